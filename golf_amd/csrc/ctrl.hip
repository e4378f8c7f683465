// Frame-rate control transform of the LPC filters: logits -> reflection coefficients -> direct-form
// coefficients.  Replaces, for tensors on the GPU, the reference's
//     rc2lpc(tanh(logits) * max_abs_value)          models/filters.py:91-97, models/utils.py:581-593
// which is a Python loop of M-1 Levinson step-up iterations, each a cat / flip / mul / add on (B,F,n) tensors: ~100
// tiny dependent kernels, measured 270 us as a hipGraph and 620 us eager for the GOLF decoder's controls at B=32 --
// three to seven times the whole synthesis step they feed.  Here: one thread per frame, the polynomial in LDS
// (column per lane, conflict-free), 253 FMAs per frame; the backward re-runs the step-up to the stage it needs and
// applies the adjoint of one iteration at a time (M^3/6 FMAs per frame, still nothing).
#include "common.h"

namespace golf {

constexpr int RC_THREADS = 64;
constexpr int RC_MAX_ORDER = 64;

// poly (length n+1, poly[0] = 1) -> poly of stage n+1 in place:  new[i] = ext[i] + k * ext[n+1-i],  ext = [poly, 0]
__device__ __forceinline__ void step_up(float* p, int stride, int n, float k) {
    // pairs (i, n+1-i) are updated together so that the update is in place
    const int len = n + 2;
    p[(n + 1) * stride] = 0.f;
    for (int i = 0; i < len / 2; ++i) {
        const float u = p[i * stride], v = p[(n + 1 - i) * stride];
        p[i * stride] = fmaf(k, v, u);
        p[(n + 1 - i) * stride] = fmaf(k, u, v);
    }
    if (len & 1) {  // the middle element pairs with itself
        const int m = len / 2;
        const float u = p[m * stride];
        p[m * stride] = fmaf(k, u, u);
    }
}

__global__ __launch_bounds__(RC_THREADS) void rc2lpc_fwd_kernel(const float* __restrict__ logits, float* __restrict__ a,
                                                                int64_t N, int M, float max_abs, int apply_tanh) {
    extern __shared__ float smem[];   // [M+1][RC_THREADS]
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + lane;
    if (row >= N) return;
    float* p = smem + lane;
    const float* lg = logits + row * M;
    auto refl = [&](int i) { return apply_tanh ? tanhf(lg[i]) * max_abs : lg[i]; };
    p[0] = 1.f;
    p[RC_THREADS] = refl(0);
    for (int n = 1; n < M; ++n) step_up(p, RC_THREADS, n, refl(n));
    float* out = a + row * M;
    for (int i = 0; i < M; ++i) out[i] = p[(i + 1) * RC_THREADS];
}

// g_logits from g_a.  For n = M-1 .. 1: rebuild poly_{n} (the input of iteration n), then
//   g_k_n = sum_i g_new[i] * ext[n+1-i];   g_ext[i] = g_new[i] + k_n * g_new[n+1-i];   g_poly = g_ext[0..n]
__global__ __launch_bounds__(RC_THREADS) void rc2lpc_bwd_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ g_a,
                                                                float* __restrict__ g_logits, int64_t N, int M,
                                                                float max_abs, int apply_tanh) {
    extern __shared__ float smem[];   // poly [M+1][T], grad [M+1][T], refl [M][T]
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + lane;
    if (row >= N) return;
    float* p = smem + lane;
    float* g = smem + (size_t)(M + 1) * RC_THREADS + lane;
    float* k = smem + (size_t)2 * (M + 1) * RC_THREADS + lane;
    const float* lg = logits + row * M;
    for (int i = 0; i < M; ++i) k[i * RC_THREADS] = apply_tanh ? tanhf(lg[i]) * max_abs : lg[i];
    const float* ga = g_a + row * M;
    g[0] = 0.f;                       // poly[0] = 1 is a constant
    for (int i = 0; i < M; ++i) g[(i + 1) * RC_THREADS] = ga[i];
    float* gl = g_logits + row * M;
    for (int n = M - 1; n >= 1; --n) {
        // poly of stage n (length n+1): re-run the step-up from the start
        p[0] = 1.f;
        p[RC_THREADS] = k[0];
        for (int j = 1; j < n; ++j) step_up(p, RC_THREADS, j, k[j * RC_THREADS]);
        p[(n + 1) * RC_THREADS] = 0.f;                       // ext = [poly, 0]
        const float kn = k[n * RC_THREADS];
        float gk = 0.f;
        for (int i = 0; i <= n + 1; ++i) gk = fmaf(g[i * RC_THREADS], p[(n + 1 - i) * RC_THREADS], gk);
        // g_ext[i] = g_new[i] + kn * g_new[n+1-i], in place over pairs
        const int len = n + 2;
        for (int i = 0; i < len / 2; ++i) {
            const float u = g[i * RC_THREADS], v = g[(n + 1 - i) * RC_THREADS];
            g[i * RC_THREADS] = fmaf(kn, v, u);
            g[(n + 1 - i) * RC_THREADS] = fmaf(kn, u, v);
        }
        if (len & 1) {
            const int m = len / 2;
            const float u = g[m * RC_THREADS];
            g[m * RC_THREADS] = fmaf(kn, u, u);
        }
        // g_ext[n+1] belongs to the appended zero: dropped
        const float d = apply_tanh ? max_abs * (1.f - (kn / max_abs) * (kn / max_abs)) : 1.f;
        gl[n] = gk * d;
    }
    const float k0 = k[0];
    gl[0] = g[RC_THREADS] * (apply_tanh ? max_abs * (1.f - (k0 / max_abs) * (k0 / max_abs)) : 1.f);
}

// Register versions for orders up to 32: the polynomial lives in MP+1 registers and every loop is unrolled (orders below
// MP are padded with k = 0, which leaves the polynomial unchanged).  The LDS version above is latency-bound on its ~1000
// dependent LDS accesses per frame (14 us for 6400 frames of order 22); this one is ~300 straight-line FMAs.
template <int MP>
__device__ __forceinline__ void step_up_reg(float (&p)[MP + 1], int n, float k) {   // n compile-time after unrolling
#pragma unroll
    for (int i = 0; i < MP / 2 + 1; ++i) {
        const int j = n + 1 - i;
        if (i < j) {
            const float u = p[i], v = j <= n ? p[j] : 0.f;
            p[i] = fmaf(k, v, u);
            p[j] = fmaf(k, u, v);
        } else if (i == j) {
            p[i] = fmaf(k, p[i], p[i]);
        }
    }
}

template <int MP>
__global__ __launch_bounds__(RC_THREADS) void rc2lpc_fwd_reg_kernel(const float* __restrict__ logits,
                                                                    float* __restrict__ a, int64_t N, int M,
                                                                    float max_abs, int apply_tanh) {
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + threadIdx.x;
    if (row >= N) return;
    const float* lg = logits + row * M;
    float k[MP], p[MP + 1];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const float v = i < M ? lg[i] : 0.f;
        k[i] = apply_tanh ? tanhf(v) * max_abs : v;
    }
    p[0] = 1.f;
    p[1] = k[0];
#pragma unroll
    for (int i = 2; i <= MP; ++i) p[i] = 0.f;
#pragma unroll
    for (int n = 1; n < MP; ++n) step_up_reg<MP>(p, n, k[n]);
    float* out = a + row * M;
#pragma unroll
    for (int i = 0; i < MP; ++i)
        if (i < M) out[i] = p[i + 1];
}

template <int MP>
__global__ __launch_bounds__(RC_THREADS) void rc2lpc_bwd_reg_kernel(const float* __restrict__ logits,
                                                                    const float* __restrict__ g_a,
                                                                    float* __restrict__ g_logits, int64_t N, int M,
                                                                    float max_abs, int apply_tanh) {
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + threadIdx.x;
    if (row >= N) return;
    const float* lg = logits + row * M;
    const float* ga = g_a + row * M;
    float k[MP], g[MP + 1], gk[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const float v = i < M ? lg[i] : 0.f;
        k[i] = apply_tanh ? tanhf(v) * max_abs : v;
        g[i + 1] = i < M ? ga[i] : 0.f;
    }
    g[0] = 0.f;
#pragma unroll
    for (int n = MP - 1; n >= 1; --n) {
        float p[MP + 1];   // poly of stage n, rebuilt
        p[0] = 1.f;
        p[1] = k[0];
#pragma unroll
        for (int i = 2; i <= MP; ++i) p[i] = 0.f;
#pragma unroll
        for (int j = 1; j < n; ++j) step_up_reg<MP>(p, j, k[j]);
        float acc = 0.f;
#pragma unroll
        for (int i = 1; i <= n + 1; ++i) acc = fmaf(g[i], p[n + 1 - i], acc);   // ext[n+1] = 0 kills the i = 0 term
        gk[n] = acc;
        const float kn = k[n];
#pragma unroll
        for (int i = 0; i < MP / 2 + 1; ++i) {
            const int j = n + 1 - i;
            if (i < j) {
                const float u = g[i], v = g[j];
                g[i] = fmaf(kn, v, u);
                g[j] = fmaf(kn, u, v);
            } else if (i == j) {
                g[i] = fmaf(kn, g[i], g[i]);
            }
        }
    }
    gk[0] = g[1];
    float* gl = g_logits + row * M;
#pragma unroll
    for (int i = 0; i < MP; ++i)
        if (i < M) {
            const float t = k[i] / max_abs;
            gl[i] = gk[i] * (apply_tanh ? max_abs * (1.f - t * t) : 1.f);
        }
}

// ------------------------------------------------------------------------------------------------------------------
// Biquad parameterisations of the ISMIR'23 configs ("coef", "conj", "real"; models/utils.py:487-525) followed by the
// product of the K second-order sections into direct form (biquads2lpc / coeff_product, models/utils.py:444-484: a tree
// of grouped conv1d calls).  In PyTorch ops that is 112 kernels, 630 us eager / 243 us as a hipGraph for ONE filter at
// B=32 -- and the ISMIR GOLF decoder has two.  One thread per frame, polynomial in LDS.
//   rep 0 "coef": a1 = 2 rho tanh(l0);            a2 = ((2 - |a1|) rho tanh(l1) + |a1|) / 2
//   rep 1 "conj": r = rho sigmoid(l0);            a1 = -2 r tanh(l1);   a2 = r^2
//   rep 2 "real": z = rho tanh(l);                a1 = -(z0 + z1);      a2 = z0 z1
struct Sos { float a1, a2; };
__device__ __forceinline__ Sos sos_from_logits(float l0, float l1, float rho, int rep) {
    Sos s;
    if (rep == 0) {
        s.a1 = 2.f * rho * tanhf(l0);
        const float m = fabsf(s.a1);
        s.a2 = 0.5f * ((2.f - m) * tanhf(l1) * rho + m);
    } else if (rep == 1) {
        const float r = rho / (1.f + expf(-l0));
        s.a1 = -2.f * r * tanhf(l1);
        s.a2 = r * r;
    } else {
        const float z0 = rho * tanhf(l0), z1 = rho * tanhf(l1);
        s.a1 = -(z0 + z1);
        s.a2 = z0 * z1;
    }
    return s;
}
// (g_a1, g_a2) -> (g_l0, g_l1)
__device__ __forceinline__ void sos_backward(float l0, float l1, float rho, int rep, float g1, float g2, float& gl0,
                                             float& gl1) {
    if (rep == 0) {
        const float t0 = tanhf(l0), t1 = tanhf(l1);
        const float a1 = 2.f * rho * t0, m = fabsf(a1);
        const float sgn = a1 > 0.f ? 1.f : (a1 < 0.f ? -1.f : 0.f);
        const float da2_da1 = 0.5f * sgn * (1.f - t1 * rho);
        gl0 = (g1 + g2 * da2_da1) * 2.f * rho * (1.f - t0 * t0);
        gl1 = g2 * 0.5f * (2.f - m) * rho * (1.f - t1 * t1);
    } else if (rep == 1) {
        const float sg = 1.f / (1.f + expf(-l0)), r = rho * sg, t1 = tanhf(l1);
        const float dr = rho * sg * (1.f - sg);
        gl0 = (g1 * (-2.f * t1) + g2 * 2.f * r) * dr;
        gl1 = g1 * (-2.f * r) * (1.f - t1 * t1);
    } else {
        const float t0 = tanhf(l0), t1 = tanhf(l1);
        const float z0 = rho * t0, z1 = rho * t1;
        gl0 = (-g1 + g2 * z1) * rho * (1.f - t0 * t0);
        gl1 = (-g1 + g2 * z0) * rho * (1.f - t1 * t1);
    }
}
// poly (length n+1) times (1 + a1 z^-1 + a2 z^-2), in place, length n+3
__device__ __forceinline__ void mul_sos(float* p, int stride, int n, Sos s) {
    p[(n + 1) * stride] = 0.f;
    p[(n + 2) * stride] = 0.f;
    for (int i = n + 2; i >= 1; --i) {
        const float pm1 = p[(i - 1) * stride], pm2 = i >= 2 ? p[(i - 2) * stride] : 0.f;
        p[i * stride] = fmaf(s.a2, pm2, fmaf(s.a1, pm1, p[i * stride]));
    }
}

__global__ __launch_bounds__(RC_THREADS) void sos2lpc_fwd_kernel(const float* __restrict__ logits, float* __restrict__ a,
                                                                 int64_t N, int K, float rho, int rep) {
    extern __shared__ float smem[];   // [2K+1][RC_THREADS]
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + lane;
    if (row >= N) return;
    float* p = smem + lane;
    const float* lg = logits + row * 2 * K;
    p[0] = 1.f;
    for (int k = 0; k < K; ++k) mul_sos(p, RC_THREADS, 2 * k, sos_from_logits(lg[2 * k], lg[2 * k + 1], rho, rep));
    float* out = a + row * 2 * K;
    for (int i = 0; i < 2 * K; ++i) out[i] = p[(i + 1) * RC_THREADS];
}

// Sections are applied in index order, so going backwards: rebuild the product of sections 0..k-1, take the two
// correlations for (g_a1, g_a2), pull the gradient back through section k.
__global__ __launch_bounds__(RC_THREADS) void sos2lpc_bwd_kernel(const float* __restrict__ logits,
                                                                 const float* __restrict__ g_a,
                                                                 float* __restrict__ g_logits, int64_t N, int K,
                                                                 float rho, int rep) {
    extern __shared__ float smem[];   // poly [2K+1][T], grad [2K+3][T]
    const int lane = threadIdx.x;
    const int64_t row = (int64_t)blockIdx.x * RC_THREADS + lane;
    if (row >= N) return;
    const int M = 2 * K;
    float* p = smem + lane;
    float* g = smem + (size_t)(M + 1) * RC_THREADS + lane;
    const float* lg = logits + row * M;
    const float* ga = g_a + row * M;
    g[0] = 0.f;
    for (int i = 0; i < M; ++i) g[(i + 1) * RC_THREADS] = ga[i];
    g[(M + 1) * RC_THREADS] = 0.f;
    g[(M + 2) * RC_THREADS] = 0.f;
    float* gl = g_logits + row * M;
    for (int k = K - 1; k >= 0; --k) {
        p[0] = 1.f;
        for (int j = 0; j < k; ++j) mul_sos(p, RC_THREADS, 2 * j, sos_from_logits(lg[2 * j], lg[2 * j + 1], rho, rep));
        const int n = 2 * k;                       // degree of the product so far; new poly has degree n + 2
        const Sos s = sos_from_logits(lg[2 * k], lg[2 * k + 1], rho, rep);
        float g1 = 0.f, g2 = 0.f;
        for (int i = 1; i <= n + 2; ++i) {
            if (i - 1 <= n) g1 = fmaf(g[i * RC_THREADS], p[(i - 1) * RC_THREADS], g1);
            if (i >= 2) g2 = fmaf(g[i * RC_THREADS], p[(i - 2) * RC_THREADS], g2);
        }
        // g_prev[i] = g_new[i] + a1 g_new[i+1] + a2 g_new[i+2], ascending in place (reads run ahead of writes)
        for (int i = 0; i <= n; ++i)
            g[i * RC_THREADS] = fmaf(s.a2, g[(i + 2) * RC_THREADS], fmaf(s.a1, g[(i + 1) * RC_THREADS], g[i * RC_THREADS]));
        float gl0, gl1;
        sos_backward(lg[2 * k], lg[2 * k + 1], rho, rep, g1, g2, gl0, gl1);
        gl[2 * k] = gl0;
        gl[2 * k + 1] = gl1;
    }
}

static int rc_check(const char* who, const void* x, const void* y, int64_t N, int M, float max_abs) {
    if (!x || !y) return fail(GOLF_EINVAL, "%s: null pointer", who);
    if (N < 1 || M < 1 || M > RC_MAX_ORDER) return fail(GOLF_EINVAL, "%s: bad size (N=%lld, order %d, max %d)", who, (long long)N, M, RC_MAX_ORDER);
    if (!(max_abs > 0.f)) return fail(GOLF_EINVAL, "%s: max_abs must be positive", who);
    return GOLF_OK;
}

}  // namespace golf

using namespace golf;

extern "C" int golf_rc2lpc_fwd_f32(const float* logits, float* a, int64_t N, int M, float max_abs, int apply_tanh,
                                   void* stream) {
    if (int rc = rc_check("rc2lpc_fwd", logits, a, N, M, max_abs)) return rc;
    const dim3 grid((unsigned)ceil_div(N, RC_THREADS)), block(RC_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (M <= 8)
        hipLaunchKernelGGL(rc2lpc_fwd_reg_kernel<8>, grid, block, 0, st, logits, a, N, M, max_abs, apply_tanh);
    else if (M <= 16)
        hipLaunchKernelGGL(rc2lpc_fwd_reg_kernel<16>, grid, block, 0, st, logits, a, N, M, max_abs, apply_tanh);
    else if (M <= 24)
        hipLaunchKernelGGL(rc2lpc_fwd_reg_kernel<24>, grid, block, 0, st, logits, a, N, M, max_abs, apply_tanh);
    else if (M <= 32)
        hipLaunchKernelGGL(rc2lpc_fwd_reg_kernel<32>, grid, block, 0, st, logits, a, N, M, max_abs, apply_tanh);
    else
        hipLaunchKernelGGL(rc2lpc_fwd_kernel, grid, block, sizeof(float) * (size_t)(M + 1) * RC_THREADS, st, logits, a,
                           N, M, max_abs, apply_tanh);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_rc2lpc_bwd_f32(const float* logits, const float* g_a, float* g_logits, int64_t N, int M,
                                   float max_abs, int apply_tanh, void* stream) {
    if (int rc = rc_check("rc2lpc_bwd", logits, g_a, N, M, max_abs)) return rc;
    if (!g_logits) return fail(GOLF_EINVAL, "rc2lpc_bwd: null pointer");
    const dim3 grid((unsigned)ceil_div(N, RC_THREADS)), block(RC_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (M <= 8)
        hipLaunchKernelGGL(rc2lpc_bwd_reg_kernel<8>, grid, block, 0, st, logits, g_a, g_logits, N, M, max_abs, apply_tanh);
    else if (M <= 16)
        hipLaunchKernelGGL(rc2lpc_bwd_reg_kernel<16>, grid, block, 0, st, logits, g_a, g_logits, N, M, max_abs, apply_tanh);
    else if (M <= 24)
        hipLaunchKernelGGL(rc2lpc_bwd_reg_kernel<24>, grid, block, 0, st, logits, g_a, g_logits, N, M, max_abs, apply_tanh);
    else
        hipLaunchKernelGGL(rc2lpc_bwd_kernel, grid, block, sizeof(float) * (size_t)(3 * M + 2) * RC_THREADS, st, logits,
                           g_a, g_logits, N, M, max_abs, apply_tanh);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_sos2lpc_fwd_f32(const float* logits, float* a, int64_t N, int K, float max_abs_pole, int rep,
                                    void* stream) {
    if (!logits || !a) return fail(GOLF_EINVAL, "sos2lpc_fwd: null pointer");
    if (N < 1 || K < 1 || 2 * K > RC_MAX_ORDER || rep < 0 || rep > 2 || !(max_abs_pole > 0.f))
        return fail(GOLF_EINVAL, "sos2lpc_fwd: bad size / parameterisation (N=%lld K=%d rep=%d)", (long long)N, K, rep);
    hipLaunchKernelGGL(sos2lpc_fwd_kernel, dim3((unsigned)ceil_div(N, RC_THREADS)), dim3(RC_THREADS),
                       sizeof(float) * (size_t)(2 * K + 1) * RC_THREADS, (hipStream_t)stream, logits, a, N, K,
                       max_abs_pole, rep);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_sos2lpc_bwd_f32(const float* logits, const float* g_a, float* g_logits, int64_t N, int K,
                                    float max_abs_pole, int rep, void* stream) {
    if (!logits || !g_a || !g_logits) return fail(GOLF_EINVAL, "sos2lpc_bwd: null pointer");
    if (N < 1 || K < 1 || 2 * K > RC_MAX_ORDER || rep < 0 || rep > 2 || !(max_abs_pole > 0.f))
        return fail(GOLF_EINVAL, "sos2lpc_bwd: bad size / parameterisation (N=%lld K=%d rep=%d)", (long long)N, K, rep);
    hipLaunchKernelGGL(sos2lpc_bwd_kernel, dim3((unsigned)ceil_div(N, RC_THREADS)), dim3(RC_THREADS),
                       sizeof(float) * (size_t)(4 * K + 4) * RC_THREADS, (hipStream_t)stream, logits, g_a, g_logits, N, K,
                       max_abs_pole, rep);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
