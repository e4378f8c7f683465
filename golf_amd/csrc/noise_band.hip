// Filtered-noise-band generator for gfx950.
//
// Replaces NoiseBand.forward (reference models/noise.py:114-124): every band k owns one period of a pre-filtered,
// loopable noise signal (noise_bands (K, Lb), Lb a power of two); per utterance and band a random start offset is
// drawn, and
//     out[b,t] = sum_k noise_bands[k][(t + off[b,k]) mod Lb] * up(exp(log_gain))[b,t,k]
// with the gains linearly upsampled from the frame rate (AudioTensor broadcasting).  The reference gathers a
// (B, K, T) tensor (6.3 GB at B=32, K=1024, T=48000) and multiplies it by a (B, T, K) upsampled gain tensor.
// Here one thread owns one output sample and walks the bands; the gain rows its block interpolates between are staged
// in LDS with the exp applied, consecutive lanes read consecutive noise samples (coalesced), nothing is materialised.
// Backward w.r.t. log_gain: one wave per (utterance, gain segment), lanes over the segment's samples, 32 bands at a time
// in registers, one cross-lane reduction per band chunk -- the structure of harm_bwd_kernel.
#include "common.h"
#include "device_common.h"
#include <algorithm>

namespace golf {

constexpr int NB_THREADS = 256;
constexpr int NB_CHUNK = 32;

__global__ __launch_bounds__(NB_THREADS) void noise_band_fwd_kernel(const float* __restrict__ bands, int Lb,
                                                                    const int* __restrict__ offs,
                                                                    const float* __restrict__ log_gain, int F, int hop,
                                                                    float* __restrict__ out, int64_t out_stride, int T,
                                                                    int K, int nrows, int KC) {
    // The bands are walked KC at a time (KC = K when the block's gain rows fit the LDS budget -- the GOLF configurations;
    // short gain hops, down to sample-rate gains, stage (256 / hop + 3) rows and take several passes: the reference's
    // NoiseBand works at any hop, models/noise.py:114-124)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* rows = sm;                                        // [nrows][KC] exp(log_gain) of bands k0 .. k0 + KC - 1
    int* off = reinterpret_cast<int*>(sm + (size_t)nrows * KC);  // [KC]
    const int tid = threadIdx.x, b = blockIdx.y;
    const int t_lo = blockIdx.x * NB_THREADS;
    const int row_lo = F >= 2 ? min(t_lo / hop, F - 2) : 0;
    const int nr = min(nrows, F - row_lo);
    const float* lg = log_gain + ((size_t)b * F + row_lo) * K;
    const int t = t_lo + tid;
    int f = 0;
    float w = 0.f;
    if (F >= 2) { f = min(min(t, T - 1) / hop, F - 2); w = (float)(t - f * hop) / (float)hop; }
    const unsigned mask = (unsigned)Lb - 1u;
    float acc0 = 0.f, acc1 = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int kc = min(KC, K - k0);
        if (k0) __syncthreads();                             // the previous pass's rows are still being read
        for (int e = tid; e < nr * kc; e += NB_THREADS) {
            const int r = e / kc, k = e - r * kc;
            rows[(size_t)r * KC + k] = __expf(lg[(size_t)r * K + k0 + k]);
        }
        for (int k = tid; k < kc; k += NB_THREADS) off[k] = offs[(size_t)b * K + k0 + k];
        __syncthreads();
        if (t < T) {
            const float* r0 = rows + (size_t)(f - row_lo) * KC;
            const float* r1 = F >= 2 ? r0 + KC : r0;
            const float* bk = bands + (size_t)k0 * Lb;
            for (int k = 0; k < kc; k += 2) {
                const float g0 = fmaf(w, r1[k] - r0[k], r0[k]);
                acc0 = fmaf(bk[(size_t)k * Lb + (((unsigned)t + (unsigned)off[k]) & mask)], g0, acc0);
                if (k + 1 < kc) {
                    const float g1 = fmaf(w, r1[k + 1] - r0[k + 1], r0[k + 1]);
                    acc1 = fmaf(bk[(size_t)(k + 1) * Lb + (((unsigned)t + (unsigned)off[k + 1]) & mask)], g1, acc1);
                }
            }
        }
    }
    if (t < T) out[(size_t)b * out_stride + t] = acc0 + acc1;
}

// part[b][sg][2][K]: segment sg = samples [sg*hop, (sg+1)*hop) (the last also owns the clamped tail), weight (1-w) -> row
// sg (slot 0), w -> row sg+1 (slot 1)
__global__ __launch_bounds__(64) void noise_band_bwd_kernel(const float* __restrict__ bands, int Lb,
                                                            const int* __restrict__ offs,
                                                            const float* __restrict__ g_out, int64_t g_stride, int hop,
                                                            float* __restrict__ part, int nseg, int T, int K, int F) {
    const int lane = threadIdx.x, sg = blockIdx.x, b = blockIdx.y;
    const int t_lo = sg * hop;
    const int t_hi = sg == nseg - 1 ? T : min((sg + 1) * hop, T);
    const float inv = 1.0f / (float)hop;
    const unsigned mask = (unsigned)Lb - 1u;
    float* p0 = part + (((size_t)b * nseg + sg) * 2) * K;
    float* p1 = p0 + K;
    for (int k0 = 0; k0 < K; k0 += NB_CHUNK) {
        float a0[NB_CHUNK], a1[NB_CHUNK];
        int off[NB_CHUNK];
#pragma unroll
        for (int i = 0; i < NB_CHUNK; ++i) {
            a0[i] = 0.f;
            a1[i] = 0.f;
            off[i] = k0 + i < K ? offs[(size_t)b * K + k0 + i] : 0;
        }
        for (int t = t_lo + lane; t < t_hi; t += 64) {
            const float g = g_out[(size_t)b * g_stride + t];
            const float w = F >= 2 ? (float)(t - t_lo) * inv : 0.f;
            const float gw0 = g * (1.0f - w), gw1 = g * w;
#pragma unroll
            for (int i = 0; i < NB_CHUNK; ++i) {
                const int k = k0 + i < K ? k0 + i : K - 1;
                const float v = bands[(size_t)k * Lb + (((unsigned)t + (unsigned)off[i]) & mask)];
                a0[i] = fmaf(gw0, v, a0[i]);
                a1[i] = fmaf(gw1, v, a1[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NB_CHUNK; ++i) {
            float v0 = a0[i], v1 = a1[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
            if (lane == 0 && k0 + i < K) { p0[k0 + i] = v0; p1[k0 + i] = v1; }
        }
    }
}

// g_log_gain[b][f][k] = exp(log_gain) * (part[b][f][0][k] + part[b][f-1][1][k])
__global__ void noise_band_bwd_combine_kernel(const float* __restrict__ part, const float* __restrict__ log_gain,
                                              float* __restrict__ g_lg, int B, int F, int K, int nseg) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * F * K) return;
    const int k = (int)(idx % K), f = (int)((idx / K) % F), b = (int)(idx / ((int64_t)K * F));
    float v = 0.f;
    if (f < nseg) v += part[(((size_t)b * nseg + f) * 2 + 0) * K + k];
    if (f >= 1 && f - 1 < nseg) v += part[(((size_t)b * nseg + f - 1) * 2 + 1) * K + k];
    g_lg[idx] = v * __expf(log_gain[idx]);
}

static int nb_check(const char* who, const float* bands, const int* offs, const float* lg, int B, int T, int F, int K,
                    int Lb, int hop) {
    if (!bands || !offs || !lg || B < 1 || T < 1 || F < 1 || K < 1 || hop < 1 || Lb < 1)
        return fail(GOLF_EINVAL, "%s: bad argument", who);
    if (Lb & (Lb - 1)) return fail(GOLF_EINVAL, "%s: band length %d is not a power of two", who, Lb);
    const int64_t up = F >= 2 ? (int64_t)(F - 1) * hop + 1 : T;
    if (T > up) return fail(GOLF_EINVAL, "%s: T=%d exceeds the upsampled gain length %lld", who, T, (long long)up);
    return GOLF_OK;
}

}  // namespace golf

using namespace golf;

extern "C" size_t golf_noise_band_workspace_bytes(int B, int F, int K) {
    if (B < 1 || F < 1 || K < 1) return 0;
    return align_up(sizeof(float) * (size_t)B * (F > 1 ? F - 1 : 1) * 2 * K, 256);
}

extern "C" int golf_noise_band_fwd_f32(const float* noise_bands, int Lb, const int* offsets, const float* log_gain,
                                       int F, int hop, float* out, int64_t out_stride, int B, int T, int K,
                                       void* stream) {
    if (int rc = nb_check("noise_band_fwd", noise_bands, offsets, log_gain, B, T, F, K, Lb, hop)) return rc;
    if (!out || out_stride < T) return fail(GOLF_EINVAL, "noise_band_fwd: bad output / stride");
    const int nrows = NB_THREADS / hop + 3;
    // bands per pass: all of them if their gain rows fit 60 KB of LDS, else as many (an even number) as do
    int KC = K;
    if (sizeof(float) * ((size_t)nrows + 1) * K > 60 * 1024) KC = std::max(2, (int)(60 * 1024 / sizeof(float) / (nrows + 1)) & ~1);
    const size_t lds = sizeof(float) * ((size_t)nrows + 1) * KC;
    hipLaunchKernelGGL(noise_band_fwd_kernel, dim3((unsigned)ceil_div(T, NB_THREADS), B), dim3(NB_THREADS), lds,
                       (hipStream_t)stream, noise_bands, Lb, offsets, log_gain, F, hop, out, out_stride, T, K, nrows, KC);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_noise_band_bwd_f32(const float* g_out, int64_t g_out_stride, const float* noise_bands, int Lb,
                                       const int* offsets, const float* log_gain, int F, int hop, float* g_log_gain,
                                       int B, int T, int K, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = nb_check("noise_band_bwd", noise_bands, offsets, log_gain, B, T, F, K, Lb, hop)) return rc;
    if (!g_out || !g_log_gain || g_out_stride < T) return fail(GOLF_EINVAL, "noise_band_bwd: bad pointer / stride");
    const size_t need = golf_noise_band_workspace_bytes(B, F, K);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "noise_band_bwd: workspace needs %zu bytes, 256-aligned (got %zu)", need, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const int nseg = F > 1 ? F - 1 : 1;
    // segments past the end of the signal contribute nothing: clear the partials, then fill the live ones
    hipError_t e = hipMemsetAsync(part, 0, need, st);
    if (e != hipSuccess) return fail((int)e, "noise_band_bwd: memset failed: %s", hipGetErrorString(e));
    const int live = (int)std::min<int64_t>(nseg, ceil_div(T, hop));
    hipLaunchKernelGGL(noise_band_bwd_kernel, dim3((unsigned)live, B), dim3(64), 0, st, noise_bands, Lb, offsets, g_out,
                       g_out_stride, hop, part, nseg, T, K, F);
    GOLF_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * F * K;
    hipLaunchKernelGGL(noise_band_bwd_combine_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                       (const float*)part, log_gain, g_log_gain, B, F, K, nseg);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
