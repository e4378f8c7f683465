// Zero-phase FIR noise filter of the GOLF decoders for gfx950.
//
// Reference: LTVZeroPhaseFIRFilter, models/filters.py:286-384
//   kernel[b,f,:] = fftshift(irfft(exp(log_mag[b,f,:]))) * window                (filters.py:294-306)
//   y[b, f*hop+n] = sum_k pad(ex)[b, f*hop+n+k] * kernel[b,f,k]                  (filters.py:355-383: unfold +
//                                                                                  grouped F.conv1d, one group per frame)
// Two different shapes of work, two different engines:
//   * the inverse real FFT of a zero-phase spectrum is a cosine transform, i.e. a dense contraction
//       u[g,d] = sum_k exp(log_mag[g,k]) * basis[k,d],   basis[k,d] = c_k cos(2 pi k d / N) / N,
//     (G = B*F = 6400 rows, K = D = n_mag = 256): that one goes to the matrix cores with the exact-fp32
//     v_mfma_f32_16x16x4_f32 (bf16 would cost 1e-3 of accuracy; the parity bar is 1e-4).  exp() is fused into the
//     A-operand staging, the window / fftshift / mirror (the kernel is symmetric about tap N/2) into the epilogue.
//   * the per-frame FIR has a different kernel in every group and no reuse across groups: matrix-vector work,
//     done on the VALU.  A wave owns one frame; the frame's taps are wave-uniform, so they arrive as SGPRs
//     (s_load_dwordx16) and every multiply-add is a v_pk_fma_f32 with a scalar operand; the signal window slides
//     through registers fed by one ds_read_b128 per 4 taps.  Packed math needs 8-byte aligned register pairs, which
//     a sliding window only offers for every other tap; the odd taps therefore accumulate into a second family of
//     accumulators that belongs to the outputs shifted by one sample ("E/O split"), so that they see the same aligned
//     pairs, and the two families are merged once at the end with a single cross-lane shift.
//   The same routine serves the backward pass: with the roles swapped (taps = the incoming gradient of the frame,
//   signal = the padded excitation, resp. the reversed kernel row) it yields g_kernel and g_ex.
#include "common.h"
#include "device_common.h"

namespace golf {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// Packed FIR core: out[o] = sum_{n < ntaps} coef[n] * sig[o + n],  o = 4*lane + j, j = 0..3   (ntaps % 4 == 0)
// sig: 16-byte aligned LDS region of >= 256 + ntaps + 4 floats owned by this wave; coef: wave-uniform global pointer.
// Lane 63 only contributes its shifted accumulator: a pass yields FIR_TILE = 252 valid outputs.
// ------------------------------------------------------------------------------------------------------------
constexpr int FIR_TILE = 252;
constexpr int FIR_WAVES = 4;

struct FirAcc {
    f32x2 e01, e23;  // even taps -> outputs 4l .. 4l+3
    f32x2 o01, o23;  // odd taps  -> outputs 4l-1 .. 4l+2
};

__device__ __forceinline__ void fir_zero(FirAcc& A) {
    A.e01 = A.e23 = A.o01 = A.o23 = (f32x2){0.f, 0.f};
}

// one group of 4 taps on the aligned window x0..x5 (x = sig[4(l+q) ...])
__device__ __forceinline__ void fir_group(FirAcc& A, const f32x4 cur, const f32x2 x45, float c0, float c1, float c2,
                                          float c3) {
    const f32x2 x01 = {cur.x, cur.y}, x23 = {cur.z, cur.w};
    A.e01 = __builtin_elementwise_fma((f32x2){c0, c0}, x01, A.e01);
    A.e23 = __builtin_elementwise_fma((f32x2){c0, c0}, x23, A.e23);
    A.o01 = __builtin_elementwise_fma((f32x2){c1, c1}, x01, A.o01);
    A.o23 = __builtin_elementwise_fma((f32x2){c1, c1}, x23, A.o23);
    A.e01 = __builtin_elementwise_fma((f32x2){c2, c2}, x23, A.e01);
    A.e23 = __builtin_elementwise_fma((f32x2){c2, c2}, x45, A.e23);
    A.o01 = __builtin_elementwise_fma((f32x2){c3, c3}, x23, A.o01);
    A.o23 = __builtin_elementwise_fma((f32x2){c3, c3}, x45, A.o23);
}

// The tap loop is left to the compiler (it unrolls 8 groups: one s_load_dwordx16 x2 + 8 ds_read_b128 per 64 packed
// FMAs, then waits for all of them).  That wait is exposed in one wave, and is hidden by the other waves of the SIMD:
// the loop needs 44 VGPRs, so all ~6 waves per SIMD of the B=32 problem are resident at once.  A hand-pipelined
// version (taps and window of the next 32 taps prefetched during the FMAs of the current ones; 152 VGPRs, 3 waves per
// SIMD) measured no faster (22.2 vs 22.4 us): what cost the time was the staging loop below, not this one.
template <bool REV = false>   // REV: the taps applied in reverse order (coef[ntaps-1-k] in place of coef[k]: the adjoint of an LTI FIR)
__device__ __forceinline__ void fir_accum(FirAcc& A, const float* sig, const float* __restrict__ coef, int ntaps,
                                          int lane) {
    const f32x4* sig4 = reinterpret_cast<const f32x4*>(sig);
    f32x4 cur = sig4[lane];
    const int nq = ntaps >> 2;
    for (int q = 0; q < nq; ++q) {
        const f32x4 nxt = sig4[lane + q + 1];
        if (REV) {
            const float* c = coef + ntaps - 4 - 4 * q;
            fir_group(A, cur, (f32x2){nxt.x, nxt.y}, c[3], c[2], c[1], c[0]);
        } else {
            fir_group(A, cur, (f32x2){nxt.x, nxt.y}, coef[4 * q], coef[4 * q + 1], coef[4 * q + 2], coef[4 * q + 3]);
        }
        cur = nxt;
    }
}

// Stage `span` floats src[base + i] into the wave's LDS region, FIR_STAGE_U loads per lane in flight at a time.
// The region must hold fir_region(span) floats: loads past `span` are harmless (the descriptor bounds them) and are
// stored too, so that the loop has no branch — a per-element guard made hipcc wait for every single load (13 serial
// HBM round trips per wave: 26 us for a kernel whose FMAs need 13).
constexpr int FIR_STAGE_U = 8;
__host__ __device__ constexpr int fir_region(int span) {
    return (span + 64 * FIR_STAGE_U - 1) / (64 * FIR_STAGE_U) * (64 * FIR_STAGE_U);
}
template <bool REVERSE>
__device__ __forceinline__ void fir_stage(float* sig, const BufRow& src, int base, int span, int lane) {
    for (int i0 = 0; i0 < span; i0 += 64 * FIR_STAGE_U) {
        float v[FIR_STAGE_U];
#pragma unroll
        for (int u = 0; u < FIR_STAGE_U; ++u) {
            const int i = i0 + u * 64 + lane;
            const int idx = REVERSE ? base - i : base + i;
            // negative indices are masked here, not by the descriptor: hipcc folds the unrolled `u * 64` into the
            // instruction's immediate offset, and (negative voffset) + (positive immediate) is range-checked WITHOUT
            // wrapping, so elements that are in range only after the addition came back as 0.  max() keeps voffset
            // non-negative (and un-foldable); indices past the end are still dropped by the hardware.
            const float x = src.ld(max(idx, 0));
            v[u] = idx >= 0 ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < FIR_STAGE_U; ++u) sig[i0 + u * 64 + lane] = v[u];
    }
}

// merge the families: out[4l+j] = E_j + O_{j+1}; the last one needs O_0 of the next lane
__device__ __forceinline__ f32x4 fir_finish(const FirAcc& A) {
    const float onext = __shfl_down(A.o01.x, 1);
    return (f32x4){A.e01.x + A.o01.y, A.e01.y + A.o23.x, A.e23.x + A.o23.y, A.e23.y + onext};
}

// ------------------------------------------------------------------------------------------------------------
// forward: unit = (utterance b, frame f, pass c over the frame's hop outputs)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FIR_WAVES) void fir_frames_fwd_kernel(
    const float* __restrict__ ex, int64_t ex_stride, const float* __restrict__ kern, int KS, float* __restrict__ y,
    int64_t y_stride, int B, int T, int nfr, int F, int N, int hop, int npass, int RS, int frame0) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * nfr * npass) return;
    const int c = unit % npass, f = (unit / npass) % nfr, b = unit / (npass * nfr);
    float* sig = fir_lds + wv * RS;
    const int P = (N - 1) >> 1;
    const int ntaps = (N + 3) & ~3;  // kernel rows are zero padded to KS >= ntaps
    const int span = 256 + ntaps + 4;
    const int t0 = f * hop + c * FIR_TILE;  // first output of this pass
    const BufRow xr(ex + b * ex_stride, T);
    fir_stage<false>(sig, xr, t0 - P, span, lane);
    wave_lds_fence();
    FirAcc A;
    fir_zero(A);
    fir_accum(A, sig, kern + (size_t)(b * F + f + frame0) * KS, ntaps, lane);  // output frame f uses kernel row f + frame0
    const f32x4 r = fir_finish(A);
    const BufRow yr(y + b * y_stride, nfr * hop);
    const int o = 4 * lane;
    const int lim = min(FIR_TILE, hop - c * FIR_TILE);
    yr.st(o + 0 < lim ? t0 + o + 0 : -1, r.x);
    yr.st(o + 1 < lim ? t0 + o + 1 : -1, r.y);
    yr.st(o + 2 < lim ? t0 + o + 2 : -1, r.z);
    yr.st(o + 3 < lim ? t0 + o + 3 : -1, r.w);
}

// ------------------------------------------------------------------------------------------------------------
// backward w.r.t. the kernels: g_kern[b,f,k] = sum_{n<hop} gy[b,f*hop+n] * pad(ex)[b, f*hop+n+k]
// unit = (b, f, pass c over the N taps); taps of the core = the frame's hop gradient samples
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FIR_WAVES) void fir_frames_bwd_kern_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ ex, int64_t ex_stride,
    float* __restrict__ g_kern, int KS, int B, int T, int nfr, int F, int N, int hop, int npass, int RS, int frame0) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * F * npass) return;
    const int c = unit % npass, fr = (unit / npass) % F, b = unit / (npass * F);
    const int f = fr - frame0;  // output frame that used kernel row fr (rows outside [frame0, frame0+nfr) get zeros)
    const BufRow gr(g_kern + (size_t)(b * F + fr) * KS, KS);
    const int k0 = c * FIR_TILE, o = 4 * lane;
    const int lim = min(FIR_TILE, N - k0);
    const int limz = min(FIR_TILE, KS - k0);   // the row's padding taps [N, KS) (last pass) get their zero gradient here:
                                               // the caller used to fill them with a strided torch kernel of its own
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (f >= 0 && f < nfr && k0 < N) {  // wave-uniform (a pass that holds padding taps only has nothing to correlate)
        float* sig = fir_lds + wv * RS;
        const int P = (N - 1) >> 1;
        const int span = 256 + hop + 4;
        const BufRow xr(ex + b * ex_stride, T);
        fir_stage<false>(sig, xr, f * hop + k0 - P, span, lane);
        wave_lds_fence();
        FirAcc A;
        fir_zero(A);
        fir_accum(A, sig, gy + b * gy_stride + (size_t)f * hop, hop, lane);
        r = fir_finish(A);
    }
    gr.st(o + 0 < limz ? k0 + o + 0 : -1, o + 0 < lim ? r.x : 0.f);
    gr.st(o + 1 < limz ? k0 + o + 1 : -1, o + 1 < lim ? r.y : 0.f);
    gr.st(o + 2 < limz ? k0 + o + 2 : -1, o + 2 < lim ? r.z : 0.f);
    gr.st(o + 3 < limz ? k0 + o + 3 : -1, o + 3 < lim ? r.w : 0.f);
}

// ------------------------------------------------------------------------------------------------------------
// backward w.r.t. the excitation, in padded coordinates m = t + P:
//   g_xp[m] = sum_f sum_{n<hop} gy[f*hop+n] * kernel[f][m - f*hop - n]
// unit = (b, tile of TILE consecutive m); outputs are taken in DEscending m so that the signal index ascends with
// the tap index: sig_f[i] = kernel[f][m_hi - f*hop - i] (0 outside the kernel), taps = gy of frame f; the frames
// whose support reaches the tile are accumulated one after the other.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FIR_WAVES) void fir_frames_bwd_ex_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ kern, int KS,
    float* __restrict__ g_ex, int64_t g_ex_stride, int B, int T, int nfr, int F, int N, int hop, int TILE,
    int tile_lo, int ntile, int RS, int frame0) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * ntile) return;
    const int tile = tile_lo + unit % ntile, b = unit / ntile;
    float* sig = fir_lds + wv * RS;
    const int P = (N - 1) >> 1;
    const int m_lo = tile * TILE, m_hi = m_lo + TILE - 1;
    int f_lo = m_lo - (N - 1) - (hop - 1);  // f*hop >= this
    f_lo = f_lo <= 0 ? 0 : (f_lo + hop - 1) / hop;
    const int f_hi = min(nfr - 1, m_hi / hop);
    const int span = 256 + hop + 4;
    FirAcc A;
    fir_zero(A);
    for (int f = f_lo; f <= f_hi; ++f) {
        const BufRow kr(kern + (size_t)(b * F + f + frame0) * KS, N);
        wave_lds_fence();
        fir_stage<true>(sig, kr, m_hi - f * hop, span, lane);
        wave_lds_fence();
        fir_accum(A, sig, gy + b * gy_stride + (size_t)f * hop, hop, lane);
    }
    const f32x4 r = fir_finish(A);
    const BufRow gr(g_ex + b * g_ex_stride, T);
    const int o = 4 * lane;
    const int t_hi = m_hi - P;  // output o <-> t = t_hi - o; hardware drops t outside [0,T)
    gr.st(o + 0 < TILE ? t_hi - o - 0 : -1, r.x);
    gr.st(o + 1 < TILE ? t_hi - o - 1 : -1, r.y);
    gr.st(o + 2 < TILE ? t_hi - o - 2 : -1, r.z);
    gr.st(o + 3 < TILE ? t_hi - o - 3 : -1, r.w);
}

// ------------------------------------------------------------------------------------------------------------
// LTI FIR shared by the whole batch (room filter; reference LTIAcousticFilter,
// models/filters.py:426-449):  y[b,t] = sum_{n<ntaps} taps[n] * ex[b, t - lead + n],  zero outside [0,T).
// unit = (b, tile of FIR_TILE outputs).  The same kernel with flipped taps and lead' = ntaps-1-lead is its adjoint.
// ------------------------------------------------------------------------------------------------------------
template <bool REV>
__global__ __launch_bounds__(64 * FIR_WAVES) void lti_fir_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                                 const float* __restrict__ taps, int ntaps, int lead,
                                                                 float* __restrict__ y, int64_t y_stride, int B, int T,
                                                                 int ntile, int RS) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * ntile) return;
    const int tile = unit % ntile, b = unit / ntile;
    float* sig = fir_lds + wv * RS;
    const int t0 = tile * FIR_TILE;
    const int span = 256 + ntaps + 4;
    const BufRow xr(ex + b * ex_stride, T);
    fir_stage<false>(sig, xr, t0 - lead, span, lane);
    wave_lds_fence();
    FirAcc A;
    fir_zero(A);
    fir_accum<REV>(A, sig, taps, ntaps, lane);
    const f32x4 r = fir_finish(A);
    const BufRow yr(y + b * y_stride, T);
    const int o = 4 * lane;
    yr.st(o + 0 < FIR_TILE ? t0 + o + 0 : -1, r.x);
    yr.st(o + 1 < FIR_TILE ? t0 + o + 1 : -1, r.y);
    yr.st(o + 2 < FIR_TILE ? t0 + o + 2 : -1, r.z);
    yr.st(o + 3 < FIR_TILE ? t0 + o + 3 : -1, r.w);
}

// gradient w.r.t. the shared taps: g_taps[n] = sum_{b,t} gy[b,t] * ex[b, t - lead + n].
// unit = (b, stretch of LT samples, pass over the taps): the stretch's gradient samples are the packed taps of the
// core, the outputs are the filter taps; per-unit partial sums, then a deterministic reduction.
constexpr int LTI_GRAD_LT = 960;

__global__ __launch_bounds__(64 * FIR_WAVES) void lti_fir_taps_grad_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ ex, int64_t ex_stride,
    float* __restrict__ part, int ntaps, int lead, int B, int T, int nstretch, int npass, int RS) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * nstretch * npass) return;
    const int c = unit % npass, st = (unit / npass) % nstretch, b = unit / (npass * nstretch);
    float* sig = fir_lds + wv * RS;
    const int t0 = st * LTI_GRAD_LT, n0 = c * FIR_TILE;
    const int len = min(LTI_GRAD_LT, T - t0), len4 = len & ~3;
    const int span = 256 + LTI_GRAD_LT + 4;
    const BufRow xr(ex + b * ex_stride, T);
    fir_stage<false>(sig, xr, t0 - lead + n0, span, lane);
    wave_lds_fence();
    FirAcc A;
    fir_zero(A);
    const float* g = gy + b * gy_stride + t0;
    fir_accum(A, sig, g, len4, lane);
    f32x4 r = fir_finish(A);
    const int o = 4 * lane;
    for (int n = len4; n < len; ++n) {  // ragged tail of the last stretch (< 4 samples)
        const float gv = g[n];
        r.x += gv * sig[o + n];
        r.y += gv * sig[o + 1 + n];
        r.z += gv * sig[o + 2 + n];
        r.w += gv * sig[o + 3 + n];
    }
    const BufRow pr(part + (size_t)(b * nstretch + st) * ntaps, ntaps);
    const int lim = min(FIR_TILE, ntaps - n0);
    pr.st(o + 0 < lim ? n0 + o + 0 : -1, r.x);
    pr.st(o + 1 < lim ? n0 + o + 1 : -1, r.y);
    pr.st(o + 2 < lim ? n0 + o + 2 : -1, r.z);
    pr.st(o + 3 < lim ? n0 + o + 3 : -1, r.w);
}

__global__ __launch_bounds__(256) void lti_fir_taps_reduce_kernel(const float* __restrict__ part,
                                                                  float* __restrict__ g_taps, int ntaps, int nrows) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < nrows; r += 256) s += part[(size_t)r * ntaps + n];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) g_taps[n] = red[0];
}

// ------------------------------------------------------------------------------------------------------------
// cosine basis (both orientations), fp64 evaluation with exact integer argument reduction, rounded once to fp32
//   bas [k*Pd + d] = c_k cos(2 pi k d / N) / N   (k, d < n_mag; 0 in the padding),  basT[d*Pd + k] = same
// ------------------------------------------------------------------------------------------------------------
__global__ void zp_basis_kernel(float* __restrict__ bas, float* __restrict__ basT, int n_mag, int Pd) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (d >= Pd) return;
    const int N = 2 * (n_mag - 1);
    float v = 0.f;
    if (k < n_mag && d < n_mag) {
        const long long r = ((long long)k * d) % N;
        const double ck = (k == 0 || k == n_mag - 1) ? 1.0 : 2.0;
        v = (float)(ck * cospi(2.0 * (double)r / (double)N) / (double)N);
    }
    bas[(size_t)k * Pd + d] = v;
    basT[(size_t)d * Pd + k] = v;
}

// ------------------------------------------------------------------------------------------------------------
// exact-fp32 MFMA contraction  C[g, c] = sum_k A[g, k] * Bm[k, c]   (Bm: Pd x Pd, zero padded)
//   MODE 0 (forward):  A = exp(log_mag), Bm = bas;  epilogue: kernel row g gets u*window at taps N/2 +- c
//   MODE 1 (backward): A[g, d] = window-folded g_kernel, Bm = basT; epilogue: g_log_mag = C * exp(log_mag)
// block = 4 waves, 32 rows x 128 columns; wave w owns all 32 rows x 32 columns = 2 x 2 tiles of 16x16.  (With 64-row
// blocks the 200 workgroups ran one wave per SIMD and staging, MFMAs and epilogue simply added up: 5.8 + 7.4 + 2.7 +
// 1.7 + 2.2 us measured piecewise, tools/ubench/gemm_parts.hip; 400 smaller workgroups let two of them share a CU.)
// ------------------------------------------------------------------------------------------------------------
constexpr int ZG_ROWS = 32, ZG_COLS = 128, ZG_KC = 64;
constexpr int ZG_RT = ZG_ROWS / 16;  // 16-row MFMA tiles per wave
constexpr int ZG_RPW = ZG_ROWS / 4;  // epilogue rows per wave
constexpr int ZG_LDA = ZG_KC + 2;  // row stride = 2 mod 32: the (row, k) pattern of the A operand hits 32 distinct banks
constexpr int ZG_LDC = ZG_COLS + 1;
constexpr int ZG_EPT = ZG_ROWS * ZG_KC / 256;  // A elements staged per thread per chunk

// A-chunk staging, split in two so that the loads of chunk c+1 fly during the MFMAs of chunk c:
// zg_fetch issues the (bounds-checked, branch-free) loads, zg_commit applies the fused transform and writes LDS.
template <int MODE>
struct ZgStage {
    float x[ZG_EPT], x2[ZG_EPT];
    float wa, wb;
    bool kin;
};
template <int MODE>
__device__ __forceinline__ void zg_fetch(ZgStage<MODE>& st, const float* __restrict__ src, int src_stride,
                                         const float* __restrict__ window, int g0, int G, int n_mag, int kc0, int tid) {
    const int N = 2 * (n_mag - 1), H = N >> 1;
    // one column per thread, rows r0, r0+4, ...; r0 = wave index: provably uniform, so the row descriptors below
    // stay in SGPRs without a waterfall loop
    const int k = tid & (ZG_KC - 1), r0 = __builtin_amdgcn_readfirstlane(tid / ZG_KC);
    const int kk = kc0 + k;
    st.kin = kk < n_mag;
    int ia = -1, ib = -1;  // element offsets inside a row; -1 (out of range for the descriptor) reads 0
    st.wa = st.wb = 0.f;
    if (MODE == 0) {
        ia = st.kin ? kk : -1;
    } else {  // adjoint of (mirror + window): taps H+d (d < H) and H-d (d >= 1) both came from u[d]
        const BufRow wr(window, N);
        if (st.kin && kk < H) ia = H + kk;
        if (st.kin && kk >= 1) ib = H - kk;
        st.wa = wr.ld(ia);
        st.wb = wr.ld(ib);
    }
    const int nrow = min(ZG_ROWS, G - g0);
#pragma unroll
    for (int j = 0; j < ZG_EPT; ++j) {
        const int r = r0 + j * (256 / ZG_KC);
        // one descriptor per row (wave-uniform): rows past G are empty descriptors
        const BufRow row(src + (size_t)min(g0 + r, G - 1) * src_stride, r < nrow ? src_stride : 0);
        st.x[j] = row.ld(ia);
        if (MODE == 1) st.x2[j] = row.ld(ib);
    }
}
template <int MODE>
__device__ __forceinline__ void zg_commit(const ZgStage<MODE>& st, float* As, int tid) {
    const int k = tid & (ZG_KC - 1), r0 = __builtin_amdgcn_readfirstlane(tid / ZG_KC);
#pragma unroll
    for (int j = 0; j < ZG_EPT; ++j) {
        const int r = r0 + j * (256 / ZG_KC);
        float v;
        if (MODE == 0) {
            v = __expf(st.x[j]);
            v = st.kin ? v : 0.f;
        } else {
            v = st.x[j] * st.wa + st.x2[j] * st.wb;
        }
        As[r * ZG_LDA + k] = v;
    }
}

#ifdef ZPG_TIMING   // dev build (tools/zp_phases.py): s_memtime stamps of thread 0 of every workgroup
__device__ unsigned long long g_zpg_stamps[8 * 4096];
#define ZPG_STAMP(i) do { if (threadIdx.x == 0) g_zpg_stamps[8 * (blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int golf_debug_zpg_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_zpg_stamps), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
}
#else
#define ZPG_STAMP(i) do { } while (0)
#endif
template <int MODE>
__global__ __launch_bounds__(256) void zp_gemm_kernel(const float* __restrict__ src, int src_stride,
                                                      const float* __restrict__ log_mag,
                                                      const float* __restrict__ window,
                                                      const float* __restrict__ Bm, float* __restrict__ out,
                                                      int out_stride, int G, int n_mag, int Pd) {
    __shared__ float Asm[2 * ZG_ROWS * ZG_LDA > ZG_ROWS * ZG_LDC ? 2 * ZG_ROWS * ZG_LDA : ZG_ROWS * ZG_LDC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g0 = blockIdx.x * ZG_ROWS, c0 = blockIdx.y * ZG_COLS;
    const int N = 2 * (n_mag - 1), H = N >> 1;
    const int li = lane & 15, lk = lane >> 4;
    ZPG_STAMP(0);

    f32x4 acc[ZG_RT][2];
#pragma unroll
    for (int rt = 0; rt < ZG_RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The epilogue's own operands, fetched HERE: read where they are used (the window taps of the thread's two columns; in
    // the backward the log-magnitudes of its 2 x 8 outputs) they were dependent loads in front of the stores -- 4.2 k of a
    // workgroup's 27 k ticks in the forward (tools/zp_phases.py).
    float ep_wu[2], ep_wd[2], ep_lm[2][ZG_RPW];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + lane + 64 * h;
        if (MODE == 0) {
            ep_wu[h] = window[H + min(c, H - 1)];
            ep_wd[h] = window[H - min(c, H)];
        } else {
            const int ccl = min(c, n_mag - 1);
#pragma unroll
            for (int rr = 0; rr < ZG_RPW; ++rr)
                ep_lm[h][rr] = log_mag[(size_t)min(g0 + w * ZG_RPW + rr, G - 1) * n_mag + ccl];
        }
    }
    // B fragments run 8 k-steps ahead across the whole K range (Bm rows are contiguous in k)
    const int ksteps_all = Pd >> 2;  // multiple of 32 (Pd is a multiple of 128)
    const float* bp = Bm + (size_t)lk * Pd + c0 + w * 32 + li;
    float bq[2][8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bq[0][i][0] = bp[(size_t)(4 * i) * Pd];
        bq[0][i][1] = bp[(size_t)(4 * i) * Pd + 16];
    }
    ZgStage<MODE> st;
    zg_fetch<MODE>(st, src, src_stride, window, g0, G, n_mag, 0, tid);
    zg_commit<MODE>(st, Asm, tid);
    __syncthreads();
    ZPG_STAMP(1);
    const int nchunk = Pd / ZG_KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        float* Acur = Asm + (ch & 1) * (ZG_ROWS * ZG_LDA);
        float* Anxt = Asm + ((ch + 1) & 1) * (ZG_ROWS * ZG_LDA);
        const bool more = ch + 1 < nchunk;
        if (more) zg_fetch<MODE>(st, src, src_stride, window, g0, G, n_mag, (ch + 1) * ZG_KC, tid);
        const float* ap = Acur + li * ZG_LDA + lk;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {  // ZG_KC / 4 = 16 k-steps per chunk = 2 phases of 8
            const int sg = ch * (ZG_KC / 4) + 8 * ph;  // global k-step of this phase
            const int sn = min(sg + 8, ksteps_all - 8);  // clamped at the very end: a redundant reload
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bq[ph ^ 1][i][0] = bp[(size_t)(4 * (sn + i)) * Pd];
                bq[ph ^ 1][i][1] = bp[(size_t)(4 * (sn + i)) * Pd + 16];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the 16 prefetch loads ahead of this phase's MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = 8 * ph + i;
                float a[ZG_RT];
#pragma unroll
                for (int rt = 0; rt < ZG_RT; ++rt) a[rt] = ap[rt * 16 * ZG_LDA + 4 * s];
#pragma unroll
                for (int rt = 0; rt < ZG_RT; ++rt) {
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ph][i][0], acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ph][i][1], acc[rt][1], 0, 0, 0);
                }
            }
        }
        if (more) zg_commit<MODE>(st, Anxt, tid);
        __syncthreads();
    }
    ZPG_STAMP(2);
    float* As = Asm;
    // epilogue through LDS so that global stores run along rows (the last loop iteration ended with a barrier)
    float* Cs = As;
#pragma unroll
    for (int rt = 0; rt < ZG_RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(rt * 16 + lk * 4 + r) * ZG_LDC + w * 32 + ct * 16 + li] = acc[rt][ct][r];
    __syncthreads();
    ZPG_STAMP(3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cc = lane + 64 * h, c = c0 + cc;
        if (MODE == 0) {
            const bool up = c < H, dn = c >= 1 && c <= H;
            const float wu = ep_wu[h], wd = ep_wd[h];
#pragma unroll 4
            for (int rr = 0; rr < ZG_RPW; ++rr) {
                const int row = w * ZG_RPW + rr, g = g0 + row;
                const float u = Cs[row * ZG_LDC + cc];
                float* orow = out + (size_t)g * out_stride;
                if (g < G && up) orow[H + c] = u * wu;
                if (g < G && dn) orow[H - c] = u * wd;
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < ZG_RPW; ++rr) {
                const int row = w * ZG_RPW + rr, g = g0 + row;
                const float u = Cs[row * ZG_LDC + cc];
                const float e = __expf(ep_lm[h][rr]);
                if (g < G && c < n_mag) out[(size_t)g * out_stride + c] = u * e;
            }
        }
    }
    ZPG_STAMP(4);
    if (MODE == 0 && blockIdx.y == 0) {  // zero the row padding [N, out_stride)
        for (int rr = 0; rr < ZG_RPW; ++rr) {
            const int g = g0 + w * ZG_RPW + rr;
            if (g >= G) break;
            for (int j = N + lane; j < out_stride; j += 64) out[(size_t)g * out_stride + j] = 0.f;
        }
    }
}

static inline int zp_pad(int n_mag) { return (int)align_up((size_t)n_mag, ZG_COLS); }

}  // namespace golf

using namespace golf;

extern "C" {

int golf_zero_phase_fir_row_stride(int n_mag) {
    if (n_mag < 2) return 0;
    return (int)align_up((size_t)(2 * (n_mag - 1)), 16);
}

size_t golf_zero_phase_fir_basis_bytes(int n_mag) {
    if (n_mag < 2) return 0;
    const size_t Pd = zp_pad(n_mag);
    return 2 * Pd * Pd * sizeof(float);
}

int golf_zero_phase_fir_basis_f32(int n_mag, void* basis, size_t basis_bytes, void* stream) {
    if (n_mag < 2 || !basis) return fail(GOLF_EINVAL, "zero_phase_fir_basis: n_mag=%d basis=%p", n_mag, basis);
    if (basis_bytes < golf_zero_phase_fir_basis_bytes(n_mag))
        return fail(GOLF_EWORKSPACE, "zero_phase_fir_basis: need %zu bytes, got %zu",
                    golf_zero_phase_fir_basis_bytes(n_mag), basis_bytes);
    const int Pd = zp_pad(n_mag);
    float* bas = (float*)basis;
    hipLaunchKernelGGL(zp_basis_kernel, dim3((Pd + 255) / 256, Pd), dim3(256), 0, (hipStream_t)stream, bas,
                       bas + (size_t)Pd * Pd, n_mag, Pd);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

int golf_zero_phase_fir_kernels_f32(const float* log_mag, const float* window, const void* basis, float* kern,
                                    int G, int n_mag, void* stream) {
    if (!log_mag || !window || !basis || !kern || G < 1 || n_mag < 2)
        return fail(GOLF_EINVAL, "zero_phase_fir_kernels: bad argument (G=%d n_mag=%d)", G, n_mag);
    const int Pd = zp_pad(n_mag), KS = golf_zero_phase_fir_row_stride(n_mag);
    hipLaunchKernelGGL(zp_gemm_kernel<0>, dim3((G + ZG_ROWS - 1) / ZG_ROWS, Pd / ZG_COLS), dim3(256), 0,
                       (hipStream_t)stream, log_mag, n_mag, (const float*)nullptr, window, (const float*)basis, kern,
                       KS, G, n_mag, Pd);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

int golf_zero_phase_fir_kernels_bwd_f32(const float* g_kern, const float* log_mag, const float* window,
                                        const void* basis, float* g_log_mag, int G, int n_mag, void* stream) {
    if (!g_kern || !log_mag || !window || !basis || !g_log_mag || G < 1 || n_mag < 2)
        return fail(GOLF_EINVAL, "zero_phase_fir_kernels_bwd: bad argument (G=%d n_mag=%d)", G, n_mag);
    const int Pd = zp_pad(n_mag), KS = golf_zero_phase_fir_row_stride(n_mag);
    const float* basT = (const float*)basis + (size_t)Pd * Pd;
    hipLaunchKernelGGL(zp_gemm_kernel<1>, dim3((G + ZG_ROWS - 1) / ZG_ROWS, Pd / ZG_COLS), dim3(256), 0,
                       (hipStream_t)stream, g_kern, KS, log_mag, window, basT, g_log_mag, n_mag, G, n_mag, Pd);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

static int fir_geometry(const char* who, int B, int T, int F, int N, int hop, int KS, int frame0, int* nfr_out) {
    if (B < 1 || T < 1 || F < 1 || N < 2 || hop < 1 || KS < ((N + 3) & ~3) || frame0 < 0 || frame0 >= F)
        return fail(GOLF_EINVAL, "%s: bad sizes B=%d T=%d F=%d N=%d hop=%d row_stride=%d frame0=%d", who, B, T, F, N, hop,
                    KS, frame0);
    const int P = (N - 1) / 2, span = N + hop - 1;
    if (T + 2 * P < span)
        return fail(GOLF_EINVAL, "%s: excitation (T=%d) shorter than one frame span (%d)", who, T, span - 2 * P);
    int nfr = (T + 2 * P - span) / hop + 1;
    if (nfr > F - frame0) nfr = F - frame0;
    *nfr_out = nfr;
    return GOLF_OK;
}

int golf_ltv_fir_frames_length(int T, int F, int N, int hop) {
    int nfr = 0;
    if (fir_geometry("ltv_fir_frames_length", 1, T, F, N, hop, (N + 3) & ~3, 0, &nfr)) return -1;
    return nfr * hop;
}

int golf_ltv_fir_frames_fwd_f32(const float* ex, int64_t ex_stride, const float* kern, int kern_row_stride, float* y,
                                int64_t y_stride, int B, int T, int F, int N, int hop, int frame0, void* stream) {
    if (!ex || !kern || !y) return fail(GOLF_EINVAL, "ltv_fir_frames_fwd: null pointer");
    int nfr = 0;
    if (int rc = fir_geometry("ltv_fir_frames_fwd", B, T, F, N, hop, kern_row_stride, frame0, &nfr)) return rc;
    const int npass = (hop + FIR_TILE - 1) / FIR_TILE;
    const int RS = fir_region(256 + ((N + 3) & ~3) + 4);
    const long long units = (long long)B * nfr * npass;
    hipLaunchKernelGGL(fir_frames_fwd_kernel, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)),
                       dim3(64 * FIR_WAVES), FIR_WAVES * RS * sizeof(float), (hipStream_t)stream, ex, ex_stride, kern,
                       kern_row_stride, y, y_stride, B, T, nfr, F, N, hop, npass, RS, frame0);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

int golf_ltv_fir_frames_bwd_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride,
                                const float* kern, int kern_row_stride, float* g_ex, int64_t g_ex_stride,
                                float* g_kern, int B, int T, int F, int N, int hop, int frame0, void* stream) {
    if (!gy || !ex || !kern) return fail(GOLF_EINVAL, "ltv_fir_frames_bwd: null pointer");
    int nfr = 0;
    if (int rc = fir_geometry("ltv_fir_frames_bwd", B, T, F, N, hop, kern_row_stride, frame0, &nfr)) return rc;
    if (hop % 4 != 0)
        return fail(GOLF_EUNSUPPORTED, "ltv_fir_frames_bwd: hop=%d must be a multiple of 4 (the frame's gradient "
                    "samples are the packed taps of the backward kernels)", hop);
    const int RS = fir_region(256 + hop + 4);
    hipStream_t st = (hipStream_t)stream;
    if (g_kern) {
        // passes over the whole ROW (kern_row_stride >= N): the padding taps [N, KS) are part of the contract ("zeroed") and a
        // pass count taken from N left [npass * 252, KS) unwritten whenever N sat just under a multiple of 252
        const int npass = (kern_row_stride + FIR_TILE - 1) / FIR_TILE;
        const long long units = (long long)B * F * npass;
        hipLaunchKernelGGL(fir_frames_bwd_kern_kernel, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)),
                           dim3(64 * FIR_WAVES), FIR_WAVES * RS * sizeof(float), st, gy, gy_stride, ex, ex_stride,
                           g_kern, kern_row_stride, B, T, nfr, F, N, hop, npass, RS, frame0);
        GOLF_LAUNCH_CHECK();
    }
    if (g_ex) {
        const int P = (N - 1) / 2;
        const int TILE = hop <= FIR_TILE ? hop : FIR_TILE;
        const int tile_lo = P / TILE, tile_hi = (P + T - 1) / TILE;
        const int ntile = tile_hi - tile_lo + 1;
        const long long units = (long long)B * ntile;
        hipLaunchKernelGGL(fir_frames_bwd_ex_kernel, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)),
                           dim3(64 * FIR_WAVES), FIR_WAVES * RS * sizeof(float), st, gy, gy_stride, kern,
                           kern_row_stride, g_ex, g_ex_stride, B, T, nfr, F, N, hop, TILE, tile_lo, ntile, RS, frame0);
        GOLF_LAUNCH_CHECK();
    }
    return GOLF_OK;
}

static int lti_check(const char* who, int B, int T, int ntaps, int lead) {
    if (B < 1 || T < 1 || ntaps < 4 || (ntaps & 3) || lead < 0 || lead >= ntaps)
        return fail(GOLF_EINVAL, "%s: bad sizes B=%d T=%d ntaps=%d (multiple of 4) lead=%d (in [0,ntaps))", who, B, T,
                    ntaps, lead);
    return GOLF_OK;
}

int golf_lti_fir_f32(const float* ex, int64_t ex_stride, const float* taps, int ntaps, int lead, float* y,
                     int64_t y_stride, int B, int T, void* stream) {
    if (!ex || !taps || !y) return fail(GOLF_EINVAL, "lti_fir: null pointer");
    const bool rev = ntaps < 0;   // ABI 5: a negative tap count applies the |ntaps| taps in reverse order (the adjoint, no flipped copy)
    if (rev) ntaps = -ntaps;
    if (int rc = lti_check("lti_fir", B, T, ntaps, lead)) return rc;
    const int ntile = (T + FIR_TILE - 1) / FIR_TILE;
    const int RS = fir_region(256 + ntaps + 4);
    const long long units = (long long)B * ntile;
    if (rev)
        hipLaunchKernelGGL(lti_fir_kernel<true>, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)), dim3(64 * FIR_WAVES),
                           FIR_WAVES * RS * sizeof(float), (hipStream_t)stream, ex, ex_stride, taps, ntaps, lead, y,
                           y_stride, B, T, ntile, RS);
    else
        hipLaunchKernelGGL(lti_fir_kernel<false>, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)), dim3(64 * FIR_WAVES),
                           FIR_WAVES * RS * sizeof(float), (hipStream_t)stream, ex, ex_stride, taps, ntaps, lead, y,
                           y_stride, B, T, ntile, RS);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

size_t golf_lti_fir_taps_grad_workspace_bytes(int B, int T, int ntaps) {
    if (B < 1 || T < 1 || ntaps < 1) return 0;
    const size_t nstretch = (T + LTI_GRAD_LT - 1) / LTI_GRAD_LT;
    return align_up((size_t)B * nstretch * ntaps * sizeof(float), 256);
}

int golf_lti_fir_taps_grad_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride, float* g_taps,
                               int ntaps, int lead, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!gy || !ex || !g_taps || !ws) return fail(GOLF_EINVAL, "lti_fir_taps_grad: null pointer");
    if (int rc = lti_check("lti_fir_taps_grad", B, T, ntaps, lead)) return rc;
    if (ws_bytes < golf_lti_fir_taps_grad_workspace_bytes(B, T, ntaps) || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "lti_fir_taps_grad: workspace too small or misaligned (%zu < %zu)", ws_bytes,
                    golf_lti_fir_taps_grad_workspace_bytes(B, T, ntaps));
    const int nstretch = (T + LTI_GRAD_LT - 1) / LTI_GRAD_LT;
    const int npass = (ntaps + FIR_TILE - 1) / FIR_TILE;
    const int RS = fir_region(256 + LTI_GRAD_LT + 4);
    const long long units = (long long)B * nstretch * npass;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lti_fir_taps_grad_kernel, dim3((unsigned)((units + FIR_WAVES - 1) / FIR_WAVES)),
                       dim3(64 * FIR_WAVES), FIR_WAVES * RS * sizeof(float), st, gy, gy_stride, ex, ex_stride,
                       (float*)ws, ntaps, lead, B, T, nstretch, npass, RS);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(lti_fir_taps_reduce_kernel, dim3(ntaps), dim3(256), 0, st, (const float*)ws, g_taps, ntaps,
                       B * nstretch);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

}  // extern "C"
