// Frame-wise LTI all-pole filter + windowed overlap-add for gfx950 — GOLF-ff end filter.
//
// Replaces LTVMinimumPhaseFilter.forward (reference models/filters.py:131-184): zero-pad, unfold to
// (B*F, W) frames, torchaudio.functional.lfilter per frame (models/lpc.py:11-16), and the dense
// diagonal conv_transpose1d that the reference uses for the windowed OLA (6.1 GMAC for 6.1 MMAC of
// useful work at B=32) plus the ones-row normaliser.
//
// Here: F1  one lane per frame runs the Wl-step LTI recursion in a rotating register window
//           (coefficients constant per frame => one FMA per tap), multiplies by the window and
//           stores the windowed frame;  6400 lanes at B=32.
//       F2  gathers the <= Wl/hop overlapping frames per output sample and divides by the
//           window sum (computed on the fly from the same window values, like the reference's
//           extra ones row).
#include "common.h"
#include "device_common.h"

namespace golf {

template <int W, int NT>
__global__ __launch_bounds__(64) void ff_frames_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                       const float* __restrict__ gain, const float* __restrict__ a,
                                                       const float* __restrict__ window, float* __restrict__ wf,
                                                       int Tx, int F, int M, int hop, int Wl, int nfr, int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int b = q / nfr, f = q - b * nfr;
    const int pad = Wl / 2;
    float a0[NT];
    {
        const float* pa = a + ((size_t)b * F + f) * M;
#pragma unroll
        for (int i = 0; i < NT; ++i) a0[i] = i < M ? pa[i] : 0.f;
    }
    float h[W];
#pragma unroll
    for (int k = 0; k < W; ++k) h[k] = 0.f;
    const float inv_hop = 1.0f / (float)hop;
    const float* exb = ex + (size_t)b * ex_stride;
    const float* gb = gain + (size_t)b * F;
    float* out = wf + (size_t)q * Wl;
    const int tstart = f * hop - pad;
    const int nblk = (Wl + W - 1) / W;
    for (int blk = 0; blk < nblk; ++blk) {
        const int k0 = blk * W;
        const int t0 = tstart + k0;
        // gain line(s) for this block: at most one frame boundary inside (W <= hop)
        const int tb = t0 > 0 ? t0 : 0;
        int ft = tb / hop;
        if (ft > F - 2) ft = F - 2;
        const float gA = gb[ft];
        const float gB = gb[ft + 1];
        const float dA = (gB - gA) * inv_hop;
        const float gC = ft + 2 < F ? gb[ft + 2] : gB;
        const float dB = (gC - gB) * inv_hop;
        const bool can_cross = ft < F - 2;
        const int nbase = t0 - ft * hop;
        float xin[W];
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const int t = t0 + s;
            xin[s] = (t >= 0 && t < Tx) ? exb[t] : 0.f;
        }
        float res[W];
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const int n = nbase + s;
            const float G = (can_cross && n >= hop) ? fmaf((float)(n - hop), dB, gB) : fmaf((float)n, dA, gA);
            const float x = xin[s] * G;
            float ra = 0.f, rb = 0.f;
#pragma unroll
            for (int i = NT - 1; i >= 1; --i) {
                const int slot = (s - 1 - i + 2 * W) % W;
                if (i & 1) ra = fmaf(a0[i], h[slot], ra);
                else       rb = fmaf(a0[i], h[slot], rb);
            }
            const float y = fmaf(-a0[0], h[(s - 1 + W) % W], x - (ra + rb));
            h[s] = y;
            res[s] = y;
        }
#pragma unroll
        for (int s = 0; s < W; ++s)
            if (k0 + s < Wl) out[k0 + s] = res[s] * window[k0 + s];
    }
}

// Quad version (fast path, Wl % W == 0): 4 lanes per frame, each owning TPL taps and a TPL-deep systolic window
// (see lpc_ss.hip / device_common.h), 16 frames per wave, coalesced bounds-checked tile I/O.  The frame's
// zero padding is what the buffer descriptor returns outside [0,Tx).
template <int W, int NT>
__global__ __launch_bounds__(64) void ff_framesq_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                        const float* __restrict__ gain, const float* __restrict__ a,
                                                        const float* __restrict__ window, float* __restrict__ wf,
                                                        int Tx, int F, int M, int hop, int Wl, int nfr) {
    constexpr int TPL = quad_tpl(W, NT);
    constexpr int R = 16;
    using TL = Tile<W, R>;
    __shared__ float xt[TL::SIZE];
    __shared__ float yt[TL::SIZE];
    const int b = blockIdx.y, fg = blockIdx.x;
    const int lane = threadIdx.x;
    const int lq = lane / W, lr = lane % W;
    const int row = lane >> 2, r = lane & 3;
    const int f0 = fg * R;
    const int f = f0 + row;
    const bool mine = f < nfr;
    const int pad = Wl / 2;
    const BufRow xrow(ex + (size_t)b * ex_stride, Tx);
    const BufRow orow(wf + (size_t)b * nfr * Wl, nfr * Wl);
    float cf[TPL];
    {
        const float* pa = a + ((size_t)b * F + (mine ? f : 0)) * M;
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            const int i = r * TPL + k;
            cf[k] = (mine && i < M) ? pa[i] : 0.f;
        }
    }
    float w[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) w[k] = 0.f;
    const float inv_hop = 1.0f / (float)hop;
    const float* gb = gain + (size_t)b * F;
    const int nblk = Wl / W;
    // window tile in LDS (applied per element in the coalesced store phase, not per recursion step)
    extern __shared__ __attribute__((aligned(16))) float wl[];
    for (int k = lane; k < Wl; k += 64) wl[k] = window[k];
    float gA = 0.f, dA = 0.f, gB = 0.f, dB = 0.f;
    bool can_cross = false;
    int ftcur = -1;
    float nx[TL::ITS];
    TL::fetch(nx, xrow, f0 * hop - pad, hop, lq, lr);
    for (int blk = 0; blk < nblk; ++blk) {
        TL::scatter(xt, nx, lq, lr);
        __syncthreads();
        float xin[W];
        TL::rows_load(xin, xt, row);
        TL::fetch(nx, xrow, f0 * hop - pad + (blk + 1) * W, hop, lq, lr);
        const int k0 = blk * W;
        const int t0 = f * hop - pad + k0;
        // gain line(s) for this block: at most one frame boundary inside (W <= hop); reload only on change
        const int tb = t0 > 0 ? t0 : 0;
        int ft = tb / hop;
        if (ft > F - 2) ft = F - 2;
        if (ft != ftcur) {
            ftcur = ft;
            gA = gb[ft];
            gB = gb[ft + 1];
            dA = (gB - gA) * inv_hop;
            const float gC = ft + 2 < F ? gb[ft + 2] : gB;
            dB = (gC - gB) * inv_hop;
            can_cross = ft < F - 2;
        }
        const int nbase = t0 - ft * hop;
        float keep[W / 4];
#pragma unroll
        for (int j = 0; j < W / 4; ++j) keep[j] = 0.f;
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const int n = nbase + s;
            const float G = (can_cross && n >= hop) ? fmaf((float)(n - hop), dB, gB) : fmaf((float)n, dA, gA);
            const float x = xin[s] * G;
            float pa_ = 0.f, pb_ = 0.f;
#pragma unroll
            for (int k = TPL - 1; k >= 1; --k) {
                const int slot = (s - 1 - k + 4 * TPL) % TPL;
                if (k & 1) pa_ = fmaf(cf[k], w[slot], pa_);
                else       pb_ = fmaf(cf[k], w[slot], pb_);
            }
            float part = fmaf(cf[0], w[(s - 1 + TPL) % TPL], pa_ + pb_);
            part += dppf<DPP_XOR1>(part);
            part += dppf<DPP_XOR2>(part);
            const float y = x - part;
            const float inc = dppf<DPP_SHR1>(w[s % TPL]);
            w[s % TPL] = r == 0 ? y : inc;
            keep[s >> 2] = ((s & 3) == r) ? y : keep[s >> 2];
        }
#pragma unroll
        for (int j = 0; j < W / 4; ++j) yt[row * TL::LD + 4 * j + r] = keep[j];
        __syncthreads();
        float o[TL::ITS];
        TL::gather(o, yt, lq, lr);
#pragma unroll
        for (int it = 0; it < TL::ITS; ++it) {
            int trow, tcol;
            TL::rowcol(it, lq, lr, trow, tcol);
            o[it] *= wl[k0 + tcol];
        }
        TL::store(o, orow, f0 * Wl + k0, Wl, lq, lr);
        __syncthreads();
    }
}

__global__ void ff_ola_kernel(const float* __restrict__ wf, const float* __restrict__ window, float* __restrict__ y,
                              int64_t y_stride, int B, int Ty, int hop, int Wl, int nfr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Ty) return;
    const int b = (int)(idx / Ty), n = (int)(idx - (int64_t)b * Ty);
    const int pad = Wl / 2;
    const int m = n + pad;          // position in the padded signal
    int fhi = m / hop;              // k = m - f*hop >= 0
    if (fhi > nfr - 1) fhi = nfr - 1;
    int flo = (m - Wl + hop) / hop; // smallest f with m - f*hop <= Wl-1  (ceil((m-Wl+1)/hop))
    if (m - Wl + 1 <= 0) flo = 0;
    float acc = 0.f, norm = 0.f;
    for (int f = flo; f <= fhi; ++f) {
        const int k = m - f * hop;
        if (k < 0 || k >= Wl) continue;
        acc += wf[((size_t)b * nfr + f) * Wl + k];
        norm += window[k];
    }
    y[(size_t)b * y_stride + n] = acc / norm;
}

template <int W, int NT>
static int launch_ff(const float* ex, int64_t ex_stride, const float* gain, const float* a, const float* window,
                     float* y, int64_t y_stride, int B, int Tx, int F, int M, int hop, int Wl, int Ty, int nfr,
                     float* wf, hipStream_t st) {
    const int nq = B * nfr;
    if (Wl % W == 0 && (int64_t)nfr * Wl < (1ll << 29) && Wl <= 32768) {
        hipLaunchKernelGGL((ff_framesq_kernel<W, NT>), dim3((unsigned)ceil_div(nfr, 16), B), dim3(64),
                           sizeof(float) * (size_t)Wl, st, ex,
                           ex_stride, gain, a, window, wf, Tx, F, M, hop, Wl, nfr);
    } else {
        hipLaunchKernelGGL((ff_frames_kernel<W, NT>), dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, st, ex, ex_stride,
                           gain, a, window, wf, Tx, F, M, hop, Wl, nfr, nq);
    }
    GOLF_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * Ty;
    hipLaunchKernelGGL(ff_ola_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)wf, window, y,
                       y_stride, B, Ty, hop, Wl, nfr);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

static int ff_geometry(int Tx, int F, int hop, int Wl, int* nfr, int* Ty) {
    const int pad = Wl / 2;
    *nfr = (Tx + 2 * pad - Wl) / hop + 1;
    *Ty = (*nfr - 1) * hop + Wl - 2 * pad;
    return 0;
}

}  // namespace golf

using namespace golf;

extern "C" size_t golf_lti_frames_workspace_bytes(int B, int Tx, int F, int M, int hop, int W) {
    if (B < 1 || Tx < 1 || F < 1 || hop < 1 || W < 1) return 0;
    int nfr, Ty;
    ff_geometry(Tx, F, hop, W, &nfr, &Ty);
    if (nfr < 1) return 256;
    return align_up(sizeof(float) * (size_t)B * nfr * W, 256);
}

extern "C" int golf_lti_frames_ola_fwd_f32(const float* ex, int64_t ex_stride, const float* gain, const float* a,
                                           const float* window, float* y, int64_t y_stride, int B, int Tx, int F,
                                           int M, int hop, int W, int Ty, void* ws, size_t ws_bytes, void* stream) {
    if (B < 1 || Tx < 1 || F < 2 || M < 1 || hop < 1 || W < 1)
        return fail(GOLF_EINVAL, "lti_frames: bad size (need F >= 2)");
    if (!ex || !gain || !a || !window || !y) return fail(GOLF_EINVAL, "lti_frames: null pointer");
    if (W < 2 * hop) return fail(GOLF_EINVAL, "lti_frames: window %d < 2*hop %d", W, 2 * hop);
    if ((int64_t)Tx > (int64_t)(F - 1) * hop + 1) return fail(GOLF_EINVAL, "lti_frames: Tx exceeds (F-1)*hop+1");
    int nfr, ty;
    ff_geometry(Tx, F, hop, W, &nfr, &ty);
    if (nfr < 1 || nfr > F) return fail(GOLF_EINVAL, "lti_frames: %d frames vs %d coefficient frames", nfr, F);
    if (ty != Ty) return fail(GOLF_EINVAL, "lti_frames: Ty=%d, expected %d", Ty, ty);
    if (ex_stride < Tx || y_stride < Ty) return fail(GOLF_EINVAL, "lti_frames: row stride too small");
    const size_t need = align_up(sizeof(float) * (size_t)B * nfr * W, 256);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "lti_frames: workspace needs %zu bytes, 256-aligned (got %zu)", need, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* wf = (float*)ws;
#define GOLF_FF_TRY(w, nt)                                                                                   \
    if (M <= (nt) && (w) <= hop)                                                                             \
        return launch_ff<w, nt>(ex, ex_stride, gain, a, window, y, y_stride, B, Tx, F, M, hop, W, Ty, nfr, wf, st);
    GOLF_FF_TRY(8, 6)
    GOLF_FF_TRY(16, 14)
    GOLF_FF_TRY(24, 22)
    GOLF_FF_TRY(32, 30)
    GOLF_FF_TRY(40, 38)
#undef GOLF_FF_TRY
    return fail(GOLF_EUNSUPPORTED, "lti_frames: need M <= 38 and hop >= ring width (M=%d hop=%d)", M, hop);
}
