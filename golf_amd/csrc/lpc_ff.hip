// Frame-wise LTI all-pole filter + windowed overlap-add for gfx950 — GOLF-ff end filter.
//
// Replaces LTVMinimumPhaseFilter.forward (reference models/filters.py:131-184): zero-pad, unfold to
// (B*F, W) frames, torchaudio.functional.lfilter per frame (models/lpc.py:11-16), and the dense
// diagonal conv_transpose1d that the reference uses for the windowed OLA (6.1 GMAC for 6.1 MMAC of
// useful work at B=32) plus the ones-row normaliser.
//
// Here: F1  a quad of lanes per frame runs the Wl-step LTI recursion (coefficients constant per frame) and
//           stores the filtered frame y_f (kept for the backward pass);  6400 frames at B=32.
//       F2  gathers the <= Wl/hop overlapping frames per output sample, applies the window and divides by the
//           window sum (computed on the fly from the same window values, like the reference's extra ones row).
// Backward (what autograd does in the reference through conv_transpose1d, lfilter, unfold and the gain product):
//       B0  g_q = gy / norm
//       B1  the SAME recursion kernel run backwards in time on window * g_q frames -> u_f (the adjoint of an
//           LTI all-pole filter is the filter itself applied to the time-reversed signal)
//       B2  g_a[f,i] = -sum_k u_f[k] * y_f[k-1-i]   (one wave per frame)
//       B3  g_x = overlap-add of the u_f; g_ex = g_x * G; hat-weighted partial sums of g_x * ex -> g_gain
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
            if (k0 + s < Wl) out[k0 + s] = res[s];  // unwindowed: the window is applied by the overlap-add
    }
}

// Quad version (fast path, Wl % W == 0): 4 lanes per frame, each owning TPL taps and a TPL-deep systolic window
// (see lpc_ss.hip / device_common.h), 16 frames per wave, coalesced bounds-checked tile I/O.  The frame's
// zero padding is what the buffer descriptor returns outside [0,Tx).
//   REV = false  forward:  input ex * up(gain), frame position k ascending, output y_f[k] (unwindowed)
//   REV = true   adjoint:  input window[k] * g_q[f*hop + k - pad], k DEscending, no gain, output u_f[k]
//                (`ex` = g_q with Tx = Ty; the recursion is the same one, run on the time-reversed frame)
template <int W, int NT, bool REV>
__global__ __launch_bounds__(64) void ff_framesq_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                        const float* __restrict__ gain, const float* __restrict__ a,
                                                        const float* __restrict__ window, float* __restrict__ wf,
                                                        int Tx, int F, int M, int hop, int Wl, int nfr) {
    constexpr int TPL = quad_tpl(W, NT);
    constexpr int R = 16;
    constexpr int DIR = REV ? -1 : 1;
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
    // window in LDS (REV only: applied to the incoming gradient frame)
    extern __shared__ __attribute__((aligned(16))) float wl[];
    if (REV) {   // (batches of 8 loads in flight: one waited-for round trip per 64 window values otherwise -- 15 of them)
        for (int k0 = lane; k0 < Wl; k0 += 64 * 8) {
            float wv_[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv_[u] = window[k0 + 64 * u < Wl ? k0 + 64 * u : 0];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 64 * u < Wl) wl[k0 + 64 * u] = wv_[u];
        }
    }
    float nx[TL::ITS], ng[TL::ITS];
    // Forward: the interpolated gain G(t) is evaluated per ELEMENT in the coalesced (parallel) fetch phase, one block
    // ahead, instead of per step inside the serial recursion (5 of its ~23 instructions per step: 76 -> 47 us).
    // Rows are consecutive frames (stride hop) and a block spans W <= hop positions, so the gain segment of element
    // (row, col) is f0 + row + q0 (+1 if the remainder wraps) with q0, rem0 wave-uniform.
    auto fetch_gain = [&](int blkx) {
        const int kk0 = blkx * W - pad;
        int q0 = kk0 / hop, rem0 = kk0 - q0 * hop;
        if (rem0 < 0) { rem0 += hop; q0 -= 1; }
#pragma unroll
        for (int it = 0; it < TL::ITS; ++it) {
            int trow, tcol;
            TL::rowcol(it, lq, lr, trow, tcol);
            int n = rem0 + tcol, ft = f0 + trow + q0;
            if (n >= hop) { n -= hop; ft += 1; }
            if (ft > F - 2) { n += (ft - (F - 2)) * hop; ft = F - 2; }
            if (ft < 0) { ft = 0; n = 0; }  // t < 0: the sample is zero padding anyway
            const float g0 = gb[ft], g1 = gb[ft + 1];
            ng[it] = fmaf((float)n, (g1 - g0) * inv_hop, g0);
        }
    };
    // block blk covers frame positions k0 .. k0+W-1 (forward) resp. Wl-1-k0 .. Wl-W-k0 (adjoint, descending)
    const int in0 = REV ? f0 * hop - pad + Wl - 1 : f0 * hop - pad;
    TL::template fetch<DIR>(nx, xrow, in0, hop, lq, lr);
    if (!REV) fetch_gain(0);
    for (int blk = 0; blk < nblk; ++blk) {
        if (!REV) {
#pragma unroll
            for (int it = 0; it < TL::ITS; ++it) nx[it] *= ng[it];
        }
        TL::scatter(xt, nx, lq, lr);
        __syncthreads();
        float xin[W];
        TL::rows_load(xin, xt, row);
        TL::template fetch<DIR>(nx, xrow, in0 + DIR * (blk + 1) * W, hop, lq, lr);
        if (!REV) fetch_gain(blk + 1);
        const int k0 = blk * W;
        float keep[W / 4];
#pragma unroll
        for (int j = 0; j < W / 4; ++j) keep[j] = 0.f;
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const float x = REV ? xin[s] * wl[Wl - 1 - k0 - s] : xin[s];
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
        TL::template store<DIR>(o, orow, REV ? f0 * Wl + Wl - 1 - k0 : f0 * Wl + k0, Wl, lq, lr);
        __syncthreads();
    }
}

__global__ void ff_ola_kernel(const float* __restrict__ wf, const float* __restrict__ window, float* __restrict__ y,
                              int64_t y_stride, int B, int Ty, int hop, int Wl, int nfr, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Ty) return;
    const int b = (int)(idx / Ty), n = (int)(idx - (int64_t)b * Ty);
    const int m = n + pad;          // position in the padded signal
    int fhi = m / hop;              // k = m - f*hop >= 0
    if (fhi > nfr - 1) fhi = nfr - 1;
    int flo = (m - Wl + hop) / hop; // smallest f with m - f*hop <= Wl-1  (ceil((m-Wl+1)/hop))
    if (m - Wl + 1 <= 0) flo = 0;
    float acc = 0.f, norm = 0.f;
    if (Wl <= 4 * hop) {
        // at most 4 frames overlap a sample (every shipped config: Wl = 4 hop or 2 hop): all loads issued before the first
        // use -- as a loop with a data-dependent trip count this was up to 4 serial pairs of round trips per sample
        float wk[4], vf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = flo + u;
            const int k = m - f * hop;
            const bool ok = f <= fhi && k >= 0 && k < Wl;
            const int kc = ok ? k : 0, fc = ok ? f : flo;
            wk[u] = window[kc];
            vf[u] = wf[((size_t)b * nfr + fc) * Wl + kc];
            if (!ok) { wk[u] = 0.f; vf[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = fmaf(wk[u], vf[u], acc);
            norm += wk[u];
        }
    } else {
        for (int f = flo; f <= fhi; ++f) {
            const int k = m - f * hop;
            if (k < 0 || k >= Wl) continue;
            const float wk = window[k];
            acc = fmaf(wk, wf[((size_t)b * nfr + f) * Wl + k], acc);
            norm += wk;
        }
    }
    y[(size_t)b * y_stride + n] = acc / norm;
}

// ------------------------------------------------------------------------------------------
// Block recursion (round 3): 16 samples of a frame per step instead of one.
// A frame's coefficients are CONSTANT, so the recursion's look-ahead form costs its set-up once per frame:
//     y[n + r] = sum_{m <= r} h[r - m] x[n + m]  +  sum_j G[r][j] y[n - 1 - j],        r = 0 .. 15,
// h = the frame's impulse response, G[r] = e_0^T C^(r+1) (C = companion matrix of the frame's coefficients) -- both from one
// 16-step row recurrence v <- v^T C (v[j] <- v[j+1] - a_j v[0]; h[s] = v_s[0], G[s] = v_(s+1)) that every lane of the frame
// runs.  16 lanes own one frame, lane r owns row r (16 + NS coefficients in registers): a block is 38 independent FMAs per
// lane on operands broadcast from LDS (the block's 16 inputs, the last NS outputs), and the serial chain per frame is Wl / 16
// = 60 blocks instead of 960 samples: ~45 instructions per 16 samples and lane against ~18 per sample for a quad of the
// direct form (ff_framesq_kernel) -- the same lane-instructions per sample, a quarter of the chain, four times the waves
// (1 600 at B = 32, where the quads' 400 left 60 % of the SIMDs idle).
// Arithmetic: not the reference's order of operations (each output is a dot product with impulse-response values instead of
// a feedback sum with the coefficients), same conditioning (h and G are what the feedback sum builds implicitly); measured
// against the float64 oracle next to the direct form in tests/test_gpu_lpc_ff.py.
// Inputs: forward -- the union of the wave's 4 overlapping frames, ex * up(gain), staged once in LDS; adjoint (REV) -- the 4
// frames' window * g_q products, staged per frame, the recursion walking them backwards.
//   grid (ceil(nfr / 4), B), 64 threads; needs Wl % 32 == 0, hop % 4 == 0, M <= NT <= 24; dynamic LDS = the staged inputs.
#ifndef FF_SETUP_T
#define FF_SETUP_T double
#endif
template <int NT, bool REV>
__global__ __launch_bounds__(64) void ff_framesb_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                        const float* __restrict__ gain, const float* __restrict__ a,
                                                        const float* __restrict__ window, float* __restrict__ wf,
                                                        int Tx, int F, int M, int hop, int Wl, int nfr, float kappa_max) {
    constexpr int NS = (NT + 3) & ~3;          // state values a block reads (whole 16-byte words)
    constexpr int NS4 = NS / 4;
    static_assert(NS <= 24, "the output ring holds two blocks");
    __shared__ __attribute__((aligned(16))) double Gs[4][16][NS];
    __shared__ __attribute__((aligned(16))) float hs[4][32];
    __shared__ __attribute__((aligned(16))) double yr[4][32];   // the last two blocks' outputs, kept in double: the FEEDBACK
    // part of a block (G . state) runs in double -- C^16's rows cancel against the state as the coefficients themselves do
    // in the direct form, only sixteen-fold, and in fp32 that cost 1.5 digits (5e-3 instead of 1e-4 on the order-22 filter
    // with poles at 0.97); the input part (h . x) has no such cancellation and stays fp32.
    extern __shared__ __attribute__((aligned(16))) float stage[];   // forward: U[3 hop + Wl]; adjoint: P[4][Wl]
    const int lane = threadIdx.x, fr = lane >> 4, r = lane & 15;
    const int b = blockIdx.y, f0 = blockIdx.x * 4;
    const int f = f0 + fr;
    const bool mine = f < nfr;
    const int fc = mine ? f : nfr - 1;
    const int pad = Wl / 2;
    const BufRow xrow(ex + (size_t)b * ex_stride, Tx);
    // ---- inputs -> LDS (coalesced, zero outside [0, Tx): the frames' zero padding)
    if (!REV) {
        const float inv_hop = 1.0f / (float)hop;
        const float* gb = gain + (size_t)b * F;
        const int nu = 3 * hop + Wl;
        constexpr int UB = 8;   // loads of UB elements in flight before the first is used (a loop of dependent round trips otherwise)
        for (int i0 = lane; i0 < nu; i0 += 64 * UB) {
            float xv_[UB], g0_[UB], g1_[UB], nn_[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + 64 * u;
                const int d = i - pad;
                int q = (int)floorf((float)d * inv_hop);
                int n = d - q * hop;
                if (n < 0) { n += hop; q -= 1; }
                if (n >= hop) { n -= hop; q += 1; }
                int ft = f0 + q;
                if (ft > F - 2) { n += (ft - (F - 2)) * hop; ft = F - 2; }
                if (ft < 0) { ft = 0; n = 0; }   // t < 0: the sample is zero padding anyway
                g0_[u] = gb[ft];
                g1_[u] = gb[ft + 1];
                nn_[u] = (float)n;
                xv_[u] = xrow.ld(f0 * hop - pad + i);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + 64 * u;
                if (i < nu) stage[i] = xv_[u] * fmaf(nn_[u], (g1_[u] - g0_[u]) * inv_hop, g0_[u]);
            }
        }
    } else {
        constexpr int UB = 8;
        const BufRow wrow(window, Wl);
        for (int k0s = lane; k0s < Wl; k0s += 64 * UB) {   // 4 frames x UB positions per pass
            float xv_[4][UB], wv_[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0s + 64 * u;
                wv_[u] = wrow.ld(k);
#pragma unroll
                for (int ff = 0; ff < 4; ++ff) xv_[ff][u] = xrow.ld(k < Wl ? (f0 + ff) * hop - pad + k : -1);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0s + 64 * u;
                if (k < Wl) {
#pragma unroll
                    for (int ff = 0; ff < 4; ++ff) stage[ff * Wl + k] = xv_[ff][u] * wv_[u];
                }
            }
        }
    }
    // ---- set-up: impulse response and look-ahead rows of this lane's frame
    // (in double: the rows of C^r cancel heavily while they are built; rounded to fp32 once, they carry only their own
    //  rounding into the main loop -- fp32 set-up: 5e-3 instead of 1e-4 on the order-22 filter with poles at 0.97)
    FF_SETUP_T av[NT], v[NT];
    {
        // (unconditional loads from a clamped index, the select after the conversion: `i < M ? (double)pa[i] : 0` became NT
        //  conditional loads, each waited for -- 22 serial round trips.  No buffer descriptor: the four frames of a wave have
        //  four bases, a descriptor has to be wave-uniform.)
        const float* pa = a + ((size_t)b * F + fc) * M;
        float af[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) af[i] = pa[i < M ? i : M - 1];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const FF_SETUP_T w = (FF_SETUP_T)af[i];
            av[i] = i < M ? w : (FF_SETUP_T)0;
            v[i] = i == 0 ? (FF_SETUP_T)1 : (FF_SETUP_T)0;
        }
    }
    hs[fr][r] = 0.f;
    yr[fr][r] = 0.0;
    yr[fr][16 + r] = 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (r == 0) hs[fr][16 + s] = (float)v[0];             // h[s] = (C^s)[0][0]
        const FF_SETUP_T v0 = v[0];
#pragma unroll
        for (int j = 0; j + 1 < NT; ++j) v[j] = __builtin_fma(-av[j], v0, v[j + 1]);
        v[NT - 1] = -av[NT - 1] * v0;
        if (r == 0) {                                         // row s of G = e_0^T C^(s+1)
#pragma unroll
            for (int j2 = 0; j2 < NS / 2; ++j2) {
                double2 w2;
                w2.x = 2 * j2 < NT ? v[(2 * j2) < NT ? 2 * j2 : 0] : 0.0;
                w2.y = 2 * j2 + 1 < NT ? v[(2 * j2 + 1) < NT ? 2 * j2 + 1 : 0] : 0.0;
                *reinterpret_cast<double2*>(&Gs[fr][s][2 * j2]) = w2;
            }
        }
    }
    wave_lds_fence();
    double Grow[NS];
    float hrow[16];
#pragma unroll
    for (int j2 = 0; j2 < NS / 2; ++j2) {
        const double2 w2 = *reinterpret_cast<const double2*>(&Gs[fr][r][2 * j2]);
        Grow[2 * j2] = w2.x; Grow[2 * j2 + 1] = w2.y;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) hrow[m] = hs[fr][16 + r - m];   // h[r - m], 0 for m > r
    // ---- conditioning tier of the wave: kappa = the largest row sum of |G| among its 4 frames.  The feedback part G . state
    // cancels like the direct form's coefficient sum does, sixteen-fold: in fp32 it is as accurate as the direct form while
    // kappa stays small and loses up to 1.5 digits on badly conditioned frames, so those waves -- and only
    // those -- keep the ring, G and the feedback sum in double.
    float kap = 0.f;
#pragma unroll
    for (int j = 0; j < NS; ++j) kap += fabsf((float)Grow[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) kap = fmaxf(kap, __shfl_xor(kap, o));
    const bool precise = __builtin_amdgcn_ballot_w64(!(kap <= kappa_max)) != 0ull;   // wave-uniform (NaN -> precise)
    // ---- main loop: two blocks per iteration (the output ring's phase is then a compile-time constant)
    const float* xsrc = REV ? stage + (size_t)fr * Wl : stage + (size_t)fr * hop;
    float* orow = wf + ((size_t)b * nfr + fc) * Wl;
    const int nblk = Wl / 16;
    auto xpart = [&](int k0) -> float {
        float xv[16];
        const float* xp = REV ? xsrc + (Wl - 16 - k0) : xsrc + k0;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            const float4 w4 = *reinterpret_cast<const float4*>(xp + 4 * m4);
            if (REV) { xv[15 - 4 * m4] = w4.x; xv[14 - 4 * m4] = w4.y; xv[13 - 4 * m4] = w4.z; xv[12 - 4 * m4] = w4.w; }
            else     { xv[4 * m4] = w4.x; xv[4 * m4 + 1] = w4.y; xv[4 * m4 + 2] = w4.z; xv[4 * m4 + 3] = w4.w; }
        }
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
        for (int m = 0; m < 16; m += 4) {
            acc0 = fmaf(hrow[m], xv[m], acc0);
            acc1 = fmaf(hrow[m + 1], xv[m + 1], acc1);
            acc2 = fmaf(hrow[m + 2], xv[m + 2], acc2);
            acc3 = fmaf(hrow[m + 3], xv[m + 3], acc3);
        }
        return (acc0 + acc1) + (acc2 + acc3);
    };
    // sv[j] = y[k0 - 1 - j] = ring[(k0 - 1 - j) & 31], k0 & 31 = 16 * par: the word at i0 = (16 par - 4 - 4 j4) & 31 holds
    // j = 4 j4 + 3 .. 4 j4 in ascending memory order
    if (precise) {
        for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int k0 = (blk + par) * 16;           // recursion index of the block's first sample
                double d0 = (double)xpart(k0), d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
                for (int j4 = 0; j4 < NS4; ++j4) {
                    const int i0 = ((16 * par - 4 - 4 * j4) & 31);
                    const double2 lo = *reinterpret_cast<const double2*>(&yr[fr][i0]);
                    const double2 hi = *reinterpret_cast<const double2*>(&yr[fr][i0 + 2]);
                    d0 = __builtin_fma(Grow[4 * j4], hi.y, d0);
                    d1 = __builtin_fma(Grow[4 * j4 + 1], hi.x, d1);
                    d2 = __builtin_fma(Grow[4 * j4 + 2], lo.y, d2);
                    d3 = __builtin_fma(Grow[4 * j4 + 3], lo.x, d3);
                }
                const double yd = (d0 + d1) + (d2 + d3);
                wave_lds_fence();                          // every lane has read the ring before it is overwritten
                yr[fr][16 * par + r] = yd;
                if (mine) orow[REV ? Wl - 1 - k0 - r : k0 + r] = (float)yd;
                wave_lds_fence();
            }
        }
    } else {
        float Gf[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) Gf[j] = (float)Grow[j];
        float* yrf = reinterpret_cast<float*>(&yr[fr][0]);   // the same ring, 32 floats (all zeros so far)
        for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int k0 = (blk + par) * 16;
                float a0 = xpart(k0), a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int j4 = 0; j4 < NS4; ++j4) {
                    const int i0 = ((16 * par - 4 - 4 * j4) & 31);
                    const float4 w4 = *reinterpret_cast<const float4*>(yrf + i0);
                    a0 = fmaf(Gf[4 * j4], w4.w, a0);
                    a1 = fmaf(Gf[4 * j4 + 1], w4.z, a1);
                    a2 = fmaf(Gf[4 * j4 + 2], w4.y, a2);
                    a3 = fmaf(Gf[4 * j4 + 3], w4.x, a3);
                }
                const float y = (a0 + a1) + (a2 + a3);
                wave_lds_fence();
                yrf[16 * par + r] = y;
                if (mine) orow[REV ? Wl - 1 - k0 - r : k0 + r] = y;
                wave_lds_fence();
            }
        }
    }
}

// ---- cascade of second-order sections ---------------------------------------------------------
// BatchSecondOrderLPCSynth.forward, models/lpc.py:94-131: every frame runs through K all-pole biquads
// 1/(a0 + a1 z^-1 + a2 z^-2) one after the other.  A cascade is a pipeline: section k can work on sample m while section
// k+1 works on sample m-1.  Here the pipeline is laid ACROSS LANES: a frame owns a 16-lane DPP row, lane k is section
// k, and at step n it filters sample n-k, taking its input from lane k-1's output of the previous step (row_shr:1).
// The loop-carried dependency is one DPP move + one FMA (the state part  -a1*s1 - a2*s2  does not wait for the
// input), against K dependent sections per sample if one lane ran the whole cascade, or the 6-FMA + 2-DPP reduction
// chain of the direct form (ff_framesq_kernel).  4 frames per wave; frames are staged in LDS (gain applied) and the
// last section's outputs collected there for a coalesced write-out.
constexpr int BQ_ROW = 16;   // lanes per frame (K <= 16)
constexpr int BQ_FPW = 64 / BQ_ROW;
#define DPP_ROW_SHR1 0x111
__global__ __launch_bounds__(64) void ff_biquad_frames_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                              const float* __restrict__ gain,
                                                              const float* __restrict__ bq, float* __restrict__ wf,
                                                              int Tx, int F, int K, int hop, int Wl, int pad, int nfr,
                                                              int gain_mode, int WS, int XS) {
    extern __shared__ __attribute__((aligned(16))) float bq_lds[];
    float* xin = bq_lds;                      // [XS]: union of the wave's frames
    float* yout = bq_lds + XS;                // [BQ_FPW][WS]
    const int lane = threadIdx.x, b = blockIdx.y;
    const int slot = lane >> 4, k = lane & 15;
    const int f0 = blockIdx.x * BQ_FPW;
    const BufRow xrow(ex + (size_t)b * ex_stride, Tx);
    const float* gb = gain + (size_t)b * F;
    const float inv_hop = 1.0f / (float)hop;
    // ---- stage the input: the wave's BQ_FPW consecutive frames overlap (hop apart), so their union
    // [f0*hop - pad, f0*hop - pad + (BQ_FPW-1)*hop + Wl) is staged ONCE, gain applied; frame s starts at s*hop.
    // (per-frame gain mode scales a frame as a whole, which the section of lane 0 does: its g carries the gain.)
    // Eight elements per lane at a time, every load unconditional (clamped index, mask applied to the value): a guarded
    // loop made hipcc wait for each of its loads in turn (145 us for this kernel instead of 40).
    {
        const int tb = f0 * hop - pad;
        for (int i0 = 0; i0 < XS; i0 += 8 * 64) {
            float xv[8], ga[8], gd[8];
            int nn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = tb + i0 + u * 64 + lane;
                xv[u] = xrow.ld(max(t, -1));
                int ft = max(t, 0) / hop;
                ft = max(min(ft, F - 2), 0);
                nn[u] = t - ft * hop;
                ga[u] = gain_mode == 0 ? gb[ft] : 1.f;
                gd[u] = gain_mode == 0 ? gb[min(ft + 1, F - 1)] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 64 + lane;
                const float G = gain_mode == 1 ? 1.f : fmaf((float)nn[u], (gd[u] - ga[u]) * inv_hop, ga[u]);
                if (i < XS) xin[i] = xv[u] * G;
            }
        }
    }
    // ---- this lane's section
    const int f = f0 + slot;
    float a1 = 0.f, a2 = 0.f, g = 1.f;
    if (k < K && f < nfr) {
        const float* c = bq + (((size_t)b * F + f) * K + k) * 3;
        const float ia0 = 1.0f / c[0];
        g = ia0 * ((gain_mode == 1 && k == 0) ? gb[f] : 1.0f);
        a1 = c[1] * ia0;
        a2 = c[2] * ia0;
    }
    wave_lds_fence();
    const float* xs = xin + slot * hop;      // frame `slot` of this wave inside the staged union
    float* ys = yout + slot * WS + BQ_ROW;   // ys[m], m in [-(K-1), Wl + 3]
    float s1 = 0.f, s2 = 0.f, outp = 0.f;   // outp = this lane's output of the previous step
    const int nsteps = Wl + K - 1;
    const bool first = k == 0, last = k == K - 1;
    // four steps of the pipeline on the inputs x4; the last section's outputs go to ys[n0 - (K-1) ...]
    auto steps4 = [&](const float4 x4, int n0) {
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        float yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = fmaf(-a1, s1, -a2 * s2);          // does not depend on this step's input
            // the shift is executed by ALL lanes (a DPP read from a lane that sits out of a branch returns the old
            // value): select afterwards.  bound_ctrl: lane 0 of a row reads 0.
            const float sh = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, outp), DPP_ROW_SHR1, 0xF, 0xF, true));
            const float in = first ? xv[j] : sh;
            const float y = fmaf(g, in, p);
            s2 = s1;
            s1 = y;
            outp = y;
            yv[j] = y;
        }
        if (last) {  // yout rows carry BQ_ROW floats of slack in front: the pipeline's fill outputs land there
#pragma unroll
            for (int j = 0; j < 4; ++j) ys[n0 + j - (K - 1)] = yv[j];
        }
    };
    // lane 0 of the row reads its next four input samples (zeros past the frame: the tail is staged as 0) one block of
    // steps AHEAD, into the other of two register quads: the LDS latency stays off the recursion's critical path
    const float4* xs4 = reinterpret_cast<const float4*>(xs);
    const int nblk4 = (nsteps + 3) / 4, last4 = (XS - slot * hop) / 4 - 1;
    float4 xa = xs4[0], xb;
    for (int q = 0; q < nblk4; q += 2) {
        xb = xs4[min(q + 1, last4)];
        steps4(xa, 4 * q);
        xa = xs4[min(q + 2, last4)];
        if (q + 1 < nblk4) steps4(xb, 4 * q + 4);
    }
    wave_lds_fence();
    // ---- coalesced write-out of the filtered frames
    for (int s_ = 0; s_ < BQ_FPW; ++s_) {
        const int fo = f0 + s_;
        if (fo >= nfr) break;
        float* o = wf + ((size_t)b * nfr + fo) * Wl;
        for (int i = lane; i < Wl; i += 64) o[i] = yout[s_ * WS + BQ_ROW + i];
    }
}

// ---- backward -----------------------------------------------------------------------------------------------
// B0: g_q[b,n] = gy[b,n] / norm[n]
__global__ void ff_gq_kernel(const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ window,
                             float* __restrict__ gq, int B, int Ty, int hop, int Wl, int nfr) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Ty) return;
    const int b = (int)(idx / Ty), n = (int)(idx - (int64_t)b * Ty);
    const int m = n + Wl / 2;
    int fhi = m / hop;
    if (fhi > nfr - 1) fhi = nfr - 1;
    int flo = (m - Wl + hop) / hop;
    if (m - Wl + 1 <= 0) flo = 0;
    float norm = 0.f;
    const float gv = gy[(size_t)b * gy_stride + n];
    if (Wl <= 4 * hop) {   // (loads issued together: see ff_ola_kernel)
        float wk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = flo + u, k = m - f * hop;
            const bool ok = f <= fhi && k >= 0 && k < Wl;
            wk[u] = window[ok ? k : 0];
            if (!ok) wk[u] = 0.f;
        }
        norm = (wk[0] + wk[1]) + (wk[2] + wk[3]);
    } else {
        for (int f = flo; f <= fhi; ++f) {
            const int k = m - f * hop;
            if (k >= 0 && k < Wl) norm += window[k];
        }
    }
    gq[idx] = gv / norm;
}

// B2: g_a[b,f,i] = -sum_k u_f[k] * y_f[k-1-i].  One wave per frame; both rows staged in LDS (y_f behind NT zeros).
template <int NT>
__global__ __launch_bounds__(256) void ff_grad_a_kernel(const float* __restrict__ uf, const float* __restrict__ yf,
                                                        float* __restrict__ g_a, int F, int M, int Wl, int nfr,
                                                        int nq, int RS) {
    extern __shared__ __attribute__((aligned(16))) float ga_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x * 4 + wv;   // (b, f) over ALL F coefficient frames: unused frames get zeros
    if (q >= nq) return;
    const int b = q / F, f = q - b * F;
    float acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = 0.f;
    if (f < nfr) {
        float* us = ga_lds + wv * RS;          // Wl floats
        float* ys = us + Wl;                   // NT zeros, then Wl floats
        const size_t base = ((size_t)b * nfr + f) * Wl;
        constexpr int UB = 8;   // (loads of a batch issued before the first LDS write: 15 serial round trips otherwise)
        for (int k0 = lane; k0 < Wl; k0 += 64 * UB) {
            float uv[UB], yv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0 + 64 * u, kc = k < Wl ? k : 0;
                uv[u] = uf[base + kc];
                yv[u] = yf[base + kc];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = k0 + 64 * u;
                if (k < Wl) { us[k] = uv[u]; ys[NT + k] = yv[u]; }
            }
        }
        if (lane < NT) ys[lane] = 0.f;
        wave_lds_fence();
        for (int k = lane; k < Wl; k += 64) {
            const float u = us[k];
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = fmaf(u, ys[NT + k - 1 - i], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[i] = v;
    }
    if (lane == 0) {
        float* o = g_a + (size_t)q * M;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (i < M) o[i] = -acc[i];
    }
}

// B3: g_x[t] = sum_f u_f[t + pad - f*hop]; g_ex = g_x * G(t); per gain segment s (t in [s*hop, (s+1)*hop), the last
// one also takes t = (F-1)*hop) the hat-weighted sums P0 = sum (1-w) g_x ex, P1 = sum w g_x ex.
__global__ __launch_bounds__(256) void ff_bwd_ola_kernel(const float* __restrict__ uf, const float* __restrict__ ex,
                                                         int64_t ex_stride, const float* __restrict__ gain,
                                                         float* __restrict__ g_ex, int64_t g_ex_stride,
                                                         float* __restrict__ part, int Tx, int Tfull, int F, int hop,
                                                         int Wl, int nfr, int g_ex_len) {
    __shared__ float red0[256], red1[256];
    const int sgm = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (sgm == F - 2)   // the excitation beyond the last coefficient frame reaches no output: its gradient is zero (the
        for (int t = Tfull + tid; t < g_ex_len; t += 256) g_ex[(size_t)b * g_ex_stride + t] = 0.f;   // caller used to fill it)
    const int pad = Wl / 2;
    const float g0 = gain[(size_t)b * F + sgm], g1 = gain[(size_t)b * F + sgm + 1];
    const float inv_hop = 1.0f / (float)hop;
    const int t_lo = sgm * hop;
    const int t_hi = (sgm == F - 2) ? t_lo + hop + 1 : t_lo + hop;  // exclusive
    float p0 = 0.f, p1 = 0.f;
    for (int t = t_lo + tid; t < t_hi && t < Tfull; t += 256) {
        float gx = 0.f;
        if (t < Tx) {
            const int m = t + pad;
            int fhi = m / hop;
            if (fhi > nfr - 1) fhi = nfr - 1;
            int flo = (m - Wl + hop) / hop;
            if (m - Wl + 1 <= 0) flo = 0;
            const float e = ex[(size_t)b * ex_stride + t];
            if (Wl <= 4 * hop) {   // (loads issued together: see ff_ola_kernel)
                float uv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int f = flo + u, k = m - f * hop;
                    const bool ok = f <= fhi && k >= 0 && k < Wl;
                    uv[u] = uf[((size_t)b * nfr + (ok ? f : flo)) * Wl + (ok ? k : 0)];
                    if (!ok) uv[u] = 0.f;
                }
                gx = ((uv[0] + uv[1]) + uv[2]) + uv[3];
            } else {
                for (int f = flo; f <= fhi; ++f) {
                    const int k = m - f * hop;
                    if (k >= 0 && k < Wl) gx += uf[((size_t)b * nfr + f) * Wl + k];
                }
            }
            const float w = (float)(t - t_lo) * inv_hop;
            g_ex[(size_t)b * g_ex_stride + t] = gx * fmaf(w, g1 - g0, g0);
            p0 = fmaf((1.0f - w) * gx, e, p0);
            p1 = fmaf(w * gx, e, p1);
        } else {
            g_ex[(size_t)b * g_ex_stride + t] = 0.f;
        }
    }
    red0[tid] = p0;
    red1[tid] = p1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red0[tid] += red0[tid + off]; red1[tid] += red1[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        part[((size_t)b * (F - 1) + sgm) * 2 + 0] = red0[0];
        part[((size_t)b * (F - 1) + sgm) * 2 + 1] = red1[0];
    }
}

__global__ void ff_gain_reduce_kernel(const float* __restrict__ part, float* __restrict__ g_gain, int B, int F) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * F) return;
    const int b = idx / F, f = idx - b * F;
    float v = 0.f;
    if (f < F - 1) v += part[((size_t)b * (F - 1) + f) * 2 + 0];
    if (f >= 1) v += part[((size_t)b * (F - 1) + f - 1) * 2 + 1];
    g_gain[idx] = v;
}

struct FfBwdPlan {
    size_t off_gq, off_uf, off_part, total;
};
static FfBwdPlan ff_bwd_plan(int B, int F, int Wl, int nfr, int Ty) {
    FfBwdPlan p;
    size_t o = 0;
    p.off_gq = o;   o += align_up(sizeof(float) * (size_t)B * Ty, 256);
    p.off_uf = o;   o += align_up(sizeof(float) * (size_t)B * nfr * Wl, 256);
    p.off_part = o; o += align_up(sizeof(float) * (size_t)B * (F - 1) * 2, 256);
    p.total = o;
    return p;
}

// Row sum of |G| up to which a wave of the block-recursion kernel keeps its feedback part in fp32 (dev knob GOLF_FF_KAPPA)
static float ff_kappa_max() {
    static const float v = [] { const char* e = getenv("GOLF_FF_KAPPA"); return e ? (float)atof(e) : 64.f; }();
    return v;
}
// The block-recursion kernel's conditions (dev knob GOLF_FF_QUADS=1: the direct-form quad kernel, A/B)
static bool ff_block_ok(int Wl, int hop, size_t lds_bytes) {
    static const bool quads = [] { const char* e = getenv("GOLF_FF_QUADS"); return e && atoi(e) != 0; }();
    return !quads && Wl % 32 == 0 && hop % 4 == 0 && lds_bytes <= 56 * 1024;
}

template <int W, int NT>
static int launch_ff_bwd(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride, const float* gain,
                         const float* a, const float* window, float* g_ex, int64_t g_ex_stride, float* g_gain,
                         float* g_a, int B, int Tx, int Tfull, int F, int M, int hop, int Wl, int Ty, int nfr,
                         const float* yf, char* ws, hipStream_t st, int g_ex_len) {
    if (Wl % W != 0 || (int64_t)nfr * Wl >= (1ll << 29) || Wl > 16384)
        return fail(GOLF_EUNSUPPORTED, "lti_frames_bwd: window length %d must be a multiple of the ring width %d "
                    "(and <= 16384)", Wl, W);
    const FfBwdPlan p = ff_bwd_plan(B, F, Wl, nfr, Ty);
    float* gq = (float*)(ws + p.off_gq);
    float* uf = (float*)(ws + p.off_uf);
    float* part = (float*)(ws + p.off_part);
    const int64_t n = (int64_t)B * Ty;
    hipLaunchKernelGGL(ff_gq_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, gy, gy_stride, window, gq, B,
                       Ty, hop, Wl, nfr);
    GOLF_LAUNCH_CHECK();
    bool blocked = false;
    if constexpr (NT <= 24) {
        const size_t ldsb = sizeof(float) * 4 * (size_t)Wl;
        // the adjoint stages 4 whole frames of window * g_q per wave instead of their union: 56 us against the quads' 45 at
        // B = 32 -- measured, so the backward keeps the quad kernel unless GOLF_FF_BLOCK_BWD=1 (A/B)
        static const bool bwd_block = [] { const char* e = getenv("GOLF_FF_BLOCK_BWD"); return e && atoi(e) != 0; }();
        if (bwd_block && ff_block_ok(Wl, hop, ldsb)) {   // block recursion (ff_framesb_kernel)
            static const hipError_t attr = hipFuncSetAttribute((const void*)ff_framesb_kernel<NT, true>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (attr != hipSuccess) return fail((int)attr, "lti_frames_bwd: cannot raise the dynamic LDS limit");
            hipLaunchKernelGGL((ff_framesb_kernel<NT, true>), dim3((unsigned)ceil_div(nfr, 4), B), dim3(64), ldsb, st,
                               (const float*)gq, (int64_t)Ty, gain, a, window, uf, Ty, F, M, hop, Wl, nfr, ff_kappa_max());
            blocked = true;
        }
    }
    if (!blocked)
        hipLaunchKernelGGL((ff_framesq_kernel<W, NT, true>), dim3((unsigned)ceil_div(nfr, 16), B), dim3(64),
                           sizeof(float) * (size_t)Wl, st, (const float*)gq, (int64_t)Ty, gain, a, window, uf, Ty, F, M,
                           hop, Wl, nfr);
    GOLF_LAUNCH_CHECK();
    const int nq = B * F;
    const int RS = 2 * Wl + NT + 8;
    hipLaunchKernelGGL((ff_grad_a_kernel<NT>), dim3((unsigned)ceil_div(nq, 4)), dim3(256), 4 * RS * sizeof(float), st,
                       (const float*)uf, yf, g_a, F, M, Wl, nfr, nq, RS);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ff_bwd_ola_kernel, dim3((unsigned)(F - 1), B), dim3(256), 0, st, (const float*)uf, ex,
                       ex_stride, gain, g_ex, g_ex_stride, part, Tx, Tfull, F, hop, Wl, nfr, g_ex_len);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ff_gain_reduce_kernel, dim3((unsigned)ceil_div(B * F, 256)), dim3(256), 0, st,
                       (const float*)part, g_gain, B, F);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

template <int W, int NT>
static int launch_ff(const float* ex, int64_t ex_stride, const float* gain, const float* a, const float* window,
                     float* y, int64_t y_stride, int B, int Tx, int F, int M, int hop, int Wl, int Ty, int nfr,
                     float* wf, hipStream_t st) {
    const int nq = B * nfr;
    bool blocked = false;
    if constexpr (NT <= 24) {
        const size_t ldsb = sizeof(float) * (3 * (size_t)hop + (size_t)Wl);
        if (ff_block_ok(Wl, hop, ldsb)) {   // block recursion (ff_framesb_kernel)
            static const hipError_t attr = hipFuncSetAttribute((const void*)ff_framesb_kernel<NT, false>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (attr != hipSuccess) return fail((int)attr, "lti_frames_ola: cannot raise the dynamic LDS limit");
            hipLaunchKernelGGL((ff_framesb_kernel<NT, false>), dim3((unsigned)ceil_div(nfr, 4), B), dim3(64), ldsb, st, ex,
                               ex_stride, gain, a, window, wf, Tx, F, M, hop, Wl, nfr, ff_kappa_max());
            blocked = true;
        }
    }
    if (blocked) {
    } else if (Wl % W == 0 && (int64_t)nfr * Wl < (1ll << 29) && Wl <= 32768) {
        hipLaunchKernelGGL((ff_framesq_kernel<W, NT, false>), dim3((unsigned)ceil_div(nfr, 16), B), dim3(64), 0, st, ex,
                           ex_stride, gain, a, window, wf, Tx, F, M, hop, Wl, nfr);
    } else {
        hipLaunchKernelGGL((ff_frames_kernel<W, NT>), dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, st, ex, ex_stride,
                           gain, a, window, wf, Tx, F, M, hop, Wl, nfr, nq);
    }
    GOLF_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * Ty;
    hipLaunchKernelGGL(ff_ola_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)wf, window, y,
                       y_stride, B, Ty, hop, Wl, nfr, Wl / 2);
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

extern "C" size_t golf_lti_frames_bwd_workspace_bytes(int B, int Tx, int F, int M, int hop, int W) {
    if (B < 1 || Tx < 1 || F < 2 || hop < 1 || W < 1) return 0;
    int nfr, Ty;
    ff_geometry(Tx, F, hop, W, &nfr, &Ty);
    if (nfr < 1) return 256;
    return ff_bwd_plan(B, F, W, nfr, Ty).total;
}

extern "C" int golf_lti_frames_ola_bwd_f32(const float* gy, int64_t gy_stride, const float* ex, int64_t ex_stride,
                                           const float* gain, const float* a, const float* window, float* g_ex,
                                           int64_t g_ex_stride, int g_ex_len, float* g_gain, float* g_a, int B,
                                           int Tx, int F, int M, int hop, int W, int Ty, const void* ws_fwd,
                                           void* ws, size_t ws_bytes, void* stream) {
    if (B < 1 || Tx < 1 || F < 2 || M < 1 || hop < 1 || W < 1)
        return fail(GOLF_EINVAL, "lti_frames_bwd: bad size (need F >= 2)");
    if (!gy || !ex || !gain || !a || !window || !g_ex || !g_gain || !g_a || !ws_fwd)
        return fail(GOLF_EINVAL, "lti_frames_bwd: null pointer");
    if (W < 2 * hop) return fail(GOLF_EINVAL, "lti_frames_bwd: window %d < 2*hop %d", W, 2 * hop);
    if ((int64_t)Tx > (int64_t)(F - 1) * hop + 1) return fail(GOLF_EINVAL, "lti_frames_bwd: Tx exceeds (F-1)*hop+1");
    int nfr, ty;
    ff_geometry(Tx, F, hop, W, &nfr, &ty);
    if (nfr < 1 || nfr > F) return fail(GOLF_EINVAL, "lti_frames_bwd: %d frames vs %d coefficient frames", nfr, F);
    if (ty != Ty) return fail(GOLF_EINVAL, "lti_frames_bwd: Ty=%d, expected %d", Ty, ty);
    if (g_ex_len < Tx) return fail(GOLF_EINVAL, "lti_frames_bwd: g_ex_len %d < Tx %d", g_ex_len, Tx);
    const size_t need = ff_bwd_plan(B, F, W, nfr, Ty).total;
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "lti_frames_bwd: workspace needs %zu bytes, 256-aligned (got %zu)", need,
                    ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    // g_ex is written on [0, g_ex_len): zeros beyond Tx and beyond the last coefficient frame's sample (F-1)*hop
    int Tfull = (F - 1) * hop + 1;
    if (Tfull > g_ex_len) Tfull = g_ex_len;
#define GOLF_FF_TRY(w, nt)                                                                                     \
    if (M <= (nt) && (w) <= hop)                                                                               \
        return launch_ff_bwd<w, nt>(gy, gy_stride, ex, ex_stride, gain, a, window, g_ex, g_ex_stride, g_gain, g_a, B, \
                                    Tx, Tfull, F, M, hop, W, Ty, nfr, (const float*)ws_fwd, (char*)ws, st, g_ex_len);
    GOLF_FF_TRY(8, 6)
    GOLF_FF_TRY(16, 14)
    GOLF_FF_TRY(24, 22)
    GOLF_FF_TRY(32, 30)
    GOLF_FF_TRY(40, 38)
#undef GOLF_FF_TRY
    return fail(GOLF_EUNSUPPORTED, "lti_frames_bwd: need M <= 38 and hop >= ring width (M=%d hop=%d)", M, hop);
}

// ---- backward of the cascade (the reference is differentiable through its K lfilter calls,
// models/lpc.py:115-118).  One wave per frame; both cascades are laid across lanes like the forward's:
//   phase 1  re-runs the forward cascade and keeps EVERY section's output in LDS: Y[0] = the gain-scaled input frame,
//            Y[k+1] = output of section k (the correlations of phase 2 need them);
//   phase 2  the adjoint cascade: the adjoint of an LTI all-pole section is the same section run BACKWARDS in time,
//            u_{k-1}[n] = (u_k[n] - a1 u_{k-1}[n+1] - a2 u_{k-1}[n+2]) / a0, fed at the last section with
//            u_K = window * g_q of the frame; lane k takes its input from lane k+1's previous output (row_shl:1) and
//            accumulates d/d(a0,a1,a2) of its section = -sum_n u_{k-1}[n] * y_k[n - i];
//   lane 0's output u_0 is the gradient w.r.t. the scaled input frame: stored (times the frame gain in gain mode 1) for the
//   overlap-add kernel below; gain mode 1 also reduces g_gain[b,f] = sum_n u_0[n] x[n] here.
#define DPP_ROW_SHL1 0x101
__global__ __launch_bounds__(64) void ff_biquad_bwd_kernel(const float* __restrict__ gq, int64_t gq_stride,
                                                           const float* __restrict__ ex, int64_t ex_stride,
                                                           const float* __restrict__ gain, const float* __restrict__ bq,
                                                           const float* __restrict__ window, float* __restrict__ ufr,
                                                           float* __restrict__ g_bq, float* __restrict__ g_gain_f,
                                                           int Tx, int Ty, int F, int K, int hop, int Wl, int pad,
                                                           int nfr, int gain_mode) {
    extern __shared__ __attribute__((aligned(16))) float bqb_lds[];
    const int WS = Wl + 4;
    float* Y = bqb_lds;                    // [(K+1)][WS], row k at Y + k*WS + 2 (two zeros in front: y[-1], y[-2])
    float* U = bqb_lds + (size_t)(K + 1) * WS;   // [WS]: u_K on the way in, u_0 on the way out
    const int lane = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
    const int k = lane;
    const float* gb = gain + (size_t)b * F;
    const float inv_hop = 1.0f / (float)hop;
    const int tb = f * hop - pad;          // input sample of frame position 0
    for (int i = lane; i < (K + 1) * WS; i += 64) Y[i] = 0.f;
    wave_lds_fence();
    for (int n = lane; n < Wl; n += 64) {
        const int t = tb + n;
        float xv = (t >= 0 && t < Tx) ? ex[(size_t)b * ex_stride + t] : 0.f;
        float G = 1.f;
        if (gain_mode == 0) {
            int ft = max(t, 0) / hop;
            ft = max(min(ft, F - 2), 0);
            G = fmaf((float)(t - ft * hop), (gb[min(ft + 1, F - 1)] - gb[ft]) * inv_hop, gb[ft]);
        } else {
            G = gb[f];
        }
        Y[2 + n] = xv * G;
        const int to = tb + n;              // output sample index of frame position n: f*hop - pad + n
        U[n] = (to >= 0 && to < Ty) ? window[n] * gq[(size_t)b * gq_stride + to] : 0.f;
    }
    float a1 = 0.f, a2 = 0.f, ia0 = 1.f;
    if (k < K) {
        const float* c = bq + (((size_t)b * F + f) * K + k) * 3;
        ia0 = 1.0f / c[0];
        a1 = c[1] * ia0;
        a2 = c[2] * ia0;
    }
    wave_lds_fence();
    // ---- phase 1: forward cascade, lane k = section k filters sample m - k at step m
    {
        float s1 = 0.f, s2 = 0.f, outp = 0.f;
        for (int m = 0; m < Wl + K - 1; ++m) {
            const float sh = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, outp), DPP_ROW_SHR1, 0xF, 0xF, true));
            const int n = m - k;
            const bool on = k < K && n >= 0 && n < Wl;
            const float in = k == 0 ? Y[2 + (on ? n : 0)] : sh;
            const float y = on ? fmaf(ia0, in, fmaf(-a1, s1, -a2 * s2)) : 0.f;
            if (on) { s2 = s1; s1 = y; Y[(size_t)(k + 1) * WS + 2 + n] = y; }
            outp = y;
        }
    }
    wave_lds_fence();
    // ---- phase 2: adjoint cascade in reverse time, lane k = section k handles sample Wl-1 - (m - (K-1-k)) at step m
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    {
        float s1 = 0.f, s2 = 0.f, outp = 0.f;
        const float* yk = Y + (size_t)(k < K ? k + 1 : 0) * WS + 2;   // this section's OUTPUT
        for (int m = 0; m < Wl + K - 1; ++m) {
            const float sh = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, outp), DPP_ROW_SHL1, 0xF, 0xF, true));
            const int n = Wl - 1 - (m - (K - 1 - k));
            const bool on = k < K && n >= 0 && n < Wl;
            const float in = k == K - 1 ? U[on ? n : 0] : sh;
            const float u = on ? fmaf(ia0, in, fmaf(-a1, s1, -a2 * s2)) : 0.f;   // (a1, a2 already divided by a0)
            if (on) {
                s2 = s1; s1 = u;
                g0 = fmaf(-u, yk[n], g0);
                g1 = fmaf(-u, yk[n - 1], g1);
                g2 = fmaf(-u, yk[n - 2], g2);
            }
            outp = u;
            // lane 0's outputs replace U behind the read front of lane K-1 (it is K-1 samples ahead): no hazard, U[n] of
            // lane K-1 at this step has index n - (K-1) < n of lane 0
            if (on && k == 0) U[n] = u;
        }
    }
    if (k < K) {
        float* o = g_bq + (((size_t)b * F + f) * K + k) * 3;
        o[0] = g0; o[1] = g1; o[2] = g2;
    }
    wave_lds_fence();
    // ---- u_0 -> frame gradient store; gain mode 1: g_gain[b,f] = sum_n u_0[n] * x[n]
    float gg = 0.f;
    const float gf = gain_mode == 1 ? gb[f] : 1.f;
    float* uo = ufr + ((size_t)b * nfr + f) * Wl;
    for (int n = lane; n < Wl; n += 64) {
        const float u0 = U[n];
        uo[n] = u0 * gf;
        const int t = tb + n;
        if (gain_mode == 1 && t >= 0 && t < Tx) gg = fmaf(u0, ex[(size_t)b * ex_stride + t], gg);
    }
    if (gain_mode == 1) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) gg += __shfl_xor(gg, off);
        if (lane == 0) g_gain_f[(size_t)b * F + f] = gg;
    }
}

// overlap-add of the frame gradients: g_x[b,t] = sum_f ufr[b,f,t - f*hop + pad]; gain mode 0 multiplies by up(gain)[t]
// (g_ex) and also writes g_x * ex (the host folds it onto the gain frames: up^T).
__global__ void ff_biquad_bwd_ola_kernel(const float* __restrict__ ufr, const float* __restrict__ ex, int64_t ex_stride,
                                         const float* __restrict__ gain, float* __restrict__ g_ex, int64_t g_ex_stride,
                                         float* __restrict__ gxe, int B, int Tx, int F, int hop, int Wl, int pad, int nfr,
                                         int gain_mode) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Tx) return;
    const int b = (int)(idx / Tx), t = (int)(idx - (int64_t)b * Tx);
    const int m = t + pad;
    int fhi = m / hop;
    if (fhi > nfr - 1) fhi = nfr - 1;
    int flo = (m - Wl + hop) / hop;
    if (m - Wl + 1 <= 0) flo = 0;
    if (flo < 0) flo = 0;
    float acc = 0.f;
    for (int f = flo; f <= fhi; ++f) {
        const int k = m - f * hop;
        if (k < 0 || k >= Wl) continue;
        acc += ufr[((size_t)b * nfr + f) * Wl + k];
    }
    if (gain_mode == 0) {
        int ft = t / hop;
        ft = max(min(ft, F - 2), 0);
        const float* gb = gain + (size_t)b * F;
        const float G = fmaf((float)(t - ft * hop), (gb[min(ft + 1, F - 1)] - gb[ft]) / (float)hop, gb[ft]);
        g_ex[(size_t)b * g_ex_stride + t] = acc * G;
        gxe[(size_t)b * Tx + t] = acc * ex[(size_t)b * ex_stride + t];
    } else {
        g_ex[(size_t)b * g_ex_stride + t] = acc;
    }
}

extern "C" int golf_biquad_frames_ola_fwd_f32(const float* ex, int64_t ex_stride, const float* gain,
                                              const float* biquads, const float* window, float* y, int64_t y_stride,
                                              int B, int Tx, int F, int K, int hop, int W, int pad, int gain_mode,
                                              int Ty, void* ws, size_t ws_bytes, void* stream) {
    if (B < 1 || Tx < 1 || F < 1 || K < 1 || hop < 1 || W < 1 || pad < 0 || (gain_mode != 0 && gain_mode != 1))
        return fail(GOLF_EINVAL, "biquad_frames: bad size / mode");
    if (gain_mode == 0 && F < 2) return fail(GOLF_EINVAL, "biquad_frames: interpolated gain needs F >= 2");
    if (!ex || !gain || !biquads || !window || !y) return fail(GOLF_EINVAL, "biquad_frames: null pointer");
    if (K > BQ_ROW) return fail(GOLF_EUNSUPPORTED, "biquad_frames: %d sections > %d (one DPP row per frame)", K, BQ_ROW);
    if (hop % 4 != 0) return fail(GOLF_EUNSUPPORTED, "biquad_frames: hop=%d must be a multiple of 4", hop);
    if (Tx + 2 * pad < W) return fail(GOLF_EINVAL, "biquad_frames: signal shorter than one frame");
    const int nfr = (Tx + 2 * pad - W) / hop + 1;
    const int ty = (nfr - 1) * hop + W - 2 * pad;
    if (nfr > F) return fail(GOLF_EINVAL, "biquad_frames: %d frames vs %d coefficient frames", nfr, F);
    if (ty != Ty || ty < 1) return fail(GOLF_EINVAL, "biquad_frames: Ty=%d, expected %d", Ty, ty);
    if (ex_stride < Tx || y_stride < Ty) return fail(GOLF_EINVAL, "biquad_frames: row stride too small");
    const size_t need = align_up(sizeof(float) * (size_t)B * nfr * W, 256);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "biquad_frames: workspace needs %zu bytes, 256-aligned (got %zu)", need, ws_bytes);
    const int WS = ((W + 3) & ~3) + 2 * BQ_ROW;
    const int XS = (((BQ_FPW - 1) * hop + W + 2 * BQ_ROW + 8) + 3) & ~3;
    const size_t lds = sizeof(float) * ((size_t)XS + (size_t)BQ_FPW * WS);
    if (lds > 60 * 1024) return fail(GOLF_EUNSUPPORTED, "biquad_frames: window %d too long for the LDS staging", W);
    hipStream_t st = (hipStream_t)stream;
    float* wf = (float*)ws;
    hipLaunchKernelGGL(ff_biquad_frames_kernel, dim3((unsigned)ceil_div(nfr, BQ_FPW), B), dim3(64), lds, st, ex,
                       ex_stride, gain, biquads, wf, Tx, F, K, hop, W, pad, nfr, gain_mode, WS, XS);
    GOLF_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * Ty;
    hipLaunchKernelGGL(ff_ola_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)wf, window, y,
                       y_stride, B, Ty, hop, W, nfr, pad);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" size_t golf_biquad_frames_bwd_workspace_bytes(int B, int Tx, int F, int K, int hop, int W, int pad) {
    if (B < 1 || Tx < 1 || F < 1 || K < 1 || hop < 1 || W < 1 || pad < 0 || Tx + 2 * pad < W) return 0;
    const int nfr = (Tx + 2 * pad - W) / hop + 1;
    return align_up(sizeof(float) * (size_t)B * nfr * W, 256) + align_up(sizeof(float) * (size_t)B * Tx, 256);
}

extern "C" int golf_biquad_frames_ola_bwd_f32(const float* gq, int64_t gq_stride, const float* ex, int64_t ex_stride,
                                              const float* gain, const float* biquads, const float* window, float* g_ex,
                                              int64_t g_ex_stride, float* g_gain_frames, float* g_biquads, float* gx_ex,
                                              int B, int Tx, int F, int K, int hop, int W, int pad, int gain_mode, int Ty,
                                              void* ws, size_t ws_bytes, void* stream) {
    if (B < 1 || Tx < 1 || F < 1 || K < 1 || hop < 1 || W < 1 || pad < 0 || (gain_mode != 0 && gain_mode != 1))
        return fail(GOLF_EINVAL, "biquad_frames_bwd: bad size / mode");
    if (!gq || !ex || !gain || !biquads || !window || !g_ex || !g_biquads || (gain_mode == 1 && !g_gain_frames) ||
        (gain_mode == 0 && !gx_ex))
        return fail(GOLF_EINVAL, "biquad_frames_bwd: null pointer");
    if (K > BQ_ROW) return fail(GOLF_EUNSUPPORTED, "biquad_frames_bwd: %d sections > %d (one DPP row per frame)", K, BQ_ROW);
    if (Tx + 2 * pad < W) return fail(GOLF_EINVAL, "biquad_frames_bwd: signal shorter than one frame");
    const int nfr = (Tx + 2 * pad - W) / hop + 1;
    const int ty = (nfr - 1) * hop + W - 2 * pad;
    if (nfr > F) return fail(GOLF_EINVAL, "biquad_frames_bwd: %d frames vs %d coefficient frames", nfr, F);
    if (ty != Ty || ty < 1) return fail(GOLF_EINVAL, "biquad_frames_bwd: Ty=%d, expected %d", Ty, ty);
    if (gq_stride < Ty || ex_stride < Tx || g_ex_stride < Tx) return fail(GOLF_EINVAL, "biquad_frames_bwd: row stride too small");
    const size_t need = golf_biquad_frames_bwd_workspace_bytes(B, Tx, F, K, hop, W, pad);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "biquad_frames_bwd: workspace needs %zu bytes, 256-aligned (got %zu)", need, ws_bytes);
    const size_t lds = sizeof(float) * (size_t)(K + 2) * (W + 4);
    if (lds > 64 * 1024) return fail(GOLF_EUNSUPPORTED, "biquad_frames_bwd: %d sections x window %d exceed the LDS staging", K, W);
    hipStream_t st = (hipStream_t)stream;
    float* ufr = (float*)ws;
    // frames beyond nfr own no samples: their coefficient / gain gradients are zero
    if (hipMemsetAsync(g_biquads, 0, sizeof(float) * (size_t)B * F * K * 3, st) != hipSuccess)
        return fail((int)hipErrorUnknown, "biquad_frames_bwd: memset failed");
    if (gain_mode == 1 && hipMemsetAsync(g_gain_frames, 0, sizeof(float) * (size_t)B * F, st) != hipSuccess)
        return fail((int)hipErrorUnknown, "biquad_frames_bwd: memset failed");
    hipLaunchKernelGGL(ff_biquad_bwd_kernel, dim3((unsigned)nfr, B), dim3(64), lds, st, gq, gq_stride, ex, ex_stride, gain,
                       biquads, window, ufr, g_biquads, g_gain_frames, Tx, Ty, F, K, hop, W, pad, nfr, gain_mode);
    GOLF_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * Tx;
    hipLaunchKernelGGL(ff_biquad_bwd_ola_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)ufr, ex,
                       ex_stride, gain, g_ex, g_ex_stride, gx_ex, B, Tx, F, hop, W, pad, nfr, gain_mode);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
