// Sample-wise time-varying all-pole (LPC) synthesis filter for gfx950 — GOLF-ss end filter.
//
// Replaces LTVMinimumPhaseFilterPrecise.forward (reference models/filters.py:99-113), i.e.
// AudioTensor gain broadcast + a.reduce_hop_length() (models/utils.py:171-191,538-544) +
// torchlpc.sample_wise_lpc (models/filters.py:112), and its autograd backward.
//
// Algorithm (MI355X-first; nothing like the reference's serial 22-thread numba kernel):
//   The recursion y[t] = x[t] - sum_i A[t,i] y[t-1-i] is linear in the state
//   s_t = (y[t-1..t-M]).  Time is cut into chunks of L samples.  Per chunk c:
//     P1h  M homogeneous trajectories (unit initial states, no input)  -> Phi_c (MxM)   [fp64]
//     P1z  one zero-state trajectory with the real input               -> z_c   (M)     [fp32]
//   so that s_{c+1} = Phi_c s_c + z_c.  Then
//     P2   one wave per utterance scans the NC chunk boundaries (lane = state component)
//     P3   every chunk re-runs its L-step recursion from its now-known initial state -> y
//   B*NC*(M+1) independent in-lane recursions instead of B serial ones: at B=32, T=47761
//   that is 146k lanes x 240 steps instead of 32 lanes x 47761 steps.
//   Phi is computed in fp64 (fp32 homogeneous trajectories lose ~2e-5 relative accuracy each,
//   which the boundary scan amplifies to >1e-4; measured in DESIGN.md §numerics) and stored fp32.
//   Frame-rate coefficients (B,F,M) are interpolated on the fly (a_f + n*d_f, one FMA per tap);
//   the (B,T,M) tensor the reference materialises (134 MB at B=32) never exists.
//   Each lane keeps its M-sample history in a statically indexed rotating register window
//   (time loop unrolled by W, W | hop), so there are no moves and no LDS traffic in the loop.
//
// Backward: the adjoint of the recursion in transposed form
//     g[t] = gy[t] + lam[0];   lam[k] <- lam[k+1] - A[t,k] g[t]
//   (same-time coefficients: no tap-shifted A[t+1+i,i]) has chunk transition Phi_c^T, so the
//   forward's Phi is reused: B1 local adjoint per chunk, B2 boundary scan with Phi^T, B3 per-chunk
//   reverse recursion that also accumulates d/d gain and d/d a at frame rate (hat weights),
//   B4 tiny segment->frame reduction.  No (B,T,M) gradient tensor either.
#include "common.h"

namespace golf {

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
// (W, NT) kernel instantiations: NT taps computed (zero padded above M), ring width W >= NT+1
// (the adjoint ring needs one free slot), W | hop.  Keep in sync with GOLF_SS_DISPATCH below.
struct WNT { int W, NT; };
static const WNT kTable[] = {{8, 2},   {8, 4},   {8, 6},   {16, 8},  {16, 12}, {16, 14}, {24, 8},  {24, 12},
                             {24, 16}, {24, 20}, {24, 22}, {32, 16}, {32, 22}, {32, 26}, {32, 30}, {40, 22},
                             {40, 26}, {40, 32}, {40, 38}};

bool make_ss_plan(int B, int T, int F, int M, int hop, SsPlan* p) {
    p->W = 0;
    p->NT = 0;
    if (F >= 2) {
        for (const WNT& e : kTable) {
            if (e.NT < M || hop % e.W != 0) continue;
            if (p->W == 0 || e.NT < p->NT || (e.NT == p->NT && e.W < p->W)) { p->W = e.W; p->NT = e.NT; }
        }
    }
    if (p->W == 0) { p->total = 256; return false; }
    const int W = p->W;
    int L;
    const int target = 240;
    if (hop >= target) {
        L = W;
        for (int cand = W; cand <= 256 && cand <= hop; cand += W)
            if (hop % cand == 0) L = cand;
    } else {
        L = hop * (target / hop);
    }
    p->L = L;
    p->NC = (int)ceil_div(T, L);
    p->NP = p->NC - 1;
    p->seg = L < hop ? L : hop;
    p->NSEG = (int)ceil_div(T, p->seg);
    size_t o = 0;
    p->off_phi = o;  o = align_up(o + sizeof(float) * (size_t)B * (p->NP > 0 ? p->NP : 1) * p->NT * W, 256);
    p->off_z = o;    o = align_up(o + sizeof(float) * (size_t)B * (p->NP > 0 ? p->NP : 1) * W, 256);
    p->off_S = o;    o = align_up(o + sizeof(float) * (size_t)B * p->NC * W, 256);
    p->off_zadj = o; o = align_up(o + sizeof(float) * (size_t)B * p->NC * W, 256);
    p->off_lam = o;  o = align_up(o + sizeof(float) * (size_t)B * p->NC * W, 256);
    p->off_pa = o;   o = align_up(o + sizeof(float) * (size_t)B * p->NSEG * 2 * W, 256);
    p->off_pg = o;   o = align_up(o + sizeof(float) * (size_t)B * p->NSEG * 2, 256);
    p->total = o;
    return true;
}

// ------------------------------------------------------------------------------------------
// P1h: homogeneous trajectories in fp64 -> Phi[b][c][j][i] = d s_end[i] / d s_start[j]
//   lane = flat chunk q = b*NP + c;  blockIdx.y = trajectory pair (j0, j0+1)
// ------------------------------------------------------------------------------------------
template <int W, int NT>
__global__ __launch_bounds__(64) void lpc_p1_hom_kernel(const float* __restrict__ a, float* __restrict__ Phi,
                                                        int F, int M, int hop, int L, int NP, int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int j0 = 2 * blockIdx.y, j1 = j0 + 1;
    float* out0 = Phi + ((size_t)q * NT + j0) * W;
    float* out1 = out0 + W;
    if (j0 >= M) {  // padding rows: exact zeros
#pragma unroll
        for (int i = 0; i < W; ++i) { out0[i] = 0.f; out1[i] = 0.f; }
        return;
    }
    const int b = q / NP, c = q - b * NP;
    double h0[W], h1[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        h0[k] = (W - 1 - k == j0) ? 1.0 : 0.0;
        h1[k] = (W - 1 - k == j1 && j1 < M) ? 1.0 : 0.0;
    }
    double a0[NT], dd[NT];
    const double inv_hop = 1.0 / (double)hop;
    int fcur = -1;
    const int nblk = L / W;
    for (int blk = 0; blk < nblk; ++blk) {
        const int t0 = c * L + blk * W;
        const int f = t0 / hop;  // <= F-2: chunks with a transition matrix end before (F-1)*hop
        if (f != fcur) {
            fcur = f;
            const float* pa0 = a + ((size_t)b * F + f) * M;
            const float* pa1 = pa0 + M;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const double v0 = i < M ? (double)pa0[i] : 0.0;
                const double v1 = i < M ? (double)pa1[i] : 0.0;
                a0[i] = v0;
                dd[i] = (v1 - v0) * inv_hop;
            }
        }
        const double n0 = (double)(t0 - f * hop);
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const double n = n0 + (double)s;
            double r0a = 0.0, r0b = 0.0, r1a = 0.0, r1b = 0.0;
#pragma unroll
            for (int i = NT - 1; i >= 1; --i) {
                const double cf = fma(n, dd[i], a0[i]);
                const int slot = (s - 1 - i + 2 * W) % W;
                if (i & 1) { r0a = fma(cf, h0[slot], r0a); r1a = fma(cf, h1[slot], r1a); }
                else       { r0b = fma(cf, h0[slot], r0b); r1b = fma(cf, h1[slot], r1b); }
            }
            const double cf0 = fma(n, dd[0], a0[0]);
            const int sp = (s - 1 + W) % W;
            const double y0 = fma(-cf0, h0[sp], -(r0a + r0b));
            const double y1 = fma(-cf0, h1[sp], -(r1a + r1b));
            h0[s] = y0;
            h1[s] = y1;
        }
    }
#pragma unroll
    for (int i = 0; i < W; ++i) {
        out0[i] = i < M ? (float)h0[W - 1 - i] : 0.f;
        out1[i] = (i < M && j1 < M) ? (float)h1[W - 1 - i] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// P1z / P3: fp32 in-lane recursion over one chunk.  lane = flat chunk q = b*NCQ + c
//   MODE 0 (P1z): zero initial state, store final state to zout[q][W]
//   MODE 1 (P3) : initial state from S[q][W], store y[b][t]
// ------------------------------------------------------------------------------------------
template <int W, int NT, int MODE>
__global__ __launch_bounds__(64) void lpc_chunk_f32_kernel(const float* __restrict__ ex, int64_t ex_stride,
                                                           const float* __restrict__ gain,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ S, float* __restrict__ out,
                                                           int64_t y_stride, int T, int F, int M, int hop, int L,
                                                           int NCQ, int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int b = q / NCQ, c = q - b * NCQ;
    float h[W];
    if (MODE == 1) {
        const float* sp = S + (size_t)q * W;
#pragma unroll
        for (int i = 0; i < W; ++i) h[W - 1 - i] = sp[i];
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) h[k] = 0.f;
    }
    float a0[NT], dd[NT];
    float g0 = 0.f, dg = 0.f;
    const float inv_hop = 1.0f / (float)hop;
    const float* exb = ex + (size_t)b * ex_stride;
    float* yb = MODE == 1 ? out + (size_t)b * y_stride : nullptr;
    int fcur = -1;
    const int nblk = L / W;
    for (int blk = 0; blk < nblk; ++blk) {
        const int t0 = c * L + blk * W;
        if (t0 >= T) break;
        int f = t0 / hop;
        if (f > F - 2) f = F - 2;
        if (f != fcur) {
            fcur = f;
            const float* pa0 = a + ((size_t)b * F + f) * M;
            const float* pa1 = pa0 + M;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float v0 = i < M ? pa0[i] : 0.f;
                const float v1 = i < M ? pa1[i] : 0.f;
                a0[i] = v0;
                dd[i] = (v1 - v0) * inv_hop;
            }
            g0 = gain[(size_t)b * F + f];
            dg = (gain[(size_t)b * F + f + 1] - g0) * inv_hop;
        }
        const float n0 = (float)(t0 - f * hop);
        float xin[W];
#pragma unroll
        for (int s = 0; s < W; ++s) xin[s] = (t0 + s < T) ? exb[t0 + s] : 0.f;
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const float n = n0 + (float)s;
            const float x = xin[s] * fmaf(n, dg, g0);
            float ra = 0.f, rb = 0.f;
#pragma unroll
            for (int i = NT - 1; i >= 1; --i) {
                const float cf = fmaf(n, dd[i], a0[i]);
                const int slot = (s - 1 - i + 2 * W) % W;
                if (i & 1) ra = fmaf(cf, h[slot], ra);
                else       rb = fmaf(cf, h[slot], rb);
            }
            const float cf0 = fmaf(n, dd[0], a0[0]);
            const float base = x - (ra + rb);
            const float y = fmaf(-cf0, h[(s - 1 + W) % W], base);
            h[s] = y;
        }
        if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < W; ++s)
                if (t0 + s < T) yb[t0 + s] = h[s];
        }
    }
    if (MODE == 0) {
        float* zp = out + (size_t)q * W;
#pragma unroll
        for (int i = 0; i < W; ++i) zp[i] = i < M ? h[W - 1 - i] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// P2: chunk-boundary scan, one wave per utterance, lane i = state component.
//   S[b][c][:] = state at the start of chunk c;  s_{c+1} = Phi_c s_c + z_c
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int W, int NT>
__global__ __launch_bounds__(64) void lpc_p2_scan_kernel(const float* __restrict__ Phi, const float* __restrict__ z,
                                                         float* __restrict__ S, int NC, int NP) {
    const int b = blockIdx.x;
    const int i = threadIdx.x;
    const bool act = i < W;
    const int ii = act ? i : 0;
    const float* phib = Phi + (size_t)b * NP * NT * W;
    const float* zb = z + (size_t)b * NP * W;
    float* Sb = S + (size_t)b * NC * W;
    float s = 0.f;
    float col[NT], zc = 0.f;
    if (NP > 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) col[j] = phib[(size_t)j * W + ii];
        zc = zb[ii];
    }
    for (int c = 0; c < NC; ++c) {
        if (act) Sb[(size_t)c * W + i] = s;
        if (c >= NP) break;
        float ncol[NT], nz = 0.f;
        const int cn = c + 1 < NP ? c + 1 : c;  // harmless re-load on the last step
#pragma unroll
        for (int j = 0; j < NT; ++j) ncol[j] = phib[((size_t)cn * NT + j) * W + ii];
        nz = zb[(size_t)cn * W + ii];
        float acc0 = zc, acc1 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; j += 2) {
            acc0 = fmaf(col[j], lane_bcast(s, j), acc0);
            acc1 = fmaf(col[j + 1], lane_bcast(s, j + 1), acc1);
        }
        s = acc0 + acc1;
#pragma unroll
        for (int j = 0; j < NT; ++j) col[j] = ncol[j];
        zc = nz;
    }
}

// ------------------------------------------------------------------------------------------
// Backward.  Adjoint ring p[(k + r) % W] = lam_r[k], r = steps done in this block (W >= NT+1).
// ------------------------------------------------------------------------------------------
// B1 (MODE 0): lam_end = 0, store lam at chunk start -> zadj[q][W]
// B3 (MODE 1): lam_end from lamEnd[q][W]; writes g_ex and per-segment partial sums.
template <int W, int NT, int MODE>
__global__ __launch_bounds__(64) void lpc_adj_chunk_kernel(
    const float* __restrict__ gy, int64_t gy_stride, const float* __restrict__ y, int64_t y_stride,
    const float* __restrict__ ex, int64_t ex_stride, const float* __restrict__ gain, const float* __restrict__ a,
    const float* __restrict__ lamEnd, float* __restrict__ zadj, float* __restrict__ g_ex, int64_t g_ex_stride,
    float* __restrict__ pa, float* __restrict__ pg, int T, int F, int M, int hop, int L, int NC, int seg, int NSEG,
    int nq) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int b = q / NC, c = q - b * NC;
    float p[W];
    if (MODE == 1) {
        const float* lp = lamEnd + (size_t)q * W;
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = k < NT ? lp[k] : 0.f;
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = 0.f;
    }
    float a0[NT], dd[NT];
    float g0 = 0.f, dg = 0.f;
    float V0[NT], V1[NT], U0 = 0.f, U1 = 0.f;
    float ycur[W], yprev[W];
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < NT; ++k) { V0[k] = 0.f; V1[k] = 0.f; }
    }
    const float inv_hop = 1.0f / (float)hop;
    const float* gyb = gy + (size_t)b * gy_stride;
    const float* yb = MODE == 1 ? y + (size_t)b * y_stride : nullptr;
    const float* exb = MODE == 1 ? ex + (size_t)b * ex_stride : nullptr;
    float* gxb = MODE == 1 ? g_ex + (size_t)b * g_ex_stride : nullptr;
    int fcur = -1;
    const int nblk = L / W;
    bool have_prev = false;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int t0 = c * L + blk * W;
        if (t0 >= T) continue;  // lam is still identically zero there
        int f = t0 / hop;
        if (f > F - 2) f = F - 2;
        if (f != fcur) {
            fcur = f;
            const float* pa0 = a + ((size_t)b * F + f) * M;
            const float* pa1 = pa0 + M;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float v0 = i < M ? pa0[i] : 0.f;
                const float v1 = i < M ? pa1[i] : 0.f;
                a0[i] = v0;
                dd[i] = (v1 - v0) * inv_hop;
            }
            if (MODE == 1) {
                g0 = gain[(size_t)b * F + f];
                dg = (gain[(size_t)b * F + f + 1] - g0) * inv_hop;
            }
        }
        const float n0 = (float)(t0 - f * hop);
        float gin[W], xin[W];
#pragma unroll
        for (int s = 0; s < W; ++s) gin[s] = (t0 + s < T) ? gyb[t0 + s] : 0.f;
        if (MODE == 1) {
#pragma unroll
            for (int s = 0; s < W; ++s) xin[s] = (t0 + s < T) ? exb[t0 + s] : 0.f;
            if (have_prev) {
#pragma unroll
                for (int s = 0; s < W; ++s) ycur[s] = yprev[s];
            } else {
#pragma unroll
                for (int s = 0; s < W; ++s) ycur[s] = (t0 + s < T) ? yb[t0 + s] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < W; ++s) yprev[s] = (t0 - W + s >= 0) ? yb[t0 - W + s] : 0.f;
            have_prev = true;
        }
#pragma unroll
        for (int s = W - 1; s >= 0; --s) {
            const int r = W - 1 - s;
            const float n = n0 + (float)s;
            const float g = gin[s] + p[r];
            p[r] = 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const float cf = fmaf(n, dd[k], a0[k]);
                p[(k + r + 1) % W] = fmaf(-cf, g, p[(k + r + 1) % W]);
            }
            if (MODE == 1) {
                const float gn = g * n;
                const float G = fmaf(n, dg, g0);
                if (t0 + s < T) gxb[t0 + s] = g * G;
                U0 = fmaf(g, xin[s], U0);
                U1 = fmaf(gn, xin[s], U1);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int idx = s - 1 - k;
                    const float yv = idx >= 0 ? ycur[idx >= 0 ? idx : 0] : yprev[idx >= 0 ? 0 : W + idx];
                    V0[k] = fmaf(-g, yv, V0[k]);
                    V1[k] = fmaf(-gn, yv, V1[k]);
                }
            }
        }
        if (MODE == 1 && (t0 % seg) == 0) {  // finished a gradient segment: flush
            const int sg = t0 / seg;
            float* pp = pa + ((size_t)b * NSEG + sg) * 2 * W;
#pragma unroll
            for (int k = 0; k < NT; ++k) { pp[k] = V0[k]; pp[W + k] = V1[k]; V0[k] = 0.f; V1[k] = 0.f; }
            pg[((size_t)b * NSEG + sg) * 2 + 0] = U0;
            pg[((size_t)b * NSEG + sg) * 2 + 1] = U1;
            U0 = 0.f;
            U1 = 0.f;
        }
    }
    if (MODE == 0) {
        float* zp = zadj + (size_t)q * W;
#pragma unroll
        for (int k = 0; k < W; ++k) zp[k] = k < NT ? p[k] : 0.f;
    }
}

// B2: lamEnd[b][c][:] = adjoint state at the END of chunk c; lam_start(c) = Phi_c^T lam_end(c) + zadj_c
template <int W, int NT>
__global__ __launch_bounds__(64) void lpc_adj_scan_kernel(const float* __restrict__ Phi, const float* __restrict__ zadj,
                                                          float* __restrict__ lamEnd, int NC, int NP) {
    const int b = blockIdx.x;
    const int j = threadIdx.x;
    const bool act = j < NT;
    const int jj = act ? j : 0;
    const float* phib = Phi + (size_t)b * NP * NT * W;
    float lam = 0.f;
    for (int c = NC - 1; c >= 0; --c) {
        if (j < W) lamEnd[((size_t)b * NC + c) * W + j] = act ? lam : 0.f;
        float acc0 = zadj[((size_t)b * NC + c) * W + (j < W ? j : 0)], acc1 = 0.f;
        if (c < NP) {
            const float* row = phib + ((size_t)c * NT + jj) * W;
            float r[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) r[i] = row[i];
#pragma unroll
            for (int i = 0; i < NT; i += 2) {
                acc0 = fmaf(r[i], lane_bcast(lam, i), acc0);
                acc1 = fmaf(r[i + 1], lane_bcast(lam, i + 1), acc1);
            }
        }
        lam = act ? acc0 + acc1 : 0.f;
    }
}

// B4: per-segment partial sums -> frame-rate gradients (adjoint of the hat interpolation).
__global__ void lpc_grad_reduce_kernel(const float* __restrict__ pa, const float* __restrict__ pg,
                                       float* __restrict__ g_a, float* __restrict__ g_gain, int B, int F, int M,
                                       int W, int hop, int seg, int NSEG) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = M + 1;
    if (idx >= B * F * per) return;
    const int k = idx % per;
    const int f = (idx / per) % F;
    const int b = idx / (per * F);
    const int R = hop / seg;
    const float inv_hop = 1.0f / (float)hop;
    float acc = 0.f;
    // segments whose frame is f contribute V0 - V1/hop; segments whose frame is f-1 contribute V1/hop
    const int lo = (f - 1) * R < 0 ? 0 : (f - 1) * R;
    const int hi = (f + 2) * R < NSEG ? (f + 2) * R : NSEG;
    for (int sg = lo; sg < hi; ++sg) {
        int fs = sg / R;
        if (fs > F - 2) fs = F - 2;
        float v0, v1;
        if (k < M) {
            const float* pp = pa + ((size_t)b * NSEG + sg) * 2 * W;
            v0 = pp[k];
            v1 = pp[W + k];
        } else {
            v0 = pg[((size_t)b * NSEG + sg) * 2 + 0];
            v1 = pg[((size_t)b * NSEG + sg) * 2 + 1];
        }
        if (fs == f) acc += v0 - v1 * inv_hop;
        if (fs == f - 1) acc += v1 * inv_hop;
    }
    if (k < M) g_a[((size_t)b * F + f) * M + k] = acc;
    else g_gain[(size_t)b * F + f] = acc;
}

// ------------------------------------------------------------------------------------------
// Generic fallback (any hop / M <= 64 / F >= 1): one lane per utterance, serial in t, history read
// back from the output row.  Correct for every shape, slow; only used when no W divides hop.
// ------------------------------------------------------------------------------------------
__global__ void lpc_ss_generic_kernel(const float* __restrict__ ex, int64_t ex_stride, const float* __restrict__ gain,
                                      const float* __restrict__ a, float* y, int64_t y_stride, int B, int T, int F,
                                      int M, int hop) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* exb = ex + (size_t)b * ex_stride;
    volatile float* yb = y + (size_t)b * y_stride;
    const float inv_hop = 1.0f / (float)hop;
    for (int t = 0; t < T; ++t) {
        int f = F >= 2 ? t / hop : 0;
        if (F >= 2 && f > F - 2) f = F - 2;
        const float n = (float)(t - f * hop);
        const float* pa0 = a + ((size_t)b * F + f) * M;
        const float* pa1 = F >= 2 ? pa0 + M : pa0;
        const float g0 = gain[(size_t)b * F + f];
        const float g1 = F >= 2 ? gain[(size_t)b * F + f + 1] : g0;
        float acc = exb[t] * fmaf(n, (g1 - g0) * inv_hop, g0);
        float ra = 0.f;
        for (int i = M - 1; i >= 0; --i) {
            if (t - 1 - i < 0) continue;
            const float cf = fmaf(n, (pa1[i] - pa0[i]) * inv_hop, pa0[i]);
            ra = fmaf(cf, yb[t - 1 - i], ra);
        }
        yb[t] = acc - ra;
    }
}

// a-5 inverse filter: fully parallel FIR with interpolated coefficients.
__global__ void lpc_inverse_kernel(const float* __restrict__ y, int64_t y_stride, const float* __restrict__ a,
                                   float* __restrict__ e, int64_t e_stride, int B, int T, int F, int M, int hop) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * T) return;
    const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
    int f = F >= 2 ? t / hop : 0;
    if (F >= 2 && f > F - 2) f = F - 2;
    const float w = (float)(t - f * hop) / (float)hop;
    const float* pa0 = a + ((size_t)b * F + f) * M;
    const float* pa1 = F >= 2 ? pa0 + M : pa0;
    const float* yb = y + (size_t)b * y_stride;
    float acc = yb[t];
    for (int i = 0; i < M; ++i) {
        if (t - 1 - i < 0) break;
        const float cf = fmaf(w, pa1[i] - pa0[i], pa0[i]);
        acc = fmaf(cf, yb[t - 1 - i], acc);
    }
    e[(size_t)b * e_stride + t] = acc;
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
template <int W, int NT>
static int launch_fwd(const SsPlan& p, const float* ex, int64_t ex_stride, const float* gain, const float* a, float* y,
                      int64_t y_stride, int B, int T, int F, int M, int hop, char* ws, hipStream_t st) {
    float* Phi = (float*)(ws + p.off_phi);
    float* z = (float*)(ws + p.off_z);
    float* S = (float*)(ws + p.off_S);
    if (p.NP > 0) {
        const int nq = B * p.NP;
        dim3 g1((unsigned)ceil_div(nq, 64), NT / 2);
        hipLaunchKernelGGL((lpc_p1_hom_kernel<W, NT>), g1, dim3(64), 0, st, a, Phi, F, M, hop, p.L, p.NP, nq);
        GOLF_LAUNCH_CHECK();
        hipLaunchKernelGGL((lpc_chunk_f32_kernel<W, NT, 0>), dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, st, ex,
                           ex_stride, gain, a, (const float*)nullptr, z, (int64_t)0, T, F, M, hop, p.L, p.NP, nq);
        GOLF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((lpc_p2_scan_kernel<W, NT>), dim3(B), dim3(64), 0, st, Phi, z, S, p.NC, p.NP);
    GOLF_LAUNCH_CHECK();
    const int nq3 = B * p.NC;
    hipLaunchKernelGGL((lpc_chunk_f32_kernel<W, NT, 1>), dim3((unsigned)ceil_div(nq3, 64)), dim3(64), 0, st, ex,
                       ex_stride, gain, a, (const float*)S, y, y_stride, T, F, M, hop, p.L, p.NC, nq3);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

template <int W, int NT>
static int launch_bwd(const SsPlan& p, const float* gy, int64_t gy_stride, const float* y, int64_t y_stride,
                      const float* ex, int64_t ex_stride, const float* gain, const float* a, float* g_ex,
                      int64_t g_ex_stride, float* g_gain, float* g_a, int B, int T, int F, int M, int hop, char* ws,
                      hipStream_t st) {
    const float* Phi = (const float*)(ws + p.off_phi);
    float* zadj = (float*)(ws + p.off_zadj);
    float* lam = (float*)(ws + p.off_lam);
    float* pa = (float*)(ws + p.off_pa);
    float* pg = (float*)(ws + p.off_pg);
    const int nq = B * p.NC;
    const dim3 gq((unsigned)ceil_div(nq, 64));
    hipLaunchKernelGGL((lpc_adj_chunk_kernel<W, NT, 0>), gq, dim3(64), 0, st, gy, gy_stride, (const float*)nullptr,
                       (int64_t)0, (const float*)nullptr, (int64_t)0, gain, a, (const float*)nullptr, zadj,
                       (float*)nullptr, (int64_t)0, (float*)nullptr, (float*)nullptr, T, F, M, hop, p.L, p.NC, p.seg,
                       p.NSEG, nq);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL((lpc_adj_scan_kernel<W, NT>), dim3(B), dim3(64), 0, st, Phi, (const float*)zadj, lam, p.NC,
                       p.NP);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL((lpc_adj_chunk_kernel<W, NT, 1>), gq, dim3(64), 0, st, gy, gy_stride, y, y_stride, ex,
                       ex_stride, gain, a, (const float*)lam, (float*)nullptr, g_ex, g_ex_stride, pa, pg, T, F, M, hop,
                       p.L, p.NC, p.seg, p.NSEG, nq);
    GOLF_LAUNCH_CHECK();
    const int n4 = B * F * (M + 1);
    hipLaunchKernelGGL(lpc_grad_reduce_kernel, dim3((unsigned)ceil_div(n4, 256)), dim3(256), 0, st, (const float*)pa,
                       (const float*)pg, g_a, g_gain, B, F, M, W, hop, p.seg, p.NSEG);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

// (W, NT) instantiation table — must list exactly kTable.
#define GOLF_SS_CASE(FN, w, nt, ...) case (w) * 100 + (nt): return FN<w, nt>(__VA_ARGS__);
#define GOLF_SS_DISPATCH(FN, ...)               \
    switch (p.W * 100 + p.NT) {                 \
        GOLF_SS_CASE(FN, 8, 2, __VA_ARGS__)     \
        GOLF_SS_CASE(FN, 8, 4, __VA_ARGS__)     \
        GOLF_SS_CASE(FN, 8, 6, __VA_ARGS__)     \
        GOLF_SS_CASE(FN, 16, 8, __VA_ARGS__)    \
        GOLF_SS_CASE(FN, 16, 12, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 16, 14, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 24, 8, __VA_ARGS__)    \
        GOLF_SS_CASE(FN, 24, 12, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 24, 16, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 24, 20, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 24, 22, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 32, 16, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 32, 22, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 32, 26, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 32, 30, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 40, 22, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 40, 26, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 40, 32, __VA_ARGS__)   \
        GOLF_SS_CASE(FN, 40, 38, __VA_ARGS__)   \
        default: break;                         \
    }

static bool plan_fast(int B, int T, int F, int M, int hop, SsPlan* p) {
    return make_ss_plan(B, T, F, M, hop, p);
}

}  // namespace golf

using namespace golf;

static int check_ss_args(int B, int T, int F, int M, int hop) {
    if (B < 1 || T < 1 || F < 1 || M < 1 || hop < 1) return fail(GOLF_EINVAL, "ltv_allpole: non-positive size");
    if (M > 64) return fail(GOLF_EUNSUPPORTED, "ltv_allpole: M=%d > 64", M);
    if ((int64_t)T > (int64_t)(F - 1) * hop + 1)
        return fail(GOLF_EINVAL, "ltv_allpole: T=%d exceeds (F-1)*hop+1=%lld", T, (long long)(F - 1) * hop + 1);
    return GOLF_OK;
}

extern "C" size_t golf_ltv_allpole_workspace_bytes(int B, int T, int F, int M, int hop) {
    SsPlan p;
    if (B < 1 || T < 1 || F < 1 || M < 1 || hop < 1) return 0;
    if (!plan_fast(B, T, F, M, hop, &p)) return 256;
    return p.total;
}

extern "C" int golf_ltv_allpole_fwd_f32(const float* ex, int64_t ex_stride, const float* gain, const float* a,
                                        float* y, int64_t y_stride, int B, int T, int F, int M, int hop, void* ws,
                                        size_t ws_bytes, void* stream) {
    if (int rc = check_ss_args(B, T, F, M, hop)) return rc;
    if (!ex || !gain || !a || !y) return fail(GOLF_EINVAL, "ltv_allpole_fwd: null pointer");
    if (ex_stride < T || y_stride < T) return fail(GOLF_EINVAL, "ltv_allpole_fwd: row stride < T");
    hipStream_t st = (hipStream_t)stream;
    SsPlan p;
    if (!plan_fast(B, T, F, M, hop, &p)) {
        hipLaunchKernelGGL(lpc_ss_generic_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, st, ex, ex_stride, gain,
                           a, y, y_stride, B, T, F, M, hop);
        GOLF_LAUNCH_CHECK();
        return GOLF_OK;
    }
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "ltv_allpole_fwd: workspace needs %zu bytes, 256-aligned (got %zu)", p.total,
                    ws_bytes);
    GOLF_SS_DISPATCH(launch_fwd, p, ex, ex_stride, gain, a, y, y_stride, B, T, F, M, hop, (char*)ws, st)
    return fail(GOLF_EUNSUPPORTED, "ltv_allpole_fwd: no kernel for W=%d NT=%d", p.W, p.NT);
}

extern "C" int golf_ltv_allpole_bwd_f32(const float* gy, int64_t gy_stride, const float* y, int64_t y_stride,
                                        const float* ex, int64_t ex_stride, const float* gain, const float* a,
                                        float* g_ex, int64_t g_ex_stride, float* g_gain, float* g_a, int B, int T,
                                        int F, int M, int hop, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_ss_args(B, T, F, M, hop)) return rc;
    if (!gy || !y || !ex || !gain || !a || !g_ex || !g_gain || !g_a)
        return fail(GOLF_EINVAL, "ltv_allpole_bwd: null pointer");
    if (gy_stride < T || y_stride < T || ex_stride < T || g_ex_stride < T)
        return fail(GOLF_EINVAL, "ltv_allpole_bwd: row stride < T");
    SsPlan p;
    if (!plan_fast(B, T, F, M, hop, &p))
        return fail(GOLF_EUNSUPPORTED,
                    "ltv_allpole_bwd: needs a ring width W in {8,16,24,32,40} with W >= M+1 and hop %% W == 0 "
                    "(M=%d hop=%d)", M, hop);
    if (!ws || ws_bytes < p.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "ltv_allpole_bwd: workspace needs %zu bytes, 256-aligned (got %zu)", p.total,
                    ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    GOLF_SS_DISPATCH(launch_bwd, p, gy, gy_stride, y, y_stride, ex, ex_stride, gain, a, g_ex, g_ex_stride, g_gain, g_a,
                     B, T, F, M, hop, (char*)ws, st)
    return fail(GOLF_EUNSUPPORTED, "ltv_allpole_bwd: no kernel for W=%d NT=%d", p.W, p.NT);
}

extern "C" int golf_ltv_inverse_f32(const float* y, int64_t y_stride, const float* a, float* e, int64_t e_stride,
                                    int B, int T, int F, int M, int hop, void* stream) {
    if (int rc = check_ss_args(B, T, F, M, hop)) return rc;
    if (!y || !a || !e) return fail(GOLF_EINVAL, "ltv_inverse: null pointer");
    if (y_stride < T || e_stride < T) return fail(GOLF_EINVAL, "ltv_inverse: row stride < T");
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(lpc_inverse_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, y,
                       y_stride, a, e, e_stride, B, T, F, M, hop);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
