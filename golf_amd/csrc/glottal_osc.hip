// Indexed glottal-flow wavetable oscillator for gfx950 — GOLF's harmonic source.
//
// Replaces IndexedGlottalFlowTable.forward (reference models/synth.py:213-263) and
// GlottalFlowTable.generate (models/synth.py:124-177).  The reference materialises, at B=32:
// (B,21,2048) blended tables, >= 6 oversampled (B,191997) temporaries, a (B,191997,1,2) sampling
// grid for F.grid_sample, and runs torch.cumsum in fp32 over 192k samples.  Here:
//
//   O1  osc_phase_scan   per utterance: exclusive prefix of the per-segment phase advance in fp64,
//                        wrapped to [0,1).  Linear interpolation of the phase increment has a closed
//                        form inside a segment, so only the Tp coarse samples are scanned, not the
//                        N = (Tp-1)*hop*os+1 oversampled ones.
//   O2  osc_render       one workgroup per (utterance, control interval): the two blended table rows
//                        of that interval are staged in LDS (2 x (L+1) floats), every oversampled
//                        sample is then phase -> bilinear LDS lookup -> equal-energy scaling.
//   O3  osc_decimate     strided FIR (kazane.Decimate stand-in; taps are an input) with a polyphase,
//                        bank-conflict-free LDS tile.
//   The running phase is exact to ~1e-13 cycles (the reference's fp32 cumsum drifts ~3e-5 cycles).
#include "common.h"

namespace golf {

struct OscGeom {
    int P;        // fine samples per coarse phase sample = phase_hop * os
    int N;        // oversampled length
    int hop_t;    // fine samples per control (table) frame = w_hop * os
    int nint;     // control intervals = ceil(N / hop_t)
    int ntile;    // phase-scan tiles of OSC_SCAN_TILE coarse samples
    size_t off_cw, off_ttot, off_pre, off_part, total;
};
#define OSC_SCAN_TILE 1024

static void osc_geom(int B, int Tp, int phase_hop, int Fw, int w_hop, int os, OscGeom* g) {
    g->P = phase_hop * os;
    g->N = g->P > 1 ? (Tp - 1) * g->P + 1 : Tp;
    g->hop_t = w_hop * os;
    g->nint = (int)ceil_div(g->N, g->hop_t);
    size_t o = 0;
    g->ntile = (int)ceil_div(Tp, OSC_SCAN_TILE);
    g->off_cw = o;   o = align_up(o + sizeof(double) * (size_t)B * Tp, 256);
    g->off_ttot = o; o = align_up(o + sizeof(double) * (size_t)B * g->ntile, 256);
    g->off_pre = o;  o = align_up(o + sizeof(float) * (size_t)B * g->N, 256);
    g->off_part = o; o = align_up(o + sizeof(float) * (size_t)B * g->nint * 2, 256);
    g->total = o;
}

// ---- O1 ---------------------------------------------------------------------------------------
// segsum_j = phase advance over coarse segment j = (P*p_j + (p_{j+1}-p_j)(P-1)/2)/os  (closed form of the
// linearly interpolated increment).  One workgroup per (tile of 1024 coarse samples, utterance):
//   Cloc[b][j] = frac(sum of segsums of the tile before j)   (exclusive, fp64)
//   Ttot[b][tile] = frac(tile total)
// The render kernel adds the (<= ntile-term) prefix of Ttot itself.  Coalesced loads, wave shuffles.
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
    }
    return v;
}

__global__ __launch_bounds__(256) void osc_phase_tile_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                             double* __restrict__ Cloc, double* __restrict__ Ttot,
                                                             int Tp, int P, int os, int ntile) {
    __shared__ float ps[OSC_SCAN_TILE + 1];
    __shared__ double wsum[4];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const float* pb = phase + (size_t)b * phase_stride;
    const int j0 = tile * OSC_SCAN_TILE;
    for (int u = tid; u < OSC_SCAN_TILE + 1; u += 256) {
        const int j = j0 + u;
        ps[u] = pb[j < Tp ? j : Tp - 1];
    }
    __syncthreads();
    const double inv_os = 1.0 / (double)os;
    const double half = 0.5 * (double)(P - 1);
    double seg[4];
    double tsum = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = tid * 4 + r;
        const int j = j0 + u;
        const double p0 = (double)ps[u], p1 = (double)ps[u + 1];
        seg[r] = j < Tp - 1 ? ((double)P * p0 + (p1 - p0) * half) * inv_os : 0.0;  // segments 0..Tp-2
        tsum += seg[r];
    }
    const double incl = wave_incl_scan(tsum, lane);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    double run = base + incl - tsum;  // exclusive prefix of this thread's first segment
    double* cb = Cloc + (size_t)b * Tp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + tid * 4 + r;
        if (j < Tp) cb[j] = run - floor(run);
        run += seg[r];
    }
    if (tid == 255) Ttot[(size_t)b * ntile + tile] = run - floor(run);
}

// ---- O2 ---------------------------------------------------------------------------------------
// MODE 0: forward render (writes fine samples);  MODE 1: backward w.r.t. table_select_weight
// (reduces g_pre * d(pre)/d(p_row) over the interval into part[b][interval][2]).
#define OSC_RENDER_THREADS 512  // 4 blocks/CU: the 640 blocks of the B=32 config run in one round
template <int MODE>
__global__ __launch_bounds__(OSC_RENDER_THREADS) void osc_render_kernel(
    const float* __restrict__ phase, int64_t phase_stride, const double* __restrict__ Cloc,
    const double* __restrict__ Ttot, int ntile, const float* __restrict__ wsel, int Fw,
    const float* __restrict__ table, int n_tab, int L, int Tp, int P, int os, int hop_t, int N, int equal_energy,
    float* __restrict__ dst, int64_t dst_stride, const float* __restrict__ g_pre, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ double toff[256];  // prefix of the tile totals (ntile <= 256 tiles = 262144 coarse samples)
    float* row0 = smem;
    float* row1 = smem + (L + 1);
    constexpr int NTH = OSC_RENDER_THREADS;
    const int r0 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < ntile) {
        double acc = 0.0;
        for (int t = 0; t < tid; ++t) acc += Ttot[(size_t)b * ntile + t];
        toff[tid] = acc;
    }
    // ---- stage rows r0, r0+1 (blended tables in MODE 0; table differences in MODE 1)
    for (int rr = 0; rr < 2; ++rr) {
        int k = r0 + rr;
        if (k > Fw - 1) k = Fw - 1;
        const float idx = wsel[(size_t)b * Fw + k] * (float)(n_tab - 1);
        int i0 = (int)idx;
        i0 = i0 < 0 ? 0 : (i0 > n_tab - 2 ? n_tab - 2 : i0);
        const float p = idx - (float)i0;
        const float* t0 = table + (size_t)i0 * L;
        const float* t1 = t0 + L;
        float* row = rr ? row1 : row0;
        for (int c = tid; c < L; c += NTH) {
            if (MODE == 0) row[c] = t0[c] * (1.0f - p) + t1[c] * p;
            else row[c] = t1[c] - t0[c];
        }
        if (tid == 0) {
            if (MODE == 0) row[L] = t0[0] * (1.0f - p) + t1[0] * p;
            else row[L] = t1[0] - t0[0];
        }
    }
    __syncthreads();
    const float* pb = phase + (size_t)b * phase_stride;
    const double* cb = Cloc + (size_t)b * Tp;
    const int m_lo = r0 * hop_t;
    const int m_hi = m_lo + hop_t < N ? m_lo + hop_t : N;
    const float inv_hop_t = 1.0f / (float)hop_t;
    const float inv_P = 1.0f / (float)P;
    const float inv_osf = 1.0f / (float)os;
    const double inv_os = 1.0 / (double)os;
    const int dj = NTH / P, dk = NTH % P;  // advance of (j,k) per NTH fine samples
    int m = m_lo + tid;
    int j = m / P;
    int k = m - j * P;
    float acc0 = 0.f, acc1 = 0.f;
    for (; m < m_hi; m += NTH) {
        // coarse sample j (clamped for the final point: k == 0 there, and d == 0 because j+1 clamps to j)
        const int jc = j < Tp - 1 ? j : Tp - 1;
        const int jn = jc + 1 < Tp ? jc + 1 : Tp - 1;
        const float p0 = pb[jc];
        const float d = (pb[jn] - p0) * inv_P;
        const double cj = cb[jc] + toff[jc / OSC_SCAN_TILE];
        const int kk_i = m - jc * P;
        const double kk = (double)kk_i;
        // inclusive cumulative phase: C_j + ((k+1) p0 + d k(k+1)/2)/os, in fp64, wrapped
        double ph = cj + ((kk + 1.0) * (double)p0 + (double)d * (kk * (kk + 1.0) * 0.5)) * inv_os;
        ph -= floor(ph);
        const float c = (float)ph * (float)L;
        int c0 = (int)c;
        c0 = c0 > L - 1 ? L - 1 : c0;  // ph rounded up to 1.0f
        const float cf = c - (float)c0;
        const float rf = (float)(m - m_lo) * inv_hop_t;
        const float a00 = row0[c0], a01 = row0[c0 + 1], a10 = row1[c0], a11 = row1[c0 + 1];
        const float top = fmaf(cf, a01 - a00, a00);
        const float bot = fmaf(cf, a11 - a10, a10);
        float scale = 1.0f;
        if (equal_energy) scale = rsqrtf(fmaf((float)kk_i, d, p0) * inv_osf);
        if (MODE == 0) {
            dst[(size_t)b * dst_stride + m] = fmaf(rf, bot - top, top) * scale;
        } else {
            const float g = g_pre[(size_t)b * N + m] * scale;
            acc0 = fmaf(g * (1.0f - rf), top, acc0);
            acc1 = fmaf(g * rf, bot, acc1);
        }
        j += dj;
        k += dk;
        if (k >= P) { k -= P; j += 1; }
    }
    if (MODE == 1) {
        __syncthreads();
        float* red = smem;  // reuse (rows are dead): 2 x NTH floats
        red[tid] = acc0;
        red[NTH + tid] = acc1;
        __syncthreads();
        for (int off = NTH / 2; off > 0; off >>= 1) {
            if (tid < off) { red[tid] += red[tid + off]; red[NTH + tid] += red[NTH + tid + off]; }
            __syncthreads();
        }
        if (tid == 0) {
            part[((size_t)b * gridDim.x + r0) * 2 + 0] = red[0];
            part[((size_t)b * gridDim.x + r0) * 2 + 1] = red[NTH];
        }
    }
}

// ---- O3 ---------------------------------------------------------------------------------------
// out[o] = sum_k taps[k] * pre[o*os + k - half], zero padded (kazane.Decimate stand-in).
// Polyphase form: k - half = os*d + ph  =>  out[o] = sum_ph sum_d h_ph[d] * X_ph[o + d],
//   X_ph[i] = pre[i*os + ph].  Each thread produces 4 consecutive outputs so that every staged sample feeds
//   4 FMAs; the LDS tile splits X_ph further by (i & 3) so that lanes (stride-4 outputs) hit consecutive banks:
//   addr(ph, i) = (ph*4 + (i&3))*RS4 + (i>>2).  Taps sit in LDS as 4-aligned groups (broadcast ds_read_b128).
#define OSC_TILE 1024
__global__ __launch_bounds__(256) void osc_decimate_kernel(const float* __restrict__ pre, int N,
                                                           const float* __restrict__ taps, int K, int os,
                                                           float* __restrict__ out, int64_t out_stride, int Tout,
                                                           int RS4, int dmin, int ngrp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: X = smem[0 .. os*4*RS4), H = 16-aligned after it: H[ph][ngrp*4 + 4] (3 leading zeros + taps + zero tail)
    float* X = smem;
    const int hoff = (os * 4 * RS4 + 3) & ~3;
    float* H = smem + hoff;
    const int HS = ngrp * 4 + 8;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int o0 = blockIdx.x * OSC_TILE;
    const float* pb = pre + (size_t)b * N;
    const int half = (K - 1) / 2;
    const int64_t m_lo = (int64_t)(o0 + dmin) * os;
    const int span = OSC_TILE + ngrp * 4 + 4;  // polyphase indices staged per phase
    for (int e = tid; e < span * os; e += 256) {
        const int64_t m = m_lo + e;
        const int ph = e % os, i = e / os;
        X[(ph * 4 + (i & 3)) * RS4 + (i >> 2)] = (m >= 0 && m < N) ? pb[m] : 0.f;
    }
    // H[ph][3 + q] = tap of (ph, d = dmin + q), zero elsewhere
    for (int e = tid; e < os * HS; e += 256) {
        const int ph = e / HS, q = e - ph * HS - 3;
        const int k = half + os * (dmin + q) + ph;
        H[e] = (q >= 0 && k >= 0 && k < K) ? taps[k] : 0.f;
    }
    __syncthreads();
    const int u = tid;  // outputs o0 + 4u .. 4u+3
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int ph = 0; ph < os; ++ph) {
        const float* Xp = X + (size_t)ph * 4 * RS4 + u;
        const float4* Hp = reinterpret_cast<const float4*>(H + (size_t)ph * HS);
        // window of taps: w[3 + e' - r] with e = 4*g + e'; hprev = taps q in [4g-4, 4g) (as H idx 4g-1 .. 4g+2)
        float4 hprev = Hp[0];  // H idx 0..3  = q -3..0  (three zeros + tap q=0)
        for (int g = 0; g <= ngrp; ++g) {
            const float4 hcur = Hp[g + 1];  // H idx 4g+4 .. 4g+7 = q 4g+1 .. 4g+4
            const float x0 = Xp[0 * RS4 + g], x1 = Xp[1 * RS4 + g], x2 = Xp[2 * RS4 + g], x3 = Xp[3 * RS4 + g];
            // e = 4g + 0: taps for r=0..3 are q = e - r  -> H idx 3 + e - r = 4g+3-r : hprev.w, .z, .y, .x
            acc0 = fmaf(hprev.w, x0, acc0); acc1 = fmaf(hprev.z, x0, acc1);
            acc2 = fmaf(hprev.y, x0, acc2); acc3 = fmaf(hprev.x, x0, acc3);
            // e = 4g + 1: H idx 4g+4-r : hcur.x, hprev.w, hprev.z, hprev.y
            acc0 = fmaf(hcur.x, x1, acc0); acc1 = fmaf(hprev.w, x1, acc1);
            acc2 = fmaf(hprev.z, x1, acc2); acc3 = fmaf(hprev.y, x1, acc3);
            // e = 4g + 2: H idx 4g+5-r : hcur.y, hcur.x, hprev.w, hprev.z
            acc0 = fmaf(hcur.y, x2, acc0); acc1 = fmaf(hcur.x, x2, acc1);
            acc2 = fmaf(hprev.w, x2, acc2); acc3 = fmaf(hprev.z, x2, acc3);
            // e = 4g + 3: H idx 4g+6-r : hcur.z, hcur.y, hcur.x, hprev.w
            acc0 = fmaf(hcur.z, x3, acc0); acc1 = fmaf(hcur.y, x3, acc1);
            acc2 = fmaf(hcur.x, x3, acc2); acc3 = fmaf(hprev.w, x3, acc3);
            hprev = hcur;
        }
    }
    const int o = o0 + 4 * u;
    float* ob = out + (size_t)b * out_stride;
    if (o < Tout) ob[o] = acc0;
    if (o + 1 < Tout) ob[o + 1] = acc1;
    if (o + 2 < Tout) ob[o + 2] = acc2;
    if (o + 3 < Tout) ob[o + 3] = acc3;
}

// transpose of the decimator: g_pre[m] = sum_o taps[m - o*os + half] * g_out[o]
__global__ void osc_decimate_T_kernel(const float* __restrict__ g_out, int64_t g_out_stride, int Tout,
                                      const float* __restrict__ taps, int K, int os, float* __restrict__ g_pre, int N,
                                      int B) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * N) return;
    const int b = (int)(idx / N), m = (int)(idx - (int64_t)b * N);
    const int half = (K - 1) / 2;
    // k = m - o*os + half in [0,K)  =>  o in [ceil((m+half-K+1)/os), floor((m+half)/os)]
    int ohi = (m + half) / os;
    if (ohi > Tout - 1) ohi = Tout - 1;
    int num = m + half - K + 1;
    int olo = num <= 0 ? 0 : (num + os - 1) / os;
    const float* gb = g_out + (size_t)b * g_out_stride;
    float acc = 0.f;
    for (int o = olo; o <= ohi; ++o) acc = fmaf(taps[m - o * os + half], gb[o], acc);
    g_pre[idx] = acc;
}

// Register-blocked transposed decimator for os == 4 (the GOLF configs): for every output phase p,
//   g_pre[j*4 + p] = sum_d h_p[d] g_out[j - d]  — a plain 33-tap FIR over g_out with reversed taps, evaluated with the
//   same 4-outputs-per-thread sliding window as the forward decimator; the 16 results of a thread are the 16
//   consecutive fine samples (4j..4j+15), staged through LDS for a coalesced store.
__global__ __launch_bounds__(256) void osc_decimate_T4_kernel(const float* __restrict__ g_out, int64_t g_out_stride,
                                                              int Tout, const float* __restrict__ taps, int K,
                                                              float* __restrict__ g_pre, int N, int RS4, int dmax,
                                                              int ngrp) {
    constexpr int OS = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: G = smem[0 .. 4*RS4), H (16-aligned) [OS][ngrp*4+8], O = output tile [OSC_TILE*OS]
    float* G = smem;
    const int hoff = (4 * RS4 + 3) & ~3;
    float* H = smem + hoff;
    const int HS = ngrp * 4 + 8;
    float* O = H + OS * HS;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int j0 = blockIdx.x * OSC_TILE;
    const float* gb = g_out + (size_t)b * g_out_stride;
    const int half = (K - 1) / 2;
    const int span = OSC_TILE + ngrp * 4 + 4;
    for (int i = tid; i < span; i += 256) {  // Gt[i] = g_out[j0 - dmax + i]
        const int j = j0 - dmax + i;
        G[(i & 3) * RS4 + (i >> 2)] = (j >= 0 && j < Tout) ? gb[j] : 0.f;
    }
    for (int e = tid; e < OS * HS; e += 256) {  // H[p][3 + q'] = h_p[dmax - q'] = taps[half + 4*(dmax-q') + p]
        const int p = e / HS, q = e - p * HS - 3;
        const int k = half + OS * (dmax - q) + p;
        H[e] = (q >= 0 && k >= 0 && k < K) ? taps[k] : 0.f;
    }
    __syncthreads();
    const int u = tid;
    float acc[4][OS];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int p = 0; p < OS; ++p) acc[r][p] = 0.f;
    const float* Gp = G + u;
    float4 hprev[OS];
#pragma unroll
    for (int p = 0; p < OS; ++p) hprev[p] = reinterpret_cast<const float4*>(H + p * HS)[0];
    for (int g = 0; g <= ngrp; ++g) {
        const float x0 = Gp[0 * RS4 + g], x1 = Gp[1 * RS4 + g], x2 = Gp[2 * RS4 + g], x3 = Gp[3 * RS4 + g];
#pragma unroll
        for (int p = 0; p < OS; ++p) {
            const float4 hc = reinterpret_cast<const float4*>(H + p * HS)[g + 1];
            const float4 hp = hprev[p];
            acc[0][p] = fmaf(hp.w, x0, acc[0][p]); acc[1][p] = fmaf(hp.z, x0, acc[1][p]);
            acc[2][p] = fmaf(hp.y, x0, acc[2][p]); acc[3][p] = fmaf(hp.x, x0, acc[3][p]);
            acc[0][p] = fmaf(hc.x, x1, acc[0][p]); acc[1][p] = fmaf(hp.w, x1, acc[1][p]);
            acc[2][p] = fmaf(hp.z, x1, acc[2][p]); acc[3][p] = fmaf(hp.y, x1, acc[3][p]);
            acc[0][p] = fmaf(hc.y, x2, acc[0][p]); acc[1][p] = fmaf(hc.x, x2, acc[1][p]);
            acc[2][p] = fmaf(hp.w, x2, acc[2][p]); acc[3][p] = fmaf(hp.z, x2, acc[3][p]);
            acc[0][p] = fmaf(hc.z, x3, acc[0][p]); acc[1][p] = fmaf(hc.y, x3, acc[1][p]);
            acc[2][p] = fmaf(hc.x, x3, acc[2][p]); acc[3][p] = fmaf(hp.w, x3, acc[3][p]);
            hprev[p] = hc;
        }
    }
    // fine index within the tile: (4u + r)*4 + p ; transposed so that the copy-out is conflict-light and coalesced
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int p = 0; p < OS; ++p) O[(4 * u + r) * OS + p] = acc[r][p];
    __syncthreads();
    float* ob = g_pre + (size_t)b * N;
    const int64_t m0 = (int64_t)j0 * OS;
    for (int e = tid; e < OSC_TILE * OS; e += 256) {
        const int64_t m = m0 + e;
        if (m < N) ob[m] = O[e];
    }
}

// partials -> g_wsel:  row k receives interval k (as row0) and interval k-1 (as row1); rows beyond
// Fw-1 were clamped onto Fw-1.  d pre / d wsel = (n_tab-1) * d pre / d p.
__global__ void osc_wsel_reduce_kernel(const float* __restrict__ part, float* __restrict__ g_wsel, int B, int Fw,
                                       int nint, int n_tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Fw) return;
    const int b = idx / Fw, k = idx - b * Fw;
    float acc = 0.f;
    for (int r = 0; r < nint; ++r) {
        const int ra = r > Fw - 1 ? Fw - 1 : r;
        const int rb = r + 1 > Fw - 1 ? Fw - 1 : r + 1;
        if (ra == k) acc += part[((size_t)b * nint + r) * 2 + 0];
        if (rb == k) acc += part[((size_t)b * nint + r) * 2 + 1];
    }
    g_wsel[idx] = acc * (float)(n_tab - 1);
}

static int osc_check(int B, int Tp, int phase_hop, int Fw, int w_hop, int n_tab, int L, int os, int K,
                     const float* taps) {
    if (B < 1 || Tp < 1 || phase_hop < 1 || Fw < 1 || w_hop < 1 || n_tab < 2 || L < 2 || os < 1)
        return fail(GOLF_EINVAL, "glottal_osc: bad size");
    if (L > 16384) return fail(GOLF_EUNSUPPORTED, "glottal_osc: table length %d > 16384", L);
    if (os > 1 && (!taps || K < 1 || (K & 1) == 0))
        return fail(GOLF_EINVAL, "glottal_osc: oversampling needs an odd number of decimation taps");
    if (os > 64) return fail(GOLF_EUNSUPPORTED, "glottal_osc: oversampling %d > 64", os);
    return GOLF_OK;
}

}  // namespace golf

using namespace golf;

extern "C" size_t golf_glottal_osc_workspace_bytes(int B, int Tp, int phase_hop, int Fw, int w_hop, int L, int os) {
    if (B < 1 || Tp < 1 || phase_hop < 1 || Fw < 1 || w_hop < 1 || os < 1) return 0;
    OscGeom g;
    osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
    return g.total;
}

extern "C" int golf_glottal_osc_fwd_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                        const float* wsel, int Fw, int w_hop, const float* table, int n_tab, int L,
                                        int os, int equal_energy, const float* taps, int K, float* pre, float* out,
                                        int64_t out_stride, int B, int Tout, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = osc_check(B, Tp, phase_hop, Fw, w_hop, n_tab, L, os, K, taps)) return rc;
    if (!phase || !wsel || !table || !out) return fail(GOLF_EINVAL, "glottal_osc_fwd: null pointer");
    OscGeom g;
    osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
    const int tout = os > 1 ? (g.N - 1) / os + 1 : g.N;
    if (Tout != tout) return fail(GOLF_EINVAL, "glottal_osc_fwd: Tout=%d, expected %d", Tout, tout);
    if (phase_stride < Tp || out_stride < Tout) return fail(GOLF_EINVAL, "glottal_osc_fwd: row stride too small");
    if (!ws || ws_bytes < g.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "glottal_osc_fwd: workspace needs %zu bytes, 256-aligned (got %zu)", g.total,
                    ws_bytes);
    if (g.ntile > 256) return fail(GOLF_EUNSUPPORTED, "glottal_osc_fwd: Tp=%d > 262144 coarse phase samples", Tp);
    hipStream_t st = (hipStream_t)stream;
    double* Cw = (double*)((char*)ws + g.off_cw);
    double* Ttot = (double*)((char*)ws + g.off_ttot);
    hipLaunchKernelGGL(osc_phase_tile_kernel, dim3(g.ntile, B), dim3(256), 0, st, phase, phase_stride, Cw, Ttot, Tp,
                       g.P, os, g.ntile);
    GOLF_LAUNCH_CHECK();
    float* fine = os > 1 ? (pre ? pre : (float*)((char*)ws + g.off_pre)) : out;
    const int64_t fine_stride = os > 1 ? g.N : out_stride;
    const size_t lds = sizeof(float) * 2 * (size_t)(L + 1);
    hipLaunchKernelGGL((osc_render_kernel<0>), dim3(g.nint, B), dim3(OSC_RENDER_THREADS), lds, st, phase, phase_stride,
                       (const double*)Cw, (const double*)Ttot, g.ntile, wsel, Fw, table, n_tab, L, Tp, g.P, os, g.hop_t,
                       g.N, equal_energy, fine, fine_stride, (const float*)nullptr, (float*)nullptr);
    GOLF_LAUNCH_CHECK();
    if (os > 1) {
        const int half = (K - 1) / 2;
        const int dmin = -((half + os - 1) / os);  // floor(-half/os)
        const int dmax = half / os;
        const int nq = dmax - dmin + 1;             // taps per polyphase branch (upper bound)
        const int ngrp = (nq + 2) / 4;
        int RS4 = OSC_TILE / 4 + ngrp + 2;
        while (RS4 % 32 != 2) ++RS4;                // (ph,i&3) sub-arrays land 2 banks apart: conflict-free fill at os=4
        const int hoff = (os * 4 * RS4 + 3) & ~3;
        const size_t lds3 = sizeof(float) * ((size_t)hoff + (size_t)os * (ngrp * 4 + 8));
        if (lds3 > 160 * 1024) return fail(GOLF_EUNSUPPORTED, "glottal_osc_fwd: %d taps x os %d exceed LDS", K, os);
        hipLaunchKernelGGL(osc_decimate_kernel, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), lds3, st,
                           (const float*)fine, g.N, taps, K, os, out, out_stride, Tout, RS4, dmin, ngrp);
        GOLF_LAUNCH_CHECK();
    }
    return GOLF_OK;
}

extern "C" int golf_glottal_osc_bwd_wsel_f32(const float* g_out, int64_t g_out_stride, const float* phase,
                                             int64_t phase_stride, int Tp, int phase_hop, const float* wsel, int Fw,
                                             int w_hop, const float* table, int n_tab, int L, int os, int equal_energy,
                                             const float* taps, int K, float* g_wsel, int B, int Tout, void* ws,
                                             size_t ws_bytes, void* stream) {
    if (int rc = osc_check(B, Tp, phase_hop, Fw, w_hop, n_tab, L, os, K, taps)) return rc;
    if (!g_out || !phase || !wsel || !table || !g_wsel) return fail(GOLF_EINVAL, "glottal_osc_bwd: null pointer");
    OscGeom g;
    osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
    const int tout = os > 1 ? (g.N - 1) / os + 1 : g.N;
    if (Tout != tout) return fail(GOLF_EINVAL, "glottal_osc_bwd: Tout=%d, expected %d", Tout, tout);
    if (!ws || ws_bytes < g.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "glottal_osc_bwd: workspace needs %zu bytes, 256-aligned (got %zu)", g.total,
                    ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    const double* Cw = (const double*)((char*)ws + g.off_cw);  // still valid from the forward
    float* g_pre = (float*)((char*)ws + g.off_pre);
    float* part = (float*)((char*)ws + g.off_part);
    if (os == 4) {
        const int half = (K - 1) / 2;
        const int dmin = -((half + os - 1) / os);
        const int dmax = half / os;
        const int nq = dmax - dmin + 1;
        const int ngrp = (nq + 2) / 4;
        int RS4 = OSC_TILE / 4 + ngrp + 2;
        while (RS4 % 32 != 8) ++RS4;
        const int hoff = (4 * RS4 + 3) & ~3;
        const size_t ldsT = sizeof(float) * ((size_t)hoff + (size_t)os * (ngrp * 4 + 8) + (size_t)OSC_TILE * os);
        if (ldsT > 160 * 1024) return fail(GOLF_EUNSUPPORTED, "glottal_osc_bwd: %d taps exceed LDS", K);
        hipLaunchKernelGGL(osc_decimate_T4_kernel, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), ldsT, st,
                           g_out, g_out_stride, Tout, taps, K, g_pre, g.N, RS4, dmax, ngrp);
        GOLF_LAUNCH_CHECK();
    } else if (os > 1) {
        const int64_t n = (int64_t)B * g.N;
        hipLaunchKernelGGL(osc_decimate_T_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, g_out,
                           g_out_stride, Tout, taps, K, os, g_pre, g.N, B);
        GOLF_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemcpy2DAsync(g_pre, sizeof(float) * g.N, g_out, sizeof(float) * g_out_stride,
                                        sizeof(float) * g.N, B, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail((int)e, "glottal_osc_bwd: copy failed: %s", hipGetErrorString(e));
    }
    size_t lds = sizeof(float) * 2 * (size_t)(L + 1);
    if (lds < sizeof(float) * 2 * OSC_RENDER_THREADS) lds = sizeof(float) * 2 * OSC_RENDER_THREADS;
    const double* Ttot = (const double*)((char*)ws + g.off_ttot);
    hipLaunchKernelGGL((osc_render_kernel<1>), dim3(g.nint, B), dim3(OSC_RENDER_THREADS), lds, st, phase, phase_stride,
                       Cw, Ttot, g.ntile, wsel, Fw, table, n_tab, L, Tp, g.P, os, g.hop_t, g.N, equal_energy,
                       (float*)nullptr, (int64_t)0, (const float*)g_pre, part);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(osc_wsel_reduce_kernel, dim3((unsigned)ceil_div(B * Fw, 256)), dim3(256), 0, st,
                       (const float*)part, g_wsel, B, Fw, g.nint, n_tab);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
