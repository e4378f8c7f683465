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
    size_t off_cw, off_pre, off_part, total;
};

static void osc_geom(int B, int Tp, int phase_hop, int Fw, int w_hop, int os, OscGeom* g) {
    g->P = phase_hop * os;
    g->N = g->P > 1 ? (Tp - 1) * g->P + 1 : Tp;
    g->hop_t = w_hop * os;
    g->nint = (int)ceil_div(g->N, g->hop_t);
    size_t o = 0;
    g->off_cw = o;   o = align_up(o + sizeof(double) * (size_t)B * Tp, 256);
    g->off_pre = o;  o = align_up(o + sizeof(float) * (size_t)B * g->N, 256);
    g->off_part = o; o = align_up(o + sizeof(float) * (size_t)B * g->nint * 2, 256);
    g->total = o;
}

// ---- O1 ---------------------------------------------------------------------------------------
// Cw[b][j] = frac( sum_{j'<j} segsum_{j'} ),  segsum_j = (P*p_j + d_j*P(P-1)/2)/os, d_j=(p_{j+1}-p_j)/P
__global__ __launch_bounds__(1024) void osc_phase_scan_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                              double* __restrict__ Cw, int Tp, int P, int os) {
    __shared__ double part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = phase + (size_t)b * phase_stride;
    double* cb = Cw + (size_t)b * Tp;
    const int nseg = Tp - 1;  // segments j = 0..Tp-2
    const int per = (nseg + 1023) / 1024;
    const int j0 = tid * per;
    const int j1 = j0 + per < nseg ? j0 + per : nseg;
    const double inv_os = 1.0 / (double)os;
    const double half = 0.5 * (double)(P - 1);
    double sum = 0.0;
    for (int j = j0; j < j1; ++j) {
        const double p0 = (double)pb[j], p1 = (double)pb[j + 1];
        sum += ((double)P * p0 + (p1 - p0) * half) * inv_os;  // d_j*P(P-1)/2 = (p1-p0)*(P-1)/2
    }
    sum -= floor(sum);
    part[tid] = sum;
    __syncthreads();
    // inclusive Hillis-Steele scan over 1024 partials (wrapped each step to keep magnitude small)
    for (int off = 1; off < 1024; off <<= 1) {
        double v = part[tid];
        if (tid >= off) v += part[tid - off];
        __syncthreads();
        part[tid] = v - floor(v);
        __syncthreads();
    }
    double run = tid > 0 ? part[tid - 1] : 0.0;
    for (int j = j0; j < j1; ++j) {
        cb[j] = run;
        const double p0 = (double)pb[j], p1 = (double)pb[j + 1];
        run += ((double)P * p0 + (p1 - p0) * half) * inv_os;
        run -= floor(run);
    }
    // exactly one thread's range ends at the last segment: it owns C for the final coarse sample
    if (nseg == 0) {
        if (tid == 0) cb[0] = 0.0;
    } else if (j0 < nseg && j1 == nseg) {
        cb[nseg] = run;
    }
}

// ---- O2 ---------------------------------------------------------------------------------------
// MODE 0: forward render (writes fine samples);  MODE 1: backward w.r.t. table_select_weight
// (reduces g_pre * d(pre)/d(p_row) over the interval into part[b][interval][2]).
template <int MODE>
__global__ __launch_bounds__(256) void osc_render_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                         const double* __restrict__ Cw,
                                                         const float* __restrict__ wsel, int Fw,
                                                         const float* __restrict__ table, int n_tab, int L, int Tp,
                                                         int P, int os, int hop_t, int N, int equal_energy,
                                                         float* __restrict__ dst, int64_t dst_stride,
                                                         const float* __restrict__ g_pre, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* row0 = smem;
    float* row1 = smem + (L + 1);
    const int r0 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
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
        for (int c = tid; c < L; c += 256) {
            if (MODE == 0) row[c] = t0[c] * (1.0f - p) + t1[c] * p;
            else row[c] = t1[c] - t0[c];
        }
    }
    __syncthreads();
    if (tid == 0) { row0[L] = row0[0]; row1[L] = row1[0]; }
    __syncthreads();
    const float* pb = phase + (size_t)b * phase_stride;
    const double* cb = Cw + (size_t)b * Tp;
    const int m_lo = r0 * hop_t;
    const int m_hi = m_lo + hop_t < N ? m_lo + hop_t : N;
    const float inv_hop_t = 1.0f / (float)hop_t;
    const float inv_P = 1.0f / (float)P;
    const double inv_os = 1.0 / (double)os;
    float acc0 = 0.f, acc1 = 0.f;
    for (int m = m_lo + tid; m < m_hi; m += 256) {
        int j = m / P;
        int k = m - j * P;
        float p0, d;
        if (j >= Tp - 1) {  // last coarse sample (k == 0) or P == 1 tail
            j = Tp - 1;
            k = m - j * P;
            p0 = pb[j];
            d = 0.f;
        } else {
            p0 = pb[j];
            d = (pb[j + 1] - p0) * inv_P;
        }
        // inclusive cumulative phase: C_j + ((k+1) p0 + d k(k+1)/2)/os, in fp64, wrapped
        const double kk = (double)k;
        double ph = cb[j] + ((kk + 1.0) * (double)p0 + (double)d * (kk * (kk + 1.0) * 0.5)) * inv_os;
        ph -= floor(ph);
        float c = (float)ph * (float)L;
        int c0 = (int)c;
        if (c0 > L - 1) c0 = L - 1;  // ph rounded up to 1.0f
        const float cf = c - (float)c0;
        const float rf = (float)(m - m_lo) * inv_hop_t;
        const float top = fmaf(cf, row0[c0 + 1] - row0[c0], row0[c0]);
        const float bot = fmaf(cf, row1[c0 + 1] - row1[c0], row1[c0]);
        float scale = 1.0f;
        if (equal_energy) scale = rsqrtf(fmaf((float)k, d, p0) / (float)os);
        if (MODE == 0) {
            dst[(size_t)b * dst_stride + m] = fmaf(rf, bot - top, top) * scale;
        } else {
            const float g = g_pre[(size_t)b * N + m] * scale;
            acc0 = fmaf(g * (1.0f - rf), top, acc0);
            acc1 = fmaf(g * rf, bot, acc1);
        }
    }
    if (MODE == 1) {
        __syncthreads();
        float* red = smem;  // reuse (rows are dead)
        red[tid] = acc0;
        red[256 + tid] = acc1;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) { red[tid] += red[tid + off]; red[256 + tid] += red[256 + tid + off]; }
            __syncthreads();
        }
        if (tid == 0) {
            part[((size_t)b * gridDim.x + r0) * 2 + 0] = red[0];
            part[((size_t)b * gridDim.x + r0) * 2 + 1] = red[256];
        }
    }
}

// ---- O3 ---------------------------------------------------------------------------------------
// out[o] = sum_k taps[k] * pre[o*os + k - half], zero padded.  Polyphase LDS tile:
// X[ph][i] = pre[(o0 + dmin + i)*os + ph].
#define OSC_TILE 1024
__global__ __launch_bounds__(256) void osc_decimate_kernel(const float* __restrict__ pre, int N,
                                                           const float* __restrict__ taps, int K, int os,
                                                           float* __restrict__ out, int64_t out_stride, int Tout,
                                                           int RS, int dmin) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int o0 = blockIdx.x * OSC_TILE;
    const float* pb = pre + (size_t)b * N;
    const int half = (K - 1) / 2;
    const int64_t m_lo = (int64_t)(o0 + dmin) * os;
    const int nload = RS * os;
    for (int e = tid; e < nload; e += 256) {
        const int64_t m = m_lo + e;
        const int ph = e % os, i = e / os;
        smem[ph * RS + i] = (m >= 0 && m < N) ? pb[m] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OSC_TILE / 256; ++r) {
        const int ol = tid + r * 256;
        const int o = o0 + ol;
        if (o >= Tout) break;
        float acc = 0.f;
        // u = k - half = os*d + ph, ph in [0,os)
        int d = dmin;
        int ph = (-half) - dmin * os;
        for (int k = 0; k < K; ++k) {
            acc = fmaf(taps[k], smem[ph * RS + (ol + d - dmin)], acc);
            if (++ph == os) { ph = 0; ++d; }
        }
        out[(size_t)b * out_stride + o] = acc;
    }
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
    hipStream_t st = (hipStream_t)stream;
    double* Cw = (double*)((char*)ws + g.off_cw);
    hipLaunchKernelGGL(osc_phase_scan_kernel, dim3(B), dim3(1024), 0, st, phase, phase_stride, Cw, Tp, g.P, os);
    GOLF_LAUNCH_CHECK();
    float* fine = os > 1 ? (pre ? pre : (float*)((char*)ws + g.off_pre)) : out;
    const int64_t fine_stride = os > 1 ? g.N : out_stride;
    const size_t lds = sizeof(float) * 2 * (size_t)(L + 1);
    hipLaunchKernelGGL((osc_render_kernel<0>), dim3(g.nint, B), dim3(256), lds, st, phase, phase_stride,
                       (const double*)Cw, wsel, Fw, table, n_tab, L, Tp, g.P, os, g.hop_t, g.N, equal_energy, fine,
                       fine_stride, (const float*)nullptr, (float*)nullptr);
    GOLF_LAUNCH_CHECK();
    if (os > 1) {
        const int half = (K - 1) / 2;
        const int dmin = -((half + os - 1) / os);          // floor(-half/os)
        const int dmax = half / os;
        int RS = OSC_TILE + dmax - dmin + 1;
        while (RS % 32 != 8) ++RS;                          // phase rows land 8 banks apart (os=4: conflict-free fill)
        const size_t lds3 = sizeof(float) * (size_t)RS * os;
        if (lds3 > 160 * 1024) return fail(GOLF_EUNSUPPORTED, "glottal_osc_fwd: %d taps x os %d exceed LDS", K, os);
        hipLaunchKernelGGL(osc_decimate_kernel, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), lds3, st,
                           (const float*)fine, g.N, taps, K, os, out, out_stride, Tout, RS, dmin);
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
    if (os > 1) {
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
    if (lds < sizeof(float) * 512) lds = sizeof(float) * 512;
    hipLaunchKernelGGL((osc_render_kernel<1>), dim3(g.nint, B), dim3(256), lds, st, phase, phase_stride, Cw, wsel, Fw,
                       table, n_tab, L, Tp, g.P, os, g.hop_t, g.N, equal_energy, (float*)nullptr, (int64_t)0,
                       (const float*)g_pre, part);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(osc_wsel_reduce_kernel, dim3((unsigned)ceil_div(B * Fw, 256)), dim3(256), 0, st,
                       (const float*)part, g_wsel, B, Fw, g.nint, n_tab);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
