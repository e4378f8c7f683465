// Indexed glottal-flow wavetable oscillator for gfx950 — GOLF's harmonic source.
//
// Replaces IndexedGlottalFlowTable.forward (reference models/synth.py:213-263) and
// GlottalFlowTable.generate (models/synth.py:124-177).  The reference materialises, at B=32:
// (B,21,2048) blended tables, >= 6 oversampled (B,191997) temporaries, a (B,191997,1,2) sampling
// grid for F.grid_sample, and runs torch.cumsum in fp32 over 192k samples.  Here:
//
//   O1  osc_phase_scan   per utterance: exclusive prefix of the per-segment phase advance.  The phase is kept
//                        in 64-bit FIXED POINT (2^64 = one cycle): wrapping is the integer overflow, sums are
//                        exact and associative (so the scan order cannot matter), and the per-sample work in O2
//                        is integer adds instead of fp64 multiply/floor chains.  Linear interpolation of the
//                        increment has a closed form inside a segment, so only the Tp coarse samples are
//                        scanned, not the N = (Tp-1)*hop*os+1 oversampled ones.
//   O2  osc_render       one workgroup per (utterance, control interval): the two blended table rows
//                        of that interval are staged in LDS (2 x (L+1) floats), every oversampled
//                        sample is then phase -> bilinear LDS lookup -> equal-energy scaling.
//   O3  osc_decimate     strided FIR (kazane.Decimate stand-in; taps are an input) with a polyphase,
//                        bank-conflict-free LDS tile.
//   The running phase is exact to ~1e-19 cycles per term (the reference's fp32 cumsum drifts ~3e-5 cycles).
#include "common.h"
#include "device_common.h"
#include "lpc_p1f.h"
#include <algorithm>
#include <cstdlib>

namespace golf {

struct OscGeom {
    int P;        // fine samples per coarse phase sample = phase_hop * os
    int N;        // oversampled length
    int hop_t;    // fine samples per control (table) frame = w_hop * os
    int nint;     // control intervals = ceil(N / hop_t)
    int ntile;    // phase-scan tiles of OSC_SCAN_TILE coarse samples
    int pre_stride;  // row stride of the internal oversampled buffer (multiple of 4 floats)
    size_t off_cw, off_ttot, off_pre, off_part, off_bfr, off_bf4, off_look, total;
};
#define OSC_SCAN_TILE 1024

static void osc_geom(int B, int Tp, int phase_hop, int Fw, int w_hop, int os, OscGeom* g) {
    g->P = phase_hop * os;
    g->N = g->P > 1 ? (Tp - 1) * g->P + 1 : Tp;
    g->hop_t = w_hop * os;
    g->nint = (int)ceil_div(g->N, g->hop_t);
    size_t o = 0;
    g->ntile = (int)ceil_div(Tp, OSC_SCAN_TILE);
    g->off_cw = o;   o = align_up(o + sizeof(unsigned long long) * (size_t)B * Tp, 256);
    g->off_ttot = o; o = align_up(o + sizeof(unsigned long long) * (size_t)B * g->ntile, 256);
    g->pre_stride = (g->N + 3) & ~3;
    g->off_pre = o;  o = align_up(o + sizeof(float) * (size_t)B * g->pre_stride, 256);
    g->off_part = o; o = align_up(o + sizeof(float) * (size_t)B * g->nint * 2, 256);
    g->off_bfr = o;  o = align_up(o + sizeof(float) * 4 * 16 * 64, 256);   // Toeplitz tap fragments of the fused backward (transposed FIR)
    g->off_bf4 = o;  o = align_up(o + sizeof(float) * 4 * 16 * 64, 256);   // ... and of the fused forward, four K-steps per 16-byte word (osc_fused2)
    // round 6: the single-pass phase scan of osc_fused2 (decoupled look-back): B launch counters + one tagged 16-byte entry per
    // (utterance, 2048-sample tile, wave) -- see OscLook
    g->off_look = o; o = align_up(o + 256 * ceil_div((size_t)B * 4, 256) + 16 * (size_t)B * ((size_t)(g->ntile + 1) / 2 + 1) * 8, 256);
    g->total = o;
}

// ---- O1 ---------------------------------------------------------------------------------------
// Fixed-point increments (both kernels use exactly these two functions, so their phases agree bit for bit):
//   inc_j(k) = a_j + k * d_j,  a_j = p_j/os,  d_j = (p_{j+1} - p_j)/(os*P)      [cycles per fine sample, Q0.64]
//   inclusive phase after fine step k of coarse sample j:  C_j + (k+1) a_j + d_j k(k+1)/2
//   segment total T_j = P a_j + d_j P(P-1)/2;   C_j = sum_{i<j} T_i   (all modulo 2^64 = modulo one cycle)
typedef unsigned long long u64;
__device__ __forceinline__ u64 osc_fix_a(float p, double scale_a) {   // scale_a = 2^64 / os
    return (u64)__double2ull_rn((double)p * scale_a);                  // p in [0, 0.5]: below 2^63
}
__device__ __forceinline__ u64 osc_fix_d(float p0, float p1, double scale_d) {  // scale_d = 2^64 / (os*P)
    return (u64)__double2ll_rn(((double)p1 - (double)p0) * scale_d);   // signed, two's complement
}
// The same two numbers by exponent arithmetic when os and P are powers of two (the fused kernels: os = P = 4).  A float
// is m 2^(e-150): p 2^(64 - log2 os) is a shift of the mantissa -- exact, and equal to osc_fix_a's rounded double product
// whenever that product is an integer (p >= 2^-39 cycles per sample; below that this truncates where that rounds to
// nearest, 2^-64 of a cycle; denormal increments count as 0).  8 integer instructions where the double-precision route
// takes ~8 fp64 ones (v_ldexp_f64, v_floor_f64, v_rndne_f64, two v_cvt_u32_f64 ...); measured: fewer instructions, the
// same kernel time -- osc_fused_kernel is bound by workgroup latency x occupancy, DESIGN.md 4.3 "Round 3".
__device__ __forceinline__ u64 osc_fix_a_pow2(float p, int log2os) {
    // mantissa (hidden bit included) at the top of a 64-bit word, shifted down by what the exponent says:
    // p = 1.m 2^(e-127)  ->  p 2^(64 - log2os) = (1.m 2^63) >> (126 + log2os - e)
    const unsigned bits = __float_as_uint(p);
    const int e = (int)((bits >> 23) & 0xffu);
    int sft = 126 + log2os - e;                                        // 0 .. 63 for 2^-63 os <= p < 2 / os ... clamp the rest
    sft = sft < 0 ? 0 : (sft > 63 ? 63 : sft);
    const unsigned hi = (int)bits >= 0x00800000 ? ((bits << 8) | 0x80000000u) : 0u;   // zero, denormals, negatives -> 0
    return ((u64)hi << 32) >> sft;
}
__device__ __forceinline__ u64 osc_fix_d_pow2(u64 a0, u64 a1, int log2P) {   // (a1 - a0) / P, two's complement
    return (u64)((long long)(a1 - a0) >> log2P);
}
// One workgroup per (tile of 1024 coarse samples, utterance):
//   Cloc[b][j] = sum of the segment totals of the tile before j   (exclusive)
//   Ttot[b][tile] = tile total
// The render kernel adds the (<= ntile-term) prefix of Ttot itself.  Coalesced loads, wave shuffles.
// Inclusive wave scan of 64-bit integers on the DPP network: row_shr 1/2/4/8 inside each row of 16 lanes (lanes without
// a source read 0), then row_bcast15 / row_bcast31 across rows -- 6 steps of two moves + one 64-bit add.  (__shfl_up
// goes through ds_bpermute: ~60 instructions and 12 serial LDS-crossbar round trips per scan.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_add_u64(u64 v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + (((u64)(unsigned)hi << 32) | (u64)(unsigned)lo);
}
__device__ __forceinline__ u64 wave_incl_scan(u64 v, int /*lane*/) {
    v = dpp_add_u64<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add_u64<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add_u64<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add_u64<0x118, 0xF>(v);  // row_shr:8
    v = dpp_add_u64<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_u64<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

__global__ __launch_bounds__(256) void osc_phase_tile_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                             u64* __restrict__ Cloc, u64* __restrict__ Ttot,
                                                             int Tp, int P, int os, int ntile) {
    __shared__ float ps[OSC_SCAN_TILE + 1];
    __shared__ u64 wsum[4];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const float* pb = phase + (size_t)b * phase_stride;
    const int j0 = tile * OSC_SCAN_TILE;
    for (int u = tid; u < OSC_SCAN_TILE + 1; u += 256) {
        const int j = j0 + u;
        ps[u] = pb[j < Tp ? j : Tp - 1];
    }
    __syncthreads();
    const double scale_a = 18446744073709551616.0 / (double)os;
    const double scale_d = scale_a / (double)P;
    const u64 tri = (u64)P * (u64)(P - 1) / 2;
    u64 seg[4];
    u64 tsum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = tid * 4 + r;
        const int j = j0 + u;
        const u64 a = osc_fix_a(ps[u], scale_a), d = osc_fix_d(ps[u], ps[u + 1], scale_d);
        seg[r] = j < Tp - 1 ? (u64)P * a + d * tri : 0;  // segments 0..Tp-2
        tsum += seg[r];
    }
    const u64 incl = wave_incl_scan(tsum, lane);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u64 base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    u64 run = base + incl - tsum;  // exclusive prefix of this thread's first segment
    u64* cb = Cloc + (size_t)b * Tp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + tid * 4 + r;
        if (j < Tp) cb[j] = run;
        run += seg[r];
    }
    if (tid == 255) Ttot[(size_t)b * ntile + tile] = run;
}

// Long inputs (more than 256 tiles = 262144 coarse phase samples, e.g. > 10.9 s of audio with a per-sample phase): the
// consumers keep the tile prefix in a 256-entry LDS array, so for longer inputs the prefix is folded into Cloc itself --
// Ttot becomes its own exclusive prefix, every Cloc[j] gets its tile's offset added, Ttot is cleared -- and the
// consumers' LDS prefix is all zeros (they clamp the tile index).  Two light extra passes, only on this path.
__global__ __launch_bounds__(64) void osc_long_prefix_kernel(u64* __restrict__ Ttot, int ntile) {
    u64* t = Ttot + (size_t)blockIdx.x * ntile;
    const int lane = threadIdx.x;
    u64 carry = 0;
    for (int base = 0; base < ntile; base += 64) {
        const int i = base + lane;
        const u64 v = i < ntile ? t[i] : 0;
        const u64 incl = wave_incl_scan(v, lane);
        if (i < ntile) t[i] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}
__global__ void osc_long_add_kernel(u64* __restrict__ Cloc, const u64* __restrict__ Toff, int Tp, int ntile, int B) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Tp) return;
    const int b = (int)(idx / Tp), j = (int)(idx - (int64_t)b * Tp);
    Cloc[idx] += Toff[(size_t)b * ntile + j / OSC_SCAN_TILE];
}

// phase tile scan of every utterance (+ the long-input passes): Cloc, Ttot as every consumer expects them
static int launch_phase_tiles(const float* phase, int64_t phase_stride, u64* Cw, u64* Ttot, int Tp, int P, int os,
                              int ntile, int B, hipStream_t st) {
    hipLaunchKernelGGL(osc_phase_tile_kernel, dim3(ntile, B), dim3(256), 0, st, phase, phase_stride, Cw, Ttot, Tp, P, os,
                       ntile);
    GOLF_LAUNCH_CHECK();
    if (ntile > 256) {
        hipLaunchKernelGGL(osc_long_prefix_kernel, dim3(B), dim3(64), 0, st, Ttot, ntile);
        GOLF_LAUNCH_CHECK();
        hipLaunchKernelGGL(osc_long_add_kernel, dim3((unsigned)ceil_div((int64_t)B * Tp, 256)), dim3(256), 0, st, Cw,
                           (const u64*)Ttot, Tp, ntile, B);
        GOLF_LAUNCH_CHECK();
        hipError_t e = hipMemsetAsync(Ttot, 0, sizeof(u64) * (size_t)B * ntile, st);
        if (e != hipSuccess) return fail((int)e, "phase scan: memset failed: %s", hipGetErrorString(e));
    }
    return GOLF_OK;
}

// ---- O2 ---------------------------------------------------------------------------------------
// MODE 0: forward render (writes fine samples);  MODE 1: backward w.r.t. table_select_weight
// (reduces g_pre * d(pre)/d(p_row) over the interval into part[b][interval][2]).
#define OSC_RENDER_THREADS 512  // 4 blocks/CU: the 640 blocks of the B=32 config run in one round
// PT = fine samples per coarse phase sample when known at compile time (0: runtime P).  FLAGS >= 0: the table length
// is a power of two and equal_energy == FLAGS, both known at compile time -- as runtime flags they were wave-uniform
// branches around every fine sample, which kept hipcc from overlapping the LDS lookups of one sample with the
// arithmetic of the previous one (75 -> ~45 instructions per fine sample).
template <int MODE, int PT, int FLAGS = -1>
__global__ __launch_bounds__(OSC_RENDER_THREADS) void osc_render_kernel(
    const float* __restrict__ phase, int64_t phase_stride, const u64* __restrict__ Cloc,
    const u64* __restrict__ Ttot, int ntile, const float* __restrict__ wsel, int Fw,
    const float* __restrict__ table, int n_tab, int L, int Tp, int P, int os, int hop_t, int N, int equal_energy,
    float* __restrict__ dst, int64_t dst_stride, const float* __restrict__ g_pre, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ u64 toff[256];  // prefix of the tile totals (ntile <= 256 tiles = 262144 coarse samples)
    float* row0 = smem;
    float* row1 = smem + (L + 1);
    constexpr int NTH = OSC_RENDER_THREADS;
    const int r0 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    // exclusive prefix of the tile totals: one load per thread + a wave scan.  (First version: thread t summed
    // Ttot[0..t) in a loop of dependent global loads, up to 46 serial L2 round trips at the head of every workgroup
    // -- most of this kernel's duration.)
    __shared__ u64 twsum[4];
    if (tid < 256) {
        const u64 v = tid < ntile ? Ttot[(size_t)b * ntile + tid] : 0;
        const u64 incl = wave_incl_scan(v, tid & 63);
        if ((tid & 63) == 63) twsum[tid >> 6] = incl;
        toff[tid] = incl - v;
    }
    __syncthreads();
    if (tid < 256) {
        u64 base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += twsum[w];
        toff[tid] += base;
    }
    // ---- stage rows r0, r0+1 (blended tables in MODE 0; table differences in MODE 1).  All table loads of a batch
    // (2 rows x 2 tables x 4 columns per thread) are issued before the first blend: clamped indices, no branches.
    {
        const float* t0[2];
        float pw[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            int k = r0 + rr;
            if (k > Fw - 1) k = Fw - 1;
            const float idx = wsel[(size_t)b * Fw + k] * (float)(n_tab - 1);
            int i0 = (int)idx;
            i0 = i0 < 0 ? 0 : (i0 > n_tab - 2 ? n_tab - 2 : i0);
            pw[rr] = idx - (float)i0;
            t0[rr] = table + (size_t)i0 * L;
        }
        constexpr int SU = 4;
        for (int cb0 = 0; cb0 < L + 1; cb0 += SU * NTH) {  // column L = wrap-around copy of column 0
            float va[2][SU], vb[2][SU];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    int c = cb0 + u * NTH + tid;
                    c = c > L ? L : c;
                    c = c == L ? 0 : c;
                    va[rr][u] = t0[rr][c];
                    vb[rr][u] = t0[rr][L + c];
                }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int c = cb0 + u * NTH + tid;
                    const float v = MODE == 0 ? va[rr][u] * (1.0f - pw[rr]) + vb[rr][u] * pw[rr] : vb[rr][u] - va[rr][u];
                    if (c <= L) (rr ? row1 : row0)[c] = v;
                }
        }
    }
    __syncthreads();
    const float* pb = phase + (size_t)b * phase_stride;
    const u64* cb = Cloc + (size_t)b * Tp;
    const int m_lo = r0 * hop_t;
    const int m_hi = m_lo + hop_t < N ? m_lo + hop_t : N;
    const float inv_hop_t = 1.0f / (float)hop_t;
    const float inv_P = 1.0f / (float)P;
    const float inv_osf = 1.0f / (float)os;
    const double scale_a = 18446744073709551616.0 / (double)os;
    const double scale_d = scale_a / (double)P;
    // One thread per COARSE phase sample of the interval, walking its P fine samples with two integer adds each
    // (ph += inc; inc += d).  The first version evaluated the closed form in fp64 for every fine sample (77 VALU
    // instructions per sample, 42 % VALU-active with the rest of the time waiting for its four loads per sample).
    const int j_lo = m_lo / P, j_hi = (m_hi - 1) / P;  // coarse samples overlapping [m_lo, m_hi)
    const int lshift = (FLAGS >= 0 || (L & (L - 1)) == 0) ? 31 - __clz(L) : -1;  // log2(L) for power-of-two tables
    const bool ee = FLAGS >= 0 ? FLAGS == 1 : equal_energy != 0;
    float acc0 = 0.f, acc1 = 0.f;
    for (int j = j_lo + tid; j <= j_hi; j += NTH) {
        // coarse sample j (the final point: j = Tp-1 has only k = 0, and d == 0 because j+1 clamps to j)
        const int jc = j < Tp - 1 ? j : Tp - 1;
        const int jn = jc + 1 < Tp ? jc + 1 : Tp - 1;
        const float p0 = pb[jc], p1 = pb[jn];
        const float d = (p1 - p0) * inv_P;
        u64 inc = osc_fix_a(p0, scale_a);
        const u64 dinc = osc_fix_d(p0, p1, scale_d);
        u64 ph = cb[jc] + toff[min(jc / OSC_SCAN_TILE, 255)];   // > 256 tiles: prefix already in cb, toff = 0
        const int mb = jc * P;
        float o4[PT > 0 ? PT : 1];
        const int pcount = PT > 0 ? PT : P;
#pragma unroll
        for (int k = 0; k < pcount; ++k) {
            ph += inc;   // inclusive cumulative phase of fine sample mb + k
            inc += dinc;
            const int m = mb + k;
            const bool in = m >= m_lo && m < m_hi;
            // table position: (ph / 2^64) * L from the top 32 phase bits, exact; a power-of-two table needs no multiply
            const unsigned hi = (unsigned)(ph >> 32);
            int c0;
            float cf;
            if (FLAGS >= 0 || lshift >= 0) {
                c0 = (int)(hi >> (32 - lshift));
                cf = (float)((hi << lshift) >> 8) * (1.0f / 16777216.0f);
            } else {
                const u64 pos = (u64)hi * (u64)L;
                c0 = (int)(pos >> 32);
                cf = (float)((unsigned)pos >> 8) * (1.0f / 16777216.0f);
            }
            const float rf = (float)(m - m_lo) * inv_hop_t;
            const float a00 = row0[c0], a01 = row0[c0 + 1], a10 = row1[c0], a11 = row1[c0 + 1];
            const float top = fmaf(cf, a01 - a00, a00);
            const float bot = fmaf(cf, a11 - a10, a10);
            float scale = 1.0f;
            if (ee) scale = rsqrtf(fmaf((float)k, d, p0) * inv_osf);
            if (MODE == 0) {
                const float v = fmaf(rf, bot - top, top) * scale;
                if (PT > 0) o4[k] = v;
                else if (in) dst[(size_t)b * dst_stride + m] = v;
            } else if (in) {
                const float g = g_pre[(size_t)b * N + m] * scale;
                acc0 = fmaf(g * (1.0f - rf), top, acc0);
                acc1 = fmaf(g * rf, bot, acc1);
            }
        }
        if (MODE == 0 && PT > 0) {
            float* o = dst + (size_t)b * dst_stride + mb;
            if (PT == 4 && mb >= m_lo && mb + 4 <= m_hi && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                *reinterpret_cast<float4*>(o) = make_float4(o4[0], o4[1], o4[2], o4[3]);  // one 16-byte store per thread
            } else {
#pragma unroll
                for (int k = 0; k < PT; ++k)
                    if (mb + k >= m_lo && mb + k < m_hi) o[k] = o4[k];
            }
        }
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
template <int OST>  // OST > 0: oversampling factor known at compile time (index arithmetic becomes shifts); 0: runtime
__global__ __launch_bounds__(256) void osc_decimate_kernel(const float* __restrict__ pre, int N, int64_t pre_stride,
                                                           const float* __restrict__ taps, int K, int os_rt,
                                                           float* __restrict__ out, int64_t out_stride, int Tout,
                                                           int RS4, int dmin, int ngrp, int vec4,
                                                           const float* __restrict__ addend, int64_t addend_stride,
                                                           int Tadd) {
    const int os = OST > 0 ? OST : os_rt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // layout: X = smem[0 .. os*4*RS4), H = 16-aligned after it: H[ph][ngrp*4 + 4] (3 leading zeros + taps + zero tail)
    float* X = smem;
    const int hoff = (os * 4 * RS4 + 3) & ~3;
    float* H = smem + hoff;
    const int HS = ngrp * 4 + 8;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int o0 = blockIdx.x * OSC_TILE;
    const float* pb = pre + (size_t)b * pre_stride;
    const int half = (K - 1) / 2;
    const int64_t m_lo = (int64_t)(o0 + dmin) * os;
    const int span = OSC_TILE + ngrp * 4 + 4;  // polyphase indices staged per phase
    if (OST == 4 && vec4 && m_lo >= 0 && m_lo + 4 * (int64_t)span <= N) {
        // fill of an interior tile (all but the first and last of an utterance), os = 4, 16-byte aligned rows: one
        // 16-byte load brings the four polyphase components of a coarse index (m_lo is a multiple of 4), so the index
        // arithmetic is paid once per 4 elements: ~4 loads per thread instead of ~17 (the fill was about as many VALU
        // instructions as the FIR itself).  No masks at all here: with `cond ? v : 0` in front of the LDS stores
        // hipcc turns the selects into branches and sinks (and splits) the loads into them.
        // (__builtin_amdgcn_raw_buffer_load_b128 is miscompiled by this hipcc: it emits buffer_load_dword and
        // splats the one dword over the four components -- hence plain loads.)
        const float4* src = reinterpret_cast<const float4*>(pb + m_lo);
        auto put = [&](int i, const float4& q) {
            float* xp = X + (i & 3) * RS4 + (i >> 2);
            xp[0 * 4 * RS4] = q.x;
            xp[1 * 4 * RS4] = q.y;
            xp[2 * 4 * RS4] = q.z;
            xp[3 * 4 * RS4] = q.w;
        };
        // span >= OSC_TILE = 4 x 256: four unconditional batches, then the few groups that are left
        static_assert(OSC_TILE == 4 * 256, "fill assumes 4 full batches");
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[u * 256 + tid];
#pragma unroll
        for (int u = 0; u < 4; ++u) put(u * 256 + tid, v[u]);
        for (int i = OSC_TILE + tid; i < span; i += 256) put(i, src[i]);
    } else {
        // fill: 8 bounds-checked loads per thread in flight at a time (a guarded `cond ? pb[m] : 0` made hipcc branch
        // and wait for every one of the ~17 loads per thread in turn: most of this kernel's former 18.9 us)
        const BufRow prow(pb, N);
        const int total = span * os;
        for (int e0 = 0; e0 < total; e0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t m = m_lo + e0 + u * 256 + tid;
                v[u] = prow.ld((int)(m < 0 ? -1 : m));   // before the start and past the end read 0
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256 + tid;
                if (e < total) {
                    const int ph = e % os, i = e / os;
                    X[(ph * 4 + (i & 3)) * RS4 + (i >> 2)] = v[u];
                }
            }
        }
    }
    // H[ph][3 + q] = tap of (ph, d = dmin + q), zero elsewhere
    for (int e = tid; e < os * HS; e += 256) {
        const int ph = e / HS, q = e - ph * HS - 3;
        const int k = half + os * (dmin + q) + ph;
        H[e] = (q >= 0 && k >= 0 && k < K) ? taps[k] : 0.f;
    }
    __syncthreads();
    const int u = tid;  // outputs o0 + 4u .. 4u+3
    // optional fused `+ addend` (the decoder's harm_osc + noise, models/sf.py:53): loads issued before the FIR so they
    // are free; a null addend is a zero-length descriptor, which reads 0 without touching memory -- no branch
    const BufRow arow(addend ? addend + (size_t)b * addend_stride : nullptr, addend ? Tadd : 0);
    const float ad0 = arow.ld(o0 + 4 * u), ad1 = arow.ld(o0 + 4 * u + 1), ad2 = arow.ld(o0 + 4 * u + 2),
                ad3 = arow.ld(o0 + 4 * u + 3);
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
    if (o < Tout) ob[o] = acc0 + ad0;
    if (o + 1 < Tout) ob[o + 1] = acc1 + ad1;
    if (o + 2 < Tout) ob[o + 2] = acc2 + ad2;
    if (o + 3 < Tout) ob[o + 3] = acc3 + ad3;
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

// ---- O2+O3 fused (the GOLF configuration: phase at hop 1, 4x oversampling, power-of-two table) --------------------
// The three-kernel path above moves 105 MB for 12 MB of algorithmic traffic at B=32 (a 64-bit phase prefix per coarse
// sample written and read back, the 4x oversampled signal on a round trip through HBM) and -- what the counters showed
// to matter more -- spends 433 VALU lane-instructions per output sample at 75 % VALU-busy: it is instruction-issue
// bound.  Here one workgroup owns a tile of 2048 (or 1536) output samples of one utterance:
//   0. (separate, tiny) osc_tile_totals_kernel: phase advance of every 256-sample stretch and of every 2048-sample tile
//   1. base phase of the tile = sum of the earlier stretches' totals (exact: Q0.64 integers, order cannot matter)
//   2. the tile's coarse phase samples (+ the decimator's halo) are scanned in the block; a thread keeps the converted
//      increments of its 5 consecutive coarse samples in registers for step 3
//   3. render into LDS.  Table rows are staged as PAIRS (row_k[c], row_{k+1}[c] - row_k[c]): two 8-byte reads bring
//      what a bilinear lookup needs, and interpolating along the control frame first makes it 2 + 2 instructions;
//      the equal-energy factor rsqrt(p) is linear over the 4 fine samples of a coarse sample to 1e-6 whenever p moves by
//      less than 0.2 % per sample (speech f0; otherwise the exact v_rsq_f32 runs)
//   4. the 129-tap polyphase FIR runs on the MATRIX pipe: per polyphase branch the outputs of a 256-sample stretch are the
//      product of 16 overlapping signal windows (A: 16 x 48) with a banded Toeplitz matrix of the branch's 33 taps
//      (B: 48 x 16), accumulated over the 4 branches in exact fp32 (v_mfma_f32_16x16x4_f32, 48 of them per wave).
//      132 VALU multiply-adds per output become 12 LDS reads.
// Same exact phases as the three-kernel path; the values agree to 1e-6 (tests/test_gpu_osc.py).  The kernel itself (round 5's
// osc_fused2_kernel) follows the totals kernel below.
#define OSCF_TO 2048          // tile of the totals launch (and of the fused backward, OSCB_TO)
#define OSCF_MAXROWS 4
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// (block (0,0) also lays the decimation taps out as the MFMA B fragments of the fused kernels: lane `lane` of K-step kk of
//  branch ph holds the tap at d = dmin + (4*kk + lane/16 - lane%16), 0 outside the filter)
// threads of the totals kernel: whole passes of the workgroup over a tile
constexpr int osct_threads(int TO) { return TO % 512 == 0 ? 512 : (TO % 384 == 0 ? 384 : 256); }
template <int TO>   // coarse samples per tile: the forward's OSCF_TO, the backward's OSCB_TO
__global__ __launch_bounds__(osct_threads(TO)) void osc_tile_totals_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                              u64* __restrict__ Ttot, int Tp, int P, int os, int ntile,
                                                              const float* __restrict__ taps, int K, int dmin, int KS,
                                                              float* __restrict__ Bfr, int dmax,
                                                              float* __restrict__ Bf4, u64* __restrict__ T256) {
    constexpr int OSCT_THREADS = osct_threads(TO);
    __shared__ u64 wsum[OSCT_THREADS / 64];
    light_wave_priority();
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tile == 0 && b == 0) {
        const int half = (K - 1) / 2;
        for (int e = tid; e < 4 * KS * 64; e += OSCT_THREADS) {
            const int lane = e & 63, kk = (e >> 6) % KS, ph = (e >> 6) / KS;
            const int q = 4 * kk + (lane >> 4) - (lane & 15);
            // forward: tap of branch ph at d = dmin + q; backward (transposed FIR, osc_fused_bwd_kernel): at d = dmax - q.
            // The forward leaves both behind: a backward handed the untouched workspace needs no totals launch of its own.
            if (Bf4) {   // osc_fused2 reads four K-steps of a lane as one 16-byte word: [(ph * KS/4 + kk/4) * 64 + lane][kk % 4]
                const int d = dmin + q, k = half + 4 * d + ph;
                Bf4[(((ph * (KS >> 2) + (kk >> 2)) * 64 + lane) << 2) + (kk & 3)] = (q >= 0 && d >= dmin && k >= 0 && k < K) ? taps[k] : 0.f;
            }
            if (Bfr) {   // (the same 16-byte layout)
                const int d = dmax - q, k = half + 4 * d + ph;
                Bfr[(((ph * (KS >> 2) + (kk >> 2)) * 64 + lane) << 2) + (kk & 3)] =
                    (q >= 0 && d >= dmin && d <= dmax && k >= 0 && k < K) ? taps[k] : 0.f;
            }
        }
    }
    const BufRow prow(phase + (size_t)b * phase_stride, Tp);
    // A pure reduction over the tile.  A wave takes 256 CONSECUTIVE samples, so that its total is the total of one 256-sample
    // stretch (osc_fused2's base phase is a sum of those), and a thread 4 consecutive ones: one 16-byte load + one dword for the
    // successor of its last sample, 5 conversions for 4 segments.  (Round 2 - 4 form: lane-consecutive samples, p_j and p_{j+1}
    // loaded and converted separately -- 8 loads and 8 conversions per thread, 190 VALU instructions per wave, 4.7 % of the
    // B = 32 step's instructions in a kernel that only sums.  Only the fused kernels' configuration, os = P = 4, launches this.)
    constexpr int PER = TO / OSCT_THREADS;
    static_assert(PER == 4, "a thread's samples are one 16-byte word; a wave's, one 256-sample stretch");
    (void)os; (void)P;
    const int jq = tile * TO + (tid >> 6) * 256 + 4 * (tid & 63);
    const bool edge = tile * TO + TO > Tp - 1;               // (uniform) the row ends in this tile: segments 0 .. Tp-2 advance the phase
    float pq[4];
    if (!edge) {
        const auto q4 = __builtin_amdgcn_raw_buffer_load_b128(prow.rs, jq * 4, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) pq[r] = __uint_as_float(q4[r]);
    } else {                                                 // (dword loads: a 16-byte load that straddles the end of the row is not relied on)
#pragma unroll
        for (int r = 0; r < 4; ++r) pq[r] = prow.ld(jq + r);
    }
    const float pn = prow.ld(jq + 4);
    u64 av[5];
#pragma unroll
    for (int r = 0; r < 4; ++r) av[r] = osc_fix_a_pow2(pq[r], 2);
    av[4] = osc_fix_a_pow2(pn, 2);
    u64 tsum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const u64 d = osc_fix_d_pow2(av[r], av[r + 1], 2);
        const u64 sg = ((av[r] + d) << 2) + (d << 1);        // 4 a + 6 d
        tsum += (!edge || jq + r < Tp - 1) ? sg : 0;
    }
    const u64 incl = wave_incl_scan(tsum, tid & 63);
    if ((tid & 63) == 63) {
        wsum[tid >> 6] = incl;
        if (T256 && PER * 64 == 256) T256[((size_t)b * ntile + tile) * (OSCT_THREADS / 64) + (tid >> 6)] = incl;
    }
    __syncthreads();
    if (tid == 0) {
        u64 tot = 0;
#pragma unroll
        for (int w = 0; w < OSCT_THREADS / 64; ++w) tot += wsum[w];
        Ttot[(size_t)b * ntile + tile] = tot;
    }
}


#ifdef OSCF_TIMING   // dev build (tools/osc_phases.py): s_memtime stamps of the workgroup's phases, thread 0 of every workgroup
__device__ unsigned long long g_oscf_stamps[8 * 4096];
// (s_memtime counters are per XCD and not synchronised: the XCC id rides in the top 4 bits of every stamp)
#define OSCF_STAMP(i) do { if (threadIdx.x == 0) g_oscf_stamps[8 * ((blockIdx.y * gridDim.x + blockIdx.x) & 4095) + (i)] = (__builtin_amdgcn_s_memtime() & 0x0fffffffffffffffull) | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 60); } while (0)
// ... and the workgroup's entry / exit on the 100 MHz real-time counter, which all XCDs share (s_memtime offsets differ per SE)
__device__ unsigned long long g_oscf_rt[2 * 4096];
#define OSCF_RT(i) do { if (threadIdx.x == 0) g_oscf_rt[2 * ((blockIdx.y * gridDim.x + blockIdx.x) & 4095) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int golf_debug_oscf_rt(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_oscf_rt), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
}
extern "C" int golf_debug_oscf_stamps(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_oscf_stamps), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
}
#else
#define OSCF_STAMP(i) do { } while (0)
#define OSCF_RT(i) do { } while (0)
#endif
// ---- round 5: the fused forward rebuilt (osc_fused2) ----------------------------------------------------------------
// Same algorithm and the same exact phases as the round 2 - 4 kernel (osc_fused_kernel, gone), reorganised around what stamps,
// counters and timing proxies showed it to be bound by:
//   * table rows arrive as 16-byte loads (6 - 8 per thread instead of 24 - 32 dword loads with a 64-bit address each), all in
//     flight together, and are written as 16-byte pair stores; the phase samples, the tile's base phase and the fused addend are
//     fetched first, and the phases are converted and scanned while the rows are in flight;
//   * ONE barrier publishes the row pairs, the wave totals, the tile's base phase, the in-wave prefix at the halo and the
//     Toeplitz fragments of the taps; one more stands between the render and the matrix phase;
//   * the render issues the 4 gathers of a coarse sample (as two 8-byte reads each: ds_read_b64 is served 32 lanes per LDS cycle
//     on 64 banks, ds_read2_b64 16 lanes on 32) before it consumes the first, and the next coarse sample's before it stores;
//   * interior tiles (every sample exists) run a body without the existence masks; 64-bit phase steps are single
//     v_lshl_add_u64; the control-frame position is one multiply + floor;
//   * three of the four branches' Toeplitz fragments reach the waves through LDS (9 KB staged once per workgroup, 9
//     ds_read_b128 per wave): as global loads all four were 96 KB per workgroup through the vector cache -- three times the
//     workgroup's own HBM traffic; the fourth stays a global load because 12 KB would not fit beside the tile in half a CU's LDS;
//   * the signal tile is padded 2 words per 16 (18 li + lk is a permutation of the 32 banks a ds_read_b32 is served on; the
//     old 20 li + lk was 2-way throughout; 1 word per 16 was tried to make room for all four branches' fragments and made the
//     render's stores FIVE-way: +76 M conflict cycles at B = 2048);
// Measured and not adopted: two half-tile passes per table staging with 3 workgroups per CU; persistent
// workgroups with the next unit's loads in flight; a wave-autonomous variant (one wave = one 256-output stretch end to end,
// no barrier after the staging); 1536-output tiles.  What they have in common: a workgroup's lifetime is a latency chain of
// ~16 k cycles whatever the tile, and the outputs in flight per CU are bounded by LDS (17 B of signal tile per output + 33 - 49 KB
// of table rows per workgroup), so every variant lands on the same ~0.25 outputs per cycle and CU.
// ---- round 6: the phase scan inside the kernel --------------------------------------------------------
// osc_tile_totals_kernel read the whole phase a second time (6.4 MB at B = 32, 13 % of the oscillator's time at B = 4096) and was
// a launch on the critical path of a lone batch.  Now every wave of a workgroup, as soon as its phase samples are converted,
// PUBLISHES the phase advance of the tile's segments it owns -- one 16-byte entry per (utterance, tile, wave) -- and wave 0 sums
// the entries of the tiles in front of its own (same utterance) before the workgroup's first barrier: a single-pass scan with
// decoupled look-back.  Integer adds: the base phase is the bit pattern the totals kernel gave.
//   * An entry is two 8-byte words, each (32 bits of the 64-bit advance | 32-bit tag << 32), stored and polled at agent scope
//     (the XCDs' L2s are not coherent with each other for ordinary accesses).  An 8-byte access is single-copy atomic, and the
//     tags are in the words they guard: no flag word, no fence, no second round trip.
//   * Tags are a hash of (launch counter of the utterance, entry index).  The counter gen[b] lives in the workspace and is
//     incremented by the workgroup of the utterance's LAST tile after its look-back has seen every other tile's entries -- i.e.
//     after every wave of every workgroup of that utterance has read gen[b] for this launch (a wave reads it before it can
//     publish).  Entries of earlier launches therefore never validate, whatever the workspace held before (a fresh, zeroed or
//     NaN-poisoned buffer is as good as a used one: 2^-64 per entry for random contents), with no clearing pass and nothing
//     per launch from the host -- which a captured hipGraph could not supply.
//   * Progress: a workgroup only waits for lower workgroup ids of its own utterance (<= ntile - 1 of them), each of which
//     publishes in its first microsecond without waiting for anything.  The lowest unfinished id of a launch is always next in
//     its XCD's dispatch order, and at most ntile - 1 workgroups per utterance can be waiting at any time, so other kernels on
//     the chip (batches in flight) delay a launch but cannot wedge it.  The poll is bounded all the same; running out puts NaN
//     into the tile's output.
struct OscLook {
    unsigned* gen;           // [B] launch counters
    unsigned long long* ent; // [B][ntile][8][2]
};
__device__ __forceinline__ unsigned osc_look_tag(unsigned gen, unsigned idx, int half) {
    const unsigned a = (gen + 0x9E3779B1u) * (half ? 0xC2B2AE3Du : 0x85EBCA77u);
    const unsigned h = a ^ (idx * (half ? 0x27D4EB2Fu : 0x165667B1u)) ^ (half ? 0x5BD1E995u : 0xA54FF53Au);
    return h ^ (h >> 15);
}
// the decimation taps as the MFMA B fragments of the fused kernels (lane `lane` of K-step kk of branch ph holds the tap at
// d = dmin + (4 kk + lane/16 - lane%16), 0 outside the filter), four K-steps of a lane per 16-byte word:
// [(ph * KS/4 + kk/4) * 64 + lane][kk % 4].  Bf4: forward; Bfr: backward (transposed FIR: the tap at d = dmax - q).
// A function of the taps alone: golf_glottal_osc_tap_fragments_f32 builds them ONCE per tap set (round 5 re-laid them every step).
__global__ __launch_bounds__(256) void osc_tap_frags_kernel(const float* __restrict__ taps, int K, int dmin, int dmax, int KS,
                                                            float* __restrict__ Bf4, float* __restrict__ Bfr) {
    const int half = (K - 1) / 2;
    for (int e = threadIdx.x + blockIdx.x * 256; e < 4 * KS * 64; e += 256 * gridDim.x) {
        const int lane = e & 63, kk = (e >> 6) % KS, ph = (e >> 6) / KS;
        const int q = 4 * kk + (lane >> 4) - (lane & 15);
        const int o = (((ph * (KS >> 2) + (kk >> 2)) * 64 + lane) << 2) + (kk & 3);
        if (Bf4) {
            const int d = dmin + q, k = half + 4 * d + ph;
            Bf4[o] = (q >= 0 && d >= dmin && k >= 0 && k < K) ? taps[k] : 0.f;
        }
        if (Bfr) {
            const int d = dmax - q, k = half + 4 * d + ph;
            Bfr[o] = (q >= 0 && d >= dmin && d <= dmax && k >= 0 && k < K) ? taps[k] : 0.f;
        }
    }
}

#ifndef OSCF2_XPAD
#define OSCF2_XPAD 2          // pad words per 16 of the signal tile: 2 = reads conflict-free (18 li + lk is a permutation of the 32 banks),
#endif                        // render stores 2-way (free); 1 = one 2-way pair per read and FIVE-way stores (measured: +76 M conflict cycles at B = 2048)
#ifndef OSCF2_FRAG_LDS
#define OSCF2_FRAG_LDS 3      // polyphase branches whose Toeplitz fragments reach the waves through LDS (the rest by 16-byte global loads):
#endif                        // 3 is what fits beside a 2-padded tile within half a CU's LDS
#ifndef OSCF2_STAGES
#define OSCF2_STAGES 2        // gathers of the next coarse sample in flight while the current one is blended and stored (1: not)
#endif
template <int KS, int TO>
struct Oscf2Geom {
    static constexpr int HALO = 4 * KS, SPAN = TO + HALO;
    static constexpr int XS = (SPAN + OSCF2_XPAD * ((SPAN + 15) >> 4) + 3) & ~3;   // padded polyphase row: i + XPAD * (i >> 4)
    static constexpr int FRAG = OSCF2_FRAG_LDS * KS * 64;                    // floats: Toeplitz fragments behind the row pairs
    static constexpr int SCRATCH = 160;                                      // bytes behind those: wave totals, bases
};
__device__ __forceinline__ int oscf2_xaddr(int i) { return i + OSCF2_XPAD * (i >> 4); }

// blended control-frame rows r_first .. r_first + NR - 1 of one utterance -> (value, row difference) pairs in LDS.
// Round 6: the table-select weights of the rows are fetched at entry (oscf2_row_weights: a dependent round trip in front of the
// row loads), the rows themselves after the phase scan -- which is where the scan sat anyway, behind the wait for the rows: the
// sum is the same, but the wave's share of the tile's phase advance is published ~1.3 us after the workgroup started instead of
// ~2.5, and the row registers do not live through the scan.
template <int NR>
__device__ __forceinline__ void oscf2_row_weights(const float* __restrict__ wrow, int Fw, int r_first, int nrw, float (&wk)[OSCF_MAXROWS]) {
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        int k = r_first + (e < nrw ? e : nrw - 1);
        if (k > Fw - 1) k = Fw - 1;              // replicate-padded frames (models/synth.py:141-146)
        wk[e] = wrow[k];
    }
}
template <int NR>
struct OscRows {                                 // a thread's first 16-byte row loads, in flight through the phase scan
    int i0[NR];                                  // (uniform) first of the two table rows a control frame blends
    float pw[NR];
    f32x4_t va[NR], vb[NR];
};
template <int NR, int NTH>
__device__ __forceinline__ void oscf2_rows_issue(const float (&wk)[OSCF_MAXROWS], const float* __restrict__ table, int n_tab, int L,
                                                 int tid, OscRows<NR>& R) {
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const float idx = wk[e] * (float)(n_tab - 1);
        int i0 = __builtin_amdgcn_readfirstlane((int)idx);
        i0 = i0 < 0 ? 0 : (i0 > n_tab - 2 ? n_tab - 2 : i0);
        R.pw[e] = idx - (float)i0;
        R.i0[e] = i0;
    }
    const int c4 = 4 * tid < L ? tid : 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        const float* t0 = table + (size_t)R.i0[e] * L;
        R.va[e] = *reinterpret_cast<const f32x4_t*>(t0 + 4 * c4);
        R.vb[e] = *reinterpret_cast<const f32x4_t*>(t0 + L + 4 * c4);
    }
}
template <int NR, int NTH>
__device__ __forceinline__ void oscf2_rows_finish(const float* __restrict__ table, int L, float2* pairs, int LRP, int tid, OscRows<NR>& R) {
    for (int c4 = tid; 4 * c4 < L; c4 += NTH) {
        if (c4 != tid) {
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const float* t0 = table + (size_t)__builtin_amdgcn_readfirstlane(R.i0[e]) * L;
                R.va[e] = *reinterpret_cast<const f32x4_t*>(t0 + 4 * c4);
                R.vb[e] = *reinterpret_cast<const f32x4_t*>(t0 + L + 4 * c4);
            }
        }
#pragma unroll
        for (int e = 0; e < NR; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) R.va[e][q] = fmaf(R.vb[e][q], R.pw[e], R.va[e][q] * (1.0f - R.pw[e]));
#pragma unroll
        for (int rr = 0; rr + 1 < NR; ++rr) {
            f32x4_t* dst = reinterpret_cast<f32x4_t*>(pairs + (size_t)rr * LRP + 4 * c4);
            const f32x4_t lo = {R.va[rr][0], R.va[rr + 1][0] - R.va[rr][0], R.va[rr][1], R.va[rr + 1][1] - R.va[rr][1]};
            const f32x4_t hi = {R.va[rr][2], R.va[rr + 1][2] - R.va[rr][2], R.va[rr][3], R.va[rr + 1][3] - R.va[rr][3]};
            dst[0] = lo;
            dst[1] = hi;
            if (c4 == 0) pairs[(size_t)rr * LRP + L] = make_float2(lo[0], lo[1]);   // column L = column 0
        }
    }
}

template <int EE, int KS, int TO, int NTH, bool EDGE, int NRT>   // NRT: control-frame rows staged (= nrows, 2 .. OSCF_MAXROWS)
__device__ __forceinline__ void oscf2_body(
    const float* __restrict__ phase, int64_t phase_stride, OscLook look, u64* __restrict__ Ttot,
    const float* __restrict__ wsel, int Fw, const float* __restrict__ table, int n_tab, int L, int lshift, int Tp,
    int hop_t, const float* __restrict__ Bf4, float* __restrict__ out, int64_t out_stride, int Tout,
    int dmin, int nrows, const float* __restrict__ addend, int64_t addend_stride, int Tadd, float* smem,
    int tile, int b, int ntile) {
    typedef Oscf2Geom<KS, TO> G;
    constexpr int SPAN = G::SPAN, XS = G::XS;
    constexpr int CPT = (SPAN + NTH - 1) / NTH, NW = NTH / 64, NT = TO / 256;
    static_assert(NTH % 64 == 0 && TO % 256 == 0 && NT <= NW, "whole waves; one 256-output wave tile per wave at most");
    const int LRP = L + 2;                       // entries per pair row: L columns + the wrap-around copy of column 0, even
    float* X = smem;
    float2* pairs = reinterpret_cast<float2*>(smem + 4 * XS);
    float* frag = reinterpret_cast<float*>(pairs + (size_t)(nrows - 1) * LRP);
    u64* scr = reinterpret_cast<u64*>(frag + G::FRAG);
    u64* wtot = scr;                             // [NW]
    u64* base_p = scr + NW;                      // base phase of the tile start
    u64* halo_p = base_p + 1;                    // prefix at index -dmin
    u64* wown = halo_p + 1;                      // [NW] the waves' shares of the tile's own phase advance (-> Ttot)
    unsigned* lost_p = reinterpret_cast<unsigned*>(wown + NW);   // the look-back ran out
    static_assert((NW + 2 + NW) * 8 + 4 <= G::SCRATCH, "scratch behind the fragments");
    const int tid = threadIdx.x, lane = tid & 63;
    // this launch's counter of the utterance (scalar load; see OscLook)
    // (look.gen == nullptr: the two-launch form -- osc_tile_totals_kernel ran first and Ttot holds every tile's advance)
    const bool lookback = look.gen != nullptr;   // (uniform)
    const unsigned gnow = lookback ? (unsigned)__builtin_amdgcn_readfirstlane((int)look.gen[b]) : 0u;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int o0 = tile * TO;
    const int j_lo = o0 + dmin;
    OSCF_STAMP(0);
    OSCF_RT(0);
    // ---- 1. loads: coarse phase samples (clamped at the ends of the utterance), the stretch totals in front of the tile, the
    //         fused addend of the wave's outputs, the tap fragments, the table rows
    const BufRow prow(phase + (size_t)b * phase_stride, Tp);
    const int i0t = tid * CPT;
    float pv[CPT + 1];
#pragma unroll
    for (int r = 0; r <= CPT; ++r) {
        const int j = j_lo + i0t + r;
        pv[r] = prow.ld(EDGE ? (j < 0 ? 0 : (j > Tp - 1 ? Tp - 1 : j)) : j);
    }
    // D rows 4 lk + r, column li of wave tile wv  ->  output o0 + 256 wv + 16 (4 lk + r) + li
    const BufRow arow(addend ? addend + (size_t)b * addend_stride : nullptr, addend ? Tadd : 0);
    const int ob = o0 + 256 * wv + 64 * lk + li;
    constexpr int NFQ4 = OSCF2_FRAG_LDS * (KS / 4) * 64;      // 16-byte words of the branches staged in LDS
    const int m_first = max(j_lo, 0) * 4;        // first fine sample that exists in this tile
    const int r_first = m_first / hop_t;         // its control frame; rows r_first .. r_first + nrows - 1 are staged
    float wk[OSCF_MAXROWS];
    {
        const int m_last = min(j_lo + SPAN - 1, Tp - 1) * 4 + 3;
        const int nrw = min(nrows, m_last / hop_t - r_first + 2);   // rows this tile's samples really touch
        // rows beyond nrw repeat row nrw - 1 (their pair rows hold zeros as differences and are never read by a sample that exists)
        oscf2_row_weights<NRT>(wsel + (size_t)b * Fw, Fw, r_first, nrw, wk);
    }
    constexpr int LBQ = 3;                       // polls in flight per lane: 3 x 64 entries = 24 tiles (2 s of audio) per round
    const int nent = tile * NW;
    u64 lw0[LBQ], lw1[LBQ];
    // `optimistic`: ordinary loads, served from this XCD's L2 -- which may hold a line from before its owner wrote it (another
    // workgroup of this XCD polled it too early).  That is harmless: the tags are in the words they guard, a stale line can only
    // fail to validate, and what fails is polled again at agent scope.  At a device-saturating batch the tiles in front finished
    // rounds ago and the cheap poll is the only one.
    auto look_poll = [&](int e0, unsigned pend, bool optimistic = false) {
#pragma unroll
        for (int u = 0; u < LBQ; ++u) {
            const unsigned idx = (unsigned)(b * ntile * NW + (((pend >> u) & 1u) ? e0 + 64 * u + lane : 0));
            const u64* e = look.ent + 2 * (size_t)idx;
            if (optimistic) {
                lw0[u] = __builtin_nontemporal_load(e);
                lw1[u] = __builtin_nontemporal_load(e + 1);
            } else {
                lw0[u] = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lw1[u] = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto look_pending = [&](int e0) {
        unsigned pend = 0;
#pragma unroll
        for (int u = 0; u < LBQ; ++u) pend |= (e0 + 64 * u + lane < nent) ? 1u << u : 0u;
        return pend;
    };
    // The look-back's first poll (wave 0) goes out HERE, behind the phase samples and the row weights and in front of everything
    // else: an agent-scope load takes 1 - 2 us and loads return in order, so whatever is issued behind it waits for it -- the rows,
    // ~1.3 us from now, by which time it is nearly back (measured the other way round, the poll issued with the rows: every
    // workgroup 10 % longer at B = 16 384).  In every round of workgroups but the first the tiles in front published long ago
    // and this poll is the only one.
    u64 lacc = 0;                                // (wave 0) advances collected so far
    unsigned lpend = look_pending(0);            // entries of the first round still missing
    auto look_eval = [&](int e0) {
#pragma unroll
        for (int u = 0; u < LBQ; ++u) {
            const unsigned idx = (unsigned)(b * ntile * NW + e0 + 64 * u + lane);
            const bool ok = (unsigned)(lw0[u] >> 32) == osc_look_tag(gnow, idx, 0) &&
                            (unsigned)(lw1[u] >> 32) == osc_look_tag(gnow, idx, 1);
            if (((lpend >> u) & 1u) && ok) {
                lacc += (lw0[u] & 0xffffffffull) | (lw1[u] << 32);
                lpend &= ~(1u << u);
            }
        }
    };
    if (wv == 0 && nent > 0 && lookback) look_poll(0, lpend, true);
    // the rows: their weights were fetched with the phase samples; the loads go out now and are consumed behind the scan
    OscRows<NRT> rows;
    oscf2_rows_issue<NRT, NTH>(wk, table, n_tab, L, tid, rows);
    OSCF_STAMP(1);
    // ---- 2. conversions and the in-wave scan
    u64 av[CPT + 1];
#pragma unroll
    for (int r = 0; r <= CPT; ++r) av[r] = osc_fix_a_pow2(pv[r], 2);
    const int jb = j_lo + i0t;
    auto seg_of = [&](int r) -> u64 {
        const u64 d = osc_fix_d_pow2(av[r], av[r + 1], 2);
        const u64 sg = ((av[r] + d) << 2) + (d << 1);             // 4 a + 6 d
        if (!EDGE) return sg;
        const int j = jb + r;
        return (j >= 0 && j < Tp - 1) ? sg : 0;                   // segments 0 .. Tp-2 advance the phase
    };
    u64 excl;
    {
        // One pass over the thread's CPT segments gives its total AND its partial sums up to the two positions the tile's own span
        // cuts a thread at (uniform: a wave starts at a multiple of CPT samples): r_lo = (-dmin) % CPT -- the tile's first output --
        // and r_hi = (-dmin + TO) % CPT -- one past its last.  (Until round 6 the prefix at -dmin re-evaluated the segments, and the
        // wave's share of the tile's own advance was a third evaluation plus a second wave scan: ~110 VALU instructions per wave.)
        const int r_lo = (-dmin) % CPT, r_hi = (-dmin + TO) % CPT;   // (uniform)
        u64 tsum = 0, p_lo = 0, p_hi = 0;
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            tsum += seg_of(r);
            p_lo = r + 1 == r_lo ? tsum : p_lo;
            p_hi = r + 1 == r_hi ? tsum : p_hi;
        }
        const u64 incl = wave_incl_scan(tsum, lane);
        excl = incl - tsum;
        if (lane == 63) wtot[wv] = incl;
        // in-wave prefix at -dmin (the tile's first output, whose phase the look-back gives): a thread of wave 0
        const int th = (-dmin) / CPT;                             // (uniform)
        const u64 at_lo = excl + p_lo, at_hi = excl + p_hi;      // this wave's prefix at a cut that falls into this thread
        if (tid == th) *halo_p = at_lo;
        // ---- 2b. publish this wave's share of the tile's OWN advance (segments o0 .. o0 + TO - 1 = indices -dmin .. -dmin + TO - 1):
        //          (wave prefix at min(hi cut, wave end)) - (wave prefix at max(lo cut, wave start)), from the scan that exists
        if (lookback) {
            auto lane_u64 = [](u64 v, int l) -> u64 {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
                return ((u64)hi << 32) | lo;
            };
            const int wstart = wv * 64 * CPT, wend = wstart + 64 * CPT;   // sample indices of this wave (uniform)
            const int x_lo = -dmin, x_hi = -dmin + TO;
            const u64 wtotal = lane_u64(incl, 63);
            const u64 up = x_hi >= wend ? wtotal : (x_hi <= wstart ? 0 : lane_u64(at_hi, (x_hi - wstart) / CPT));
            const u64 dn = x_lo <= wstart ? 0 : (x_lo >= wend ? wtotal : lane_u64(at_lo, (x_lo - wstart) / CPT));
            const u64 own = x_hi <= wstart || x_lo >= wend ? 0 : up - dn;
            if (lane == 63) {
                wown[wv] = own;
                const unsigned idx = (unsigned)((b * ntile + tile) * NW + wv);
                u64* e = look.ent + 2 * (size_t)idx;
                const u64 w0 = (own & 0xffffffffull) | ((u64)osc_look_tag(gnow, idx, 0) << 32);
                const u64 w1 = (own >> 32) | ((u64)osc_look_tag(gnow, idx, 1) << 32);
                __hip_atomic_store(e, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(e + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // A workgroup whose neighbours in front started with it (a lone batch: 512 of its 768 workgroups enter within a microsecond)
    // finds nothing in its first poll.  By now those neighbours have published as well: the second poll goes out at once and is
    // back with the rows, instead of starting behind them.
    if (wv == 0 && nent > 0 && lookback) {
        look_eval(0);
        if (__builtin_amdgcn_ballot_w64(lpend != 0u) != 0ull) look_poll(0, lpend);
    }
    // ---- 2c. the rows and the LDS-staged fragments
    f32x4_t fq[NFQ4 > 0 ? (NFQ4 + NTH - 1) / NTH : 1];
#pragma unroll
    for (int q = 0; q < (NFQ4 + NTH - 1) / NTH; ++q) {
        const int e = tid + q * NTH;
        fq[q] = *reinterpret_cast<const f32x4_t*>(Bf4 + 4 * (e < NFQ4 ? e : 0));
    }
    oscf2_rows_finish<NRT, NTH>(table, L, pairs, LRP, tid, rows);
#pragma unroll
    for (int q = 0; q < (NFQ4 + NTH - 1) / NTH; ++q) {
        const int e = tid + q * NTH;
        if (e < NFQ4) reinterpret_cast<f32x4_t*>(frag)[e] = fq[q];
    }
    if (wv == 0) {   // ---- 2d. look back: the base phase = the advance of every tile in front of this one
        bool lost = false;
        if (!lookback)
            for (int i = lane; i < tile; i += 64) lacc += Ttot[(size_t)b * ntile + i];
        for (int e0 = 0; lookback && e0 < nent; e0 += 64 * LBQ) {
            if (e0 > 0) lpend = look_pending(e0);
            for (unsigned it = 0u;; ++it) {
                // (round 0: whatever is still pending has a poll in flight since the publication above)
                if (e0 > 0 || it > 0) look_poll(e0, lpend);
                look_eval(e0);
                if (__builtin_amdgcn_ballot_w64(lpend != 0u) == 0ull) break;
                __builtin_amdgcn_s_sleep(2);
                if (it > (1u << 18)) { lost = true; break; }       // (never observed; bounded like every device-side wait here)
            }
        }
        lacc = wave_incl_scan(lacc, lane);
        if (lane == 63) { *base_p = lacc; *lost_p = lost ? 1u : 0u; }
    }
    __syncthreads();                             // row pairs, fragments, wave totals, bases
    if (tid == 64 && lookback) {                 // the tile's own total, for the backward (golf_glottal_osc_bwd_wsel_f32, GOLF_OSC_WS_KEPT)
        u64 t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += wown[w];
        Ttot[(size_t)b * ntile + tile] = t;
    }
    // every wave of every workgroup of this utterance has read gen[b] by now (wave 0 saw all their entries): the next launch's value
    if (tile == ntile - 1 && tid == 0 && lookback) look.gen[b] = gnow + 1u;
    const bool lost = *lost_p != 0u;             // (uniform) the look-back ran out: this tile's output is NaN, not a wrong phase
    OSCF_STAMP(2);
    // (the remaining branches' fragments: 16-byte global loads of the waves that multiply, in flight through the render.  Round 6:
    //  issued here, not at entry -- the row registers now live through the scan, and these twelve were what spilled)
    f32x4_t fg[OSCF2_FRAG_LDS < 4 ? (4 - OSCF2_FRAG_LDS) * (KS / 4) : 1];
    if (OSCF2_FRAG_LDS < 4 && wv < NT) {
#pragma unroll
        for (int q = 0; q < (4 - OSCF2_FRAG_LDS) * (KS / 4); ++q)
            fg[q] = *reinterpret_cast<const f32x4_t*>(Bf4 + (((OSCF2_FRAG_LDS * (KS / 4) + q) * 64 + lane) << 2));
    }
    u64 ph = *base_p - *halo_p + excl;
#pragma unroll
    for (int w = 0; w < NW; ++w) ph += w < wv ? wtot[w] : 0;
    // ---- 3. render the 4 fine samples of every owned coarse sample into the polyphase tile
    const float inv_hop_t = 1.0f / (float)hop_t;
    const int fw = 32 - lshift < 24 ? 32 - lshift : 24, fo = 32 - lshift - fw;
    const float fscale = __uint_as_float((unsigned)(127 - fw) << 23);   // 2^-fw
    const int row_bytes = LRP * 8;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const f32x2_t lds_cf2;   // (LDS byte offsets as integers: see the gathers)
    const unsigned pbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)reinterpret_cast<const char*>(pairs);
    if (i0t < SPAN) {
        // gathers of one coarse sample: (row pair address, column) -> (value, row difference) of columns c0 and c0 + 1
        f32x2_t e0[2][4], e1[2][4];
        unsigned hik[2][4];
        float rf0[2];
        u64 phn = ph;
        const float x0 = (float)(4 * jb - r_first * hop_t) * inv_hop_t;   // control frames from r_first at the thread's first sample
        auto issue = [&](int r) {
            const int s = r & 1;
            // (the second difference and the segment total are recomputed from an opaque copy of the increment: kept from the
            //  scan they are 6 registers per coarse sample, which is what decides the waves per SIMD here)
            u64 a = av[r];
            asm volatile("" : "+v"(a));
            const u64 d = osc_fix_d_pow2(a, av[r + 1], 2), t = a + d;
            u64 sg = (t << 2) + (d << 1);
            if (EDGE) {
                const int j = jb + r;
                sg = (j >= 0 && j < Tp - 1) ? sg : 0;
            }
            // the three fine phases inside the coarse sample -- ph + a, ph + 2 a + d, ph + 3 a + 3 d -- from the HIGH words alone: the
            // carries out of the low words are at most 3 units of 2^-32 cycle (1.4e-6 table columns; the lookup is continuous in
            // the phase), the running phase itself stays exact in 64 bits.  Four 32-bit adds instead of eight 64-bit operations.
            const unsigned phi = (unsigned)(phn >> 32), ahi = (unsigned)(a >> 32), dhi = (unsigned)(d >> 32), thi = ahi + dhi;
            (void)t;
            hik[s][0] = phi + ahi;
            hik[s][1] = hik[s][0] + thi;
            hik[s][2] = hik[s][1] + thi + dhi;
            phn += sg;
            hik[s][3] = (unsigned)(phn >> 32);
            // control-frame position of the coarse sample's first fine sample: frames from r_first, integer + fraction
            const float x = fmaf((float)r, 4.0f * inv_hop_t, x0);
            rf0[s] = __builtin_amdgcn_fractf(x);
            const int ri = (int)x;                   // (x >= 0 wherever the sample exists)
            const unsigned rowaddr = pbase + (unsigned)(EDGE ? (ri < 0 ? 0 : ri) : ri) * (unsigned)row_bytes;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned c0 = hik[s][k] >> (32 - lshift);
                const unsigned q = rowaddr + c0 * 8;
                e0[s][k] = *(lds_cf2*)(uintptr_t)q;
                unsigned q1 = q;
                asm volatile("" : "+v"(q1));       // an opaque copy of the address: two ds_read_b64, not one ds_read2_b64
                e1[s][k] = *(lds_cf2*)(uintptr_t)(q1 + 8);
            }
        };
        if (OSCF2_STAGES == 2) issue(0);
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int s = r & 1;
            const int i = i0t + r;
            if (OSCF2_STAGES == 2) { if (r + 1 < CPT) issue(r + 1); } else issue(r);
            const float p0 = pv[r], p1 = pv[r + 1];
            float sck[4] = {1.f, 1.f, 1.f, 1.f};
            if (EE) {
                const float q0 = p0 * 0.25f, dq = (p1 - p0) * 0.0625f;   // fine increment q0 + k dq (cycles per fine sample)
                const float s0 = __builtin_amdgcn_rsqf(q0);              // raw v_rsq_f32: q0 is a normal number
                const float ds = -0.5f * s0 * s0 * s0 * dq;              // d/dk rsqrt(q0 + k dq) at k = 0
                const bool lin = fabsf(p1 - p0) <= 0.002f * p0;          // second-order term below 1e-6: speech f0 always is
#pragma unroll
                for (int k = 0; k < 4; ++k) sck[k] = fmaf((float)k, ds, s0);
                if (__builtin_amdgcn_ballot_w64(!lin) != 0) {
                    // f0 jumps (voicing boundaries): exact, wave-uniform and behind an opaque statement so that the four
                    // quarter-rate v_rsq_f32 are not issued in the common case
                    asm volatile("; exact equal-energy factors" ::: "memory");
#pragma unroll
                    for (int k = 0; k < 4; ++k) sck[k] = lin ? sck[k] : __builtin_amdgcn_rsqf(fmaf((float)k, dq, q0));
                }
            }
            if (EDGE) {
                const int j = jb + r;
                const bool v0 = j >= 0 && j <= Tp - 1, vk = j >= 0 && j < Tp - 1;   // the last coarse sample has only k = 0
                sck[0] = v0 ? sck[0] : 0.f;
#pragma unroll
                for (int k = 1; k < 4; ++k) sck[k] = vk ? sck[k] : 0.f;
            }
            float* xp = X + oscf2_xaddr(i);
            if (i < SPAN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float rf = fmaf((float)k, inv_hop_t, rf0[s]);
                    const float cf = (float)__builtin_amdgcn_ubfe(hik[s][k], (unsigned)fo, (unsigned)fw) * fscale;
                    const float t0 = fmaf(rf, e0[s][k][1], e0[s][k][0]), t1 = fmaf(rf, e1[s][k][1], e1[s][k][0]);
                    float v = fmaf(cf, t1 - t0, t0);
                    if (EE || EDGE) v *= sck[k];
                    if (EDGE) v = sck[k] == 0.f ? 0.f : v;   // (a row that is not staged may hold anything: 0 x NaN)
                    xp[k * XS] = v;
                }
            }
        }
    }
    OSCF_STAMP(3);
    __syncthreads();
    OSCF_STAMP(4);
    // ---- 4. polyphase FIR on the matrix pipe: wave wv owns outputs o0 + 256 wv .. + 255 as a 16 x 16 tile
    //         D[m][n] (output 256 wv + 16 m + n) = sum_ph sum_k' X_ph[256 wv + 16 m + k'] * B_ph[k'][n]
    if (wv < NT) {
        // the fused addend of the wave's outputs (round 6: fetched here, behind the render, not at entry -- four registers that
        // lived through every phase; the matrix phase below hides the round trip)
        float ad[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ad[r] = lost ? __builtin_nanf("") : arow.ld(ob + 16 * r);
        float bfrag[4][KS];
#pragma unroll
        for (int phs = 0; phs < 4; ++phs)
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const f32x4_t v = phs < OSCF2_FRAG_LDS
                    ? *reinterpret_cast<const f32x4_t*>(frag + (((phs * (KS / 4) + q) * 64 + lane) << 2))
                    : fg[(phs - OSCF2_FRAG_LDS) * (KS / 4) + q];
#pragma unroll
                for (int j = 0; j < 4; ++j) bfrag[phs][4 * q + j] = v[j];
            }
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // A[m = li][k' = 4 kk + lk]: window element e = 256 wv + 16 li + 4 kk + lk -> address e + XPAD (e >> 4)
        constexpr int XP = OSCF2_XPAD;
        const float* ap = X + (256 + 16 * XP) * wv + (16 + XP) * li + lk;
#pragma unroll
        for (int phs = 0; phs < 4; ++phs) {
            float a[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[kk] = ap[phs * XS + 4 * kk + XP * (kk >> 2)];
#pragma unroll
            for (int kk = 0; kk < KS; kk += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], bfrag[phs][kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk + 1], bfrag[phs][kk + 1], acc1, 0, 0, 0);
            }
        }
        OSCF_STAMP(5);
        const BufRow orow(out + (size_t)b * out_stride, Tout);
#pragma unroll
        for (int r = 0; r < 4; ++r) orow.st(ob + 16 * r, acc0[r] + acc1[r] + ad[r]);
    }
    OSCF_STAMP(7);
    OSCF_RT(1);
}

// one (utterance, tile) unit: the body of osc_fused2_kernel, and of the oscillator workgroups of the source + transition-map launch
template <int EE, int KS, int TO, int NTH>
__device__ __forceinline__ void osc_fused2_tile(
    const float* __restrict__ phase, int64_t phase_stride, OscLook look, u64* __restrict__ Ttot,
    const float* __restrict__ wsel, int Fw, const float* __restrict__ table, int n_tab, int L, int lshift, int Tp,
    int hop_t, const float* __restrict__ Bf4, float* __restrict__ out, int64_t out_stride, int Tout,
    int dmin, int nrows, const float* __restrict__ addend, int64_t addend_stride, int Tadd, float* smem, int tile, int b,
    int ntile) {
    typedef Oscf2Geom<KS, TO> G;
    const int j_lo = tile * TO + dmin;
    // every coarse sample j_lo .. j_lo + SPAN (the last one as a segment's right end) exists and is not the last
    const bool edge = j_lo < 0 || j_lo + G::SPAN > Tp - 1;
#define OSCF2_BODY(EDGEV, NRV)                                                                                                  \
    oscf2_body<EE, KS, TO, NTH, EDGEV, NRV>(phase, phase_stride, look, Ttot, wsel, Fw, table, n_tab, L, lshift, Tp, hop_t, Bf4, out, \
                                            out_stride, Tout, dmin, nrows, addend, addend_stride, Tadd, smem, tile, b, ntile)
    // (the row count is a property of the launch, the edge of the tile: one of these six bodies runs per workgroup)
    if (nrows <= 2)      { if (edge) OSCF2_BODY(true, 2); else OSCF2_BODY(false, 2); }
    else if (nrows == 3) { if (edge) OSCF2_BODY(true, 3); else OSCF2_BODY(false, 3); }
    else                 { if (edge) OSCF2_BODY(true, 4); else OSCF2_BODY(false, 4); }
#undef OSCF2_BODY
}

template <int EE, int KS, int TO, int NTH>
__global__ __launch_bounds__(NTH, 4) void osc_fused2_kernel(
    const float* __restrict__ phase, int64_t phase_stride, OscLook look, u64* __restrict__ Ttot,
    const float* __restrict__ wsel, int Fw, const float* __restrict__ table, int n_tab, int L, int lshift, int Tp,
    int hop_t, const float* __restrict__ Bf4, float* __restrict__ out, int64_t out_stride, int Tout,
    int dmin, int nrows, const float* __restrict__ addend, int64_t addend_stride, int Tadd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    light_wave_priority();
    // grid x = utterance, y = tile: workgroup ids run TILE-major, so that the tiles a workgroup looks back to were dispatched a
    // whole batch of workgroups earlier -- at a device-saturating batch, many rounds earlier: their entries are there at the first
    // poll.  (Utterance-major, all tiles of an utterance enter within a fraction of a microsecond of each other whatever the
    // batch, and every workgroup pays the second poll: measured 7.7 - 8.0 ms against 7.05 for round 5's two launches at B = 16 384.)
    osc_fused2_tile<EE, KS, TO, NTH>(phase, phase_stride, look, Ttot, wsel, Fw, table, n_tab, L, lshift, Tp, hop_t, Bf4, out,
                                     out_stride, Tout, dmin, nrows, addend, addend_stride, Tadd, smem, (int)blockIdx.y,
                                     (int)blockIdx.x, (int)gridDim.y);
}

// ---- fused backward w.r.t. table_select_weight (round 3): the forward's structure run the other way ----------------
// One workgroup owns OSCB_TO coarse samples (no halo on the sample side: every fine sample is counted once) of one
// utterance:
//   1. the gradient tile Y[v] = g_out[o0 - dmax + v] (+ the FIR's reach on both sides) -> LDS, the table DIFFERENCE rows
//      D_k = T[i0_k + 1] - T[i0_k] of the control frames the tile touches -> LDS, the phase scan as in the forward;
//   2. the transposed polyphase FIR on the matrix pipe: g_pre[4j + ph] = sum_q' Y[j - o0 + q'] h_ph[dmax - q'] -- the same
//      16-window Toeplitz product as the forward's with reversed taps, one accumulator per branch -> LDS, laid out so that
//      the thread that owns coarse samples 4t .. 4t+3 finds its 16 values in one row of 20 words (four 16-byte reads);
//   3. every thread walks the 4 fine samples of its 4 coarse samples: d pre / d p of frame k is (1 - rf) D_k(phase), of
//      frame k + 1 it is rf D_{k+1}(phase), times the equal-energy factor; sums per staged row, block reduction ->
//      part[b][tile][row].  osc_wsel_reduce_tiles_kernel adds the tiles' rows into g_wsel.
// Replaces osc_phase_tile + osc_decimate_T4 + osc_render<1> (46 us and a 12 MB prefix + a 24 MB oversampled gradient on a
// round trip through HBM at B = 32) for the GOLF configuration; everything else keeps the three-kernel path.
// Toeplitz fragments of the taps as osc_tile_totals_kernel lays them out: four K-steps of a lane per 16-byte word
template <int KS>
__device__ __forceinline__ void osc_load_frags4(const float* __restrict__ Bf4, int lane, float (&bfrag)[4][KS]) {
#pragma unroll
    for (int phs = 0; phs < 4; ++phs)
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(Bf4 + (((phs * (KS / 4) + q) * 64 + lane) << 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfrag[phs][4 * q + j] = v[j];
        }
}

// difference rows D_k = T[i0_k + 1] - T[i0_k] of control frames r_first .. r_first + NR - 1 -> LDS rows of LR floats
template <int NR, int NTH>
__device__ __forceinline__ void oscb_stage_rows(const float* __restrict__ wrow, int Fw, const float* __restrict__ table, int n_tab,
                                                int L, int r_first, int nrw, float* rows, int LR, int tid) {
    const float* t0[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        int k = r_first + (e < nrw ? e : nrw - 1);
        if (k > Fw - 1) k = Fw - 1;              // replicate-padded frames (models/synth.py:141-146)
        const float idx = wrow[k] * (float)(n_tab - 1);
        int i0 = __builtin_amdgcn_readfirstlane((int)idx);
        i0 = i0 < 0 ? 0 : (i0 > n_tab - 2 ? n_tab - 2 : i0);
        t0[e] = table + (size_t)i0 * L;
    }
    for (int c4 = tid; 4 * c4 < L; c4 += NTH) {
        f32x4_t va[NR], vb[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            va[e] = *reinterpret_cast<const f32x4_t*>(t0[e] + 4 * c4);
            vb[e] = *reinterpret_cast<const f32x4_t*>(t0[e] + L + 4 * c4);
        }
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            const f32x4_t dv = vb[e] - va[e];
            *reinterpret_cast<f32x4_t*>(rows + (size_t)e * LR + 4 * c4) = dv;
            if (c4 == 0) rows[(size_t)e * LR + L] = dv[0];   // column L = wrap-around copy of column 0
        }
    }
}

#define OSCB_TO 2048          // the backward's own tile geometry (independent of the forward's build parameters)
#define OSCB_THREADS 512
#define OSCB_CPT (OSCB_TO / OSCB_THREADS)
template <int EE, int KS>
__global__ __launch_bounds__(OSCB_THREADS) void osc_fused_bwd_kernel(
    const float* __restrict__ phase, int64_t phase_stride, const u64* __restrict__ Ttot, int ntile,
    const float* __restrict__ wsel, int Fw, const float* __restrict__ table, int n_tab, int L, int lshift, int Tp,
    int hop_t, const float* __restrict__ Bf, const float* __restrict__ g_out, int64_t g_out_stride, int Tout, int dmax,
    int nrows, float* __restrict__ part) {
    constexpr int P = 4, NTH = OSCB_THREADS, CPT = OSCB_CPT;
    static_assert(OSCB_TO == 2048 && OSCB_THREADS == 512, "4 consecutive coarse samples per thread, 8 waves x 256 outputs");
    constexpr int spanY = OSCB_TO + 4 * KS;                     // gradient samples the windows reach
    constexpr int YS = (spanY + 2 * (spanY >> 4) + 2 + 3) & ~3;  // padded like the forward's signal tile: i + 2 * (i >> 4)
    constexpr int GR = 20;                                       // words per thread row of the fine-gradient tile: 16 + 4, so
    // that a thread reads its row as four 16-byte words and 16 consecutive lanes cover all 64 banks exactly once
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ u64 wtot[NTH / 64];
    __shared__ u64 base_sh;
    __shared__ float red[OSCF_MAXROWS][NTH / 64];
    // layout: Y [YS] | G [NTH][20] | difference rows [nrows][L + 4] (column L = column 0; rows 16-byte aligned)
    float* Y = smem;
    float* G = smem + YS;
    float* rows = G + NTH * GR;
    const int LR = L + 4;
    const int b = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int o0 = tile * OSCB_TO;
    // ---- 0. reversed Toeplitz fragments (osc_tile_totals_kernel, reversed = 1)
    float bfrag[4][KS];
    osc_load_frags4<KS>(Bf, lane, bfrag);            // 12 - 16 loads of 16 bytes (round 5; were 48 - 64 dword loads)
    // ---- 1. base phase, own phase samples, gradient tile, difference rows
    if (wv == 0) {
        u64 acc = 0;
        for (int i = lane; i < tile; i += 64) acc += Ttot[(size_t)b * ntile + i];
        acc = wave_incl_scan(acc, lane);
        if (lane == 63) base_sh = acc;
    }
    const BufRow prow(phase + (size_t)b * phase_stride, Tp);
    const int u0 = tid * CPT;
    float pv[CPT + 1];
#pragma unroll
    for (int r = 0; r <= CPT; ++r) {
        const int j = o0 + u0 + r;
        pv[r] = prow.ld(j > Tp - 1 ? Tp - 1 : j);
    }
    {
        const BufRow grow(g_out + (size_t)b * g_out_stride, Tout);
        constexpr int NY = (spanY + NTH - 1) / NTH;          // 5 loads per thread, all issued before the first LDS write (as
        float xv[NY];                                           // a loop: five serial round trips in front of everything)
#pragma unroll
        for (int q = 0; q < NY; ++q) {
            const int o = o0 - dmax + tid + q * NTH;
            xv[q] = grow.ld(o < 0 ? 0 : o);                     // (past the end: dropped by the descriptor -> 0)
        }
#pragma unroll
        for (int q = 0; q < NY; ++q) {
            const int v = tid + q * NTH, o = o0 - dmax + v;
            if (v < spanY) Y[oscf2_xaddr(v)] = o >= 0 ? xv[q] : 0.f;
        }
    }
    const int m_first = o0 * P;
    const int r_first = m_first / hop_t;
    {
        const int m_last = min(o0 + OSCB_TO - 1, Tp - 1) * P + (P - 1);
        const int nrw = min(nrows, m_last / hop_t - r_first + 2);
        // difference rows D_k = T[i0_k + 1] - T[i0_k] as 16-byte loads and stores, every row's loads in flight together (rows
        // beyond nrw repeat row nrw - 1: never read by a sample that exists) -- the forward's staging, round 5
        const float* wrow = wsel + (size_t)b * Fw;
        if (nrows <= 2)      oscb_stage_rows<2, NTH>(wrow, Fw, table, n_tab, L, r_first, nrw, rows, LR, tid);
        else if (nrows == 3) oscb_stage_rows<3, NTH>(wrow, Fw, table, n_tab, L, r_first, nrw, rows, LR, tid);
        else                 oscb_stage_rows<4, NTH>(wrow, Fw, table, n_tab, L, r_first, nrw, rows, LR, tid);
    }
    // ---- 2. phase scan: thread owns coarse samples u0 .. u0 + 3
    u64 av[CPT + 1], dv[CPT], tv[CPT];
    u64 tsum = 0;
#pragma unroll
    for (int r = 0; r <= CPT; ++r) av[r] = osc_fix_a_pow2(pv[r], 2);
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int j = o0 + u0 + r;
        dv[r] = osc_fix_d_pow2(av[r], av[r + 1], 2);
        tv[r] = j < Tp - 1 ? (av[r] << 2) + dv[r] * (u64)6 : 0;
        tsum += tv[r];
    }
    const u64 incl = wave_incl_scan(tsum, lane);
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();                                             // Y, rows, base_sh, wtot
    u64 ph = base_sh + incl - tsum;
    for (int w = 0; w < wv; ++w) ph += wtot[w];
    // ---- 3. transposed FIR: wave wv owns coarse samples 256 wv .. +255 as a 16 x 16 tile per polyphase branch
    {
        f32x4_t acc[4];
#pragma unroll
        for (int phs = 0; phs < 4; ++phs) acc[phs] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const float* ap = Y + 288 * wv + 18 * li + lk;           // A[m = li][k' = 4 kk + lk]: see the forward
        float a[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[kk] = ap[4 * kk + 2 * (kk >> 2)];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int phs = 0; phs < 4; ++phs)
                acc[phs] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], bfrag[phs][kk], acc[phs], 0, 0, 0);
        // D rows 4 lk + r, column li -> coarse sample i = 256 wv + 16 (4 lk + r) + li -> owner thread i >> 2, slot 4 (i & 3) + ph
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 256 * wv + 16 * (4 * lk + r) + li;
            float* gp = G + (size_t)(i >> 2) * GR + 4 * (i & 3);
#pragma unroll
            for (int phs = 0; phs < 4; ++phs) gp[phs] = acc[phs][r];
        }
    }
    __syncthreads();
    // ---- 4. per fine sample: (1 - rf) D_rr(phase) and rf D_rr+1(phase), weighted by the incoming gradient
    const float inv_hop_t = 1.0f / (float)hop_t;
    const int bnd1 = nrows > 2 ? (r_first + 1) * hop_t : 0x7fffffff, bnd2 = nrows > 3 ? (r_first + 2) * hop_t : 0x7fffffff;
    const int fw = 32 - lshift < 24 ? 32 - lshift : 24, fo = 32 - lshift - fw;
    const float fscale = __uint_as_float((unsigned)(127 - fw) << 23);
    float racc[OSCF_MAXROWS] = {0.f, 0.f, 0.f, 0.f};
    f32x4_t gq[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) gq[r] = *reinterpret_cast<const f32x4_t*>(G + (size_t)tid * GR + 4 * r);
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int j = o0 + u0 + r;
        const u64 ph_next = ph + tv[r];
        const float p0 = pv[r], p1 = pv[r + 1];
        // (the three fine phases inside the coarse sample from the high words alone, as in the forward: <= 3 units of 2^-32 cycle)
        const unsigned phi = (unsigned)(ph >> 32), ahi = (unsigned)(av[r] >> 32), dhi = (unsigned)(dv[r] >> 32), thi = ahi + dhi;
        const unsigned h0 = phi + ahi, h1 = h0 + thi, h2 = h1 + thi + dhi;
        const unsigned hik[P] = {h0, h1, h2, (unsigned)(ph_next >> 32)};
        const int m0 = j * P;
        const int rr = (m0 >= bnd1) + (m0 >= bnd2);
        const float* ra = rows + (size_t)rr * LR;
        float rf = (float)(m0 - (r_first + rr) * hop_t) * inv_hop_t;
        const bool v0 = j <= Tp - 1, vk = j < Tp - 1;            // the last coarse sample has only k = 0
        const float q0 = p0 * 0.25f, dq = (p1 - p0) * 0.0625f;
        float sck[P] = {1.f, 1.f, 1.f, 1.f};
        if (EE) {   // the forward's equal-energy factors: linear over the 4 fine samples unless f0 jumps (wave-uniform branch)
            const float s0 = __builtin_amdgcn_rsqf(q0), ds = -0.5f * s0 * s0 * s0 * dq;
            const bool lin = fabsf(p1 - p0) <= 0.002f * p0;
#pragma unroll
            for (int k = 0; k < P; ++k) sck[k] = fmaf((float)k, ds, s0);
            if (__builtin_amdgcn_ballot_w64(!lin) != 0) {
                asm volatile("; exact equal-energy factors" ::: "memory");
#pragma unroll
                for (int k = 0; k < P; ++k) sck[k] = lin ? sck[k] : __builtin_amdgcn_rsqf(fmaf((float)k, dq, q0));
            }
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const unsigned hi = hik[k];
            const int c0 = (int)(hi >> (32 - lshift));
            const float cf = (float)__builtin_amdgcn_ubfe(hi, (unsigned)fo, (unsigned)fw) * fscale;
            const float d00 = ra[c0], d01 = ra[c0 + 1], d10 = ra[LR + c0], d11 = ra[LR + c0 + 1];
            const float top = fmaf(cf, d01 - d00, d00), bot = fmaf(cf, d11 - d10, d10);
            const float g = gq[r][k] * sck[k];
            // fine samples past the end of the signal: a select on the PRODUCT (their rows may not be staged: 0 x garbage)
            const bool ok = k == 0 ? v0 : vk;
            const float t0_ = g * (1.0f - rf) * top, t1_ = g * rf * bot;
            a0 += ok ? t0_ : 0.f;
            a1 += ok ? t1_ : 0.f;
            rf += inv_hop_t;
        }
#pragma unroll
        for (int e = 0; e < OSCF_MAXROWS; ++e) racc[e] += (rr == e ? a0 : 0.f) + (rr + 1 == e ? a1 : 0.f);
        ph = ph_next;
    }
    // ---- 5. block reduction per staged row
#pragma unroll
    for (int e = 0; e < OSCF_MAXROWS; ++e) {
        float v = racc[e];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[e][wv] = v;
    }
    __syncthreads();
    if (tid < OSCF_MAXROWS) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NTH / 64; ++w) v += red[tid][w];
        part[((size_t)b * ntile + tile) * OSCF_MAXROWS + tid] = v;
    }
}

// partials of the fused backward -> g_wsel: staged row e of tile t is control frame min(r_first(t) + e, Fw - 1)
__global__ void osc_wsel_reduce_tiles_kernel(const float* __restrict__ part, float* __restrict__ g_wsel, int B, int Fw,
                                             int ntile, int n_tab, int hop_t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Fw) return;
    const int b = idx / Fw, k = idx - b * Fw;
    float acc = 0.f;
    // only the tiles whose staged rows can be frame k: r_first(t) in [k - 3, k] (the last frame also collects the clamped rows)
    const int fine = OSCB_TO * 4;
    int t_lo = (int)(((int64_t)(k - (OSCF_MAXROWS - 1)) * hop_t) / fine) - 1;
    int t_hi = k >= Fw - 1 ? ntile - 1 : (int)(((int64_t)(k + 1) * hop_t) / fine) + 1;
    t_lo = t_lo < 0 ? 0 : t_lo;
    t_hi = t_hi > ntile - 1 ? ntile - 1 : t_hi;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int r_first = (t * fine) / hop_t;
#pragma unroll
        for (int e = 0; e < OSCF_MAXROWS; ++e) {
            const int row = r_first + e > Fw - 1 ? Fw - 1 : r_first + e;
            if (row == k) acc += part[((size_t)b * ntile + t) * OSCF_MAXROWS + e];
        }
    }
    g_wsel[idx] = acc * (float)(n_tab - 1);
}

// ---- decimator launches (shared by the oscillator entry points and golf_decimate_fir_*) -----------------------------
static int launch_decimate(const float* fine, int N, int64_t fine_stride, const float* taps, int K, int os, float* out,
                           int64_t out_stride, int Tout, int B, const float* addend, int64_t addend_stride, int Tadd,
                           hipStream_t st) {
    const int half = (K - 1) / 2;
    const int dmin = -((half + os - 1) / os);  // floor(-half/os)
    const int dmax = half / os;
    const int nq = dmax - dmin + 1;             // taps per polyphase branch (upper bound)
    const int ngrp = (nq + 2) / 4;
    int RS4 = OSC_TILE / 4 + ngrp + 2;
    while (RS4 % 32 != 2) ++RS4;                // (ph,i&3) sub-arrays land 2 banks apart: conflict-free fill at os=4
    const int hoff = (os * 4 * RS4 + 3) & ~3;
    const size_t lds3 = sizeof(float) * ((size_t)hoff + (size_t)os * (ngrp * 4 + 8));
    if (lds3 > 64 * 1024) return fail(GOLF_EUNSUPPORTED, "decimate: %d taps x os %d exceed LDS", K, os);
    const int vec4 = (fine_stride % 4 == 0 && ((uintptr_t)fine & 15) == 0) ? 1 : 0;
    if (os == 4)
        hipLaunchKernelGGL(osc_decimate_kernel<4>, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), lds3, st, fine,
                           N, fine_stride, taps, K, os, out, out_stride, Tout, RS4, dmin, ngrp, vec4, addend,
                           addend_stride, Tadd);
    else
        hipLaunchKernelGGL(osc_decimate_kernel<0>, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), lds3, st, fine,
                           N, fine_stride, taps, K, os, out, out_stride, Tout, RS4, dmin, ngrp, vec4, addend,
                           addend_stride, Tadd);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

// g_pre (B, N) dense <- transposed decimator of g_out (B, Tout); os == 1: plain copy
static int launch_decimate_T(const float* g_out, int64_t g_out_stride, int Tout, const float* taps, int K, int os,
                             float* g_pre, int N, int B, hipStream_t st) {
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
        if (ldsT > 64 * 1024) return fail(GOLF_EUNSUPPORTED, "decimate (transposed): %d taps exceed LDS", K);
        hipLaunchKernelGGL(osc_decimate_T4_kernel, dim3((unsigned)ceil_div(Tout, OSC_TILE), B), dim3(256), ldsT, st,
                           g_out, g_out_stride, Tout, taps, K, g_pre, N, RS4, dmax, ngrp);
        GOLF_LAUNCH_CHECK();
    } else if (os > 1) {
        const int64_t n = (int64_t)B * N;
        hipLaunchKernelGGL(osc_decimate_T_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, g_out,
                           g_out_stride, Tout, taps, K, os, g_pre, N, B);
        GOLF_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemcpy2DAsync(g_pre, sizeof(float) * N, g_out, sizeof(float) * g_out_stride,
                                        sizeof(float) * N, B, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return fail((int)e, "decimate (transposed): copy failed: %s", hipGetErrorString(e));
    }
    return GOLF_OK;
}

// =============================================================================================
// Generic wavetable lookup: GlottalFlowTable.generate (models/synth.py:124-177, F.grid_sample bilinear over
// (control frame, phase)) for ARBITRARY per-frame tables (B,K,L) and a given wrapped phase (B,N) in [0,1).
// This is what every table oscillator of the reference reduces to once its tables are formed --
// WeightedGlottalFlowTable (:266-294, tables = softmax weights @ table), WrappedPhaseDownsampledIndexed... (:343-375),
// and IndexedGlottalFlowTable itself when the phase, a phase offset or a trainable table must receive gradients (the
// fused fast path above differentiates w.r.t. table_select_weight only).
//   out[b,n] = (1-rf) lerp_c(T[ra]) + rf lerp_c(T[rb]),   r0 = n / hop_t, rf = (n - r0 hop_t)/hop_t,
//   ra = min(r0, K-1), rb = min(r0+1, K-1)  (replicate-padded frames, synth.py:141-146),
//   c = phi L, c0 = floor(c), cf = c - c0, columns c0 and (c0+1) mod L  (wrap column, synth.py:148-150)
// Backward: d/d phi = g L [(1-rf)(T[ra][c1]-T[ra][c0]) + rf (T[rb][c1]-T[rb][c0])]; d/d T scattered with the four
// bilinear weights -- one workgroup per (utterance, control interval) accumulates its two rows in LDS (ds_add_f32),
// then adds them to g_tables (each row receives from at most two intervals: the result does not depend on their order).
// =============================================================================================
__global__ __launch_bounds__(256) void wt_lookup_fwd_kernel(const float* __restrict__ wrapped, int64_t w_stride,
                                                            const float* __restrict__ tables, int K, int L, int hop_t,
                                                            int N, float* __restrict__ out, int64_t out_stride, int B) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * N) return;
    const int b = (int)(idx / N), n = (int)(idx - (int64_t)b * N);
    const int r0 = n / hop_t;
    const float rf = (float)(n - r0 * hop_t) / (float)hop_t;
    const int ra = r0 < K - 1 ? r0 : K - 1, rb = r0 + 1 < K - 1 ? r0 + 1 : K - 1;
    const float c = wrapped[(size_t)b * w_stride + n] * (float)L;
    int c0 = (int)floorf(c);
    c0 = c0 < 0 ? 0 : (c0 > L - 1 ? L - 1 : c0);
    const float cf = c - (float)c0;
    const int c1 = c0 + 1 == L ? 0 : c0 + 1;
    const float* Ta = tables + ((size_t)b * K + ra) * L;
    const float* Tb = tables + ((size_t)b * K + rb) * L;
    const float a00 = Ta[c0], a01 = Ta[c1], a10 = Tb[c0], a11 = Tb[c1];
    const float top = fmaf(cf, a01 - a00, a00), bot = fmaf(cf, a11 - a10, a10);
    out[(size_t)b * out_stride + n] = fmaf(rf, bot - top, top);
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void wt_lookup_bwd_kernel(const float* __restrict__ g_out, int64_t g_stride,
                                                            const float* __restrict__ wrapped, int64_t w_stride,
                                                            const float* __restrict__ tables, int K, int L, int hop_t,
                                                            int N, float* __restrict__ g_wrapped, int64_t gw_stride,
                                                            float* __restrict__ g_tables) {
    extern __shared__ float acc[];   // [2][L] when USE_LDS
    const int r0 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ra = r0 < K - 1 ? r0 : K - 1, rb = r0 + 1 < K - 1 ? r0 + 1 : K - 1;
    if (USE_LDS && g_tables) {
        for (int e = tid; e < 2 * L; e += 256) acc[e] = 0.f;
        __syncthreads();
    }
    const float* Ta = tables + ((size_t)b * K + ra) * L;
    const float* Tb = tables + ((size_t)b * K + rb) * L;
    float* Ga = g_tables ? g_tables + ((size_t)b * K + ra) * L : nullptr;
    float* Gb = g_tables ? g_tables + ((size_t)b * K + rb) * L : nullptr;
    const int n_lo = r0 * hop_t, n_hi = n_lo + hop_t < N ? n_lo + hop_t : N;
    const float inv = 1.0f / (float)hop_t;
    for (int n = n_lo + tid; n < n_hi; n += 256) {
        const float g = g_out[(size_t)b * g_stride + n];
        const float rf = (float)(n - n_lo) * inv;
        const float c = wrapped[(size_t)b * w_stride + n] * (float)L;
        int c0 = (int)floorf(c);
        c0 = c0 < 0 ? 0 : (c0 > L - 1 ? L - 1 : c0);
        const float cf = c - (float)c0;
        const int c1 = c0 + 1 == L ? 0 : c0 + 1;
        if (g_wrapped) {
            const float dt = Ta[c1] - Ta[c0], db = Tb[c1] - Tb[c0];
            g_wrapped[(size_t)b * gw_stride + n] = g * (float)L * fmaf(rf, db - dt, dt);
        }
        if (g_tables) {
            const float wa = g * (1.0f - rf), wb = g * rf;
            if (USE_LDS) {
                atomicAdd(&acc[c0], wa * (1.0f - cf));
                atomicAdd(&acc[c1], wa * cf);
                atomicAdd(&acc[L + c0], wb * (1.0f - cf));
                atomicAdd(&acc[L + c1], wb * cf);
            } else {
                atomicAdd(&Ga[c0], wa * (1.0f - cf));
                atomicAdd(&Ga[c1], wa * cf);
                atomicAdd(&Gb[c0], wb * (1.0f - cf));
                atomicAdd(&Gb[c1], wb * cf);
            }
        }
    }
    if (USE_LDS && g_tables) {
        __syncthreads();
        for (int e = tid; e < L; e += 256) {
            atomicAdd(&Ga[e], acc[e]);
            atomicAdd(&Gb[e], acc[L + e]);
        }
    }
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


// =============================================================================================
// Harmonic oscillator bank (the source of the DDSP / NHV / WORLD / MLSA / SawSing / PULF
// baselines).  Replaces HarmonicOscillator.forward, models/synth.py:403-446, and what its subclasses feed it
// (AdditiveSynthesizer :449-468, SawToothOscillator :486-504, AdditivePulseTrain :526-547):
//     out[t] = sum_{h=1..H} [h * p(t) < 0.5] * amp(t,h) * sin(2 pi h Phi(t)),   Phi = inclusive cumsum of p = up(phase)
//     amp(t,h) = up(A)[t,h] * up(tscale)[t] * hscale[h]      (each factor optional)
// The reference materialises (B,T,H) tensors (0.95 GB each at B=32, T=48000, H=155) for the harmonic phases, their
// cumsum, the mask, the amplitudes and the sines.  Here one thread owns one output sample: the phase comes from the
// same exact fixed-point prefix as the wavetable oscillator (h * Phi wraps exactly in 64-bit integers), sin(h theta)
// follows by a rotation recurrence that is re-anchored from the exact phase every HARM_ANCHOR harmonics, and the
// frame-rate amplitude rows the block needs are staged in LDS.
// =============================================================================================
constexpr int HARM_THREADS = 256;
constexpr int HARM_ANCHOR = 32;

struct HarmSample {
    u64 Phi;     // inclusive phase, Q0.64 cycles
    float p;     // instantaneous increment up(phase)[t]  (cycles per sample)
};
__device__ __forceinline__ HarmSample harm_sample(const float* __restrict__ pb, const u64* __restrict__ cb,
                                                  const u64* toff, int t, int Tp, int P, double scale_a,
                                                  double scale_d) {
    const int j = t / P, k = t - j * P;
    const int jc = j < Tp - 1 ? j : Tp - 1;
    const int jn = jc + 1 < Tp ? jc + 1 : Tp - 1;
    const float p0 = pb[jc], p1 = pb[jn];
    const u64 a = osc_fix_a(p0, scale_a), d = osc_fix_d(p0, p1, scale_d);
    HarmSample r;
    r.Phi = cb[jc] + toff[min(jc / OSC_SCAN_TILE, 255)] + (u64)(k + 1) * a + d * ((u64)k * (u64)(k + 1) / 2);
    r.p = fmaf((float)k, (p1 - p0) / (float)P, p0);
    return r;
}
// Optional phase terms of HarmonicOscillator.forward (models/synth.py:434-440): harmonic h runs at
//   h * (Phi(t) + up(phase_offset)(t)) + initial_phase[b, h]   (cycles).
struct HarmPhase {
    const float* poff;   // (B, Fo) at hop po_hop, linearly upsampled; or null
    int Fo, po_hop;
    const float* phi0;   // (B, H) cycles; or null
};
// the offset as a Q0.64 fraction of a cycle (any real value: only its fractional part matters)
__device__ __forceinline__ u64 harm_offset_q64(const HarmPhase& hp, int b, int t) {
    if (!hp.poff) return 0;
    int f = 0;
    float w = 0.f;
    if (hp.Fo >= 2) { f = min(t / hp.po_hop, hp.Fo - 2); w = (float)(t - f * hp.po_hop) / (float)hp.po_hop; }
    const float v0 = hp.poff[(size_t)b * hp.Fo + f], v1 = hp.poff[(size_t)b * hp.Fo + (hp.Fo >= 2 ? f + 1 : f)];
    const double x = (double)fmaf(w, v1 - v0, v0);
    return (u64)((x - floor(x)) * 18446744073709551616.0);
}
// sin and cos of 2*pi*(x / 2^64)
__device__ __forceinline__ void harm_sincos(u64 x, float& s, float& c) {
    const float rev = (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f);  // top 24 bits: exact in fp32, [0,1)
    sincospif(2.0f * rev, &s, &c);
}

// Forward: one block per 256 consecutive output samples.
// DERIV: d out / d Phi(t) = 2 pi * sum_h [..] amp(t,h) * h * cos(2 pi h Phi(t)) instead of the signal itself (the
// Nyquist mask is piecewise constant in the phase); feeds the gradient w.r.t. the phase input.
template <bool DERIV>
__global__ __launch_bounds__(HARM_THREADS) void harm_kernel(
    const float* __restrict__ phase, int64_t phase_stride, const u64* __restrict__ Cloc, const u64* __restrict__ Ttot,
    int ntile, int Tp, int P, const float* __restrict__ amp, int Fa, int amp_hop, const float* __restrict__ tscale,
    int Fs, int ts_hop, const float* __restrict__ hscale, int H, float* __restrict__ out, int64_t out_stride, int Tout,
    int nrows_lds, HarmPhase hp) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    __shared__ u64 toff[256];
    __shared__ u64 twsum[4];
    const int tid = threadIdx.x, b = blockIdx.y;
    {   // exclusive prefix of the phase tile totals (as in osc_render_kernel)
        const u64 v = tid < ntile ? Ttot[(size_t)b * ntile + tid] : 0;
        const u64 incl = wave_incl_scan(v, tid & 63);
        if ((tid & 63) == 63) twsum[tid >> 6] = incl;
        toff[tid] = incl - v;
        __syncthreads();
        u64 base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += twsum[w];
        toff[tid] += base;
    }
    float* hs = hsm;                     // [H] per-harmonic scale (used directly when there is no amplitude tensor)
    float* rows = hsm + ((H + 3) & ~3);  // [nrows_lds][H] amplitude rows, per-harmonic scale already folded in
    float* ph0 = rows + (size_t)(amp ? nrows_lds : 1) * H;   // [2][H] cos / sin of 2 pi initial_phase (only if given)
    for (int h = tid; h < H; h += HARM_THREADS) hs[h] = hscale ? hscale[h] : 1.0f;
    if (hp.phi0)
        for (int h = tid; h < H; h += HARM_THREADS) {
            const float x = hp.phi0[(size_t)b * H + h];
            sincospif(2.0f * (x - floorf(x)), &ph0[H + h], &ph0[h]);
        }
    const float* pb = phase + (size_t)b * phase_stride;
    const u64* cb = Cloc + (size_t)b * Tp;
    const double scale_a = 18446744073709551616.0, scale_d = scale_a / (double)P;
    const float inv_ah = 1.0f / (float)amp_hop, inv_sh = 1.0f / (float)ts_hop;
    const int t_lo = blockIdx.x * HARM_THREADS;
    int row_lo = 0;
    if (amp) {  // stage the amplitude rows this block interpolates between
        row_lo = Fa >= 2 ? min(t_lo / amp_hop, Fa - 2) : 0;
        const float* ab = amp + ((size_t)b * Fa + row_lo) * H;
        const int nr = min(nrows_lds, Fa - row_lo);
        for (int e = tid; e < nr * H; e += HARM_THREADS) rows[e] = ab[e] * (hscale ? hscale[e % H] : 1.0f);
    }
    __syncthreads();
    const int t = t_lo + tid;
    if (t >= Tout) return;
    HarmSample sm = harm_sample(pb, cb, toff, t, Tp, P, scale_a, scale_d);
    sm.Phi += harm_offset_q64(hp, b, t);   // the harmonics multiply Phi + offset: wraps exactly, like Phi itself
    int fa = 0;
    float wa = 0.f;
    if (Fa >= 2) { fa = min(t / amp_hop, Fa - 2); wa = (float)(t - fa * amp_hop) * inv_ah; }
    float ts = 1.0f;
    if (tscale) {
        int fs = 0;
        float ws = 0.f;
        if (Fs >= 2) { fs = min(t / ts_hop, Fs - 2); ws = (float)(t - fs * ts_hop) * inv_sh; }
        const float s0 = tscale[(size_t)b * Fs + fs], s1 = tscale[(size_t)b * Fs + (Fs >= 2 ? fs + 1 : fs)];
        ts = fmaf(ws, s1 - s0, s0);
    }
    // number of harmonics below Nyquist for this sample: the largest hl with (float)h * p < 0.5 for all h <= hl,
    // found from the quotient and corrected by the very comparison the reference makes (synth.py:440)
    int hl = sm.p > 0.f ? (int)fminf(0.5f / sm.p, (float)H) : H;
    while (hl < H && (float)(hl + 1) * sm.p < 0.5f) ++hl;
    while (hl > 0 && !((float)hl * sm.p < 0.5f)) --hl;
    float rs, rc;  // rotation by theta = 2 pi Phi
    harm_sincos(sm.Phi, rs, rc);
    float acc = 0.f;
    const float* r0 = amp ? rows + (size_t)(fa - row_lo) * H : hs;
    const float* r1 = amp ? r0 + H : hs;
    for (int h0 = 1; h0 <= H; h0 += HARM_ANCHOR) {   // blocks of 32 harmonics, fully unrolled, exact re-anchor each
        float s, c;
        harm_sincos((u64)h0 * sm.Phi, s, c);
#pragma unroll
        for (int i = 0; i < HARM_ANCHOR; ++i) {
            const int h = h0 + i;
            if (h <= H) {  // uniform
                const float a0 = r0[h - 1];
                const float a = fmaf(wa, r1[h - 1] - a0, a0);
                float se = s, ce = c;   // sin / cos of 2 pi (h (Phi + offset) + initial_phase[h])
                if (hp.phi0) {          // uniform
                    const float cp = ph0[h - 1], sp = ph0[H + h - 1];
                    se = fmaf(s, cp, c * sp);
                    ce = fmaf(c, cp, -s * sp);
                }
                if (DERIV) acc = h <= hl ? fmaf(a * (float)h, ce, acc) : acc;
                else       acc = h <= hl ? fmaf(a, se, acc) : acc;
                const float sn = fmaf(s, rc, c * rs), cn = fmaf(c, rc, -s * rs);
                s = sn;
                c = cn;
            }
        }
    }
    out[(size_t)b * out_stride + t] = DERIV ? acc * ts * 6.283185307179586f : acc * ts;
}

// Gradient w.r.t. the amplitude rows.  One wave per (utterance, amplitude segment sg): every sample of the segment is
// visited ONCE and feeds both rows it interpolates between (row sg with 1-w, row sg+1 with w).  Harmonics are
// processed in chunks of HARM_ANCHOR = 32 (one exact re-anchor per sample and chunk): a lane keeps 2 x 32 running sums
// in registers over its samples (stride 64) and only at the end of the chunk are they reduced across the wave -- the
// first version reduced every harmonic of every 64 samples across lanes (930 cross-lane operations per 64 samples) and
// visited each sample from both of its rows: 617 us at B=32, H=155.  Partial sums part[b][sg][2][H] are combined by
// harm_bwd_combine_kernel (deterministic, no atomics).
__global__ __launch_bounds__(64) void harm_bwd_kernel(
    const float* __restrict__ phase, int64_t phase_stride, const u64* __restrict__ Cloc, const u64* __restrict__ Ttot,
    int ntile, int Tp, int P, int Fa, int amp_hop, const float* __restrict__ tscale, int Fs, int ts_hop,
    const float* __restrict__ hscale, int H, const float* __restrict__ g_out, int64_t g_out_stride,
    float* __restrict__ part, int nseg, int Tout, HarmPhase hp) {
    __shared__ u64 toff[256];
    const int lane = threadIdx.x, sg = blockIdx.x, b = blockIdx.y;
    {   // exclusive prefix of the phase tile totals, 4 passes of one wave
        u64 carry = 0;
        for (int base = 0; base < 256; base += 64) {
            const int i = base + lane;
            const u64 v = i < ntile ? Ttot[(size_t)b * ntile + i] : 0;
            const u64 incl = wave_incl_scan(v, lane);
            toff[i] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const float* pb = phase + (size_t)b * phase_stride;
    const u64* cb = Cloc + (size_t)b * Tp;
    const double scale_a = 18446744073709551616.0, scale_d = scale_a / (double)P;
    const float inv_ah = 1.0f / (float)amp_hop, inv_sh = 1.0f / (float)ts_hop;
    // samples of segment sg: [sg*hop, (sg+1)*hop), the last segment also owns the clamped tail
    const int t_lo = sg * amp_hop;
    const int t_hi = sg == nseg - 1 ? Tout : min((sg + 1) * amp_hop, Tout);
    float* p0 = part + (((size_t)b * nseg + sg) * 2) * H;
    float* p1 = p0 + H;
    for (int h0 = 1; h0 <= H; h0 += HARM_ANCHOR) {
        float a0[HARM_ANCHOR], a1[HARM_ANCHOR];
#pragma unroll
        for (int i = 0; i < HARM_ANCHOR; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        for (int t = t_lo + lane; t < t_hi; t += 64) {
            HarmSample sm = harm_sample(pb, cb, toff, t, Tp, P, scale_a, scale_d);
            sm.Phi += harm_offset_q64(hp, b, t);
            const float wa = Fa >= 2 ? (float)(t - t_lo) * inv_ah : 0.f;
            float ts = 1.0f;
            if (tscale) {
                int fs = 0;
                float ws = 0.f;
                if (Fs >= 2) { fs = min(t / ts_hop, Fs - 2); ws = (float)(t - fs * ts_hop) * inv_sh; }
                const float s0 = tscale[(size_t)b * Fs + fs], s1 = tscale[(size_t)b * Fs + (Fs >= 2 ? fs + 1 : fs)];
                ts = fmaf(ws, s1 - s0, s0);
            }
            const float g = g_out[(size_t)b * g_out_stride + t] * ts;
            const float g0 = g * (1.0f - wa), g1 = g * wa;
            float rs, rc, s, c;
            harm_sincos(sm.Phi, rs, rc);
            harm_sincos((u64)h0 * sm.Phi, s, c);
#pragma unroll
            for (int i = 0; i < HARM_ANCHOR; ++i) {
                const int h = h0 + i;
                float se = s;
                if (hp.phi0 && h <= H) {   // sin(2 pi (h Phi + initial_phase[h])); two loads per term: the rare path
                    float sp, cp;
                    const float x = hp.phi0[(size_t)b * H + h - 1];
                    sincospif(2.0f * (x - floorf(x)), &sp, &cp);
                    se = fmaf(s, cp, c * sp);
                }
                const float v = (h <= H && (float)h * sm.p < 0.5f) ? se : 0.f;
                a0[i] = fmaf(g0, v, a0[i]);
                a1[i] = fmaf(g1, v, a1[i]);
                const float sn = fmaf(s, rc, c * rs), cn = fmaf(c, rc, -s * rs);
                s = sn;
                c = cn;
            }
        }
#pragma unroll
        for (int i = 0; i < HARM_ANCHOR; ++i) {
            float v0 = a0[i], v1 = a1[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); }
            const int h = h0 + i;
            if (lane == 0 && h <= H) {
                const float hsv = hscale ? hscale[h - 1] : 1.0f;
                p0[h - 1] = v0 * hsv;
                p1[h - 1] = v1 * hsv;
            }
        }
    }
}

// g_amp[b][f][h] = part[b][f][0][h] (segment f, weight 1-w) + part[b][f-1][1][h] (segment f-1, weight w)
__global__ void harm_bwd_combine_kernel(const float* __restrict__ part, float* __restrict__ g_amp, int B, int Fa, int H,
                                        int nseg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Fa * H) return;
    const int h = idx % H, f = (idx / H) % Fa, b = idx / (H * Fa);
    float v = 0.f;
    if (f < nseg) v += part[(((size_t)b * nseg + f) * 2 + 0) * H + h];
    if (f >= 1) v += part[(((size_t)b * nseg + f - 1) * 2 + 1) * H + h];
    g_amp[idx] = v;
}

struct HarmGeom {
    int P, N, ntile, nrows;
    size_t off_cw, off_ttot, off_part, total;
};
static int harm_geom(int B, int Tp, int phase_hop, int amp_hop, HarmGeom* g, int Fa = 0, int H = 0) {
    g->P = phase_hop;
    g->N = phase_hop > 1 ? (Tp - 1) * phase_hop + 1 : Tp;
    g->ntile = (int)ceil_div(Tp, OSC_SCAN_TILE);
    g->nrows = HARM_THREADS / (amp_hop > 0 ? amp_hop : 1) + 3;
    size_t o = 0;
    g->off_cw = o;   o = align_up(o + sizeof(u64) * (size_t)B * Tp, 256);
    g->off_ttot = o; o = align_up(o + sizeof(u64) * (size_t)B * g->ntile, 256);
    g->off_part = o; o = align_up(o + sizeof(float) * (size_t)B * (Fa > 1 ? Fa - 1 : 1) * 2 * H, 256);
    g->total = o;
    return 0;
}
static int harm_check(const char* who, const float* phase, int B, int Tp, int phase_hop, const float* amp, int Fa,
                      int amp_hop, const float* tscale, int Fs, int ts_hop, int H, int Tout, const HarmGeom& g) {
    if (!phase || B < 1 || Tp < 1 || phase_hop < 1 || H < 1 || H > 4096)
        return fail(GOLF_EINVAL, "%s: bad size (B=%d Tp=%d phase_hop=%d H=%d)", who, B, Tp, phase_hop, H);
    if ((amp && (Fa < 1 || amp_hop < 1)) || (tscale && (Fs < 1 || ts_hop < 1)))
        return fail(GOLF_EINVAL, "%s: bad amplitude / scale geometry", who);
    int expect = g.N;
    if (amp) expect = std::min(expect, amp_hop > 1 ? (Fa - 1) * amp_hop + 1 : Fa);
    if (tscale) expect = std::min(expect, ts_hop > 1 ? (Fs - 1) * ts_hop + 1 : Fs);
    if (Tout != expect) return fail(GOLF_EINVAL, "%s: Tout=%d, expected %d", who, Tout, expect);
    if (amp && (size_t)(g.nrows + 1) * H * sizeof(float) > 60 * 1024)
        return fail(GOLF_EUNSUPPORTED, "%s: amplitude hop %d too fine for %d harmonics (LDS staging); pass amplitudes "
                    "at a coarser hop or fold them into tscale/hscale", who, amp_hop, H);
    return GOLF_OK;
}

// ---- round 6: the source and the filter's transition maps in ONE launch -------------------------------------
// A lone batch used to walk oscillator totals -> oscillator -> transition maps (|| zero-state pass) -> pre-pass -> chunk passes.
// The maps need only the coefficients, not the source: their 637 issue-bound waves (one per SIMD of 160 CUs, 38 us) and the
// oscillator's LDS-heavy, latency-bound workgroups are complementary, and the HIP graph executor does not run two branches of a
// graph side by side to any effect (round 5) -- so the two kernels are one grid:
//   workgroups [0, nblk_f)   p1f_body: waves 0..3 run a transition-map wave each (one per SIMD), waves 4..7 leave at once;
//   the rest                 one oscillator unit (utterance, 2048-output tile) each, tile-major like osc_fused2_kernel's grid.
// The transition workgroups come FIRST in dispatch order: they are the long pole and want a CU each; the oscillator's workgroups
// fill what is left -- the 96 idle CUs and, beside the transition waves, the two free wave slots per SIMD of the 160 busy ones (the
// register allocation is held to three waves per SIMD for that).  LDS: one launch-wide dynamic size (the oscillator's ~80 KB), so
// a transition workgroup and an oscillator workgroup share a CU's 160 KB.
struct SrcArgs {
    const float* phase; int64_t phase_stride; OscLook look; u64* Ttot; const float* wsel; int Fw; const float* table; int n_tab, L,
        lshift, Tp, hop_t; const float* Bf4; float* out; int64_t out_stride; int Tout, dmin, nrows; const float* addend;
    int64_t addend_stride; int Tadd, B, ntile;
};
struct MapArgs { const float* a; float* PhiT; int F, M, hop, L, NP, nq; float* pmax; unsigned* fixcnt; int B; float* Phi; };
#ifdef SRCMAPS_TIMING   // dev build (tools/srcmaps_timeline.py): entry / exit on the 100 MHz clock all XCDs share + where the workgroup ran
__device__ unsigned long long g_srcmaps_rt[4 * 2048];
extern "C" int golf_debug_srcmaps_rt(unsigned long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_srcmaps_rt), sizeof(unsigned long long) * (size_t)n);
}
#define SRCMAPS_RT(i, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048 && (threadIdx.x == 0 || (i) == 3)) g_srcmaps_rt[4 * blockIdx.x + (i)] = (v); } while (0)
#else
#define SRCMAPS_RT(i, v) do { } while (0)
#endif
template <int EE, int KS, int W, int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void source_maps_kernel(SrcArgs o, MapArgs m,
                                                                                                     int nblk_f) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SRCMAPS_RT(0, __builtin_amdgcn_s_memrealtime());
    SRCMAPS_RT(2, ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11))) | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 32));
    if ((int)blockIdx.x < nblk_f) {
        if (threadIdx.x >= 64 * P1F_WPB) return;   // (the waves of a transition workgroup never synchronise with each other)
#ifndef SRCMAPS_PRIO_MAPS
#define SRCMAPS_PRIO_MAPS 0
#endif
        __builtin_amdgcn_s_setprio(SRCMAPS_PRIO_MAPS);
        p1f_body<W, NT>(m.a, m.PhiT, m.F, m.M, m.hop, m.L, m.NP, m.nq, smem, (int)blockIdx.x, m.pmax, m.fixcnt, m.B, m.Phi);
        SRCMAPS_RT(1, __builtin_amdgcn_s_memrealtime());
    } else {
#ifndef SRCMAPS_PRIO_OSC
#define SRCMAPS_PRIO_OSC 1
#endif
        const int u = (int)blockIdx.x - nblk_f;
#ifdef SRCMAPS_HI_UNITS
        if (u < SRCMAPS_HI_UNITS) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(SRCMAPS_PRIO_OSC);
#else
        __builtin_amdgcn_s_setprio(SRCMAPS_PRIO_OSC);
#endif
        const int tile = u / o.B, b = u - tile * o.B;
        osc_fused2_tile<EE, KS, 2048, 512>(o.phase, o.phase_stride, o.look, o.Ttot, o.wsel, o.Fw, o.table, o.n_tab, o.L, o.lshift,
                                           o.Tp, o.hop_t, o.Bf4, o.out, o.out_stride, o.Tout, o.dmin, o.nrows, o.addend,
                                           o.addend_stride, o.Tadd, smem, tile, b, o.ntile);
        SRCMAPS_RT(1, __builtin_amdgcn_s_memrealtime());
    }
}

}  // namespace golf

using namespace golf;

extern "C" size_t golf_glottal_osc_workspace_bytes(int B, int Tp, int phase_hop, int Fw, int w_hop, int L, int os) {
    if (B < 1 || Tp < 1 || phase_hop < 1 || Fw < 1 || w_hop < 1 || os < 1) return 0;
    OscGeom g;
    osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
    return g.total;
}

// The fused forward's geometry, and whether it serves this call (the backward asks too: a workspace the fused forward filled
// still holds the tile totals and the transposed tap fragments, GOLF_OSC_WS_KEPT).
struct OscFused2 { int dmin, dmax, KS, nrows, ntile2, lshift, TO, ntile_f; size_t lds; };
static int osc_unfused_env() {
    static const int v = [] { const char* e = getenv("GOLF_OSC_UNFUSED"); return e ? atoi(e) : 0; }();  // A/B knob
    return v;
}
static bool osc_fused2_plan(const OscGeom& g, const float* table, int L, int os, int K, bool want_pre, int Tout, int B, OscFused2* f) {
    if (!(os == 4 && g.P == 4 && (L & (L - 1)) == 0 && L >= 8 && !want_pre && !osc_unfused_env())) return false;
    if ((uintptr_t)table & 15) return false;                    // table rows are fetched as 16-byte words
    const int half = (K - 1) / 2;
    f->dmin = -((half + os - 1) / os);
    f->dmax = half / os;
    const int nq = f->dmax - f->dmin + 1;                       // taps per polyphase branch
    f->KS = nq + 15 <= 48 ? 12 : 16;                            // K-steps of the 16-window Toeplitz product
    f->ntile2 = (int)ceil_div(Tout, OSCF_TO);                   // tiles of the totals launch (<= g.ntile: fits the workspace)
    f->lshift = 31 - __builtin_clz((unsigned)L);
    if (!(nq + 15 <= 64 && -f->dmin < 64)) return false;
    // (a 1536-output tile on 384 threads quantises B = 32 x 2 s better -- 1024 tiles = two full rounds of the chip's 512 workgroup
    //  slots instead of 768 = one and a half -- and measured WORSE, 21.0 against 18.0 us, and 29 % worse at B = 16 384: a
    //  workgroup's lifetime is a latency chain that does not shorten with the tile, so the rate is tiles in flight x outputs per
    //  tile.  One shape.)
    const int span = 2048 + 4 * f->KS;
    f->TO = 2048;
    f->nrows = (span * 4 - 2) / g.hop_t + 3;                    // a run of span*4 fine samples at any alignment touches so many frames
    const int XS = (span + OSCF2_XPAD * ((span + 15) >> 4) + 3) & ~3;
    f->lds = sizeof(float) * (4 * (size_t)XS + OSCF2_FRAG_LDS * (size_t)f->KS * 64) + 8 * (size_t)(f->nrows - 1) * (L + 2) + 160;
    f->ntile_f = (int)ceil_div(Tout, 2048);
    return f->nrows <= OSCF_MAXROWS && f->lds <= 160 * 1024;
}

// The taps' Toeplitz fragments, prepared once per tap set (ABI 6): [Bfr: 4 x 16 x 64 floats][Bf4: the same].
static constexpr size_t kTapFragFloats = 4 * 16 * 64;
struct TapGeom { int dmin, dmax, KS; bool ok; };
static TapGeom tap_geom(int K, int os) {
    TapGeom t;
    const int half = (K - 1) / 2;
    t.dmin = -((half + os - 1) / os);
    t.dmax = half / os;
    const int nq = t.dmax - t.dmin + 1;
    t.KS = nq + 15 <= 48 ? 12 : 16;
    t.ok = os == 4 && K >= 1 && (K & 1) && nq + 15 <= 64 && -t.dmin < 64;
    return t;
}
extern "C" size_t golf_glottal_osc_tap_fragments_bytes(int K, int os) {
    return tap_geom(K, os).ok ? 2 * kTapFragFloats * sizeof(float) : 0;
}
extern "C" int golf_glottal_osc_tap_fragments_f32(const float* taps, int K, int os, void* frags, size_t frags_bytes, void* stream) {
    const TapGeom t = tap_geom(K, os);
    if (!t.ok) return fail(GOLF_EUNSUPPORTED, "glottal_osc_tap_fragments: the fused oscillator takes os = 4 and an odd tap count of at most 195 (K=%d, os=%d)", K, os);
    if (!taps || !frags || frags_bytes < 2 * kTapFragFloats * sizeof(float) || ((uintptr_t)frags & 15))
        return fail(GOLF_EINVAL, "glottal_osc_tap_fragments: needs taps and a 16-byte aligned buffer of %zu bytes", 2 * kTapFragFloats * sizeof(float));
    float* Bfr = (float*)frags;
    hipLaunchKernelGGL(osc_tap_frags_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, taps, K, t.dmin, t.dmax, t.KS, Bfr + kTapFragFloats, Bfr);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_glottal_osc_fwd_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                        const float* wsel, int Fw, int w_hop, const float* table, int n_tab, int L,
                                        int os, int equal_energy, const float* taps, int K, float* pre, float* out,
                                        int64_t out_stride, int B, int Tout, void* ws, size_t ws_bytes, void* stream,
                                        const float* addend, int64_t addend_stride, int Tadd, const void* tap_frags) {
    if (int rc = osc_check(B, Tp, phase_hop, Fw, w_hop, n_tab, L, os, K, taps)) return rc;
    if (!phase || !wsel || !table || !out) return fail(GOLF_EINVAL, "glottal_osc_fwd: null pointer");
    const bool throughput = (equal_energy & GOLF_OSC_THROUGHPUT) != 0;   // the caller keeps batches in flight (see the header)
    equal_energy &= 1;
    OscGeom g;
    osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
    const int tout = os > 1 ? (g.N - 1) / os + 1 : g.N;
    if (Tout != tout) return fail(GOLF_EINVAL, "glottal_osc_fwd: Tout=%d, expected %d", Tout, tout);
    if (phase_stride < Tp || out_stride < Tout) return fail(GOLF_EINVAL, "glottal_osc_fwd: row stride too small");
    if (addend && (os <= 1 || Tadd < 0 || addend_stride < Tadd))
        return fail(GOLF_EINVAL, "glottal_osc_fwd: the fused addend needs oversampling > 1 and addend_stride >= Tadd >= 0");
    if (!ws || ws_bytes < g.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "glottal_osc_fwd: workspace needs %zu bytes, 256-aligned (got %zu)", g.total,
                    ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    u64* Cw = (u64*)((char*)ws + g.off_cw);
    u64* Ttot = (u64*)((char*)ws + g.off_ttot);
    // ---- fused path (the GOLF configuration): phase at hop 1, 4x oversampling, power-of-two table, no `pre` wanted
    OscFused2 f2;
    if (osc_fused2_plan(g, table, L, os, K, pre != nullptr, Tout, B, &f2)) {
        // ONE launch (round 6): the phase scan is inside the kernel (OscLook).  The taps' fragments come prepared (tap_frags,
        // golf_glottal_osc_tap_fragments_f32: once per tap set) or are laid out into the workspace by a small launch of their own.
        const float* Bf4 = tap_frags ? (const float*)tap_frags + kTapFragFloats : (const float*)((char*)ws + g.off_bf4);
        if (!tap_frags) {
            hipLaunchKernelGGL(osc_tap_frags_kernel, dim3(4), dim3(256), 0, st, taps, K, f2.dmin, f2.dmax, f2.KS,
                               (float*)((char*)ws + g.off_bf4), (float*)((char*)ws + g.off_bfr));
            GOLF_LAUNCH_CHECK();
        }
        OscLook look;
        look.gen = (unsigned*)((char*)ws + g.off_look);
        look.ent = (unsigned long long*)((char*)ws + g.off_look + 256 * ceil_div((size_t)B * 4, 256));
        // One launch or two.  The single-pass scan saves a launch and the second read of the phase (+3 % at B = 16 384: 112.0 vs
        // 109.0 G samples/s) and is what lets the oscillator share a grid with the transition maps; for one B = 32 batch alone the
        // two forms are equal (122.8 vs 122.5 us for the step).  With FOUR such batches in flight every oscillator launch is a
        // "first round" -- the tiles of an utterance enter together and wave 0 of every workgroup polls twice -- and the two
        // launches are cheaper: 69.4 - 69.9 against 70.7 - 70.8 us/step on one box, three alternations (profiles/r06_ab_r05_vs_r06.txt).
        // GOLF_OSC_THROUGHPUT asks for them below a device-filling batch; GOLF_OSC_TWO_LAUNCH=0/1 forces either (A/B knob).
        static const int tl_env = [] { const char* e = getenv("GOLF_OSC_TWO_LAUNCH"); return e ? atoi(e) : -1; }();
        const bool two_launch = tl_env >= 0 ? tl_env != 0 : (throughput && (int64_t)B * f2.ntile_f < 8192);
        if (two_launch) {   // round 5's form: the tile totals by a launch of their own (the phase is read twice)
            hipLaunchKernelGGL(osc_tile_totals_kernel<OSCF_TO>, dim3(f2.ntile2, B), dim3(osct_threads(OSCF_TO)), 0, st, phase, phase_stride,
                               Ttot, Tp, g.P, os, f2.ntile2, taps, K, f2.dmin, f2.KS, (float*)nullptr, f2.dmax, (float*)nullptr, (u64*)nullptr);
            GOLF_LAUNCH_CHECK();
            look.gen = nullptr;
        }
#define GOLF_FUSED2(EE, KSV, TOV, NTHV)                                                                               \
    do {                                                                                                              \
        static const hipError_t lds_attr = hipFuncSetAttribute(                                                       \
            (const void*)osc_fused2_kernel<EE, KSV, TOV, NTHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (lds_attr != hipSuccess) /* > 64 KB of dynamic LDS per workgroup needs the opt-in */                       \
            return fail((int)lds_attr, "glottal_osc_fwd: cannot raise the dynamic LDS limit: %s",                     \
                        hipGetErrorString(lds_attr));                                                                 \
        hipLaunchKernelGGL((osc_fused2_kernel<EE, KSV, TOV, NTHV>), dim3(B, f2.ntile_f), dim3(NTHV), f2.lds, st, phase, \
                           phase_stride, look, Ttot, wsel, Fw, table, n_tab, L,                                       \
                           f2.lshift, Tp, g.hop_t, (const float*)Bf4, out, out_stride, Tout, f2.dmin, f2.nrows,       \
                           addend, addend_stride, Tadd);                                                              \
    } while (0)
#define GOLF_FUSED2_T(EE, KSV) GOLF_FUSED2(EE, KSV, 2048, 512)
        if (f2.KS == 12) { if (equal_energy) GOLF_FUSED2_T(1, 12); else GOLF_FUSED2_T(0, 12); }
        else             { if (equal_energy) GOLF_FUSED2_T(1, 16); else GOLF_FUSED2_T(0, 16); }
#undef GOLF_FUSED2_T
#undef GOLF_FUSED2
        GOLF_LAUNCH_CHECK();
        return GOLF_OK;
    }
    if (int rc = launch_phase_tiles(phase, phase_stride, Cw, Ttot, Tp, g.P, os, g.ntile, B, st)) return rc;
    float* fine = os > 1 ? (pre ? pre : (float*)((char*)ws + g.off_pre)) : out;
    // the internal oversampled buffer uses a row stride that is a multiple of 4 floats (16-byte stores / loads); a
    // caller-provided `pre` is dense (B, N)
    const int64_t fine_stride = os > 1 ? (pre ? (int64_t)g.N : g.pre_stride) : out_stride;
    const size_t lds = sizeof(float) * 2 * (size_t)(L + 1);
    const bool pow2 = (L & (L - 1)) == 0 && L >= 2;
#define GOLF_RENDER(KERNEL)                                                                                            \
    hipLaunchKernelGGL(KERNEL, dim3(g.nint, B), dim3(OSC_RENDER_THREADS), lds, st, phase, phase_stride,               \
                       (const u64*)Cw, (const u64*)Ttot, g.ntile, wsel, Fw, table, n_tab, L, Tp, g.P, os, g.hop_t,    \
                       g.N, equal_energy, fine, fine_stride, (const float*)nullptr, (float*)nullptr)
    if (g.P == 4 && pow2 && equal_energy)
        GOLF_RENDER((osc_render_kernel<0, 4, 1>));
    else if (g.P == 4 && pow2)
        GOLF_RENDER((osc_render_kernel<0, 4, 0>));
    else if (g.P == 4)
        GOLF_RENDER((osc_render_kernel<0, 4>));
    else
        hipLaunchKernelGGL((osc_render_kernel<0, 0>), dim3(g.nint, B), dim3(OSC_RENDER_THREADS), lds, st, phase,
                           phase_stride, (const u64*)Cw, (const u64*)Ttot, g.ntile, wsel, Fw, table, n_tab, L, Tp, g.P,
                           os, g.hop_t, g.N, equal_energy, fine, fine_stride, (const float*)nullptr, (float*)nullptr);
    GOLF_LAUNCH_CHECK();
    if (os > 1) {
        if (int rc = launch_decimate((const float*)fine, g.N, fine_stride, taps, K, os, out, out_stride, Tout, B, addend,
                                     addend_stride, Tadd, st))
            return rc;
    }
    return GOLF_OK;
}

// golf_glottal_osc_fwd_f32 + golf_ltv_allpole_transitions_f32(.. | GOLF_SS_FAST_TRANSITIONS | GOLF_SS_MAPS_ONLY) as ONE launch where
// the shapes allow it (the fused oscillator's configuration, filter ring 24 / 22 taps: lpc_order 19 .. 22 at a hop that is a
// multiple of 24), as the two calls everywhere else: the results are the two calls' results bit for bit either way.
extern "C" int golf_source_transitions_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop, const float* wsel,
                                           int Fw, int w_hop, const float* table, int n_tab, int L, int os, int equal_energy,
                                           const float* taps, int K, float* out, int64_t out_stride, int B, int Tout,
                                           void* osc_ws, size_t osc_ws_bytes, const float* addend, int64_t addend_stride,
                                           int Tadd, const void* tap_frags, const float* a, int T, int F, int M, int hop,
                                           void* ss_ws, size_t ss_ws_bytes, int ss_flags, void* stream) {
    static const bool split_env = [] { const char* e = getenv("GOLF_SOURCE_MAPS_SPLIT"); return e && atoi(e) != 0; }();   // A/B knob
    OscGeom g;
    OscFused2 f2;
    SsPlan p;
    bool fuse = !split_env && phase && wsel && table && out && a && osc_ws && ss_ws && B >= 1 && Tp >= 1 && phase_hop >= 1 &&
                Fw >= 1 && w_hop >= 1 && os >= 1 && T >= 1 && F >= 1 && M >= 1 && hop >= 1 &&
                (ss_flags & GOLF_SS_FAST_TRANSITIONS) && (ss_flags & GOLF_SS_MAPS_ONLY) && !(ss_flags & GOLF_SS_SERIAL);
    if (fuse) fuse = osc_check(B, Tp, phase_hop, Fw, w_hop, n_tab, L, os, K, taps) == GOLF_OK;
    if (fuse) {
        osc_geom(B, Tp, phase_hop, Fw, w_hop, os, &g);
        const int tout = os > 1 ? (g.N - 1) / os + 1 : g.N;
        fuse = Tout == tout && phase_stride >= Tp && out_stride >= Tout && osc_ws_bytes >= g.total && !((uintptr_t)osc_ws & 255) &&
               (!addend || (Tadd >= 0 && addend_stride >= Tadd)) && osc_fused2_plan(g, table, L, os, K, false, Tout, B, &f2);
    }
    if (fuse) {
        const int mode = (ss_flags & GOLF_SS_CHUNKED) ? GOLF_SS_CHUNKED : 0;
        fuse = make_ss_plan(B, T, F, M, hop, &p, mode) && !p.serial && p.NP > 0 && p.W == 24 && p.NT == 22 &&
               ss_ws_bytes >= p.total && !((uintptr_t)ss_ws & 255);
    }
    if (!fuse) {   // the two calls, with their own argument checks and messages
        if (int rc = golf_glottal_osc_fwd_f32(phase, phase_stride, Tp, phase_hop, wsel, Fw, w_hop, table, n_tab, L, os, equal_energy,
                                              taps, K, nullptr, out, out_stride, B, Tout, osc_ws, osc_ws_bytes, stream, addend,
                                              addend_stride, Tadd, tap_frags))
            return rc;
        return golf_ltv_allpole_transitions_f32(a, B, T, F, M, hop, ss_ws, ss_ws_bytes, ss_flags, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const float* Bf4 = tap_frags ? (const float*)tap_frags + kTapFragFloats : (const float*)((char*)osc_ws + g.off_bf4);
    if (!tap_frags) {
        hipLaunchKernelGGL(osc_tap_frags_kernel, dim3(4), dim3(256), 0, st, taps, K, f2.dmin, f2.dmax, f2.KS,
                           (float*)((char*)osc_ws + g.off_bf4), (float*)((char*)osc_ws + g.off_bfr));
        GOLF_LAUNCH_CHECK();
    }
    SrcArgs o;
    o.phase = phase; o.phase_stride = phase_stride;
    o.look.gen = (unsigned*)((char*)osc_ws + g.off_look);
    o.look.ent = (unsigned long long*)((char*)osc_ws + g.off_look + 256 * ceil_div((size_t)B * 4, 256));
    o.Ttot = (u64*)((char*)osc_ws + g.off_ttot);
    o.wsel = wsel; o.Fw = Fw; o.table = table; o.n_tab = n_tab; o.L = L; o.lshift = f2.lshift; o.Tp = Tp; o.hop_t = g.hop_t;
    o.Bf4 = Bf4; o.out = out; o.out_stride = out_stride; o.Tout = Tout; o.dmin = f2.dmin; o.nrows = f2.nrows;
    o.addend = addend; o.addend_stride = addend_stride; o.Tadd = Tadd; o.B = B; o.ntile = f2.ntile_f;
    char* sw = (char*)ss_ws;
    MapArgs m;
    m.a = a; m.PhiT = (float*)(sw + p.off_phiT); m.F = F; m.M = M; m.hop = hop; m.L = p.L; m.NP = p.NP; m.nq = B * p.NP;
    m.pmax = (float*)(sw + p.off_pmax); m.fixcnt = (unsigned*)(sw + p.off_fixcnt); m.B = B;
    m.Phi = (ss_flags & GOLF_SS_TRAINING) ? (float*)(sw + p.off_phi) : (float*)nullptr;
    const int nblk_f = (int)ceil_div(m.nq, P1fGeom<24, 22, 2>::CPW * P1F_WPB);
    const unsigned grid = (unsigned)(nblk_f + B * f2.ntile_f);
#define GOLF_SRC_MAPS(EE, KSV)                                                                                        \
    do {                                                                                                              \
        static const hipError_t lds_attr = hipFuncSetAttribute(                                                       \
            (const void*)source_maps_kernel<EE, KSV, 24, 22>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (lds_attr != hipSuccess)                                                                                   \
            return fail((int)lds_attr, "source_transitions: cannot raise the dynamic LDS limit: %s",                  \
                        hipGetErrorString(lds_attr));                                                                 \
        hipLaunchKernelGGL((source_maps_kernel<EE, KSV, 24, 22>), dim3(grid), dim3(512), f2.lds, st, o, m, nblk_f);   \
    } while (0)
    if (f2.KS == 12) { if (equal_energy & 1) GOLF_SRC_MAPS(1, 12); else GOLF_SRC_MAPS(0, 12); }
    else             { if (equal_energy & 1) GOLF_SRC_MAPS(1, 16); else GOLF_SRC_MAPS(0, 16); }
#undef GOLF_SRC_MAPS
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_glottal_osc_bwd_wsel_f32(const float* g_out, int64_t g_out_stride, const float* phase,
                                             int64_t phase_stride, int Tp, int phase_hop, const float* wsel, int Fw,
                                             int w_hop, const float* table, int n_tab, int L, int os, int equal_energy,
                                             const float* taps, int K, float* g_wsel, int B, int Tout, void* ws,
                                             size_t ws_bytes, void* stream, const void* tap_frags) {
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
    // GOLF_OSC_WS_KEPT: the caller vouches that `ws` is as the forward of these same arguments (with pre == NULL) left it
    const bool ws_kept = (equal_energy & GOLF_OSC_WS_KEPT) != 0;
    equal_energy &= 1;
    // ---- fused path (the GOLF configuration, as in the forward): tile totals + osc_fused_bwd_kernel + a reduction
    if (os == 4 && g.P == 4 && (L & (L - 1)) == 0 && !osc_unfused_env()) {
        const int half = (K - 1) / 2;
        const int dmin = -((half + os - 1) / os);
        const int dmax = half / os;
        const int nq = dmax - dmin + 1;
        const int KS = nq + 15 <= 48 ? 12 : 16;
        const int nint_touched = (OSCB_TO * 4 - 2) / g.hop_t + 2;
        const int nrows = nint_touched + 1;
        const int spanY = OSCB_TO + 4 * KS;
        const int YS = (spanY + 2 * (spanY >> 4) + 2 + 3) & ~3;
        const size_t ldsb = sizeof(float) * ((size_t)YS + (size_t)OSCB_THREADS * 20 + (size_t)nrows * (L + 4));
        const int ntile2 = (int)ceil_div(Tp, OSCB_TO);            // tiles of COARSE SAMPLES here (Tp = Tout at hop 1)
        if (nq + 15 <= 64 && nrows <= OSCF_MAXROWS && ldsb <= 80 * 1024 && ntile2 <= g.ntile && L >= 8 && !((uintptr_t)table & 15) &&
            sizeof(float) * (size_t)B * ntile2 * OSCF_MAXROWS <= sizeof(float) * (size_t)B * g.pre_stride) {
            const int lshift = 31 - __builtin_clz((unsigned)L);
            u64* Ttot = (u64*)((char*)ws + g.off_ttot);
            float* Bfw = (float*)((char*)ws + g.off_bfr);
            const float* Bf = tap_frags ? (const float*)tap_frags : (const float*)Bfw;   // prepared fragments (ABI 6) or the workspace's
            float* part2 = (float*)((char*)ws + g.off_pre);      // the oversampled-gradient buffer is not needed here
            // The fused forward computed exactly these totals (same tiles: OSCB_TO == OSCF_TO, Tp == Tout at hop 1) and wrote
            // the transposed tap fragments next to its own: with its workspace intact the backward is two launches, not three
            // (5.9 us of the B = 32 training step).  Otherwise -- a caller that does not say, a forward that took the
            // three-kernel path -- they are recomputed: the backward must not depend on which forward variant ran.
            OscFused2 f2;
            const bool have_totals = ws_kept && OSCB_TO == OSCF_TO && osc_fused2_plan(g, table, L, os, K, false, Tout, B, &f2) &&
                                     f2.ntile2 == ntile2 && f2.KS == KS && f2.dmax == dmax;
            if (!have_totals) {
                hipLaunchKernelGGL(osc_tile_totals_kernel<OSCB_TO>, dim3(ntile2, B), dim3(osct_threads(OSCB_TO)), 0, st, phase, phase_stride, Ttot, Tp,
                                   g.P, os, ntile2, taps, K, dmin, KS, tap_frags ? (float*)nullptr : Bfw, dmax, (float*)nullptr, (u64*)nullptr);
                GOLF_LAUNCH_CHECK();
            }
#define GOLF_FUSED_BWD(EE, KSV)                                                                                       \
    do {                                                                                                              \
        static const hipError_t lds_attr = hipFuncSetAttribute(                                                       \
            (const void*)osc_fused_bwd_kernel<EE, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);       \
        if (lds_attr != hipSuccess)                                                                                   \
            return fail((int)lds_attr, "glottal_osc_bwd: cannot raise the dynamic LDS limit: %s",                     \
                        hipGetErrorString(lds_attr));                                                                 \
        hipLaunchKernelGGL((osc_fused_bwd_kernel<EE, KSV>), dim3(ntile2, B), dim3(OSCB_THREADS), ldsb, st, phase,     \
                           phase_stride, (const u64*)Ttot, ntile2, wsel, Fw, table, n_tab, L, lshift, Tp, g.hop_t,    \
                           (const float*)Bf, g_out, g_out_stride, Tout, dmax, nrows, part2);                          \
    } while (0)
            if (KS == 12) { if (equal_energy) GOLF_FUSED_BWD(1, 12); else GOLF_FUSED_BWD(0, 12); }
            else          { if (equal_energy) GOLF_FUSED_BWD(1, 16); else GOLF_FUSED_BWD(0, 16); }
#undef GOLF_FUSED_BWD
            GOLF_LAUNCH_CHECK();
            hipLaunchKernelGGL(osc_wsel_reduce_tiles_kernel, dim3((unsigned)ceil_div(B * Fw, 256)), dim3(256), 0, st,
                               (const float*)part2, g_wsel, B, Fw, ntile2, n_tab, g.hop_t);
            GOLF_LAUNCH_CHECK();
            return GOLF_OK;
        }
    }
    // three-kernel path; the phase prefix is recomputed: the fused forward leaves no per-sample prefix behind, and the
    // backward must not depend on which forward variant ran or on the caller keeping the workspace untouched in between
    u64* Cw = (u64*)((char*)ws + g.off_cw);
    if (int rc = launch_phase_tiles(phase, phase_stride, Cw, (u64*)((char*)ws + g.off_ttot), Tp, g.P, os, g.ntile, B,
                                    (hipStream_t)stream))
        return rc;
    float* g_pre = (float*)((char*)ws + g.off_pre);
    float* part = (float*)((char*)ws + g.off_part);
    if (int rc = launch_decimate_T(g_out, g_out_stride, Tout, taps, K, os, g_pre, g.N, B, st)) return rc;
    size_t lds = sizeof(float) * 2 * (size_t)(L + 1);
    if (lds < sizeof(float) * 2 * OSC_RENDER_THREADS) lds = sizeof(float) * 2 * OSC_RENDER_THREADS;
    const u64* Ttot = (const u64*)((char*)ws + g.off_ttot);
    hipLaunchKernelGGL((osc_render_kernel<1, 0>), dim3(g.nint, B), dim3(OSC_RENDER_THREADS), lds, st, phase, phase_stride,
                       Cw, Ttot, g.ntile, wsel, Fw, table, n_tab, L, Tp, g.P, os, g.hop_t, g.N, equal_energy,
                       (float*)nullptr, (int64_t)0, (const float*)g_pre, part);
    GOLF_LAUNCH_CHECK();
    hipLaunchKernelGGL(osc_wsel_reduce_kernel, dim3((unsigned)ceil_div(B * Fw, 256)), dim3(256), 0, st,
                       (const float*)part, g_wsel, B, Fw, g.nint, n_tab);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_wavetable_lookup_fwd_f32(const float* wrapped, int64_t wrapped_stride, const float* tables, int K,
                                             int L, int hop_t, float* out, int64_t out_stride, int B, int N,
                                             void* stream) {
    if (!wrapped || !tables || !out || B < 1 || N < 1 || K < 1 || L < 2 || hop_t < 1)
        return fail(GOLF_EINVAL, "wavetable_lookup_fwd: bad argument");
    if (wrapped_stride < N || out_stride < N) return fail(GOLF_EINVAL, "wavetable_lookup_fwd: row stride < N");
    const int64_t n = (int64_t)B * N;
    hipLaunchKernelGGL(wt_lookup_fwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, wrapped,
                       wrapped_stride, tables, K, L, hop_t, N, out, out_stride, B);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_wavetable_lookup_bwd_f32(const float* g_out, int64_t g_out_stride, const float* wrapped,
                                             int64_t wrapped_stride, const float* tables, int K, int L, int hop_t,
                                             float* g_wrapped, int64_t g_wrapped_stride, float* g_tables, int B, int N,
                                             void* stream) {
    if (!g_out || !wrapped || !tables || B < 1 || N < 1 || K < 1 || L < 2 || hop_t < 1)
        return fail(GOLF_EINVAL, "wavetable_lookup_bwd: bad argument");
    if (g_out_stride < N || wrapped_stride < N || (g_wrapped && g_wrapped_stride < N))
        return fail(GOLF_EINVAL, "wavetable_lookup_bwd: row stride < N");
    hipStream_t st = (hipStream_t)stream;
    if (g_tables) {
        hipError_t e = hipMemsetAsync(g_tables, 0, sizeof(float) * (size_t)B * K * L, st);
        if (e != hipSuccess) return fail((int)e, "wavetable_lookup_bwd: memset failed: %s", hipGetErrorString(e));
    }
    const int nint = (int)ceil_div(N, hop_t);
    const size_t lds = sizeof(float) * 2 * (size_t)L;
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(wt_lookup_bwd_kernel<true>, dim3(nint, B), dim3(256), g_tables ? lds : 0, st, g_out,
                           g_out_stride, wrapped, wrapped_stride, tables, K, L, hop_t, N, g_wrapped, g_wrapped_stride,
                           g_tables);
    else
        hipLaunchKernelGGL(wt_lookup_bwd_kernel<false>, dim3(nint, B), dim3(256), 0, st, g_out, g_out_stride, wrapped,
                           wrapped_stride, tables, K, L, hop_t, N, g_wrapped, g_wrapped_stride, g_tables);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_decimate_fir_f32(const float* x, int64_t x_stride, int N, const float* taps, int K, int os,
                                     float* out, int64_t out_stride, int B, int Tout, void* stream) {
    if (!x || !taps || !out || B < 1 || N < 1 || os < 2 || os > 64 || K < 1 || (K & 1) == 0)
        return fail(GOLF_EINVAL, "decimate_fir: bad argument (needs os in [2,64] and an odd number of taps)");
    if (Tout != (N - 1) / os + 1) return fail(GOLF_EINVAL, "decimate_fir: Tout=%d, expected %d", Tout, (N - 1) / os + 1);
    if (x_stride < N || out_stride < Tout) return fail(GOLF_EINVAL, "decimate_fir: row stride too small");
    return launch_decimate(x, N, x_stride, taps, K, os, out, out_stride, Tout, B, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int golf_decimate_fir_adj_f32(const float* g_out, int64_t g_out_stride, int Tout, const float* taps, int K,
                                       int os, float* g_x, int N, int B, void* stream) {
    if (!g_out || !taps || !g_x || B < 1 || N < 1 || os < 2 || os > 64 || K < 1 || (K & 1) == 0)
        return fail(GOLF_EINVAL, "decimate_fir_T: bad argument");
    if (Tout != (N - 1) / os + 1 || g_out_stride < Tout) return fail(GOLF_EINVAL, "decimate_fir_T: bad Tout / stride");
    return launch_decimate_T(g_out, g_out_stride, Tout, taps, K, os, g_x, N, B, (hipStream_t)stream);
}

extern "C" size_t golf_harmonic_osc_workspace_bytes(int B, int Tp, int phase_hop, int Fa, int H) {
    if (B < 1 || Tp < 1 || phase_hop < 1 || Fa < 0 || H < 1) return 0;
    HarmGeom g;
    harm_geom(B, Tp, phase_hop, 1, &g, Fa, H);
    return g.total;
}

template <bool DERIV>
static int harmonic_osc_run(const char* who, const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                            const float* amp, int Fa, int amp_hop, const float* tscale, int Fs, int ts_hop,
                            const float* hscale, int H, float* out, int64_t out_stride, int B, int Tout, void* ws,
                            size_t ws_bytes, void* stream, HarmPhase hp);
static int harm_phase_check(const char* who, const float* phase_offset, int Fo, int po_hop, int Tout) {
    // every frame count, Fo = 1 included: linear upsampling of Fo frames at hop po_hop yields (Fo - 1) po_hop + 1 samples
    // (a lone frame with a hop > 1 used to pass and was then read as a constant track: ADVICE r3)
    if (phase_offset && (Fo < 1 || po_hop < 1 || (int64_t)(Fo - 1) * po_hop + 1 < Tout))
        return fail(GOLF_EINVAL, "%s: phase_offset (%d frames at hop %d) does not cover %d samples", who, Fo, po_hop, Tout);
    return GOLF_OK;
}

extern "C" int golf_harmonic_osc_fwd_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                         const float* amp, int Fa, int amp_hop, const float* tscale, int Fs,
                                         int ts_hop, const float* hscale, int H, float* out, int64_t out_stride, int B,
                                         int Tout, void* ws, size_t ws_bytes, void* stream, const float* phase_offset,
                                         int Fo, int po_hop, const float* initial_phase) {
    if (int rc = harm_phase_check("harmonic_osc_fwd", phase_offset, Fo, po_hop, Tout)) return rc;
    return harmonic_osc_run<false>("harmonic_osc_fwd", phase, phase_stride, Tp, phase_hop, amp, Fa, amp_hop, tscale, Fs,
                                   ts_hop, hscale, H, out, out_stride, B, Tout, ws, ws_bytes, stream,
                                   HarmPhase{phase_offset, Fo, po_hop, initial_phase});
}

extern "C" int golf_harmonic_osc_dphase_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                                            const float* amp, int Fa, int amp_hop, const float* tscale, int Fs,
                                            int ts_hop, const float* hscale, int H, float* out, int64_t out_stride,
                                            int B, int Tout, void* ws, size_t ws_bytes, void* stream,
                                            const float* phase_offset, int Fo, int po_hop, const float* initial_phase) {
    if (int rc = harm_phase_check("harmonic_osc_dphase", phase_offset, Fo, po_hop, Tout)) return rc;
    return harmonic_osc_run<true>("harmonic_osc_dphase", phase, phase_stride, Tp, phase_hop, amp, Fa, amp_hop, tscale,
                                  Fs, ts_hop, hscale, H, out, out_stride, B, Tout, ws, ws_bytes, stream,
                                  HarmPhase{phase_offset, Fo, po_hop, initial_phase});
}

template <bool DERIV>
static int harmonic_osc_run(const char* who, const float* phase, int64_t phase_stride, int Tp, int phase_hop,
                            const float* amp, int Fa, int amp_hop, const float* tscale, int Fs, int ts_hop,
                            const float* hscale, int H, float* out, int64_t out_stride, int B, int Tout, void* ws,
                            size_t ws_bytes, void* stream, HarmPhase hp) {
    HarmGeom g;
    harm_geom(B > 0 ? B : 1, Tp > 0 ? Tp : 1, phase_hop > 0 ? phase_hop : 1, amp ? amp_hop : HARM_THREADS, &g,
              amp ? Fa : 0, H > 0 ? H : 1);
    if (int rc = harm_check(who, phase, B, Tp, phase_hop, amp, Fa, amp_hop, tscale, Fs, ts_hop, H, Tout, g))
        return rc;
    if (!out || out_stride < Tout || phase_stride < Tp) return fail(GOLF_EINVAL, "%s: bad output / stride", who);
    if (!ws || ws_bytes < g.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "%s: workspace needs %zu bytes, 256-aligned (got %zu)", who, g.total, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    u64* Cw = (u64*)((char*)ws + g.off_cw);
    u64* Ttot = (u64*)((char*)ws + g.off_ttot);
    if (int rc = launch_phase_tiles(phase, phase_stride, Cw, Ttot, Tp, g.P, 1, g.ntile, B, st)) return rc;
    const size_t lds = sizeof(float) * (((H + 3) & ~3) + (size_t)(amp ? g.nrows : 1) * H + (hp.phi0 ? 2 * (size_t)H : 0));
    hipLaunchKernelGGL(harm_kernel<DERIV>, dim3((unsigned)ceil_div(Tout, HARM_THREADS), B), dim3(HARM_THREADS), lds, st,
                       phase, phase_stride, (const u64*)Cw, (const u64*)Ttot, g.ntile, Tp, g.P, amp, amp ? Fa : 1,
                       amp ? amp_hop : 1, tscale, Fs, ts_hop, hscale, H, out, out_stride, Tout, g.nrows, hp);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}

extern "C" int golf_harmonic_osc_bwd_amp_f32(const float* g_out, int64_t g_out_stride, const float* phase,
                                             int64_t phase_stride, int Tp, int phase_hop, int Fa, int amp_hop,
                                             const float* tscale, int Fs, int ts_hop, const float* hscale, int H,
                                             float* g_amp, int B, int Tout, void* ws, size_t ws_bytes, void* stream,
                                             const float* phase_offset, int Fo, int po_hop, const float* initial_phase) {
    if (int rc = harm_phase_check("harmonic_osc_bwd_amp", phase_offset, Fo, po_hop, Tout)) return rc;
    HarmGeom g;
    harm_geom(B > 0 ? B : 1, Tp > 0 ? Tp : 1, phase_hop > 0 ? phase_hop : 1, amp_hop > 0 ? amp_hop : 1, &g, Fa,
              H > 0 ? H : 1);
    if (int rc = harm_check("harmonic_osc_bwd_amp", phase, B, Tp, phase_hop, g_amp, Fa, amp_hop, tscale, Fs, ts_hop, H,
                            Tout, g))
        return rc;
    if (!g_out || !g_amp || g_out_stride < Tout) return fail(GOLF_EINVAL, "harmonic_osc_bwd_amp: bad pointer / stride");
    if (!ws || ws_bytes < g.total || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "harmonic_osc_bwd_amp: workspace needs %zu bytes, 256-aligned (got %zu)", g.total,
                    ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    u64* Cw = (u64*)((char*)ws + g.off_cw);   // recomputed: the backward does not rely on the forward's scratch
    u64* Ttot = (u64*)((char*)ws + g.off_ttot);
    float* part = (float*)((char*)ws + g.off_part);
    if (int rc = launch_phase_tiles(phase, phase_stride, Cw, Ttot, Tp, g.P, 1, g.ntile, B, st)) return rc;
    const int nseg = Fa > 1 ? Fa - 1 : 1;
    hipLaunchKernelGGL(harm_bwd_kernel, dim3((unsigned)nseg, B), dim3(64), 0, st, phase, phase_stride, (const u64*)Cw,
                       (const u64*)Ttot, g.ntile, Tp, g.P, Fa, amp_hop, tscale, Fs, ts_hop, hscale, H, g_out,
                       g_out_stride, part, nseg, Tout, HarmPhase{phase_offset, Fo, po_hop, initial_phase});
    GOLF_LAUNCH_CHECK();
    const int n = B * Fa * H;
    hipLaunchKernelGGL(harm_bwd_combine_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)part,
                       g_amp, B, Fa, H, nseg);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}


// =============================================================================================
// Running phase of the general (fully differentiable) table oscillators, golf_amd.functional.wavetable_osc:
//   wrapped[b,n] = frac( cumsum( up(phase / os) )[n] + phase_offset[b,n] ),  n < N = (Tp-1)*phase_hop*os + 1 (Tp if hop*os == 1)
// Replaces F.interpolate + torch.cumsum + % 1 of IndexedGlottalFlowTable.forward, models/synth.py:239-255, with the same
// exact 64-bit fixed-point prefix the fused oscillator uses (osc_phase_tile_kernel): one thread per fine sample.
// =============================================================================================
namespace golf {
__global__ __launch_bounds__(256) void osc_wrapped_phase_kernel(const float* __restrict__ phase, int64_t phase_stride,
                                                                const u64* __restrict__ Cloc,
                                                                const u64* __restrict__ Ttot, int ntile, int Tp, int P,
                                                                int os, const float* __restrict__ poff,
                                                                int64_t poff_stride, float* __restrict__ out,
                                                                int64_t out_stride, int N) {
    __shared__ u64 toff[256];
    __shared__ u64 twsum[4];
    const int tid = threadIdx.x, b = blockIdx.y;
    {   // exclusive prefix of the phase tile totals (as in harm_kernel)
        const u64 v = tid < ntile ? Ttot[(size_t)b * ntile + tid] : 0;
        const u64 incl = wave_incl_scan(v, tid & 63);
        if ((tid & 63) == 63) twsum[tid >> 6] = incl;
        toff[tid] = incl - v;
        __syncthreads();
        u64 base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += twsum[w];
        toff[tid] += base;
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    if (n >= N) return;
    const float* pb = phase + (size_t)b * phase_stride;
    const double scale_a = 18446744073709551616.0 / (double)os, scale_d = scale_a / (double)P;
    const int j = n / P, k = n - j * P;
    const int jc = j < Tp - 1 ? j : Tp - 1;
    const int jn = jc + 1 < Tp ? jc + 1 : Tp - 1;
    const u64 a = osc_fix_a(pb[jc], scale_a), d = osc_fix_d(pb[jc], pb[jn], scale_d);
    u64 Phi = Cloc[(size_t)b * Tp + jc] + toff[min(jc / OSC_SCAN_TILE, 255)] + (u64)(k + 1) * a + d * ((u64)k * (u64)(k + 1) / 2);
    if (poff) {
        const double x = (double)poff[(size_t)b * poff_stride + n];
        Phi += (u64)((x - floor(x)) * 18446744073709551616.0);
    }
    out[(size_t)b * out_stride + n] = (float)((double)Phi * 5.421010862427522e-20);   // * 2^-64
}
}  // namespace golf

extern "C" size_t golf_phase_accumulate_workspace_bytes(int B, int Tp) {
    if (B < 1 || Tp < 1) return 0;
    const int ntile = (int)golf::ceil_div(Tp, OSC_SCAN_TILE);
    return golf::align_up(sizeof(u64) * (size_t)B * Tp, 256) + golf::align_up(sizeof(u64) * (size_t)B * ntile, 256);
}

extern "C" int golf_phase_accumulate_f32(const float* phase, int64_t phase_stride, int Tp, int phase_hop, int os,
                                         const float* phase_offset, int64_t offset_stride, float* wrapped,
                                         int64_t wrapped_stride, int B, int N, void* ws, size_t ws_bytes, void* stream) {
    using namespace golf;
    if (!phase || !wrapped || B < 1 || Tp < 1 || phase_hop < 1 || os < 1 || N < 1)
        return fail(GOLF_EINVAL, "phase_accumulate: bad argument");
    const int P = phase_hop * os;
    const int Nmax = P > 1 ? (Tp - 1) * P + 1 : Tp;
    if (N > Nmax) return fail(GOLF_EINVAL, "phase_accumulate: N=%d exceeds the upsampled length %d", N, Nmax);
    if (phase_stride < Tp || wrapped_stride < N || (phase_offset && offset_stride < N))
        return fail(GOLF_EINVAL, "phase_accumulate: row stride too small");
    const size_t need = golf_phase_accumulate_workspace_bytes(B, Tp);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return fail(GOLF_EWORKSPACE, "phase_accumulate: workspace needs %zu bytes, 256-aligned (got %zu)", need, ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int ntile = (int)ceil_div(Tp, OSC_SCAN_TILE);
    u64* Cw = (u64*)ws;
    u64* Ttot = (u64*)((char*)ws + align_up(sizeof(u64) * (size_t)B * Tp, 256));
    if (int rc = launch_phase_tiles(phase, phase_stride, Cw, Ttot, Tp, P, os, ntile, B, st)) return rc;
    hipLaunchKernelGGL(osc_wrapped_phase_kernel, dim3((unsigned)ceil_div(N, 256), B), dim3(256), 0, st, phase, phase_stride,
                       (const u64*)Cw, (const u64*)Ttot, ntile, Tp, P, os, phase_offset, offset_stride, wrapped,
                       wrapped_stride, N);
    GOLF_LAUNCH_CHECK();
    return GOLF_OK;
}
