// The fp32 transition-map trajectories of the sample-wise filter (p1f_body) -- in a header since round 6: lpc_ss.hip launches
// them as lpc_p1f_kernel / lpc_p1fz_kernel, glottal_osc.hip as the transition-map workgroups of golf_source_transitions_f32's ONE
// launch (oscillator || transition maps; reference: models/sf.py:47-64 source -> end filter, models/filters.py:99-113).
#pragma once
#include "device_common.h"

namespace golf {

// fp32 homogeneous trajectories for the inference path (GOLF_SS_FAST_TRANSITIONS): 4 trajectories per lane held as
// two float2 rings, so that every dot-product step is a v_pk_fma_f32 with the coefficient broadcast to both halves
// and the coefficient interpolation is packed over tap pairs.  (The scalar fp32 instantiation of p1_hom_body is
// unusable: hipcc's SLP vectoriser re-packs the unrolled body into 256 VGPR + 256 AGPR + 1 KB of scratch, 425 us.)
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int P1F_WPB = 4;  // waves per workgroup of the fp32 transition kernel
template <int W, int NT, int NR = 2>
struct P1fGeom {
    static constexpr int KT = 2 * NR;               // trajectories per lane: NR float2 rings
    static constexpr int NG = (NT + KT - 1) / KT;   // trajectory groups per chunk
    static constexpr int CPW = 64 / NG;             // whole chunks per wave: lane = cl*NG + grp
    static constexpr int WAVE_TILE = 64;            // LDS per wave: the 64-word scratch of the per-chunk maximum (the maps leave
                                                    // by direct stores; a 16 KB transposition tile per wave was round 3's losing arm)
    static constexpr int TILE_FLOATS = P1F_WPB * WAVE_TILE;
};
// NR = 2 (four trajectories per lane) is what runs: 637 waves at B = 32, one per SIMD of 160 CUs, 36 - 38 us.
// NR = 3 (six per lane: the 11 coefficient interpolations of a step amortised over 66 instead of 44 dot-product FMAs, all 64
//   lanes of a wave used, -13 % packed FMAs on 398 waves) was measured in round 4 for the throughput chain: 232 VGPRs, 55 us
//   alone, and 71.5 instead of 69.0 us/step with four batches in flight -- with four streams the step rate is four chains per
//   (contended) chain time, and a longer kernel lengthens the chain more than its smaller instruction count shortens it.
template <int W, int NT, int NR = 2>
__device__ __forceinline__ void p1f_body(const float* __restrict__ a, float* __restrict__ PhiT, int F, int M, int hop,
                                         int L, int NP, int nq, float* __restrict__ tile_all, int blk_id,
                                         float* __restrict__ pmax, unsigned* __restrict__ fixcnt = nullptr, int B = 0,
                                         float* __restrict__ Phi = nullptr) {
    // Workgroups of P1F_WPB = 4 independent waves: there are fewer waves than SIMDs (637 for B=32) and every wave is
    // FMA-issue bound, so two waves sharing a SIMD double the kernel.  With single-wave workgroups the dispatcher's
    // SIMD choice depended on what ran before (measured: the same launch took 42 us or 63 us); a 4-wave workgroup
    // puts one wave on each SIMD of its CU.
    using G = P1fGeom<W, NT, NR>;
    constexpr int KT = G::KT, NG = G::NG, CPW = G::CPW;
    constexpr int NP2 = NT / 2;              // tap pairs (NT is even)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tile = tile_all + wv * G::WAVE_TILE;
    const int cl = lane / NG, grp = lane - cl * NG;
    const int q0 = (blk_id * P1F_WPB + wv) * CPW;
    if (fixcnt && blk_id == 0 && wv == 0)   // counters of the fix-up that follows in the next launch (see fixup_wave)
        for (int e = lane; e < 5 * B + 1; e += 64) fixcnt[e] = 0u;
    if (q0 >= nq) return;  // wave-uniform; the waves of a workgroup never synchronise with each other
    const int q = q0 + cl;
    const bool live = cl < CPW && q < nq;
    const int jb = KT * grp;
    const int qq = live ? q : (nq - 1);
    const int b = qq / NP, c = qq - b * NP;
    f32x2 h[NR][W];  // ring r = trajectories (jb + 2 r, jb + 2 r + 1)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int j = W - 1 - k, j0 = jb + 2 * r;
            h[r][k] = f32x2{(j == j0 && j0 < M) ? 1.f : 0.f, (j == j0 + 1 && j0 + 1 < M) ? 1.f : 0.f};
        }
    f32x2 a0p[NP2], ddp[NP2];
    const float inv_hop = 1.0f / (float)hop;
    int fcur = -1;
    const int nblk = L / W;
    for (int blk = 0; blk < nblk; ++blk) {
        const int t0 = c * L + blk * W;
        const int f = t0 / hop;
        if (f != fcur) {
            fcur = f;
            const float* pa0 = a + ((size_t)b * F + f) * M;
            const float* pa1 = pa0 + M;
#pragma unroll
            for (int pp = 0; pp < NP2; ++pp) {
                const int i0 = 2 * pp, i1 = 2 * pp + 1;
                const float u0 = i0 < M ? pa0[i0] : 0.f, u1 = i1 < M ? pa0[i1] : 0.f;
                const float v0 = i0 < M ? pa1[i0] : 0.f, v1 = i1 < M ? pa1[i1] : 0.f;
                a0p[pp] = f32x2{u0, u1};
                ddp[pp] = f32x2{(v0 - u0) * inv_hop, (v1 - u1) * inv_hop};
            }
        }
        const float n0 = (float)(t0 - f * hop);
#pragma unroll
        for (int s = 0; s < W; ++s) {
            const float n = n0 + (float)s;
            const f32x2 n2 = f32x2{n, n};
            f32x2 cfp[NP2];
#pragma unroll
            for (int pp = 0; pp < NP2; ++pp) cfp[pp] = __builtin_elementwise_fma(n2, ddp[pp], a0p[pp]);
            // NCH independent accumulation chains per ring (NR x NCH in flight).  Measured: 2 and 4 chains per ring run
            // the same 41.5 us -- the loop is bound by v_pk_fma_f32 issue (~6.4 cycles each for a lone wave), not by the
            // dependent-result latency
            constexpr int NCH = 2;
            f32x2 acc[NR][NCH];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int u = 0; u < NCH; ++u) acc[r][u] = f32x2{0.f, 0.f};
#pragma unroll
            for (int i = NT - 1; i >= 1; --i) {
                const float cf = (i & 1) ? cfp[i / 2].y : cfp[i / 2].x;
                const f32x2 c2 = f32x2{cf, cf};
                const int slot = (s - 1 - i + 2 * W) % W;
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r][i % NCH] = __builtin_elementwise_fma(c2, h[r][slot], acc[r][i % NCH]);
            }
#pragma unroll
            for (int st = NCH / 2; st >= 1; st /= 2)
#pragma unroll
                for (int u = 0; u < st; ++u)
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc[r][u] += acc[r][u + st];
            const float cf0 = cfp[0].x;
            const f32x2 c0 = f32x2{-cf0, -cf0};
            const int sp = (s - 1 + W) % W;
#pragma unroll
            for (int r = 0; r < NR; ++r) h[r][s] = __builtin_elementwise_fma(c0, h[r][sp], -acc[r][0]);
        }
    }
    // largest |entry| of the chunk's matrix -> pmax[q] (the conditioning guard of the chunked algorithm, see
    // kPhiGuard): compared as bit patterns, so that a NaN ranks above +inf and cannot hide
    if (pmax) {
        unsigned mx = 0u;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int k = 0; k < W; ++k) {
                mx = max(mx, __float_as_uint(fabsf(h[r][k].x)));
                mx = max(mx, __float_as_uint(fabsf(h[r][k].y)));
            }
        wave_lds_fence();
        reinterpret_cast<unsigned*>(tile)[lane] = mx;
        wave_lds_fence();
        if (live && grp == 0) {
#pragma unroll
            for (int u = 1; u < NG; ++u) mx = max(mx, reinterpret_cast<const unsigned*>(tile)[lane + u]);
            pmax[q] = __uint_as_float(mx);
        }
        wave_lds_fence();
    }
    // element (row i, trajectory jb + u) of the chunk's map: d s_end[i] / d s_start[jb + u]
    auto entry = [&](int i, int u) -> float {
        const f32x2 v = h[u / 2][(W - 1 - i + W) % W];
        return (i < M && jb + u < M) ? ((u & 1) ? v.y : v.x) : 0.f;
    };
    // Training (GOLF_SS_TRAINING): the backward's adjoint scan reads the maps in the other orientation, Phi[q][j][i] -- rows
    // j = this lane's trajectories, W contiguous floats each: direct float4 stores.
    if (Phi && live) {
#pragma unroll
        for (int u = 0; u < KT; ++u) {
            const int j = jb + u;
            if (j < NT) {
                float4* o = reinterpret_cast<float4*>(Phi + ((size_t)q * NT + j) * W);
#pragma unroll
                for (int i4 = 0; i4 < W / 4; ++i4)
                    o[i4] = make_float4(entry(4 * i4, u), entry(4 * i4 + 1, u), entry(4 * i4 + 2, u), entry(4 * i4 + 3, u));
            }
        }
    }
    // PhiT[q][i][j] = d s_end[i] / d s_start[j]: this lane owns columns jb..jb+KT-1 of every row i of its chunk.
    // Straight from the registers: 4 KT bytes per lane and row (16-byte stores with four trajectories per lane, 8-byte ones with
    // six: 24 grp bytes is only 8-byte aligned), the NG lanes of a chunk cover the row, NT rows.  No LDS tile: the workgroup's
    // footprint is the zero-state units' 7 KB, so that it fits a CU beside TWO oscillator workgroups.
    if (live) {
        float* prow = PhiT + (size_t)q * NT * W + jb;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if constexpr (KT == 4) {
                if (jb + 3 < W)
                    *reinterpret_cast<float4*>(prow + (size_t)i * W) = make_float4(entry(i, 0), entry(i, 1), entry(i, 2), entry(i, 3));
            } else {
#pragma unroll
                for (int u = 0; u < KT; u += 2)
                    if (jb + u + 1 < W)
                        *reinterpret_cast<float2*>(prow + u + (size_t)i * W) = make_float2(entry(i, u), entry(i, u + 1));
            }
            if constexpr (KT * NG < W) {   // columns no trajectory group covers: zeros (the group composites read whole rows)
                if (grp == NG - 1) {
#pragma unroll
                    for (int cc = KT * NG; cc < W; cc += 2)
                        *reinterpret_cast<float2*>(prow - jb + cc + (size_t)i * W) = make_float2(0.f, 0.f);
                }
            }
        }
    }
}

}  // namespace golf
