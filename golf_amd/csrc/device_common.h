// Device-side helpers shared by the recursion kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

namespace golf {

// ------------------------------------------------------------------------------------------
// Tile I/O: a wave owns 64 consecutive chunks of one utterance.  For each W-step block the wave moves
// a 64 x W tile between HBM and registers THROUGH LDS so that global accesses are coalesced
// (element e = it*64 + lane  <->  row e/W (= chunk), col e%W) while each lane computes on its own row.
// Row stride W+1 floats keeps both access patterns bank-conflict free.  Single-wave workgroups.
// ------------------------------------------------------------------------------------------
// Bounds-checked view of one utterance row of T floats (raw buffer descriptor, wave-uniform): loads outside
// [0,T) return 0 and stores outside are dropped BY THE HARDWARE — masking without branches, which is what lets
// hipcc keep the prefetch loads in flight (per-element `if (t<T)` made it wait for every load, 24 serial HBM
// round trips per block).
struct BufRow {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ BufRow(const float* p, int T)
        : rs(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, T * 4, 0x00020000)) {}
    __device__ __forceinline__ float ld(int t) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t * 4, 0, 0));
    }
    __device__ __forceinline__ void st(int t, float v) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, t * 4, 0, 0);
    }
};

template <int W, int ROWS = 64>
struct Tile {
    static constexpr int LD = W + 1;
    static constexpr int SIZE = ROWS * LD;
    static constexpr int ITS = ROWS * W / 64;  // elements per lane

    __device__ static __forceinline__ void rowcol(int it, int lq, int lr, int& row, int& col) {
        const int q = (it * 64) / W, r = (it * 64) % W;  // constants after unrolling
        col = r + lr;
        row = q + lq;
        if (col >= W) { col -= W; row += 1; }
        if (col >= W) { col -= W; row += 1; }            // W < 64: lr + r < 2W, at most two wraps (W >= 8: lr<W)
    }
    // global -> registers, coalesced order; element at t = tbase + row*L + col (0 outside [0,T))
    // DIR = -1 walks each row backwards in memory (time-reversed recursions): element (row, col) at tbase + row*L - col
    template <int DIR = 1>
    __device__ static __forceinline__ void fetch(float (&r)[ITS], const BufRow& src, int tbase, int L, int lq,
                                                 int lr) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            int row, col;
            rowcol(it, lq, lr, row, col);
            r[it] = src.ld(tbase + row * L + DIR * col);
        }
    }
    template <int DIR = 1>
    __device__ static __forceinline__ void store(const float (&r)[ITS], const BufRow& dst, int tbase, int L, int lq,
                                                 int lr) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            int row, col;
            rowcol(it, lq, lr, row, col);
            dst.st(tbase + row * L + DIR * col, r[it]);
        }
    }
    __device__ static __forceinline__ void scatter(float* lds, const float (&r)[ITS], int lq, int lr) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            int row, col;
            rowcol(it, lq, lr, row, col);
            lds[row * LD + col] = r[it];
        }
    }
    __device__ static __forceinline__ void gather(float (&r)[ITS], const float* lds, int lq, int lr) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            int row, col;
            rowcol(it, lq, lr, row, col);
            r[it] = lds[row * LD + col];
        }
    }
    __device__ static __forceinline__ void rows_load(float (&x)[W], const float* lds, int row) {
#pragma unroll
        for (int s = 0; s < W; ++s) x[s] = lds[row * LD + s];
    }
    __device__ static __forceinline__ void rows_store(float* lds, const float (&x)[W], int row) {
#pragma unroll
        for (int s = 0; s < W; ++s) lds[row * LD + s] = x[s];
    }
};

// Issue priority of the LIGHT waves (round 4).  A SIMD arbitrates VALU issue between its resident waves by priority, then
// age: with several batches in flight the transition kernel's long-running, issue-bound waves are the oldest on their
// SIMDs and win every slot they can use, while the short latency-bound waves of the other batches' kernels (oscillator
// tiles, zero-state pass, pre-pass, chunk passes) queue behind them although they need few slots.  Those kernels raise
// their priority once at entry; the transition waves stay at 0.  Build parameter for the A/B (0 = no s_setprio at all).
#ifndef GOLF_PRIO_LIGHT
#define GOLF_PRIO_LIGHT 1   // measured (tools/ab2.sh ab_prio, B = 32, 4 batches in flight): 0 -> 73.0 - 73.9 us/step, 1 -> 71.1 - 71.4, 3 -> 71.0 - 71.1; one batch alone unchanged
#endif
__device__ __forceinline__ void light_wave_priority() {
#if GOLF_PRIO_LIGHT > 0
    __builtin_amdgcn_s_setprio(GOLF_PRIO_LIGHT);
#endif
}

// Orders the LDS traffic of ONE wave (a wave that owns its LDS region needs no workgroup barrier).  The fences name the
// LDS address space: a fence over all address spaces also waits for the wave's outstanding GLOBAL stores and prefetch
// loads (s_waitcnt vmcnt(0)) -- in the kernels that store a tile and then fence, once per block of the recursion, that was
// a store round trip on the critical path of every block (found in round 3 with the block-recursion frame kernel, where it
// was 2/3 of the kernel: DESIGN.md 4.2).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// (Scalar-operand v_fmac_f32 by inline asm instead of what hipcc emits for the scans' broadcast-FMA chains measured slower,
// 31.2 -> 36.6 us: it SLP-packs pairs into v_pk_fma_f32.  Round 2; the helper is gone, the finding stays.)
__device__ __forceinline__ float f4get(const float4& v, int k) {
    return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}


// ---- quad (4-lane) tap-parallel helpers ------------------------------------------------------
constexpr int quad_tpl(int W, int NT) {
    for (int d = 1; d <= W; ++d)
        if (W % d == 0 && 4 * d >= NT) return d;
    return W;
}
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
#define DPP_XOR1 0xB1   /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E   /* quad_perm [2,3,0,1] */
#define DPP_SHR1 0x90   /* quad_perm [0,0,1,2]: lane r reads lane r-1 */
#define DPP_SHL1 0xF9   /* quad_perm [1,2,3,3]: lane r reads lane r+1 */
#define DPP_BC0  0x00   /* quad_perm [0,0,0,0]: broadcast lane 0 of the quad */


}  // namespace golf
