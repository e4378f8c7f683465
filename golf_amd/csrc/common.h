// Shared host-side helpers for libgolf_hip.so (gfx950 only; no other backend exists).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "golf_amd.h"

namespace golf {

char* err_buf();
int fail(int code, const char* fmt, ...);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Every launch goes through this: returns the hipError_t (>0) of a failed launch.
#define GOLF_LAUNCH_CHECK()                                                              \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            snprintf(golf::err_buf(), 512, "%s:%d launch failed: %s", __FILE__, __LINE__, \
                     hipGetErrorString(e__));                                            \
            return (int)e__;                                                             \
        }                                                                                \
    } while (0)

// ---- chunk plan shared by forward and backward of the sample-wise filter -------------------
struct SsPlan {
    int W;      // ring/unroll width: W >= M+1, hop % W == 0  (0 => generic fallback)
    int NT;     // taps computed (>= M, zero padded)
    int L;      // chunk length: L % W == 0 and (L % hop == 0 || hop % L == 0)
    int NC;     // chunks per utterance = ceil(T/L)
    int NP;     // chunks that own a transition matrix = NC-1
    int seg;    // gradient segment length = min(L, hop)
    int NSEG;   // ceil(T/seg)
    bool serial;  // batch-parallel serial kernels (large batches): no transition matrices / boundary states in ws
    // workspace offsets (bytes)
    int NG, GS; // two-level boundary scan: NG groups of GS chunk maps (NG == 0: flat scan)
    size_t off_phi, off_phiT, off_z, off_E, off_z2, off_S, off_zadj, off_lam, off_g, off_pa, off_pg, off_mt, off_gv, off_pmax, total;
    // conditioning tiers (see lpc_fixup_kernel): per-utterance tier words, first-pass chunk start states of the two-level
    // scan (the delta-form refinement adds its correction to exactly these), the status words, and -- touched only for the
    // rare tier-3 utterances -- the transition matrices as doubles
    size_t off_tier, off_S1, off_status, off_phi64, off_fixcnt;
    size_t off_m64, off_v64, off_g64;   // tier 3 on the two-level path: fp64 group composites, group responses, group start states
    size_t off_mtT, off_L1, off_wadj, off_dadj;   // backward: two-level adjoint scan
    size_t off_gflag;   // merged chunk pass (lpc_fwdq2m_kernel): [B][NG] "defect response published" + [B] "fp64 states ready" words
};
bool make_ss_plan(int B, int T, int F, int M, int hop, SsPlan* p, int mode = 0);
int ss_serial_min_batch();

}  // namespace golf
