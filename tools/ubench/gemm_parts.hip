// dev: time the parts of the cosine-transform GEMM kernel (generated from noise_fir.hip by tools/ubench/gen_gemm_parts.py)
#include "../../golf_amd/csrc/noise_fir.hip"
#include <cstdio>
namespace golf {
template <int MODE, int DBG>
__global__ __launch_bounds__(256) void zp_gemm_dbg(const float* __restrict__ src, int src_stride,
                                                      const float* __restrict__ log_mag,
                                                      const float* __restrict__ window,
                                                      const float* __restrict__ Bm, float* __restrict__ out,
                                                      int out_stride, int G, int n_mag, int Pd) {
    __shared__ float Asm[2 * ZG_ROWS * ZG_LDA > ZG_ROWS * ZG_LDC ? 2 * ZG_ROWS * ZG_LDA : ZG_ROWS * ZG_LDC];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g0 = blockIdx.x * ZG_ROWS, c0 = blockIdx.y * ZG_COLS;
    const int N = 2 * (n_mag - 1), H = N >> 1;
    const int li = lane & 15, lk = lane >> 4;

    f32x4 acc[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
    if (!(DBG & 1)) { zg_fetch<MODE>(st, src, src_stride, window, g0, G, n_mag, 0, tid);
    zg_commit<MODE>(st, Asm, tid); }
    __syncthreads();
    const int nchunk = Pd / ZG_KC;
    for (int ch = 0; ch < nchunk; ++ch) {
        float* Acur = Asm + (ch & 1) * (ZG_ROWS * ZG_LDA);
        float* Anxt = Asm + ((ch + 1) & 1) * (ZG_ROWS * ZG_LDA);
        const bool more = ch + 1 < nchunk;
        if (more && !(DBG & 1)) zg_fetch<MODE>(st, src, src_stride, window, g0, G, n_mag, (ch + 1) * ZG_KC, tid);
        const float* ap = Acur + li * ZG_LDA + lk;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {  // ZG_KC / 4 = 16 k-steps per chunk = 2 phases of 8
            const int sg = ch * (ZG_KC / 4) + 8 * ph;  // global k-step of this phase
            const int sn = min(sg + 8, ksteps_all - 8);  // clamped at the very end: a redundant reload
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (!(DBG & 2)) { bq[ph ^ 1][i][0] = bp[(size_t)(4 * (sn + i)) * Pd];
                bq[ph ^ 1][i][1] = bp[(size_t)(4 * (sn + i)) * Pd + 16]; }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the 16 prefetch loads ahead of this phase's MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = 8 * ph + i;
                float a[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) a[rt] = ap[rt * 16 * ZG_LDA + 4 * s];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) { if (DBG & 4) { acc[rt][0][0] += a[rt] * bq[ph][i][0]; continue; }
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ph][i][0], acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], bq[ph][i][1], acc[rt][1], 0, 0, 0);
                }
            }
        }
        if (more && !(DBG & 1)) zg_commit<MODE>(st, Anxt, tid);
        __syncthreads();
    }
    float* As = Asm;
    // epilogue through LDS so that global stores run along rows (the last loop iteration ended with a barrier)
    float* Cs = As;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(rt * 16 + lk * 4 + r) * ZG_LDC + w * 32 + ct * 16 + li] = acc[rt][ct][r];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cc = lane + 64 * h, c = c0 + cc;
        if (MODE == 0) {
            const bool up = c < H, dn = c >= 1 && c <= H;
            const float wu = window[H + min(c, H - 1)], wd = window[H - min(c, H)];
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {
                const int row = w * 16 + rr, g = g0 + row;
                const float u = Cs[row * ZG_LDC + cc];
                float* orow = out + (size_t)g * out_stride;
                if (!(DBG & 8)) { if (g < G && up) orow[H + c] = u * wu;
                if (g < G && dn) orow[H - c] = u * wd; } else if (u == 123.f) orow[0] = u;
            }
        } else {
            const int ccl = min(c, n_mag - 1);
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {
                const int row = w * 16 + rr, g = g0 + row;
                const float u = Cs[row * ZG_LDC + cc];
                const float e = __expf(log_mag[(size_t)min(g, G - 1) * n_mag + ccl]);
                if (g < G && c < n_mag) out[(size_t)g * out_stride + c] = u * e;
            }
        }
    }
    if (MODE == 0 && blockIdx.y == 0) {  // zero the row padding [N, out_stride)
        for (int rr = 0; rr < 16; ++rr) {
            const int g = g0 + w * 16 + rr;
            if (g >= G) break;
            for (int j = N + lane; j < out_stride; j += 64) out[(size_t)g * out_stride + j] = 0.f;
        }
    }
}

}
using namespace golf;
template <int DBG>
void run(const char* name, float* lm, float* win, float* bas, float* kern) {
    const int G = 6400, n_mag = 256, Pd = 256, KS = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((zp_gemm_dbg<0, DBG>), dim3(G / 64, 2), dim3(256), 0, 0, lm, n_mag, (const float*)nullptr, win, bas, kern, KS, G, n_mag, Pd);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us per launch\n", name, ms * 100);
}
int main() {
    float *lm, *win, *bas, *kern;
    hipMalloc(&lm, 4 * 6400 * 256); hipMalloc(&win, 4 * 512); hipMalloc(&bas, 4 * 2 * 256 * 256); hipMalloc(&kern, 4 * 6400 * 512);
    hipMemset(lm, 0, 4 * 6400 * 256); hipMemset(win, 0, 4 * 512); hipMemset(bas, 0, 4 * 2 * 256 * 256);
    run<0>("full", lm, win, bas, kern);
    run<1>("no A staging", lm, win, bas, kern);
    run<2>("no B loads", lm, win, bas, kern);
    run<4>("no MFMA", lm, win, bas, kern);
    run<8>("no epilogue stores", lm, win, bas, kern);
    run<3>("no A staging, no B loads", lm, win, bas, kern);
    run<11>("MFMA only", lm, win, bas, kern);
    run<15>("nothing", lm, win, bas, kern);
    return 0;
}
