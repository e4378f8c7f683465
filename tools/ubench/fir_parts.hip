// dev: time the parts of the frame-FIR forward kernel separately (staging only / taps only / both)
#include "../../golf_amd/csrc/noise_fir.hip"
#include <cstdio>
using namespace golf;

template <int MODE>
__global__ __launch_bounds__(64 * FIR_WAVES) void k(const float* __restrict__ ex, int64_t ex_stride,
    const float* __restrict__ kern, int KS, float* __restrict__ y, int64_t y_stride, int B, int T, int nfr, int F, int N,
    int hop, int npass, int RS) {
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * FIR_WAVES + wv;
    if (unit >= B * nfr * npass) return;
    const int c = unit % npass, f = (unit / npass) % nfr, b = unit / (npass * nfr);
    float* sig = fir_lds + wv * RS;
    const int P = (N - 1) >> 1;
    const int ntaps = (N + 3) & ~3;
    const int span = 256 + ntaps + 4;
    const int t0 = f * hop + c * FIR_TILE;
    const BufRow xr(ex + b * ex_stride, T);
    if (MODE & 1) fir_stage<false>(sig, xr, t0 - P, span, lane);
    wave_lds_fence();
    FirAcc A;
    fir_zero(A);
    if (MODE & 2) fir_accum(A, sig, kern + (size_t)((MODE & 4) ? ((b * F + f) & 1) : (b * F + f)) * KS, ntaps, lane);
    else { A.e01.x = sig[lane]; }
    const f32x4 r = fir_finish(A);
    const BufRow yr(y + b * y_stride, nfr * hop);
    const int o = 4 * lane;
    const int lim = min(FIR_TILE, hop - c * FIR_TILE);
    yr.st(o + 0 < lim ? t0 + o + 0 : -1, r.x);
    yr.st(o + 1 < lim ? t0 + o + 1 : -1, r.y);
    yr.st(o + 2 < lim ? t0 + o + 2 : -1, r.z);
    yr.st(o + 3 < lim ? t0 + o + 3 : -1, r.w);
}

template <int MODE>
void run(const char* name, float* ex, float* kern, float* y) {
    const int B = 32, T = 48000, F = 200, N = 510, hop = 240, nfr = 199, KS = 512;
    const int RS = fir_region(256 + 512 + 4);
    const int units = B * nfr;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL((k<MODE>), dim3((units + 3) / 4), dim3(256), 4 * RS * 4, 0, ex, (int64_t)T, kern, KS, y, (int64_t)(nfr * hop), B, T, nfr, F, N, hop, 1, RS);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.2f us per launch\n", name, ms * 100);
}

int main() {
    float *ex, *kern, *y;
    hipMalloc(&ex, 4 * 32 * 48000); hipMalloc(&kern, 4 * 6400 * 512); hipMalloc(&y, 4 * 32 * 47760);
    hipMemset(ex, 0, 4 * 32 * 48000); hipMemset(kern, 0, 4 * 6400 * 512);
    run<0>("neither (launch + store)", ex, kern, y);
    run<1>("staging only", ex, kern, y);
    run<2>("taps only", ex, kern, y);
    run<3>("both", ex, kern, y);
    run<6>("taps only, cached coefs", ex, kern, y);
    return 0;
}
