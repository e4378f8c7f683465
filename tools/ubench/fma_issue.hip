// Microbenchmark: fp64/fp32 FMA issue cost for ONE wave and for several waves per SIMD, with all-VGPR operands
// (acc = fma(cf, h, acc), cf and h per-lane registers) as in the recursion kernels.  dev tool; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int C>
__global__ void chains(T* out, const T* in, int iters) {
    T acc[C], h[C], cf[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { acc[c] = in[threadIdx.x + c]; h[c] = in[64 + threadIdx.x + c]; cf[c] = in[128 + threadIdx.x + c]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if constexpr (sizeof(T) == 8) acc[c] = __builtin_fma(cf[(c + u) % C], h[c], acc[c]);
                else acc[c] = __builtin_fmaf(cf[(c + u) % C], h[c], acc[c]);
            }
        }
    }
    T s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, int C>
void run(const char* name, int blocks, int threads) {
    T *out, *in;
    hipMalloc(&out, sizeof(T) * blocks * threads);
    hipMalloc(&in, sizeof(T) * 512);
    hipMemset(in, 0, sizeof(T) * 512);
    const int iters = 4000;
    hipLaunchKernelGGL((chains<T, C>), dim3(blocks), dim3(threads), 0, 0, out, in, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chains<T, C>), dim3(blocks), dim3(threads), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 8 * C;
    double waves = (double)blocks * threads / 64;
    double per_simd = waves / 1024.0 < 1 ? 1 : waves / 1024.0;
    printf("%-4s chains=%2d waves=%5.0f  ns/FMA/wave=%6.2f (=%5.2f cyc @2.4GHz)  SIMD-time per wave-FMA=%5.2f cyc\n", name, C, waves,
           ms * 1e6 / n, ms * 1e6 / n * 2.4, ms * 1e6 / n * 2.4 / per_simd);
    hipFree(out); hipFree(in);
}

int main() {
    run<double, 2>("f64", 1, 64); run<double, 4>("f64", 1, 64); run<double, 6>("f64", 1, 64); run<double, 8>("f64", 1, 64); run<double, 16>("f64", 1, 64);
    run<float, 2>("f32", 1, 64); run<float, 4>("f32", 1, 64); run<float, 8>("f32", 1, 64); run<float, 16>("f32", 1, 64);
    run<double, 6>("f64", 1024, 64); run<double, 6>("f64", 2048, 64); run<double, 6>("f64", 3072, 64); run<double, 6>("f64", 4096, 64);
    run<float, 8>("f32", 1024, 64); run<float, 8>("f32", 2048, 64); run<float, 8>("f32", 4096, 64); run<float, 8>("f32", 8192, 64);
    return 0;
}
