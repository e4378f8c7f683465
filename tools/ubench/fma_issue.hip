// Microbenchmark: cycles per FMA instruction for one wave, by precision and number of independent chains,
// and aggregate rate with 1, 2, 4 waves per SIMD.  (dev tool; results recorded in DESIGN.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int C>
__global__ void chains(T* out, long long* cyc, int iters, T m) {
    T acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = (T)(threadIdx.x + c);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_fma(acc[c], m, (T)1.0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    T s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename T, int C>
void run(const char* name, int blocks, int threads) {
    T* out; long long* cyc;
    hipMalloc(&out, sizeof(T) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    const int iters = 2000;
    hipLaunchKernelGGL((chains<T, C>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10, (T)0.999);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((chains<T, C>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, (T)0.999);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double n = (double)iters * 16 * C;
    double waves = (double)blocks * threads / 64;
    printf("%-6s C=%d blocks=%5d thr=%4d  cyc/FMA(wave0)=%6.2f  wall %.3f ms  => %.2f wave-FMA/us/SIMD-equivalent(1024)\n", name, C,
           blocks, threads, h[0] / n, ms, waves * n / (ms * 1e3) / 1024.0);
    hipFree(out); hipFree(cyc);
}

int main() {
    // one wave alone
    run<double, 1>("f64", 1, 64); run<double, 2>("f64", 1, 64); run<double, 4>("f64", 1, 64); run<double, 8>("f64", 1, 64);
    run<float, 1>("f32", 1, 64); run<float, 2>("f32", 1, 64); run<float, 4>("f32", 1, 64); run<float, 8>("f32", 1, 64);
    // whole chip: 1, 2, 4 waves per SIMD
    run<double, 8>("f64", 1024, 64); run<double, 8>("f64", 2048, 64); run<double, 8>("f64", 4096, 64);
    run<float, 8>("f32", 1024, 64); run<float, 8>("f32", 2048, 64); run<float, 8>("f32", 4096, 64);
    run<double, 8>("f64", 256, 256); run<double, 8>("f64", 512, 256);
    return 0;
}
