// Microbenchmark: SIMD-time per v_pk_fma_f32 / v_fma_f32 wave-instruction with several waves per SIMD, coefficient from
// a VGPR pair or from an SGPR pair (as in the FIR kernels).  dev tool; result quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0: v_fma_f32 vgpr coef, 1: v_pk_fma_f32 vgpr coef, 2: v_pk_fma_f32 sgpr coef
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    constexpr int C = 8;
    f32x2 acc[C], h[C];
    f32x2 cv = {in[300 + threadIdx.x], in[400 + threadIdx.x]};
    const float cs0 = in[blockIdx.x & 1], cs1 = in[2 + (blockIdx.x & 1)];  // uniform -> SGPR
#pragma unroll
    for (int c = 0; c < C; ++c) { acc[c] = (f32x2){in[threadIdx.x + c], in[c]}; h[c] = (f32x2){in[64 + threadIdx.x + c], in[c + 9]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (MODE == 0) { acc[c].x = __builtin_fmaf(cv.x, h[c].x, acc[c].x); }
                else if (MODE == 1) acc[c] = __builtin_elementwise_fma(cv, h[c], acc[c]);
                else acc[c] = __builtin_elementwise_fma((u & 1) ? (f32x2){cs1, cs1} : (f32x2){cs0, cs0}, h[c], acc[c]);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks) {
    float *out, *in;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&in, sizeof(float) * 1024);
    hipMemset(in, 0, sizeof(float) * 1024);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, in, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 64;          // instructions per wave
    double wps = blocks * 4 / 1024.0;       // waves per SIMD
    printf("%-22s waves/SIMD=%4.1f  SIMD cycles per wave-instruction = %5.2f (@2.4 GHz)\n", name, wps, ms * 1e6 * 2.4 / n / wps);
    hipFree(out); hipFree(in);
}

int main() {
    for (int b : {256, 512, 1024, 2048}) {
        run<0>("v_fma_f32 (vgpr)", b);
        run<1>("v_pk_fma_f32 (vgpr)", b);
        run<2>("v_pk_fma_f32 (sgpr)", b);
    }
    return 0;
}
