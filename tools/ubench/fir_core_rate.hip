// Microbenchmark: issue rate of the FIR core's FMA batch (64 v_pk_fma_f32 with SGPR coefficients) with the operands
// already in registers — no LDS, no scalar loads inside the loop.  dev tool.
#include "../../golf_amd/csrc/noise_fir.hip"
#include <cstdio>
using namespace golf;

__global__ __launch_bounds__(256) void k(float* out, const float* in, const float* coef, int iters) {
    __shared__ __attribute__((aligned(16))) float sig[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) sig[i] = in[i];
    __syncthreads();
    FirBatch b0;
    fir_batch_load(b0, reinterpret_cast<const f32x4*>(sig), coef + (blockIdx.x & 1) * 64, lane, 0);
    FirAcc A;
    fir_zero(A);
    for (int it = 0; it < iters; ++it) {
        fir_batch_fma(A, b0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 r = fir_finish(A);
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y + r.z + r.w;
}

int main() {
    float *out, *in, *coef;
    hipMalloc(&out, 4 * 8192 * 256); hipMalloc(&in, 4096); hipMalloc(&coef, 4096);
    hipMemset(in, 0, 4096); hipMemset(coef, 0, 4096);
    for (int blocks : {256, 512, 768, 1024, 2048}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, coef, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, in, coef, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wps = blocks * 4 / 1024.0;
        printf("waves/SIMD=%4.1f  SIMD cycles per v_pk_fma_f32 (FIR batch pattern) = %5.2f (@2.4 GHz)\n", wps,
               ms * 1e6 * 2.4 / (iters * 64.0) / wps);
    }
    return 0;
}
