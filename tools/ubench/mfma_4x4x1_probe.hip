// Microbenchmark / layout probe for v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per wave instruction):
// (1) which lane supplies A_b[i], B_b[j] and which lane/register receives D_b[i][j];
// (2) SIMD cycles per block-step of the transition kernel's inner loop shape: NM dependent-accumulator MFMAs
//     (1 or 2 chains) + NV VALU FMAs, at 1..4 waves per SIMD.   dev tool; result quoted in DESIGN.md §4.1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // A = 1000 + lane, B = 1 + lane/1000  ->  D = (1000 + la) * (1 + lb/1000)
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(1000.f + l, 1.f + l * 0.001f, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

template <int NM, int NCHAIN, int NV>
__global__ __launch_bounds__(256) void rate(float* out, const float* in, int iters) {
    float a[NM], b[NM];
#pragma unroll
    for (int h = 0; h < NM; ++h) { a[h] = in[threadIdx.x + h]; b[h] = in[64 + threadIdx.x + h]; }
    float v0 = in[threadIdx.x], v1 = in[threadIdx.x + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 d[NCHAIN];
#pragma unroll
        for (int u = 0; u < NCHAIN; ++u) d[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NM; ++h) d[h % NCHAIN] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[h], b[h], d[h % NCHAIN], 0, 0, 0);
#pragma unroll
        for (int u = 1; u < NCHAIN; ++u) d[0] += d[u];
        // independent VALU work (coefficient interpolation stand-in)
#pragma unroll
        for (int h = 0; h < NV; ++h) a[h % NM] = __builtin_fmaf(v0, a[h % NM], v1);
        // dependent tail (triangular solve stand-in): feeds the newest B operands
        float y0 = -d[0][0];
        float y1 = __builtin_fmaf(-v0, y0, -d[0][1]);
        float y2 = __builtin_fmaf(-v0, y1, __builtin_fmaf(-v1, y0, -d[0][2]));
        float y3 = __builtin_fmaf(-v0, y2, __builtin_fmaf(-v1, y1, __builtin_fmaf(-v0, y0, -d[0][3])));
        b[0] = y0; b[1] = y1; b[2] = y2; b[3] = y3;
        acc += d[0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a[0] + b[3];
}

template <int NM, int NCHAIN, int NV>
void run(int blocks) {
    float *out, *in;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&in, sizeof(float) * 1024);
    hipMemset(in, 0, sizeof(float) * 1024);
    const int iters = 4000;
    hipLaunchKernelGGL((rate<NM, NCHAIN, NV>), dim3(blocks), dim3(256), 0, 0, out, in, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<NM, NCHAIN, NV>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wps = blocks * 4 / 1024.0;
    printf("NM=%d chains=%d NV=%d waves/SIMD=%4.1f  SIMD cycles per block-step per wave = %6.1f (@2.4 GHz)\n", NM, NCHAIN, NV, wps,
           ms * 1e6 * 2.4 / iters / (wps < 1 ? 1 : wps));
    hipFree(out); hipFree(in);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    std::vector<float> h(256);
    hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 4, 5, 9, 63}) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            // decode: D = (1000 + la)(1 + lb/1000) -> la, lb
            double v = h[l * 4 + r];
            int best_a = -1, best_b = -1; double be = 1e9;
            for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
                double e = fabs((double)(float)(1000.f + la) * (double)(float)(1.f + lb * 0.001f) - v);
                if (e < be) { be = e; best_a = la; best_b = lb; }
            }
            printf("  reg%d = A(lane %2d) x B(lane %2d)", r, best_a, best_b);
        }
        printf("\n");
    }
    for (int b : {256, 512, 768, 1024}) {
        run<22, 1, 28>(b);
        run<22, 2, 28>(b);
        run<22, 2, 0>(b);
        run<22, 1, 0>(b);
    }
    return 0;
}
