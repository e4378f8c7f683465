#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x;
    // A[i][k] = 100*i + k (i = l%16, k = l/16);  B[k][j] = (k == 0) ? 1 : 0  -> D[i][j] = A[i][0] = 100*i  (tests row mapping)
    // second: B[k][j] = (k==K0) ... do general: a = 1000*i + 10*k, b = (k==2 ? 1 : 0) + 0.001*j*(k==2)
    const int i = l % 16, kk = l / 16;
    double a = 1000.0 * i + 10.0 * kk;
    double b = (kk == 2) ? (1.0 + 0.001 * (l % 16)) : 0.0;
    f64x4 c = {0, 0, 0, 0};
    f64x4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][2] * B[2][j] = (1000 i + 20) * (1 + 0.001 j)
    for (int l = 0; l < 64; l += 5) {
        printf("lane %2d:", l);
        for (int v = 0; v < 4; ++v) {
            double x = h[l * 4 + v];
            // decode: j from fractional factor, i from magnitude
            int ibest = -1, jbest = -1;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double e = (1000.0 * i + 20.0) * (1.0 + 0.001 * j);
                if (fabs(e - x) < 1e-6) { ibest = i; jbest = j; }
            }
            printf("  v%d=(%d,%d)", v, ibest, jbest);
        }
        printf("\n");
    }
    return 0;
}
