/* Numerics lab for the time-chunked LTV all-pole filter (dev tool, CPU): emulates the arithmetic of the HIP kernels
 * (fp32 FMA recursions, fp32 / fp64 transition trajectories) so that conditioning questions can be answered without a GPU.
 * One utterance per call; Python (tools/numlab/lab.py) does the orchestration and the scans. */
#include <math.h>
#include <stddef.h>

static inline int frame_of(int t, int hop, int F) { int f = t / hop; return f > F - 2 ? F - 2 : f; }

/* sequential fp32 recursion, coefficients a0 + n*d by FMA (as the kernels do), from state s0 (s0[i] = y[t0-1-i]) */
void seq_f32(const float* ex, const float* gain, const float* a, float* y, int t0, int t1, int F, int M, int hop,
             const float* s0, float* s_end) {
    float h[64];
    for (int i = 0; i < M; ++i) h[i] = s0 ? s0[i] : 0.f;
    const float inv_hop = 1.0f / (float)hop;
    for (int t = t0; t < t1; ++t) {
        const int f = frame_of(t, hop, F);
        const float n = (float)(t - f * hop);
        const float* p0 = a + (size_t)f * M;
        const float* p1 = p0 + M;
        const float g = fmaf(n, (gain[f + 1] - gain[f]) * inv_hop, gain[f]);
        float acc = 0.f;
        for (int i = M - 1; i >= 0; --i) acc = fmaf(fmaf(n, (p1[i] - p0[i]) * inv_hop, p0[i]), h[i], acc);
        const float v = (ex ? ex[t] * g : 0.f) - acc;
        for (int i = M - 1; i > 0; --i) h[i] = h[i - 1];
        h[0] = v;
        if (y) y[t] = v;
    }
    if (s_end) for (int i = 0; i < M; ++i) s_end[i] = h[i];
}

void seq_f64(const float* ex, const float* gain, const float* a, double* y, int t0, int t1, int F, int M, int hop,
             const double* s0, double* s_end) {
    double h[64];
    for (int i = 0; i < M; ++i) h[i] = s0 ? s0[i] : 0.0;
    for (int t = t0; t < t1; ++t) {
        const int f = frame_of(t, hop, F);
        const double w = (double)(t - f * hop) / (double)hop;
        const float* p0 = a + (size_t)f * M;
        const float* p1 = p0 + M;
        const double g = (double)gain[f] * (1 - w) + (double)gain[f + 1] * w;
        double acc = 0.0;
        for (int i = M - 1; i >= 0; --i) acc += ((double)p0[i] * (1 - w) + (double)p1[i] * w) * h[i];
        const double v = (ex ? (double)ex[t] * g : 0.0) - acc;
        for (int i = M - 1; i > 0; --i) h[i] = h[i - 1];
        h[0] = v;
        if (y) y[t] = v;
    }
    if (s_end) for (int i = 0; i < M; ++i) s_end[i] = h[i];
}

/* transition matrices of chunks c = 0..NP-1: Phi[c][i][j] = d s_end[i] / d s_start[j]; prec 32: fp32 trajectories */
void phi_all(const float* a, double* Phi, int NP, int L, int F, int M, int hop, int prec) {
    const float zg[2] = {0.f, 0.f};
    (void)zg;
    for (int c = 0; c < NP; ++c)
        for (int j = 0; j < M; ++j) {
            if (prec == 32) {
                float h[64];
                for (int i = 0; i < M; ++i) h[i] = i == j ? 1.f : 0.f;
                const float inv_hop = 1.0f / (float)hop;
                for (int t = c * L; t < (c + 1) * L; ++t) {
                    const int f = frame_of(t, hop, F);
                    const float n = (float)(t - f * hop);
                    const float* p0 = a + (size_t)f * M;
                    const float* p1 = p0 + M;
                    float ra = 0.f, rb = 0.f;
                    for (int i = M - 1; i >= 1; --i) {
                        const float cf = fmaf(n, (p1[i] - p0[i]) * inv_hop, p0[i]);
                        if (i & 1) ra = fmaf(cf, h[i], ra); else rb = fmaf(cf, h[i], rb);
                    }
                    const float cf0 = fmaf(n, (p1[0] - p0[0]) * inv_hop, p0[0]);
                    const float v = fmaf(-cf0, h[0], -(ra + rb));
                    for (int i = M - 1; i > 0; --i) h[i] = h[i - 1];
                    h[0] = v;
                }
                for (int i = 0; i < M; ++i) Phi[((size_t)c * M + i) * M + j] = h[i];
            } else {
                double h[64];
                for (int i = 0; i < M; ++i) h[i] = i == j ? 1.0 : 0.0;
                const double inv_hop = 1.0 / (double)hop;
                for (int t = c * L; t < (c + 1) * L; ++t) {
                    const int f = frame_of(t, hop, F);
                    const double n = (double)(t - f * hop);
                    const float* p0 = a + (size_t)f * M;
                    const float* p1 = p0 + M;
                    double acc = 0.0;
                    for (int i = M - 1; i >= 0; --i)
                        acc = fma(fma(n, ((double)p1[i] - (double)p0[i]) * inv_hop, (double)p0[i]), h[i], acc);
                    for (int i = M - 1; i > 0; --i) h[i] = h[i - 1];
                    h[0] = -acc;
                }
                for (int i = 0; i < M; ++i) Phi[((size_t)c * M + i) * M + j] = h[i];
            }
        }
}
