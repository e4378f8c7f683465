/*
 * C restatement of the reference algorithm for the GOLF-ss / GOLF-ff end filters
 * (TEST INFRASTRUCTURE — never linked or called by the product package golf_amd).
 *
 * Used (a) as a fast float64 checker at full BASELINE sizes and (b) as the timed "port" CPU baseline in
 * bench.py: it keeps the *structure* of the reference's CPU path —
 *   1. materialise sample-rate coefficients by linear interpolation
 *      (AudioTensor.reduce_hop_length -> F.interpolate, reference models/utils.py:171-191,538-544;
 *       models/filters.py:107-109),
 *   2. fp32 recursion with sequential taps, parallel over batch items only
 *      (torchlpc.sample_wise_lpc, third-party, call site models/filters.py:112; its CPU path is a
 *       numba prange over the batch — SURVEY.md §2.1 [recollection]).
 * Parity status: the third-party kernels themselves are not in /root/reference => "parity unpinned"
 * for them; this file is pinned against oracle/golf_oracle.py (itself pinned on tests/golden).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DEF_UPSAMPLE(NAME, TY)                                                                       \
    /* z (B,F,C) -> out (B,T,C), T <= (F-1)*hop+1 */                                                 \
    void NAME(const TY* z, TY* out, int B, int F, int C, int hop, int T) {                           \
        _Pragma("omp parallel for schedule(static)") for (int b = 0; b < B; ++b) {                   \
            for (int t = 0; t < T; ++t) {                                                            \
                int f = F >= 2 ? t / hop : 0;                                                        \
                if (F >= 2 && f > F - 2) f = F - 2;                                                  \
                const TY w = F >= 2 ? (TY)(t - f * hop) / (TY)hop : (TY)0;                           \
                const TY* z0 = z + ((size_t)b * F + f) * C;                                          \
                const TY* z1 = F >= 2 ? z0 + C : z0;                                                 \
                TY* o = out + ((size_t)b * T + t) * C;                                               \
                for (int c = 0; c < C; ++c) o[c] = z0[c] * ((TY)1 - w) + z1[c] * w;                  \
            }                                                                                        \
        }                                                                                            \
    }
DEF_UPSAMPLE(golf_oracle_upsample_f32, float)
DEF_UPSAMPLE(golf_oracle_upsample_f64, double)

#define DEF_SWLPC(NAME, TY)                                                                          \
    /* y[b,t] = x[b,t] - sum_i A[b,t,i]*y[b,t-1-i]; x,y (B,T) contiguous, A (B,T,M) */               \
    void NAME(const TY* x, const TY* A, TY* y, int B, int T, int M) {                                \
        _Pragma("omp parallel for schedule(static)") for (int b = 0; b < B; ++b) {                   \
            const TY* xb = x + (size_t)b * T;                                                        \
            const TY* Ab = A + (size_t)b * T * M;                                                    \
            TY* yb = y + (size_t)b * T;                                                              \
            for (int t = 0; t < T; ++t) {                                                            \
                TY acc = xb[t];                                                                      \
                const TY* at = Ab + (size_t)t * M;                                                   \
                const int mm = t < M ? t : M;                                                        \
                for (int i = 0; i < mm; ++i) acc -= at[i] * yb[t - 1 - i];                           \
                yb[t] = acc;                                                                         \
            }                                                                                        \
        }                                                                                            \
    }
DEF_SWLPC(golf_oracle_sample_wise_lpc_f32, float)
DEF_SWLPC(golf_oracle_sample_wise_lpc_f64, double)

#define DEF_SS(NAME, TY, UPS, SW)                                                                    \
    /* LTVMinimumPhaseFilterPrecise.forward restated: scratch must hold B*T*(M+2) elements */        \
    void NAME(const TY* ex, int64_t ex_stride, const TY* gain, const TY* a, TY* y, int B, int T,     \
              int F, int M, int hop, TY* scratch) {                                                  \
        TY* A = scratch;                                                                             \
        TY* G = A + (size_t)B * T * M;                                                               \
        TY* x = G + (size_t)B * T;                                                                   \
        UPS(a, A, B, F, M, hop, T);                                                                  \
        UPS(gain, G, B, F, 1, hop, T);                                                               \
        _Pragma("omp parallel for schedule(static)") for (int b = 0; b < B; ++b)                     \
            for (int t = 0; t < T; ++t) x[(size_t)b * T + t] = ex[(size_t)b * ex_stride + t] * G[(size_t)b * T + t]; \
        SW(x, A, y, B, T, M);                                                                        \
    }
DEF_SS(golf_oracle_ltv_ss_f32, float, golf_oracle_upsample_f32, golf_oracle_sample_wise_lpc_f32)
DEF_SS(golf_oracle_ltv_ss_f64, double, golf_oracle_upsample_f64, golf_oracle_sample_wise_lpc_f64)

/* Closed-form backward in float64: checker for full-size gradients.
 * gy,y (B,T) contiguous; ex (B,>=T) with stride; outputs g_ex (B,T), g_gain (B,F), g_a (B,F,M).
 * scratch: B*T*(M+2) doubles. */
void golf_oracle_ltv_ss_bwd_f64(const double* gy, const double* y, const double* ex, int64_t ex_stride,
                                const double* gain, const double* a, double* g_ex, double* g_gain, double* g_a,
                                int B, int T, int F, int M, int hop, double* scratch) {
    double* A = scratch;
    double* G = A + (size_t)B * T * M;
    double* g = G + (size_t)B * T;
    golf_oracle_upsample_f64(a, A, B, F, M, hop, T);
    golf_oracle_upsample_f64(gain, G, B, F, 1, hop, T);
    memset(g_gain, 0, sizeof(double) * (size_t)B * F);
    memset(g_a, 0, sizeof(double) * (size_t)B * F * M);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        const double* Ab = A + (size_t)b * T * M;
        double* gb = g + (size_t)b * T;
        const double* yb = y + (size_t)b * T;
        for (int t = T - 1; t >= 0; --t) {
            double acc = gy[(size_t)b * T + t];
            for (int i = 0; i < M && t + 1 + i < T; ++i) acc -= Ab[(size_t)(t + 1 + i) * M + i] * gb[t + 1 + i];
            gb[t] = acc;
        }
        for (int t = 0; t < T; ++t) {
            int f = F >= 2 ? t / hop : 0;
            if (F >= 2 && f > F - 2) f = F - 2;
            const double w = F >= 2 ? (double)(t - f * hop) / (double)hop : 0.0;
            const double e = ex[(size_t)b * ex_stride + t];
            g_ex[(size_t)b * T + t] = gb[t] * G[(size_t)b * T + t];
            g_gain[(size_t)b * F + f] += (1.0 - w) * gb[t] * e;
            if (F >= 2) g_gain[(size_t)b * F + f + 1] += w * gb[t] * e;
            for (int i = 0; i < M && t - 1 - i >= 0; ++i) {
                const double v = -gb[t] * yb[t - 1 - i];
                g_a[((size_t)b * F + f) * M + i] += (1.0 - w) * v;
                if (F >= 2) g_a[((size_t)b * F + f + 1) * M + i] += w * v;
            }
        }
    }
}

/* torchaudio.functional.lfilter(x, [1,a], [1,0..], clamp=False) per row (lpc_synthesis,
 * reference models/lpc.py:11-16): rows (R,W), a (R,M). */
#define DEF_LFILT(NAME, TY)                                                                          \
    void NAME(const TY* x, const TY* a, TY* y, int R, int W, int M) {                                \
        _Pragma("omp parallel for schedule(static)") for (int r = 0; r < R; ++r) {                   \
            const TY* xr = x + (size_t)r * W;                                                        \
            const TY* ar = a + (size_t)r * M;                                                        \
            TY* yr = y + (size_t)r * W;                                                              \
            for (int t = 0; t < W; ++t) {                                                            \
                TY acc = xr[t];                                                                      \
                const int mm = t < M ? t : M;                                                        \
                for (int i = 0; i < mm; ++i) acc -= ar[i] * yr[t - 1 - i];                           \
                yr[t] = acc;                                                                         \
            }                                                                                        \
        }                                                                                            \
    }
DEF_LFILT(golf_oracle_lfilter_rows_f32, float)
DEF_LFILT(golf_oracle_lfilter_rows_f64, double)

int golf_oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
