/*
 * zernike_oracle.c -- plain-C restatement of the two transform definitions.
 * TEST INFRASTRUCTURE ONLY (see oracle/zernike_oracle.py for the policy): built by
 * __graft_entry__.build() into oracle/_build/libzernike_oracle.so and loaded only by tests/,
 * smoke() and bench.py's cpu_baseline leg -- never by the product package.
 *
 * Parity status: PINNED through tests/test_oracle_golden.py (compared with the golden vectors
 * captured from the reference and with the NumPy restatement).
 *
 * The arithmetic is the definition both reference paths approximate with library calls:
 *   batch  (mtflearn/features/_zps.py:146-157): Z[p][j] = sum_t patch[p][t] * V[j][t] / (pi K^2 / 4)
 *   dense  (mtflearn/features/_zps.py:159-193): Z[j][i][k] = sum_{r,c} pad(img)[i-ea+r][k-ea+c] * V[j][r][c] / area
 *          with eb = (K-1)/2, ea = K-1-eb  (the alignment of fftconvolve(mode='same') * (-1)^n)
 * Sums run in row-major pixel order in double precision.
 */
#include <stddef.h>
#include <stdint.h>

static const double ZK_PI = 3.14159265358979323846;

/* patches: (n_patches, K, K) float32 or float64 (is_f64), basis: (n_poly, K, K), out: (n_patches, n_poly) */
void zko_patches(const void *patches, int is_f64, int64_t n_patches, int K, int n_poly,
                 const double *basis, double *out)
{
    const double area = ZK_PI * (double)K * (double)K / 4.0;
    const int64_t kk = (int64_t)K * K;
    for (int64_t p = 0; p < n_patches; ++p) {
        for (int j = 0; j < n_poly; ++j) {
            const double *v = basis + (size_t)j * kk;
            double acc = 0.0;
            if (is_f64) {
                const double *f = (const double *)patches + p * kk;
                for (int64_t t = 0; t < kk; ++t) acc += f[t] * v[t];
            } else {
                const float *f = (const float *)patches + p * kk;
                for (int64_t t = 0; t < kk; ++t) acc += (double)f[t] * v[t];
            }
            out[p * n_poly + j] = acc / area;
        }
    }
}

/* image: (H, W); out: (n_poly, n_rows, W) for output rows [row0, row0 + n_rows) */
void zko_frame(const void *image, int is_f64, int64_t H, int64_t W, int K, int n_poly,
               const double *basis, int64_t row0, int64_t n_rows, double *out)
{
    const double area = ZK_PI * (double)K * (double)K / 4.0;
    const int eb = (K - 1) / 2, ea = K - 1 - eb;
    for (int j = 0; j < n_poly; ++j) {
        const double *v = basis + (size_t)j * K * K;
        for (int64_t i = row0; i < row0 + n_rows; ++i) {
            for (int64_t k = 0; k < W; ++k) {
                double acc = 0.0;
                for (int r = 0; r < K; ++r) {
                    const int64_t ii = i - ea + r;
                    if (ii < 0 || ii >= H) continue;
                    for (int c = 0; c < K; ++c) {
                        const int64_t kc = k - ea + c;
                        if (kc < 0 || kc >= W) continue;
                        const double f = is_f64 ? ((const double *)image)[ii * W + kc]
                                                : (double)((const float *)image)[ii * W + kc];
                        acc += f * v[r * K + c];
                    }
                }
                out[((size_t)j * n_rows + (i - row0)) * W + k] = acc / area;
            }
        }
    }
}
