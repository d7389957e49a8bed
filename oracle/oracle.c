/*
 * oracle.c -- plain-C restatement of the radar-ml hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this;
 * the product (radar-ml_amd/) never does.  It restates, loop for loop, what the reference and
 * its scikit-learn/libsvm dependency compute on the CPU, so that it can (a) check the HIP path
 * at sizes NumPy would be too slow for and (b) be timed as the CPU "port" baseline.
 *
 * Citations: paths relative to the reference tree; "sk:" = scikit-learn (pinned 0.24.0 in
 * requirements.txt:57; the libsvm sources quoted are those of the installed 1.7.2).
 * Pinned against the golden vectors in tests/golden/ by tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Max-projection named by BASELINE.json: xz=max_j V, yz=max_i V, xy=max_k V,
 * tuple order (xz, yz, xy) of common.py:40; NumPy-equivalent np.max(V, axis). */
void oracle_project_max(const float* V, int64_t B, int X, int Y, int Z, float* xz, float* yz, float* xy, int threads) {
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* v = V + b * (int64_t)X * Y * Z;
        float* pxz = xz + b * (int64_t)X * Z;
        float* pyz = yz + b * (int64_t)Y * Z;
        float* pxy = xy + b * (int64_t)X * Y;
        for (int64_t t = 0; t < (int64_t)X * Z; ++t) pxz[t] = -INFINITY;
        for (int64_t t = 0; t < (int64_t)Y * Z; ++t) pyz[t] = -INFINITY;
        for (int64_t t = 0; t < (int64_t)X * Y; ++t) pxy[t] = -INFINITY;
        for (int i = 0; i < X; ++i)
            for (int j = 0; j < Y; ++j) {
                const float* row = v + ((int64_t)i * Y + j) * Z;
                float m = -INFINITY;
                for (int k = 0; k < Z; ++k) {
                    float x = row[k];
                    if (x > pxz[i * Z + k]) pxz[i * Z + k] = x;
                    if (x > pyz[j * Z + k]) pyz[j * Z + k] = x;
                    if (x > m) m = x;
                }
                pxy[i * Y + j] = m;
            }
    }
}

/* Plane slices through (i,j,k): predict.py:102-107 / ground_truth_samples.py:413-419
 * (yz=V[i,:,:], xz=V[:,j,:], xy=V[:,:,k]); Python negative-index wrap. */
void oracle_project_slice(const float* V, int64_t B, int X, int Y, int Z, const int32_t* ijk, float* xz, float* yz, float* xy) {
    for (int64_t b = 0; b < B; ++b) {
        const float* v = V + b * (int64_t)X * Y * Z;
        int i = ijk[b * 3], j = ijk[b * 3 + 1], k = ijk[b * 3 + 2];
        if (i < 0) i += X;
        if (j < 0) j += Y;
        if (k < 0) k += Z;
        for (int ii = 0; ii < X; ++ii)
            for (int kk = 0; kk < Z; ++kk) xz[(b * X + ii) * Z + kk] = v[((int64_t)ii * Y + j) * Z + kk];
        for (int jj = 0; jj < Y; ++jj)
            for (int kk = 0; kk < Z; ++kk) yz[(b * Y + jj) * Z + kk] = v[((int64_t)i * Y + jj) * Z + kk];
        for (int ii = 0; ii < X; ++ii)
            for (int jj = 0; jj < Y; ++jj) xy[(b * X + ii) * Y + jj] = v[((int64_t)ii * Y + jj) * Z + k];
    }
}

/* common.process_samples at zoom 1 (common.py:141-148): ravel + concatenate the selected planes
 * in (xz,yz,xy) order, optional float32 "/ RADAR_MAX". */
void oracle_features(const float* xz, const float* yz, const float* xy, int64_t B, int X, int Y, int Z,
                     int mask, int scale, float* feat) {
    int64_t D = ((mask & 1) ? (int64_t)X * Z : 0) + ((mask & 2) ? (int64_t)Y * Z : 0) + ((mask & 4) ? (int64_t)X * Y : 0);
    for (int64_t b = 0; b < B; ++b) {
        float* f = feat + b * D;
        int64_t o = 0;
        if (mask & 1) { memcpy(f + o, xz + b * (int64_t)X * Z, sizeof(float) * X * Z); o += (int64_t)X * Z; }
        if (mask & 2) { memcpy(f + o, yz + b * (int64_t)Y * Z, sizeof(float) * Y * Z); o += (int64_t)Y * Z; }
        if (mask & 4) { memcpy(f + o, xy + b * (int64_t)X * Y, sizeof(float) * X * Y); o += (int64_t)X * Y; }
        if (scale) for (int64_t t = 0; t < D; ++t) f[t] = f[t] / 255.0f;
    }
}

static double expit_d(double x) {
    if (x >= 0.0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}

/*
 * C-SVC decision function + votes + ovr + sigmoid calibration for N float32 rows.
 *   sk:svm/_base.py:610-620            X is cast to float64
 *   sk:svm/src/libsvm/svm.cpp:461-475,514   RBF: m = x - sv; sum = dot(m, m); exp(-gamma*sum)
 *   sk:svm/src/libsvm/svm.cpp:457           linear: dot(x, sv)
 *   sk:svm/src/libsvm/svm.cpp:2847-2894     kvalue[], pair loop (coef1 = sv_coef[j-1] on class i,
 *                                           coef2 = sv_coef[i] on class j), sum -= rho, votes, first max
 *   sk:utils/multiclass.py:542-584          ovr = votes + s / (3 (|s| + 1))
 *   sk:calibration.py:727-784,928-942       expit(-(a T + b)), normalise, clip (1, 1+1e-5] -> 1, argmax
 * Outputs may be NULL.  kernel: 0 rbf, 1 linear.  threads parallelises over rows only (each row is the
 * reference's serial loop).
 */
void oracle_svm(const float* Xf, int64_t N, int64_t D, const double* sv, int64_t M,
                const double* dual_coef, const double* intercept, const int32_t* n_support, int C,
                int kernel, double gamma, const double* calib_a, const double* calib_b,
                double* dec_ovo, double* dec_ovr, double* proba, int32_t* label_vote, int32_t* label_calib,
                int threads) {
    const int P = C * (C - 1) / 2;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        double* x = (double*)malloc(sizeof(double) * D);
        double* kvalue = (double*)malloc(sizeof(double) * M);
        double* dec = (double*)malloc(sizeof(double) * (P > 0 ? P : 1));
        int* start = (int*)malloc(sizeof(int) * C);
        int* vote = (int*)malloc(sizeof(int) * C);
        double* T = (double*)malloc(sizeof(double) * C);
        double* soc = (double*)malloc(sizeof(double) * C);
        double* pr = (double*)malloc(sizeof(double) * C);
#pragma omp for schedule(dynamic, 4)
        for (int64_t n = 0; n < N; ++n) {
            for (int64_t d = 0; d < D; ++d) x[d] = (double)Xf[n * D + d];
            for (int64_t m = 0; m < M; ++m) {
                const double* s = sv + m * D;
                double sum = 0.0;
                if (kernel == 0) {
                    for (int64_t d = 0; d < D; ++d) { double t = x[d] - s[d]; sum += t * t; }
                    kvalue[m] = exp(-gamma * sum);
                } else {
                    for (int64_t d = 0; d < D; ++d) sum += x[d] * s[d];
                    kvalue[m] = sum;
                }
            }
            start[0] = 0;
            for (int i = 1; i < C; ++i) start[i] = start[i - 1] + n_support[i - 1];
            for (int i = 0; i < C; ++i) vote[i] = 0;
            int p = 0;
            for (int i = 0; i < C; ++i)
                for (int j = i + 1; j < C; ++j) {
                    double sum = 0.0;
                    int si = start[i], sj = start[j], ci = n_support[i], cj = n_support[j];
                    const double* coef1 = dual_coef + (int64_t)(j - 1) * M;
                    const double* coef2 = dual_coef + (int64_t)i * M;
                    for (int k = 0; k < ci; ++k) sum += coef1[si + k] * kvalue[si + k];
                    for (int k = 0; k < cj; ++k) sum += coef2[sj + k] * kvalue[sj + k];
                    sum += intercept[p];              /* sum -= rho[p], rho = -intercept_ */
                    dec[p] = sum;
                    if (dec[p] > 0) ++vote[i]; else ++vote[j];
                    ++p;
                }
            int best = 0;
            for (int i = 1; i < C; ++i) if (vote[i] > vote[best]) best = i;
            if (dec_ovo) for (int q = 0; q < P; ++q) dec_ovo[n * P + q] = dec[q];
            if (label_vote) label_vote[n] = best;
            if (C == 2) {
                T[0] = -dec[0];                       /* sk:svm/_base.py:546-547 */
                if (dec_ovr) dec_ovr[n] = T[0];
            } else {
                for (int c = 0; c < C; ++c) { soc[c] = 0.0; T[c] = 0.0; }
                p = 0;
                for (int i = 0; i < C; ++i)
                    for (int j = i + 1; j < C; ++j) {
                        double conf = -dec[p];
                        soc[i] -= conf; soc[j] += conf;
                        if (dec[p] < 0) T[j] += 1.0; else T[i] += 1.0;
                        ++p;
                    }
                for (int c = 0; c < C; ++c) T[c] = T[c] + soc[c] / (3.0 * (fabs(soc[c]) + 1.0));
                if (dec_ovr) for (int c = 0; c < C; ++c) dec_ovr[n * C + c] = T[c];
            }
            if (calib_a && (proba || label_calib)) {
                if (C == 2) {
                    pr[1] = expit_d(-(calib_a[0] * T[0] + calib_b[0]));
                    pr[0] = 1.0 - pr[1];
                } else {
                    double den = 0.0;
                    for (int c = 0; c < C; ++c) { pr[c] = expit_d(-(calib_a[c] * T[c] + calib_b[c])); den += pr[c]; }
                    for (int c = 0; c < C; ++c) pr[c] = den != 0.0 ? pr[c] / den : 1.0 / C;
                }
                for (int c = 0; c < C; ++c) if (pr[c] > 1.0 && pr[c] <= 1.0 + 1e-5) pr[c] = 1.0;
                int bc = 0;
                for (int c = 1; c < C; ++c) if (pr[c] > pr[bc]) bc = c;
                if (proba) for (int c = 0; c < C; ++c) proba[n * C + c] = pr[c];
                if (label_calib) label_calib[n] = bc;
            }
        }
        free(x); free(kvalue); free(dec); free(start); free(vote); free(T); free(soc); free(pr);
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
