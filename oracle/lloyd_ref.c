/*
 * lloyd_ref.c -- CPU oracle for sklearn.cluster.KMeans(n_clusters=k, init=<k x d array>, n_init=1)
 * .fit(X) / .predict(X) as the reference calls it (k-means-color-clustering/color_kmeans.py:66-78,
 * KmeanGrids.py:300-304).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_ref.h).
 *
 * The arithmetic lives in scikit-learn (unpinned by the reference; 1.7.2 installed in this
 * image).  Restated from sklearn/cluster/_kmeans.py:279-287 (_tolerance), :624-752
 * (_kmeans_single_lloyd), :1427-1554 (fit: cast, centring, un-centring), :1066-1098 (predict),
 * _k_means_lloyd.pyx:23-218 (lloyd_iter_chunked_dense / _update_chunk_dense) and
 * _k_means_common.pyx:16-43 (_euclidean_dense_dense), :167-211 (_relocate_empty_clusters_dense),
 * :274-311 (_average_centers, _center_shift).  All math in float64 (the reference's data is uint8,
 * which sklearn casts to float64, _kmeans.py:1454-1462).  Single-threaded, samples visited in index
 * order.  Pinned by goldens generated with the installed sklearn
 * (tests/golden/make_lloyd_goldens.py) and, for k=1, by the reference's recorded CSV (KAT-B).
 *
 * Distance kernel: D[i][j] = |c_j|^2 - 2 x_i.c_j with the dot product as a sequential FMA chain
 * (what a BLAS dgemm micro-kernel does for an inner dimension of 2..4) -- fma() is explicit here,
 * everything else is built with -ffp-contract=off.
 */
#include "oracle_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double xval(const void *X, int dtype, int64_t idx)
{
    switch (dtype) {
    case 0: return (double)((const uint8_t *)X)[idx];
    case 1: return (double)((const float *)X)[idx];
    default: return ((const double *)X)[idx];
    }
}

/* numpy's pairwise summation for n <= 128 (np.mean / np.sum over a short contiguous vector) */
static double np_sum_small(const double *a, int n)
{
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* _k_means_common.pyx:16-43 */
static double sq_euclid(const double *a, const double *b, int d)
{
    int n = d / 4, rem = d % 4;
    double result = 0;
    for (int i = 0; i < n; i++) {
        result += ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) +
                   (a[2] - b[2]) * (a[2] - b[2]) + (a[3] - b[3]) * (a[3] - b[3]));
        a += 4; b += 4;
    }
    for (int i = 0; i < rem; i++) result += (a[i] - b[i]) * (a[i] - b[i]);
    return result;
}

static double dot_fma(const double *a, const double *b, int d)
{
    double acc = a[0] * b[0];
    for (int i = 1; i < d; i++) acc = fma(a[i], b[i], acc);
    return acc;
}

/* E-step for one sample against k centres; first strict minimum wins */
static int assign_one(const double *x, const double *c, const double *cn, int k, int d)
{
    double best = cn[0] - 2.0 * dot_fma(x, c, d);
    int label = 0;
    for (int j = 1; j < k; j++) {
        double dj = cn[j] - 2.0 * dot_fma(x, c + (size_t)j * d, d);
        if (dj < best) { best = dj; label = j; }
    }
    return label;
}

/* one Lloyd iteration on centred data Xc (N x d doubles) */
static void lloyd_iter(const double *Xc, int64_t N, int d, int k, const double *c_old,
                       double *c_new, double *w, int32_t *labels, double *shift,
                       int update_centers)
{
    double *cn = (double *)malloc(sizeof(double) * k);
    for (int j = 0; j < k; j++) cn[j] = dot_fma(c_old + (size_t)j * d, c_old + (size_t)j * d, d);
    if (update_centers) {
        memset(c_new, 0, sizeof(double) * k * d);
        memset(w, 0, sizeof(double) * k);
    }
    for (int64_t i = 0; i < N; i++) {
        const double *x = Xc + (size_t)i * d;
        int l = assign_one(x, c_old, cn, k, d);
        labels[i] = l;
        if (update_centers) {
            w[l] += 1.0;
            for (int f = 0; f < d; f++) c_new[(size_t)l * d + f] += x[f];
        }
    }
    free(cn);
    if (!update_centers) return;

    /* _relocate_empty_clusters_dense */
    int n_empty = 0;
    for (int j = 0; j < k; j++) n_empty += (w[j] == 0);
    if (n_empty > 0) {
        double *dist = (double *)malloc(sizeof(double) * N);
        double dmax = 0;
        for (int64_t i = 0; i < N; i++) {
            const double *x = Xc + (size_t)i * d, *c = c_old + (size_t)labels[i] * d;
            double s = 0;
            for (int f = 0; f < d; f++) s += (x[f] - c[f]) * (x[f] - c[f]);
            dist[i] = s;
            if (s > dmax) dmax = s;
        }
        if (dmax != 0) {
            /* the n_empty farthest samples, farthest first; ties -> lowest index */
            int e = 0;
            for (int j = 0; j < k && e < n_empty; j++) {
                if (w[j] != 0) continue;
                int64_t far = -1;
                double best = -1;
                for (int64_t i = 0; i < N; i++)
                    if (dist[i] > best) { best = dist[i]; far = i; }
                dist[far] = -2;               /* taken */
                int old = labels[far];
                const double *x = Xc + (size_t)far * d;
                for (int f = 0; f < d; f++) {
                    c_new[(size_t)old * d + f] -= x[f];
                    c_new[(size_t)j * d + f] = x[f];
                }
                w[j] = 1.0;
                w[old] -= 1.0;
                e++;
            }
        }
        free(dist);
    }
    /* _average_centers */
    int amax = 0;
    for (int j = 1; j < k; j++) if (w[j] > w[amax]) amax = j;
    for (int j = 0; j < k; j++) {
        if (w[j] > 0) {
            double alpha = 1.0 / w[j];
            for (int f = 0; f < d; f++) c_new[(size_t)j * d + f] *= alpha;
        } else {
            for (int f = 0; f < d; f++) c_new[(size_t)j * d + f] = c_new[(size_t)amax * d + f];
        }
    }
    /* _center_shift */
    for (int j = 0; j < k; j++)
        shift[j] = sqrt(sq_euclid(c_new + (size_t)j * d, c_old + (size_t)j * d, d));
}

int ofc_ref_kmeans_fit(const void *X, int dtype, int64_t N, int d, int k, const double *init,
                       int max_iter, double tol_rel, double *centers, int32_t *labels,
                       double *inertia, int *n_iter)
{
    if (N < k || k < 1 || d < 1 || k > 256) return -2;   /* ValueError, _kmeans.py:870-873 */
    double *Xc = (double *)malloc(sizeof(double) * (size_t)N * d);
    double *mean = (double *)calloc(d, sizeof(double)), *var = (double *)calloc(d, sizeof(double));
    for (int64_t i = 0; i < N; i++)
        for (int f = 0; f < d; f++) {
            double v = xval(X, dtype, i * d + f);
            Xc[(size_t)i * d + f] = v;
            mean[f] += v;
        }
    for (int f = 0; f < d; f++) mean[f] /= (double)N;
    /* _tolerance: mean(np.var(X, axis=0)) * tol, on the un-centred data */
    for (int64_t i = 0; i < N; i++)
        for (int f = 0; f < d; f++) {
            double t = Xc[(size_t)i * d + f] - mean[f];
            var[f] += t * t;
        }
    for (int f = 0; f < d; f++) var[f] /= (double)N;
    double tol = (tol_rel == 0) ? 0 : np_sum_small(var, d) / (double)d * tol_rel;
    /* centring */
    for (int64_t i = 0; i < N; i++)
        for (int f = 0; f < d; f++) Xc[(size_t)i * d + f] -= mean[f];
    double *c = (double *)malloc(sizeof(double) * k * d), *cnew = (double *)malloc(sizeof(double) * k * d);
    double *w = (double *)malloc(sizeof(double) * k), *shift = (double *)malloc(sizeof(double) * k);
    double *sh2 = (double *)malloc(sizeof(double) * k);
    int32_t *labels_old = (int32_t *)malloc(sizeof(int32_t) * N);
    for (int j = 0; j < k * d; j++) c[j] = init[j] - mean[j % d];
    for (int64_t i = 0; i < N; i++) { labels[i] = -1; labels_old[i] = -1; }
    int strict = 0, it = 0;
    for (it = 0; it < max_iter; it++) {
        lloyd_iter(Xc, N, d, k, c, cnew, w, labels, shift, 1);
        double *t = c; c = cnew; cnew = t;
        if (memcmp(labels, labels_old, sizeof(int32_t) * N) == 0) { strict = 1; break; }
        for (int j = 0; j < k; j++) sh2[j] = shift[j] * shift[j];
        if (np_sum_small(sh2, k) <= tol) break;
        memcpy(labels_old, labels, sizeof(int32_t) * N);
    }
    if (it == max_iter) it = max_iter - 1;      /* loop exhausted: n_iter_ = max_iter */
    if (!strict) lloyd_iter(Xc, N, d, k, c, c, w, labels, shift, 0);
    double in = 0;
    for (int64_t i = 0; i < N; i++)
        in += sq_euclid(Xc + (size_t)i * d, c + (size_t)labels[i] * d, d);
    if (inertia) *inertia = in;
    if (n_iter) *n_iter = it + 1;
    for (int j = 0; j < k * d; j++) centers[j] = c[j] + mean[j % d];
    free(Xc); free(mean); free(var); free(c); free(cnew); free(w); free(shift); free(sh2);
    free(labels_old);
    return 0;
}

/* KMeans.predict: one E-step of the un-centred X against the un-centred centres */
int ofc_ref_kmeans_predict(const void *X, int dtype, int64_t N, int d, int k,
                           const double *centers, int32_t *labels)
{
    if (k < 1 || d < 1 || d > 64) return -2;
    double *cn = (double *)malloc(sizeof(double) * k), x[64];
    for (int j = 0; j < k; j++) cn[j] = dot_fma(centers + (size_t)j * d, centers + (size_t)j * d, d);
    for (int64_t i = 0; i < N; i++) {
        for (int f = 0; f < d; f++) x[f] = xval(X, dtype, i * d + f);
        labels[i] = assign_one(x, centers, cn, k, d);
    }
    free(cn);
    return 0;
}

/* shard-level E+M accumulation used by the multi-rank tests: labels in/out (pass -1 first),
 * out = [sum (x-mean) per cluster: k*d][count: k][n_changed: 1] */
int ofc_ref_lloyd_partials(const void *X, int dtype, int64_t N, int d, int k, const double *mean,
                           const double *centers_c, int32_t *labels, double *out)
{
    if (d > 64) return -2;
    double *cn = (double *)malloc(sizeof(double) * k), x[64];
    for (int j = 0; j < k; j++)
        cn[j] = dot_fma(centers_c + (size_t)j * d, centers_c + (size_t)j * d, d);
    memset(out, 0, sizeof(double) * (k * d + k + 1));
    for (int64_t i = 0; i < N; i++) {
        for (int f = 0; f < d; f++) x[f] = xval(X, dtype, i * d + f) - mean[f];
        int l = assign_one(x, centers_c, cn, k, d);
        if (l != labels[i]) out[k * d + k] += 1.0;
        labels[i] = l;
        out[k * d + l] += 1.0;
        for (int f = 0; f < d; f++) out[(size_t)l * d + f] += x[f];
    }
    free(cn);
    return 0;
}
