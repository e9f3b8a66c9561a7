// lloyd_kernels.hip -- hand-written gfx950 kernels for Lloyd's k-means in the two shapes the path needs
// (SURVEY.md 2.2 K11):
//   (B) streaming: one huge problem resident in HBM ((u,v) of a whole clip, f32 d=2; also u8/f64, d<=8),
//       10 B/point/iteration algorithmic (4d read + 1 label read + 1 label write);
//   (A) batched-small: hundreds of independent problems (grid cells, u8 RGBA), one work-group per
//       problem with the points resident in LDS for all iterations (lloyd_batched.hip).
// Arithmetic = the oracle's (oracle/lloyd_ref.c == sklearn): f64 everywhere, data centred by the column
// mean, D_j = |c_j|^2 - 2 x.c_j with the dot product as a sequential FMA chain, first strict minimum wins.
// Reductions are deterministic: per-lane partials -> wave shuffle tree -> LDS -> one record per
// work-group -> fixed-order finishing kernel.  No atomics anywhere.
#include "lloyd_common.h"

#include <type_traits>

namespace ofc {

// ------------------------------------------------------------------------------------------------
// loaders: 4 consecutive points per lane as doubles
// ------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void load4(const uint8_t *X, int64_t i, double (&x)[4][D])
{
    if constexpr (D == 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(X + i * 4);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int f = 0; f < 4; f++) x[p][f] = (double)((w[p] >> (8 * f)) & 0xffu);
    } else {
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int f = 0; f < D; f++) x[p][f] = (double)X[(i + p) * D + f];
    }
}
template <int D>
__device__ __forceinline__ void load4(const float *X, int64_t i, double (&x)[4][D])
{
    if constexpr (D == 2) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        // streamed once per iteration and far larger than any cache: non-temporal
        const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(X + i * 2));
        const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(X + i * 2 + 4));
        x[0][0] = a.x; x[0][1] = a.y; x[1][0] = a.z; x[1][1] = a.w;
        x[2][0] = b.x; x[2][1] = b.y; x[3][0] = b.z; x[3][1] = b.w;
    } else if constexpr (D == 4) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float4 a = *reinterpret_cast<const float4 *>(X + (i + p) * 4);
            x[p][0] = a.x; x[p][1] = a.y; x[p][2] = a.z; x[p][3] = a.w;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int f = 0; f < D; f++) x[p][f] = (double)X[(i + p) * D + f];
    }
}
template <int D>
__device__ __forceinline__ void load4(const double *X, int64_t i, double (&x)[4][D])
{
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int f = 0; f < D; f++) x[p][f] = X[(i + p) * D + f];
}
template <int D, class T>
__device__ __forceinline__ void load1(const T *X, int64_t i, double (&x)[D])
{
#pragma unroll
    for (int f = 0; f < D; f++) x[f] = (double)X[i * D + f];
}

// deterministic work-group sum of NV doubles per thread -> out[NV] (thread 0..NV-1 write)
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV], double *lds /*[4][NV]*/, double *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double a = v[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) lds[wave * NV + i] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int i = threadIdx.x;
        out[i] = ((lds[i] + lds[NV + i]) + lds[2 * NV + i]) + lds[3 * NV + i];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// column statistics: pass 0 = sum x, pass 1 = sum (x-mean)^2   -> partial[block][D]
// ------------------------------------------------------------------------------------------------
template <int D, class T>
__global__ __launch_bounds__(256) void k_colstats(const T *__restrict__ X, int64_t N,
                                                  const double *__restrict__ mean, int pass,
                                                  double *__restrict__ partial)
{
    __shared__ double lds[4 * D];
    double acc[D], m[D];
#pragma unroll
    for (int f = 0; f < D; f++) { acc[f] = 0; m[f] = pass ? mean[f] : 0.0; }
    const int64_t n4 = N / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        double x[4][D];
        load4<D>(X, q * 4, x);
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int f = 0; f < D; f++) {
                const double t = x[p][f] - m[f];
                acc[f] += pass ? t * t : t;
            }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(N - n4 * 4)) {
        double x[D];
        load1<D>(X, n4 * 4 + threadIdx.x, x);
#pragma unroll
        for (int f = 0; f < D; f++) {
            const double t = x[f] - m[f];
            acc[f] += pass ? t * t : t;
        }
    }
    block_reduce_store<D>(acc, lds, partial + (size_t)blockIdx.x * D);
}

// fixed-order sum of per-block records: out[i] = sum_b partial[b][i].  1024 threads: component i is
// summed by nstripes = 1024/nv threads over interleaved record subsets, then folded in stripe order.
__global__ __launch_bounds__(1024) void k_reduce_records(const double *__restrict__ partial, int nblocks,
                                                         int nv, double *__restrict__ out, const int *halt)
{
    __shared__ double lds[1024];
    if (halt && *halt) return;
    const int nstripes = 1024 / nv;
    const int i = threadIdx.x % nv, st = threadIdx.x / nv;
    double a = 0;
    if (st < nstripes) {
        // eight records in flight per thread (the loop used to walk ~40 dependent L2 round trips: 15 us per call, a tenth
        // of a pruned Lloyd iteration); partial sums combined in a fixed order
        double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int b = st;
        for (; b + 7 * nstripes < nblocks; b += 8 * nstripes) {
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = partial[(size_t)(b + j * nstripes) * nv + i];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] += t[j];
        }
        for (int j = 0; b < nblocks; b += nstripes, j++) q[j & 7] += partial[(size_t)b * nv + i];
        a = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    lds[threadIdx.x] = a;
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double r = 0;
        for (int q = 0; q < nstripes; q++) r += lds[q * nv + threadIdx.x];
        out[threadIdx.x] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// E-step (+ M-step accumulation).  Record per work-group: [k*d sums of (x-mean)][k counts][n_changed]
// (laid out with stride KMAX: sums[j*D+f], counts at KMAX*D + j, changed at KMAX*D + KMAX).
// ------------------------------------------------------------------------------------------------
template <int D, int KMAX>
__device__ __forceinline__ int assign_point(const double (&x)[D], const double *c /*[KMAX*D] regs*/,
                                            const double *cn, int k)
{
    double best = 0;
    int label = 0;
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        if (j < k) {
            double acc = x[0] * c[j * D];
#pragma unroll
            for (int f = 1; f < D; f++) acc = fma(x[f], c[j * D + f], acc);
            const double dj = cn[j] - 2.0 * acc;
            if (j == 0 || dj < best) { best = dj; label = j; }
        }
    }
    return label;
}

template <int D>
__device__ __forceinline__ double sq_euclid_grouped(const double (&a)[D], const double *b)
{
#pragma clang fp contract(off)
    double r = 0;
    int f = 0;
#pragma unroll
    for (; f + 4 <= D; f += 4)
        r += ((a[f] - b[f]) * (a[f] - b[f]) + (a[f + 1] - b[f + 1]) * (a[f + 1] - b[f + 1]) +
              (a[f + 2] - b[f + 2]) * (a[f + 2] - b[f + 2]) + (a[f + 3] - b[f + 3]) * (a[f + 3] - b[f + 3]));
#pragma unroll
    for (; f < D; f++) r += (a[f] - b[f]) * (a[f] - b[f]);
    return r;
}

// The M-step partials are NOT kept as k*(d+1) predicated register accumulators (8 x 3 conditional f64 adds per
// point made the kernel VALU-bound at 2.5 TB/s): every lane owns a private column of LDS accumulators
// [k*d sums f64][k counts u32] x 256 lanes and adds each point into the slot of its label with one ds_add per
// component.  A slot is only ever touched by its own lane, in point order, so the sums are deterministic; the
// 256 columns are folded in a fixed order at the end.  LDS: k*(8d+4)*256 B (25 KB for k=5, d=2).
// MODE 0: labels only (predict / final E-step); 1: labels + M-step partials (one Lloyd iteration; with `first` also
// sum (x-mean)^2 per column for sklearn's tol, so that pass costs no extra sweep); 2: labels + inertia in one sweep;
// 3: M-step partials only -- labels are neither read nor written (2 of the 10 B/point of a float2 stream), record's
// n_changed slot is 0: lloyd_fit_dev's loop does not need it while no cluster is empty (see there).
template <int D, int KMAX, class T, int MODE>
__global__ __launch_bounds__(256) void k_lloyd_assign(const T *__restrict__ X, int64_t N, int k_arg,
                                                      const LloydState *__restrict__ st,
                                                      uint8_t *__restrict__ labels,
                                                      double *__restrict__ partial, int first)
{
    // the launcher instantiates KMAX == k for k <= 8 (lloyd_kmax): a compile-time k removes the `j < k` masks from
    // the distance loop (3 of 10 VALU instructions per cluster and sample)
    const int k = KMAX <= 8 ? KMAX : k_arg;
    constexpr bool ACCUM = (MODE == 1 || MODE == 3), LABELS = (MODE != 3);
    if (ACCUM && st->halt) return;      // speculatively enqueued behind the iteration that converged (uniform)
    constexpr int NV = KMAX * D + KMAX + LLOYD_REC_EXTRA;    // [sums][counts][changed][sum (x-mean)^2 per column][tile counts: 0 here]
    extern __shared__ __align__(16) unsigned char smem[];
    double *sacc = reinterpret_cast<double *>(smem);                       // [k*D][256]
    unsigned *scnt = reinterpret_cast<unsigned *>(sacc + (size_t)k * D * 256);   // [k][256]
    __shared__ unsigned s_changed[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double c[KMAX * D], cn[KMAX], m[D];
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        cn[j] = (j < k) ? st->cn[j] : 0.0;
#pragma unroll
        for (int f = 0; f < D; f++) c[j * D + f] = (j < k) ? st->centers[j * D + f] : 0.0;
    }
#pragma unroll
    for (int f = 0; f < D; f++) m[f] = st->mean[f];
    if (ACCUM) {
        for (int i = 0; i < k * D; i++) sacc[i * 256 + tid] = 0.0;
        for (int j = 0; j < k; j++) scnt[j * 256 + tid] = 0u;
    }
    unsigned changed = 0;
    double sq[D], inert = 0;
#pragma unroll
    for (int f = 0; f < D; f++) sq[f] = 0;
    auto accumulate = [&](int l, const double (&x)[D]) {
#pragma unroll
        for (int f = 0; f < D; f++) sacc[(l * D + f) * 256 + tid] += x[f];
        scnt[l * 256 + tid] += 1u;
    };

    const int64_t n4 = N / 4;
    auto process = [&](int64_t q, double (&x)[4][D], unsigned lo) {
        const int old[4] = {(int)(lo & 255u), (int)((lo >> 8) & 255u), (int)((lo >> 16) & 255u), (int)(lo >> 24)};
        int nl[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int f = 0; f < D; f++) x[p][f] -= m[f];
            nl[p] = assign_point<D, KMAX>(x[p], c, cn, k);
        }
        if (ACCUM) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                accumulate(nl[p], x[p]);
                if (MODE == 1) changed += (nl[p] != old[p]);
            }
            if (first) {
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int f = 0; f < D; f++) sq[f] += x[p][f] * x[p][f];
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int p = 0; p < 4; p++) inert += sq_euclid_grouped<D>(x[p], st->centers + nl[p] * D);
        }
        if (LABELS)
            __builtin_nontemporal_store((unsigned)(nl[0] | (nl[1] << 8) | (nl[2] << 16) | (nl[3] << 24)),
                                        reinterpret_cast<unsigned *>(labels) + q);
    };
    const int64_t q0 = (int64_t)blockIdx.x * 256 + tid, qs = (int64_t)gridDim.x * 256;
    if constexpr (std::is_same<T, float>::value && D == 2) {
        // the (u,v) stream: the next quad's 32 bytes are requested before the current quad is processed, which doubles
        // the bytes each lane keeps in flight (1024 work-groups x 256 lanes x 32 B = 8 MB is less than what 5+ TB/s
        // times the memory latency needs)
        typedef float v4f __attribute__((ext_vector_type(4)));
        auto raw = [&](int64_t q, v4f &a, v4f &b, unsigned &lo) {
            a = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(X + q * 8));
            b = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(X + q * 8 + 4));
            lo = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(labels) + q) : 0u;
        };
        v4f na = {0, 0, 0, 0}, nb = {0, 0, 0, 0};
        unsigned nlo = 0;
        if (q0 < n4) raw(q0, na, nb, nlo);
        for (int64_t q = q0; q < n4; q += qs) {
            const v4f a = na, b = nb;
            const unsigned lo = nlo;
            if (q + qs < n4) raw(q + qs, na, nb, nlo);
            double x[4][D];
            x[0][0] = a.x; x[0][1] = a.y; x[1][0] = a.z; x[1][1] = a.w;
            x[2][0] = b.x; x[2][1] = b.y; x[3][0] = b.z; x[3][1] = b.w;
            process(q, x, lo);
        }
    } else {
        for (int64_t q = q0; q < n4; q += qs) {
            double x[4][D];
            load4<D>(X, q * 4, x);
            const unsigned lo = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(labels) + q) : 0u;
            process(q, x, lo);
        }
    }
    if (blockIdx.x == 0 && tid < (int)(N - n4 * 4)) {
        const int64_t i = n4 * 4 + tid;
        double x[D];
        load1<D>(X, i, x);
#pragma unroll
        for (int f = 0; f < D; f++) x[f] -= m[f];
        const int l = assign_point<D, KMAX>(x, c, cn, k);
        if (ACCUM) {
            accumulate(l, x);
            if (MODE == 1) changed += (l != labels[i]);
            if (first) {
#pragma unroll
                for (int f = 0; f < D; f++) sq[f] += x[f] * x[f];
            }
        }
        if (MODE == 2) inert += sq_euclid_grouped<D>(x, st->centers + l * D);
        if (LABELS) labels[i] = (uint8_t)l;
    }
    if (MODE == 2) {
        __shared__ double lds1[4];
        double a1[1] = {inert};
        block_reduce_store<1>(a1, lds1, partial + blockIdx.x);
    }
    if (ACCUM) {
        double *rec = partial + (size_t)blockIdx.x * NV;
        if (tid < NV) rec[tid] = 0.0;
        unsigned ch = changed;
        for (int off = 32; off >= 1; off >>= 1) ch += __shfl_down(ch, off, 64);
        if (lane == 0) s_changed[wave] = ch;
        __syncthreads();
        // fold the 256 private columns: component i by wave i % 4; lanes add their 4 columns in a fixed order,
        // then a fixed shuffle tree
        for (int i = wave; i < k * D + k; i += 4) {
            double a;
            if (i < k * D) {
                const double *col = sacc + (size_t)i * 256;
                a = ((col[lane] + col[lane + 64]) + col[lane + 128]) + col[lane + 192];
            } else {
                const unsigned *col = scnt + (size_t)(i - k * D) * 256;
                a = (double)(col[lane] + col[lane + 64] + col[lane + 128] + col[lane + 192]);
            }
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
            if (lane == 0) rec[i < k * D ? i : KMAX * D + (i - k * D)] = a;
        }
        if (tid == 0) rec[KMAX * D + KMAX] = (double)(s_changed[0] + s_changed[1] + s_changed[2] + s_changed[3]);
        if (first) {                                    // uniform
            __shared__ double lds_sq[4 * D];
            block_reduce_store<D>(sq, lds_sq, rec + KMAX * D + KMAX + 1);
        }
    }
}

// numpy's pairwise sum for n <= 128 (sequential below 8 elements, 8 lanes above)
__device__ inline double np_sum_small_dev(const double *a, int n)
{
    if (n < 8) {
        double r = 0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
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

// ------------------------------------------------------------------------------------------------
// M-step finish on one lane: (optional relocation is done by the host before this), average, shift,
// convergence inputs.  tot: [KMAX*D sums][KMAX counts][changed] (already all-reduced if distributed).
// Follows _average_centers / _center_shift (_k_means_common.pyx:274-311) incl. the in-place quirk for
// still-empty clusters.
// ------------------------------------------------------------------------------------------------
__global__ void k_lloyd_update(LloydState *st, const double *__restrict__ tot_g, int k, int d, int kmax,
                               int after_reloc, int labelled, int first, double n_total, double tol_rel,
                               LloydStatus *status /* pinned host */, int tiles)
{
    // one lane does the (tiny, strictly ordered) arithmetic; all 64 first stage its inputs in LDS so that it does not
    // walk through ~150 dependent global loads (8.6 -> 4 us per iteration, which matters on a 1/8 shard)
    constexpr int NVMAX = LLOYD_NVMAX;
    __shared__ double tot[NVMAX], cold[LLOYD_KMAX * LLOYD_DMAX], cnew[LLOYD_KMAX * LLOYD_DMAX];
    __shared__ int s_halt;
    const int NV = kmax * d + kmax + LLOYD_REC_EXTRA;
    for (int i = threadIdx.x; i < NV; i += blockDim.x) tot[i] = tot_g[i];
    for (int i = threadIdx.x; i < k * d; i += blockDim.x) cold[i] = st->centers[i];
    if (threadIdx.x == 0) s_halt = st->halt;
    __syncthreads();
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (s_halt && !after_reloc) return;             // no-op behind a converged / stalled iteration
    const double *w = tot + kmax * d;
    int amax = 0, n_empty = 0;
    for (int j = 0; j < k; j++) {
        if (w[j] > w[amax]) amax = j;
        n_empty += (w[j] == 0.0);
    }
    if (n_empty > 0 && !after_reloc) {
        // _relocate_empty_clusters_dense needs a pass over the data: hand back to the host, which
        // patches `tot` and calls again with after_reloc = 1.  Nothing is modified here except the halt flag.
        st->halt = 1;
        status->n_empty = n_empty;
        status->n_changed = tot[kmax * d + kmax];
        status->shift_tot = -1;
        status->converged = 0;
        status->strict = 0;
        for (int j = 0; j < k; j++) status->counts[j] = w[j];
        status->tiles_mode = -1;
        status->tiles_next = LLOYD_TILES_FULL;
        status->tiles_tested = status->tiles_pure = 0;
        __threadfence_system();
        status->valid = 1;
        return;
    }
    for (int j = 0; j < k; j++)
        for (int f = 0; f < d; f++) cnew[j * d + f] = tot[j * d + f];
    for (int j = 0; j < k; j++) {
        if (w[j] > 0) {
            const double alpha = 1.0 / w[j];
            for (int f = 0; f < d; f++) cnew[j * d + f] *= alpha;
        } else {
            for (int f = 0; f < d; f++) cnew[j * d + f] = cnew[amax * d + f];
        }
    }
    // shift (4-way grouped squared distance, then sqrt, squared again and summed numpy-style)
    double sh2[LLOYD_KMAX];
    for (int j = 0; j < k; j++) {
        const double *a = cnew + j * d, *b = cold + j * d;
        double r = 0;
        int f = 0;
        for (; f + 4 <= d; f += 4)
            r += ((a[f] - b[f]) * (a[f] - b[f]) + (a[f + 1] - b[f + 1]) * (a[f + 1] - b[f + 1]) +
                  (a[f + 2] - b[f + 2]) * (a[f + 2] - b[f + 2]) + (a[f + 3] - b[f + 3]) * (a[f + 3] - b[f + 3]));
        for (; f < d; f++) r += (a[f] - b[f]) * (a[f] - b[f]);
        const double s = sqrt(r);
        sh2[j] = s * s;
    }
    const double tot_shift = np_sum_small_dev(sh2, k);
    // swap: centers <- centers_new, new |c|^2
    for (int j = 0; j < k; j++) {
        double acc = cnew[j * d] * cnew[j * d];
        for (int f = 1; f < d; f++) acc = fma(cnew[j * d + f], cnew[j * d + f], acc);
        st->cn[j] = acc;
        for (int f = 0; f < d; f++) {
            st->centers[j * d + f] = cnew[j * d + f];
            st->centers_new[j * d + f] = cnew[j * d + f];
        }
    }
    // tol = mean(X.var(axis=0)) * tol_rel (_tolerance, _kmeans.py:279-287): the column sums of (x-mean)^2 ride along
    // with iteration 0's record
    double tol = st->tol;
    if (first) {
        double var[LLOYD_DMAX];
        for (int f = 0; f < d; f++) var[f] = tot[kmax * d + kmax + 1 + f] / n_total;
        tol = tol_rel != 0 ? np_sum_small_dev(var, d) / (double)d * tol_rel : 0.0;
        st->tol = tol;
    }
    const double n_changed = tot[kmax * d + kmax];
    const int strict = labelled && n_changed == 0.0;
    const int converged = strict || tot_shift <= tol;
    st->halt = converged;
    // ---- how the next tile sweep runs (lloyd_tiles.hip).  Every rank sees the same all-reduced counts, so every rank
    // takes the same decision.  A pruned sweep costs about (metadata 6 %) + 1.2 x (share of tiles it still walks by
    // sample) of a full one: it pays while about a third of the tiles pass the box test; a sweep that is switched off is
    // re-examined by a PROBE (box tests counted, nothing skipped) after 6, 12, 24, ... full sweeps.
    status->tiles_mode = -1;
    status->tiles_next = LLOYD_TILES_FULL;
    status->tiles_tested = status->tiles_pure = 0;
    if (tiles) {
        const double tested = tot[kmax * d + kmax + 1 + LLOYD_DMAX], pure = tot[kmax * d + kmax + 2 + LLOYD_DMAX];
        const double frac = tested > 0 ? pure / tested : 0.0;
        const int ran = st->prune_mode;
        int next = ran;
        if (st->prune_policy == LLOYD_PRUNE_ALWAYS) {
            next = LLOYD_TILES_PRUNED;
        } else if (ran == LLOYD_TILES_PRUNED) {
            if (frac < 0.3) {
                next = LLOYD_TILES_FULL;
                st->prune_cooldown = st->prune_backoff;
                st->prune_backoff *= 2;
            }
        } else if (ran == LLOYD_TILES_PROBE) {
            next = frac >= 0.45 ? LLOYD_TILES_PRUNED : LLOYD_TILES_FULL;
            st->prune_cooldown = st->prune_backoff;
            st->prune_backoff *= 2;
        } else {                          // a full sweep (policy OFF: k_tile_decide found the field incoherent, no metadata)
            if (st->prune_policy != LLOYD_PRUNE_OFF && --st->prune_cooldown <= 0) next = LLOYD_TILES_PROBE;
        }
        st->prune_mode = next;
        status->tiles_next = next;
        status->tiles_mode = ran;
        status->tiles_tested = tested;
        status->tiles_pure = pure;
    }
    status->n_changed = n_changed;
    for (int f = 0; f < d; f++) status->sqsum[f] = tot[kmax * d + kmax + 1 + f];
    status->shift_tot = tot_shift;
    status->tol = tol;
    status->n_empty = 0;
    status->converged = converged;
    status->strict = strict;
    for (int j = 0; j < k; j++) status->counts[j] = w[j];
    __threadfence_system();
    status->valid = 1;
}

// sets centres (centred) + |c|^2 from host-provided values
__global__ void k_lloyd_set_centers(LloydState *st, int k, int d, int prune_policy)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st->prune_policy = prune_policy;
    st->prune_mode = LLOYD_TILES_FULL;
    st->prune_cooldown = st->prune_backoff = 6;
    for (int j = 0; j < k; j++) {
        double acc = st->centers[j * d] * st->centers[j * d];
        for (int f = 1; f < d; f++) acc = fma(st->centers[j * d + f], st->centers[j * d + f], acc);
        st->cn[j] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// inertia = sum ||x - c_label||^2 (centred), _inertia_dense grouping; partial[block][1]
// and farthest-point search for empty-cluster relocation: partial[block] = {max dist, index}
// ------------------------------------------------------------------------------------------------
template <int D, class T>
__global__ __launch_bounds__(256) void k_lloyd_inertia(const T *__restrict__ X, int64_t N,
                                                       const LloydState *__restrict__ st,
                                                       const uint8_t *__restrict__ labels,
                                                       double *__restrict__ partial)
{
    __shared__ double lds[4];
    double acc[1] = {0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        double x[D];
        load1<D>(X, i, x);
#pragma unroll
        for (int f = 0; f < D; f++) x[f] -= st->mean[f];
        acc[0] += sq_euclid_grouped<D>(x, st->centers + (int)labels[i] * D);
    }
    block_reduce_store<1>(acc, lds, partial + blockIdx.x);
}

// distance of every sample to its (old) assigned centre; work-group argmax, ties -> lowest index;
// samples listed in excl[0..n_excl) are skipped.  out[block] = {dist, (double)index}
template <int D, class T>
__global__ __launch_bounds__(256) void k_lloyd_farthest(const T *__restrict__ X, int64_t N,
                                                        const LloydState *__restrict__ st,
                                                        const double *__restrict__ c_old,
                                                        const uint8_t *__restrict__ labels,
                                                        const int64_t *__restrict__ excl, int n_excl,
                                                        double *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double sv[256];
    __shared__ int64_t si[256];
    double best = -1;
    int64_t bi = -1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        bool skip = false;
        for (int e = 0; e < n_excl; e++) skip |= (excl[e] == i);
        if (skip) continue;
        double x[D];
        load1<D>(X, i, x);
        const double *c = c_old + (int)labels[i] * D;
        double s = 0;
#pragma unroll
        for (int f = 0; f < D; f++) {
            const double t = (x[f] - st->mean[f]) - c[f];
            s += t * t;
        }
        if (s > best) { best = s; bi = i; }   // i increases per thread: first max kept
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const int64_t oi = si[threadIdx.x + off];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi >= 0 && (si[threadIdx.x] < 0 || oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = sv[0];
        out[2 * blockIdx.x + 1] = (double)si[0];
    }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch over (dtype, D, KMAX)
// ------------------------------------------------------------------------------------------------
// the distance loop is unrolled over a compile-time K: exact for k <= 8, 16 above
int lloyd_kmax(int k) { return k <= 8 ? k : (k <= 16 ? 16 : 0); }

template <int D, int KMAX, class T>
static void launch_assign_t(const void *X, int64_t N, int k, const LloydState *st, uint8_t *labels,
                            double *partial, int nblocks, int mode, int first, hipStream_t s)
{
    if (mode == 1 || mode == 3) {
        const size_t lds = (size_t)k * (8 * D + 4) * 256;
        auto kern = mode == 1 ? &k_lloyd_assign<D, KMAX, T, 1> : &k_lloyd_assign<D, KMAX, T, 3>;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, s, (const T *)X, N, k, st, labels, partial, first);
    } else if (mode == 2) {
        hipLaunchKernelGGL((k_lloyd_assign<D, KMAX, T, 2>), dim3(nblocks), dim3(256), 0, s,
                           (const T *)X, N, k, st, labels, partial, 0);
    } else {
        hipLaunchKernelGGL((k_lloyd_assign<D, KMAX, T, 0>), dim3(nblocks), dim3(256), 0, s,
                           (const T *)X, N, k, st, labels, partial, 0);
    }
}

template <int D, int KMAX>
static int launch_assign_d(const void *X, int dtype, int64_t N, int k, const LloydState *st,
                           uint8_t *labels, double *partial, int nblocks, int mode, int first, hipStream_t s)
{
    switch (dtype) {
    case OFC_U8: launch_assign_t<D, KMAX, uint8_t>(X, N, k, st, labels, partial, nblocks, mode, first, s); break;
    case OFC_F32: launch_assign_t<D, KMAX, float>(X, N, k, st, labels, partial, nblocks, mode, first, s); break;
    case OFC_F64: launch_assign_t<D, KMAX, double>(X, N, k, st, labels, partial, nblocks, mode, first, s); break;
    default: set_error("bad dtype %d", dtype); return OFC_EINVAL;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

#define OFC_D_SWITCH(d, ...)                                     \
    switch (d) {                                                 \
    case 1: { constexpr int DD = 1; __VA_ARGS__; } break;        \
    case 2: { constexpr int DD = 2; __VA_ARGS__; } break;        \
    case 3: { constexpr int DD = 3; __VA_ARGS__; } break;        \
    case 4: { constexpr int DD = 4; __VA_ARGS__; } break;        \
    default: set_error("d=%d unsupported (1..4)", d); return OFC_EUNSUPPORTED; \
    }

int launch_lloyd_assign(const void *X, int dtype, int64_t N, int d, int k, const LloydState *st,
                        uint8_t *labels, double *partial, int nblocks, int mode, int first, hipStream_t s)
{
    const int kmax = lloyd_kmax(k);
    if (!kmax) { set_error("k=%d unsupported by the streaming kernel (1..16)", k); return OFC_EUNSUPPORTED; }
    int rc = OFC_OK;
    OFC_D_SWITCH(d, {
        switch (kmax) {
        case 1: rc = launch_assign_d<DD, 1>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 2: rc = launch_assign_d<DD, 2>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 3: rc = launch_assign_d<DD, 3>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 4: rc = launch_assign_d<DD, 4>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 5: rc = launch_assign_d<DD, 5>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 6: rc = launch_assign_d<DD, 6>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 7: rc = launch_assign_d<DD, 7>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        case 8: rc = launch_assign_d<DD, 8>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        default: rc = launch_assign_d<DD, 16>(X, dtype, N, k, st, labels, partial, nblocks, mode, first, s); break;
        }
    })
    return rc;
}

template <int D>
static int launch_colstats_d(const void *X, int dtype, int64_t N, const double *mean, int pass,
                             double *partial, int nblocks, hipStream_t s)
{
    switch (dtype) {
    case OFC_U8: hipLaunchKernelGGL((k_colstats<D, uint8_t>), dim3(nblocks), dim3(256), 0, s, (const uint8_t *)X, N, mean, pass, partial); break;
    case OFC_F32: hipLaunchKernelGGL((k_colstats<D, float>), dim3(nblocks), dim3(256), 0, s, (const float *)X, N, mean, pass, partial); break;
    case OFC_F64: hipLaunchKernelGGL((k_colstats<D, double>), dim3(nblocks), dim3(256), 0, s, (const double *)X, N, mean, pass, partial); break;
    default: set_error("bad dtype %d", dtype); return OFC_EINVAL;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_lloyd_colstats(const void *X, int dtype, int64_t N, int d, const double *mean, int pass,
                          double *partial, int nblocks, hipStream_t s)
{
    int rc = OFC_OK;
    OFC_D_SWITCH(d, { rc = launch_colstats_d<DD>(X, dtype, N, mean, pass, partial, nblocks, s); })
    return rc;
}

int launch_reduce_records(const double *partial, int nblocks, int nv, double *out, hipStream_t s, const int *halt)
{
    if (nv > 256) { set_error("record too long"); return OFC_EINVAL; }
    hipLaunchKernelGGL(k_reduce_records, dim3(1), dim3(1024), 0, s, partial, nblocks, nv, out, halt);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_lloyd_update(LloydState *st, const double *tot, int k, int d, int after_reloc, int labelled, int first,
                        double n_total, double tol_rel, LloydStatus *status, hipStream_t s, int tiles)
{
    hipLaunchKernelGGL(k_lloyd_update, dim3(1), dim3(64), 0, s, st, tot, k, d, lloyd_kmax(k), after_reloc, labelled,
                       first, n_total, tol_rel, status, tiles);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_lloyd_set_centers(LloydState *st, int k, int d, hipStream_t s, int prune_policy)
{
    hipLaunchKernelGGL(k_lloyd_set_centers, dim3(1), dim3(64), 0, s, st, k, d, prune_policy);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

template <int D>
static int launch_inertia_d(const void *X, int dtype, int64_t N, const LloydState *st,
                            const uint8_t *labels, double *partial, int nblocks, hipStream_t s)
{
    switch (dtype) {
    case OFC_U8: hipLaunchKernelGGL((k_lloyd_inertia<D, uint8_t>), dim3(nblocks), dim3(256), 0, s, (const uint8_t *)X, N, st, labels, partial); break;
    case OFC_F32: hipLaunchKernelGGL((k_lloyd_inertia<D, float>), dim3(nblocks), dim3(256), 0, s, (const float *)X, N, st, labels, partial); break;
    case OFC_F64: hipLaunchKernelGGL((k_lloyd_inertia<D, double>), dim3(nblocks), dim3(256), 0, s, (const double *)X, N, st, labels, partial); break;
    default: set_error("bad dtype %d", dtype); return OFC_EINVAL;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_lloyd_inertia(const void *X, int dtype, int64_t N, int d, const LloydState *st,
                         const uint8_t *labels, double *partial, int nblocks, hipStream_t s)
{
    int rc = OFC_OK;
    OFC_D_SWITCH(d, { rc = launch_inertia_d<DD>(X, dtype, N, st, labels, partial, nblocks, s); })
    return rc;
}

template <int D>
static int launch_farthest_d(const void *X, int dtype, int64_t N, const LloydState *st, const double *c_old,
                             const uint8_t *labels, const int64_t *excl, int n_excl, double *out,
                             int nblocks, hipStream_t s)
{
    switch (dtype) {
    case OFC_U8: hipLaunchKernelGGL((k_lloyd_farthest<D, uint8_t>), dim3(nblocks), dim3(256), 0, s, (const uint8_t *)X, N, st, c_old, labels, excl, n_excl, out); break;
    case OFC_F32: hipLaunchKernelGGL((k_lloyd_farthest<D, float>), dim3(nblocks), dim3(256), 0, s, (const float *)X, N, st, c_old, labels, excl, n_excl, out); break;
    case OFC_F64: hipLaunchKernelGGL((k_lloyd_farthest<D, double>), dim3(nblocks), dim3(256), 0, s, (const double *)X, N, st, c_old, labels, excl, n_excl, out); break;
    default: set_error("bad dtype %d", dtype); return OFC_EINVAL;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_lloyd_farthest(const void *X, int dtype, int64_t N, int d, const LloydState *st,
                          const double *c_old, const uint8_t *labels, const int64_t *excl, int n_excl,
                          double *out, int nblocks, hipStream_t s)
{
    int rc = OFC_OK;
    OFC_D_SWITCH(d, { rc = launch_farthest_d<DD>(X, dtype, N, st, c_old, labels, excl, n_excl, out, nblocks, s); })
    return rc;
}

// ------------------------------------------------------------------------------------------------
// k-means++ seeding step (_kmeans_plusplus, sklearn/cluster/_kmeans.py:230-262): squared distances of every (centred)
// sample to NC candidate rows in sklearn's expanded form max(0, (|c|^2 - 2 c.x) + |x|^2), minimum with the running
// closest distance, per-work-group potential partials.  out: [NC][N] f64, partial: [block][KPP_MAXC].
// ------------------------------------------------------------------------------------------------
constexpr int KPP_MAXC = 8;
struct KppArgs {
    double mean[LLOYD_DMAX];
    double cand[KPP_MAXC][LLOYD_DMAX];     // centred candidate rows
    int n_cand;
};

template <int D, class T>
__global__ __launch_bounds__(256) void k_kpp_candidates(const T *__restrict__ X, int64_t N, KppArgs a,
                                                        const double *__restrict__ closest /* or null */,
                                                        double *__restrict__ out, double *__restrict__ partial)
{
#pragma clang fp contract(off)
    __shared__ double lds[4 * KPP_MAXC];
    double cc[KPP_MAXC], pot[KPP_MAXC];
#pragma unroll
    for (int c = 0; c < KPP_MAXC; c++) {
        double t = 0;
#pragma unroll
        for (int f = 0; f < D; f++) t += a.cand[c][f] * a.cand[c][f];
        cc[c] = t;
        pot[c] = 0;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        double x[D];
        load1<D>(X, i, x);
        double xx = 0;
#pragma unroll
        for (int f = 0; f < D; f++) {
            x[f] -= a.mean[f];
            xx += x[f] * x[f];
        }
        const double cl = closest ? closest[i] : 0.0;
#pragma unroll
        for (int c = 0; c < KPP_MAXC; c++) {
            if (c < a.n_cand) {
                double dot = 0;
#pragma unroll
                for (int f = 0; f < D; f++) dot += a.cand[c][f] * x[f];
                double dd = (-2.0 * dot + cc[c]) + xx;
                dd = dd > 0.0 ? dd : 0.0;
                if (closest) dd = dd < cl ? dd : cl;
                out[(size_t)c * N + i] = dd;
                pot[c] += dd;
            }
        }
    }
    block_reduce_store<KPP_MAXC>(pot, lds, partial + (size_t)blockIdx.x * KPP_MAXC);
}

template <int D>
static int launch_kpp_d(const void *X, int dtype, int64_t N, const KppArgs &a, const double *closest, double *out,
                        double *partial, int nblocks, hipStream_t s)
{
    switch (dtype) {
    case OFC_U8: hipLaunchKernelGGL((k_kpp_candidates<D, uint8_t>), dim3(nblocks), dim3(256), 0, s, (const uint8_t *)X, N, a, closest, out, partial); break;
    case OFC_F32: hipLaunchKernelGGL((k_kpp_candidates<D, float>), dim3(nblocks), dim3(256), 0, s, (const float *)X, N, a, closest, out, partial); break;
    case OFC_F64: hipLaunchKernelGGL((k_kpp_candidates<D, double>), dim3(nblocks), dim3(256), 0, s, (const double *)X, N, a, closest, out, partial); break;
    default: set_error("bad dtype %d", dtype); return OFC_EINVAL;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_kpp_candidates(const void *X, int dtype, int64_t N, int d, const double *mean, const double *cand_centred,
                          int n_cand, const double *closest, double *out, double *partial, int nblocks, hipStream_t s)
{
    if (n_cand < 1 || n_cand > KPP_MAXC) { set_error("n_cand %d outside 1..%d", n_cand, KPP_MAXC); return OFC_EUNSUPPORTED; }
    KppArgs a;
    memset(&a, 0, sizeof(a));
    a.n_cand = n_cand;
    for (int f = 0; f < d; f++) a.mean[f] = mean[f];
    for (int c = 0; c < n_cand; c++)
        for (int f = 0; f < d; f++) a.cand[c][f] = cand_centred[c * d + f];
    int rc = OFC_OK;
    OFC_D_SWITCH(d, { rc = launch_kpp_d<DD>(X, dtype, N, a, closest, out, partial, nblocks, s); })
    return rc;
}

// test hook (ofc_dist_loopback): the all-reduce of W ranks that all hold the same shard -- a sum becomes W-fold repeated
// addition of the local value, max / min / owner-broadcast leave it unchanged
__global__ void k_loopback_reduce(const double *__restrict__ send, double *__restrict__ recv, int count, int world, int sum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double v = send[i];
    double r = v;
    if (sum)
        for (int w = 1; w < world; w++) r += v;
    recv[i] = r;
}

int launch_loopback_reduce(const double *send, double *recv, int count, int world, int sum, hipStream_t s)
{
    hipLaunchKernelGGL(k_loopback_reduce, dim3(cdiv(count, 64)), dim3(64), 0, s, send, recv, count, world, sum);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
