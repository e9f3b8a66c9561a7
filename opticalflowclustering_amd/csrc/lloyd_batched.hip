// lloyd_batched.hip -- Lloyd's k-means for MANY SMALL independent problems in one launch: the shape the
// reference actually runs (KmeanGrids.py:376-392: 350 grid cells per frame, each a few thousand RGBA
// pixels, KMeans(n_clusters=k) per cell).  One 256-thread work-group per problem; the problem's points
// (u8 x 4, packed in one dword) live in LDS for the whole fit, so HBM is touched once (read) per problem.
// All Lloyd iterations, the empty-cluster relocation, the final E-step, predict()'s bincount and the
// reference's "dominant cluster -> rint -> BGR2HSV" epilogue (KmeanGrids.py:307-339) run inside the one
// launch.  Semantics = oracle/lloyd_ref.c (== sklearn).  Per-cluster sums are accumulated as exact
// integers (the data is uint8) and centred afterwards: sum(x - mean) = sum(x) - n*mean.
#include "color_common.h"
#include "lloyd_common.h"

namespace ofc {

__device__ __forceinline__ unsigned gray15(unsigned c0, unsigned c1, unsigned c2)
{
    return (c0 * 3735u + c1 * 19235u + c2 * 9798u + (1u << 14)) >> 15;
}

// block-wide argmax of (value, lowest index); result broadcast through sv[0] / si[0]
__device__ __forceinline__ void block_argmax(double v, int idx, double *sv, int *si)
{
    sv[threadIdx.x] = v;
    si[threadIdx.x] = idx;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            const double mv = sv[threadIdx.x];
            const int mi = si[threadIdx.x];
            if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi < mi))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
}

template <int KMAX>
__global__ __launch_bounds__(256) void k_lloyd_batched(BatchedArgs a, int max_points)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *pts = reinterpret_cast<uint32_t *>(smem);
    uint8_t *lab = reinterpret_cast<uint8_t *>(pts + max_points);
    __shared__ double s_mean[4], s_c[KMAX * 4], s_cn[KMAX], s_cnew[KMAX * 4], s_w[KMAX];
    __shared__ double s_tol, s_dred[4][4];
    __shared__ unsigned s_ired[4][KMAX * 5 + 1], s_itot[KMAX * 5 + 1];
    __shared__ double sv[256];
    __shared__ int si[256];
    __shared__ int s_flag, s_nempty;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x, k = a.k;
    int N;
    int64_t base = 0;
    // ---- load the problem into LDS ----
    if (a.X) {
        base = a.offsets[p];
        N = (int)(a.offsets[p + 1] - base);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(a.X) + base;
        for (int i = tid; i < N; i += 256) pts[i] = src[i];
    } else {
        const int cells = a.rows * a.cols;
        const int frame = p / cells, cell = p - frame * cells;
        const int cy = cell / a.cols, cx = cell - cy * a.cols;
        const int xs = a.W / a.cols, ys = a.H / a.rows;
        N = xs * ys;
        const uint8_t *fr = a.bgr + (size_t)frame * a.W * a.H * 3;
        for (int i = tid; i < N; i += 256) {
            const int ly = i / xs, lx = i - ly * xs;
            unsigned c0, c1, c2;
            if (ly == 0 || lx == 0) {
                c0 = c1 = c2 = 255u;                       // cv2.rectangle's white lines
            } else {
                const uint8_t *q = fr + ((size_t)(cy * ys + ly) * a.W + cx * xs + lx) * 3;
                c0 = q[0]; c1 = q[1]; c2 = q[2];
            }
            if (a.channel_order) { const unsigned t = c0; c0 = c2; c2 = t; }   // read_image: BGR2RGB
            c0 = c0 < (unsigned)a.thresh ? 0u : c0;        // image[image < 30] = 0
            c1 = c1 < (unsigned)a.thresh ? 0u : c1;
            c2 = c2 < (unsigned)a.thresh ? 0u : c2;
            const unsigned alpha = gray15(c0, c1, c2) > 0 ? 255u : 0u;
            pts[i] = c0 | (c1 << 8) | (c2 << 16) | (alpha << 24);
        }
    }
    for (int i = tid; i < N; i += 256) lab[i] = 0xFF;
    __syncthreads();
    if (N < k || N == 0) {       // ValueError in sklearn; reported through n_iter = -1
        if (tid == 0 && a.n_iter) a.n_iter[p] = -1;
        return;
    }

    // ---- column mean (exact integer sums) and tol = mean(var) * tol_rel ----
    {
        unsigned s[4] = {0, 0, 0, 0};
        for (int i = tid; i < N; i += 256) {
            const unsigned w = pts[i];
            s[0] += w & 255u; s[1] += (w >> 8) & 255u; s[2] += (w >> 16) & 255u; s[3] += w >> 24;
        }
#pragma unroll
        for (int f = 0; f < 4; f++) {
            unsigned v = s[f];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_ired[wave][f] = v;
        }
        __syncthreads();
        if (tid < 4) s_mean[tid] = (double)(s_ired[0][tid] + s_ired[1][tid] + s_ired[2][tid] + s_ired[3][tid]) / (double)N;
        __syncthreads();
        double q[4] = {0, 0, 0, 0};
        for (int i = tid; i < N; i += 256) {
            const unsigned w = pts[i];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const double t = (double)((w >> (8 * f)) & 255u) - s_mean[f];
                q[f] += t * t;
            }
        }
#pragma unroll
        for (int f = 0; f < 4; f++) {
            double v = q[f];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_dred[wave][f] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double var = 0;
            for (int f = 0; f < 4; f++)
                var += (((s_dred[0][f] + s_dred[1][f]) + s_dred[2][f]) + s_dred[3][f]) / (double)N;
            s_tol = (a.tol_rel == 0) ? 0 : var / 4.0 * a.tol_rel;
        }
    }
    // ---- initial centres (centred) ----
    if (a.init) {
        if (tid < k * 4) s_c[tid] = a.init[(size_t)p * k * 4 + tid] - s_mean[tid & 3];
        __syncthreads();
    } else {
        // deterministic maximin seeding: c0 = first point, c_{j+1} = the point farthest (integer squared
        // distance) from the centres chosen so far, ties -> lowest index
        __shared__ unsigned s_seed[KMAX];
        if (tid == 0) s_seed[0] = pts[0];
        __syncthreads();
        for (int j = 1; j < k; j++) {
            double best = -1;
            int bi = -1;
            for (int i = tid; i < N; i += 256) {
                const unsigned w = pts[i];
                unsigned dmin = 0xffffffffu;
                for (int c = 0; c < j; c++) {
                    const unsigned sd = s_seed[c];
                    unsigned d2 = 0;
#pragma unroll
                    for (int f = 0; f < 4; f++) {
                        const int t = (int)((w >> (8 * f)) & 255u) - (int)((sd >> (8 * f)) & 255u);
                        d2 += (unsigned)(t * t);
                    }
                    dmin = min(dmin, d2);
                }
                if ((double)dmin > best) { best = (double)dmin; bi = i; }
            }
            block_argmax(best, bi, sv, si);
            if (tid == 0) s_seed[j] = pts[si[0]];
            __syncthreads();
        }
        if (tid < k * 4) s_c[tid] = (double)((s_seed[tid >> 2] >> (8 * (tid & 3))) & 255u) - s_mean[tid & 3];
        __syncthreads();
    }

    // ---- E-step helper ----
    auto assign = [&](const double (&x)[4]) -> int {
        double best = 0;
        int label = 0;
        for (int j = 0; j < k; j++) {
            double acc = x[0] * s_c[j * 4];
            acc = fma(x[1], s_c[j * 4 + 1], acc);
            acc = fma(x[2], s_c[j * 4 + 2], acc);
            acc = fma(x[3], s_c[j * 4 + 3], acc);
            const double dj = s_cn[j] - 2.0 * acc;
            if (j == 0 || dj < best) { best = dj; label = j; }
        }
        return label;
    };
    auto compute_cn = [&]() {
        if (tid < k) {
            double acc = s_c[tid * 4] * s_c[tid * 4];
            acc = fma(s_c[tid * 4 + 1], s_c[tid * 4 + 1], acc);
            acc = fma(s_c[tid * 4 + 2], s_c[tid * 4 + 2], acc);
            acc = fma(s_c[tid * 4 + 3], s_c[tid * 4 + 3], acc);
            s_cn[tid] = acc;
        }
    };

    // ---- Lloyd iterations ----
    const double m0 = s_mean[0], m1 = s_mean[1], m2 = s_mean[2], m3 = s_mean[3];
    int it = 0;
    bool strict = false;
    for (it = 0; it < a.max_iter; it++) {
        compute_cn();
        __syncthreads();
        unsigned ps[KMAX][4], pc[KMAX], changed = 0;
#pragma unroll
        for (int j = 0; j < KMAX; j++) { pc[j] = 0; ps[j][0] = ps[j][1] = ps[j][2] = ps[j][3] = 0; }
        for (int i = tid; i < N; i += 256) {
            const unsigned w = pts[i];
            const unsigned b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
            const double x[4] = {(double)b0 - m0, (double)b1 - m1, (double)b2 - m2, (double)b3 - m3};
            const int l = assign(x);
            changed += (l != (int)lab[i]);
            lab[i] = (uint8_t)l;
#pragma unroll
            for (int j = 0; j < KMAX; j++) {
                const bool hit = (l == j);
                pc[j] += hit ? 1u : 0u;
                ps[j][0] += hit ? b0 : 0u; ps[j][1] += hit ? b1 : 0u;
                ps[j][2] += hit ? b2 : 0u; ps[j][3] += hit ? b3 : 0u;
            }
        }
        // integer block reduction: exact, hence order-free
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            if (j < k) {
#pragma unroll
                for (int f = 0; f < 5; f++) {
                    unsigned v = f < 4 ? ps[j][f] : pc[j];
                    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
                    if (lane == 0) s_ired[wave][j * 5 + f] = v;
                }
            }
        }
        {
            unsigned v = changed;
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_ired[wave][KMAX * 5] = v;
        }
        __syncthreads();
        if (tid < k * 5 || tid == KMAX * 5)
            s_itot[tid] = s_ired[0][tid] + s_ired[1][tid] + s_ired[2][tid] + s_ired[3][tid];
        __syncthreads();
        if (tid == 0) {
            int ne = 0;
            for (int j = 0; j < k; j++) {
                const double w = (double)s_itot[j * 5 + 4];
                s_w[j] = w;
                ne += (w == 0.0);
                for (int f = 0; f < 4; f++) s_cnew[j * 4 + f] = (double)s_itot[j * 5 + f] - w * s_mean[f];
            }
            s_nempty = ne;
        }
        __syncthreads();
        // ---- _relocate_empty_clusters_dense ----
        if (s_nempty > 0) {
            bool first = true, give_up = false;
            for (int j = 0; j < k && !give_up; j++) {
                if (s_w[j] != 0.0) continue;            // uniform: s_w is shared
                double best = -1;
                int bi = -1;
                for (int i = tid; i < N; i += 256) {
                    if (lab[i] & 0x80) continue;        // already taken in this relocation round
                    const unsigned w = pts[i];
                    const double *c = s_c + (int)lab[i] * 4;
                    double s = 0;
                    {
#pragma clang fp contract(off)
                        const double t0 = ((double)(w & 255u) - m0) - c[0];
                        const double t1 = ((double)((w >> 8) & 255u) - m1) - c[1];
                        const double t2 = ((double)((w >> 16) & 255u) - m2) - c[2];
                        const double t3 = ((double)(w >> 24) - m3) - c[3];
                        s = t0 * t0; s += t1 * t1; s += t2 * t2; s += t3 * t3;
                    }
                    if (s > best) { best = s; bi = i; }
                }
                block_argmax(best, bi, sv, si);
                const double dmax = sv[0];
                const int far = si[0];
                __syncthreads();
                if (first && !(dmax > 0)) { give_up = true; break; }
                first = false;
                if (tid == 0) {
                    const unsigned w = pts[far];
                    const int old = lab[far] & 0x7f;
                    const double x[4] = {(double)(w & 255u) - m0, (double)((w >> 8) & 255u) - m1,
                                         (double)((w >> 16) & 255u) - m2, (double)(w >> 24) - m3};
                    for (int f = 0; f < 4; f++) {
                        s_cnew[old * 4 + f] -= x[f];
                        s_cnew[j * 4 + f] = x[f];
                    }
                    s_w[j] = 1.0;
                    s_w[old] -= 1.0;
                    lab[far] |= 0x80;                   // mark taken (k <= 16, so bit 7 is free)
                }
                __syncthreads();
            }
            for (int i = tid; i < N; i += 256) lab[i] &= 0x7f;
            __syncthreads();
        }
        // ---- _average_centers, _center_shift, convergence ----
        if (tid == 0) {
            int amax = 0;
            for (int j = 1; j < k; j++) if (s_w[j] > s_w[amax]) amax = j;
            for (int j = 0; j < k; j++) {
                if (s_w[j] > 0) {
                    const double alpha = 1.0 / s_w[j];
                    for (int f = 0; f < 4; f++) s_cnew[j * 4 + f] *= alpha;
                } else {
                    for (int f = 0; f < 4; f++) s_cnew[j * 4 + f] = s_cnew[amax * 4 + f];
                }
            }
            double sh2[KMAX];
            for (int j = 0; j < k; j++) {
#pragma clang fp contract(off)
                const double *x = s_cnew + j * 4, *y = s_c + j * 4;
                double r = 0;
                r += ((x[0] - y[0]) * (x[0] - y[0]) + (x[1] - y[1]) * (x[1] - y[1]) +
                      (x[2] - y[2]) * (x[2] - y[2]) + (x[3] - y[3]) * (x[3] - y[3]));
                const double s = sqrt(r);
                sh2[j] = s * s;
            }
            double tot = 0;
            if (k < 8) {
                for (int j = 0; j < k; j++) tot += sh2[j];
            } else {
                double r[8];
                int i;
                for (i = 0; i < 8; i++) r[i] = sh2[i];
                for (i = 8; i < k - (k % 8); i += 8)
                    for (int j = 0; j < 8; j++) r[j] += sh2[i + j];
                tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < k; i++) tot += sh2[i];
            }
            for (int j = 0; j < k * 4; j++) s_c[j] = s_cnew[j];
            s_flag = (s_itot[KMAX * 5] == 0) ? 2 : (tot <= s_tol ? 1 : 0);
        }
        __syncthreads();
        if (s_flag == 2) { strict = true; break; }
        if (s_flag == 1) break;
    }
    if (it == a.max_iter) it = a.max_iter - 1;
    if (!strict) {                               // final E-step with the final centres
        compute_cn();
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            const unsigned w = pts[i];
            const double x[4] = {(double)(w & 255u) - m0, (double)((w >> 8) & 255u) - m1,
                                 (double)((w >> 16) & 255u) - m2, (double)(w >> 24) - m3};
            lab[i] = (uint8_t)assign(x);
        }
        __syncthreads();
    }
    if (a.labels && a.X)
        for (int i = tid; i < N; i += 256) a.labels[base + i] = lab[i];
    if (a.n_iter && tid == 0) a.n_iter[p] = it + 1;
    __syncthreads();

    // ---- cluster_centers_ = centres + mean; predict(): E-step on the un-centred data ----
    if (tid < k * 4) s_c[tid] += s_mean[tid & 3];
    __syncthreads();
    if (a.centers && tid < k * 4) a.centers[(size_t)p * k * 4 + tid] = s_c[tid];
    compute_cn();
    __syncthreads();
    unsigned pc[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) pc[j] = 0;
    for (int i = tid; i < N; i += 256) {
        const unsigned w = pts[i];
        const double x[4] = {(double)(w & 255u), (double)((w >> 8) & 255u), (double)((w >> 16) & 255u), (double)(w >> 24)};
        const int l = assign(x);
#pragma unroll
        for (int j = 0; j < KMAX; j++) pc[j] += (l == j) ? 1u : 0u;
    }
#pragma unroll
    for (int j = 0; j < KMAX; j++) {
        if (j < k) {
            unsigned v = pc[j];
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_ired[wave][j] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int dom = 0;
        unsigned best = 0;
        for (int j = 0; j < k; j++) {
            const unsigned c = s_ired[0][j] + s_ired[1][j] + s_ired[2][j] + s_ired[3][j];
            if (a.counts) a.counts[(size_t)p * k + j] = (int)c;
            if (j == 0 || c > best) { best = c; dom = j; }     // stable sort, descending share
        }
        // np.rint(centre) -> uint8 -> cv2.cvtColor(BGR2HSV)   (KmeanGrids.py:325-339)
        double r[4];
        for (int f = 0; f < 4; f++) r[f] = rint(s_c[dom * 4 + f]);
        if (a.dom_center)
            for (int f = 0; f < 4; f++) a.dom_center[(size_t)p * 4 + f] = r[f];
        if (a.dom_hsv) {
            unsigned h, s, v;
            bgr2hsv_u8((unsigned)(int)r[0] & 255u, (unsigned)(int)r[1] & 255u, (unsigned)(int)r[2] & 255u, h, s, v);
            a.dom_hsv[(size_t)p * 3] = h; a.dom_hsv[(size_t)p * 3 + 1] = s; a.dom_hsv[(size_t)p * 3 + 2] = v;
        }
    }
}

int launch_lloyd_batched(const BatchedArgs &a, int max_points, hipStream_t s)
{
    if (a.k < 1 || a.k > LLOYD_KMAX) {
        set_error("k=%d outside the batched kernel's range (1..%d)", a.k, LLOYD_KMAX);
        return OFC_EUNSUPPORTED;
    }
    const int mp = (max_points + 15) & ~15;
    const size_t lds = (size_t)mp * 5;
    if (lds > 120 * 1024) {
        set_error("a problem of %d points does not fit the LDS-resident kernel (max 24576); use ofc_kmeans_fit", max_points);
        return OFC_EUNSUPPORTED;
    }
    const int kmax = lloyd_kmax(a.k);
    dim3 grid(a.n_problems), block(256);
#define OFC_LB(KM)                                                                                     \
    {                                                                                                  \
        if (lds > 48 * 1024)                                                                           \
            OFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lloyd_batched<KM>),          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(k_lloyd_batched<KM>, grid, block, lds, s, a, mp);                           \
    }
    if (kmax <= 4) OFC_LB(4) else if (kmax <= 8) OFC_LB(8) else OFC_LB(16)
#undef OFC_LB
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
