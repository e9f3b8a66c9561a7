/*
 * ofc.h -- C ABI of libofc.so: the MI355X (gfx950) implementation of the dense-Farneback-flow ->
 * k-means hot path of menmitsu/opticalFlowClustering.
 *
 * The reference has no FFI of its own for this path: it reaches the arithmetic through the Python
 * call signatures of cv2 and scikit-learn.  Each entry point below names the reference call it
 * stands in for (file:line relative to k-means-color-clustering/ in the reference).  The Python
 * host side (opticalflowclustering_amd/) binds these with ctypes and re-exposes the reference's
 * own class / function / CLI names on top.
 *
 * Conventions
 *   - plain C types only; all buffers C-contiguous; images are row-major, stride == width.
 *   - "host" pointers are ordinary process memory; "_dev" entry points take device pointers
 *     obtained from ofc_malloc (or any hipMalloc'ed memory of the same device).
 *   - every call returns 0 on success or a negative OFC_E* code; ofc_last_error() gives the
 *     message for the calling host thread.  No C++ exception crosses the ABI.
 *   - handles are not thread-safe; use one per (host thread, GPU).
 *   - there is NO CPU fallback: without a usable gfx950 device every compute call fails with
 *     OFC_ENODEV.
 */
#ifndef OFC_H
#define OFC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFC_VERSION 100 /* 0.1.0 */

enum {
    OFC_OK = 0,
    OFC_EINVAL = -1,   /* bad argument (maps to Python ValueError) */
    OFC_ENODEV = -2,   /* no usable GPU / HIP runtime error at init */
    OFC_EHIP = -3,     /* HIP runtime error during the call */
    OFC_ENOMEM = -4,   /* device allocation failed */
    OFC_ENOTREADY = -5,/* streaming flow: first frame pushed, no pair yet */
    OFC_EUNSUPPORTED = -6, /* parameter combination outside what the kernels implement */
    OFC_ECOMM = -7     /* RCCL error */
};

enum { OFC_U8 = 0, OFC_F32 = 1, OFC_F64 = 2 };

int ofc_version(void);
const char *ofc_last_error(void);
int ofc_device_count(int *n);

/* ---- device memory plumbing (so a host without torch can keep inputs resident in HBM) ---- */
int ofc_malloc(int device, size_t bytes, void **dptr);
int ofc_free(int device, void *dptr);
int ofc_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
int ofc_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int ofc_memset(int device, void *dst_dev, int value, size_t bytes);
int ofc_device_sync(int device);

/* ------------------------------------------------------------------------------------------
 * Dense Farneback flow.
 * Replaces cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, 3, 15, 3, 5, 1.2, 0)
 * (computeOpticalFlowModule.py:20-22, computeOpticalFlow.py:99-101).
 * ------------------------------------------------------------------------------------------ */
typedef struct ofc_fb_params {
    double pyr_scale;  /* 0.5 */
    int levels;        /* 3   */
    int winsize;       /* 15  */
    int iterations;    /* 3   */
    int poly_n;        /* 5   */
    double poly_sigma; /* 1.2 */
    int flags;         /* 0   */
} ofc_fb_params;

void ofc_fb_default_params(ofc_fb_params *p);

typedef struct ofc_flow ofc_flow_t;

/* one engine per (device, resolution).  max_batch = the largest number of frame PAIRS processed
 * per call of ofc_flow_calc_frames_dev; all scratch (pyramid images, polynomial expansions R,
 * matrices M, per-level flows) is allocated here once. */
int ofc_flow_create(int device, int W, int H, const ofc_fb_params *p, int max_batch,
                    ofc_flow_t **out);
void ofc_flow_destroy(ofc_flow_t *f);

/* one isolated pair, host buffers: prev/next HxW u8 -> flow HxWx2 f32 (u = x-disp, v = y-disp) */
int ofc_flow_calc(ofc_flow_t *f, const uint8_t *prev_gray, const uint8_t *next_gray,
                  float *flow_out);

/* batched, device-resident: n_frames consecutive HxW u8 frames -> (n_frames-1) flows,
 * flow_dev[(t*H + y)*W + x][2] between frame t and t+1.  n_frames-1 <= max_batch.
 * Asynchronous on the engine's stream; ofc_flow_sync() (or ofc_device_sync) completes it. */
int ofc_flow_calc_frames_dev(ofc_flow_t *f, const uint8_t *frames_dev, int n_frames,
                             float *flow_dev);
/* As above, and uv_sum_dev[0..1] (device) receive sum(u), sum(v) over the n_frames-1 flow fields in float64: the column
 * sums the clip-wide k-means needs for sklearn's centring (X.mean(axis=0), _kmeans.py:1478-1484) come out of the
 * epilogue of the iteration that writes the field instead of an extra sweep over it (pass them to
 * ofc_kmeans_fit_dev_stats).  Engines without that epilogue (one iteration per level, another winsize, the staged or
 * experimental kernels) form the same two sums with one sweep over the finished field: the call always succeeds. */
int ofc_flow_calc_frames_dev_stats(ofc_flow_t *f, const uint8_t *frames_dev, int n_frames,
                                   float *flow_dev, double *uv_sum_dev);
int ofc_flow_sync(ofc_flow_t *f);

/* streaming form of ComputeOpticalFLow (computeOpticalFlowModule.py:6-36): keeps the previous
 * frame; the first push returns OFC_ENOTREADY and writes nothing. */
int ofc_flow_push_gray(ofc_flow_t *f, const uint8_t *gray, float *flow_out);

/* ComputeOpticalFLow.compute (computeOpticalFlowModule.py:18-36) in one call: BGR frame (host) ->
 * BGR2GRAY on the device -> flow against the previous frame -> HSV-coded BGR visualisation.  The first
 * push stores the frame and returns OFC_ENOTREADY.  vis_out (HxWx3 u8), mean_mag, flow_out (HxWx2 f32)
 * may each be NULL.  The visualisation also stays resident: ofc_flow_last_vis_dev(). */
int ofc_flow_push_bgr(ofc_flow_t *f, const uint8_t *bgr, uint8_t *vis_out, float *mean_mag,
                      float *flow_out);
int ofc_flow_last_vis_dev(ofc_flow_t *f, const uint8_t **vis_dev);

/* ---- single stages, host buffers (parity-test / bench hooks; interleaved layouts as OpenCV's
 *      internals so they compare 1:1 with the oracle) ---- */
/* u8 frame -> f32 pyramid image of level k (GaussianBlur at full res + INTER_LINEAR resize) */
int ofc_level_image(int device, const uint8_t *gray, int W, int H, const ofc_fb_params *p, int k,
                    float *out, int *w_out, int *h_out);
/* FarnebackPolyExp: f32 HxW -> f32 HxWx5 {y-lin, x-lin, y^2, x^2, xy} */
int ofc_polyexp(int device, const float *img, int W, int H, int n, double sigma, float *R5);
/* the same for pyramid level 0 straight from the u8 frame (level-0 blur fused in; what ofc_flow_* runs at level 0):
 * must equal ofc_level_image(k=0) followed by ofc_polyexp bit for bit.  OFC_EUNSUPPORTED when W < 4 or H < 2 */
int ofc_polyexp_u8(int device, const uint8_t *gray, int W, int H, int n, double sigma, float *R5);
/* FarnebackUpdateMatrices: R0,R1 HxWx5, flow HxWx2 -> M HxWx5 */
int ofc_update_matrices(int device, const float *R0, const float *R1, const float *flow, int W,
                        int H, float *M);
/* box mean (winsize) of M + 2x2 solve: M HxWx5 -> flow HxWx2 */
int ofc_box_solve(int device, const float *M, int W, int H, int winsize, float *flow);
/* resize(flow, (w,h), INTER_LINEAR) * mul */
int ofc_flow_resize(int device, const float *flow, int sw, int sh, int dw, int dh, float mul,
                    float *out);

/* `iters` Farneback iterations (update matrices + box mean + solve, as ofc_update_matrices / ofc_box_solve chained:
 * oracle/farneback_ref.c, the loop of FarnebackUpdateFlow_Blur) from flow_in, with the engine's fused kernels.
 * mode 0: one launch per iteration (k_flow_iter); mode 1: two iterations per launch where possible (k_flow_iter2);
 * mode 2: one launch per iteration with the 3-waves-per-SIMD kernel (k_flow_iter_w3); rows_per_block 0 = automatic
 * strip height.  Parity-test hook. */
int ofc_flow_iterate(int device, const float *R0, const float *R1, const float *flow_in, int W, int H,
                     int winsize, int iters, int mode, int rows_per_block, float *flow_out);
/* bench hook: one 1080p-style level of `n_pairs` resident pairs, the last two of the three iterations of a level
 * timed with HIP events on their stream: mode 0 = two single-iteration launches, mode 1 = one two-iteration launch,
 * mode 2 = two launches of the 3-waves-per-SIMD kernel.
 * *ms = average duration of those two iterations per batch. */
int ofc_bench_flow_iters(int device, int W, int H, int n_pairs, int reps, int mode, float *ms);

/* bench hook: time `iters` launches of the polyexp kernel over n_images distinct resident images
 * with HIP events on the kernel's own stream; *ms_per_launch = average launch duration. */
int ofc_bench_polyexp(int device, int W, int H, int n_images, int iters, int rows_per_block,
                      float *ms_per_launch);

/* ------------------------------------------------------------------------------------------
 * Flow visualisation and grid.
 * ------------------------------------------------------------------------------------------ */
/* cvtColor(BGR2GRAY) u8 (computeOpticalFlowModule.py:16,19) */
int ofc_bgr2gray(int device, const uint8_t *bgr, int W, int H, uint8_t *gray);
/* cartToPolar + hue/normalize(MINMAX)/HSV2BGR (computeOpticalFlowModule.py:25-33) and
 * np.mean(magnitude) (computeOpticalFlow.py:114-117): flow HxWx2 f32 -> bgr HxWx3 u8 */
int ofc_flow_to_bgr(int device, const float *flow, int W, int H, uint8_t *bgr, float *mean_mag);
int ofc_flow_to_bgr_dev(int device, const float *flow_dev, int W, int H, int n_frames,
                        uint8_t *bgr_dev, float *mean_mag_dev);
/* cvtColor(BGR2HSV) u8, H in [0,180) (KmeanGrids.py:86,336; color_kmeans.py:121) */
int ofc_bgr2hsv(int device, const uint8_t *bgr, int64_t npix, uint8_t *hsv);
/* preprocess_image (KmeanGrids.py:269-286, color_kmeans.py:35-52): per channel <thresh -> 0,
 * alpha = 255 where BGR2GRAY > 0; HxWx3 u8 -> HxWx4 u8 (channel order kept as given) */
int ofc_preprocess_rgba(int device, const uint8_t *img3, int64_t npix, int thresh, uint8_t *rgba);
/* overlayGridAndComputeAvgColor (KmeanGrids.py:52-113): per-cell mean BGR -> u8 -> BGR2HSV.
 * mean_bgr, hsv: rows*cols x 3 u8 */
int ofc_grid_cell_means(int device, const uint8_t *bgr, int W, int H, int rows, int cols,
                        uint8_t *mean_bgr, uint8_t *hsv);

/* ------------------------------------------------------------------------------------------
 * Lloyd k-means.  Replaces sklearn.cluster.KMeans(n_clusters=k, init=<k x d>, n_init=1,
 * max_iter, tol).fit(X) / .predict(X) (color_kmeans.py:66-78, KmeanGrids.py:300-304).
 * All arithmetic in f64 whatever the storage dtype (sklearn casts u8 to f64).
 * ------------------------------------------------------------------------------------------ */
/* host buffers. X: N x d of `dtype`; init: k x d f64 (required: explicit init = determinism);
 * centers k x d f64, labels N i32, inertia, n_iter as sklearn's attributes of the same name. */
int ofc_kmeans_fit(int device, const void *X, int dtype, int64_t N, int d, int k,
                   const double *init, int max_iter, double tol_rel, double *centers,
                   int32_t *labels, double *inertia, int *n_iter);
/* KMeans.predict: one E-step of the un-centred X against `centers` */
int ofc_kmeans_predict(int device, const void *X, int dtype, int64_t N, int d, int k,
                       const double *centers, int32_t *labels);
/* device-resident X (e.g. the (u,v) field ofc_flow_calc_frames_dev just wrote); labels_dev is
 * N x u8 (k <= 255) or NULL.  If a communicator was set up with ofc_dist_init the partial sums are
 * all-reduced over the ranks every iteration (X is then this rank's shard). */
int ofc_kmeans_fit_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k,
                       const double *init, int max_iter, double tol_rel, double *centers,
                       uint8_t *labels_dev, double *inertia, int *n_iter);
/* ofc_kmeans_fit_dev with the column sums of THIS RANK'S rows supplied by the caller (host pointer, d doubles; e.g. added
 * up from ofc_flow_calc_frames_dev_stats): the fit skips its own column-sum sweep.  colsum == NULL: as ofc_kmeans_fit_dev. */
int ofc_kmeans_fit_dev_stats(int device, const void *X_dev, int dtype, int64_t N, int d, int k,
                             const double *init, int max_iter, double tol_rel, const double *colsum,
                             double *centers, uint8_t *labels_dev, double *inertia, int *n_iter);
/* How the last ofc_kmeans_fit_dev[_stats] on `device` swept its samples (no reference counterpart: sklearn's Lloyd,
 * _k_means_lloyd.pyx:23-218, reads every sample in every iteration).  For a float32 d=2 stream (the per-pixel (u,v) vectors
 * of computeOpticalFlow.py:99-101's field) of >= 2^20 samples and k <= 8 the label-less iterations run over 64-sample
 * tiles; a tile whose (u,v) bounding box lies inside one Voronoi cell of the current centres contributes its cached sum
 * without being read, in the iterations and in the final E-step (exact: same labels, same n_iter, centres and inertia equal
 * to rounding).  out6 = [tile sweeps, of them pruned,
 * tiles tested by the pruned sweeps, tiles they skipped, probe sweeps, 1 if the final E-step ran pruned].  Environment: OFC_LLOYD_PRUNE=0 switches
 * the tile sweeps off, 2 enables them for any N, 3 also forces every one of them to run pruned. */
int ofc_lloyd_prune_stats(int device, double *out6);
/* Measurement hook (bench.py's roofline.lloyd): average duration, by HIP events on the Lloyd stream, of `iters` launches of
 * one sweep over the resident float32 (u,v) stream X_dev[N][2] with fixed centres (k x 2, uncentred) and column mean.
 * what = 0: the full label-less sweep (k_lloyd_assign mode 3, 8 B/sample); 1: the pruned tile sweep (metadata built first,
 * untimed); 2: the streaming pass that builds the tile metadata before iteration 0; 3: the full final E-step (labels + inertia, every sample
 * read); 4: the pruned final E-step (tiles inside one cell labelled without being read); 5: the tile sweep in its full mode
 * (every tile walked by sample: what an incoherent field gets). */
int ofc_bench_lloyd_sweep(int device, const float *X_dev, int64_t N, int k, const double *centers, const double *mean,
                          int what, int iters, float *ms_per_launch);
/* ---- building blocks of a HOST-driven sharded fit (opticalflowclustering_amd/sharded.py): the same
 * kernels, one pass per call, records returned to the host so that ANY collective (RCCL, or
 * torch.distributed/gloo across nodes) can combine the shards.  X_dev is this rank's shard. ---- */
/* pass 0: out[f] = sum_i x[i][f]; pass 1: out[f] = sum_i (x[i][f]-mean[f])^2 */
int ofc_lloyd_colstats_dev(int device, const void *X_dev, int dtype, int64_t N, int d, const double *mean,
                           int pass, double *out);
/* one E-step (+ M-step accumulation when accumulate != 0) against the CENTRED centres centers_c (k x d):
 * labels_dev (N x u8, 0xFF = unassigned) is read and rewritten; record = [k*d sums of (x-mean) | k counts |
 * number of labels that changed] */
int ofc_lloyd_step_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                       const double *centers_c, uint8_t *labels_dev, int accumulate, double *record);
/* sum_i ||(x_i - mean) - centers_c[label_i]||^2 */
int ofc_lloyd_inertia_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                          const double *centers_c, const uint8_t *labels_dev, double *inertia);
/* the sample farthest from its assigned (old) centre, skipping the local indices in excl[0..n_excl):
 * *dist2 (-1 if none), *index, x_c[d] = its centred coordinates, *label */
int ofc_lloyd_farthest_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                           const double *centers_c, const uint8_t *labels_dev, const int64_t *excl, int n_excl,
                           double *dist2, int64_t *index, double *x_c, int *label);

/* one step of k-means++ seeding (sklearn's default init for KMeans(n_clusters=k), the reference's construction at
 * color_kmeans.py:66, KmeanGrids.py:300; algorithm: sklearn/cluster/_kmeans.py:230-262): squared distances of all N
 * samples (centred by `mean`) to the rows X[cand[c]], c < n_cand <= 8, in the expanded form sklearn evaluates,
 * element-wise minimum with closest[] when it is not NULL; out_min [n_cand][N] f64, pots[c] = sum_i out_min[c][i].
 * The random draws, the cumulative sum and searchsorted stay on the host (cluster.kmeans_plusplus). */
int ofc_kpp_candidates(int device, const void *X, int dtype, int64_t N, int d, const double *mean,
                       const int64_t *cand, int n_cand, const double *closest, double *out_min, double *pots);
/* many small independent problems in one launch (the per-grid-cell shape of KmeanGrids.py:376-392):
 * problem p owns rows [offsets[p], offsets[p+1]) of X (u8, d = 4); init/centers: P x k x d f64;
 * counts: P x k i32 = np.bincount(predict(X)); labels may be NULL. */
int ofc_kmeans_fit_batched(int device, const uint8_t *X, const int64_t *offsets, int n_problems,
                           int d, int k, const double *init, int max_iter, double tol_rel,
                           double *centers, int32_t *counts, int32_t *labels, int *n_iter);
/* the whole per-frame tail of KmeanGrids.py:376-392 on the device: grid cells of a BGR frame
 * (white row 0 / column 0 as cv2.rectangle leaves them) -> preprocess_image -> KMeans(k) ->
 * dominant cluster -> rint -> BGR2HSV.  init: cells x k x 4 f64.  centers: cells x 4 f64 (dominant
 * centre, rounded); hsv: cells x 3 u8.  channel_order: 0 = BGR as in memory (KmeanGrids.py:385),
 * 1 = swap to RGB first (the disk path, color_kmeansChange.py:33). */
int ofc_grid_kmeans(int device, const uint8_t *bgr, int W, int H, int rows, int cols, int k,
                    const double *init, int max_iter, double tol_rel, int channel_order,
                    double *centers, uint8_t *hsv);

/* same on device-resident frames (n_frames x HxWx3 u8): centers n_frames*cells x 4 f64 and
 * hsv n_frames*cells x 3 u8 are HOST buffers; init is NULL (device seeding) or host n_frames*cells*k*4 */
int ofc_grid_kmeans_dev(int device, const uint8_t *bgr_dev, int W, int H, int n_frames, int rows,
                        int cols, int k, const double *init, int max_iter, double tol_rel,
                        int channel_order, double *centers, uint8_t *hsv);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU; frames (hence (u,v) points) are sharded by rank; the only
 * exchange is one all-reduce (sum, f64) of [k*d sums | k counts | n_changed] per Lloyd iteration.
 * ------------------------------------------------------------------------------------------ */
#define OFC_UNIQUE_ID_BYTES 128
int ofc_dist_unique_id(uint8_t id[OFC_UNIQUE_ID_BYTES]);          /* rank 0, then broadcast */
int ofc_dist_init(int device, int rank, int world, const uint8_t id[OFC_UNIQUE_ID_BYTES]);
int ofc_dist_allreduce_f64(int device, double *buf_dev, int count); /* test hook */
/* The same exchange over a transport the CALLER provides: `fn` all-reduces `count` doubles in place across the ranks on
 * the host (op 0 sum, 1 max, 2 min; returns 0 on success) -- gloo or MPI across nodes without xGMI/RCCL connectivity, or a
 * pipe between processes that share one GPU (how the in-library N > 1 control flow -- all-reduced statistics, iterations
 * behind the halt flag, the relocation exchange with its owner election -- is tested with DIFFERENT shards per rank).
 * Every collective then costs a stream synchronisation and two small copies. */
typedef int (*ofc_host_allreduce_fn)(double *buf, int count, int op, void *user);
int ofc_dist_init_host(int device, int rank, int world, ofc_host_allreduce_fn fn, void *user);
/* TEST HOOK: emulate `world` ranks that all hold the caller's shard, without a communicator (sums become world-fold,
 * max/min unchanged, rank 0): lets one GPU exercise the N>1 control flow of ofc_kmeans_fit_dev -- the result must equal
 * a single-rank fit over `world` concatenated copies of the shard.  world = 1 switches it off. */
int ofc_dist_loopback(int world);
int ofc_dist_finalize(void);

/* ------------------------------------------------------------------------------------------
 * Streaming ingest (BASELINE.json configs[4] shape): frames arrive from the host one at a time (a decoder);
 * they are packed into pinned ring buffers, uploaded with hipMemcpyAsync on a copy stream while the previous batch
 * computes, and reduced on the device to the grid-cell averaged flow: per pair rows*cols (u,v) means.
 * No reference counterpart beyond the cv2.VideoCapture loop (KmeanGrids.py:180-187); the grid geometry is
 * overlayGridAndComputeAvgColor's (KmeanGrids.py:56-59).
 * ------------------------------------------------------------------------------------------ */
typedef struct ofc_stream ofc_stream_t;
int ofc_stream_create(int device, int W, int H, const ofc_fb_params *p, int batch_pairs, int rows, int cols,
                      ofc_stream_t **out);
/* push one HxW u8 frame; *pairs_done = pairs whose cell means are complete so far (may lag the pushes) */
int ofc_stream_push_gray(ofc_stream_t *s, const uint8_t *gray, int *pairs_done);
/* flush the partial batch, wait, and copy all cell means out: cell_uv[n_pairs][rows*cols][2] f32 */
int ofc_stream_finish(ofc_stream_t *s, float *cell_uv, int max_pairs, int *n_pairs);
void ofc_stream_destroy(ofc_stream_t *s);
/* per-cell mean of a flow field (host buffers): flow HxWx2 f32 -> cell_uv rows*cols x 2 f32 */
int ofc_grid_cell_mean_flow(int device, const float *flow, int W, int H, int rows, int cols, float *cell_uv);

/* ------------------------------------------------------------------------------------------
 * Downstream consumer of the hue CSVs (findCosineDifferentVectors.py:5-61): cosine similarity between
 * `small` (n_small values) and every window large[i : i+n_small], i = 0 .. n_large-n_small.
 * sims has n_large-n_small+1 entries; 0 where either norm is 0.  Integer-valued input is summed exactly.
 * ------------------------------------------------------------------------------------------ */
int ofc_sliding_cosine(int device, const double *small_v, int n_small, const double *large_v, int n_large,
                       double *sims);

/* ---- synthetic input generator used by bench.py (not part of the reference path) ---- */
int ofc_synth_frames_dev(int device, uint8_t *frames_dev, int W, int H, int n_frames,
                         int t0, int seed);

#ifdef __cplusplus
}
#endif
#endif /* OFC_H */
