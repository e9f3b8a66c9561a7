/*
 * oracle_ref.h -- CPU restatement ("oracle") of the dense-flow -> k-means hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What is restated, and where it comes from:
 *   - Farneback dense flow: the arithmetic lives in OpenCV (pip `opencv-python`,
 *     version UNPINNED by the reference; call sites
 *     k-means-color-clustering/computeOpticalFlowModule.py:20-22 and
 *     computeOpticalFlow.py:99-101).  OpenCV is not vendored and not installed, so this
 *     is a restatement of its published algorithm (modules/video/src/optflowgf.cpp,
 *     SURVEY.md App. A).  PARITY UNPINNED: the reference holds no flow input->output pair.
 *   - Lloyd k-means: scikit-learn (unpinned; 1.7.2 installed here),
 *     sklearn/cluster/_kmeans.py:279-287,624-752,1427-1554, _k_means_lloyd.pyx:23-218,
 *     _k_means_common.pyx:16-43,167-311.  Pinned by goldens generated
 *     with the installed sklearn (tests/golden/make_lloyd_goldens.py).
 *   - 8-bit colour routines (BGR2GRAY, BGR2HSV, HSV2BGR, cartToPolar, normalize):
 *     OpenCV imgproc/core.  BGR2HSV + the k=1 path are pinned by the two
 *     recorded CSVs of the reference (tests/golden/kat_cells.npz).
 */
#ifndef OFC_ORACLE_REF_H
#define OFC_ORACLE_REF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double pyr_scale;   /* 0.5  */
    int    levels;      /* 3    */
    int    winsize;     /* 15   */
    int    iterations;  /* 3    */
    int    poly_n;      /* 5    */
    double poly_sigma;  /* 1.2  */
    int    flags;       /* 0    */
} ofc_ref_fb_params;

/* ---- Farneback pieces ---- */
void ofc_ref_gaussian_kernel(int n, double sigma, float *k);
void ofc_ref_gaussian_blur(const float *src, int W, int H, int ksize, double sigma, float *dst);
void ofc_ref_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);
void ofc_ref_polyexp_setup(int n, double sigma, float *g, float *xg, float *xxg, double *ig4);
void ofc_ref_polyexp(const float *src, int W, int H, int n, double sigma, float *dst5);
void ofc_ref_update_matrices(const float *R0, const float *R1, const float *flow, float *M,
                             int W, int H, int y0, int y1);
void ofc_ref_update_flow_blur(const float *R0, const float *R1, float *flow, float *M,
                              int W, int H, int block_size, int update_mats);
int  ofc_ref_pyramid_levels(int W, int H, const ofc_ref_fb_params *p);
void ofc_ref_level_geometry(int W, int H, const ofc_ref_fb_params *p, int k,
                            int *w, int *h, int *ksize, double *sigma);
void ofc_ref_level_image(const uint8_t *img, int W, int H, const ofc_ref_fb_params *p, int k,
                         float *I);
int  ofc_ref_farneback(const uint8_t *prev, const uint8_t *next, int W, int H,
                       const ofc_ref_fb_params *p, float *flow);

/* ---- colour / visualisation ---- */
void ofc_ref_bgr2gray(const uint8_t *bgr, int64_t npix, uint8_t *gray);
void ofc_ref_bgr2hsv(const uint8_t *bgr, int64_t npix, uint8_t *hsv);
void ofc_ref_hsv2bgr(const uint8_t *hsv, int64_t npix, uint8_t *bgr);
void ofc_ref_cart_to_polar(const float *x, const float *y, int64_t n, float *mag, float *ang);
void ofc_ref_flow_to_bgr(const float *flow, int W, int H, uint8_t *bgr, float *mean_mag);
void ofc_ref_grid_cell_means(const uint8_t *bgr, int W, int H, int rows, int cols,
                             uint8_t *mean_bgr, uint8_t *hsv);
void ofc_ref_extract_cell(const uint8_t *bgr, int W, int H, int rows, int cols, int cell,
                          uint8_t *cell_bgr);
void ofc_ref_preprocess_rgba(const uint8_t *bgr, int64_t npix, int thresh, uint8_t *rgba);

/* ---- Lloyd.  dtype: 0=u8, 1=f32, 2=f64; all math in f64 ---- */
int ofc_ref_kmeans_fit(const void *X, int dtype, int64_t N, int d, int k, const double *init,
                       int max_iter, double tol_rel, double *centers, int32_t *labels,
                       double *inertia, int *n_iter);
int ofc_ref_kmeans_predict(const void *X, int dtype, int64_t N, int d, int k,
                           const double *centers, int32_t *labels);
/* one E+M accumulation over a shard (centred data): out = [sums k*d][counts k][n_changed] */
int ofc_ref_lloyd_partials(const void *X, int dtype, int64_t N, int d, int k, const double *mean,
                           const double *centers_c, int32_t *labels, double *out);

#ifdef __cplusplus
}
#endif
#endif
