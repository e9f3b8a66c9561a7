// temporary: entry points not implemented yet (removed as the kernels land)
#include "ofc_common.h"
#define STUB(...) { ofc::set_error("%s: not implemented yet", __func__); return OFC_EUNSUPPORTED; }
extern "C" {
int ofc_bgr2gray(int, const uint8_t *, int, int, uint8_t *) STUB()
int ofc_flow_to_bgr(int, const float *, int, int, uint8_t *, float *) STUB()
int ofc_flow_to_bgr_dev(int, const float *, int, int, int, uint8_t *, float *) STUB()
int ofc_grid_cell_means(int, const uint8_t *, int, int, int, int, uint8_t *, uint8_t *) STUB()
int ofc_kmeans_fit_batched(int, const uint8_t *, const int64_t *, int, int, int, const double *, int, double, double *, int32_t *, int32_t *, int *) STUB()
int ofc_grid_kmeans(int, const uint8_t *, int, int, int, int, int, const double *, int, double, int, double *, uint8_t *) STUB()
int ofc_synth_frames_dev(int, uint8_t *, int, int, int, int, int) STUB()
}
