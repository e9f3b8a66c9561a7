// stream_api.cpp -- C ABI of libofc.so, part 4: streaming ingest with double-buffered pinned uploads
// (BASELINE.json configs[4] shape).  Two slots; while slot A's batch computes on the engine's stream, the host fills
// slot B's pinned buffer and its hipMemcpyAsync runs on the copy stream.  Consecutive batches overlap by one frame.
#include "color_common.h"

#include <memory>

using namespace ofc;

struct ofc_stream {
    int device = 0, W = 0, H = 0, batch = 0, rows = 0, cols = 0;
    ofc_flow_t *flow = nullptr;
    hipStream_t copy = nullptr, compute = nullptr;
    struct Slot {
        uint8_t *pinned = nullptr;     // (batch+1) frames, host
        DevBuf frames, flows;          // device
        hipEvent_t uploaded = nullptr, done = nullptr;
        int n_frames = 0;              // frames currently packed in `pinned`
        int inflight_pairs = 0;        // pairs of the batch in flight from this slot
        bool busy = false;             // a batch of this slot is in flight
    } slot[2];
    int cur = 0;
    DevBuf cells;                      // [max_pairs][rows*cols][2] f32, grows
    size_t cells_cap = 0;              // pairs
    int pairs_submitted = 0;
};

namespace {

int ensure_cells(ofc_stream *s, int need_pairs)
{
    if ((size_t)need_pairs <= s->cells_cap) return OFC_OK;
    size_t cap = std::max<size_t>(need_pairs, s->cells_cap ? s->cells_cap * 2 : 256);
    DevBuf nb;
    const size_t per = sizeof(float) * 2 * s->rows * s->cols;
    OFC_TRY(nb.alloc(cap * per));
    if (s->cells.p) {
        OFC_HIP(hipStreamSynchronize(s->compute));
        OFC_HIP(hipMemcpy(nb.p, s->cells.p, (size_t)s->pairs_submitted * per, hipMemcpyDeviceToDevice));
    }
    std::swap(s->cells.p, nb.p);
    std::swap(s->cells.bytes, nb.bytes);
    s->cells_cap = cap;
    return OFC_OK;
}

// submit the frames packed in slot `i` (n >= 2): async upload, flow, cell means
int submit(ofc_stream *s, int i)
{
    ofc_stream::Slot &sl = s->slot[i];
    const int npair = sl.n_frames - 1;
    const size_t P = (size_t)s->W * s->H;
    OFC_TRY(ensure_cells(s, s->pairs_submitted + npair));
    OFC_HIP(hipMemcpyAsync(sl.frames.p, sl.pinned, P * sl.n_frames, hipMemcpyHostToDevice, s->copy));
    OFC_HIP(hipEventRecord(sl.uploaded, s->copy));
    OFC_HIP(hipStreamWaitEvent(s->compute, sl.uploaded, 0));
    OFC_TRY(ofc_flow_calc_frames_dev(s->flow, sl.frames.as<uint8_t>(), sl.n_frames, sl.flows.as<float>()));
    float *dst = s->cells.as<float>() + (size_t)s->pairs_submitted * 2 * s->rows * s->cols;
    OFC_TRY(launch_grid_cell_mean_flow(sl.flows.as<float>(), s->W, s->H, npair, s->rows, s->cols, dst, s->compute));
    OFC_HIP(hipEventRecord(sl.done, s->compute));
    sl.busy = true;
    sl.inflight_pairs = npair;
    s->pairs_submitted += npair;
    return OFC_OK;
}

}  // namespace

extern "C" {

hipStream_t ofc_flow_stream_internal(ofc_flow_t *f);   // ofc_api.cpp

int ofc_stream_create(int device, int W, int H, const ofc_fb_params *p, int batch_pairs, int rows, int cols,
                      ofc_stream_t **out)
{
    OFC_REQUIRE(out, "null out pointer");
    *out = nullptr;
    OFC_REQUIRE(batch_pairs >= 1 && rows >= 1 && cols >= 1 && W >= cols && H >= rows, "bad arguments");
    OFC_TRY(ensure_device(device));
    std::unique_ptr<ofc_stream> s(new ofc_stream);
    s->device = device; s->W = W; s->H = H; s->batch = batch_pairs; s->rows = rows; s->cols = cols;
    OFC_TRY(ofc_flow_create(device, W, H, p, batch_pairs, &s->flow));
    s->compute = ofc_flow_stream_internal(s->flow);
    OFC_HIP(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
    const size_t P = (size_t)W * H;
    for (auto &sl : s->slot) {
        OFC_HIP(hipHostMalloc((void **)&sl.pinned, P * (batch_pairs + 1), hipHostMallocDefault));
        OFC_TRY(sl.frames.alloc(P * (batch_pairs + 1)));
        OFC_TRY(sl.flows.alloc(sizeof(float) * 2 * P * batch_pairs));
        OFC_HIP(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
        OFC_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    *out = s.release();
    return OFC_OK;
}

int ofc_stream_push_gray(ofc_stream_t *s, const uint8_t *gray, int *pairs_done)
{
    OFC_REQUIRE(s && gray, "null pointer");
    OFC_TRY(ensure_device(s->device));
    const size_t P = (size_t)s->W * s->H;
    ofc_stream::Slot *sl = &s->slot[s->cur];
    if (sl->busy && sl->n_frames == 0) {            // about to refill a slot whose batch may still be running
        OFC_HIP(hipEventSynchronize(sl->done));
        sl->busy = false;
    }
    memcpy(sl->pinned + P * sl->n_frames, gray, P);
    sl->n_frames++;
    if (sl->n_frames == s->batch + 1) {
        OFC_TRY(submit(s, s->cur));
        // next slot starts with this batch's last frame (one-frame overlap)
        ofc_stream::Slot *nx = &s->slot[s->cur ^ 1];
        if (nx->busy) {
            OFC_HIP(hipEventSynchronize(nx->done));
            nx->busy = false;
        }
        memcpy(nx->pinned, sl->pinned + P * s->batch, P);
        nx->n_frames = 1;
        sl->n_frames = 0;
        s->cur ^= 1;
    }
    if (pairs_done) {
        int done = s->pairs_submitted;
        for (auto &q : s->slot)
            if (q.busy && hipEventQuery(q.done) != hipSuccess) done -= q.inflight_pairs;
        *pairs_done = done;
    }
    return OFC_OK;
}

int ofc_stream_finish(ofc_stream_t *s, float *cell_uv, int max_pairs, int *n_pairs)
{
    OFC_REQUIRE(s && n_pairs, "null pointer");
    OFC_TRY(ensure_device(s->device));
    ofc_stream::Slot &sl = s->slot[s->cur];
    if (sl.n_frames >= 2) {
        if (sl.busy) { OFC_HIP(hipEventSynchronize(sl.done)); sl.busy = false; }
        OFC_TRY(submit(s, s->cur));
    }
    sl.n_frames = 0;
    OFC_HIP(hipStreamSynchronize(s->copy));
    OFC_HIP(hipStreamSynchronize(s->compute));
    for (auto &q : s->slot) q.busy = false;
    *n_pairs = s->pairs_submitted;
    if (cell_uv) {
        OFC_REQUIRE(max_pairs >= s->pairs_submitted, "cell_uv holds %d pairs, %d produced", max_pairs, s->pairs_submitted);
        OFC_HIP(hipMemcpy(cell_uv, s->cells.p, sizeof(float) * 2 * s->rows * s->cols * (size_t)s->pairs_submitted,
                          hipMemcpyDeviceToHost));
    }
    s->pairs_submitted = 0;
    return OFC_OK;
}

void ofc_stream_destroy(ofc_stream_t *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->compute) (void)hipStreamSynchronize(s->compute);
    if (s->copy) { (void)hipStreamSynchronize(s->copy); (void)hipStreamDestroy(s->copy); }
    for (auto &sl : s->slot) {
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    ofc_flow_destroy(s->flow);
    delete s;
}

int ofc_grid_cell_mean_flow(int device, const float *flow, int W, int H, int rows, int cols, float *cell_uv)
{
    OFC_REQUIRE(flow && cell_uv && rows >= 1 && cols >= 1 && W >= cols && H >= rows, "bad arguments");
    OFC_TRY(ensure_device(device));
    DevBuf f, o;
    OFC_TRY(f.alloc(sizeof(float) * 2 * W * H));
    OFC_TRY(o.alloc(sizeof(float) * 2 * rows * cols));
    OFC_HIP(hipMemcpy(f.p, flow, sizeof(float) * 2 * W * H, hipMemcpyHostToDevice));
    OFC_TRY(launch_grid_cell_mean_flow(f.as<float>(), W, H, 1, rows, cols, o.as<float>(), nullptr));
    OFC_HIP(hipMemcpy(cell_uv, o.p, sizeof(float) * 2 * rows * cols, hipMemcpyDeviceToHost));
    return OFC_OK;
}

}  // extern "C"
