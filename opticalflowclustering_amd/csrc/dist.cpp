// dist.cpp -- multi-GPU plumbing: one process per GPU, RCCL over xGMI.  The hot path has exactly one
// exchange step: an all-reduce (sum, f64) of the [k*d sums | k counts | n_changed] record per Lloyd
// iteration; the flow needs no collective.  librccl is dlopen'ed lazily so that
// single-GPU users never load it; a process that already has RCCL loaded (e.g. through
// torch.distributed) shares that instance (same soname).
#include "lloyd_common.h"

#include <dlfcn.h>

namespace ofc {

// minimal RCCL surface (rccl.h: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy)
struct ncclUniqueIdT { char internal[OFC_UNIQUE_ID_BYTES]; };
typedef void *ncclCommT;
enum { NCCL_FLOAT64 = 8 };                      // ncclDouble
enum { NCCL_SUM = 0, NCCL_PROD = 1, NCCL_MAX = 2, NCCL_MIN = 3 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueIdT *) = nullptr;
    int (*CommInitRank)(ncclCommT *, int, ncclUniqueIdT, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclCommT, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclCommT) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static Rccl g_rccl;
static ncclCommT g_comm = nullptr;
static int g_rank = 0, g_world = 1, g_device = -1;
static int g_loopback = 0;      // test hook: world size emulated without a communicator (see ofc_dist_loopback)
static ofc_host_allreduce_fn g_host_fn = nullptr;      // caller-provided transport (ofc_dist_init_host)
static void *g_host_user = nullptr;
static double *g_host_buf = nullptr;                   // pinned staging, 512 doubles

static int load_rccl()
{
    if (g_rccl.handle) return OFC_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        set_error("cannot dlopen librccl: %s", dlerror());
        return OFC_ECOMM;
    }
    g_rccl.handle = h;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        set_error("librccl lacks an expected symbol");
        return OFC_ECOMM;
    }
    return OFC_OK;
}

#define OFC_NCCL(expr)                                                                          \
    do {                                                                                        \
        int _r = (expr);                                                                        \
        if (_r != 0) {                                                                          \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return OFC_ECOMM;                                                                   \
        }                                                                                       \
    } while (0)

bool dist_active() { return (g_comm != nullptr || g_loopback > 1 || g_host_fn) && g_world > 1; }
int dist_rank() { return g_rank; }
int dist_world() { return g_world; }

bool dist_has_comm() { return g_comm != nullptr || g_loopback > 1 || g_host_fn; }

int dist_allreduce_f64(double *buf_dev, int count, int op, hipStream_t s)
{
    return dist_allreduce_f64(buf_dev, buf_dev, count, op, s);
}

// send != recv keeps the local contribution intact, so that issuing the same collective again (an iteration that was
// enqueued speculatively and found the halt flag set) reproduces the same totals instead of summing totals
int dist_allreduce_f64(const double *send_dev, double *recv_dev, int count, int op, hipStream_t s)
{
    if (g_loopback > 1) return launch_loopback_reduce(send_dev, recv_dev, count, g_loopback, op == DIST_SUM, s);
    if (g_host_fn) {
        if (count > 512) { set_error("host transport: %d values exceed the staging buffer", count); return OFC_ECOMM; }
        OFC_HIP(hipMemcpyAsync(g_host_buf, send_dev, sizeof(double) * count, hipMemcpyDeviceToHost, s));
        OFC_HIP(hipStreamSynchronize(s));
        const int hop = op == DIST_MAX ? 1 : (op == DIST_MIN ? 2 : 0);
        if (g_host_fn(g_host_buf, count, hop, g_host_user) != 0) { set_error("host all-reduce callback failed"); return OFC_ECOMM; }
        OFC_HIP(hipMemcpyAsync(recv_dev, g_host_buf, sizeof(double) * count, hipMemcpyHostToDevice, s));
        OFC_HIP(hipStreamSynchronize(s));           // the staging buffer is reused by the next collective
        return OFC_OK;
    }
    if (!g_comm) return OFC_OK;      // a world-1 communicator (OFC_FORCE_DIST rehearsal) still issues the collective
    const int nop = op == DIST_MAX ? NCCL_MAX : (op == DIST_MIN ? NCCL_MIN : NCCL_SUM);
    OFC_NCCL(g_rccl.AllReduce(send_dev, recv_dev, (size_t)count, NCCL_FLOAT64, nop, g_comm, s));
    return OFC_OK;
}

}  // namespace ofc

using namespace ofc;

extern "C" {

int ofc_dist_unique_id(uint8_t id[OFC_UNIQUE_ID_BYTES])
{
    OFC_REQUIRE(id, "null pointer");
    OFC_TRY(load_rccl());
    ncclUniqueIdT u;
    OFC_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, OFC_UNIQUE_ID_BYTES);
    return OFC_OK;
}

int ofc_dist_init(int device, int rank, int world, const uint8_t id[OFC_UNIQUE_ID_BYTES])
{
    OFC_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, "bad rank/world");
    OFC_REQUIRE(!g_comm && !g_host_fn, "communicator already initialised");
    OFC_TRY(ensure_device(device));
    OFC_TRY(load_rccl());
    ncclUniqueIdT u;
    memcpy(u.internal, id, OFC_UNIQUE_ID_BYTES);
    OFC_NCCL(g_rccl.CommInitRank(&g_comm, world, u, rank));
    g_rank = rank; g_world = world; g_device = device;
    return OFC_OK;
}

int ofc_dist_init_host(int device, int rank, int world, ofc_host_allreduce_fn fn, void *user)
{
    OFC_REQUIRE(fn && world >= 1 && rank >= 0 && rank < world, "bad rank/world/callback");
    OFC_REQUIRE(!g_comm && !g_host_fn && g_loopback == 0, "a communicator is already active");
    OFC_TRY(ensure_device(device));
    if (!g_host_buf) OFC_HIP(hipHostMalloc((void **)&g_host_buf, sizeof(double) * 512, hipHostMallocDefault));
    g_host_fn = fn; g_host_user = user;
    g_rank = rank; g_world = world; g_device = device;
    return OFC_OK;
}

int ofc_dist_allreduce_f64(int device, double *buf_dev, int count)
{
    OFC_REQUIRE(buf_dev && count >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    if (g_host_fn) return dist_allreduce_f64(buf_dev, buf_dev, count, DIST_SUM, nullptr);
    OFC_REQUIRE(g_comm, "ofc_dist_init was not called");
    OFC_NCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, NCCL_FLOAT64, NCCL_SUM, g_comm, nullptr));
    OFC_HIP(hipStreamSynchronize(nullptr));
    return OFC_OK;
}

int ofc_dist_loopback(int world)
{
    OFC_REQUIRE(world >= 1 && world <= 64, "bad world");
    OFC_REQUIRE(!g_comm && !g_host_fn, "a communicator is active");
    g_loopback = world > 1 ? world : 0;
    g_world = world > 1 ? world : 1;
    g_rank = 0;
    return OFC_OK;
}

int ofc_dist_finalize(void)
{
    g_loopback = 0;
    g_host_fn = nullptr; g_host_user = nullptr;
    if (g_comm) {
        (void)hipSetDevice(g_device);
        (void)g_rccl.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    g_rank = 0; g_world = 1;
    return OFC_OK;
}

}  // extern "C"
