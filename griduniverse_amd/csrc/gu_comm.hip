// gu_comm.hip -- gathered (obs, reward, done) view over RCCL / xGMI.
//
// The reference is single-process and has no collective; env
// instances are independent, so the step / rollout data path of a sharded batch never
// communicates.  The ONLY exchange is this optional view: one ncclAllGather of each
// rank's packed int32[3N] block (pos|reward|done are contiguous in HBM for exactly this
// reason).  With 7 direct xGMI links per MI355X the gather is a single hop per peer and,
// at N = 32768 (393 KB per rank), launch-latency- not link-bandwidth-bound; it is never
// issued inside a throughput loop.
#include "gu_internal.hpp"

#include <rccl/rccl.h>  // types and prototypes only: the library itself is loaded on first use (below)
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static_assert(sizeof(ncclUniqueId) == GU_COMM_ID_BYTES, "ncclUniqueId size changed");

// librccl.so is 570 MB of code objects.  Linking it would make EVERY process that loads libgu.so map and register it
// at HIP start-up -- seconds on a warm box, minutes on a cold one -- although stepping never communicates.  It is
// therefore dlopen'ed by the first gathered-view call, and only the eight entry points used here are resolved.
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char *error = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

template <typename F>
bool rccl_sym(F &slot, const char *name)
{
    slot = reinterpret_cast<F>(dlsym(g_rccl.lib, name));
    return slot != nullptr;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        // RCCL must sit on the SAME HIP / HSA runtime this library runs on.  A process may hold two ROCm stacks -- e.g. the
        // system one under /opt/rocm and the copy a PyTorch wheel bundles -- and an RCCL bound to the runtime that was not
        // initialised sees no device ("unhandled cuda error").  So the first candidate is the librccl next to the
        // libamdhip64 that resolved OUR HIP calls; only then the loader's default search.  RTLD_LOCAL | RTLD_DEEPBIND: a second
        // copy of RCCL in the process must keep to its own symbols.
        std::string beside;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            beside = info.dli_fname;
            const size_t slash = beside.rfind('/');
            beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
        }
        std::vector<std::string> names;
        if (const char *forced = std::getenv("GU_RCCL_LIB")) names.push_back(forced);  // an explicit choice comes first (also: the test double of tests/c_abi/)
        if (!beside.empty()) {
            names.push_back(beside + "librccl.so.1");
            names.push_back(beside + "librccl.so");
        }
        for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
        for (const std::string &name : names) {
            g_rccl.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) {
            g_rccl.error = "librccl.so.1 not found (dlopen)";
            return;
        }
        const bool ok = rccl_sym(g_rccl.GetUniqueId, "ncclGetUniqueId") && rccl_sym(g_rccl.CommInitRank, "ncclCommInitRank") &&
                        rccl_sym(g_rccl.CommInitAll, "ncclCommInitAll") && rccl_sym(g_rccl.CommDestroy, "ncclCommDestroy") &&
                        rccl_sym(g_rccl.AllGather, "ncclAllGather") && rccl_sym(g_rccl.GroupStart, "ncclGroupStart") &&
                        rccl_sym(g_rccl.GroupEnd, "ncclGroupEnd") && rccl_sym(g_rccl.GetErrorString, "ncclGetErrorString");
        if (!ok) g_rccl.error = "librccl.so.1 lacks an expected entry point";
    });
    return g_rccl.error ? nullptr : &g_rccl;
}
}  // namespace

#define GU_RCCL_OR_FAIL(R)                                                       \
    const Rccl *R = rccl();                                                      \
    if (!R) return gu_fail(GU_ERR_COMM, "RCCL unavailable: %s", g_rccl.error)

#define GU_NCCL(expr)                                                                       \
    do {                                                                                    \
        ncclResult_t _r = (expr);                                                           \
        if (_r != ncclSuccess) return gu_fail(GU_ERR_COMM, "%s failed: %s", #expr, nc->GetErrorString(_r)); \
    } while (0)

void gu_comm_free(gu_engine *h)
{
    if (h->comm) {
        if (const Rccl *nc = rccl()) (void)nc->CommDestroy((ncclComm_t)h->comm);
        h->comm = nullptr;
    }
    if (h->d_gather) {
        (void)hipFree(h->d_gather);
        h->d_gather = nullptr;
    }
    h->nranks = 0;
    h->rank = 0;
}

extern "C" {

int gu_comm_unique_id(uint8_t id[GU_COMM_ID_BYTES])
{
    GU_REQUIRE(id != nullptr, GU_ERR_INVALID, "id is NULL");
    GU_RCCL_OR_FAIL(nc);
    ncclUniqueId uid;
    GU_NCCL(nc->GetUniqueId(&uid));
    memcpy(id, &uid, GU_COMM_ID_BYTES);
    return GU_OK;
}

int gu_comm_init(gu_handle h, int32_t nranks, int32_t rank, const uint8_t id[GU_COMM_ID_BYTES])
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(id != nullptr && nranks > 0 && rank >= 0 && rank < nranks, GU_ERR_INVALID, "bad rank %d of %d", rank, nranks);
    GU_RCCL_OR_FAIL(nc);
    gu_comm_free(h);
    ncclUniqueId uid;
    memcpy(&uid, id, GU_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    GU_NCCL(nc->CommInitRank(&comm, nranks, uid, rank));
    h->comm = comm;
    h->nranks = nranks;
    h->rank = rank;
    GU_HIP(hipMalloc(&h->d_gather, (size_t)nranks * 3 * (size_t)h->N * sizeof(int32_t)));
    return GU_OK;
}

int gu_comm_destroy(gu_handle h)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_HIP(hipStreamSynchronize(h->stream));
    gu_comm_free(h);
    return GU_OK;
}

int gu_allgather_view(gu_handle h, int32_t *obs_all, int32_t *reward_all, int32_t *done_all)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->comm != nullptr, GU_ERR_STATE, "no communicator: call gu_comm_init first");
    GU_RCCL_OR_FAIL(nc);
    const size_t n = (size_t)h->N, block = 3 * n;
    GU_NCCL(nc->AllGather(h->d_out3, h->d_gather, block, ncclInt32, (ncclComm_t)h->comm, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    // unpack rank-major [rank][obs|reward|done][N] into three env-major host arrays
    int32_t *dst[3] = {obs_all, reward_all, done_all};
    for (int k = 0; k < 3; ++k) {
        if (!dst[k]) continue;
        GU_HIP(hipMemcpy2D(dst[k], n * sizeof(int32_t), h->d_gather + (size_t)k * n, block * sizeof(int32_t),
                           n * sizeof(int32_t), (size_t)h->nranks, hipMemcpyDeviceToHost));
    }
    return GU_OK;
}

// ---- single process, several handles (one per device): ncclCommInitAll + grouped all-gather -------------------
int gu_comm_init_all(gu_handle *handles, int32_t n)
{
    GU_REQUIRE(handles != nullptr && n > 0 && n <= 64, GU_ERR_INVALID, "handles is NULL or n outside 1..64");
    std::vector<int> devs((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        GU_REQUIRE(handles[i] != nullptr, GU_ERR_INVALID, "handles[%d] is NULL", i);
        GU_REQUIRE(handles[i]->N == handles[0]->N, GU_ERR_INVALID, "all shards must hold the same number of envs");
        devs[(size_t)i] = handles[i]->device;
        for (int32_t j = 0; j < i; ++j)
            GU_REQUIRE(devs[(size_t)j] != devs[(size_t)i], GU_ERR_INVALID, "RCCL needs one device per rank: handles %d and %d share device %d", j, i, devs[(size_t)i]);
    }
    GU_RCCL_OR_FAIL(nc);
    std::vector<ncclComm_t> comms((size_t)n);
    for (int32_t i = 0; i < n; ++i) gu_comm_free(handles[i]);
    GU_NCCL(nc->CommInitAll(comms.data(), n, devs.data()));
    // every handle owns its communicator from here on, so that a failure below (or later) releases all of them through
    // gu_comm_free / gu_destroy
    for (int32_t i = 0; i < n; ++i) {
        handles[i]->comm = comms[(size_t)i];
        handles[i]->nranks = n;
        handles[i]->rank = i;
    }
    for (int32_t i = 0; i < n; ++i) {
        gu_engine *h = handles[i];
        hipError_t e = hipSetDevice(h->device);
        if (e == hipSuccess) e = hipMalloc(&h->d_gather, (size_t)n * 3 * (size_t)h->N * sizeof(int32_t));
        if (e != hipSuccess) {
            for (int32_t j = 0; j < n; ++j) gu_comm_free(handles[j]);
            return gu_fail(e == hipErrorOutOfMemory ? GU_ERR_NOMEM : GU_ERR_HIP, "gather buffer of rank %d: %s", i, hipGetErrorString(e));
        }
    }
    return GU_OK;
}

int gu_allgather_view_all(gu_handle *handles, int32_t n, int32_t *obs_all, int32_t *reward_all, int32_t *done_all)
{
    GU_REQUIRE(handles != nullptr && n > 0, GU_ERR_INVALID, "handles is NULL or n <= 0");
    for (int32_t i = 0; i < n; ++i)
        GU_REQUIRE(handles[i] && handles[i]->comm && handles[i]->nranks == n && handles[i]->rank == i, GU_ERR_STATE,
                   "handles[%d] is not rank %d of a %d-rank communicator: call gu_comm_init_all first", i, i, n);
    GU_RCCL_OR_FAIL(nc);
    const size_t cnt = 3 * (size_t)handles[0]->N;
    GU_NCCL(nc->GroupStart());
    for (int32_t i = 0; i < n; ++i) {
        gu_engine *h = handles[i];
        ncclResult_t r = nc->AllGather(h->d_out3, h->d_gather, cnt, ncclInt32, (ncclComm_t)h->comm, h->stream);
        if (r != ncclSuccess) {
            (void)nc->GroupEnd();
            return gu_fail(GU_ERR_COMM, "ncclAllGather on rank %d failed: %s", i, nc->GetErrorString(r));
        }
    }
    GU_NCCL(nc->GroupEnd());
    for (int32_t i = 0; i < n; ++i) {
        GU_HIP(hipSetDevice(handles[i]->device));
        GU_HIP(hipStreamSynchronize(handles[i]->stream));
    }
    gu_engine *h = handles[0];  // every rank now holds the full view; read it from rank 0
    GU_HIP(hipSetDevice(h->device));
    const size_t nn = (size_t)h->N;
    int32_t *dst[3] = {obs_all, reward_all, done_all};
    for (int k = 0; k < 3; ++k) {
        if (!dst[k]) continue;
        GU_HIP(hipMemcpy2D(dst[k], nn * sizeof(int32_t), h->d_gather + (size_t)k * nn, cnt * sizeof(int32_t),
                           nn * sizeof(int32_t), (size_t)n, hipMemcpyDeviceToHost));
    }
    return GU_OK;
}

}  // extern "C"
