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

#include <rccl/rccl.h>
#include <cstring>

static_assert(sizeof(ncclUniqueId) == GU_COMM_ID_BYTES, "ncclUniqueId size changed");

#define GU_NCCL(expr)                                                                       \
    do {                                                                                    \
        ncclResult_t _r = (expr);                                                           \
        if (_r != ncclSuccess) return gu_fail(GU_ERR_COMM, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)

void gu_comm_free(gu_engine *h)
{
    if (h->comm) {
        (void)ncclCommDestroy((ncclComm_t)h->comm);
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
    ncclUniqueId uid;
    GU_NCCL(ncclGetUniqueId(&uid));
    memcpy(id, &uid, GU_COMM_ID_BYTES);
    return GU_OK;
}

int gu_comm_init(gu_handle h, int32_t nranks, int32_t rank, const uint8_t id[GU_COMM_ID_BYTES])
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(id != nullptr && nranks > 0 && rank >= 0 && rank < nranks, GU_ERR_INVALID, "bad rank %d of %d", rank, nranks);
    gu_comm_free(h);
    ncclUniqueId uid;
    memcpy(&uid, id, GU_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    GU_NCCL(ncclCommInitRank(&comm, nranks, uid, rank));
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
    const size_t n = (size_t)h->N, block = 3 * n;
    GU_NCCL(ncclAllGather(h->d_out3, h->d_gather, block, ncclInt32, (ncclComm_t)h->comm, h->stream));
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

}  // extern "C"
