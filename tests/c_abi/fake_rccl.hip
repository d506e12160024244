// fake_rccl.hip -- TEST DOUBLE for librccl (tests/test_gpu_comm.py only; never shipped, never loaded unless GU_RCCL_LIB
// names it).  Lets ONE GPU play several ranks: every rank is a thread of one process with its own libgu handle, and
// ncclAllGather is a rendezvous of those threads followed by device-to-device copies.  It exists to run gu_comm_init with
// nranks > 1 and the rank-major -> env-major unpack of gu_allgather_view on real hardware where only one device is at hand
// (RCCL itself refuses two ranks on one device).  Build: hipcc --offload-arch=gfx950 -shared -fPIC -o libfake_rccl.so fake_rccl.hip
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
struct Group {
    int nranks = 0, arrived = 0;
    unsigned generation = 0;
    const void *send[64] = {};
    std::mutex m;
    std::condition_variable cv;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};
struct FakeComm {
    Group *g;
    int rank;
};
std::mutex g_table_mutex;
std::map<std::string, Group *> g_table;
unsigned g_next_id = 1;
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::lock_guard<std::mutex> lk(g_table_mutex);
    memset(id, 0, sizeof(*id));
    const unsigned v = g_next_id++;
    memcpy(id, &v, sizeof(v));
    memcpy(id->internal + 8, "fake-rccl", 9);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_table_mutex);
    Group *&g = g_table[std::string(id.internal, sizeof(id.internal))];
    if (!g) {
        g = new Group;
        g->nranks = nranks;
    }
    if (g->nranks != nranks) return ncclInvalidArgument;
    *comm = reinterpret_cast<ncclComm_t>(new FakeComm{g, rank});
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<FakeComm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    if (datatype != ncclInt32) return ncclInvalidArgument;
    FakeComm *c = reinterpret_cast<FakeComm *>(comm);
    Group *g = c->g;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // this rank's block is complete
    g->send[c->rank] = sendbuff;
    g->barrier();  // every rank has published its block
    for (int r = 0; r < g->nranks; ++r)
        if (hipMemcpyAsync((char *)recvbuff + (size_t)r * count * 4, g->send[r], count * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    g->barrier();  // nobody's block is reused before everybody has copied it
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t) { return "fake rccl"; }

}  // extern "C"
