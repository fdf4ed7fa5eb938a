// RCCL leg of the sharded search for hosts that are not Python (include/mmr.h): ONE all-gather
// of the per-shard top-k over xGMI, then mmr_topk_merge.  The Python package uses torch.distributed (backend "nccl" =
// RCCL) for the same exchange; this file gives a C/C++ host the identical step without torch.
// librccl is bound at run time (dlopen) so that libmmr_hip.so keeps linking against nothing but the HIP runtime: a
// single-GPU user never needs RCCL installed, and inside a torch process the already-loaded copy is reused.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include "mmr_common.h"

namespace {
// the few RCCL declarations used (ABI of rccl.h / nccl.h 2.x)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
enum { NCCL_INT64 = 4, NCCL_FLOAT64 = 8 };
typedef int (*GetUniqueIdFn)(NcclUniqueId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, NcclComm, hipStream_t);
typedef int (*GroupFn)(void);
typedef const char *(*GetErrorStringFn)(int);

struct Rccl {
    void *h = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char *names[] = {getenv("MMR_RCCL_LIB"), "librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)           // pass 0: a copy the process already has (torch's); pass 1: load one
        for (const char *n : names) {
            if (!n || !*n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) return;
    Rccl r;
    r.h = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
    r.group_start = (GroupFn)dlsym(h, "ncclGroupStart");
    r.group_end = (GroupFn)dlsym(h, "ncclGroupEnd");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_gather && r.group_start && r.group_end) g_rccl = r;
}

const Rccl *rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.h ? &g_rccl : nullptr;
}

#define MMR_CHECK_NCCL(expr)                                                                               \
    do {                                                                                                   \
        const int _e = (expr);                                                                             \
        if (_e != 0) {                                                                                     \
            mmr::set_error("%s failed: %s", #expr, R->error_string ? R->error_string(_e) : "RCCL error");  \
            return MMR_EIO;                                                                                \
        }                                                                                                  \
    } while (0)
}  // namespace

struct mmr_comm {
    NcclComm comm;
    int rank, world;
};

extern "C" int mmr_comm_unique_id(void *id_host)
{
    MMR_CHECK_ARG(id_host != nullptr, "mmr_comm_unique_id: null pointer");
    const Rccl *R = rccl();
    if (!R) { mmr::set_error("mmr_comm_unique_id: librccl not found (set MMR_RCCL_LIB)"); return MMR_ENOTSUP; }
    NcclUniqueId id;
    MMR_CHECK_NCCL(R->get_unique_id(&id));
    memcpy(id_host, &id, sizeof id);
    return MMR_OK;
}

extern "C" int mmr_comm_init(int rank, int world, const void *id_host, mmr_comm **out)
{
    MMR_CHECK_ARG(id_host && out && world >= 1 && rank >= 0 && rank < world, "mmr_comm_init: bad argument (rank %d of %d)", rank, world);
    const Rccl *R = rccl();
    if (!R) { mmr::set_error("mmr_comm_init: librccl not found (set MMR_RCCL_LIB)"); return MMR_ENOTSUP; }
    NcclUniqueId id;
    memcpy(&id, id_host, sizeof id);
    mmr_comm *c = new (std::nothrow) mmr_comm;
    MMR_CHECK_ARG(c != nullptr, "mmr_comm_init: out of host memory");
    c->rank = rank;
    c->world = world;
    const int e = R->comm_init_rank(&c->comm, world, id, rank);
    if (e != 0) {
        mmr::set_error("ncclCommInitRank failed: %s", R->error_string ? R->error_string(e) : "RCCL error");
        delete c;
        return MMR_EIO;
    }
    *out = c;
    return MMR_OK;
}

extern "C" void mmr_comm_destroy(mmr_comm *c)
{
    if (!c) return;
    const Rccl *R = rccl();
    if (R) (void)R->comm_destroy(c->comm);
    delete c;
}

extern "C" int mmr_allgather_topk_packed(mmr_comm *c, const int64_t *packed_local, int Q, int k, int64_t *packed_parts,
                                         void *stream)
{
    MMR_CHECK_ARG(c != nullptr, "mmr_allgather_topk_packed: null communicator");
    MMR_CHECK_ARG(Q >= 0 && k >= 1, "mmr_allgather_topk_packed: bad shape Q=%d k=%d", Q, k);
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(packed_local && packed_parts, "mmr_allgather_topk_packed: null pointer");
    const Rccl *R = rccl();
    if (!R) { mmr::set_error("mmr_allgather_topk_packed: librccl not found"); return MMR_ENOTSUP; }
    // THE collective of the sharded search: one message of 2*Q*k int64 per rank -- the same exchange
    // search.ShardedGalleryIndex issues through torch.distributed (all_gather_into_tensor of the packed tensor)
    MMR_CHECK_NCCL(R->all_gather(packed_local, packed_parts, (size_t)Q * k * 2, NCCL_INT64, c->comm, (hipStream_t)stream));
    return MMR_OK;
}

extern "C" int mmr_allgather_topk(mmr_comm *c, const int64_t *idx_local, const double *dot_local, int Q, int k,
                                  int64_t *idx_parts, double *dot_parts, void *stream)
{
    MMR_CHECK_ARG(c != nullptr, "mmr_allgather_topk: null communicator");
    MMR_CHECK_ARG(Q >= 0 && k >= 1, "mmr_allgather_topk: bad shape Q=%d k=%d", Q, k);
    if (Q == 0) return MMR_OK;
    MMR_CHECK_ARG(idx_local && dot_local && idx_parts && dot_parts, "mmr_allgather_topk: null pointer");
    const Rccl *R = rccl();
    if (!R) { mmr::set_error("mmr_allgather_topk: librccl not found"); return MMR_ENOTSUP; }
    const size_t n = (size_t)Q * k;
    // the two payload arrays travel as ONE fused collective (a group of two all-gathers of 8-byte elements).
    // A failure between ncclGroupStart and ncclGroupEnd must still close the group: a thread left in group mode
    // queues every later RCCL call (torch's included) without ever launching it.  The FIRST error is reported.
    MMR_CHECK_NCCL(R->group_start());
    int e = R->all_gather(idx_local, idx_parts, n, NCCL_INT64, c->comm, (hipStream_t)stream);
    const char *what = "ncclAllGather(idx)";
    if (e == 0) {
        e = R->all_gather(dot_local, dot_parts, n, NCCL_FLOAT64, c->comm, (hipStream_t)stream);
        what = "ncclAllGather(dot)";
    }
    const int e_end = R->group_end();
    if (e == 0 && e_end != 0) { e = e_end; what = "ncclGroupEnd"; }
    if (e != 0) {
        mmr::set_error("mmr_allgather_topk: %s failed: %s", what, R->error_string ? R->error_string(e) : "RCCL error");
        return MMR_EIO;
    }
    return MMR_OK;
}
