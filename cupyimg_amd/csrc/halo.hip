// halo.hip -- slab halo exchange between neighbouring GPUs over RCCL / xGMI.
//
// New design (the reference is single-GPU: no NCCL/MPI anywhere, SURVEY.md
// section 2.2).  A volume is partitioned along axis 0 into one slab per rank;
// output plane z needs input planes z-lo .. z+hi with lo = w/2 + origin,
// hi = w - 1 - lo (offset rule of _filters_core.py:10-11), so a rank receives
// `lo` planes from its predecessor and `hi` planes from its successor.  That
// is the only communication: point-to-point ncclSend/ncclRecv pairs in one
// group, each crossing one xGMI link; no all-reduce, no global collective.
#include <rccl/rccl.h>

#include "common.hpp"

namespace mi {
static int nccl_fail(ncclResult_t r, const char *what)
{
    set_error("RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
    return MI_ERR_RCCL;
}
}  // namespace mi

#define MI_NCCL(call)                                             \
    do {                                                          \
        ncclResult_t r__ = (call);                                \
        if (r__ != ncclSuccess) return mi::nccl_fail(r__, #call); \
    } while (0)

using namespace mi;

extern "C" {

int mi_comm_unique_id(char id[MI_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= MI_UNIQUE_ID_BYTES, "unique id does not fit");
    MI_REQUIRE(id, MI_ERR_INVALID_ARG, "id is NULL");
    ncclUniqueId uid;
    MI_NCCL(ncclGetUniqueId(&uid));
    memset(id, 0, MI_UNIQUE_ID_BYTES);
    memcpy(id, &uid, sizeof(uid));
    return MI_OK;
}

int mi_comm_init_rank(mi_comm *comm, int nranks, int rank, const char id[MI_UNIQUE_ID_BYTES])
{
    MI_REQUIRE(comm && id, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, MI_ERR_INVALID_ARG, "bad rank / nranks");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c;
    MI_NCCL(ncclCommInitRank(&c, nranks, uid, rank));
    *comm = (mi_comm)c;
    return MI_OK;
}

int mi_comm_destroy(mi_comm comm)
{
    if (comm) MI_NCCL(ncclCommDestroy((ncclComm_t)comm));
    return MI_OK;
}

int mi_halo_exchange(mi_comm comm, void *slab, size_t plane_bytes, int64_t n_local, int lo, int hi,
                     int prev_rank, int next_rank, mi_stream stream)
{
    MI_REQUIRE(comm && slab, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(lo >= 0 && hi >= 0 && n_local >= 0, MI_ERR_INVALID_ARG, "negative extent");
    MI_REQUIRE(n_local >= lo && n_local >= hi, MI_ERR_INVALID_ARG,
               "slab is thinner than the halo it has to provide");
    hipStream_t s = resolve_stream(stream);
    ncclComm_t c = (ncclComm_t)comm;
    char *base = (char *)slab;
    char *local0 = base + (size_t)lo * plane_bytes;                 // first local plane
    char *local_end = local0 + (size_t)n_local * plane_bytes;       // one past the last local plane
    // Pair every send with the receive that travels in the same direction so
    // that the per-peer ordering also matches when prev == next (two ranks,
    // closed chain): first everything flowing "downwards" (to prev / from
    // next), then everything flowing "upwards".
    MI_NCCL(ncclGroupStart());
    if (hi > 0) {
        if (prev_rank >= 0) MI_NCCL(ncclSend(local0, (size_t)hi * plane_bytes, ncclUint8, prev_rank, c, s));
        if (next_rank >= 0) MI_NCCL(ncclRecv(local_end, (size_t)hi * plane_bytes, ncclUint8, next_rank, c, s));
    }
    if (lo > 0) {
        if (next_rank >= 0) MI_NCCL(ncclSend(local_end - (size_t)lo * plane_bytes, (size_t)lo * plane_bytes,
                                             ncclUint8, next_rank, c, s));
        if (prev_rank >= 0) MI_NCCL(ncclRecv(base, (size_t)lo * plane_bytes, ncclUint8, prev_rank, c, s));
    }
    MI_NCCL(ncclGroupEnd());
    return MI_OK;
}

}  // extern "C"
