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

static constexpr size_t kOverlapMinHaloBytes = (size_t)8 << 20;   // per direction

/* One filtering step of a slab rank with the exchange hidden behind the
 * interior planes (see include/mi355img.h).  Composition of the two entry
 * points above and mi_separable3d_f32_planes, kept native so that a step costs
 * one host call: at 8 ranks the per-rank kernel time is ~30 us and Python-side
 * marshalling of three calls would dominate it. */
int mi_slab_separable3d_f32(mi_comm comm, const mi_array *ext_in, const mi_array *ext_out,
                            const double *const weights[3], const int wlen[3], const int origin[3],
                            const int mode[3], double cval, int lo, int hi, int prev_rank, int next_rank,
                            int overlap, mi_stream comm_stream, mi_event input_free, mi_event halos_ready,
                            mi_stream stream)
{
    MI_REQUIRE(ext_in && ext_out, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(ext_in->ndim == 3 && ext_out->ndim == 3, MI_ERR_INVALID_ARG, "slabs are 3-D");
    MI_REQUIRE(lo >= 0 && hi >= 0, MI_ERR_INVALID_ARG, "negative halo");
    const bool has_prev = prev_rank >= 0, has_next = next_rank >= 0;
    const int64_t lo_p = has_prev ? lo : 0, hi_p = has_next ? hi : 0;
    const int64_t n_ext = ext_in->shape[0], n_local = n_ext - lo_p - hi_p;
    MI_REQUIRE(n_local >= 1 && n_local >= lo && n_local >= hi, MI_ERR_INVALID_ARG,
               "slab is thinner than the halo it has to provide");
    // the axis-0 kernel must fit the halo the plan exchanges: otherwise the planes next to a neighbour would be
    // filtered with the boundary mode applied at an interior slab edge
    MI_REQUIRE(weights && wlen && origin, MI_ERR_INVALID_ARG, "NULL argument");
    if (weights[0] && wlen[0] > 1) {
        const int need_lo = wlen[0] / 2 + origin[0], need_hi = wlen[0] - 1 - need_lo;
        MI_REQUIRE(need_lo >= 0 && need_hi >= 0, MI_ERR_INVALID_ARG, "invalid origin");
        MI_REQUIRE((!has_prev || need_lo <= lo) && (!has_next || need_hi <= hi), MI_ERR_INVALID_ARG,
                   "the axis-0 kernel reaches beyond the halo of the slab plan");
    }
    const int64_t a = lo_p, b = a + n_local;
    if (!has_prev && !has_next) {
        const int64_t all[2] = {a, b};
        return mi_separable3d_f32_planes(ext_in, ext_out, weights, wlen, origin, mode, cval, all, 1, stream);
    }
    MI_REQUIRE(comm, MI_ERR_INVALID_ARG, "a communicator is required when a neighbour exists");
    const size_t plane_bytes = (size_t)ext_in->strides[0];
    char *base = (char *)ext_in->data - (size_t)(lo - lo_p) * plane_bytes;
    // Overlapping costs two cross-stream waits and a second (small) launch,
    // ~20 us on MI355X; it pays once the exchange itself takes longer than that.
    if (overlap < 0) overlap = (size_t)(lo > hi ? lo : hi) * plane_bytes >= kOverlapMinHaloBytes;
    if (!overlap) {
        int rc = mi_halo_exchange(comm, base, plane_bytes, n_local, lo, hi, prev_rank, next_rank, stream);
        if (rc != MI_OK) return rc;
        const int64_t all[2] = {a, b};
        rc = mi_separable3d_f32_planes(ext_in, ext_out, weights, wlen, origin, mode, cval, all, 1, stream);
        // kernels that take no plane ranges (> 9 taps): filter the halo planes too, they are scratch
        if (rc == MI_ERR_UNSUPPORTED)
            rc = mi_separable3d_f32(ext_in, ext_out, weights, wlen, origin, mode, cval, 0, stream);
        return rc;
    }
    MI_REQUIRE(comm_stream && input_free && halos_ready, MI_ERR_INVALID_ARG,
               "comm stream and both events are required for the overlapped schedule");
    hipStream_t s = resolve_stream(stream);
    hipStream_t cs = (hipStream_t)comm_stream;
    MI_REQUIRE(cs != s, MI_ERR_INVALID_ARG, "the comm stream must differ from the compute stream");
    // everything queued so far (producers of the local planes, readers of the old halos)
    MI_HIP(hipEventRecord((hipEvent_t)input_free, s));
    const int64_t ib = a + lo_p, ie = b - hi_p;
    const bool has_interior = ib < ie;
    if (has_interior) {
        const int64_t interior[2] = {ib, ie};
        int rc = mi_separable3d_f32_planes(ext_in, ext_out, weights, wlen, origin, mode, cval, interior, 1, stream);
        if (rc != MI_OK) return rc;        // nothing else queued yet: the caller may fall back
    }
    MI_HIP(hipStreamWaitEvent(cs, (hipEvent_t)input_free, 0));
    int rc = mi_halo_exchange(comm, base, plane_bytes, n_local, lo, hi, prev_rank, next_rank, comm_stream);
    if (rc != MI_OK) return rc;
    MI_HIP(hipEventRecord((hipEvent_t)halos_ready, cs));
    MI_HIP(hipStreamWaitEvent(s, (hipEvent_t)halos_ready, 0));
    const int64_t edges[4] = {a, has_interior ? ib : b, has_interior ? ie : b, b};
    return mi_separable3d_f32_planes(ext_in, ext_out, weights, wlen, origin, mode, cval, edges, 2, stream);
}

}  // extern "C"
