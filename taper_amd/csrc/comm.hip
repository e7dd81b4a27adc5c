// comm.hip -- data-parallel gradient all-reduce over RCCL / xGMI (new: the
// reference is single-process).  One process per GPU; the
// communicator is bootstrapped from a 128-byte unique id shipped out of band.
// The all-reduce runs on the ctx stream so it can sit inside the captured
// step graph between loss.backward() and optim.step().
#include "common.h"

#include <rccl/rccl.h>

#include <cstring>

struct th_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
};

namespace th {
int scale_inplace(th_ctx *ctx, float *d_x, size_t n, float scale);
}

#define TH_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            th::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

extern "C" {

int th_comm_unique_id(uint8_t out_id[128]) {
    TH_REQUIRE(out_id, "th_comm_unique_id: null out");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    TH_NCCL(ncclGetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

int th_comm_init_rank(th_ctx *ctx, int n_ranks, int rank, const uint8_t id[128], th_comm **out) {
    TH_REQUIRE(ctx && id && out, "th_comm_init_rank: null argument");
    TH_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "th_comm_init_rank: bad rank %d of %d", rank, n_ranks);
    TH_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    th_comm *c = new th_comm();
    c->n_ranks = n_ranks;
    c->rank = rank;
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, uid, rank);
    if (r != ncclSuccess) {
        th::set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

int th_comm_destroy(th_comm *comm) {
    if (!comm) return 0;
    if (comm->comm) ncclCommDestroy(comm->comm);
    delete comm;
    return 0;
}

int th_allreduce_sum_scale(th_comm *comm, th_ctx *ctx, float *d_buf, size_t n, float scale) {
    TH_REQUIRE(comm && ctx && (n == 0 || d_buf), "th_allreduce_sum_scale: null argument");
    if (n == 0) return 0;
    // the data-parallel mean (scale == 1/n_ranks) is RCCL's own ncclAvg: no scale launch behind the collective
    const float mean = 1.0f / (float)comm->n_ranks;
    if (scale == mean) {
        TH_NCCL(ncclAllReduce(d_buf, d_buf, n, ncclFloat, ncclAvg, comm->comm, ctx->stream));
        return 0;
    }
    TH_NCCL(ncclAllReduce(d_buf, d_buf, n, ncclFloat, ncclSum, comm->comm, ctx->stream));
    return th::scale_inplace(ctx, d_buf, n, scale);
}

}  // extern "C"
