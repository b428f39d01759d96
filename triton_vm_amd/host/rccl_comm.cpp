// rccl_comm.cpp -> libtriton_rccl.so: the tvmh_comm (triton_host.hpp) of the multi-GPU prover over RCCL / xGMI.
//
// One process per GPU.  The collectives of the default proof are enqueued on the CONTEXT's stream (tvm_ctx_stream), behind the
// kernels that produce their operands and ahead of those that consume their results: no host synchronisation, no second stream,
// no event.  (The opt-in column split's coefficient exchange is the exception: all_gather_async / wait below, on the context's side
// lane, so that it can run under the kernels queued after it.)
// The exchanges of a proof are few and large (leaf digests L x 40 B / R per rank by all-to-all,
// the quotient codeword L x 24 B by all-gather, FRI codewords by all-to-all), so they are issued as single RCCL calls --
// xGMI is point-to-point (7 links per GPU), an all-to-all is R - 1 concurrent peer transfers, one per link.
//
// Rendezvous: rank 0 draws a ncclUniqueId (tvmh_rccl_unique_id) and hands its 128 bytes to the other ranks by whatever
// channel the launcher has (bench.py: a torch.distributed broadcast); every rank then calls tvmh_rccl_comm_create.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "triton_host.hpp"

namespace {
struct RcclComm {
    tvmh_comm vt;
    ncclComm_t comm = nullptr;
    int device = 0;
};
thread_local char g_error[256] = "";

int32_t fail(const char* what, ncclResult_t r) {
    std::snprintf(g_error, sizeof g_error, "%s: %s", what, ncclGetErrorString(r));
    return TVM_ERR_DEVICE;
}

int32_t rccl_all_gather(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words) {
    auto* c = (RcclComm*)self;
    if (!c->comm) return fail("ncclAllGather on an aborted communicator", ncclInvalidUsage);
    const ncclResult_t r = ncclAllGather(d_send, d_recv, words, ncclUint64, c->comm, (hipStream_t)tvm_ctx_stream(ctx));
    return r == ncclSuccess ? TVM_OK : fail("ncclAllGather", r);
}

int32_t rccl_all_to_all(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words) {
    auto* c = (RcclComm*)self;
    if (!c->comm) return fail("all-to-all on an aborted communicator", ncclInvalidUsage);
    hipStream_t stream = (hipStream_t)tvm_ctx_stream(ctx);
    ncclResult_t r = ncclGroupStart();
    for (uint32_t peer = 0; peer < c->vt.world && r == ncclSuccess; peer++) {
        r = ncclSend(d_send + (uint64_t)peer * words, words, ncclUint64, (int)peer, c->comm, stream);
        if (r == ncclSuccess) r = ncclRecv(d_recv + (uint64_t)peer * words, words, ncclUint64, (int)peer, c->comm, stream);
    }
    const ncclResult_t e = ncclGroupEnd();
    if (r == ncclSuccess) r = e;
    return r == ncclSuccess ? TVM_OK : fail("all-to-all (ncclSend / ncclRecv group)", r);
}

// tvmh_comm::all_gather_async / wait: the exchange runs on the CONTEXT's side lane (include/triton_hip.h: tvm_side_*), behind an event
// that marks the work queued on the context's stream so far (the operands), and the context's stream waits for the slot's mark only
// where the host says so -- kernels queued in between run under the exchange.  The ordering logic is the library's, the same calls
// the in-process communicator makes (sharded_host.cpp: local_all_gather_async), so the single-GPU suite exercises it.
int32_t rccl_all_gather_async(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words, uint32_t slot) {
    auto* c = (RcclComm*)self;
    if (!c->comm) return fail("all_gather_async on an aborted communicator", ncclInvalidUsage);
    if (slot >= TVM_SIDE_SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    hipStream_t side = (hipStream_t)tvm_ctx_side_stream(ctx);
    if (!side) return TVM_ERR_DEVICE;
    int32_t st = tvm_side_begin(ctx);
    if (st != TVM_OK) return st;
    const ncclResult_t r = ncclAllGather(d_send, d_recv, words, ncclUint64, c->comm, side);
    if (r != ncclSuccess) return fail("ncclAllGather (side lane)", r);
    return tvm_side_mark(ctx, slot);
}
int32_t rccl_wait(void*, tvm_ctx* ctx, uint32_t slot) { return tvm_side_wait(ctx, slot); }

// tvmh_comm::abort: this rank failed in the middle of a proof.  Its queued collectives are cancelled and the communicator is
// torn down without waiting for the peers (ncclCommDestroy would wait for them); the peers' pending collectives never complete,
// which their launcher resolves the usual way -- this rank's process reports the error and the job is torn down.
void rccl_abort(void* self) {
    auto* c = (RcclComm*)self;
    if (c->comm) (void)ncclCommAbort(c->comm);
    c->comm = nullptr;
}
}  // namespace

extern "C" const char* tvmh_rccl_last_error(void) { return g_error; }

extern "C" int32_t tvmh_rccl_unique_id(uint8_t out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    std::memcpy(out, &id, sizeof id);
    return TVM_OK;
}

extern "C" int32_t tvmh_rccl_comm_create(const uint8_t unique_id[128], uint32_t rank, uint32_t world, int32_t device, tvmh_comm** out) {
    if (!unique_id || !out || !world || rank >= world) return TVM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) {
        std::snprintf(g_error, sizeof g_error, "hipSetDevice(%d) failed", device);
        return TVM_ERR_DEVICE;
    }
    auto* c = new (std::nothrow) RcclComm();
    if (!c) return TVM_ERR_OUT_OF_MEMORY;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    const ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank", r);
    }
    c->device = device;
    c->vt = tvmh_comm{c, rank, world, rccl_all_gather, rccl_all_to_all, nullptr, nullptr, nullptr, rccl_abort, nullptr,
                      rccl_all_gather_async, rccl_wait};
    *out = &c->vt;
    return TVM_OK;
}

extern "C" void tvmh_rccl_comm_destroy(tvmh_comm* comm) {
    if (!comm) return;
    auto* c = (RcclComm*)comm->self;
    // (exchanges queued on a context's side lane belong to that context: ncclCommDestroy completes what this communicator has queued)
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}
