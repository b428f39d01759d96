// rccl_comm.cpp -> libtriton_rccl.so: the tvmh_comm (triton_host.hpp) of the multi-GPU prover over RCCL / xGMI.
//
// One process per GPU.  The collectives of the default proof are enqueued on the CONTEXT's stream (tvm_ctx_stream), behind the
// kernels that produce their operands and ahead of those that consume their results: no host synchronisation, no second stream,
// no event.  (The opt-in column split's coefficient exchange is the exception: all_gather_async / wait below, on a stream of the
// communicator's own, so that it can run under the kernels queued after it.)
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
constexpr uint32_t SLOTS = 16;
struct RcclComm {
    tvmh_comm vt;
    ncclComm_t comm = nullptr;
    int device = 0;
    // all_gather_async: the communicator's own stream, one "operands ready" event, one "exchange done" event per slot
    hipStream_t side = nullptr;
    hipEvent_t ready = nullptr, done[SLOTS] = {};
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

// tvmh_comm::all_gather_async / wait: the exchange runs on the communicator's own stream, behind an event that marks the work queued
// on the context's stream so far (the operands), and the context's stream waits for the slot's "done" event only where the host says
// so -- kernels queued in between run under the exchange.
int32_t hip_fail(const char* what, hipError_t e) {
    std::snprintf(g_error, sizeof g_error, "%s: %s", what, hipGetErrorString(e));
    return TVM_ERR_DEVICE;
}
int32_t rccl_all_gather_async(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words, uint32_t slot) {
    auto* c = (RcclComm*)self;
    if (!c->comm) return fail("all_gather_async on an aborted communicator", ncclInvalidUsage);
    if (slot >= SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    hipStream_t compute = (hipStream_t)tvm_ctx_stream(ctx);
    hipError_t e = hipEventRecord(c->ready, compute);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->side, c->ready, 0);
    if (e != hipSuccess) return hip_fail("all_gather_async (operands)", e);
    const ncclResult_t r = ncclAllGather(d_send, d_recv, words, ncclUint64, c->comm, c->side);
    if (r != ncclSuccess) return fail("ncclAllGather (own stream)", r);
    e = hipEventRecord(c->done[slot], c->side);
    return e == hipSuccess ? TVM_OK : hip_fail("all_gather_async (done event)", e);
}
int32_t rccl_wait(void* self, tvm_ctx* ctx, uint32_t slot) {
    auto* c = (RcclComm*)self;
    if (slot >= SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    const hipError_t e = hipStreamWaitEvent((hipStream_t)tvm_ctx_stream(ctx), c->done[slot], 0);
    return e == hipSuccess ? TVM_OK : hip_fail("wait (exchange done)", e);
}

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
    bool own_stream = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess;
    for (uint32_t k = 0; k < SLOTS && own_stream; k++) own_stream = hipEventCreateWithFlags(&c->done[k], hipEventDisableTiming) == hipSuccess;
    // (without a stream of its own the communicator simply offers no asynchronous exchange: the caller uses all_gather)
    c->vt = tvmh_comm{c, rank, world, rccl_all_gather, rccl_all_to_all, nullptr, nullptr, nullptr, rccl_abort, nullptr,
                      own_stream ? rccl_all_gather_async : nullptr, own_stream ? rccl_wait : nullptr};
    *out = &c->vt;
    return TVM_OK;
}

extern "C" void tvmh_rccl_comm_destroy(tvmh_comm* comm) {
    if (!comm) return;
    auto* c = (RcclComm*)comm->self;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    for (uint32_t k = 0; k < SLOTS; k++)
        if (c->done[k]) (void)hipEventDestroy(c->done[k]);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}
