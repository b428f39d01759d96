// context.h -- device context shared by all entry points of libtriton_hip.so.
//
// One context per host thread that proves (the reference's prove() may run concurrently on several
// threads, /root/reference/triton-vm/src/lib.rs:522-532): a context owns one HIP stream, a cache of
// power tables (twiddles, coset scalings) and a growable scratch workspace.  Nothing here
// synchronises with the host unless an entry point has to hand data back.
#pragma once
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "field.h"
#include "triton_hip.h"  // status codes and the public C ABI (include/)

// Device tables are "row-block-major" (DESIGN.md section 2): blocks of TVM_RB = 16 consecutive rows,
// column-major inside a block, i.e. element (row, v) of a table with W base-field words per row is at
// ((row / 16) * W + v) * 16 + row % 16.  Sixteen consecutive rows of one column form one 128-byte line,
// so a wavefront whose lanes are consecutive rows reads any column as four full lines (row hashing,
// AIR evaluation, linear combinations), and the last LDE pass writes full lines.
#define TVM_RB 16
#define TVM_RB_LOG 4
TVM_HD u64 tvm_tab_idx(u64 row, u64 v, u64 W) { return ((row >> TVM_RB_LOG) * W + v) * TVM_RB + (row & (TVM_RB - 1)); }
TVM_HD u64 tvm_tab_words(u64 rows, u64 W) { return ((rows + TVM_RB - 1) / TVM_RB) * TVM_RB * W; }

struct tvm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::map<std::tuple<u64, u64, u64>, u64*> tables;  // (base, count, scale) -> device table
    std::vector<void*> scratch;                         // named scratch slots
    std::vector<size_t> scratch_bytes;
    // Caching device allocator: every device block handed out by the library (tvm_malloc, table handles)
    // comes from here and returns here; a 2^20-row proof allocates ~45 GiB of tables per prove(), and
    // hipMalloc/hipFree of that size cost hundreds of milliseconds each.  Blocks are reused in stream
    // order (one stream per context), so handing a freed block to the next request needs no host sync.
    std::multimap<size_t, void*> pool_free;             // capacity -> block
    std::map<void*, size_t> pool_live;                  // block -> capacity
    size_t pool_bytes = 0;                              // bytes held from the driver (live + cached)
    size_t pool_limit = 0;                              // tvm_ctx_set_memory_limit: 0 = whatever the device has
    bool air_valid_trace = false;                       // TVM_OPTION_AIR_VALID_TRACE (capi.hip: tvm_all_quotients_combined)
    std::string last_error;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

namespace tvm {
// t[i] = scale * base^i, i < count (Montgomery words); cached for the life of the context
const u64* pow_table(tvm_ctx* c, u64 base, u64 count, u64 scale = TVM_ONE);
// scratch slot `slot` of at least `bytes` bytes (grown on demand, contents undefined)
void* scratch(tvm_ctx* c, int slot, size_t bytes);
int set_error(tvm_ctx* c, int code, const char* what);
// pool: nullptr on device out-of-memory (after the cache has been given back to the driver and the request retried)
void* pool_alloc(tvm_ctx* c, size_t bytes);
void pool_release(tvm_ctx* c, void* p);   // back to the cache (stream-ordered reuse)
void pool_trim(tvm_ctx* c);               // cached blocks back to the driver (synchronises the stream)
// a pool block that goes back to the cache on every exit path of the function that holds it
struct PoolBlock {
    tvm_ctx* c;
    void* p = nullptr;
    explicit PoolBlock(tvm_ctx* c_) : c(c_) {}
    PoolBlock(tvm_ctx* c_, size_t bytes) : c(c_), p(pool_alloc(c_, bytes)) {}
    void* alloc(size_t bytes) {
        pool_release(c, p);
        return p = pool_alloc(c, bytes);
    }
    ~PoolBlock() { pool_release(c, p); }
    PoolBlock(const PoolBlock&) = delete;
    PoolBlock& operator=(const PoolBlock&) = delete;
};
inline int ilog2(u64 n) {
    int l = 0;
    while ((1ull << l) < n) l++;
    return l;
}
inline bool is_pow2(u64 n) { return n && !(n & (n - 1)); }
}  // namespace tvm

#define TVM_HIP_CHECK(c, expr)                                                              \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return tvm::set_error((c), e_ == hipErrorOutOfMemory ? TVM_ERR_OUT_OF_MEMORY : TVM_ERR_DEVICE, #expr); \
    } while (0)
#define TVM_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != TVM_OK) return rc_; \
    } while (0)
