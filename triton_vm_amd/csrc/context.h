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

// Device tables are "row-block-major" (DESIGN.md section 2): blocks of TVM_RB = 16 consecutive STORAGE rows,
// column-major inside a block, i.e. element (s, v) of a table with W base-field words per row is at
// ((s / 16) * W + v) * 16 + s % 16.  Sixteen consecutive storage rows of one column form one 128-byte line,
// so a wavefront whose lanes are consecutive storage rows reads any column as four full lines (row hashing,
// AIR evaluation, linear combinations), and the last LDE pass writes full lines.
#define TVM_RB 16
#define TVM_RB_LOG 4
TVM_HD u64 tvm_tab_idx(u64 row, u64 v, u64 W) { return ((row >> TVM_RB_LOG) * W + v) * TVM_RB + (row & (TVM_RB - 1)); }
TVM_HD u64 tvm_tab_words(u64 rows, u64 W) { return ((rows + TVM_RB - 1) / TVM_RB) * TVM_RB * W; }

// Which storage row holds which row of the domain.  A table made by tvm_lde_table over the evaluation domain
// g*<w_L>, L = X*N, N = n1*n2 (the two axes of the transform, csrc/ntt.hip) is stored COSET-MAJOR and, inside a coset of
// the trace domain, in the order the last LDE pass produces it:
//     domain row  i = X*j + k,  j = j1 + n2*j2  (k < X the coset, j1 < n2, j2 < n1)   <->   storage row  k*pitch + j1*n1 + j2.
//   * one transform of the last pass (fixed k, j1; all j2) owns n1 CONSECUTIVE storage rows: it writes full lines whatever the
//     number of transforms a workgroup holds;
//   * the rows of a coset -- of a stride view (every s-th domain row = the cosets k = 0 mod s: the quotient domain inside a
//     longer LDT domain, a rank's or a coset-wise pass's share, the half domain of valid-trace mode) -- are contiguous;
//   * the successor row j + 1 of the AIR (domain row i + X) is storage row + n1; tables the AIR reads carry one more block of
//     n1 rows per coset behind the n2 blocks (`pitch` = (n2 + 1)*n1): block n2 = block 0 shifted by one row, the successors of
//     block n2 - 1.
// Tables in natural row order (the quotient-segment table, the verifier's row packs) are the case X = 1, n2 = 1, n1 = rows.
struct TabLayout {
    u64 X = 1, n1 = 0, n2 = 1, pitch = 0;   // X, n2 and -- unless n2 == 1 -- n1 are powers of two
    int log_x = 0, log_n1 = 0, log_n2 = 0;
    TVM_HD u64 rows() const { return X * n1 * n2; }
    TVM_HD u64 storage_rows() const { return X * pitch; }
    // position r = j1*n1 + j2 inside a coset -> j = j1 + n2*j2
    TVM_HD u64 coset_index(u64 r) const { return n2 == 1 ? r : (r >> log_n1) + ((r & (n1 - 1)) << log_n2); }
    TVM_HD u64 storage_row(u64 i) const {
        const u64 k = i & (X - 1), j = i >> log_x;
        return n2 == 1 ? k * pitch + j : k * pitch + (j & (n2 - 1)) * n1 + (j >> log_n2);
    }
};
inline TabLayout tab_layout_natural(u64 rows) {
    TabLayout l;
    l.n1 = l.pitch = rows;
    return l;
}
// The rows of a stride view in an order that walks storage contiguously: view row index t < rows/stride, in "coset-major"
// order when the stride selects whole cosets, in domain order otherwise.  -> storage row, and the view's domain index of it.
struct TabView {
    TabLayout l;
    u64 stride = 1, n_out = 0;
    bool by_coset = false;   // stride divides X: the view is the cosets k = stride*k', k' < X/stride
    int log_n = 0, log_xv = 0;
    TVM_HD void locate(u64 t, u64& s, u64& out_index) const {
        if (by_coset) {
            const u64 kv = t >> log_n, r = t & ((1ull << log_n) - 1);
            s = kv * stride * l.pitch + r;
            out_index = (l.coset_index(r) << log_xv) + kv;
        } else {
            s = l.storage_row(t * stride);
            out_index = t;
        }
    }
};
inline int tab_ilog2(u64 n) {
    int k = 0;
    while ((1ull << k) < n) k++;
    return k;
}
inline TabView tab_view(const TabLayout& l, u64 stride) {
    TabView v;
    v.l = l;
    v.stride = stride;
    v.n_out = l.rows() / stride;
    // (a single coset too: the table of a rank that owns one coset of the trace domain -- eight ranks at expansion 8 -- is stored
    // in the order of the last LDE pass like any other; walking it in domain order read every word of a row from a line of its own)
    v.by_coset = stride <= l.X && l.X % stride == 0 && l.n2 > 1;
    if (v.by_coset) {
        v.log_n = l.log_n1 + l.log_n2;
        v.log_xv = tab_ilog2(l.X / stride);
    }
    return v;
}

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
    int lde_chunk_columns = 0;                          // TVM_OPTION_LDE_CHUNK_COLUMNS: 0 = chosen by lde_table (ntt.hip)
    int lde_pass2_tiles = 0;                            // TVM_OPTION_LDE_PASS2_TILES: 1 = the tile kernels instead of k_lde_pass2_fused (A/B)
    u64 merkle_min_workgroups = 4096;                   // TVM_OPTION_MERKLE_MIN_WORKGROUPS (hash.hip: merkle_tree_from_leaves)
    bool merkle_subtrees = true;                        // TVM_OPTION_MERKLE_SUBTREES: narrow levels seven to a launch (k_merkle_subtrees)
    std::string last_error;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // the side lane (include/triton_hip.h: tvm_side_*): created on first use
    hipStream_t side = nullptr;
    hipEvent_t side_ready = nullptr, side_done[16] = {};
    // the staging ring (h2d_small): pinned host memory that small host arrays (points, weights, challenges, index lists) pass through
    // on their way to the device, so that the copy is asynchronous AND the caller's array is free when the entry point returns
    char* pin = nullptr;
    size_t pin_bytes = 0, pin_head = 0;
    bool pin_unavailable = false;   // hipHostMalloc refused once: the plain path (copy, then wait) from then on
    // the fork lanes (air.hip: all_quotients_combined): launches that are independent of one another and too small to fill the
    // chip each -- the parts of the AIR on a short quotient domain -- go out on these streams beside the context's own and meet
    // it again before the next dependent launch.  Created on first use.
    hipStream_t fork[3] = {};
    hipEvent_t fork_ready = nullptr, fork_done[3] = {};
    u64 air_fork_max_workgroups = 256;                  // TVM_OPTION_AIR_FORK_MAX_WORKGROUPS (0 after set_option(…, 0): never fork)
};

namespace tvm {
// the extension split at the coefficients, for the virtual columns [first_vcol, first_vcol + n_vcols) (a virtual column = one
// base-field component of a column): mode 1 = inverse transforms only, the coefficient form -> coeffs ([n_vcols][n_rows] words);
// mode 2 = forward only, from coeffs, into those columns of the table
struct LdeSplit {
    int mode;
    u64* coeffs;
    int first_vcol, n_vcols;
};
// t[i] = scale * base^i, i < count (Montgomery words); cached for the life of the context
const u64* pow_table(tvm_ctx* c, u64 base, u64 count, u64 scale = TVM_ONE);
// scratch slot `slot` of at least `bytes` bytes (grown on demand, contents undefined)
void* scratch(tvm_ctx* c, int slot, size_t bytes);
int set_error(tvm_ctx* c, int code, const char* what);
// pool: nullptr on device out-of-memory (after the cache has been given back to the driver and the request retried)
void* pool_alloc(tvm_ctx* c, size_t bytes);
void pool_release(tvm_ctx* c, void* p);   // back to the cache (stream-ordered reuse)
void pool_trim(tvm_ctx* c);               // cached blocks back to the driver (synchronises the stream)
size_t pool_available(tvm_ctx* c, size_t* device_total);
// Host array -> device, stream-ordered, WITHOUT draining the stream: the bytes are copied into the context's pinned staging ring and
// go from there (the caller's array may be a temporary: it is free on return).  The ring wraps after a stream synchronisation -- once
// in some tens of proofs; arrays above a quarter of the ring take the plain path (copy, then wait).  Before round 6 every such hand-over
// was hipMemcpyAsync + hipStreamSynchronize: some twenty drained streams per proof, a fifth of a proof of a 2^10-row trace.
int h2d_small(tvm_ctx* c, void* d, const void* h, size_t bytes);
// the context's fork lanes (three more streams and their events), created on first use; false if the driver refuses
bool fork_lanes(tvm_ctx* c);   // device free + own cache, capped by the context's limit
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
