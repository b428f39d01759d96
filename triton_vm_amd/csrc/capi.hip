// capi.hip -- the extern "C" boundary of libtriton_hip.so (declared in include/triton_hip.h).
// Argument validation and error mapping live here; kernels live in ntt.hip / hash.hip / poly.hip.
#include <cstring>
#include <new>
#include <vector>

#include "kernels.h"
#include "ntt_shift.h"

using namespace tvm;

#define TVM_ABI_VERSION 1

static bool valid_fk(int32_t fk) { return fk == 1 || fk == 3; }
static bool valid_domain(const tvm_domain& d) { return is_pow2(d.length) && d.length >= 1 && d.generator < TVM_P && d.offset < TVM_P; }

extern "C" {

int32_t tvm_abi_version(void) { return TVM_ABI_VERSION; }

const char* tvm_status_string(int32_t s) {
    switch (s) {
        case TVM_OK: return "ok";
        case TVM_ERR_INVALID_ARGUMENT: return "invalid argument";
        case TVM_ERR_OUT_OF_MEMORY: return "device out of memory";
        case TVM_ERR_DEVICE: return "HIP runtime error";
        case TVM_ERR_UNSUPPORTED: return "unsupported size or configuration";
        default: return "unknown status";
    }
}

int32_t tvm_ctx_create(int32_t device, void* hip_stream, tvm_ctx** out) {
    if (!out) return TVM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return TVM_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TVM_ERR_DEVICE;
    tvm_ctx* c = new (std::nothrow) tvm_ctx();
    if (!c) return TVM_ERR_OUT_OF_MEMORY;
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreate(&c->stream) != hipSuccess) {
            delete c;
            return TVM_ERR_DEVICE;
        }
        c->owns_stream = true;
    }
    *out = c;
    return TVM_OK;
}

void tvm_ctx_destroy(tvm_ctx* c) {
    if (!c) return;
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->tables) hipFree(kv.second);
    for (void* p : c->scratch)
        if (p) hipFree(p);
    for (auto& kv : c->pool_free) hipFree(kv.second);
    for (auto& kv : c->pool_live) hipFree(kv.first);
    if (c->ev_start) hipEventDestroy(c->ev_start);
    if (c->ev_stop) hipEventDestroy(c->ev_stop);
    if (c->side) {
        hipStreamSynchronize(c->side);
        hipStreamDestroy(c->side);
    }
    if (c->side_ready) hipEventDestroy(c->side_ready);
    for (hipStream_t s : c->fork)
        if (s) {
            hipStreamSynchronize(s);
            hipStreamDestroy(s);
        }
    if (c->fork_ready) hipEventDestroy(c->fork_ready);
    for (hipEvent_t e : c->fork_done)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : c->side_done)
        if (e) hipEventDestroy(e);
    if (c->pin) hipHostFree(c->pin);
    if (c->owns_stream) hipStreamDestroy(c->stream);
    delete c;
}

const char* tvm_last_error(const tvm_ctx* c) { return c ? c->last_error.c_str() : "null context"; }

int32_t tvm_sync(tvm_ctx* c) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}
int32_t tvm_malloc(tvm_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return TVM_ERR_INVALID_ARGUMENT;
    *d_ptr = pool_alloc(c, bytes);
    if (!*d_ptr) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "tvm_malloc");
    return TVM_OK;
}
int32_t tvm_free(tvm_ctx* c, void* d_ptr) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    pool_release(c, d_ptr);
    return TVM_OK;
}
int32_t tvm_ctx_set_memory_limit(tvm_ctx* c, size_t bytes) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    c->pool_limit = bytes;
    return TVM_OK;
}
int32_t tvm_ctx_memory_held(const tvm_ctx* c, size_t* bytes) {
    if (!c || !bytes) return TVM_ERR_INVALID_ARGUMENT;
    *bytes = c->pool_bytes;
    return TVM_OK;
}
int32_t tvm_ctx_memory_info(const tvm_ctx* c, size_t* available_bytes, size_t* device_total_bytes) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    int cur = -1;
    (void)hipGetDevice(&cur);
    size_t total = 0;
    const size_t avail = pool_available(const_cast<tvm_ctx*>(c), &total);   // (binds the context's device; reads the pool only)
    if (cur >= 0 && cur != c->device) (void)hipSetDevice(cur);
    if (!total) return TVM_ERR_DEVICE;
    if (available_bytes) *available_bytes = avail;
    if (device_total_bytes) *device_total_bytes = total;
    return TVM_OK;
}
int32_t tvm_ctx_set_option(tvm_ctx* c, int32_t option, uint64_t value) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    if (option == TVM_OPTION_AIR_VALID_TRACE) {
        c->air_valid_trace = value != 0;
        return TVM_OK;
    }
    if (option == TVM_OPTION_LDE_CHUNK_COLUMNS) {
        if (value > 4096) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "TVM_OPTION_LDE_CHUNK_COLUMNS: at most 4096");
        c->lde_chunk_columns = (int)value;
        return TVM_OK;
    }
    if (option == TVM_OPTION_LDE_PASS2_TILES) {
        c->lde_pass2_tiles = value ? 1 : 0;
        return TVM_OK;
    }
    if (option == TVM_OPTION_MERKLE_SUBTREES) {
        c->merkle_subtrees = value != 0;
        return TVM_OK;
    }
    if (option == TVM_OPTION_AIR_FORK_MAX_WORKGROUPS) {
        c->air_fork_max_workgroups = value;
        return TVM_OK;
    }
    if (option == TVM_OPTION_MERKLE_MIN_WORKGROUPS) {
        c->merkle_min_workgroups = value ? value : 4096;
        return TVM_OK;
    }
    return set_error(c, TVM_ERR_INVALID_ARGUMENT, "unknown option");
}
int32_t tvm_ctx_trim(tvm_ctx* c) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    pool_trim(c);
    return TVM_OK;
}
int32_t tvm_memcpy_h2d(tvm_ctx* c, void* d, const void* h, size_t bytes) {
    if (!c || (bytes && (!d || !h))) return TVM_ERR_INVALID_ARGUMENT;
    TVM_HIP_CHECK(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}
int32_t tvm_memcpy_d2h(tvm_ctx* c, void* h, const void* d, size_t bytes) {
    if (!c || (bytes && (!d || !h))) return TVM_ERR_INVALID_ARGUMENT;
    TVM_HIP_CHECK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}
int32_t tvm_memcpy_d2d(tvm_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return TVM_ERR_INVALID_ARGUMENT;
    if (bytes) TVM_HIP_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return TVM_OK;
}
void* tvm_ctx_stream(const tvm_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- the side lane (include/triton_hip.h)
static_assert(TVM_SIDE_SLOTS == 16, "tvm_ctx::side_done has sixteen slots");
static bool side_lane(tvm_ctx* c) {
    if (c->side) return true;
    int cur = -1;   // streams belong to the device that is current when they are created (see bind_device, ntt.hip)
    if ((hipGetDevice(&cur) != hipSuccess || cur != c->device) && hipSetDevice(c->device) != hipSuccess) return false;
    hipStream_t s = nullptr;
    hipEvent_t ready = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess) {
        hipStreamDestroy(s);
        return false;
    }
    c->side = s;
    c->side_ready = ready;
    return true;
}
void* tvm_ctx_side_stream(tvm_ctx* c) { return c && side_lane(c) ? (void*)c->side : nullptr; }
int32_t tvm_side_begin(tvm_ctx* c) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    if (!side_lane(c)) return tvm::set_error(c, TVM_ERR_DEVICE, "tvm_side_begin: no second stream");
    TVM_HIP_CHECK(c, hipEventRecord(c->side_ready, c->stream));
    TVM_HIP_CHECK(c, hipStreamWaitEvent(c->side, c->side_ready, 0));
    return TVM_OK;
}
int32_t tvm_side_memcpy_d2d(tvm_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return TVM_ERR_INVALID_ARGUMENT;
    if (!side_lane(c)) return tvm::set_error(c, TVM_ERR_DEVICE, "tvm_side_memcpy_d2d: no second stream");
    if (bytes) TVM_HIP_CHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->side));
    return TVM_OK;
}
int32_t tvm_side_mark(tvm_ctx* c, uint32_t slot) {
    if (!c || slot >= TVM_SIDE_SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    if (!side_lane(c)) return tvm::set_error(c, TVM_ERR_DEVICE, "tvm_side_mark: no second stream");
    if (!c->side_done[slot]) TVM_HIP_CHECK(c, hipEventCreateWithFlags(&c->side_done[slot], hipEventDisableTiming));
    TVM_HIP_CHECK(c, hipEventRecord(c->side_done[slot], c->side));
    return TVM_OK;
}
int32_t tvm_side_wait(tvm_ctx* c, uint32_t slot) {
    if (!c || slot >= TVM_SIDE_SLOTS) return TVM_ERR_INVALID_ARGUMENT;
    if (c->side_done[slot]) TVM_HIP_CHECK(c, hipStreamWaitEvent(c->stream, c->side_done[slot], 0));   // (never marked: nothing to wait for)
    return TVM_OK;
}
int32_t tvm_side_sync(tvm_ctx* c) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    if (c->side) TVM_HIP_CHECK(c, hipStreamSynchronize(c->side));
    return TVM_OK;
}

int32_t tvm_timer_start(tvm_ctx* c) {
    if (!c) return TVM_ERR_INVALID_ARGUMENT;
    if (!c->ev_start) {
        TVM_HIP_CHECK(c, hipEventCreate(&c->ev_start));
        TVM_HIP_CHECK(c, hipEventCreate(&c->ev_stop));
    }
    TVM_HIP_CHECK(c, hipEventRecord(c->ev_start, c->stream));
    return TVM_OK;
}
int32_t tvm_timer_stop(tvm_ctx* c, float* ms) {
    if (!c || !ms || !c->ev_start) return TVM_ERR_INVALID_ARGUMENT;
    TVM_HIP_CHECK(c, hipEventRecord(c->ev_stop, c->stream));
    TVM_HIP_CHECK(c, hipEventSynchronize(c->ev_stop));
    TVM_HIP_CHECK(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return TVM_OK;
}
}  // extern "C"

namespace tvm {
// splitmix64 of (seed, index), reduced into [0, p): synthetic benchmark data only
__global__ void k_synthetic_fill(u64* __restrict__ d, u64 n, u64 seed) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    d[i] = z >= TVM_P ? z - TVM_P : z;
}
}  // namespace tvm

extern "C" {
int32_t tvm_synthetic_fill(tvm_ctx* c, uint64_t* d, uint64_t n, uint64_t seed) {
    if (!c || (n && !d)) return TVM_ERR_INVALID_ARGUMENT;
    if (!n) return TVM_OK;
    TVM_LAUNCH(tvm::k_synthetic_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d, n, seed);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// ---------------------------------------------------------------------------------- field self-check
}  // extern "C"
namespace tvm {
// out[i] = a[i] op b[i] through the device arithmetic of field.h (op: 0 add, 1 sub, 2 mul, 3 a^7, 4: a * 2^b for a plain
// exponent b < 192 through the shift forms of ntt_shift.h)
template <int S>
TVM_D u64 mul_pow2_dispatch(u64 x, int s) {
    if constexpr (S < 0) return 0;
    else return s == S ? bfe_mul_pow2<S>(x) : mul_pow2_dispatch<S - 1>(x, s);
}
__global__ void k_field_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 x = a[i], y = b[i];
    u64 r;
    if (op == 0) r = bfe_add(x, y);
    else if (op == 1) r = bfe_sub(x, y);
    else if (op == 2) r = bfe_mul(x, y);
    else if (op == 3) r = bfe_mul(bfe_mul(bfe_sqr(bfe_sqr(x)), bfe_sqr(x)), x);
    else {
        const int s = (int)(y % 192);
        r = mul_pow2_dispatch<95>(x, s % 96);
        if (s >= 96) r = bfe_neg(r);
    }
    out[i] = r;
}
// the in-register power-of-two-twiddle transforms of ntt_shift.h on consecutive groups of 2^K words
template <int K, bool DIT, bool INVERSE>
__global__ void k_pow2_points(const u64* __restrict__ a, u64* __restrict__ out, u64 n_groups) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    u64 x[1 << K];
#pragma unroll
    for (int e = 0; e < (1 << K); e++) x[e] = a[(g << K) + e];
    ntt_pow2_points<K, DIT, INVERSE>(x);
#pragma unroll
    for (int e = 0; e < (1 << K); e++) out[(g << K) + e] = x[e];
}
template <int K>
static void launch_pow2_points(tvm_ctx* c, int dit, int inverse, const u64* a, u64* out, u64 n_groups) {
    const dim3 grid((unsigned)((n_groups + 63) / 64)), block(64);
    if (dit && inverse) TVM_LAUNCH((k_pow2_points<K, true, true>), grid, block, 0, c->stream, a, out, n_groups);
    else if (dit) TVM_LAUNCH((k_pow2_points<K, true, false>), grid, block, 0, c->stream, a, out, n_groups);
    else if (inverse) TVM_LAUNCH((k_pow2_points<K, false, true>), grid, block, 0, c->stream, a, out, n_groups);
    else TVM_LAUNCH((k_pow2_points<K, false, false>), grid, block, 0, c->stream, a, out, n_groups);
}
}  // namespace tvm
extern "C" {
int32_t tvm_field_op(tvm_ctx* c, int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n) {
    const bool points = op >= 32 && op < 32 + 5 * 4;  // 32 + 4 K + 2 dit + inverse: transforms of 2^K points, K = 1 .. 4
    if (!c || op < 0 || (op > 4 && !points) || (n && (!d_a || !d_b || !d_out))) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_field_op arguments");
    if (!n) return TVM_OK;
    if (points) {
        const int K = (op - 32) >> 2, dit = (op >> 1) & 1, inverse = op & 1;
        if (K < 1 || K > 4 || n % (1ull << K)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_field_op: whole groups of 2^K words");
        if (K == 1) tvm::launch_pow2_points<1>(c, dit, inverse, d_a, d_out, n >> 1);
        else if (K == 2) tvm::launch_pow2_points<2>(c, dit, inverse, d_a, d_out, n >> 2);
        else if (K == 3) tvm::launch_pow2_points<3>(c, dit, inverse, d_a, d_out, n >> 3);
        else tvm::launch_pow2_points<4>(c, dit, inverse, d_a, d_out, n >> 4);
        TVM_HIP_CHECK(c, hipGetLastError());
        return TVM_OK;
    }
    TVM_LAUNCH(tvm::k_field_op, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (int)op, d_a, d_b, d_out, n);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

// ---------------------------------------------------------------------------------- NTT family
int32_t tvm_ntt(tvm_ctx* c, int32_t fk, uint64_t* d, uint64_t n, uint64_t gen) {
    if (!c || !d || !valid_fk(fk) || !is_pow2(n)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_ntt arguments");
    if (n == 1) return TVM_OK;
    return ntt_columns(c, d, n, fk, 0, d, fk, 0, 1, 0, fk, n, gen, TVM_ONE, TVM_ONE, TVM_ONE);
}
int32_t tvm_intt(tvm_ctx* c, int32_t fk, uint64_t* d, uint64_t n, uint64_t gen) {
    if (!c || !d || !valid_fk(fk) || !is_pow2(n)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_intt arguments");
    if (n == 1) return TVM_OK;
    return ntt_columns(c, d, n, fk, 0, d, fk, 0, 1, 0, fk, n, bfe_inv(gen), TVM_ONE, TVM_ONE, bfe_inv(bfe_from_u64(n)));
}

int32_t tvm_interpolate(tvm_ctx* c, int32_t fk, const uint64_t* d_values, tvm_domain dom, uint64_t* d_coeffs) {
    if (!c || !d_values || !d_coeffs || !valid_fk(fk) || !valid_domain(dom))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_interpolate arguments");
    const u64 n = dom.length;
    if (n == 1) {
        TVM_HIP_CHECK(c, hipMemcpyAsync(d_coeffs, d_values, fk * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        return TVM_OK;
    }
    // fast_coset_interpolate: iNTT, then coefficient i times offset^-i
    return ntt_columns(c, d_values, n, fk, 0, d_coeffs, fk, 0, 1, 0, fk, n, bfe_inv(dom.generator), TVM_ONE,
                       bfe_inv(dom.offset), bfe_inv(bfe_from_u64(n)));
}

}  // extern "C"

namespace tvm {
// c'[i] = sum_k c[i + k*len] * offset^(k*len): reduction modulo X^len - offset^len, which leaves the
// values on the coset unchanged (arithmetic_domain.rs:153-167 computes the same sum chunk-wise).
__global__ void k_fold_chunks(const u64* __restrict__ co, u64 n_coeffs, int fk, u64 len, u64 offset_pow_len,
                              u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= len * (u64)fk) return;
    const u64 i = e / fk;
    const int comp = (int)(e % fk);
    u64 acc = 0, s = TVM_ONE;
    for (u64 j = i; j < n_coeffs; j += len) {
        acc = bfe_add(acc, bfe_mul(co[j * fk + comp], s));
        s = bfe_mul(s, offset_pow_len);
    }
    out[e] = acc;
}
}  // namespace tvm

extern "C" {

int32_t tvm_evaluate(tvm_ctx* c, int32_t fk, const uint64_t* d_coeffs, uint64_t n_coeffs, tvm_domain dom,
                     uint64_t* d_values) {
    if (!c || !d_values || (n_coeffs && !d_coeffs) || !valid_fk(fk) || !valid_domain(dom))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_evaluate arguments");
    const u64 L = dom.length;
    if (n_coeffs == 0) {
        TVM_HIP_CHECK(c, hipMemsetAsync(d_values, 0, L * fk * sizeof(u64), c->stream));
        return TVM_OK;
    }
    const u64* co = d_coeffs;
    if (n_coeffs > L) {
        u64* folded = (u64*)scratch(c, 3, L * fk * sizeof(u64));
        if (!folded) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "evaluate scratch");
        const u64 total = L * fk;
        TVM_LAUNCH(k_fold_chunks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_coeffs, n_coeffs,
                   (int)fk, L, bfe_pow(dom.offset, L), folded);
        co = folded;
        n_coeffs = L;
    }
    if (L == 1) {  // a single point: the constant term survives
        TVM_HIP_CHECK(c, hipMemcpyAsync(d_values, co, fk * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        return TVM_OK;
    }
    // Evaluate on X = L/M cosets of length M >= n_coeffs instead of zero-padding to L.
    u64 M = 2;
    while (M < n_coeffs) M <<= 1;
    const u64 X = L / M;
    if (X >= 4 && L <= (1ull << 22))
        // Many short cosets (the later rounds of STIR evaluate a polynomial of 2^15 ... 2^9 coefficients on domains 8 ... 256
        // times as long): one zero-padded transform of length L -- (log L / log M) times the butterflies, but two launches
        // instead of 2 X, each of which costs 20-40 us whatever its size (round 4: 217 transform pairs per STIR proof, 11.8 ms).
        return ntt_columns(c, co, n_coeffs, fk, 0, d_values, fk, 0, 1, 0, fk, L, dom.generator, dom.offset, TVM_ONE, TVM_ONE);
    const u64 gen_m = bfe_pow(dom.generator, X);
    for (u64 k = 0; k < X; k++) {
        const u64 off = bfe_mul(dom.offset, bfe_pow(dom.generator, k));
        TVM_TRY(ntt_columns(c, co, n_coeffs, fk, 0, d_values, fk, 0, X, k, fk, M, gen_m, off, TVM_ONE, TVM_ONE));
    }
    return TVM_OK;
}

// ---------------------------------------------------------------------------------- master-table LDE
int32_t tvm_lde_table(tvm_ctx* c, int32_t fk, const uint64_t* d_trace, uint64_t n_rows, uint64_t n_cols,
                      const uint64_t* d_rnd, uint64_t h, tvm_domain trace_dom, tvm_domain eval_dom, tvm_table** out) {
    if (!c || !out) return TVM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!d_trace || (h && !d_rnd) || !valid_fk(fk) || !valid_domain(trace_dom) || !valid_domain(eval_dom) ||
        trace_dom.length != n_rows || n_cols == 0 || n_cols * fk > (1u << 20))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table arguments");
    if (trace_dom.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    tvm_table* t = new (std::nothrow) tvm_table();
    if (!t) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "table handle");
    t->rows = eval_dom.length;
    t->layout = lde_table_layout(n_rows, eval_dom.length);  // coset-major, one successor block per coset (context.h)
    t->has_successor_blocks = true;
    t->n_cols = n_cols;
    t->fk = fk;
    t->W = (int)(n_cols * fk);
    t->interpolant_len = n_rows + h;  // randomized_column_interpolant: degree < n_rows + h (master_table.rs:392-403)
    t->data = (u64*)pool_alloc(c, t->bytes());
    if (!t->data) {
        delete t;
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "LDE table allocation");
    }
    if (t->layout.storage_rows() % TVM_RB) {  // zero the padding rows of the last (partial) row block
        const u64 full = t->layout.storage_rows() / TVM_RB * TVM_RB * (u64)t->W;
        (void)hipMemsetAsync(t->data + full, 0, t->bytes() - full * sizeof(u64), c->stream);
    }
    int rc = lde_table(c, fk, d_trace, n_rows, n_cols, d_rnd, h, trace_dom.generator, eval_dom.offset, eval_dom.generator,
                       eval_dom.length, t->data, 0);
    if (rc == TVM_OK) rc = fill_successor_blocks(c, t->data, t->layout, t->W);
    if (rc != TVM_OK) {
        pool_release(c, t->data);
        delete t;
        return rc;
    }
    *out = t;
    return TVM_OK;
}

// The same extension, split at the coefficients (the column sharding: SURVEY 8(e)).
int32_t tvm_lde_column_coefficients(tvm_ctx* c, int32_t fk, const uint64_t* d_trace, uint64_t n_rows, uint64_t n_cols, tvm_domain trace_dom,
                                    uint64_t first_virtual_column, uint64_t n_virtual_columns, uint64_t* d_coeffs) {
    if (!c || !d_trace || !d_coeffs || !valid_fk(fk) || !valid_domain(trace_dom) || trace_dom.length != n_rows || n_cols == 0 ||
        n_cols * fk > (1u << 20) || first_virtual_column > n_cols * fk || n_virtual_columns > n_cols * fk - first_virtual_column)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_column_coefficients arguments");
    if (trace_dom.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    if (!n_virtual_columns) return TVM_OK;
    const LdeSplit split{1, d_coeffs, (int)first_virtual_column, (int)n_virtual_columns};
    return lde_table(c, fk, d_trace, n_rows, n_cols, nullptr, 0, trace_dom.generator, TVM_ONE, trace_dom.generator, n_rows, nullptr, 0, &split);
}

int32_t tvm_lde_table_begin(tvm_ctx* c, int32_t fk, uint64_t n_rows, uint64_t n_cols, uint64_t h, tvm_domain trace_dom, tvm_domain eval_dom,
                            tvm_table** out) {
    if (!c || !out) return TVM_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!valid_fk(fk) || !valid_domain(trace_dom) || !valid_domain(eval_dom) || trace_dom.length != n_rows || n_cols == 0 ||
        n_cols * fk > (1u << 20) || eval_dom.length < n_rows)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_begin arguments");
    if (trace_dom.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    tvm_table* t = new (std::nothrow) tvm_table();
    if (!t) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "table handle");
    t->rows = eval_dom.length;
    t->layout = lde_table_layout(n_rows, eval_dom.length);
    t->has_successor_blocks = true;
    t->n_cols = n_cols;
    t->fk = fk;
    t->W = (int)(n_cols * fk);
    t->interpolant_len = n_rows + h;
    t->data = (u64*)pool_alloc(c, t->bytes());
    if (!t->data) {
        delete t;
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "LDE table allocation");
    }
    if (t->layout.storage_rows() % TVM_RB) {
        const u64 full = t->layout.storage_rows() / TVM_RB * TVM_RB * (u64)t->W;
        (void)hipMemsetAsync(t->data + full, 0, t->bytes() - full * sizeof(u64), c->stream);
    }
    t->lde_split_open = true;
    t->lde_trace_len = n_rows;
    t->lde_trace_gen = trace_dom.generator;
    t->lde_eval_offset = eval_dom.offset;
    t->lde_eval_gen = eval_dom.generator;
    t->lde_written.assign((size_t)t->W, 0);
    *out = t;
    return TVM_OK;
}

int32_t tvm_lde_table_add_columns(tvm_ctx* c, tvm_table* t, const uint64_t* d_coeffs, uint64_t first_virtual_column,
                                  uint64_t n_virtual_columns, const uint64_t* d_rnd, uint64_t h, tvm_domain trace_dom, tvm_domain eval_dom) {
    if (!c || !t || !d_coeffs || (h && !d_rnd) || !valid_domain(trace_dom) || !valid_domain(eval_dom))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_add_columns arguments");
    // only a table that tvm_lde_table_begin made and tvm_lde_table_end has not closed (its layout has the successor blocks pass 3
    // and fill_successor_blocks write into), and only with the domains that were given to begin
    if (!t->lde_split_open || !t->has_successor_blocks)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_add_columns: not a table under construction (tvm_lde_table_begin)");
    if (trace_dom.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    if (trace_dom.length != t->lde_trace_len || trace_dom.generator != t->lde_trace_gen || eval_dom.length != t->rows ||
        eval_dom.offset != t->lde_eval_offset || eval_dom.generator != t->lde_eval_gen || t->interpolant_len != trace_dom.length + h)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_add_columns: other domains / randomizer count than tvm_lde_table_begin's");
    const uint64_t W = (uint64_t)t->W;
    if (first_virtual_column > W || n_virtual_columns > W - first_virtual_column)   // (no wrap-around in the sum)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_add_columns: columns beyond the table");
    if (!n_virtual_columns) return TVM_OK;
    const LdeSplit split{2, const_cast<uint64_t*>(d_coeffs), (int)first_virtual_column, (int)n_virtual_columns};
    const int rc = lde_table(c, t->fk, nullptr, trace_dom.length, t->n_cols, d_rnd, h, trace_dom.generator, eval_dom.offset, eval_dom.generator,
                             eval_dom.length, t->data, 0, &split);
    if (rc == TVM_OK)
        for (uint64_t v = first_virtual_column; v < first_virtual_column + n_virtual_columns; v++) t->lde_written[(size_t)v] = 1;
    return rc;
}

int32_t tvm_lde_table_end(tvm_ctx* c, tvm_table* t) {
    if (!c || !t) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_end arguments");
    if (!t->lde_split_open || !t->has_successor_blocks)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_end: not a table under construction (tvm_lde_table_begin)");
    for (size_t v = 0; v < t->lde_written.size(); v++)
        if (!t->lde_written[v]) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_lde_table_end: a column was never written (tvm_lde_table_add_columns)");
    const int rc = fill_successor_blocks(c, t->data, t->layout, t->W);
    if (rc == TVM_OK) {
        t->lde_split_open = false;
        t->lde_written.clear();
    }
    return rc;
}

void tvm_table_free(tvm_ctx* c, tvm_table* t) {
    if (!t) return;
    if (c && t->data) pool_release(c, t->data);
    delete t;
}
uint64_t tvm_table_num_rows(const tvm_table* t) { return t ? t->rows : 0; }
uint64_t tvm_table_num_columns(const tvm_table* t) { return t ? t->n_cols : 0; }
int32_t tvm_table_field_kind(const tvm_table* t) { return t ? t->fk : 0; }

int32_t tvm_table_export_row_major(tvm_ctx* c, const tvm_table* t, uint64_t* d_out) {
    if (!c || !t || !d_out) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "export arguments");
    return table_to_row_major(c, t->data, t->layout, t->W, d_out);
}

int32_t tvm_table_reveal_rows(tvm_ctx* c, const tvm_table* t, uint64_t ldt_length, const uint64_t* h_idx, uint64_t n,
                              uint64_t* h_out) {
    if (!c || !t || (n && (!h_idx || !h_out)) || !is_pow2(ldt_length) || ldt_length > t->rows)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "reveal_rows arguments");
    if (!n) return TVM_OK;
    const u64 stride = t->rows / ldt_length;
    std::vector<u64> idx(n);
    for (u64 j = 0; j < n; j++) {
        if (h_idx[j] >= ldt_length) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "reveal_rows: index out of range");
        idx[j] = h_idx[j] * stride;
    }
    u64* d_idx = (u64*)scratch(c, 4, n * sizeof(u64));
    u64* d_out = (u64*)scratch(c, 5, n * t->W * sizeof(u64));
    if (!d_idx || !d_out) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "reveal scratch");
    TVM_TRY(h2d_small(c, d_idx, idx.data(), n * sizeof(u64)));   // (idx is a local)
    TVM_TRY(gather_rows(c, t->data, t->layout, t->W, d_idx, n, d_out));
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_out, d_out, n * t->W * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

// ---------------------------------------------------------------------------------- hashing
int32_t tvm_hash_rows(tvm_ctx* c, const tvm_table* t, uint64_t ldt_length, uint64_t* d_digests) {
    if (!c || !t || !d_digests || !is_pow2(ldt_length) || ldt_length > t->rows)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "hash_rows arguments");
    return hash_rows(c, t->data, t->layout, t->W, t->rows / ldt_length, d_digests);
}
int32_t tvm_merkle_tree(tvm_ctx* c, const uint64_t* d_leaves, uint64_t n, uint64_t* d_nodes) {
    if (!c || !d_leaves || !d_nodes || !is_pow2(n)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "merkle arguments");
    TVM_HIP_CHECK(c, hipMemcpyAsync(d_nodes + 5 * n, d_leaves, 5 * n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    return merkle_tree_from_leaves(c, d_nodes, n);
}
int32_t tvm_table_merkle_tree(tvm_ctx* c, const tvm_table* t, uint64_t ldt_length, uint64_t* d_nodes) {
    if (!c || !t || !d_nodes || !is_pow2(ldt_length) || ldt_length > t->rows)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "table_merkle_tree arguments");
    TVM_TRY(hash_rows(c, t->data, t->layout, t->W, t->rows / ldt_length, d_nodes + 5 * ldt_length));
    return merkle_tree_from_leaves(c, d_nodes, ldt_length);
}
}  // extern "C"

namespace tvm {
__global__ void k_xfe_aos_leaves(const u64* __restrict__ cw, u64 n, u64* __restrict__ leaves) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    leaves[5 * i + 0] = cw[3 * i];
    leaves[5 * i + 1] = cw[3 * i + 1];
    leaves[5 * i + 2] = cw[3 * i + 2];
    leaves[5 * i + 3] = 0;
    leaves[5 * i + 4] = 0;
}
}  // namespace tvm

extern "C" int32_t tvm_codeword_merkle_tree(tvm_ctx* c, const uint64_t* d_cw, uint64_t n, uint64_t* d_nodes) {
    if (!c || !d_cw || !d_nodes || !is_pow2(n)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "codeword tree arguments");
    TVM_LAUNCH(tvm::k_xfe_aos_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_cw, n, d_nodes + 5 * n);
    return merkle_tree_from_leaves(c, d_nodes, n);
}

// ---------------------------------------------------------------------------------- combination / DEEP / FRI
namespace tvm {
// small host arrays (points, weights) staged into a context scratch slot
static const u64* stage_small(tvm_ctx* c, int slot, const u64* h, size_t words) {
    u64* d = (u64*)scratch(c, slot, (words ? words : 1) * sizeof(u64));
    if (!d) return nullptr;
    if (h2d_small(c, d, h, words * sizeof(u64)) != TVM_OK) return nullptr;   // (h may be a caller temporary)
    return d;
}
// coeffs[i + N * r] += sum_k w[3 * r + k] * q[k][i]  (r, k < 3; i < N; XFE vectors q, base-field weights w): the polynomial
// A + X^N B + X^2N C from its restrictions Q_k = A + c_k B + c_k^2 C to three cosets (w = the inverse Vandermonde matrix)
struct ThreeCosetWeights {
    u64 w[9];
};
// coeffs[i * N + t] += sum_j w[4 i + j] * q_j[t]  (i, j < n <= 4; t < N; XFE vectors q_j, base-field weights): the polynomial
// sum_i X^(iN) A_i from its restrictions Q_j = sum_i c_j^i A_i to n cosets (w = the inverse Vandermonde matrix)
struct CosetCombineWeights {
    u64 w[16];
};
struct CosetCombinePointers {
    const u64* q[4];
};
__global__ void k_coset_combine(CosetCombinePointers p, int n, u64 N, CosetCombineWeights m, u64* __restrict__ coeffs) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    xfe v[4];
    for (int j = 0; j < n; j++) v[j] = xfe_make(p.q[j][3 * t], p.q[j][3 * t + 1], p.q[j][3 * t + 2]);
    for (int i = 0; i < n; i++) {
        u64* o = coeffs + 3 * ((u64)i * N + t);
        xfe acc = xfe_make(o[0], o[1], o[2]);
        for (int j = 0; j < n; j++) acc = xfe_add(acc, xfe_mul_bfe(v[j], m.w[4 * i + j]));
        o[0] = acc.c0, o[1] = acc.c1, o[2] = acc.c2;
    }
}
__global__ void k_three_coset_combine(const u64* __restrict__ q0, const u64* __restrict__ q1, const u64* __restrict__ q2, u64 n,
                                      ThreeCosetWeights m, u64* __restrict__ coeffs) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const xfe a = xfe_make(q0[3 * i], q0[3 * i + 1], q0[3 * i + 2]), b = xfe_make(q1[3 * i], q1[3 * i + 1], q1[3 * i + 2]),
              c = xfe_make(q2[3 * i], q2[3 * i + 1], q2[3 * i + 2]);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const xfe v = xfe_add(xfe_add(xfe_mul_bfe(a, m.w[3 * r]), xfe_mul_bfe(b, m.w[3 * r + 1])), xfe_mul_bfe(c, m.w[3 * r + 2]));
        u64* p = coeffs + 3 * (i + n * r);
        p[0] = bfe_add(p[0], v.c0);
        p[1] = bfe_add(p[1], v.c1);
        p[2] = bfe_add(p[2], v.c2);
    }
}
// out[i] = sum_k w[k] * v[k * stride + i]  (XFE vectors, XFE weights; k < 8)
__global__ void k_xfe_linear_combination(const u64* __restrict__ v, int n_vectors, u64 stride, u64 n, const u64* __restrict__ w,
                                         u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    xfe acc = xfe_zero();
    for (int k = 0; k < n_vectors; k++) {
        const u64* e = v + 3 * ((u64)k * stride + i);
        acc = xfe_add(acc, xfe_mul(xfe_make(e[0], e[1], e[2]), xfe_make(w[3 * k], w[3 * k + 1], w[3 * k + 2])));
    }
    out[3 * i] = acc.c0, out[3 * i + 1] = acc.c1, out[3 * i + 2] = acc.c2;
}
__global__ void k_xfe_add_assign(u64* __restrict__ a, const u64* __restrict__ b, u64 n_words) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) a[i] = bfe_add(a[i], b[i]);
}
}  // namespace tvm

extern "C" {

int32_t tvm_out_of_domain_rows(tvm_ctx* c, int32_t fk, const uint64_t* d_trace, uint64_t n, uint64_t n_cols,
                               const uint64_t* d_rnd, uint64_t h, tvm_domain td, const uint64_t* h_points, uint32_t n_points,
                               uint64_t* h_rows) {
    if (!c || !d_trace || (h && !d_rnd) || !valid_fk(fk) || !valid_domain(td) || td.length != n || !h_points || !h_rows ||
        !n_points || !n_cols)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "out_of_domain_rows arguments");
    if (td.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    const u64* d_points = stage_small(c, 9, h_points, 3 * (size_t)n_points);
    u64* d_rows = (u64*)scratch(c, 10, (size_t)n_points * n_cols * 3 * sizeof(u64));
    if (!d_points || !d_rows) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "ood staging");
    TVM_TRY(out_of_domain_rows(c, fk, d_trace, n, n_cols, d_rnd, h, td.generator, d_points, (int)n_points, d_rows));
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_rows, d_rows, (size_t)n_points * n_cols * 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

int32_t tvm_weighted_sum_of_columns(tvm_ctx* c, int32_t fk, const uint64_t* d_trace, uint64_t n, uint64_t n_cols,
                                    const uint64_t* d_rnd, uint64_t h, tvm_domain td, const uint64_t* h_weights,
                                    uint64_t* d_poly) {
    if (!c || !d_trace || (h && !d_rnd) || !valid_fk(fk) || !valid_domain(td) || td.length != n || !h_weights || !d_poly ||
        !n_cols || h > n)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "weighted_sum_of_columns arguments");
    if (td.offset != TVM_ONE) return set_error(c, TVM_ERR_UNSUPPORTED, "trace domain offset must be 1");
    const u64* d_w = stage_small(c, 9, h_weights, 3 * (size_t)n_cols);
    if (!d_w) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "weights staging");
    TVM_TRY(weighted_row_sum(c, fk, d_trace, n, n_cols, d_w, 0, d_poly));
    TVM_HIP_CHECK(c, hipMemsetAsync(d_poly + 3 * n, 0, 3 * n * sizeof(u64), c->stream));
    if (n > 1)
        TVM_TRY(ntt_columns(c, d_poly, n, 3, 0, d_poly, 3, 0, 1, 0, 3, n, bfe_inv(td.generator), TVM_ONE, TVM_ONE,
                            bfe_inv(bfe_from_u64(n))));
    return randomizer_contribution(c, fk, d_rnd, n, n_cols, h, d_w, d_poly);
}

int32_t tvm_xfe_add_assign(tvm_ctx* c, uint64_t* d_a, const uint64_t* d_b, uint64_t n) {
    if (!c || (n && (!d_a || !d_b))) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "xfe_add_assign arguments");
    if (!n) return TVM_OK;
    TVM_LAUNCH(tvm::k_xfe_add_assign, dim3((unsigned)((3 * n + 255) / 256)), dim3(256), 0, c->stream, d_a, d_b, 3 * n);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int32_t tvm_xfe_linear_combination(tvm_ctx* c, const uint64_t* d_vectors, uint32_t n_vectors, uint64_t stride, uint64_t n,
                                   const uint64_t* h_weights, uint64_t* d_out) {
    if (!c || !d_vectors || !h_weights || !d_out || !n_vectors || n_vectors > 64 || stride < n)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "xfe_linear_combination arguments");
    if (!n) return TVM_OK;
    const u64* d_w = stage_small(c, 9, h_weights, 3 * (size_t)n_vectors);
    if (!d_w) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "weights staging");
    TVM_LAUNCH(tvm::k_xfe_linear_combination, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_vectors, (int)n_vectors, stride, n,
               d_w, d_out);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

int32_t tvm_evaluate_at_points(tvm_ctx* c, const uint64_t* d_coeffs, uint64_t n, const uint64_t* h_points, uint32_t n_points,
                               uint64_t* h_out) {
    if (!c || (n && !d_coeffs) || !h_points || !h_out || !n_points)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "evaluate_at_points arguments");
    const u64* d_points = stage_small(c, 9, h_points, 3 * (size_t)n_points);
    u64* d_out = (u64*)scratch(c, 10, (size_t)n_points * 3 * sizeof(u64));
    if (!d_points || !d_out) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "points staging");
    TVM_TRY(poly_eval(c, d_coeffs, n, d_points, (int)n_points, d_out));
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_out, d_out, (size_t)n_points * 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

int32_t tvm_evaluate_polys_at_points(tvm_ctx* c, const uint64_t* d_coeffs, uint64_t n, uint64_t stride, uint32_t n_polys,
                                     const uint64_t* h_points, uint32_t n_points, uint64_t* h_out) {
    if (!c || (n && !d_coeffs) || !h_points || !h_out || !n_points || !n_polys || n_polys > 64 || stride < n)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "evaluate_polys_at_points arguments");
    const u64* d_points = stage_small(c, 9, h_points, 3 * (size_t)n_points);
    u64* d_out = (u64*)scratch(c, 10, (size_t)n_polys * n_points * 3 * sizeof(u64));
    if (!d_points || !d_out) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "points staging");
    // (the partial sums of poly_eval share one scratch slot: the evaluations follow one another on the stream)
    for (uint32_t p = 0; p < n_polys; p++)
        TVM_TRY(poly_eval(c, n ? d_coeffs + (u64)p * stride * 3 : d_coeffs, n, d_points, (int)n_points, d_out + (size_t)p * n_points * 3));
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_out, d_out, (size_t)n_polys * n_points * 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

int32_t tvm_quotient_segments(tvm_ctx* c, const uint64_t* d_cw, tvm_domain qd, tvm_domain ldt, const uint64_t* h_rnd,
                              uint64_t n_rand, uint64_t zeta, tvm_table** out_table, uint64_t* d_polys, uint64_t poly_len) {
    if (!c || !out_table) return TVM_ERR_INVALID_ARGUMENT;
    *out_table = nullptr;
    if (!d_cw || !d_polys || (n_rand && !h_rnd) || !valid_domain(qd) || !valid_domain(ldt) || qd.length < 4 ||
        poly_len < qd.length / 4 || poly_len < n_rand || ldt.length < 2)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "quotient_segments arguments");
    const u64 Q = qd.length, L = ldt.length;
    u64* coeffs = (u64*)scratch(c, 11, Q * 3 * sizeof(u64));
    const u64* d_rnd = stage_small(c, 9, h_rnd, 3 * (size_t)n_rand);
    if (!coeffs || !d_rnd) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "segments scratch");
    TVM_TRY(tvm_interpolate(c, 3, d_cw, qd, coeffs));
    TVM_TRY(randomized_segments(c, coeffs, Q, d_rnd, n_rand, zeta, poly_len, d_polys));
    u64 M = 2;
    while (M < poly_len) M <<= 1;
    const u64* polys_in = d_polys;       // what is evaluated: the segment polynomials, or their reductions below
    u64 in_len = poly_len;
    if (M > L) {
        // More coefficients than points: `ldt` is one rank's share of the LDT domain in a many-GPU split (8 ranks: one coset
        // of the trace domain, half as long as a segment polynomial).  Reduce modulo X^L - offset^L first, which leaves the
        // values on the coset unchanged (arithmetic_domain.rs:153-167); d_polys keeps the polynomials themselves.
        u64* folded = (u64*)scratch(c, 25, (size_t)15 * L * sizeof(u64));
        if (!folded) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "segment folding scratch");
        for (int k = 0; k < 5; k++)
            TVM_LAUNCH(k_fold_chunks, dim3((unsigned)((3 * L + 255) / 256)), dim3(256), 0, c->stream, d_polys + (u64)k * poly_len * 3,
                       poly_len, 3, L, bfe_pow(ldt.offset, L), folded + (u64)k * L * 3);
        polys_in = folded;
        in_len = M = L;
    }
    const u64 X = L / M;
    const bool via_lde = M >= 16 && X >= 2;   // the table kernels' shapes (ntt.hip: lde_table)
    tvm_table* t = new (std::nothrow) tvm_table();
    if (!t) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "table handle");
    t->rows = L;
    t->layout = via_lde ? lde_table_layout(M, L) : tab_layout_natural(L);
    t->n_cols = 5;
    t->fk = 3;
    t->W = 15;
    t->data = (u64*)pool_alloc(c, t->bytes());
    if (!t->data) {
        delete t;
        return set_error(c, TVM_ERR_OUT_OF_MEMORY, "segment table allocation");
    }
    int rc = TVM_OK;
    if (via_lde) {
        // The 5 polynomials have fewer than M coefficients: their values on the M-th roots of unity (one transform of 15
        // base-field columns) are a "trace" whose low-degree extension onto the LDT domain is the segment table -- the table
        // kernels write it coset-major and row-block-major directly (context.h), at a quarter of the cost per column of X
        // generic coset transforms plus a transposition (5.0 -> 2.4 ms at 2^20 rows).  (The successor blocks stay unfilled:
        // nothing reads the "next" row of a segment.)
        u64* values = (u64*)scratch(c, 12, (size_t)15 * M * sizeof(u64));
        if (!values) rc = set_error(c, TVM_ERR_OUT_OF_MEMORY, "segment values scratch");
        const u64 w_m = bfe_pow(ldt.generator, X);
        if (rc == TVM_OK) rc = ntt_columns(c, polys_in, in_len, 3, 3 * in_len, values, 3, 3 * M, 1, 0, 15, M, w_m, TVM_ONE, TVM_ONE, TVM_ONE);
        if (rc == TVM_OK && t->layout.storage_rows() % TVM_RB) {
            const u64 full = t->layout.storage_rows() / TVM_RB * TVM_RB * (u64)t->W;
            (void)hipMemsetAsync(t->data + full, 0, t->bytes() - full * sizeof(u64), c->stream);
        }
        if (rc == TVM_OK) rc = lde_table(c, 3, values, M, 5, nullptr, 0, w_m, ldt.offset, ldt.generator, L, t->data, 0);
    } else {
        // evaluate the 5 polynomials (15 base-field columns) into planar codewords, then lay them out as a table
        u64* planar = (u64*)scratch(c, 12, (size_t)15 * L * sizeof(u64));
        if (!planar) rc = set_error(c, TVM_ERR_OUT_OF_MEMORY, "segment codewords scratch");
        for (u64 k = 0; k < X && rc == TVM_OK; k++) {
            const u64 off = bfe_mul(ldt.offset, bfe_pow(ldt.generator, k));
            rc = ntt_columns(c, polys_in, in_len, 3, 3 * in_len, planar, 1, L, X, k, 15, M, bfe_pow(ldt.generator, X), off,
                             TVM_ONE, TVM_ONE);
        }
        if (rc == TVM_OK) rc = columns_to_table(c, planar, L, L, 15, t->data);
    }
    if (rc != TVM_OK) {
        pool_release(c, t->data);
        delete t;
        return rc;
    }
    *out_table = t;
    return TVM_OK;
}

int32_t tvm_table_linear_combination(tvm_ctx* c, const tvm_table* t, uint64_t ldt_length, const uint64_t* h_w, uint64_t* d_out) {
    if (!c || !t || !h_w || !d_out || !is_pow2(ldt_length) || ldt_length > t->rows)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "table_linear_combination arguments");
    const u64* d_w = stage_small(c, 9, h_w, 3 * (size_t)t->n_cols);
    if (!d_w) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "weights staging");
    return table_lincomb(c, t->data, t->layout, t->fk, t->n_cols, t->rows / ldt_length, d_w, d_out);
}

int32_t tvm_deep_codeword(tvm_ctx* c, uint32_t n_comp, const uint64_t* const* d_cw, tvm_domain dom, const uint64_t* h_points,
                          const uint64_t* h_values, const uint64_t* h_weights, uint64_t* d_out) {
    if (!c || !d_cw || !h_points || !h_values || !h_weights || !d_out || !valid_domain(dom))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "deep_codeword arguments");
    return deep_sum(c, (int)n_comp, d_cw, h_points, h_values, h_weights, dom.offset, dom.generator, dom.length, d_out);
}

int32_t tvm_fri_commit_phase(tvm_ctx* c, const uint64_t* d_cw, tvm_domain dom, uint32_t n_rounds, const uint64_t* h_state,
                             uint64_t* const* d_codewords, uint64_t* const* d_nodes, uint64_t* h_roots, uint64_t* h_challenges) {
    if (!c || !d_cw || !valid_domain(dom) || !h_state || !d_nodes || !h_roots || (n_rounds && (!d_codewords || !h_challenges)) ||
        n_rounds >= 64 || (dom.length >> n_rounds) < 1)   // every fold halves a codeword of at least two elements: the last one has at least one
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fri_commit_phase arguments");
    for (uint32_t r = 0; r <= n_rounds; r++)
        if (!d_nodes[r] || (r < n_rounds && !d_codewords[r])) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fri_commit_phase: null buffer");
    // device words: the sponge (16), the roots ((n_rounds + 1) * 5), the challenges (n_rounds * 3)
    const size_t n_words = 16 + (size_t)(n_rounds + 1) * 5 + (size_t)n_rounds * 3;
    u64* d = (u64*)scratch(c, 24, n_words * sizeof(u64));
    if (!d) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "fri_commit_phase scratch");
    u64 *d_state = d, *d_roots = d + 16, *d_ch = d_roots + (size_t)(n_rounds + 1) * 5;
    TVM_TRY(tvm::h2d_small(c, d_state, h_state, 16 * sizeof(u64)));   // (h_state may be a caller temporary)
    const u64* cw = d_cw;
    u64 offset = dom.offset, gen = dom.generator, n = dom.length;
    for (uint32_t r = 0; r <= n_rounds; r++) {
        TVM_LAUNCH(tvm::k_xfe_aos_leaves, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, cw, n, d_nodes[r] + 5 * n);
        TVM_TRY(merkle_tree_from_leaves(c, d_nodes[r], n));
        TVM_HIP_CHECK(c, hipMemcpyAsync(d_roots + 5 * r, d_nodes[r] + 5, 5 * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        TVM_TRY(sponge_absorb_root_and_sample(c, d_state, d_nodes[r] + 5, r < n_rounds ? d_ch + 3 * r : nullptr));
        if (r == n_rounds) break;
        TVM_TRY(fri_fold(c, cw, n, offset, gen, nullptr, d_codewords[r], d_ch + 3 * r));
        cw = d_codewords[r];
        offset = bfe_mul(offset, offset);   // ArithmeticDomain::pow(2) (arithmetic_domain.rs:150-160)
        gen = bfe_mul(gen, gen);
        n >>= 1;
    }
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_roots, d_roots, (size_t)(n_rounds + 1) * 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (n_rounds) TVM_HIP_CHECK(c, hipMemcpyAsync(h_challenges, d_ch, (size_t)n_rounds * 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

int32_t tvm_fri_split_and_fold(tvm_ctx* c, const uint64_t* d_cw, tvm_domain dom, const uint64_t* h_ch, uint64_t* d_out) {
    if (!c || !d_cw || !h_ch || !d_out || !valid_domain(dom))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fri_split_and_fold arguments");
    return fri_fold(c, d_cw, dom.length, dom.offset, dom.generator, h_ch, d_out);
}

}  // extern "C"

extern "C" // ---------------------------------------------------------------------------------- degree-lowering fill
int32_t tvm_fill_derived_main_columns(tvm_ctx* c, uint64_t* d_main_trace, uint64_t n_rows) {
    if (!c || !d_main_trace || n_rows < 2) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_fill_derived_main_columns arguments");
    return fill_degree_lowering(c, 0, d_main_trace, nullptr, nullptr, n_rows);
}
int32_t tvm_fill_derived_aux_columns(tvm_ctx* c, const uint64_t* d_main_trace, uint64_t* d_aux_trace, uint64_t n_rows,
                                     const uint64_t* h_challenges) {
    if (!c || !d_main_trace || !d_aux_trace || !h_challenges || n_rows < 2)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_fill_derived_aux_columns arguments");
    u64* staged = (u64*)scratch(c, 20, (size_t)3 * TVM_NUM_CHALLENGES * sizeof(u64));
    if (!staged) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "challenge staging");
    TVM_TRY(tvm::h2d_small(c, staged, h_challenges, 3 * TVM_NUM_CHALLENGES * sizeof(u64)));   // (h_challenges may be a caller temporary)
    return fill_degree_lowering(c, 1, const_cast<u64*>(d_main_trace), d_aux_trace, staged, n_rows);
}

int32_t tvm_fill_main_table(tvm_ctx* c, const tvm_aet* aet, uint64_t* d_main_trace, uint64_t n_rows, uint64_t* h_table_lengths_out) {
    if (!c || !aet || !d_main_trace || !h_table_lengths_out || !is_pow2(n_rows) || n_rows < 2 || !aet->processor_trace ||
        !aet->lookup_multiplicities || (aet->program_len && (!aet->program_words || !aet->instruction_multiplicities)) ||
        (aet->op_stack_len && !aet->op_stack_trace) || (aet->ram_len && (!aet->ram_trace || !aet->bezout_coefficients_0 != !aet->bezout_coefficients_1)) ||  // both or neither (device)
        (aet->program_hash_len && !aet->program_hash_trace) || (aet->sponge_len && !aet->sponge_trace) || (aet->hash_len && !aet->hash_trace) ||
        (aet->u32_len && !aet->u32_entries) || (aet->cascade_len && !aet->cascade_entries))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_fill_main_table arguments");
    return fill_main_table(c, aet, d_main_trace, n_rows, h_table_lengths_out);
}

int32_t tvm_pad_main_table(tvm_ctx* c, uint64_t* d_main_trace, uint64_t n_rows, const uint64_t* h_table_lengths) {
    if (!c || !d_main_trace || !h_table_lengths || !is_pow2(n_rows) || n_rows < 2)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_pad_main_table arguments");
    return pad_main_table(c, d_main_trace, n_rows, h_table_lengths);
}

int32_t tvm_extend_aux_table(tvm_ctx* c, const uint64_t* d_main_trace, uint64_t* d_aux_trace, uint64_t n_rows,
                             const uint64_t* h_challenges) {
    if (!c || !d_main_trace || !d_aux_trace || !h_challenges || n_rows < 2)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_extend_aux_table arguments");
    u64* staged = (u64*)scratch(c, 21, (size_t)3 * TVM_NUM_CHALLENGES * sizeof(u64));
    if (!staged) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "challenge staging");
    TVM_TRY(tvm::h2d_small(c, staged, h_challenges, 3 * TVM_NUM_CHALLENGES * sizeof(u64)));   // (h_challenges may be a caller temporary)
    return extend_aux_table(c, d_main_trace, d_aux_trace, staged, n_rows);
}

int32_t tvm_all_quotients_combined(tvm_ctx* c, const tvm_table* mt, const tvm_table* at, tvm_domain td,
                                              tvm_domain qd, const uint64_t* h_challenges, const uint64_t* h_weights,
                                              uint64_t* d_out) {
    if (!c || !mt || !at || !h_challenges || !h_weights || !d_out || !valid_domain(td) || !valid_domain(qd))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "all_quotients_combined arguments");
    if (mt->fk != 1 || mt->n_cols != TVM_NUM_MAIN_COLUMNS || at->fk != 3 || at->n_cols != TVM_NUM_AUX_COLUMNS ||
        mt->rows != at->rows || qd.length > mt->rows || !mt->has_successor_blocks || !at->has_successor_blocks ||
        mt->layout.X != at->layout.X || mt->layout.n1 != at->layout.n1 || mt->layout.n2 != at->layout.n2 || mt->layout.pitch != at->layout.pitch)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "all_quotients_combined: tables must be 379 BFE / 91 XFE columns wide, made by tvm_lde_table over one domain");
    u64* staged = (u64*)scratch(c, 13, (size_t)3 * (TVM_NUM_CHALLENGES + TVM_NUM_QUOTIENT_WEIGHTS) * sizeof(u64));
    if (!staged) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "challenge staging");
    TVM_TRY(tvm::h2d_small(c, staged, h_challenges, 3 * TVM_NUM_CHALLENGES * sizeof(u64)));
    TVM_TRY(tvm::h2d_small(c, staged + 3 * TVM_NUM_CHALLENGES, h_weights, 3 * TVM_NUM_QUOTIENT_WEIGHTS * sizeof(u64)));
    const u64* d_ch = staged;
    const u64* d_w = staged + 3 * TVM_NUM_CHALLENGES;
    // Valid-trace mode (tvm_ctx_set_option TVM_OPTION_AIR_VALID_TRACE; off by default).  Every column is a polynomial with at most m = interpolant_len coefficients, every constraint has degree
    // <= 4 in the columns (the AIR is degree-lowered to 4), so a constraint polynomial has degree <= 4(m - 1).  The
    // consistency and transition quotients divide by X^N - 1 (the transition one times X - w^-1): fewer than
    // 4(m - 1) + 2 - N coefficients -- about 3N + 4h, when the quotient domain has 8N points.  If that fits HALF the
    // quotient domain, those constraints (87 % of the work) are evaluated on its even points only, interpolated there and
    // evaluated on all points.  On a VALID trace -- the constraints vanish on the trace domain, so the quotients ARE
    // polynomials -- these are exactly the field elements the row-by-row evaluation yields, at half the cost; on an
    // invalid trace the row-by-row values are those of a rational function and differ (either way the unmodified
    // verifier rejects: it recomputes the quotient at the out-of-domain point).  An initial / terminal quotient (zerofier of
    // degree 1) has up to d(m - 1) coefficients for a constraint of degree d: the 100 of degree <= 3 fit the half domain too
    // (3(m - 1) <= half) and are generated as a low-degree part; the FOUR of degree 4 are evaluated on every point.
    // The constraints of lower degree have shorter quotients still (air_gen.h: TVM_AIR_PART_CLASS): those of class 2 -- a quarter
    // of the multiplications -- have fewer than N + 2h coefficients and are evaluated on a QUARTER of the points, interpolated
    // there, and their coefficients added to the half-domain interpolant before the one evaluation on all points.
    const u64 m = mt->interpolant_len > at->interpolant_len ? mt->interpolant_len : at->interpolant_len;
    const u64 half = qd.length / 2, quarter = qd.length / 4;
    // (a quotient domain short enough for the parts to run side by side is evaluated row by row: air.hip, fork lanes)
    const bool split = c->air_valid_trace && !tvm::air_parts_fork(c, qd.length) && mt->interpolant_len && at->interpolant_len && half >= 2 * td.length && half % td.length == 0 &&
                       4 * (m - 1) + 2 <= half + td.length && 3 * (m - 1) <= half && mt->rows % half == 0;
    if (!split)
        return all_quotients_combined(c, mt->data, mt->layout, (u64)mt->W, at->data, (u64)at->W, td.length, td.generator,
                                      qd.offset, qd.generator, qd.length, d_ch, d_w, d_out);
    // (class 2 on the quarter domain: transition quotients of degree-2 constraints have 2(m - 1) + 2 - N coefficients at most;
    // class 3 on THREE cosets of the trace domain: those of degree-3 constraints have 3(m - 1) + 2 - N <= 3N)
    const u64 N = td.length;
    const bool split4 = quarter >= 2 * N && quarter % N == 0 && 2 * (m - 1) + 2 <= quarter + N && m <= quarter;
    const bool split3 = split4 && half == 4 * N && 3 * (m - 1) + 2 <= 4 * N && 2 * (m - 1) <= 3 * N && mt->layout.X == at->layout.X &&
                        mt->layout.pitch == at->layout.pitch && mt->layout.X == 2 * (half / N) && mt->layout.pitch % TVM_RB == 0;
    PoolBlock low_block(c, (size_t)(2 * 3 * half + (split4 ? 2 * 3 * quarter : 0) + (split3 ? 6 * 3 * N : 0)) * sizeof(u64));  // released on every exit path
    u64* low = (u64*)low_block.p;
    if (!low) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "quotient scratch");
    u64* coeffs = low + 3 * half;
    const tvm_domain half_dom = {qd.offset, bfe_mul(qd.generator, qd.generator), half};
    const int CLASS_FULL = 1 << 0, CLASS_HALF = 1 << 1, CLASS_QUARTER = 1 << 2, CLASS_THREE = 1 << 3;
    int rc = all_quotients_combined(c, mt->data, mt->layout, (u64)mt->W, at->data, (u64)at->W, td.length, td.generator,
                                    half_dom.offset, half_dom.generator, half, d_ch, d_w, low,
                                    CLASS_HALF | (split4 ? 0 : CLASS_QUARTER) | (split3 ? 0 : CLASS_THREE), 0);
    if (rc == TVM_OK) rc = tvm_interpolate(c, 3, low, half_dom, coeffs);
    u64* extra = coeffs + 3 * half;
    if (rc == TVM_OK && split4) {
        u64* low4 = extra;
        u64* coeffs4 = low4 + 3 * quarter;
        extra = coeffs4 + 3 * quarter;
        const u64 g2 = bfe_mul(qd.generator, qd.generator);
        const tvm_domain quarter_dom = {qd.offset, bfe_mul(g2, g2), quarter};
        rc = all_quotients_combined(c, mt->data, mt->layout, (u64)mt->W, at->data, (u64)at->W, td.length, td.generator,
                                    quarter_dom.offset, quarter_dom.generator, quarter, d_ch, d_w, low4, CLASS_QUARTER, 0);
        if (rc == TVM_OK) rc = tvm_interpolate(c, 3, low4, quarter_dom, coeffs4);
        if (rc == TVM_OK) rc = tvm_xfe_add_assign(c, coeffs, coeffs4, quarter);
    }
    if (rc == TVM_OK && split3) {
        // P = A + X^N B + X^2N C with A, B, C of degree < N.  On the coset gamma_k <w_N> (gamma_k = offset * generator^k) X^N is the
        // constant c_k = gamma_k^N, so P restricted to it is the polynomial Q_k = A + c_k B + c_k^2 C of degree < N: three cosets
        // (k = 0, 2, 4: among the rows the tables hold for the half domain), three N-point interpolations, and the inverse of the
        // 3 x 3 Vandermonde matrix of (c_0, c_2, c_4) coefficient by coefficient.
        const u64 gN = bfe_mul(bfe_mul(bfe_mul(qd.generator, qd.generator), bfe_mul(qd.generator, qd.generator)),
                               bfe_mul(bfe_mul(qd.generator, qd.generator), bfe_mul(qd.generator, qd.generator)));  // generator^8: order N
        const u64 rows_per_table_coset = mt->layout.X / (qd.length / N);   // table cosets per quotient-domain coset (1 when the LDT domain is the quotient domain)
        u64 cs[3];
        u64* vals = extra;          // [3][N] XFE values, then [3][N] XFE coefficients
        u64* qk = vals + 9 * N;
        for (int j = 0; j < 3 && rc == TVM_OK; j++) {
            const u64 k = 2 * (u64)j;
            const u64 gamma = bfe_mul(qd.offset, bfe_pow(qd.generator, k));
            cs[j] = bfe_pow(gamma, N);
            const tvm_domain dom = {gamma, gN, N};
            TabLayout lm = mt->layout;   // the one coset of the tables this domain is: a table of its own
            lm.X = 1;
            lm.log_x = 0;
            const u64 first_row = k * rows_per_table_coset * mt->layout.pitch;
            const u64* sub_main = mt->data + tvm_tab_idx(first_row, 0, (u64)mt->W);
            const u64* sub_aux = at->data + tvm_tab_idx(first_row, 0, (u64)at->W);
            rc = all_quotients_combined(c, sub_main, lm, (u64)mt->W, sub_aux, (u64)at->W, td.length, td.generator, dom.offset,
                                        dom.generator, N, d_ch, d_w, vals + 3 * N * j, CLASS_THREE, 0);
            if (rc == TVM_OK) rc = tvm_interpolate(c, 3, vals + 3 * N * j, dom, qk + 3 * N * j);
        }
        if (rc == TVM_OK) {
            // inverse Vandermonde: row r of V^-1 holds the coefficients of the Lagrange basis polynomial data, i.e.
            // [A B C]^T = V^-1 [Q_0 Q_2 Q_4]^T with V = [[1, c, c^2]]
            tvm::ThreeCosetWeights w;
            const u64 c0 = cs[0], c1 = cs[1], c2 = cs[2];
            const u64 d0 = bfe_inv(bfe_mul(bfe_sub(c0, c1), bfe_sub(c0, c2)));
            const u64 d1 = bfe_inv(bfe_mul(bfe_sub(c1, c0), bfe_sub(c1, c2)));
            const u64 d2 = bfe_inv(bfe_mul(bfe_sub(c2, c0), bfe_sub(c2, c1)));
            // Lagrange polynomial L_k(t) = prod_{j != k} (t - c_j) / prod (c_k - c_j) = (t^2 - (sum of the other two) t + product) * d_k
            const u64 dk[3] = {d0, d1, d2};
            const u64 oth[3][2] = {{c1, c2}, {c0, c2}, {c0, c1}};
            for (int k3 = 0; k3 < 3; k3++) {
                w.w[0 * 3 + k3] = bfe_mul(bfe_mul(oth[k3][0], oth[k3][1]), dk[k3]);             // t^0 -> A
                w.w[1 * 3 + k3] = bfe_mul(bfe_neg(bfe_add(oth[k3][0], oth[k3][1])), dk[k3]);    // t^1 -> B
                w.w[2 * 3 + k3] = dk[k3];                                                       // t^2 -> C
            }
            TVM_LAUNCH(tvm::k_three_coset_combine, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, qk, qk + 3 * N, qk + 6 * N, N, w, coeffs);
            TVM_HIP_CHECK(c, hipGetLastError());
        }
    }
    if (rc == TVM_OK) rc = tvm_evaluate(c, 3, coeffs, half, qd, d_out);
    if (rc == TVM_OK)
        rc = all_quotients_combined(c, mt->data, mt->layout, (u64)mt->W, at->data, (u64)at->W, td.length, td.generator,
                                    qd.offset, qd.generator, qd.length, d_ch, d_w, d_out, CLASS_FULL, 1);
    return rc;
}

// ---- the valid-trace AIR piecewise: what tvm_all_quotients_combined does in one call on one device, as the pieces a multi-GPU
// host distributes over its ranks (triton_vm_amd/host/sharded_host.cpp).  A class's quotient polynomial has fewer than n * N
// coefficients (n = tvm_air_class_cosets), so its values on ANY n cosets of the trace domain determine it.
extern "C" {
int32_t tvm_air_class_cosets(const tvm_table* mt, const tvm_table* at, tvm_domain td, uint32_t out[4]) {
    if (!mt || !at || !out || !valid_domain(td)) return TVM_ERR_INVALID_ARGUMENT;
    const u64 m = mt->interpolant_len > at->interpolant_len ? mt->interpolant_len : at->interpolant_len;
    const u64 N = td.length;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!mt->interpolant_len || !at->interpolant_len) return TVM_OK;
    // the bounds of tvm_all_quotients_combined (above): consistency / transition quotients of degree-d constraints have at most
    // d (m - 1) + 2 - N coefficients, initial / terminal ones d' (m - 1) + 1 with d' one class lower
    if (4 * (m - 1) + 2 <= 5 * N && 3 * (m - 1) <= 4 * N) out[1] = 4;   // class "half"
    if (2 * (m - 1) + 2 <= 3 * N && m <= 2 * N) out[2] = 2;             // class "quarter"
    if (3 * (m - 1) + 2 <= 4 * N && 2 * (m - 1) <= 3 * N) out[3] = 3;   // class "three cosets"
    return TVM_OK;
}

int32_t tvm_air_class_values(tvm_ctx* c, const tvm_table* mt, const tvm_table* at, tvm_domain td, tvm_domain table_dom, uint32_t coset,
                             uint32_t class_mask, const uint64_t* h_challenges, const uint64_t* h_weights, uint64_t* d_out) {
    if (!c || !mt || !at || !h_challenges || !h_weights || !d_out || !valid_domain(td) || !valid_domain(table_dom) || !class_mask || class_mask > 15)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "air_class_values arguments");
    const u64 N = td.length;
    if (mt->fk != 1 || mt->n_cols != TVM_NUM_MAIN_COLUMNS || at->fk != 3 || at->n_cols != TVM_NUM_AUX_COLUMNS || mt->rows != at->rows ||
        mt->rows != table_dom.length || table_dom.length % N || coset >= table_dom.length / N || !mt->has_successor_blocks ||
        !at->has_successor_blocks || mt->layout.X != table_dom.length / N || at->layout.X != mt->layout.X || mt->layout.pitch != at->layout.pitch ||
        mt->layout.pitch % TVM_RB)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "air_class_values: tables made by tvm_lde_table over table_domain, one coset of them");
    u64* staged = (u64*)scratch(c, 13, (size_t)3 * (TVM_NUM_CHALLENGES + TVM_NUM_QUOTIENT_WEIGHTS) * sizeof(u64));
    if (!staged) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "challenge staging");
    TVM_TRY(tvm::h2d_small(c, staged, h_challenges, 3 * TVM_NUM_CHALLENGES * sizeof(u64)));
    TVM_TRY(tvm::h2d_small(c, staged + 3 * TVM_NUM_CHALLENGES, h_weights, 3 * TVM_NUM_QUOTIENT_WEIGHTS * sizeof(u64)));
    const u64 X = table_dom.length / N;
    const u64 gamma = bfe_mul(table_dom.offset, bfe_pow(table_dom.generator, coset));
    TabLayout lm = mt->layout;   // the one coset of the tables: a table of its own (the layout is coset-major, context.h)
    lm.X = 1;
    lm.log_x = 0;
    const u64 first_row = (u64)coset * mt->layout.pitch;
    return all_quotients_combined(c, mt->data + tvm_tab_idx(first_row, 0, (u64)mt->W), lm, (u64)mt->W, at->data + tvm_tab_idx(first_row, 0, (u64)at->W),
                                  (u64)at->W, N, td.generator, gamma, bfe_pow(table_dom.generator, X), N, staged, staged + 3 * TVM_NUM_CHALLENGES,
                                  d_out, (int)class_mask, 0);
}

int32_t tvm_coset_values_to_coefficients(tvm_ctx* c, tvm_domain td, uint32_t n_cosets, const uint64_t* h_offsets,
                                         const uint64_t* const* d_values, uint64_t* d_coeffs) {
    if (!c || !valid_domain(td) || !n_cosets || n_cosets > 4 || !h_offsets || !d_values || !d_coeffs)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "coset_values_to_coefficients arguments");
    const u64 N = td.length;
    // P = sum_i X^(iN) A_i with A_i of degree < N.  On the coset gamma_j <w_N>, X^N is the constant c_j = gamma_j^N, so the
    // interpolant of P's values there is Q_j = sum_i c_j^i A_i: n interpolations, then the inverse of the n x n Vandermonde
    // matrix of (c_0 .. c_{n-1}) coefficient by coefficient.  Row i of the inverse holds the coefficients of x^i in the
    // Lagrange basis polynomials L_j(x) = prod_{l != j} (x - c_l) / (c_j - c_l).
    u64 cs[4];
    for (uint32_t j = 0; j < n_cosets; j++) cs[j] = bfe_pow(h_offsets[j], N);
    tvm::CosetCombineWeights w;
    for (int i = 0; i < 16; i++) w.w[i] = 0;
    for (uint32_t j = 0; j < n_cosets; j++) {
        u64 poly[4] = {TVM_ONE, 0, 0, 0};   // running product prod (x - c_l), low coefficient first
        u64 denom = TVM_ONE;
        int deg = 0;
        for (uint32_t l = 0; l < n_cosets; l++) {
            if (l == j) continue;
            if (cs[l] == cs[j]) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "coset_values_to_coefficients: two cosets with the same X^N");
            for (int e = deg + 1; e >= 1; e--) poly[e] = bfe_sub(poly[e - 1], bfe_mul(poly[e], cs[l]));
            poly[0] = bfe_neg(bfe_mul(poly[0], cs[l]));
            deg++;
            denom = bfe_mul(denom, bfe_sub(cs[j], cs[l]));
        }
        const u64 dinv = bfe_inv(denom);
        for (uint32_t i = 0; i < n_cosets; i++) w.w[4 * i + j] = bfe_mul(poly[i], dinv);
    }
    PoolBlock block(c, (size_t)n_cosets * 3 * N * sizeof(u64));
    u64* q = (u64*)block.p;
    if (!q) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "coset interpolants");
    tvm::CosetCombinePointers ptrs;
    for (uint32_t j = 0; j < 4; j++) ptrs.q[j] = nullptr;
    for (uint32_t j = 0; j < n_cosets; j++) {
        if (!d_values[j]) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "coset_values_to_coefficients: null values");
        const tvm_domain dom = {h_offsets[j], td.generator, N};
        TVM_TRY(tvm_interpolate(c, 3, d_values[j], dom, q + (u64)j * 3 * N));
        ptrs.q[j] = q + (u64)j * 3 * N;
    }
    TVM_LAUNCH(tvm::k_coset_combine, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, ptrs, (int)n_cosets, N, w, d_coeffs);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
}  // extern "C"

// ---------------------------------------------------------------------------------- STIR
extern "C" {
int32_t tvm_stir_merkle_tree(tvm_ctx* c, const uint64_t* d_cw, uint64_t n, uint32_t stack_height, uint64_t* d_nodes) {
    if (!c || !d_cw || !d_nodes || !is_pow2(n) || !stack_height || !is_pow2(stack_height) || stack_height > n ||
        stack_height > 16)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "stir_merkle_tree arguments");
    const u64 d = n / stack_height;
    TVM_TRY(stir_hash_stacked(c, d_cw, n, (int)stack_height, d_nodes + 5 * d));
    return merkle_tree_from_leaves(c, d_nodes, d);
}
int32_t tvm_fold_polynomial(tvm_ctx* c, const uint64_t* d_poly, uint64_t n_coeffs, uint32_t folding_factor,
                            const uint64_t* h_randomness, uint64_t* d_out) {
    if (!c || (n_coeffs && !d_poly) || !d_out || !h_randomness || !folding_factor || folding_factor > 64)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "fold_polynomial arguments");
    return stir_fold_polynomial(c, d_poly, n_coeffs, (int)folding_factor, h_randomness, d_out);
}
int32_t tvm_stir_next_polynomial(tvm_ctx* c, const uint64_t* d_folded_poly, uint64_t n_coeffs, const uint64_t* h_quotient_set,
                                 const uint64_t* h_answer_poly, uint32_t k, const uint64_t* h_degree_correction_randomness,
                                 tvm_domain work_domain, uint64_t* d_out_poly) {
    if (!c || !d_folded_poly || !d_out_poly || !h_degree_correction_randomness || (k && (!h_quotient_set || !h_answer_poly)) ||
        !valid_domain(work_domain) || n_coeffs > work_domain.length)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "stir_next_polynomial arguments");
    const u64 M = work_domain.length;
    u64* vals = (u64*)scratch(c, 15, (size_t)M * 3 * sizeof(u64));
    u64* staged = (u64*)scratch(c, 16, (size_t)(k ? k : 1) * 6 * sizeof(u64));
    if (!vals || !staged) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "stir scratch");
    if (k) {
        TVM_TRY(tvm::h2d_small(c, staged, h_quotient_set, (size_t)k * 3 * sizeof(u64)));
        TVM_TRY(tvm::h2d_small(c, staged + 3 * (size_t)k, h_answer_poly, (size_t)k * 3 * sizeof(u64)));
    }
    TVM_TRY(tvm_evaluate(c, 3, d_folded_poly, n_coeffs, work_domain, vals));
    // Ans on the work domain by one zero-padded transform where that is cheaper than Horner at every point (k multiplications
    // by a base-field element per point against log2(M) butterfly layers: the first STIR round of a 2^20-row proof spent 0.7 of
    // its 1.2 ms quotient kernel in the Horner loop over 204 coefficients)
    u64* ans_values = nullptr;
    if (k >= 32 && k <= M) {
        ans_values = (u64*)scratch(c, 17, (size_t)M * 3 * sizeof(u64));
        if (!ans_values) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "stir scratch");
        TVM_TRY(tvm_evaluate(c, 3, staged + 3 * (size_t)k, k, work_domain, ans_values));
    }
    u32 kb = 0;  // leading base-field points of the quotient set
    while (kb < k && h_quotient_set[3 * kb + 1] == 0 && h_quotient_set[3 * kb + 2] == 0) kb++;
    TVM_TRY(stir_quotient(c, vals, M, work_domain.offset, work_domain.generator, staged, staged + 3 * (size_t)k, ans_values, k, kb,
                          h_degree_correction_randomness));
    return tvm_interpolate(c, 3, vals, work_domain, d_out_poly);
}

// the same interpolation on the device (one workgroup, k <= 256; stir.hip: k_xfe_interpolate); larger k: the host function below
int32_t tvm_xfe_interpolate(tvm_ctx* c, const uint64_t* h_points, const uint64_t* h_values, uint32_t k, uint64_t* h_out) {
    if (!c || (k && (!h_points || !h_values || !h_out))) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "xfe_interpolate arguments");
    if (!k) return TVM_OK;
    if (k > 256) return tvm_host_xfe_interpolate(h_points, h_values, k, h_out);
    u64* d = (u64*)scratch(c, 27, (size_t)(9 * k + 2) * sizeof(u64));
    if (!d) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "interpolation staging");
    int* d_status = (int*)(d + 9 * k);
    TVM_HIP_CHECK(c, hipMemcpyAsync(d, h_points, 3 * (size_t)k * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    TVM_HIP_CHECK(c, hipMemcpyAsync(d + 3 * k, h_values, 3 * (size_t)k * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    TVM_HIP_CHECK(c, hipMemsetAsync(d_status, 0, sizeof(int), c->stream));
    TVM_TRY(xfe_interpolate(c, d, d + 3 * k, (int)k, d + 6 * k, d_status));
    int status = 0;
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_out, d + 6 * k, 3 * (size_t)k * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return status ? set_error(c, TVM_ERR_INVALID_ARGUMENT, "xfe_interpolate: repeated point") : TVM_OK;
}

// Polynomial::interpolate for a handful of XFE points (the STIR "Ans" polynomial, stir.rs:954): Newton's divided
// differences, then expansion to monomial coefficients.  Host work, O(k^2), points pairwise distinct.
int32_t tvm_host_xfe_interpolate(const uint64_t* points, const uint64_t* values, uint32_t k, uint64_t* out_coeffs) {
    if (k && (!points || !values || !out_coeffs)) return TVM_ERR_INVALID_ARGUMENT;
    std::vector<xfe> p(k), d(k);
    for (u32 i = 0; i < k; i++) {
        p[i] = xfe_make(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        d[i] = xfe_make(values[3 * i], values[3 * i + 1], values[3 * i + 2]);
    }
    // divided differences; the k - j denominators of level j are inverted together (one inversion per level instead of
    // one per entry: at k = 204 the entry-wise form spent 12 ms of host time per STIR round in xfe_inv)
    std::vector<xfe> den(k), pre(k);
    for (u32 j = 1; j < k; j++) {
        xfe acc = xfe_one();
        for (u32 i = j; i < k; i++) {
            den[i] = xfe_sub(p[i], p[i - j]);
            if (xfe_eq(den[i], xfe_zero())) return TVM_ERR_INVALID_ARGUMENT;  // repeated point
            pre[i] = acc;                                                     // den[j] * ... * den[i-1]
            acc = xfe_mul(acc, den[i]);
        }
        xfe inv = xfe_inv(acc);                                               // 1 / (den[j] * ... * den[k-1])
        for (u32 i = k - 1; i >= j; i--) {
            const xfe den_inv = xfe_mul(inv, pre[i]);
            inv = xfe_mul(inv, den[i]);
            d[i] = xfe_mul(xfe_sub(d[i], d[i - 1]), den_inv);                 // d[i-1] still holds the previous level
        }
    }
    // Horner over the Newton basis: c(X) = d[k-1];  c <- c * (X - p[i]) + d[i]  for i = k-2 .. 0
    std::vector<xfe> co(k ? k : 1, xfe_zero()), nxt(k ? k : 1, xfe_zero());
    if (k) co[0] = d[k - 1];
    for (u32 deg = 0, i = k > 1 ? k - 1 : 0; i-- > 0; deg++) {
        for (u32 e = 0; e <= deg + 1; e++) {
            const xfe shifted = e ? co[e - 1] : xfe_zero();                       // X * c
            const xfe scaled = e <= deg ? xfe_mul(p[i], co[e]) : xfe_zero();      // p_i * c
            nxt[e] = xfe_sub(shifted, scaled);
        }
        nxt[0] = xfe_add(nxt[0], d[i]);
        co.swap(nxt);
    }
    for (u32 i = 0; i < k; i++) {
        out_coeffs[3 * i] = co[i].c0;
        out_coeffs[3 * i + 1] = co[i].c1;
        out_coeffs[3 * i + 2] = co[i].c2;
    }
    return TVM_OK;
}
}  // extern "C"

// ---------------------------------------------------------------------------------- small transfers, host helpers
#include "tip5.h"

namespace tvm {
__global__ void k_gather_elements(const u64* __restrict__ src, u32 elem_words, const u64* __restrict__ idx, u64 n,
                                  u64* __restrict__ out) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * elem_words) return;
    out[e] = src[idx[e / elem_words] * elem_words + e % elem_words];
}
}  // namespace tvm

namespace tvm {
__global__ void k_scatter_strided(const u64* __restrict__ src, u32 elem_words, u64 n, u64 stride, u64 offset,
                                  u64* __restrict__ dst) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * elem_words) return;
    dst[((e / elem_words) * stride + offset) * elem_words + e % elem_words] = src[e];
}
}  // namespace tvm

extern "C" {
int32_t tvm_scatter_strided(tvm_ctx* c, const uint64_t* d_src, uint32_t elem_words, uint64_t n, uint64_t stride,
                            uint64_t offset, uint64_t* d_dst) {
    if (!c || !elem_words || !stride || offset >= stride || (n && (!d_src || !d_dst)))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "scatter_strided arguments");
    if (!n) return TVM_OK;
    const u64 total = n * elem_words;
    TVM_LAUNCH(tvm::k_scatter_strided, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_src, elem_words, n,
               stride, offset, d_dst);
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}
int32_t tvm_gather_elements(tvm_ctx* c, const uint64_t* d_src, uint32_t elem_words, const uint64_t* h_idx, uint64_t n,
                            uint64_t* h_out) {
    if (!c || !d_src || !elem_words || (n && (!h_idx || !h_out))) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "gather arguments");
    if (!n) return TVM_OK;
    u64* d_idx = (u64*)scratch(c, 4, n * sizeof(u64));
    u64* d_out = (u64*)scratch(c, 5, n * elem_words * sizeof(u64));
    if (!d_idx || !d_out) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "gather scratch");
    TVM_TRY(tvm::h2d_small(c, d_idx, h_idx, n * sizeof(u64)));
    const u64 total = n * elem_words;
    TVM_LAUNCH(tvm::k_gather_elements, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_src, elem_words,
               d_idx, n, d_out);
    TVM_HIP_CHECK(c, hipMemcpyAsync(h_out, d_out, total * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return TVM_OK;
}

int32_t tvm_gather_elements_batch(tvm_ctx* c, uint32_t n_jobs, const uint64_t* const* d_src, const uint32_t* elem_words,
                                  const uint64_t* const* h_idx, const uint64_t* n, uint64_t* const* h_out) {
    if (!c || (n_jobs && (!d_src || !elem_words || !h_idx || !n || !h_out)))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "gather batch arguments");
    u64 n_idx = 0, n_words = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        if (n[j] && (!d_src[j] || !elem_words[j] || !h_idx[j] || !h_out[j]))
            return set_error(c, TVM_ERR_INVALID_ARGUMENT, "gather batch: null job");
        n_idx += n[j];
        n_words += n[j] * elem_words[j];
    }
    if (!n_idx) return TVM_OK;
    std::vector<u64> idx(n_idx), out(n_words);   // (pageable staging: the copies below are complete when the call returns)
    u64 at = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        std::memcpy(idx.data() + at, h_idx[j], n[j] * sizeof(u64));
        at += n[j];
    }
    u64* d_idx = (u64*)scratch(c, 4, n_idx * sizeof(u64));
    u64* d_out = (u64*)scratch(c, 5, n_words * sizeof(u64));
    if (!d_idx || !d_out) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "gather scratch");
    TVM_TRY(tvm::h2d_small(c, d_idx, idx.data(), n_idx * sizeof(u64)));
    u64 i0 = 0, w0 = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const u64 total = n[j] * elem_words[j];
        if (!total) continue;
        TVM_LAUNCH(tvm::k_gather_elements, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_src[j], elem_words[j],
                   d_idx + i0, n[j], d_out + w0);
        i0 += n[j];
        w0 += total;
    }
    TVM_HIP_CHECK(c, hipMemcpyAsync(out.data(), d_out, n_words * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    w0 = 0;
    for (uint32_t j = 0; j < n_jobs; j++) {
        const u64 total = n[j] * elem_words[j];
        if (total) std::memcpy(h_out[j], out.data() + w0, total * sizeof(u64));
        w0 += total;
    }
    return TVM_OK;
}

// (host code: an AVX2 clone beside the baseline one, picked at load time -- the MDS layer is 512 multiply-adds of 32-bit values)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("avx2", "default")))
#endif
void tvm_host_tip5_permutation(uint64_t st[16]) {
    static const u64 rc[80] = {TVM_TIP5_RC_LIST};
    static const unsigned char lut[256] = {TVM_TIP5_LUT_LIST};
    u64 s[16];
    for (int i = 0; i < 16; i++) s[i] = st[i];
    for (int r = 0; r < TIP5_ROUNDS; r++) {
        for (int i = 0; i < 4; i++) s[i] = tip5_sbox_lookup(s[i], lut);
        for (int i = 4; i < 16; i++) s[i] = tip5_pow7(s[i]);
        tip5_mds(s);
        for (int i = 0; i < 16; i++) s[i] = bfe_add(s[i], rc[16 * r + i]);
    }
    for (int i = 0; i < 16; i++) st[i] = s[i];
}
void tvm_host_sponge_pad_and_absorb(uint64_t st[16], const uint64_t* w, uint64_t n) {
    uint64_t pos = 0;
    for (;;) {
        const uint64_t rem = n - pos;
        if (rem >= 10) {
            for (int i = 0; i < 10; i++) st[i] = w[pos + i];
            tvm_host_tip5_permutation(st);
            pos += 10;
        } else {
            for (uint64_t i = 0; i < rem; i++) st[i] = w[pos + i];
            st[rem] = TVM_ONE;
            for (uint64_t i = rem + 1; i < 10; i++) st[i] = 0;
            tvm_host_tip5_permutation(st);
            return;
        }
    }
}
void tvm_host_xfe_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]) {
    xfe r = xfe_mul(xfe_make(a[0], a[1], a[2]), xfe_make(b[0], b[1], b[2]));
    out[0] = r.c0; out[1] = r.c1; out[2] = r.c2;
}
void tvm_host_xfe_inv(const uint64_t a[3], uint64_t out[3]) {
    xfe r = xfe_inv(xfe_make(a[0], a[1], a[2]));
    out[0] = r.c0; out[1] = r.c1; out[2] = r.c2;
}
void tvm_host_xfe_powers(const uint64_t x[3], uint64_t first, uint64_t n, uint64_t* out) {
    const xfe b = xfe_make(x[0], x[1], x[2]);
    xfe p = xfe_pow(b, first);
    for (uint64_t i = 0; i < n; i++) {
        out[3 * i] = p.c0; out[3 * i + 1] = p.c1; out[3 * i + 2] = p.c2;
        p = xfe_mul(p, b);
    }
}
// out[j] = sum_i coeffs[i] * points[j]^i (Horner) -- and, with `zerofier` set, prod_i (points[j] - coeffs[i]):
// the verifier's scalar polynomial work (STIR's answer polynomial and quotient-set zerofier at the queried points)
void tvm_host_xfe_poly_eval(const uint64_t* coeffs, uint64_t n, const uint64_t* points, uint64_t m, int32_t zerofier, uint64_t* out) {
    for (uint64_t j = 0; j < m; j++) {
        const xfe x = xfe_make(points[3 * j], points[3 * j + 1], points[3 * j + 2]);
        xfe acc = zerofier ? xfe_one() : xfe_zero();
        if (zerofier)
            for (uint64_t i = 0; i < n; i++) acc = xfe_mul(acc, xfe_sub(x, xfe_make(coeffs[3 * i], coeffs[3 * i + 1], coeffs[3 * i + 2])));
        else
            for (uint64_t i = n; i-- > 0;) acc = xfe_add(xfe_mul(acc, x), xfe_make(coeffs[3 * i], coeffs[3 * i + 1], coeffs[3 * i + 2]));
        out[3 * j] = acc.c0; out[3 * j + 1] = acc.c1; out[3 * j + 2] = acc.c2;
    }
}

/* StdRng of rand [not vendored in the reference tree; restated, pinned by the proof-digest snapshots through
 * tests/test_proof_snapshot.py]: ChaCha with 12 rounds, 64-bit block counter, the blocks' 16 words as a u32 stream, next_u64 =
 * two consecutive words (low first); `rng.random::<BFieldElement>()` = random_range(0..=MAX): widening multiply by p, one more
 * draw when the low half leaves room for a carry (UniformInt::sample_single_inclusive). */
namespace {
struct StdRng {
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void refill() {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) s[4 + i] = key[i];
        s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
        uint32_t x[16];
        for (int i = 0; i < 16; i++) x[i] = s[i];
#define TVM_QR(a, b, c, d) \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int r = 0; r < 6; r++) {
            TVM_QR(0, 4, 8, 12) TVM_QR(1, 5, 9, 13) TVM_QR(2, 6, 10, 14) TVM_QR(3, 7, 11, 15)
            TVM_QR(0, 5, 10, 15) TVM_QR(1, 6, 11, 12) TVM_QR(2, 7, 8, 13) TVM_QR(3, 4, 9, 14)
        }
#undef TVM_QR
        for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
        counter++;
        idx = 0;
    }
    uint32_t next_u32() { if (idx >= 16) refill(); return buf[idx++]; }
    uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | ((uint64_t)next_u32() << 32); }
    uint64_t random_bfe() {
        unsigned __int128 prod = (unsigned __int128)next_u64() * TVM_P;
        uint64_t result = (uint64_t)(prod >> 64);
        const uint64_t lo = (uint64_t)prod;
        if (lo > (uint64_t)(0 - TVM_P)) {
            const uint64_t new_hi = (uint64_t)(((unsigned __int128)next_u64() * TVM_P) >> 64);
            if (lo + new_hi < lo) result++;
        }
        return result;
    }
};
}  // namespace
void tvm_host_stdrng_elements(const uint8_t seed[32], uint64_t n, uint64_t* out) {
    StdRng rng;
    for (int i = 0; i < 8; i++)
        rng.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    for (uint64_t i = 0; i < n; i++) out[i] = bfe_from_u64(rng.random_bfe());
}
}  // extern "C"
namespace tvm {
struct StdRngKey {
    uint32_t w[8];
};
// The same stream on the device: work-item t computes ChaCha block t (16 words = 4 draws of two u64 each) and the four
// elements 4t .. 4t+3.  That is the host stream as long as every draw takes its second u64 -- it does unless the low
// half of x * p is below 2^32 (probability 2^-32 per element); such a draw is reported in *short_draws and the caller
// regenerates on the host, where the stream is consumed sequentially.
// (streams: blockIdx.y = the stream; its key is the launch's key plus the stream number as 256-bit little-endian integers)
__global__ void k_stdrng_elements(StdRngKey key, u64 n, u64* __restrict__ out, unsigned* __restrict__ short_draws) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * t >= n) return;
    if (blockIdx.y) {
        u64 carry = blockIdx.y;
        for (int i = 0; i < 8 && carry; i++) {
            carry += key.w[i];
            key.w[i] = (uint32_t)carry;
            carry >>= 32;
        }
        out += (u64)blockIdx.y * n;
    }
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = key.w[i];
    s[12] = (uint32_t)t; s[13] = (uint32_t)(t >> 32); s[14] = 0; s[15] = 0;
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define TVM_ROTL(v, k) (((v) << (k)) | ((v) >> (32 - (k))))
#define TVM_QR(a, b, c, d) \
    x[a] += x[b]; x[d] = TVM_ROTL(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = TVM_ROTL(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = TVM_ROTL(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = TVM_ROTL(x[b] ^ x[c], 7);
    for (int r = 0; r < 6; r++) {
        TVM_QR(0, 4, 8, 12) TVM_QR(1, 5, 9, 13) TVM_QR(2, 6, 10, 14) TVM_QR(3, 7, 11, 15)
        TVM_QR(0, 5, 10, 15) TVM_QR(1, 6, 11, 12) TVM_QR(2, 7, 8, 13) TVM_QR(3, 4, 9, 14)
    }
#undef TVM_QR
#undef TVM_ROTL
    for (int i = 0; i < 16; i++) x[i] += s[i];
    for (int j = 0; j < 4 && 4 * t + j < n; j++) {
        const u64 a = (u64)x[4 * j] | ((u64)x[4 * j + 1] << 32), b = (u64)x[4 * j + 2] | ((u64)x[4 * j + 3] << 32);
        const unsigned __int128 prod = (unsigned __int128)a * TVM_P;
        u64 result = (u64)(prod >> 64);
        const u64 lo = (u64)prod;
        if (lo <= (u64)(0 - TVM_P)) *short_draws = 1u;  // a flag: every writer stores the same value
        const u64 new_hi = (u64)(((unsigned __int128)b * TVM_P) >> 64);
        if (lo + new_hi < lo) result++;
        out[4 * t + j] = bfe_from_u64(result);
    }
}
}  // namespace tvm
extern "C" {
int32_t tvm_stdrng_elements(tvm_ctx* c, const uint8_t seed[32], uint64_t n, uint64_t* d_out) {
    if (!c || !seed || (n && !d_out)) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_stdrng_elements arguments");
    if (!n) return TVM_OK;
    tvm::StdRngKey key;
    for (int i = 0; i < 8; i++)
        key.w[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    unsigned* d_flag = (unsigned*)pool_alloc(c, sizeof(unsigned));
    if (!d_flag) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "stdrng flag");
    int rc = TVM_OK;
    unsigned short_draws = 0;
    if (hipMemsetAsync(d_flag, 0, sizeof(unsigned), c->stream) != hipSuccess) rc = set_error(c, TVM_ERR_DEVICE, "stdrng flag");
    if (rc == TVM_OK) {
        const u64 blocks = (n + 3) / 4;
        TVM_LAUNCH(tvm::k_stdrng_elements, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, c->stream, key, n, d_out, d_flag);
        if (hipMemcpyAsync(&short_draws, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = set_error(c, TVM_ERR_DEVICE, "stdrng elements");
    }
    pool_release(c, d_flag);
    if (rc == TVM_OK && short_draws) {  // a draw that took one u64 shifts the rest of the stream: sequential on the host
        std::vector<uint64_t> host(n);
        tvm_host_stdrng_elements(seed, n, host.data());
        if (hipMemcpyAsync(d_out, host.data(), n * sizeof(u64), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess)
            rc = set_error(c, TVM_ERR_DEVICE, "stdrng elements (host path)");
    }
    return rc;
}

int32_t tvm_stdrng_streams(tvm_ctx* c, const uint8_t seed[32], uint64_t n_streams, uint64_t per_stream, uint64_t* d_out) {
    if (!c || !seed || n_streams > 65535 || (n_streams && per_stream && !d_out))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_stdrng_streams arguments");
    if (!n_streams || !per_stream) return TVM_OK;
    tvm::StdRngKey key;
    for (int i = 0; i < 8; i++)
        key.w[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    tvm::PoolBlock flag(c, sizeof(unsigned));
    unsigned* d_flag = (unsigned*)flag.p;
    if (!d_flag) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "stdrng flag");
    unsigned short_draws = 0;
    TVM_HIP_CHECK(c, hipMemsetAsync(d_flag, 0, sizeof(unsigned), c->stream));
    const u64 blocks = (per_stream + 3) / 4;
    TVM_LAUNCH(tvm::k_stdrng_elements, dim3((unsigned)((blocks + 255) / 256), (unsigned)n_streams), dim3(256), 0, c->stream, key, per_stream, d_out, d_flag);
    TVM_HIP_CHECK(c, hipMemcpyAsync(&short_draws, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (short_draws) {  // a draw that took one u64 shifts the rest of its stream: every stream sequentially on the host
        std::vector<uint64_t> host(n_streams * per_stream);
        for (uint64_t st = 0; st < n_streams; st++) {
            uint8_t sd[32];
            uint64_t carry = st;
            for (int k = 0; k < 32; k++) {
                carry += seed[k];
                sd[k] = (uint8_t)carry;
                carry >>= 8;
            }
            tvm_host_stdrng_elements(sd, per_stream, host.data() + st * per_stream);
        }
        TVM_HIP_CHECK(c, hipMemcpyAsync(d_out, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        TVM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    }
    return TVM_OK;
}
}  // extern "C"

extern "C" {
int32_t tvm_bezout_coefficients(tvm_ctx* c, const uint64_t* d_roots, uint64_t n, uint64_t* d_a, uint64_t* d_b) {
    if (!c || (n && (!d_roots || !d_a || !d_b))) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_bezout_coefficients arguments");
    return tvm::bezout_coefficients(c, d_roots, n, d_a, d_b);
}
}  // extern "C"

extern "C" {  // ---------------------------------------------------------------------------------- verifier batch work
int32_t tvm_verifier_row_digests(tvm_ctx* c, const uint64_t* h_rows, uint64_t n_rows, uint64_t row_words, uint64_t* h_digests) {
    if (!c || !h_rows || !h_digests || !n_rows || !row_words || row_words > (1u << 20))
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_verifier_row_digests arguments");
    u64* d_rows = (u64*)pool_alloc(c, (size_t)n_rows * row_words * sizeof(u64));
    u64* d_digests = (u64*)pool_alloc(c, (size_t)n_rows * 5 * sizeof(u64));
    int rc = TVM_OK;
    if (!d_rows || !d_digests) rc = set_error(c, TVM_ERR_OUT_OF_MEMORY, "row digests scratch");
    if (rc == TVM_OK && hipMemcpyAsync(d_rows, h_rows, (size_t)n_rows * row_words * sizeof(u64), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = set_error(c, TVM_ERR_DEVICE, "row upload");
    if (rc == TVM_OK) rc = hash_varlen_rows(c, d_rows, n_rows, (int)row_words, d_digests);
    if (rc == TVM_OK && (hipMemcpyAsync(h_digests, d_digests, (size_t)n_rows * 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                         hipStreamSynchronize(c->stream) != hipSuccess))
        rc = set_error(c, TVM_ERR_DEVICE, "digest download");
    (void)hipStreamSynchronize(c->stream);  // h_rows may be a caller temporary
    pool_release(c, d_rows);
    pool_release(c, d_digests);
    return rc;
}

int32_t tvm_verifier_deep_values(tvm_ctx* c, const uint64_t* h_main_rows, const uint64_t* h_aux_rows, const uint64_t* h_quot_rows,
                                 const uint64_t* h_row_indices, uint64_t n_rows, tvm_domain ldt_domain,
                                 const uint64_t* h_weights_main_aux, const uint64_t* h_weights_quot, const uint64_t* h_weights_deep,
                                 const uint64_t* h_ood_points, const uint64_t* h_ood_values, uint64_t* h_out) {
    if (!c || !h_main_rows || !h_aux_rows || !h_quot_rows || !h_row_indices || !n_rows || !valid_domain(ldt_domain) ||
        !h_weights_main_aux || !h_weights_quot || !h_weights_deep || !h_ood_points || !h_ood_values || !h_out)
        return set_error(c, TVM_ERR_INVALID_ARGUMENT, "tvm_verifier_deep_values arguments");
    for (u64 j = 0; j < n_rows; j++)
        if (h_row_indices[j] >= ldt_domain.length) return set_error(c, TVM_ERR_INVALID_ARGUMENT, "verifier: row index out of range");
    const size_t wm = (size_t)n_rows * TVM_NUM_MAIN_COLUMNS, wa = (size_t)n_rows * TVM_NUM_AUX_COLUMNS * 3, wq = (size_t)n_rows * 15;
    const size_t ww = (size_t)3 * (TVM_NUM_MAIN_COLUMNS + TVM_NUM_AUX_COLUMNS), total = wm + wa + wq + n_rows + ww + 51;
    std::vector<u64> host(total);
    u64* p = host.data();
    memcpy(p, h_main_rows, wm * 8); p += wm;
    memcpy(p, h_aux_rows, wa * 8); p += wa;
    memcpy(p, h_quot_rows, wq * 8); p += wq;
    memcpy(p, h_row_indices, n_rows * 8); p += n_rows;
    memcpy(p, h_weights_main_aux, ww * 8); p += ww;
    memcpy(p, h_weights_quot, 15 * 8); p += 15;
    memcpy(p, h_weights_deep, 12 * 8); p += 12;
    memcpy(p, h_ood_points, 12 * 8); p += 12;
    memcpy(p, h_ood_values, 12 * 8);
    u64* d = (u64*)pool_alloc(c, (total + 3 * n_rows) * sizeof(u64));
    if (!d) return set_error(c, TVM_ERR_OUT_OF_MEMORY, "verifier scratch");
    int rc = TVM_OK;
    if (hipMemcpyAsync(d, host.data(), total * sizeof(u64), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        rc = set_error(c, TVM_ERR_DEVICE, "verifier upload");
    u64* d_out = d + total;
    if (rc == TVM_OK)
        rc = verifier_deep_values(c, d, TVM_NUM_MAIN_COLUMNS, d + wm, TVM_NUM_AUX_COLUMNS, d + wm + wa, d + wm + wa + wq, n_rows,
                                  ldt_domain.offset, ldt_domain.generator, d + wm + wa + wq + n_rows, d + wm + wa + wq + n_rows + ww, d_out);
    if (rc == TVM_OK && (hipMemcpyAsync(h_out, d_out, 3 * n_rows * sizeof(u64), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                         hipStreamSynchronize(c->stream) != hipSuccess))
        rc = set_error(c, TVM_ERR_DEVICE, "verifier download");
    pool_release(c, d);
    return rc;
}
}  // extern "C"
