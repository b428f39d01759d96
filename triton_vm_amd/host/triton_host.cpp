// triton_host.cpp -- see triton_host.hpp.  Step order, names and comments follow stark.rs:331-719.
#include "triton_host.hpp"
#include "host_internal.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <iterator>
#include <thread>

namespace triton_vm {

typedef unsigned __int128 u128;
static const u64 R_MOD_P = 0xFFFFFFFFull;  // 2^64 mod p

u64 to_mont(u64 v) { return (u64)((u128)(v % P) * R_MOD_P % P); }
static u64 redc(u128 t) {  // t * 2^-64 mod p for t < p * 2^64
    const u64 lo = (u64)t, hi = (u64)(t >> 64);
    const u64 a = lo + (lo << 32);
    const u64 e = a < lo ? 1 : 0;
    const u64 b = a - (a >> 32) - e;
    const u64 r = hi - b;
    return hi < b ? r - 0xFFFFFFFFull : r;
}
u64 mont_mul(u64 a, u64 b) { return redc((u128)a * b); }
u64 mont_pow(u64 a, u64 e) {
    u64 r = to_mont(1);
    for (; e; e >>= 1, a = mont_mul(a, a))
        if (e & 1) r = mont_mul(r, a);
    return r;
}
u64 generator() { return to_mont(7); }
u64 primitive_root_of_unity(u64 order) {
    if (!order || (order & (order - 1)) || order > (1ull << 32)) throw Error(TVM_ERR_INVALID_ARGUMENT, "PrimitiveRootNotSupported");
    return mont_pow(to_mont(7), (P - 1) / order);
}
u64 bfe_add(u64 a, u64 b) { return (u64)(((u128)a + b) % P); }
Xfe xfe_add(const Xfe& a, const Xfe& b) { return Xfe{{bfe_add(a.c[0], b.c[0]), bfe_add(a.c[1], b.c[1]), bfe_add(a.c[2], b.c[2])}}; }
Xfe xfe_mul(const Xfe& a, const Xfe& b) {
    Xfe o;
    tvm_host_xfe_mul(a.c, b.c, o.c);
    return o;
}
Xfe xfe_scale(const Xfe& a, u64 s) { return Xfe{{mont_mul(a.c[0], s), mont_mul(a.c[1], s), mont_mul(a.c[2], s)}}; }
std::vector<Xfe> xfe_powers(const Xfe& x, u64 first, u64 n) {
    std::vector<Xfe> out(n);
    if (n) tvm_host_xfe_powers(x.c, first, n, out[0].c);
    return out;
}

// ------------------------------------------------------------------------------------------------ Context
void Context::check(int32_t status, const char* what) const {
    if (status != TVM_OK) throw Error(status, std::string(what) + ": " + tvm_status_string(status) + " (" + tvm_last_error(ctx_) + ")");
}
u64* Context::alloc(u64 n_words) const {
    void* p = nullptr;
    check(tvm_malloc(ctx_, (size_t)(n_words ? n_words : 1) * sizeof(u64), &p), "tvm_malloc");
    return (u64*)p;
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
        reset();
        c_ = o.c_, p_ = o.p_, n_ = o.n_;
        o.p_ = nullptr;
    }
    return *this;
}
void DeviceBuffer::reset() {
    if (p_) c_->free(p_);
    p_ = nullptr;
}
std::vector<u64> DeviceBuffer::download(u64 first_word, u64 n_words) const {
    std::vector<u64> out(n_words);
    if (n_words) c_->check(tvm_memcpy_d2h(c_->raw(), out.data(), p_ + first_word, (size_t)n_words * sizeof(u64)), "tvm_memcpy_d2h");
    return out;
}

// ------------------------------------------------------------------------------------------------ ArithmeticDomain
ArithmeticDomain ArithmeticDomain::of_length(u64 length) { return {to_mont(1), primitive_root_of_unity(length), length}; }
ArithmeticDomain ArithmeticDomain::pow(u64 exponent) const {
    if (!exponent || (exponent & (exponent - 1))) throw Error(TVM_ERR_INVALID_ARGUMENT, "IllegalExponent");
    return {mont_pow(offset, exponent), mont_pow(generator, exponent), std::max<u64>(length / exponent, 1)};
}
DeviceBuffer ArithmeticDomain::evaluate(const Context& c, const u64* d_coeffs, u64 n_coeffs, int fk) const {
    DeviceBuffer out(c, length * fk);
    c.check(tvm_evaluate(c.raw(), fk, d_coeffs, n_coeffs, this->c(), out.ptr()), "tvm_evaluate");
    return out;
}
DeviceBuffer ArithmeticDomain::interpolate(const Context& c, const u64* d_values, int fk) const {
    DeviceBuffer out(c, length * fk);
    c.check(tvm_interpolate(c.raw(), fk, d_values, this->c(), out.ptr()), "tvm_interpolate");
    return out;
}

// ------------------------------------------------------------------------------------------------ ProofStream
// BFieldCodec of twenty-first 2.0.0 [not in the reference tree; restated, pinned by the reference's proof-digest snapshot
// through tests/test_proof_snapshot.py]: statically-sized types are their elements in order; Vec<T> of a statically-sized T
// is [number of elements, elements...], of a dynamically-sized T [number of elements, (length_i, element_i)...]; a derived
// struct is its fields LAST FIELD FIRST, each dynamically-sized one prefixed with its length; a derived enum is
// [discriminant, fields as for a struct]; a Polynomial drops its trailing zero coefficients.
namespace {
enum Kind { STATIC = 0, POLYNOMIAL = -1, RESPONSE = -2 };  // > 0: Vec of that many words per element
struct Variant {
    const char* name;
    int kind;
    bool fiat_shamir;
};
// proof_item.rs:96-150 in declaration order (= discriminant)
const Variant VARIANTS[] = {
    {"MerkleRoot", STATIC, true}, {"Log2PaddedHeight", STATIC, true}, {"OutOfDomainMainRow", STATIC, true},
    {"OutOfDomainAuxRow", STATIC, true}, {"OutOfDomainQuotientSegments", STATIC, true}, {"Polynomial", POLYNOMIAL, true},
    {"StirOutOfDomainValues", 3, true}, {"AuthenticationStructure", 5, false}, {"MasterMainTableRows", 379, false},
    {"MasterAuxTableRows", 273, false}, {"QuotientSegmentsElements", 15, false}, {"FriCodeword", 3, false},
    {"FriResponse", RESPONSE, false}, {"StirResponse", RESPONSE, false}};
// the labels this prover enqueues under -> proof item
const struct { const char* prefix; int variant; } LABELS[] = {
    {"log2 padded height", 1}, {"ood main", 2}, {"ood aux", 3}, {"ood quot", 4}, {"fri last codeword", 11},
    {"fri last polynomial", 5}, {"fri response", 12}, {"fri auth", 12}, {"main rows", 8}, {"aux rows", 9}, {"quot rows", 10},
    {"main auth", 7}, {"aux auth", 7}, {"quot auth", 7}, {"main root", 0}, {"aux root", 0}, {"quot root", 0}, {"fri root", 0},
    {"stir root", 0}, {"stir ood values", 6}, {"stir final polynomial", 5}, {"stir response leafs", 13}, {"stir response auth", 13}};
int variant_of(const std::string& label) {
    for (const auto& l : LABELS)
        if (label.compare(0, std::strlen(l.prefix), l.prefix) == 0) return l.variant;
    throw Error(TVM_ERR_INVALID_ARGUMENT, "no proof item for the label " + label);
}
typedef std::vector<u64> Words;
void push_len(Words& v, u64 n) { v.push_back(to_mont(n)); }
void append(Words& v, const Words& w) { v.insert(v.end(), w.begin(), w.end()); }
void append_dynamic(Words& v, const Words& w) { push_len(v, w.size()); append(v, w); }
Words encode_vec(const u64* w, u64 n_words, u64 elem_words) {
    Words out;
    push_len(out, n_words / elem_words);
    out.insert(out.end(), w, w + n_words);
    return out;
}
Words encode_polynomial(const u64* w, u64 n_words) {
    u64 n = n_words / 3;
    while (n && !(w[3 * n - 3] | w[3 * n - 2] | w[3 * n - 1])) n--;
    Words out;
    append_dynamic(out, encode_vec(w, 3 * n, 3));  // struct { coefficients: Vec<XFieldElement> }
    return out;
}
Words encode_item(int variant, const Words& payload, const Words* auth_structure, u64 stack_words = 0) {
    const Variant& v = VARIANTS[variant];
    Words out;
    push_len(out, (u64)variant);
    if (v.kind == STATIC) {
        append(out, payload);
    } else if (v.kind > 0) {
        append_dynamic(out, encode_vec(payload.data(), payload.size(), (u64)v.kind));
    } else if (v.kind == POLYNOMIAL) {
        append_dynamic(out, encode_polynomial(payload.data(), payload.size()));
    } else {  // FriResponse { queried_leaves: Vec<XFieldElement>, auth_structure: Vec<Digest> } (fri.rs:101-108)
        Words response, leaves;
        append_dynamic(response, encode_vec(auth_structure->data(), auth_structure->size(), 5));
        if (stack_words) {  // StirResponse { queried_leafs: Vec<Vec<XFieldElement>>, .. } (stir.rs:150-168)
            push_len(leaves, payload.size() / stack_words);
            for (u64 at = 0; at < payload.size(); at += stack_words) append_dynamic(leaves, encode_vec(payload.data() + at, stack_words, 3));
        } else {
            leaves = encode_vec(payload.data(), payload.size(), 3);
        }
        append_dynamic(response, leaves);
        append_dynamic(out, response);
    }
    return out;
}
}  // namespace

Words Claim::encode() const {  // proof.rs:62-84: program_digest, version, input, output
    Words out;
    append_dynamic(out, encode_vec(output.data(), output.size(), 1));
    append_dynamic(out, encode_vec(input.data(), input.size(), 1));
    push_len(out, version);
    out.insert(out.end(), program_digest, program_digest + 5);
    return out;
}

void ProofStream::alter_fiat_shamir_state_with(const Words& encoding) {  // proof_stream.rs:40-42
    tvm_host_sponge_pad_and_absorb(state_, encoding.data(), encoding.size());
}
void ProofStream::enqueue(const std::string& name, const u64* words, u64 n, u64 stack_words) {
    // proof_stream.rs:54-59: the item always goes into the proof; it alters the sponge only if
    // ProofItem::include_in_fiat_shamir_heuristic says so (proof_item.rs:96-150: roots, out-of-domain rows, polynomials do;
    // authentication structures, opened rows, the FRI codeword and responses do not -- the prover is already committed to
    // them through a Merkle root)
    const int variant = variant_of(name);
    const bool fiat_shamir = VARIANTS[variant].fiat_shamir;
    items_.push_back(Item{name, Words(words, words + n), fiat_shamir, stack_words});
    if (fiat_shamir) alter_fiat_shamir_state_with(encode_item(variant, items_.back().words, nullptr));
}
namespace {
// encode_item appended to `out` in place (length prefixes are reserved and filled in afterwards): the proof is 2 MB at 2^20
// rows and is assembled between two device phases
struct InPlace {
    Words& v;
    size_t open() { v.push_back(0); return v.size() - 1; }
    void close(size_t at) { v[at] = to_mont(v.size() - at - 1); }
    void vec(const u64* w, u64 n_words, u64 elem_words) { push_len(v, n_words / elem_words); v.insert(v.end(), w, w + n_words); }
};
void encode_item_into(Words& out, int variant, const Words& payload, const Words* auth_structure, u64 stack_words) {
    const Variant& var = VARIANTS[variant];
    InPlace e{out};
    push_len(out, (u64)variant);
    if (var.kind == STATIC) {
        append(out, payload);
    } else if (var.kind > 0) {
        const size_t a = e.open();
        e.vec(payload.data(), payload.size(), (u64)var.kind);
        e.close(a);
    } else if (var.kind == POLYNOMIAL) {
        append_dynamic(out, encode_polynomial(payload.data(), payload.size()));
    } else {  // see encode_item
        const size_t response = e.open();
        const size_t auth = e.open();
        e.vec(auth_structure->data(), auth_structure->size(), 5);
        e.close(auth);
        const size_t leaves = e.open();
        if (stack_words) {
            push_len(out, payload.size() / stack_words);
            for (u64 at = 0; at < payload.size(); at += stack_words) {
                const size_t one = e.open();
                e.vec(payload.data() + at, stack_words, 3);
                e.close(one);
            }
        } else {
            e.vec(payload.data(), payload.size(), 3);
        }
        e.close(leaves);
        e.close(response);
    }
}
}  // namespace
Words ProofStream::proof() const {  // impl From<&ProofStream> for Proof, proof_stream.rs:115-119
    size_t words = 16;
    for (const Item& it : items_) words += it.words.size() + it.words.size() / 8 + 16;
    Words out;
    out.reserve(words);
    InPlace e{out};
    const size_t vec = e.open();   // struct ProofStream { items: Vec<ProofItem>, .. }
    out.push_back(0);              // the number of items
    u64 count = 0;
    for (size_t k = 0; k < items_.size(); k++, count++) {
        const int variant = variant_of(items_[k].name);
        const bool response = VARIANTS[variant].kind == RESPONSE;  // its leaves and its authentication structure: one item
        const size_t item = e.open();
        encode_item_into(out, variant, items_[k].words, response ? &items_[k + 1].words : nullptr, items_[k].stack_words);
        e.close(item);
        if (response) k++;
    }
    out[vec + 1] = to_mont(count);
    e.close(vec);
    return out;
}
void ProofStream::squeeze(u64 out[10]) {
    std::memcpy(out, state_, 10 * sizeof(u64));
    tvm_host_tip5_permutation(state_);
}
std::vector<Xfe> ProofStream::sample_scalars(u64 n) {
    std::vector<u64> words;
    for (u64 k = 0; k < (3 * n + 9) / 10; k++) {
        u64 w[10];
        squeeze(w);
        words.insert(words.end(), w, w + 10);
    }
    std::vector<Xfe> out(n);
    for (u64 i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) out[i].c[k] = words[3 * i + k];
    return out;
}
std::vector<u64> ProofStream::sample_indices(u64 upper_bound, u64 n) {
    std::vector<u64> out;  // [twenty-first Tip5::sample_indices] squeezed elements in order, P - 1 skipped
    while (out.size() < n) {
        u64 w[10];
        squeeze(w);
        for (int k = 0; k < 10 && out.size() < n; k++) {
            const u64 v = mont_mul(w[k], 1);  // the canonical value
            if (v != P - 1) out.push_back(v % upper_bound);
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ MasterTable
MasterTable::MasterTable(const Context& c, int field_kind, const u64* d_trace, u64 n_rows, u64 n_cols, const u64* d_randomizers,
                         u64 num_trace_randomizers, ArithmeticDomain trace, ArithmeticDomain quotient, ArithmeticDomain ldt)
    : c_(c), fk_(field_kind), d_trace_(d_trace), d_rnd_(d_randomizers), n_rows_(n_rows), n_cols_(n_cols),
      h_(num_trace_randomizers), trace_(trace), quotient_(quotient), ldt_(ldt) {}
ArithmeticDomain MasterTable::evaluation_domain() const { return quotient_.length > ldt_.length ? quotient_ : ldt_; }
void MasterTable::maybe_low_degree_extend_all_columns() {
    clear_cache();  // the old table's block goes back to the context's pool and is reused right away
    c_.check(tvm_lde_table(c_.raw(), fk_, d_trace_, n_rows_, n_cols_, d_rnd_, h_, trace_.c(), evaluation_domain().c(), &table_),
             "tvm_lde_table");
}
void MasterTable::low_degree_extend_over(const tvmh_comm* comm, unsigned chunks, const std::function<void(u64)>& sent) {
    if (!comm || !chunks || chunks > 16) throw Error(TVM_ERR_INVALID_ARGUMENT, "low_degree_extend_over: a communicator and 1 .. 16 chunks");
    const u64 R = comm->world, me = comm->rank, W = n_cols_ * (u64)fk_, n = n_rows_;
    const u64 cpc = (W + R * chunks - 1) / (R * chunks), per = cpc * chunks;   // virtual columns per (rank, chunk) / per rank
    const ArithmeticDomain ev = evaluation_domain();
    clear_cache();
    // this rank's block, interpolated (the unused tail of a short last block travels as it is and is never read)
    DeviceBuffer mine(c_, per * n);
    const u64 first = std::min(me * per, W), count = std::min(W - first, per);
    c_.check(tvm_lde_column_coefficients(c_.raw(), fk_, d_trace_, n, n_cols_, trace_.c(), first, count, mine.ptr()), "tvm_lde_column_coefficients");
    const bool async = comm->all_gather_async && comm->wait;
    auto status = [&](int32_t st, const char* what) {
        if (st != TVM_OK) throw Error(st, std::string(what) + ": the communicator reported " + tvm_status_string(st));
    };
    // Two chunks' receive buffers at a time: the exchange of chunk k + 1 is requested before chunk k is extended (with an
    // asynchronous communicator it runs under that extension), and a chunk's buffer goes back to the pool once its columns are
    // written -- R * W / (R * chunks) columns of coefficients live per buffer, not all W.
    std::vector<DeviceBuffer> all(chunks);
    // An exchange on the communicator's side lane reads `mine` and writes all[k] outside the order of the context's stream -- the
    // order the pool reuses freed blocks in.  Whatever way this function is left (a failed launch, a failed peer), every requested
    // exchange is waited for BEFORE the buffers below go back to the pool: wait() puts the side lane's mark into the context's stream.
    struct Outstanding {
        const tvmh_comm* comm;
        tvm_ctx* ctx;
        uint32_t requested = 0, waited = 0;   // slots [waited, requested) are in flight
        ~Outstanding() {
            for (uint32_t k = waited; k < requested; k++) (void)comm->wait(comm->self, ctx, k);
        }
    } outstanding{comm, c_.raw()};
    auto request = [&](unsigned k) {
        all[k] = DeviceBuffer(c_, R * cpc * n);
        if (async) outstanding.requested = k + 1;   // (counted before the call: a request that failed half-way may have queued its exchange)
        if (async) status(comm->all_gather_async(comm->self, c_.raw(), mine.ptr() + k * cpc * n, all[k].ptr(), cpc * n, k), "coefficients (all-gather)");
        else status(comm->all_gather(comm->self, c_.raw(), mine.ptr() + k * cpc * n, all[k].ptr(), cpc * n), "coefficients (all-gather)");
        if (sent) sent(cpc * n * 8 * (R - 1));
    };
    request(0);
    c_.check(tvm_lde_table_begin(c_.raw(), fk_, n, n_cols_, h_, trace_.c(), ev.c(), &table_), "tvm_lde_table_begin");
    for (unsigned k = 0; k < chunks; k++) {
        if (k + 1 < chunks) request(k + 1);
        if (async) {
            outstanding.waited = k + 1;
            status(comm->wait(comm->self, c_.raw(), k), "coefficients (wait)");
        }
        for (u64 r = 0; r < R; r++) {
            const u64 col0 = r * per + k * cpc;
            if (col0 >= W) continue;
            c_.check(tvm_lde_table_add_columns(c_.raw(), table_, all[k].ptr() + r * cpc * n, col0, std::min(W - col0, cpc), d_rnd_, h_, trace_.c(),
                                               ev.c()), "tvm_lde_table_add_columns");
        }
        all[k].reset();   // (stream-ordered: the next request's exchange is ordered behind the kernels that read this buffer)
    }
    c_.check(tvm_lde_table_end(c_.raw(), table_), "tvm_lde_table_end");
}
void MasterTable::clear_cache() {
    if (table_) tvm_table_free(c_.raw(), table_);
    table_ = nullptr;
}
const tvm_table* MasterTable::table() const {
    if (!table_) throw Error(TVM_ERR_INVALID_ARGUMENT, "low-degree extend first (maybe_low_degree_extend_all_columns)");
    return table_;
}
DeviceBuffer MasterTable::merkle_tree() const {
    DeviceBuffer nodes(c_, 10 * ldt_.length);
    c_.check(tvm_table_merkle_tree(c_.raw(), table(), ldt_.length, nodes.ptr()), "tvm_table_merkle_tree");
    return nodes;
}
std::vector<u64> MasterTable::reveal_rows(const std::vector<u64>& idx) const {
    std::vector<u64> out(idx.size() * n_cols_ * fk_);
    c_.check(tvm_table_reveal_rows(c_.raw(), table(), ldt_.length, idx.data(), idx.size(), out.data()), "tvm_table_reveal_rows");
    return out;
}
std::vector<u64> MasterTable::out_of_domain_rows(const std::vector<Xfe>& points) const {
    std::vector<u64> out(points.size() * n_cols_ * 3);
    c_.check(tvm_out_of_domain_rows(c_.raw(), fk_, d_trace_, n_rows_, n_cols_, d_rnd_, h_, trace_.c(), points[0].c,
                                    (uint32_t)points.size(), out.data()), "tvm_out_of_domain_rows");
    return out;
}
std::vector<u64> MasterTable::out_of_domain_rows(const std::vector<Xfe>& points, u64 first_col, u64 n) const {
    std::vector<u64> out(points.size() * n * 3);
    if (n == 0) return out;
    if (first_col + n > n_cols_) throw Error(TVM_ERR_INVALID_ARGUMENT, "out_of_domain_rows: column range");
    c_.check(tvm_out_of_domain_rows(c_.raw(), fk_, d_trace_ + first_col * n_rows_ * fk_, n_rows_, n, d_rnd_ + first_col * h_ * fk_, h_,
                                    trace_.c(), points[0].c, (uint32_t)points.size(), out.data()), "tvm_out_of_domain_rows");
    return out;
}
std::vector<u64> MasterTable::reveal_rows_of(const std::vector<u64>& idx, u64 view_rows) const {
    std::vector<u64> out(idx.size() * n_cols_ * fk_);
    if (!idx.empty()) c_.check(tvm_table_reveal_rows(c_.raw(), table(), view_rows, idx.data(), idx.size(), out.data()), "tvm_table_reveal_rows");
    return out;
}
void MasterTable::set_domains(ArithmeticDomain quotient, ArithmeticDomain ldt) {
    clear_cache();
    quotient_ = quotient;
    ldt_ = ldt;
}
DeviceBuffer MasterTable::weighted_sum_of_columns(const Xfe* weights) const {
    DeviceBuffer poly(c_, 2 * n_rows_ * 3);
    c_.check(tvm_weighted_sum_of_columns(c_.raw(), fk_, d_trace_, n_rows_, n_cols_, d_rnd_, h_, trace_.c(), weights[0].c, poly.ptr()),
             "tvm_weighted_sum_of_columns");
    return poly;
}

// ------------------------------------------------------------------------------------------------ parameters
unsigned bit_length(u64 v) {
    unsigned n = 0;
    for (; v; v >>= 1) n++;
    return n;
}
StarkParameters::StarkParameters(unsigned log2_padded_height, u64 num_trace_randomizers, u64 checks, unsigned log2_expansion)
    : padded_height(1ull << log2_padded_height), h(num_trace_randomizers), num_collinearity_checks(checks) {
    const u64 rtl = std::max({padded_height + h, 2 * h + 1, (h + 1) * 5});
    randomized_trace_len = 1ull << bit_length(rtl - 1);
    trace = ArithmeticDomain::of_length(randomized_trace_len / 2);
    const u64 max_degree = 4 * (randomized_trace_len - 1) - 1;
    const u64 quotient_len = 1ull << bit_length(max_degree - 1);
    const u64 g = generator();
    ldt = ArithmeticDomain::of_length(randomized_trace_len << log2_expansion).with_offset(g);
    quotient = ArithmeticDomain::of_length(quotient_len).with_offset(g);
    const int max_rounds = (int)bit_length(randomized_trace_len - 1);
    const int rounds = max_rounds - ((int)bit_length(checks) - 1) - 1;
    fri_rounds = rounds > 0 ? (unsigned)rounds : 0;
    num_quotient_randomizers = (h + 1) * 5;
    this->log2_expansion = log2_expansion;
}

// ------------------------------------------------------------------------------------------------ helpers of prove

std::vector<u64> merkle_root(const Context& c, const DeviceBuffer& nodes) { return nodes.download(5, 5); }  // node 1; drains the stream

// [twenty-first MerkleTree::authentication_structure, restated] the nodes a verifier cannot compute from the revealed
// leaves -- the siblings along the paths that are not themselves on a path -- in descending heap order, gathered to the host
std::vector<u64> auth_node_indices(u64 n_leaves, const std::vector<u64>& indices) {
    // Level by level on the sorted list of path nodes: a sibling is needed unless it is a path node itself (then it sits next
    // to its sibling in the sorted list); nodes of different levels have disjoint index ranges, so the levels do not interact.
    // Deeper levels have the larger heap indices: appending each level's siblings in descending order, deepest level first,
    // IS the descending heap order.  (O(indices x depth), no sets: this runs between two device phases with the GPU idle.)
    std::vector<u64> k;
    k.reserve(indices.size());
    for (u64 i : indices) k.push_back(i + n_leaves);
    std::sort(k.begin(), k.end());
    k.erase(std::unique(k.begin(), k.end()), k.end());
    std::vector<u64> need, level;
    while (!k.empty() && k[0] > 1) {
        level.clear();
        for (size_t i = 0; i < k.size(); i++) {
            const u64 x = k[i], sibling = x ^ 1;
            const bool on_a_path = (x & 1) ? (i > 0 && k[i - 1] == sibling) : (i + 1 < k.size() && k[i + 1] == sibling);
            if (!on_a_path) level.push_back(sibling);
        }
        need.insert(need.end(), level.rbegin(), level.rend());
        size_t m = 0;
        for (size_t i = 0; i < k.size(); i++) {
            const u64 parent = k[i] >> 1;
            if (m == 0 || k[m - 1] != parent) k[m++] = parent;
        }
        k.resize(m);
    }
    return need;
}
static std::vector<u64> auth_nodes(const Context& c, const DeviceBuffer& nodes, u64 n_leaves, const std::vector<u64>& indices) {
    const std::vector<u64> need = auth_node_indices(n_leaves, indices);
    std::vector<u64> out(need.size() * 5);
    if (!need.empty()) c.check(tvm_gather_elements(c.raw(), nodes.ptr(), 5, need.data(), need.size(), out.data()), "tvm_gather_elements");
    return out;
}

// Challenges::new (challenges.rs:85-121): the 59 sampled challenges, then the terminals of the public input, the public
// output, the lookup table and the program digest -- EvalArg::compute_terminal(symbols, 1, indeterminate)
std::vector<Xfe> derive_challenges(std::vector<Xfe> ch, const Claim& claim) {
    auto terminal = [](const u64* symbols, u64 n, const Xfe& x) {
        Xfe acc{{to_mont(1), 0, 0}};
        for (u64 i = 0; i < n; i++) {
            acc = xfe_mul(acc, x);
            acc.c[0] = bfe_add(acc.c[0], symbols[i]);
        }
        return acc;
    };
    u64 lut[256];  // [Tip5] L(x) = (x + 1)^3 - 1 mod 257
    for (u64 x = 0; x < 256; x++) lut[x] = to_mont(((x + 1) * (x + 1) % 257 * (x + 1) % 257 + 256) % 257);
    const Xfe input = terminal(claim.input.data(), claim.input.size(), ch[1]);      // StandardInputIndeterminate
    const Xfe output = terminal(claim.output.data(), claim.output.size(), ch[2]);   // StandardOutputIndeterminate
    const Xfe lookup = terminal(lut, 256, ch[54]);                                  // LookupTablePublicIndeterminate
    const Xfe digest = terminal(claim.program_digest, 5, ch[0]);                    // CompressProgramDigestIndeterminate
    ch.insert(ch.end(), {input, output, lookup, digest});
    return ch;
}

Prover::Prover(const Context& c, const StarkParameters& p, const u64* d_main_trace, const u64* d_main_randomizers,
               const u64* d_aux_trace, const u64* d_aux_randomizers, const std::vector<Xfe>& quotient_randomizer, const Claim& claim)
    : c_(c), p_(p), claim_(claim), main_(c, 1, d_main_trace, p.trace.length, NUM_MAIN, d_main_randomizers, p.h, p.trace, p.quotient, p.ldt),
      aux_(c, 3, d_aux_trace, p.trace.length, NUM_AUX, d_aux_randomizers, p.h, p.trace, p.quotient, p.ldt),
      quotient_randomizer_(quotient_randomizer) {
    if (quotient_randomizer.size() != p.num_quotient_randomizers) throw Error(TVM_ERR_INVALID_ARGUMENT, "quotient randomizer length");
}

// Fri::prove (fri.rs:212-319, 754-772): commit and fold round by round, send the last codeword and polynomial,
// answer the queries.  Returns the first-round indices.
std::vector<u64> Prover::fri(const DeviceBuffer& combination, ProofStream& ps) {
    struct Round {
        ArithmeticDomain dom;
        const u64* cw;
        DeviceBuffer nodes;
    };
    std::vector<Round> rounds;
    std::vector<DeviceBuffer> folded;  // owns the codewords of rounds 1..
    ArithmeticDomain dom = p_.ldt;
    const u64* cw = combination.ptr();
    {
        // The commit phase in one call, the sponge on the device (tvm_fri_commit_phase): trees, roots into the transcript,
        // folding challenges, folds -- then the same enqueues and samplings are replayed on this host's sponge, which must
        // arrive at the same challenges.
        std::vector<u64*> d_cw, d_nodes;
        ArithmeticDomain d = dom;
        for (unsigned r = 0; r <= p_.fri_rounds; r++) {
            rounds.push_back(Round{d, nullptr, DeviceBuffer(c_, 10 * d.length)});
            d_nodes.push_back(rounds.back().nodes.ptr());
            if (r == p_.fri_rounds) break;
            folded.emplace_back(c_, d.length / 2 * 3);
            d_cw.push_back(folded.back().ptr());
            d = d.pow(2);
        }
        std::vector<u64> roots(5 * (p_.fri_rounds + 1)), challenges(3 * (size_t)p_.fri_rounds + 1);
        c_.check(tvm_fri_commit_phase(c_.raw(), cw, dom.c(), p_.fri_rounds, ps.sponge_state(), d_cw.data(), d_nodes.data(), roots.data(),
                                      challenges.data()), "tvm_fri_commit_phase");
        for (unsigned r = 0; r <= p_.fri_rounds; r++) {
            rounds[r].cw = r == 0 ? cw : folded[r - 1].ptr();
            ps.enqueue("fri root " + std::to_string(r), &roots[5 * r], 5);
            if (r == p_.fri_rounds) break;
            const Xfe challenge = ps.sample_scalars(1)[0];
            if (std::memcmp(challenge.c, &challenges[3 * r], 3 * sizeof(u64)) != 0)
                throw Error(TVM_ERR_DEVICE, "the device's Fiat-Shamir sponge and the host's disagree on a FRI folding challenge");
        }
        cw = rounds.back().cw;
        dom = rounds.back().dom;
    }
    std::vector<u64> last(dom.length * 3);
    c_.check(tvm_memcpy_d2h(c_.raw(), last.data(), cw, last.size() * sizeof(u64)), "last codeword");
    ps.enqueue("fri last codeword", last.data(), last.size());
    const DeviceBuffer last_poly_d = ArithmeticDomain::of_length(dom.length).interpolate(c_, cw, 3);
    const std::vector<u64> last_poly = last_poly_d.download(0, dom.length * 3);
    ps.enqueue("fri last polynomial", last_poly.data(), last_poly.size());
    last_polynomial.resize(dom.length);
    std::memcpy(last_polynomial.data(), last_poly.data(), last_poly.size() * sizeof(u64));
    const std::vector<u64> a_indices = ps.sample_indices(p_.ldt.length, p_.num_collinearity_checks);
    // the responses of all rounds in one round trip to the device (their order in the proof stream is fixed below)
    GatherBatch batch;
    struct Response {
        size_t round, leaves, auth;
    };
    std::vector<Response> responses;
    for (size_t r = 0; r < rounds.size(); r++) {
        const Round& round = rounds[r];
        std::vector<u64> b_idx;
        for (u64 i : a_indices) b_idx.push_back((i % round.dom.length + round.dom.length / 2) % round.dom.length);
        for (int which = (r == 0 ? 0 : 1); which < 2; which++) {
            if (which == 1 && r == rounds.size() - 1) continue;
            const std::vector<u64>& ix = which == 0 ? a_indices : b_idx;
            const size_t leaves = batch.add(round.cw, 3, ix);
            const size_t auth = batch.add(round.nodes.ptr(), 5, auth_node_indices(round.dom.length, ix));
            responses.push_back(Response{r, leaves, auth});
        }
    }
    batch.run(c_);
    for (const Response& q : responses) {
        const std::vector<u64>&leaves = batch.jobs[q.leaves].out, &auth = batch.jobs[q.auth].out;
        ps.enqueue("fri response " + std::to_string(q.round), leaves.data(), leaves.size());
        ps.enqueue("fri auth " + std::to_string(q.round), auth.data(), auth.size());
    }
    (void)ps.sample_scalars(1);
    return a_indices;
}

ProofStream Prover::prove() {
    ProofStream ps;
    ps.alter_fiat_shamir_state_with(claim_.encode());  // stark.rs:336-339
    {
        const u64 log2_padded_height = to_mont(bit_length(p_.padded_height) - 1);  // stark.rs:354
        ps.enqueue("log2 padded height", &log2_padded_height, 1);
    }
    const u64 L = p_.ldt.length;
    const ArithmeticDomain short_dom = p_.ldt.length <= p_.quotient.length ? p_.ldt : p_.quotient;
    const u64 zeta = to_mont(3);  // Stark::ZETA, stark.rs:1801
    auto enqueue_xfes = [&](const char* name, const std::vector<Xfe>& v) { ps.enqueue(name, v[0].c, 3 * v.size()); };

    // 4-6: main table LDE, Merkle tree, challenges  (stark.rs:367-377)
    main_.maybe_low_degree_extend_all_columns();
    const DeviceBuffer main_nodes = main_.merkle_tree();
    ps.enqueue("main root", merkle_root(c_, main_nodes).data(), 5);
    const std::vector<Xfe> challenges = derive_challenges(ps.sample_scalars(NUM_SAMPLED_CHALLENGES), claim_);
    if (extend) extend(challenges);  // MasterMainTable::extend (stark.rs:379-381)

    // 8-9: aux table (its `extend` is host work in the reference; the trace is already resident)
    aux_.maybe_low_degree_extend_all_columns();
    const DeviceBuffer aux_nodes = aux_.merkle_tree();
    ps.enqueue("aux root", merkle_root(c_, aux_nodes).data(), 5);
    const std::vector<Xfe> quotient_weights = xfe_powers(ps.sample_scalars(1)[0], 0, TVM_NUM_QUOTIENT_WEIGHTS);

    // 10: quotient codeword, segments, randomization  (stark.rs:405-423)
    DeviceBuffer quot(c_, p_.quotient.length * 3);
    if (assume_valid_trace) c_.check(tvm_ctx_set_option(c_.raw(), TVM_OPTION_AIR_VALID_TRACE, 1), "tvm_ctx_set_option");
    const int32_t quotient_status = tvm_all_quotients_combined(c_.raw(), main_.table(), aux_.table(), p_.trace.c(), p_.quotient.c(),
                                                               challenges[0].c, quotient_weights[0].c, quot.ptr());
    if (assume_valid_trace) (void)tvm_ctx_set_option(c_.raw(), TVM_OPTION_AIR_VALID_TRACE, 0);
    c_.check(quotient_status, "tvm_all_quotients_combined");
    const u64 poly_len = std::max<u64>(p_.quotient.length / 4, quotient_randomizer_.size());
    DeviceBuffer polys(c_, 5 * poly_len * 3);
    tvm_table* seg_table = nullptr;
    c_.check(tvm_quotient_segments(c_.raw(), quot.ptr(), p_.quotient.c(), p_.ldt.c(), quotient_randomizer_.data()->c,
                                   quotient_randomizer_.size(), zeta, &seg_table, polys.ptr(), poly_len), "tvm_quotient_segments");
    struct TableGuard {
        const Context& c;
        tvm_table* t;
        ~TableGuard() { tvm_table_free(c.raw(), t); }
    } seg_guard{c_, seg_table};
    quot.reset();
    // 12: quotient Merkle tree  (stark.rs:425-446)
    DeviceBuffer quot_nodes(c_, 10 * L);
    c_.check(tvm_table_merkle_tree(c_.raw(), seg_table, L, quot_nodes.ptr()), "quotient merkle tree");
    ps.enqueue("quot root", merkle_root(c_, quot_nodes).data(), 5);

    // 13: out-of-domain rows  (stark.rs:450-495)
    const Xfe alpha = ps.sample_scalars(1)[0];
    const Xfe alpha_next = xfe_scale(alpha, p_.trace.generator);
    const std::vector<u64> ood_main = main_.out_of_domain_rows({alpha, alpha_next});
    const std::vector<u64> ood_aux = aux_.out_of_domain_rows({alpha, alpha_next});
    const Xfe a4 = xfe_powers(alpha, 4, 1)[0];
    const Xfe za4 = xfe_powers(xfe_scale(alpha, zeta), 4, 1)[0];
    Xfe seg_ood[5][2];
    {
        const Xfe pts[2] = {a4, za4};   // the five segment polynomials at both points: one round trip
        c_.check(tvm_evaluate_polys_at_points(c_.raw(), polys.ptr(), poly_len, poly_len, 5, pts[0].c, 2, seg_ood[0][0].c),
                 "tvm_evaluate_polys_at_points");
    }
    ps.enqueue("ood main", ood_main.data(), NUM_MAIN * 3);
    ps.enqueue("ood aux", ood_aux.data(), NUM_AUX * 3);
    ps.enqueue("ood main next", ood_main.data() + NUM_MAIN * 3, NUM_MAIN * 3);
    ps.enqueue("ood aux next", ood_aux.data() + NUM_AUX * 3, NUM_AUX * 3);
    enqueue_xfes("ood quot p", {seg_ood[0][0], seg_ood[1][0], seg_ood[2][0], seg_ood[3][0]});
    enqueue_xfes("ood quot r", {seg_ood[1][1], seg_ood[2][1], seg_ood[3][1], seg_ood[4][1]});

    // 14-15: combination weights, linear combinations  (stark.rs:497-543)
    const std::vector<Xfe> w3 = ps.sample_scalars(3);
    const std::vector<Xfe> weights_ma = xfe_powers(w3[0], 0, NUM_MAIN + NUM_AUX);
    const std::vector<Xfe> weights_q = xfe_powers(w3[1], 0, 5);
    const std::vector<Xfe> weights_d = xfe_powers(w3[2], 0, 4);
    DeviceBuffer comb = main_.weighted_sum_of_columns(&weights_ma[0]);
    {
        const DeviceBuffer comb_aux = aux_.weighted_sum_of_columns(&weights_ma[NUM_MAIN]);
        c_.check(tvm_xfe_add_assign(c_.raw(), comb.ptr(), comb_aux.ptr(), 2 * p_.trace.length), "tvm_xfe_add_assign");
    }
    const u64 n_comb = p_.trace.length + p_.h;
    const DeviceBuffer main_aux_codeword = short_dom.evaluate(c_, comb.ptr(), n_comb, 3);
    std::vector<Xfe> wp = weights_q, wr = weights_q;
    wp[4] = Xfe{{0, 0, 0}};
    wr[0] = Xfe{{0, 0, 0}};
    // values of the P and R polynomials on the short domain (stark.rs:536-539): its points are the rows i * L/|short| of
    // the segment table, which was evaluated on the LDT domain
    DeviceBuffer cw_p(c_, short_dom.length * 3), cw_r(c_, short_dom.length * 3);
    c_.check(tvm_table_linear_combination(c_.raw(), seg_table, short_dom.length, wp[0].c, cw_p.ptr()), "tvm_table_linear_combination");
    c_.check(tvm_table_linear_combination(c_.raw(), seg_table, short_dom.length, wr[0].c, cw_r.ptr()), "tvm_table_linear_combination");
    Xfe ma_values[2];
    {
        const Xfe pts[2] = {alpha, alpha_next};
        c_.check(tvm_evaluate_at_points(c_.raw(), comb.ptr(), n_comb, pts[0].c, 2, ma_values[0].c), "tvm_evaluate_at_points");
    }
    Xfe p_value{{0, 0, 0}}, r_value{{0, 0, 0}};
    for (int k = 0; k < 4; k++) p_value = xfe_add(p_value, xfe_mul(weights_q[k], seg_ood[k][0]));
    for (int k = 1; k < 5; k++) r_value = xfe_add(r_value, xfe_mul(weights_q[k], seg_ood[k][1]));

    // 16: DEEP  (stark.rs:545-639)
    DeviceBuffer combination(c_, short_dom.length * 3);
    {
        const u64* cws[4] = {main_aux_codeword.ptr(), main_aux_codeword.ptr(), cw_p.ptr(), cw_r.ptr()};
        const Xfe points[4] = {alpha, alpha_next, a4, za4}, values[4] = {ma_values[0], ma_values[1], p_value, r_value};
        c_.check(tvm_deep_codeword(c_.raw(), 4, cws, short_dom.c(), points[0].c, values[0].c, weights_d[0].c, combination.ptr()),
                 "tvm_deep_codeword");
    }
    cw_p.reset();
    cw_r.reset();
    comb.reset();
    if (short_dom.length != L) {  // stark.rs:629-639: the quotient domain was the short one -- extend to the LDT domain
        const DeviceBuffer coeffs = p_.quotient.interpolate(c_, combination.ptr(), 3);
        combination = p_.ldt.evaluate(c_, coeffs.ptr(), p_.quotient.length, 3);
    }

    // 17: the low-degree test  (stark.rs:641-663)
    const std::vector<u64> a_indices = p_.use_stir ? p_.stir.prove(c_, combination.ptr(), ps) : fri(combination, ps);

    // 18: the out-of-domain point must not collide with a revealed in-domain point  (stark.rs:645-663)
    if (a4.c[1] == 0 && a4.c[2] == 0) {
        const u64 other = mont_mul(a4.c[0], mont_pow(zeta, 4));
        for (u64 i : a_indices) {
            const u64 x = p_.ldt.value(i);
            if (x == a4.c[0] || x == other) throw Error(TVM_ERR_INVALID_ARGUMENT, "ZeroKnowledgeViolation (stark.rs:645-663)");
        }
    }

    // 19: open the trace leafs  (stark.rs:665-716): the three trees have the same shape and the same revealed leaves, hence
    // the same authentication-structure node indices; their nodes come back in one round trip
    {
        const std::vector<u64> auth_idx = auth_node_indices(L, a_indices);
        GatherBatch batch;
        const size_t a_main = batch.add(main_nodes.ptr(), 5, auth_idx), a_aux = batch.add(aux_nodes.ptr(), 5, auth_idx),
                     a_quot = batch.add(quot_nodes.ptr(), 5, auth_idx);
        batch.run(c_);
        const std::vector<u64> main_rows = main_.reveal_rows(a_indices), aux_rows = aux_.reveal_rows(a_indices);
        std::vector<u64> qrows(a_indices.size() * 15);
        c_.check(tvm_table_reveal_rows(c_.raw(), seg_table, L, a_indices.data(), a_indices.size(), qrows.data()), "quotient rows");
        ps.enqueue("main rows", main_rows.data(), main_rows.size());
        ps.enqueue("main auth", batch.jobs[a_main].out.data(), batch.jobs[a_main].out.size());
        ps.enqueue("aux rows", aux_rows.data(), aux_rows.size());
        ps.enqueue("aux auth", batch.jobs[a_aux].out.data(), batch.jobs[a_aux].out.size());
        ps.enqueue("quot rows", qrows.data(), qrows.size());
        ps.enqueue("quot auth", batch.jobs[a_quot].out.data(), batch.jobs[a_quot].out.size());
    }
    main_.clear_cache();
    aux_.clear_cache();
    c_.check(tvm_sync(c_.raw()), "tvm_sync");
    return ps;
}

// ------------------------------------------------------------------------------------------------ STIR
namespace {
const double LOG2_FIELD_SIZE_F = 191.99999999899228;  // ReedSolomonCode::LOG2_FIELD_SIZE (low_degree_test/mod.rs:226)
const int LOG2_FIELD_SIZE = 64 * 3, LOG2_DOMAIN_SHRINKAGE = 1, LOG2_FOLDING_FACTOR = 2;  // stir.rs:404-412
// ReedSolomonCode with proven soundness (mod.rs:93-170)
double proximity_parameter(unsigned log2_expansion) {
    const double margin = std::sqrt(1.0 / (double)(1ull << log2_expansion));
    return 1.0 - margin - margin / 20.0;
}
double log2_list_size(unsigned log2_expansion) {
    const double rate = 1.0 / (double)(1ull << log2_expansion);
    return std::log2(1.0 / (2.0 * std::sqrt(rate) * (std::sqrt(rate) / 20.0)));
}
double log2_binomial_coefficient(u64 a, u64 b) {  // stir.rs:779-793: Kahan-compensated sum of log2 terms
    double log2_binom = 0.0, compensation = 0.0;
    for (u64 i = 0; i < std::min(b, a - b); i++) {
        const double summand = std::log2((double)(a - i)) - std::log2((double)(i + 1));
        const double corrected = summand - compensation;
        const double next = log2_binom + corrected;
        compensation = (next - log2_binom) - corrected;
        log2_binom = next;
    }
    return log2_binom;
}
u64 num_in_domain_queries(unsigned security_level, unsigned log2_domain_size, unsigned log2_expansion) {  // stir.rs:597-700
    u64 uniques = (u64)std::ceil(-(double)security_level / std::log2(1.0 - proximity_parameter(log2_expansion)));
    uniques = std::min<u64>(uniques, 1ull << log2_domain_size);
    const u64 k_minus_1 = uniques - 1, domain_len = 1ull << log2_domain_size;
    const double log2_u_choose_l = log2_binomial_coefficient(domain_len, std::min(k_minus_1, domain_len / 2));
    const double log2_k_minus_1 = k_minus_1 ? std::max(std::log2((double)k_minus_1), 0.0) : 0.0;
    return (u64)std::ceil(((double)security_level + log2_k_minus_1 + log2_u_choose_l) / ((double)log2_domain_size - log2_k_minus_1));
}
u64 num_ood_queries(unsigned security_level, unsigned log2_poly_degree, unsigned log2_expansion) {  // stir.rs:702-777
    return (u64)std::ceil(((double)security_level - 1.0 + 2.0 * log2_list_size(log2_expansion)) / (double)(LOG2_FIELD_SIZE - (int)log2_poly_degree));
}
// StirParameters::try_into_stir (stir.rs:420-560) with folding factor 4
bool try_into_stir(unsigned security_level, unsigned log2_expansion, unsigned log2_high_degree_bound, Stir* out) {
    if (log2_expansion == 0 || log2_high_degree_bound < (unsigned)LOG2_FOLDING_FACTOR) throw Error(TVM_ERR_INVALID_ARGUMENT, "LdtParameterError");
    const unsigned log2_len = log2_high_degree_bound + log2_expansion;
    if (log2_len > 32) throw Error(TVM_ERR_INVALID_ARGUMENT, "InitialDomainTooBig");
    Stir stir;
    stir.folding_factor = 1ull << LOG2_FOLDING_FACTOR;
    stir.initial_domain = ArithmeticDomain::of_length(1ull << log2_len).with_offset(generator());
    u64 folded_poly_degree = ((1ull << log2_high_degree_bound) - 1) / stir.folding_factor;
    unsigned log2_exp = log2_expansion, log2_folded_domain_size = log2_len - LOG2_FOLDING_FACTOR;
    auto ilog2 = [](u64 v) { unsigned n = 0; while (v >>= 1) n++; return n; };
    while (folded_poly_degree > stir.folding_factor) {
        const u64 in_domain = num_in_domain_queries(security_level, log2_folded_domain_size, log2_exp);
        const unsigned log2_next_exp = log2_exp + LOG2_FOLDING_FACTOR - LOG2_DOMAIN_SHRINKAGE;
        const u64 out_of_domain = num_ood_queries(security_level, ilog2(folded_poly_degree), log2_next_exp);
        const u64 next_degree = folded_poly_degree / stir.folding_factor;
        if (in_domain + out_of_domain > next_degree) break;
        stir.round_queries.push_back({in_domain, out_of_domain});
        folded_poly_degree = next_degree;
        log2_exp = log2_next_exp;
        log2_folded_domain_size -= LOG2_DOMAIN_SHRINKAGE;
    }
    stir.final_num_in_domain_queries = num_in_domain_queries(security_level, log2_folded_domain_size, log2_exp);
    stir.final_degree = folded_poly_degree;
    *out = stir;
    return true;
}
u64 randomized_trace_len_for(u64 padded_height, u64 h) {  // stark.rs:1885-1896
    const u64 total = std::max({padded_height + h, 2 * h + 1, (h + 1) * 5});
    u64 len = 1;
    while (len < total) len <<= 1;
    return len;
}
}  // namespace

Stir Stir::for_stark(u64 padded_height, unsigned security_level, unsigned log2_expansion) {
    unsigned log2_bound = 0;
    while ((1ull << log2_bound) < padded_height) log2_bound++;
    padded_height = 1ull << log2_bound;
    // the instance's query count fixes the number of trace randomizers, which fixes how long the domain must be: a
    // linear search over the bound at which a degree counts as high (stark.rs:2004-2031)
    for (int attempt = 0; attempt < 33; attempt++) {
        log2_bound++;
        Stir stir;
        try_into_stir(security_level, log2_expansion, log2_bound, &stir);
        if (stir.initial_domain.length >= randomized_trace_len_for(padded_height, stir.num_trace_randomizers()) << log2_expansion) return stir;
    }
    throw Error(TVM_ERR_INVALID_ARGUMENT, "no suitable STIR parameters found");
}

// Stir::prove (stir.rs:885-993).  Device work through the C ABI: stacked Merkle trees, polynomial folding, the witness
// polynomial of the next round, the answer polynomial (tvm_xfe_interpolate); host: sampling, inclusion proofs.
std::vector<u64> Stir::prove(const Context& c, const u64* d_codeword, ProofStream& ps) const {
    const u64 ff = folding_factor;
    struct Commitment {
        const u64* codeword;
        u64 length, n_leaves;
        DeviceBuffer nodes;
    };
    auto commit = [&](const u64* cw, u64 length) {
        Commitment t{cw, length, length / ff, DeviceBuffer(c, 10 * (length / ff))};
        c.check(tvm_stir_merkle_tree(c.raw(), cw, length, (uint32_t)ff, t.nodes.ptr()), "tvm_stir_merkle_tree");
        const std::vector<u64> root = merkle_root(c, t.nodes);
        ps.enqueue("stir root", root.data(), 5);
        return t;
    };
    auto respond = [&](const Commitment& t, const std::vector<u64>& folded_indices) {  // StirMerkleTree::inclusion_proof
        std::vector<u64> idx;
        for (u64 i : folded_indices)
            for (u64 j = 0; j < ff; j++) idx.push_back(i + j * t.n_leaves);
        std::vector<u64> leafs(idx.size() * 3);
        if (!idx.empty()) c.check(tvm_gather_elements(c.raw(), t.codeword, 3, idx.data(), idx.size(), leafs.data()), "stir leafs");
        ps.enqueue("stir response leafs", leafs.data(), leafs.size(), ff * 3);
        const std::vector<u64> auth = auth_nodes(c, t.nodes, t.n_leaves, folded_indices);
        ps.enqueue("stir response auth", auth.data(), auth.size());
    };
    auto unique_folded = [](const std::vector<u64>& indices, u64 folded_len) {  // .map(|i| i % len).unique()
        std::vector<u64> out;
        for (u64 i : indices) {
            const u64 f = i % folded_len;
            if (std::find(out.begin(), out.end(), f) == out.end()) out.push_back(f);
        }
        return out;
    };
    auto fold = [&](const u64* poly, u64 n_coeffs, const Xfe& randomness, u64* n_out) {
        *n_out = (n_coeffs + ff - 1) / ff;
        DeviceBuffer out(c, 3 * std::max<u64>(*n_out, 1));
        c.check(tvm_fold_polynomial(c.raw(), poly, n_coeffs, (uint32_t)ff, randomness.c, out.ptr()), "tvm_fold_polynomial");
        return out;
    };

    ArithmeticDomain domain = initial_domain;
    std::vector<DeviceBuffer> owned;  // codewords and polynomials of the rounds
    Commitment commitment = commit(d_codeword, domain.length);
    owned.push_back(domain.interpolate(c, d_codeword, 3));
    const u64* poly = owned.back().ptr();
    u64 n_coeffs = domain.length;
    std::vector<u64> first_round_indices;
    bool have_first = false;
    for (const auto& q : round_queries) {
        const Xfe folding_randomness = ps.sample_scalars(1)[0];
        u64 n_folded = 0;
        DeviceBuffer folded = fold(poly, n_coeffs, folding_randomness, &n_folded);
        ArithmeticDomain next_domain = domain.pow(1ull << LOG2_DOMAIN_SHRINKAGE);  // stir.rs:1149-1155
        next_domain = next_domain.with_offset(mont_mul(next_domain.offset, domain.offset));
        owned.push_back(next_domain.evaluate(c, folded.ptr(), n_folded, 3));
        const u64* folded_evaluations = owned.back().ptr();
        Commitment folded_commitment = commit(folded_evaluations, next_domain.length);

        const std::vector<Xfe> ood_queries = ps.sample_scalars(q.second);
        std::vector<Xfe> ood_values(q.second);
        if (q.second) c.check(tvm_evaluate_at_points(c.raw(), folded.ptr(), n_folded, ood_queries[0].c, (uint32_t)q.second, ood_values[0].c), "stir ood");
        ps.enqueue("stir ood values", q.second ? ood_values[0].c : nullptr, 3 * q.second);

        const std::vector<u64> queried = ps.sample_indices(domain.length, q.first);
        const ArithmeticDomain folded_domain = domain.pow(ff);
        const std::vector<u64> folded_queried = unique_folded(queried, folded_domain.length);
        respond(commitment, folded_queried);

        // the witness polynomial of the next round (stir.rs:945-966)
        const u64 k = folded_queried.size() + q.second;
        std::vector<Xfe> quotient_set(k), quotient_answers(k), answer_poly(k);
        {
            const DeviceBuffer on_folded_domain = folded_domain.evaluate(c, folded.ptr(), n_folded, 3);
            std::vector<u64> answers(folded_queried.size() * 3);
            c.check(tvm_gather_elements(c.raw(), on_folded_domain.ptr(), 3, folded_queried.data(), folded_queried.size(), answers.data()), "stir answers");
            for (size_t i = 0; i < folded_queried.size(); i++) {
                quotient_set[i] = Xfe{{folded_domain.value(folded_queried[i]), 0, 0}};
                std::memcpy(quotient_answers[i].c, &answers[3 * i], sizeof(Xfe));
            }
        }
        for (u64 i = 0; i < q.second; i++) quotient_set[folded_queried.size() + i] = ood_queries[i], quotient_answers[folded_queried.size() + i] = ood_values[i];
        if (tvm_xfe_interpolate(c.raw(), quotient_set[0].c, quotient_answers[0].c, (uint32_t)k, answer_poly[0].c))   // on the device (one workgroup)
            throw Error(TVM_ERR_INVALID_ARGUMENT, "STIR quotient set has repeated points");
        const Xfe degree_correction_randomness = ps.sample_scalars(1)[0];
        // any coset of >= n_folded points that avoids the quotient set: 7 generates F_p^*, so 7 * offset * <w> is disjoint from
        // offset * <w'> for every 2-power subgroup; the out-of-domain points are not in F_p
        u64 work_len = 1;
        while (work_len < n_folded) work_len <<= 1;
        const ArithmeticDomain work = ArithmeticDomain::of_length(work_len).with_offset(mont_mul(folded_domain.offset, generator()));
        DeviceBuffer next_poly(c, 3 * work_len);
        c.check(tvm_stir_next_polynomial(c.raw(), folded.ptr(), n_folded, quotient_set[0].c, answer_poly[0].c, (uint32_t)k,
                                         degree_correction_randomness.c, work.c(), next_poly.ptr()), "tvm_stir_next_polynomial");
        owned.push_back(std::move(next_poly));
        poly = owned.back().ptr();
        n_coeffs = n_folded;
        domain = next_domain;
        commitment = std::move(folded_commitment);
        if (!have_first) first_round_indices = queried, have_first = true;
    }
    // the final round has no quotienting (stir.rs:975-992)
    const Xfe folding_randomness = ps.sample_scalars(1)[0];
    u64 n_final = 0;
    const DeviceBuffer final_poly = fold(poly, n_coeffs, folding_randomness, &n_final);
    const std::vector<u64> final_words = final_poly.download(0, 3 * n_final);
    ps.enqueue("stir final polynomial", final_words.data(), final_words.size());
    const ArithmeticDomain folded_domain = domain.pow(ff);
    const std::vector<u64> queried = ps.sample_indices(domain.length, final_num_in_domain_queries);
    respond(commitment, unique_folded(queried, folded_domain.length));
    return have_first ? first_round_indices : queried;
}

// ------------------------------------------------------------------------------------------------ from an execution trace
namespace {
// TVMH_OPTION_TRACE: wall time of the steps of prove_execution on stderr (1: each step drains the stream first; 2: it does not --
// the host's own time per step, what a proof of a short trace is made of)
struct Stopwatch {
    const Context& c;
    const uint64_t mode = tvmh_get_option(TVMH_OPTION_TRACE);
    const bool on = mode != 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), start = t0;
    ~Stopwatch() {
        if (on) std::fprintf(stderr, "[tvmh] %-28s %8.2f ms\n", "total", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count());
    }
    void lap(const char* what) {
        if (!on) return;
        if (mode == 1) (void)tvm_sync(c.raw());
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[tvmh] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace
StarkParameters stark_parameters(unsigned log2_padded_height, unsigned security_level, unsigned log2_expansion, bool use_stir) {
    if (use_stir) {  // Stark::stir (stark.rs:1972-2032): the instance fixes the trace randomizers and the LDT domain
        const Stir stir = Stir::for_stark(1ull << log2_padded_height, security_level, log2_expansion);
        StarkParameters p(log2_padded_height, stir.num_trace_randomizers(), stir.num_first_round_queries(), log2_expansion);
        p.use_stir = true;
        p.stir = stir;
        p.ldt = stir.initial_domain;
        return p;
    }
    const u64 checks = (u64)std::ceil(-(double)security_level / std::log2(1.0 - proximity_parameter(log2_expansion)));
    return StarkParameters(log2_padded_height, checks + 4 * 3 * 2 + 1, checks, log2_expansion);
}

void offset_rng_seed(const uint8_t seed[32], u64 offset, uint8_t out[32]) {
    unsigned carry = 0;
    for (int k = 0; k < 32; k++) {
        const unsigned sum = (unsigned)seed[k] + (k < 8 ? (unsigned)((offset >> (8 * k)) & 0xFF) : 0u) + carry;
        out[k] = (uint8_t)(sum & 0xFF);
        carry = sum >> 8;
    }
}

// trace_randomizer_for_column for every column (master_table.rs:423-434) -> device [n_cols][h](x3)
std::vector<u64> trace_randomizers_host(const uint8_t table_seed[32], u64 n_cols, u64 h, int fk) {
    std::vector<u64> host(n_cols * h * fk);
    for (u64 col = 0; col < n_cols; col++) {
        uint8_t seed[32];
        offset_rng_seed(table_seed, col, seed);
        tvm_host_stdrng_elements(seed, h * fk, host.data() + col * h * fk);
    }
    return host;
}
DeviceBuffer upload(const Context& c, const std::vector<u64>& host) {
    DeviceBuffer d(c, host.size());
    c.check(tvm_memcpy_h2d(c.raw(), d.ptr(), host.data(), host.size() * sizeof(u64)), "tvm_memcpy_h2d");
    return d;
}

ExecutionTables::ExecutionTables(const Context& c, const StarkParameters& p, const tvm_aet& aet, const uint8_t seed[32],
                                 const std::function<void(const char*)>& lap) {
    const u64 n = p.trace.length;
    // the seeded randomness: offsets as in the table of master_table.rs:618-628.  The 470 trace-randomizer streams are drawn on the
    // device, one launch per table (tvm_stdrng_streams; round 6: they were 1.3 ms of sequential ChaCha on a helper thread, hidden
    // behind fill + pad at 2^20 rows and the largest idle gap of a proof of a short trace)
    uint8_t aux_seed[32], batch_seed[32], quotient_seed[32];
    offset_rng_seed(seed, NUM_MAIN, aux_seed);
    offset_rng_seed(aux_seed, NUM_AUX, batch_seed);
    offset_rng_seed(seed, NUM_MAIN + NUM_AUX + 1, quotient_seed);
    quotient_randomizer.resize(p.num_quotient_randomizers);
    if (!quotient_randomizer.empty()) tvm_host_stdrng_elements(quotient_seed, 3 * quotient_randomizer.size(), quotient_randomizer.data()->c);
    main_rnd = DeviceBuffer(c, NUM_MAIN * p.h);
    aux_rnd = DeviceBuffer(c, NUM_AUX * p.h * 3);
    c.check(tvm_stdrng_streams(c.raw(), seed, NUM_MAIN, p.h, main_rnd.ptr()), "tvm_stdrng_streams");
    c.check(tvm_stdrng_streams(c.raw(), aux_seed, NUM_AUX, 3 * p.h, aux_rnd.ptr()), "tvm_stdrng_streams");
    lap("trace randomizers");
    // MasterMainTable::new + pad (master_table.rs:881-983)
    main_trace = DeviceBuffer(c, NUM_MAIN * n);
    u64 lengths[9];
    c.check(tvm_fill_main_table(c.raw(), &aet, main_trace.ptr(), n, lengths), "tvm_fill_main_table");
    for (u64 len : lengths)
        if (len > p.padded_height) throw Error(TVM_ERR_INVALID_ARGUMENT, "a table is longer than the padded height");
    lap("fill from the AET");
    c.check(tvm_pad_main_table(c.raw(), main_trace.ptr(), n, lengths), "tvm_pad_main_table");
    c.check(tvm_fill_derived_main_columns(c.raw(), main_trace.ptr(), n), "tvm_fill_derived_main_columns");
    lap("pad + derived main columns");
    // MasterMainTable::extend (master_table.rs:1006-1075): the batch-randomizer column now, the rest once the challenges exist
    aux_trace = DeviceBuffer(c, NUM_AUX * n * 3);
    c.check(tvm_stdrng_elements(c.raw(), batch_seed, 3 * n, aux_trace.ptr() + (NUM_AUX - 1) * n * 3), "tvm_stdrng_elements");
    lap("batch randomizer column");
}

void ExecutionTables::extend(const Context& c, u64 n, const std::vector<Xfe>& challenges) const {
    c.check(tvm_extend_aux_table(c.raw(), main_trace.ptr(), aux_trace.ptr(), n, challenges[0].c), "tvm_extend_aux_table");
    c.check(tvm_fill_derived_aux_columns(c.raw(), main_trace.ptr(), aux_trace.ptr(), n, challenges[0].c), "tvm_fill_derived_aux_columns");
}

std::vector<u64> prove_execution(const Context& c, const StarkParameters& p, const tvm_aet& aet, const Claim& claim,
                                 const uint8_t seed[32]) {
    const u64 n = p.trace.length;
    Stopwatch watch{c};
    const ExecutionTables t(c, p, aet, seed, [&](const char* what) { watch.lap(what); });
    Prover prover(c, p, t.main_trace.ptr(), t.main_rnd.ptr(), t.aux_trace.ptr(), t.aux_rnd.ptr(), t.quotient_randomizer, claim);
    prover.assume_valid_trace = !tvmh_get_option(TVMH_OPTION_EXACT_AIR);
    prover.extend = [&](const std::vector<Xfe>& challenges) { t.extend(c, n, challenges); };
    const ProofStream stream = prover.prove();
    watch.lap("prove (extend + hot path)");
    std::vector<u64> proof = stream.proof();
    watch.lap("proof encoding");
    return proof;
}

}  // namespace triton_vm

static std::atomic<uint64_t> g_options[5] = {{0}, {0}, {0}, {0}, {0}};   // indexed by TVMH_OPTION_*
extern "C" void tvmh_set_option(uint32_t option, uint64_t value) {
    if (option >= 1 && option <= 4) g_options[option].store(value);
}
extern "C" uint64_t tvmh_get_option(uint32_t option) { return option >= 1 && option <= 4 ? g_options[option].load() : 0; }

extern "C" int32_t tvmh_prove(tvm_ctx* ctx, uint32_t log2_padded_height, uint64_t num_trace_randomizers,
                              uint64_t num_collinearity_checks, uint32_t log2_expansion, const uint64_t* d_main_trace,
                              const uint64_t* d_main_randomizers, const uint64_t* d_aux_trace,
                              const uint64_t* d_aux_randomizers, const uint64_t* h_quotient_randomizer,
                              const uint64_t* h_program_digest, const uint64_t* h_public_input, uint64_t n_public_input,
                              const uint64_t* h_public_output, uint64_t n_public_output, uint32_t use_stir, uint64_t* h_proof,
                              uint64_t capacity, uint64_t* proof_words, char* error, uint64_t error_capacity) {
    using namespace triton_vm;
    try {
        const Context c(ctx);
        const StarkParameters p = use_stir ? stark_parameters(log2_padded_height, 160, log2_expansion, true)
                                           : StarkParameters(log2_padded_height, num_trace_randomizers, num_collinearity_checks, log2_expansion);
        std::vector<Xfe> qr(p.num_quotient_randomizers);
        std::memcpy(qr.data(), h_quotient_randomizer, qr.size() * sizeof(Xfe));
        Claim claim;
        if (h_program_digest) std::memcpy(claim.program_digest, h_program_digest, sizeof(claim.program_digest));
        if (n_public_input) claim.input.assign(h_public_input, h_public_input + n_public_input);
        if (n_public_output) claim.output.assign(h_public_output, h_public_output + n_public_output);
        Prover prover(c, p, d_main_trace, d_main_randomizers, d_aux_trace, d_aux_randomizers, qr, claim);
        const std::vector<u64> proof = prover.prove().proof();
        if (proof_words) *proof_words = proof.size();
        if (h_proof && capacity >= proof.size()) std::memcpy(h_proof, proof.data(), proof.size() * sizeof(u64));
        return TVM_OK;
    } catch (const Error& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return e.status ? e.status : TVM_ERR_INVALID_ARGUMENT;
    } catch (const std::exception& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return TVM_ERR_DEVICE;
    }
}

extern "C" int32_t tvmh_prove_execution(tvm_ctx* ctx, const tvm_aet* aet, uint32_t log2_padded_height, uint32_t security_level,
                                        uint32_t log2_expansion, uint32_t use_stir, const uint8_t randomness_seed[32],
                                        const uint64_t* h_program_digest, const uint64_t* h_public_input,
                                        uint64_t n_public_input, const uint64_t* h_public_output, uint64_t n_public_output,
                                        uint64_t* h_proof, uint64_t capacity, uint64_t* proof_words, char* error,
                                        uint64_t error_capacity) {
    using namespace triton_vm;
    try {
        if (!aet || !randomness_seed) throw Error(TVM_ERR_INVALID_ARGUMENT, "tvmh_prove_execution: null execution trace or seed");
        const Context c(ctx);
        // use_stir: 0 = LdtChoice::Fri, 1 = LdtChoice::Stir, 2 = Stark::ldt's rule (STIR from 2^16 padded rows on, stark.rs:1944-1951)
        if (use_stir > 2) throw Error(TVM_ERR_INVALID_ARGUMENT, "tvmh_prove_execution: use_stir is 0 (FRI), 1 (STIR) or 2 (automatic)");
        const bool stir = use_stir == 2 ? log2_padded_height >= 16 : use_stir == 1;
        const StarkParameters p = stark_parameters(log2_padded_height, security_level, log2_expansion, stir);
        Claim claim;
        if (h_program_digest) std::memcpy(claim.program_digest, h_program_digest, sizeof(claim.program_digest));
        if (n_public_input) claim.input.assign(h_public_input, h_public_input + n_public_input);
        if (n_public_output) claim.output.assign(h_public_output, h_public_output + n_public_output);
        // The reference's memory policy (master_table.rs:268-271, stark.rs:730-768): the cached extension first; when the device (or
        // the context's memory limit) cannot hold it, the same proof coset by coset with as few passes as fit (sharded_host.cpp).
        std::vector<u64> proof;
        bool out_of_memory = false;
        try {
            proof = prove_execution(c, p, *aet, claim, randomness_seed);
        } catch (const Error& e) {
            if (e.status != TVM_ERR_OUT_OF_MEMORY || p.quotient.length != p.ldt.length) throw;
            out_of_memory = true;   // (outside the handler: the failed attempt's buffers are back in the pool by now)
        }
        if (out_of_memory) {
            (void)tvm_ctx_trim(c.raw());
            // the policy of the sharded entry from two passes on (sharded_host.cpp: pass counts by estimate, retries on out-of-memory)
            proof = prove_execution_sharded(c, p, nullptr, 0, *aet, claim, randomness_seed, false, nullptr, 1ull << 21, 2);
        }
        if (proof_words) *proof_words = proof.size();
        if (h_proof && capacity >= proof.size()) std::memcpy(h_proof, proof.data(), proof.size() * sizeof(u64));
        return TVM_OK;
    } catch (const Error& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return e.status ? e.status : TVM_ERR_INVALID_ARGUMENT;
    } catch (const std::exception& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return TVM_ERR_DEVICE;
    }
}

// Stir::prove alone, for an explicitly given instance (the tests' small instances; Stark::stir derives them otherwise):
// round_queries = [in-domain, out-of-domain] pairs.  Returns the proof of a stream that holds only the STIR items.
extern "C" int32_t tvmh_stir_prove(tvm_ctx* ctx, tvm_domain initial_domain, uint32_t folding_factor, const uint64_t* round_queries,
                                   uint32_t n_rounds, uint64_t final_num_in_domain_queries, uint64_t final_degree,
                                   const uint64_t* d_codeword, uint64_t* h_first_round_indices, uint64_t* h_proof, uint64_t capacity,
                                   uint64_t* proof_words, char* error, uint64_t error_capacity) {
    using namespace triton_vm;
    try {
        const Context c(ctx);
        Stir stir;
        stir.initial_domain = ArithmeticDomain{initial_domain.offset, initial_domain.generator, initial_domain.length};
        stir.folding_factor = folding_factor;
        for (uint32_t r = 0; r < n_rounds; r++) stir.round_queries.push_back({round_queries[2 * r], round_queries[2 * r + 1]});
        stir.final_num_in_domain_queries = final_num_in_domain_queries;
        stir.final_degree = final_degree;
        ProofStream ps;
        const std::vector<u64> first = stir.prove(c, d_codeword, ps);
        if (h_first_round_indices) std::memcpy(h_first_round_indices, first.data(), first.size() * sizeof(u64));
        const std::vector<u64> proof = ps.proof();
        if (proof_words) *proof_words = proof.size();
        if (h_proof && capacity >= proof.size()) std::memcpy(h_proof, proof.data(), proof.size() * sizeof(u64));
        return TVM_OK;
    } catch (const Error& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return e.status ? e.status : TVM_ERR_INVALID_ARGUMENT;
    } catch (const std::exception& e) {
        if (error && error_capacity) std::snprintf(error, error_capacity, "%s", e.what());
        return TVM_ERR_DEVICE;
    }
}

// Stark::stir's instance for a padded height: out = [initial domain length, folding factor, final in-domain queries, final
// degree, number of full rounds, then (in-domain, out-of-domain) per round]; returns the number of words, 0 on error
extern "C" uint64_t tvmh_stir_parameters(uint64_t padded_height, uint32_t security_level, uint32_t log2_expansion, uint64_t* out,
                                         uint64_t capacity) {
    try {
        const triton_vm::Stir stir = triton_vm::Stir::for_stark(padded_height, security_level, log2_expansion);
        std::vector<uint64_t> w = {stir.initial_domain.length, stir.folding_factor, stir.final_num_in_domain_queries, stir.final_degree,
                                   stir.round_queries.size()};
        for (const auto& q : stir.round_queries) w.push_back(q.first), w.push_back(q.second);
        if (w.size() > capacity) return 0;
        std::memcpy(out, w.data(), w.size() * sizeof(uint64_t));
        return w.size();
    } catch (...) {
        return 0;
    }
}

