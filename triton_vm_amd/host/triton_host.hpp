// triton_host.hpp -- C++ host side above the C ABI of libtriton_hip.so: the reference's interface for the hot
// path of Prover::prove, same names, argument meaning and step order (the reference is Rust; its toolchain is
// not in the build image, so the host mirror is C++ -- INTEGRATION.md has the Rust binding a maintainer would add).
//
//   ArithmeticDomain   /root/reference/triton-vm/src/arithmetic_domain.rs:34-92, 227-229, 280-296
//   MasterTable        /root/reference/triton-vm/src/table/master_table.rs:190-610 (the methods on the hot path)
//   ProofStream, Claim /root/reference/triton-vm/src/proof_stream.rs:8-119, proof_item.rs:96-150, proof.rs:62-84: the
//                      reference's transcript -- BFieldCodec encoding of the items, Fiat-Shamir sampling -- so that
//                      `ProofStream::proof()` is the reference's Proof for the same tables, randomizers and claim
//   Stark / Prover     /root/reference/triton-vm/src/stark.rs:263-286 (domains), 331-719 (prove), fri.rs:212-366
//
// Everything bulky stays in HBM behind the C ABI; this file only sequences calls and keeps the transcript.
// No device code, no HIP headers: it compiles with g++ and binds to whichever library exports the tvm_* symbols.
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "triton_hip.h"

extern "C" {
typedef struct tvmh_comm tvmh_comm;   // the collectives of the sharded prover as a table of functions: defined below
}

namespace triton_vm {

typedef uint64_t u64;
constexpr u64 P = 0xFFFFFFFF00000001ull;
struct Xfe {
    u64 c[3];
};

// ---- scalar helpers over F_p on Montgomery words (twenty-first's BFieldElement; host plumbing only) ---------
u64 to_mont(u64 v);
u64 mont_mul(u64 a, u64 b);
u64 mont_pow(u64 a, u64 e);
u64 generator();                          // BFieldElement::generator() = 7
u64 primitive_root_of_unity(u64 order);   // 7^((p-1)/order)

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

class Context {  // one per proving thread (lib.rs:522-532)
public:
    explicit Context(tvm_ctx* borrowed) : ctx_(borrowed) {}
    tvm_ctx* raw() const { return ctx_; }
    void check(int32_t status, const char* what) const;
    u64* alloc(u64 n_words) const;
    void free(u64* p) const { (void)tvm_free(ctx_, p); }

private:
    tvm_ctx* ctx_;
};

class DeviceBuffer {  // owning device array of 64-bit words
public:
    DeviceBuffer() = default;
    DeviceBuffer(const Context& c, u64 n_words) : c_(&c), p_(c.alloc(n_words)), n_(n_words) {}
    DeviceBuffer(DeviceBuffer&& o) noexcept : c_(o.c_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept;
    DeviceBuffer(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { reset(); }
    void reset();
    u64* ptr() const { return p_; }
    u64 words() const { return n_; }
    std::vector<u64> download(u64 first_word, u64 n_words) const;

private:
    const Context* c_ = nullptr;
    u64* p_ = nullptr;
    u64 n_ = 0;
};

struct ArithmeticDomain {
    u64 offset, generator, length;
    static ArithmeticDomain of_length(u64 length);                       // arithmetic_domain.rs:78-85
    ArithmeticDomain with_offset(u64 o) const { return {o, generator, length}; }  // :89-92
    ArithmeticDomain pow(u64 exponent) const;                            // :280-296
    u64 value(u64 n) const { return mont_mul(mont_pow(generator, n), offset); }   // :227-229
    tvm_domain c() const { return tvm_domain{offset, generator, length}; }
    DeviceBuffer evaluate(const Context& c, const u64* d_coeffs, u64 n_coeffs, int field_kind) const;  // :141-170
    DeviceBuffer interpolate(const Context& c, const u64* d_values, int field_kind) const;             // :182-189
};

// proof.rs:62-84 (Montgomery words)
struct Claim {
    u64 program_digest[5] = {0, 0, 0, 0, 0};
    uint32_t version = 6;  // proof.rs:33 CURRENT_VERSION
    std::vector<u64> input, output;
    std::vector<u64> encode() const;
};

// proof_stream.rs:8-104.  Items are enqueued under a label that names the ProofItem variant (see LABELS in the .cpp);
// a FRI response is enqueued as its two parts, leaves then authentication structure.
class ProofStream {
public:
    struct Item {
        std::string name;
        std::vector<u64> words;
        bool fiat_shamir;
        u64 stack_words = 0;  // a StirResponse's leaves: words per stacked leaf (Vec<Vec<XFieldElement>>), else 0
    };
    void alter_fiat_shamir_state_with(const std::vector<u64>& encoding);   // proof_stream.rs:40-42
    void enqueue(const std::string& name, const u64* words, u64 n, u64 stack_words = 0);   // proof_stream.rs:54-59
    std::vector<Xfe> sample_scalars(u64 n);
    std::vector<u64> sample_indices(u64 upper_bound, u64 n);
    const std::vector<Item>& items() const { return items_; }
    const u64* sponge_state() const { return state_; }                       // the 16 words of the Fiat-Shamir sponge
    std::vector<u64> proof() const;                                         // proof_stream.rs:115-119: Proof(encode())

private:
    void squeeze(u64 out[10]);
    u64 state_[16] = {0};  // Tip5::init(): the variable-length domain
    std::vector<Item> items_;
};

// A padded master main (field_kind 1) or auxiliary (field_kind 3) table on the device.  The trace
// [n_cols][n_rows](x3) and the trace-randomizer coefficients [n_cols][h](x3) are device arrays owned by the caller.
class MasterTable {
public:
    MasterTable(const Context& c, int field_kind, const u64* d_trace, u64 n_rows, u64 n_cols, const u64* d_randomizers,
                u64 num_trace_randomizers, ArithmeticDomain trace, ArithmeticDomain quotient, ArithmeticDomain ldt);
    ~MasterTable() { clear_cache(); }
    ArithmeticDomain evaluation_domain() const;                           // master_table.rs:215-222
    void maybe_low_degree_extend_all_columns();                           // :258-322
    void clear_cache();
    const tvm_table* table() const;
    DeviceBuffer merkle_tree() const;                                     // :443-468 -> node array [2L][5]
    std::vector<u64> reveal_rows(const std::vector<u64>& row_indices) const;          // :548-555
    std::vector<u64> out_of_domain_rows(const std::vector<Xfe>& points) const;        // :348-390, [n_points][n_cols][3]
    DeviceBuffer weighted_sum_of_columns(const Xfe* weights) const;       // :512-542 -> 2 * n_rows XFE coefficients
    int field_kind() const { return fk_; }
    u64 n_cols() const { return n_cols_; }
    // the quotient / LDT domain this table is extended onto: a coset group of the real domains when the extended rows are
    // split over ranks or evaluated pass by pass (sharded_host.cpp); drops the cached extension
    void set_domains(ArithmeticDomain quotient, ArithmeticDomain ldt);
    // maybe_low_degree_extend_all_columns the way the COLUMN sharding does it (include/triton_hip.h:
    // tvm_lde_column_coefficients ...): this rank interpolates ITS block of virtual columns, the coefficient forms are all-gathered
    // in `chunks` chunks -- all requested at once through all_gather_async where the communicator has it, so that the exchange
    // of chunk c + 1 runs under the extension of chunk c -- and every column is extended onto this table's own evaluation domain.
    // The same table as maybe_low_degree_extend_all_columns; sent(bytes) is told what each exchange sends.
    void low_degree_extend_over(const tvmh_comm* comm, unsigned chunks, const std::function<void(u64)>& sent);
    // a second table object over the same trace and randomizer arrays, with no cached extension of its own
    MasterTable sibling() const { return MasterTable(c_, fk_, d_trace_, n_rows_, n_cols_, d_rnd_, h_, trace_, quotient_, ldt_); }
    // reveal_rows over the view of `view_rows` rows of the cached extension (its own LDT domain by default)
    std::vector<u64> reveal_rows_of(const std::vector<u64>& row_indices, u64 view_rows) const;
    // out_of_domain_rows for the n columns from first_col on (a rank's share when the columns are split): [n_points][n][3]
    std::vector<u64> out_of_domain_rows(const std::vector<Xfe>& points, u64 first_col, u64 n) const;

private:
    const Context& c_;
    int fk_;
    const u64 *d_trace_, *d_rnd_;
    u64 n_rows_, n_cols_, h_;
    ArithmeticDomain trace_, quotient_, ldt_;
    tvm_table* table_ = nullptr;
};

// The STIR low-degree test (low_degree_test/stir.rs): parameter derivation (stir.rs:403-560, 597-793; the f64 formulas are
// the reference's) and the prover (stir.rs:885-993) over the C ABI.  The reference's automatic choice from 2^16 padded rows.
struct Stir {
    ArithmeticDomain initial_domain;
    u64 folding_factor = 4;
    std::vector<std::pair<u64, u64>> round_queries;  // (in-domain, out-of-domain) per full round
    u64 final_num_in_domain_queries = 0, final_degree = 0;
    u64 num_first_round_queries() const { return round_queries.empty() ? final_num_in_domain_queries : round_queries[0].first; }
    u64 num_trace_randomizers() const { return num_first_round_queries() + 4 * 3 * 2 + 1; }  // stark.rs:2083-2089
    // Stark::stir (stark.rs:1972-2032): the smallest instance whose initial domain holds the randomized trace
    static Stir for_stark(u64 padded_height, unsigned security_level, unsigned log2_expansion);
    // Stir::prove: enqueues roots, out-of-domain values, responses and the final polynomial; -> first-round indices
    std::vector<u64> prove(const Context& c, const u64* d_codeword, ProofStream& ps) const;
};

// Domains for a padded height as Stark::default() with LdtChoice::Fri derives them
// (stark.rs:263-286, 1885-1916, 2083-2089; fri.rs:832-836, 907-920).
struct StarkParameters {
    StarkParameters(unsigned log2_padded_height, u64 num_trace_randomizers = 198, u64 num_collinearity_checks = 173,
                    unsigned log2_expansion = 2);
    u64 padded_height, h, num_collinearity_checks, randomized_trace_len, num_quotient_randomizers;
    unsigned fri_rounds, log2_expansion;
    ArithmeticDomain trace, quotient, ldt;
    bool use_stir = false;  // LdtChoice::Stir: `stir` fixes h and the LDT domain (stark_parameters)
    Stir stir;
};

class Prover {
public:
    Prover(const Context& c, const StarkParameters& p, const u64* d_main_trace, const u64* d_main_randomizers,
           const u64* d_aux_trace, const u64* d_aux_randomizers, const std::vector<Xfe>& quotient_randomizer,
           const Claim& claim = Claim());
    ProofStream prove();  // the hot path of Prover::prove, stark.rs:331-719
    std::vector<Xfe> last_polynomial;
    // called with the 63 challenges before the auxiliary table is extended: MasterMainTable::extend for a prover that
    // starts from an execution trace (prove_execution); the auxiliary trace buffer is filled then
    std::function<void(const std::vector<Xfe>&)> extend;
    bool assume_valid_trace = false;  // TVM_OPTION_AIR_VALID_TRACE around the quotient evaluation

private:
    std::vector<u64> fri(const DeviceBuffer& combination, ProofStream& ps);  // Fri::prove, fri.rs:212-319
    const Context& c_;
    StarkParameters p_;
    Claim claim_;
    MasterTable main_, aux_;
    std::vector<Xfe> quotient_randomizer_;
};

// Stark::new(security_level, log2_expansion) with LdtChoice::Fri: the number of collinearity checks (fri.rs:832-836,
// low_degree_test/mod.rs:93-170, proven soundness) and of trace randomizers (stark.rs:2083-2089)
StarkParameters stark_parameters(unsigned log2_padded_height, unsigned security_level, unsigned log2_expansion,
                                 bool use_stir = false);

// offset_rng_seed (master_table.rs:630-662)
void offset_rng_seed(const uint8_t seed[32], u64 offset, uint8_t out[32]);

// What Prover::prove(claim, aet) builds before the hot path (stark.rs:331-400): MasterMainTable::new + pad on the device
// (master_table.rs:881-983), every randomizer drawn from `seed` the way the reference draws it (master_table.rs:423-434,
// 1006-1024, stark.rs:1315-1322), the auxiliary trace buffer with its batch-randomizer column; `extend` runs
// MasterMainTable::extend once the challenges exist (master_table.rs:1006-1075).  lap(name) is called after each step.
struct ExecutionTables {
    DeviceBuffer main_trace, main_rnd, aux_trace, aux_rnd;
    std::vector<Xfe> quotient_randomizer;
    ExecutionTables(const Context& c, const StarkParameters& p, const tvm_aet& aet, const uint8_t seed[32],
                    const std::function<void(const char*)>& lap);
    void extend(const Context& c, u64 n_rows, const std::vector<Xfe>& challenges) const;
};

// Prover::prove(claim, aet) from its first line (stark.rs:331-719): the master main table is filled from the algebraic
// execution trace and padded on the device, every randomizer is drawn from `seed` the way the reference draws it
// (master_table.rs:423-434, 1006-1024, stark.rs:1315-1322), the auxiliary table is extended on the device once the
// challenges are sampled, the AIR runs in valid-trace mode.  Returns the proof (the words of the reference's Proof).
std::vector<u64> prove_execution(const Context& c, const StarkParameters& p, const tvm_aet& aet, const Claim& claim,
                                 const uint8_t seed[32]);

}  // namespace triton_vm

// ---- one proof over the GPUs of a node, and/or coset by coset (sharded_host.cpp) ----------------------------------------
// The collectives the sharded prover needs, as a table of functions so that this library stays free of device code and of
// any communication library: rccl_comm.cpp (libtriton_rccl.so) fills it with RCCL calls on the context's stream, the local
// implementation below with copies between the contexts of one process (tests, the single-GPU lockstep measurement of
// bench.py --simulate-gpus), the CPU tests with callbacks that go through torch.distributed's gloo backend.  Buffers are
// DEVICE buffers of 64-bit words; a call is ordered after the work already queued on the context's stream and its result
// is visible to work queued afterwards (it need not have completed when the call returns).
extern "C" {
typedef struct tvmh_comm {
    void* self;
    uint32_t rank, world;
    /* d_recv[r * words_per_rank ...] = rank r's d_send[0 .. words_per_rank) */
    int32_t (*all_gather)(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words_per_rank);
    /* d_recv[r * words_per_pair ...] = rank r's d_send[me * words_per_pair ...] */
    int32_t (*all_to_all)(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words_per_pair);
    /* optional hooks (may be null): prove begins / a named stage begins / prove ends on this rank */
    void (*begin)(void* self, tvm_ctx* ctx);
    void (*mark)(void* self, tvm_ctx* ctx, const char* stage);
    void (*end)(void* self, tvm_ctx* ctx);
    /* optional (may be null): this rank failed outside a collective and will not take part in the ones its peers are waiting in.
     * RCCL: ncclCommAbort -- the rank's own queued collectives are cancelled and its process can report the error (the launcher
     * tears the job down); in-process communicators: every rank waiting in a collective leaves it with TVM_ERR_DEVICE. */
    void (*abort)(void* self);
    /* optional, ONLY for communicators whose ranks live in one process (may be null; null in rccl_comm.cpp): a barrier at which
     * the ranks exchange one pointer to an object in their common address space, so that READ-ONLY replicated device data --
     * the trace-side tables every rank would otherwise build and hold for itself -- exist once.  op TVMH_SHARE_RELEASE: barrier,
     * then rank 0 drops the object the group keeps (every rank has left the proof that used it).  TVMH_SHARE_PUBLISH: rank 0
     * hands over `mine` (its device work complete) and `drop`; the group keeps it until the next release or its own destruction,
     * which must happen while rank 0's context is alive; every rank's *out is rank 0's object -- and rank 0's *out == mine says "the
     * group has taken ownership" EVEN IF the call then returns an error (the publisher must not free it).  TVMH_SHARE_BARRIER: barrier only
     * (*out = the kept object).  Used by bench.py --simulate-gpus (eight ranks of a 2^22-row proof on one GPU: 21.8 GB of traces
     * once instead of eight times) under TVMH_OPTION_SHARE_REPLICATED_TABLES; a multi-process communicator never sees it. */
    int32_t (*share)(void* self, tvm_ctx* ctx, uint32_t op, const void* mine, void (*drop)(const void*), const void** out);
    /* optional (may be null: the caller then uses all_gather): an all-gather that does NOT occupy the context's stream -- ordered
     * after the work queued on it so far, carried out on the communicator's own stream, and complete for the context's stream only
     * once wait(slot) has been called -- so that several exchanges can be in flight under the kernels that consume the earlier
     * ones (the coefficient exchange of the column split, chunk by chunk: MasterTable::low_degree_extend_over).  slot < 16. */
    int32_t (*all_gather_async)(void* self, tvm_ctx* ctx, const uint64_t* d_send, uint64_t* d_recv, uint64_t words_per_rank, uint32_t slot);
    int32_t (*wait)(void* self, tvm_ctx* ctx, uint32_t slot);
} tvmh_comm;
#define TVMH_SHARE_BARRIER 0
#define TVMH_SHARE_PUBLISH 1
#define TVMH_SHARE_RELEASE 2
}

namespace triton_vm {

// rows i = g (mod G) of `domain`, as a domain (offset * generator^g, generator^G, length / G)
ArithmeticDomain coset_group(const ArithmeticDomain& domain, u64 g, u64 G);

// Prover::prove with the extended master tables split by cosets of the trace domain: over the ranks of `comm` (rank r of R
// owns the extended rows i = r (mod R): SURVEY 8(e), stark.rs:805-1006 and master_table.rs:1305-1306 are the reference's
// formulation of the same decomposition) and/or over `passes` passes per rank that never hold more than 1/passes of a
// rank's share (the reference's just-in-time path, master_table.rs:268-271, 470-503, 556-609).  Same constructor data as
// Prover; every rank passes the same traces, randomizers and claim and obtains the same transcript.  What is split: the
// table extensions, row hashing, the Merkle trees (by leaf ranges, digests by all-to-all), the AIR, the quotient-segment
// table, the out-of-domain rows (by columns), the linear combinations and DEEP (row-local), the first FRI rounds (i and
// i + n/2 share a residue mod R: the fold is rank-local; codewords by all-to-all for the split trees).  What is replicated:
// the trace-side work, the inverse transforms inside the table extension, the interpolation of the quotient codeword,
// the weighted column sums, the small FRI rounds.
class ShardedProver {
public:
    ShardedProver(const Context& c, const StarkParameters& p, const tvmh_comm* comm, unsigned passes, const u64* d_main_trace,
                  const u64* d_main_randomizers, const u64* d_aux_trace, const u64* d_aux_randomizers,
                  const std::vector<Xfe>& quotient_randomizer, const Claim& claim = Claim());
    ProofStream prove();
    std::function<void(const std::vector<Xfe>&)> extend;  // as Prover::extend
    bool assume_valid_trace = false;
    u64 split_tree_min_leaves = 1ull << 21;  // trees with fewer leaves are built whole on every rank
    bool profile = false;                    // drain the stream at every stage boundary and record stage times
    // what the last prove() did: stage -> ms (profile only), collective -> {calls, bytes sent by this rank}
    std::vector<std::pair<std::string, double>> stage_ms;
    std::vector<std::pair<std::string, std::pair<u64, u64>>> exchanges;
    u64 split_trees_built = 0;
    std::string stats_json() const;

private:
    friend struct ShardedRun;  // one prove(): sharded_host.cpp
    const Context& c_;
    StarkParameters p_;
    const tvmh_comm* comm_;
    unsigned passes_;
    Claim claim_;
    MasterTable main_, aux_;
    std::vector<Xfe> quotient_randomizer_;
};

// Prover::prove(claim, aet) over the ranks of `comm` (null: this process alone) with `passes` passes per rank; passes == 0
// is the reference's memory policy (master_table.rs:268-271, stark.rs:730-768): the cached path, and if the device (or the
// context's memory limit) cannot hold it, the coset-wise path with as few passes as fit WITH headroom (a pass count is attempted when its
// estimated footprint fits what the context can obtain, tvm_ctx_memory_info; an out-of-memory starts over with more).  stats: optional,
// see ShardedProver.
std::vector<u64> prove_execution_sharded(const Context& c, const StarkParameters& p, const tvmh_comm* comm, unsigned passes,
                                         const tvm_aet& aet, const Claim& claim, const uint8_t seed[32], bool profile = false,
                                         std::string* stats = nullptr, u64 split_tree_min_leaves = 1ull << 21,
                                         unsigned policy_first_passes = 1);   // passes == 0 on one rank: the first pass count the policy considers

}  // namespace triton_vm

// tvmh_prove_execution over a communicator and / or coset by coset (see triton_vm::prove_execution_sharded): comm may be null,
// jit_passes == 0 selects the memory policy.  stats_json (optional): a JSON object with this rank's stage times (when
// profile != 0) and its exchanges.
extern "C" int32_t tvmh_prove_execution_sharded(tvm_ctx* ctx, const tvmh_comm* comm, uint32_t jit_passes, uint64_t split_tree_min_leaves,
                                                const tvm_aet* aet, uint32_t log2_padded_height, uint32_t security_level,
                                                uint32_t log2_expansion, uint32_t use_stir, const uint8_t randomness_seed[32],
                                                const uint64_t* h_program_digest, const uint64_t* h_public_input,
                                                uint64_t n_public_input, const uint64_t* h_public_output, uint64_t n_public_output,
                                                uint64_t* h_proof, uint64_t proof_capacity_words, uint64_t* proof_words,
                                                uint32_t profile, char* stats_json, uint64_t stats_capacity, char* error,
                                                uint64_t error_capacity);
// the hot path alone on device-resident traces (as tvmh_prove), sharded / coset-wise
extern "C" int32_t tvmh_prove_sharded(tvm_ctx* ctx, const tvmh_comm* comm, uint32_t jit_passes, uint64_t split_tree_min_leaves,
                                      uint32_t log2_padded_height, uint64_t num_trace_randomizers, uint64_t num_collinearity_checks,
                                      uint32_t log2_expansion, const uint64_t* d_main_trace, const uint64_t* d_main_randomizers,
                                      const uint64_t* d_aux_trace, const uint64_t* d_aux_randomizers,
                                      const uint64_t* h_quotient_randomizer, uint32_t use_stir, uint32_t stir_security_level,
                                      uint64_t* h_proof, uint64_t proof_capacity_words, uint64_t* proof_words, char* error,
                                      uint64_t error_capacity);

// Communicators between the contexts of ONE process (one thread per rank): collectives are a rendezvous plus device-to-device
// copies on each rank's own stream.  lockstep != 0: between two collectives only one rank computes at a time, in rank
// order, and the time each rank spends per stage (its stream drained at every stage mark) is recorded -- on a single GPU
// that measures, rank by rank and without contention, the work the ranks of a real N-GPU run would do concurrently
// (bench.py --simulate-gpus).  out: `world` pointers; all of them are freed by destroying comms[0].
extern "C" int32_t tvmh_local_comms_create(uint32_t world, uint32_t lockstep, tvmh_comm** out);
extern "C" void tvmh_local_comms_destroy(tvmh_comm* first);
// a rank failed outside a collective (its caller reports the error): every rank waiting in one leaves it with TVM_ERR_DEVICE
extern "C" void tvmh_local_comms_abort(tvmh_comm* any);
// lockstep accounting of a local communicator: JSON {"stage": [ms of rank 0, rank 1, ...], ...}, stages in first-seen order
extern "C" uint64_t tvmh_local_comms_report(const tvmh_comm* any, char* json, uint64_t capacity);

// C entry for hosts without a C++ ABI (the Python tests and bench.py): runs Prover::prove on device-resident traces
// and returns the proof (the words of the reference's `Proof`).  h_program_digest (5 words) may be null (all zero), the
// public input / output may be empty.  use_stir != 0: LdtChoice::Stir at security level 160 (the two query-count arguments are
// ignored: the STIR instance fixes them).  *proof_words is the length of the proof; it is copied when capacity suffices.
extern "C" int32_t tvmh_prove(tvm_ctx* ctx, uint32_t log2_padded_height, uint64_t num_trace_randomizers,
                              uint64_t num_collinearity_checks, uint32_t log2_expansion, const uint64_t* d_main_trace,
                              const uint64_t* d_main_randomizers, const uint64_t* d_aux_trace,
                              const uint64_t* d_aux_randomizers, const uint64_t* h_quotient_randomizer,
                              const uint64_t* h_program_digest, const uint64_t* h_public_input, uint64_t n_public_input,
                              const uint64_t* h_public_output, uint64_t n_public_output, uint32_t use_stir, uint64_t* h_proof,
                              uint64_t proof_capacity_words, uint64_t* proof_words, char* error, uint64_t error_capacity);

// Prover::prove(claim, aet) -- triton_vm::prove_execution -- for hosts without a C++ ABI.  use_stir: 0 = LdtChoice::Fri,
// 1 = LdtChoice::Stir, 2 = Stark::ldt's rule, what Stark::default() does (STIR from 2^16 padded rows on, stark.rs:1944-1951).
// The arrays of `aet` may be host or device memory (include/triton_hip.h: tvm_fill_main_table).
extern "C" int32_t tvmh_prove_execution(tvm_ctx* ctx, const tvm_aet* aet, uint32_t log2_padded_height, uint32_t security_level,
                                        uint32_t log2_expansion, uint32_t use_stir, const uint8_t randomness_seed[32],
                                        const uint64_t* h_program_digest, const uint64_t* h_public_input,
                                        uint64_t n_public_input, const uint64_t* h_public_output, uint64_t n_public_output,
                                        uint64_t* h_proof, uint64_t proof_capacity_words, uint64_t* proof_words, char* error,
                                        uint64_t error_capacity);

// Process-wide switches of this host library.  TVMH_OPTION_EXACT_AIR != 0: prove_execution (plain and sharded) evaluates the AIR row
// by row on every point of the quotient domain, as the reference does (master_table.rs:1264-1363), instead of in valid-trace mode
//  -- the same proof on a valid trace; bench.py times both.
#define TVMH_OPTION_EXACT_AIR 1
// TVMH_OPTION_SHARE_REPLICATED_TABLES != 0: the ranks of an IN-PROCESS communicator (one that has a `share` hook) use ONE copy of the
// replicated trace-side tables -- rank 0 fills, pads and extends, the other ranks read its device arrays -- instead of one copy each.
// For the single-GPU lockstep measurement of the N-rank code path at heights where N copies do not fit (bench.py --simulate-gpus
// at 2^22 rows); the stage times of ranks 1..N-1 then lack the replicated stages, rank 0's has them.  Default 0.
#define TVMH_OPTION_SHARE_REPLICATED_TABLES 2
// TVMH_OPTION_TRACE != 0: host wall time of the steps of prove_execution (plain and sharded) on stderr.  Diagnostics only; the host
// library, like the backend, reads no environment variable.
#define TVMH_OPTION_TRACE 3
// TVMH_OPTION_COLUMN_SPLIT = k > 0: the sharded prover splits the INVERSE transforms of the table extensions by columns over the
// ranks and exchanges the coefficients in k chunks per table (MasterTable::low_degree_extend_over) instead of replicating them --
// north_star's column sharding for the part of the front end where it applies.  Pays only for the part of the exchange that
// overlaps with compute (at ~300 GB/s per rank moving a column's coefficients costs what recomputing them
// costs); default 0 = the coset sharding alone.  k <= 16.
#define TVMH_OPTION_COLUMN_SPLIT 4
extern "C" void tvmh_set_option(uint32_t option, uint64_t value);
extern "C" uint64_t tvmh_get_option(uint32_t option);

// Stir::prove alone for an explicitly given instance (round_queries: [in-domain, out-of-domain] pairs); the proof of a
// stream holding only the STIR items; h_first_round_indices: room for the first round's in-domain query count.
extern "C" int32_t tvmh_stir_prove(tvm_ctx* ctx, tvm_domain initial_domain, uint32_t folding_factor, const uint64_t* round_queries,
                                   uint32_t n_rounds, uint64_t final_num_in_domain_queries, uint64_t final_degree,
                                   const uint64_t* d_codeword, uint64_t* h_first_round_indices, uint64_t* h_proof,
                                   uint64_t proof_capacity_words, uint64_t* proof_words, char* error, uint64_t error_capacity);

// Stark::stir's instance for a padded height: out = [initial domain length, folding factor, final in-domain queries, final
// degree, number of full rounds, then (in-domain, out-of-domain) per round]; returns the number of words, 0 on error.
extern "C" uint64_t tvmh_stir_parameters(uint64_t padded_height, uint32_t security_level, uint32_t log2_expansion, uint64_t* out,
                                         uint64_t capacity);

