/*
 * triton_hip.h -- C ABI of libtriton_hip.so, the MI355X (gfx950) backend for the hot path of
 * triton_vm::stark::Prover::prove (/root/reference/triton-vm/src/stark.rs:331-719).
 *
 * The reference has no FFI for this path; these entry points are what a Rust shim
 * (`extern "C"` block) binds in place of the bodies of the cited reference
 * functions.  Everything [twenty-first]-only -- Fiat-Shamir sampling, BFieldCodec, the prover's RNG,
 * authentication structures -- stays on the Rust side and crosses this boundary as plain data.
 *
 * Conventions
 *  - A BFieldElement is one uint64_t holding the Montgomery word a*2^64 mod p, p = 2^64-2^32+1,
 *    exactly BFieldElement::raw_u64() (triton-constraint-builder/src/codegen.rs:28-31,926-944).
 *    An XFieldElement is 3 consecutive words, a Digest 5 words.  field_kind = 1 (BFE) or 3 (XFE).
 *  - Pointers named d_* are DEVICE pointers (hipMalloc / tvm_malloc / a torch tensor's data_ptr);
 *    pointers named h_* are host pointers.  No entry point takes ownership of caller memory.
 *  - Domains are passed as the reference holds them: (offset, generator, length)
 *    (arithmetic_domain.rs:34-47); the library never invents a root of unity for a caller's domain.
 *  - Every function returns a status (0 = ok) mapped 1:1 on the reference's error enums
 *    (triton-vm/src/error.rs:150-186) and never aborts or throws across the boundary.  Device
 *    out-of-memory is TVM_ERR_OUT_OF_MEMORY and leaves the context usable, mirroring the
 *    reference's try_reserve_exact fallback (master_table.rs:268-271).
 *  - Work is enqueued on the context's HIP stream; functions that return host data synchronise
 *    that stream, the others do not.  One context per proving thread (lib.rs:522-532); the calling
 *    thread's current HIP device must be the one the context was created on.
 */
#ifndef TRITON_HIP_H
#define TRITON_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVM_OK 0
#define TVM_ERR_INVALID_ARGUMENT 1 /* ArithmeticDomainError, TableRowConversionError, length mismatches */
#define TVM_ERR_OUT_OF_MEMORY 2    /* ProvingError::OutOfMemory -- recoverable */
#define TVM_ERR_DEVICE 3           /* HIP runtime failure */
#define TVM_ERR_UNSUPPORTED 4

typedef struct tvm_ctx tvm_ctx;
typedef struct tvm_table tvm_table; /* device-resident low-degree-extended master table */

typedef struct {
    uint64_t offset;    /* Montgomery word */
    uint64_t generator; /* Montgomery word; order == length */
    uint64_t length;    /* power of two */
} tvm_domain;

/* ---- context ------------------------------------------------------------------------------ */
int32_t tvm_abi_version(void);
/* `hip_stream` may be NULL (the context creates its own stream) or an existing hipStream_t. */
int32_t tvm_ctx_create(int32_t device, void* hip_stream, tvm_ctx** out);
void tvm_ctx_destroy(tvm_ctx* ctx);
const char* tvm_last_error(const tvm_ctx* ctx);
const char* tvm_status_string(int32_t status);
int32_t tvm_sync(tvm_ctx* ctx);
/* Device memory comes from a per-context caching allocator (a prove() at 2^20 rows turns over ~45 GiB of
 * tables; the driver's allocator costs hundreds of ms at that size).  tvm_free and tvm_table_free return
 * blocks to the cache for stream-ordered reuse; tvm_ctx_trim gives the cache back to the driver. */
int32_t tvm_malloc(tvm_ctx* ctx, size_t bytes, void** d_ptr);
int32_t tvm_free(tvm_ctx* ctx, void* d_ptr);
int32_t tvm_ctx_trim(tvm_ctx* ctx);
/* Options.  TVM_OPTION_AIR_VALID_TRACE (default 0): the caller guarantees that the master tables come from a valid
 * execution trace -- what Prover::prove is for.  tvm_all_quotients_combined may then use that the constraint quotients
 * are polynomials of known degree: the consistency / transition constraints are evaluated on half of the quotient
 * domain and the codeword completed by interpolation; bit-identical to the row-by-row evaluation on a valid trace,
 * different on an invalid one (where both yield a proof the verifier rejects). */
#define TVM_OPTION_AIR_VALID_TRACE 1
/* Tuning values (per context; they change launch shapes, never results).  TVM_OPTION_LDE_CHUNK_COLUMNS: columns per chunk of
 * tvm_lde_table's three-pass transform, 0 (default) = 96 while the chunk's intermediates fit comfortably, else 32.
 * TVM_OPTION_MERKLE_MIN_WORKGROUPS: tvm_merkle_tree gives a workgroup several groups of parents while at least this many
 * workgroups remain (default 4096; 0 restores it) -- the tests lower it to reach that path with small trees.
 * The library reads NO environment variable: a prover behind triton_vm::prove() must not change kernels on an inherited
 * environment. */
#define TVM_OPTION_LDE_CHUNK_COLUMNS 2
#define TVM_OPTION_MERKLE_MIN_WORKGROUPS 3
/* TVM_OPTION_LDE_PASS2_TILES = 1: the middle pass of tvm_lde_table on 2048-point axes (2^21 / 2^22 rows) runs the position-major tile
 * kernel (k_lde_pass2_v3) instead of k_lde_pass2_fused (and the generic kernel on 1024-point axes): the A/B switch of profiles/r05_*. */
#define TVM_OPTION_LDE_PASS2_TILES 4
/* TVM_OPTION_AIR_FORK_MAX_WORKGROUPS (default 256; 0 = never): tvm_all_quotients_combined / tvm_air_class_values on a quotient domain of at
 * most this many workgroups of 256 rows launch the parts of the AIR on four streams side by side (the context's and three of its
 * own, joined before the call returns to the context's stream), and on such a domain valid-trace mode evaluates row by row. */
#define TVM_OPTION_AIR_FORK_MAX_WORKGROUPS 5
/* TVM_OPTION_MERKLE_SUBTREES (default 1): the levels of a Merkle tree between 32768 and 64 parents are built up to seven to a launch
 * (a workgroup per subtree of 64 parents); 0: one launch per level (A/B). */
#define TVM_OPTION_MERKLE_SUBTREES 6
int32_t tvm_ctx_set_option(tvm_ctx* ctx, int32_t option, uint64_t value);
/* Cap on the bytes this context may hold through tvm_malloc / table handles (0 = no cap).  Requests beyond it fail
 * with TVM_ERR_OUT_OF_MEMORY exactly like a full device: the knob a host uses to share a GPU, and what the tests use to
 * walk the reference's out-of-memory fallback (master_table.rs:268-271 -> the coset-wise path). */
int32_t tvm_ctx_set_memory_limit(tvm_ctx* ctx, size_t bytes);
/* bytes currently held from the driver through this context (live blocks + cached blocks) */
int32_t tvm_ctx_memory_held(const tvm_ctx* ctx, size_t* bytes);
/* what this context could still obtain through tvm_malloc / table handles right now: the device's free memory plus the
 * context's own cached blocks (they are given back before a request fails), capped by the memory limit; and the device's
 * total.  Either pointer may be null.  The figure the host's memory policy plans with (master_table.rs:268-271: "does the
 * cached extension fit?") instead of running into the failure. */
int32_t tvm_ctx_memory_info(const tvm_ctx* ctx, size_t* available_bytes, size_t* device_total_bytes);
int32_t tvm_memcpy_h2d(tvm_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int32_t tvm_memcpy_d2h(tvm_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* device to device, ordered on the context's stream, NOT synchronised (same device, or a peer-accessible one). */
int32_t tvm_memcpy_d2d(tvm_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* the hipStream_t every call of this context is ordered on (so that a collective library -- RCCL -- can be enqueued
 * behind the kernels that produce its operands and ahead of those that consume its results; triton_vm_amd/host/rccl_comm.cpp) */
void* tvm_ctx_stream(const tvm_ctx* ctx);
/* The context's SIDE LANE: a second stream (created on first use) for exchanges that run UNDER the kernels of the context's stream --
 * the coefficient exchange of the column split (MasterTable::low_degree_extend_over, triton_vm_amd/host/triton_host.cpp) and its
 * communicators (rccl_comm.cpp enqueues ncclAllGather on it; the in-process communicator of sharded_host.cpp copies on it).  Ordering is
 * by events only, no host synchronisation:
 *   tvm_side_begin       the side lane waits for everything queued on the context's stream so far (the exchange's operands);
 *   tvm_side_memcpy_d2d  a device-to-device copy on the side lane;
 *   tvm_side_mark(slot)  records "everything queued on the side lane so far is done" in slot < TVM_SIDE_SLOTS;
 *   tvm_side_wait(slot)  the context's stream waits for that mark -- which also orders the pool's reuse of a freed block behind
 *                        the exchange that read or wrote it (a mark never recorded is no wait);
 *   tvm_side_sync        the HOST waits for the side lane (in-process communicators: a peer's copies out of this rank's buffer). */
#define TVM_SIDE_SLOTS 16
void* tvm_ctx_side_stream(tvm_ctx* ctx); /* NULL if it cannot be created */
int32_t tvm_side_begin(tvm_ctx* ctx);
int32_t tvm_side_memcpy_d2d(tvm_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int32_t tvm_side_mark(tvm_ctx* ctx, uint32_t slot);
int32_t tvm_side_wait(tvm_ctx* ctx, uint32_t slot);
int32_t tvm_side_sync(tvm_ctx* ctx);

/* HIP-event stopwatch on the context's stream (bench.py's live kernel timing) */
int32_t tvm_timer_start(tvm_ctx* ctx);
int32_t tvm_timer_stop(tvm_ctx* ctx, float* h_elapsed_ms); /* synchronises the stream */
/* Synthetic tables for benchmarks: n canonical Montgomery words < p from a counter-based generator */
int32_t tvm_synthetic_fill(tvm_ctx* ctx, uint64_t* d_data, uint64_t n_words, uint64_t seed);

/* Elementwise d_out[i] = d_a[i] op d_b[i] on Montgomery words through the device's field arithmetic
 * (op 0: +, 1: -, 2: *, 3: a^7, 4: a * 2^b for a plain integer b < 192): the known-answer hook for the hand-scheduled carry
 * chains in csrc/field.h (BFieldElement's Add/Sub/Mul, twenty-first; KAT triton-constraint-builder/src/codegen.rs:926-944)
 * and for the shift forms of csrc/ntt_shift.h.  op 32 + 4K + 2*dit + inverse (K = 1..4): the in-register transforms of
 * 2^K points with power-of-two twiddles on consecutive groups of 2^K words of d_a (d_b unused): dit = bit-reversed in /
 * natural out, else natural in / bit-reversed out; inverse = with the inverse root (no scaling by 2^-K). */
int32_t tvm_field_op(tvm_ctx* ctx, int32_t op, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n);

/* ---- L2 / L3: ArithmeticDomain::{evaluate, interpolate} (arithmetic_domain.rs:141-189) -------
 * d_coeffs: n_coeffs elements, d_values: domain.length elements, both arrays of field_kind-word
 * elements.  evaluate accepts n_coeffs > domain.length (chunk folding, :153-167). */
int32_t tvm_evaluate(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_coeffs, uint64_t n_coeffs,
                     tvm_domain domain, uint64_t* d_values);
int32_t tvm_interpolate(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_values, tvm_domain domain,
                        uint64_t* d_coeffs);
/* twenty-first `ntt` / `intt` (used at stark.rs:872,877,997,1002,1176): in place, natural order,
 * with the root of unity given explicitly. */
int32_t tvm_ntt(tvm_ctx* ctx, int32_t field_kind, uint64_t* d_data, uint64_t length, uint64_t generator);
int32_t tvm_intt(tvm_ctx* ctx, int32_t field_kind, uint64_t* d_data, uint64_t length, uint64_t generator);

/* ---- L1 + L4: MasterTable::maybe_low_degree_extend_all_columns (master_table.rs:258-322) -----
 * d_trace: column-major [n_cols][n_rows] elements (Array2 `.f()` order, master_table.rs:888,1013);
 * d_randomizers: the h coefficients of trace_randomizer_for_column(c) (master_table.rs:423-434),
 * [n_cols][h] elements, generated by the host RNG.  The extended table stays on the device behind
 * the handle (row i = evaluation at eval.offset * eval.generator^i). */
int32_t tvm_lde_table(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_trace, uint64_t n_rows,
                      uint64_t n_cols, const uint64_t* d_randomizers, uint64_t num_trace_randomizers,
                      tvm_domain trace_domain, tvm_domain evaluation_domain, tvm_table** out);
/* The same extension split at the coefficients -- the COLUMN sharding of SURVEY 8(e): a rank interpolates a range of columns,
 * the coefficients are exchanged (an all-gather), every rank extends all columns onto its own part of the evaluation domain.
 * A "virtual column" is one base-field component of a column (n_cols * field_kind of them, in the order (column, component)).
 *   tvm_lde_column_coefficients   the inverse transforms of the virtual columns [first, first + n): n_rows words each into
 *                                 d_coeffs[(v - first) * n_rows ...], in the library's COEFFICIENT FORM -- the scaled
 *                                 coefficients in the order its extension kernels read them; opaque, position-independent
 *                                 (a block of it can be moved anywhere), consumed only by tvm_lde_table_add_columns of the
 *                                 same trace length.
 *   tvm_lde_table_begin           a table handle like tvm_lde_table's, its columns not yet written
 *   tvm_lde_table_add_columns     the virtual columns [first, first + n) of the table from their coefficient form, with the
 *                                 trace randomizers of master_table.rs:392-403 (d_randomizers as in tvm_lde_table: all columns)
 *   tvm_lde_table_end             once every column is written (the successor rows of the row-pair views)
 * begin + add_columns over all columns + end == tvm_lde_table, word for word. */
int32_t tvm_lde_column_coefficients(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_trace, uint64_t n_rows, uint64_t n_cols,
                                    tvm_domain trace, uint64_t first_virtual_column, uint64_t n_virtual_columns, uint64_t* d_coeffs);
int32_t tvm_lde_table_begin(tvm_ctx* ctx, int32_t field_kind, uint64_t n_rows, uint64_t n_cols, uint64_t num_trace_randomizers,
                            tvm_domain trace, tvm_domain eval, tvm_table** out);
int32_t tvm_lde_table_add_columns(tvm_ctx* ctx, tvm_table* table, const uint64_t* d_coeffs, uint64_t first_virtual_column,
                                  uint64_t n_virtual_columns, const uint64_t* d_randomizers, uint64_t num_trace_randomizers,
                                  tvm_domain trace, tvm_domain eval);
int32_t tvm_lde_table_end(tvm_ctx* ctx, tvm_table* table);
void tvm_table_free(tvm_ctx* ctx, tvm_table* table);
uint64_t tvm_table_num_rows(const tvm_table* table);
uint64_t tvm_table_num_columns(const tvm_table* table);
int32_t tvm_table_field_kind(const tvm_table* table);
/* the reference's row-major Array2 [rows][n_cols] (master_table.rs:304-305), into device memory */
int32_t tvm_table_export_row_major(tvm_ctx* ctx, const tvm_table* table, uint64_t* d_out);
/* MasterTable::reveal_rows, cached branch (master_table.rs:548-555): rows of the ldt-domain view
 * (stride rows/ldt_length) at the given indices, row-major into host memory */
int32_t tvm_table_reveal_rows(tvm_ctx* ctx, const tvm_table* table, uint64_t ldt_length,
                              const uint64_t* h_row_indices, uint64_t n_indices, uint64_t* h_out);

/* ---- H1 / H2: row hashing and Merkle trees (master_table.rs:443-468) ------------------------
 * Digest i = Tip5::hash_varlen(row i of the ldt-domain view, XFE rows flattened c0,c1,c2). */
int32_t tvm_hash_rows(tvm_ctx* ctx, const tvm_table* table, uint64_t ldt_length, uint64_t* d_digests);
/* MerkleTree::par_new: d_nodes holds 2*n_leaves digests in heap order (node 1 = root, node
 * n_leaves + i = leaf i, node 0 zeroed). */
int32_t tvm_merkle_tree(tvm_ctx* ctx, const uint64_t* d_leaf_digests, uint64_t n_leaves, uint64_t* d_nodes);
/* MasterTable::merkle_tree: both of the above, leaves hashed straight into the node array */
int32_t tvm_table_merkle_tree(tvm_ctx* ctx, const tvm_table* table, uint64_t ldt_length, uint64_t* d_nodes);
/* ProverRound::merkle_tree_from_codeword (fri.rs:343-347): leaf = Digest::from(xfe), no hashing */
int32_t tvm_codeword_merkle_tree(tvm_ctx* ctx, const uint64_t* d_xfe_codeword, uint64_t length, uint64_t* d_nodes);

/* ---- O1: MasterTable::out_of_domain_row (master_table.rs:348-390) ---------------------------
 * Every column of the trace-randomized table evaluated at n_points XFE indeterminates (the prover
 * uses alpha and omega*alpha, stark.rs:451-469).  h_rows_out: [n_points][n_cols] XFE on the host. */
int32_t tvm_out_of_domain_rows(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_trace, uint64_t n_rows,
                               uint64_t n_cols, const uint64_t* d_randomizers, uint64_t num_trace_randomizers,
                               tvm_domain trace_domain, const uint64_t* h_points, uint32_t n_points,
                               uint64_t* h_rows_out);

/* ---- C1: MasterTable::weighted_sum_of_columns (master_table.rs:512-542) -----------------------
 * d_poly receives the 2*n_rows XFE coefficients (zero padded above n_rows + h) of
 * sum_c w_c * (column_c interpolant + zerofier * randomizer_c).  h_weights: n_cols XFE. */
int32_t tvm_weighted_sum_of_columns(tvm_ctx* ctx, int32_t field_kind, const uint64_t* d_trace, uint64_t n_rows,
                                    uint64_t n_cols, const uint64_t* d_randomizers, uint64_t num_trace_randomizers,
                                    tvm_domain trace_domain, const uint64_t* h_weights, uint64_t* d_poly);
/* d_a[i] += d_b[i] for n XFE (main + aux combination polynomial, stark.rs:516) */
int32_t tvm_xfe_add_assign(tvm_ctx* ctx, uint64_t* d_a, const uint64_t* d_b, uint64_t n);
/* d_out[i] = sum_k h_weights[k] * d_vectors[k * stride + i], i < n: a weighted sum of XFE vectors (polynomials) that lie
 * `stride` elements apart -- the P and R combinations of the segment POLYNOMIALS (stark.rs:520-536), which the sharded
 * host evaluates on its own rows of the short domain when the quotient domain is shorter than the LDT domain. */
int32_t tvm_xfe_linear_combination(tvm_ctx* ctx, const uint64_t* d_vectors, uint32_t n_vectors, uint64_t stride, uint64_t n,
                                   const uint64_t* h_weights, uint64_t* d_out);
/* Polynomial::evaluate at XFE points (stark.rs:480,491,568,579,590,600): d_coeffs n XFE -> h_out n_points XFE */
int32_t tvm_evaluate_at_points(tvm_ctx* ctx, const uint64_t* d_coeffs, uint64_t n, const uint64_t* h_points,
                               uint32_t n_points, uint64_t* h_out);
/* The same for n_polys polynomials of n coefficients each, `stride` XFE apart in d_coeffs, at the same points, in ONE round trip:
 * h_out[(p * n_points + j) * 3 ..] = polynomial p at point j (the five quotient-segment polynomials at the two out-of-domain
 * points, stark.rs:474-495). */
int32_t tvm_evaluate_polys_at_points(tvm_ctx* ctx, const uint64_t* d_coeffs, uint64_t n, uint64_t stride, uint32_t n_polys,
                                     const uint64_t* h_points, uint32_t n_points, uint64_t* h_out);

/* ---- degree-lowering fill (second half) ---------------------------------------
 * The generated DegreeLoweringTable::fill_derived_main_columns / fill_derived_aux_columns
 * (triton-constraint-builder/src/substitutions.rs:128-205, row loops :236-400; called at the end of
 * MasterMainTable::pad, master_table.rs:980-982, and of MasterMainTable::extend, :1066-1072): 230 of the 379
 * main columns and 41 of the 90 non-randomizer aux columns are values of the substitution rules the degree
 * lowering introduced.  Tables are the column-major traces tvm_lde_table takes: main [379][n_rows] words, aux
 * [>= 90][n_rows][3] words, filled in place.  Single-row sections derive every row; the transition section
 * derives rows 0 .. n_rows-2 from a row and its successor and leaves 0 in the last row (the reference's table is
 * zero-initialised there).  main: columns 149..378 from columns 0..148; aux: columns 49..89 from the main
 * table (already filled), aux columns 0..48 and the 63 challenges (host, XFE). */
int32_t tvm_fill_derived_main_columns(tvm_ctx* ctx, uint64_t* d_main_trace, uint64_t n_rows);
int32_t tvm_fill_derived_aux_columns(tvm_ctx* ctx, const uint64_t* d_main_trace, uint64_t* d_aux_trace, uint64_t n_rows,
                                     const uint64_t* h_challenges);

/* ---- main-table fill from the AET (the `fill` half) ----------------------------------
 * MasterMainTable::new's table fills (master_table.rs:881-931; table/{op_stack,ram,jump_stack,processor,program,hash,
 * cascade,lookup,u32}.rs `fill`).  The AET is handed over as AlgebraicExecutionTrace holds it (aet.rs:41-96): trace arrays
 * row-major in Montgomery words, multiplicities as plain integers.  Every array may live in HOST memory (it is staged
 * through the context's pool: 327 MB of processor trace at 2^20 cycles, ~6 ms over PCIe) or in DEVICE memory of the
 * context's GPU (a host that keeps the trace resident next to the prover: the library reads it where it lies); the
 * library tells the two apart per pointer (hipPointerGetAttributes). */
typedef struct {
    const uint64_t* program_words;              /* Program::to_bwords(), program_len words */
    const uint32_t* instruction_multiplicities; /* [program_len] */
    uint64_t program_len;
    const uint64_t* processor_trace;            /* [processor_len][39] */
    uint64_t processor_len;
    const uint64_t* op_stack_trace;             /* op_stack_underflow_trace [op_stack_len][4] */
    uint64_t op_stack_len;
    const uint64_t* ram_trace;                  /* [ram_len][7] (RamTableCall::to_table_row: the last 3 columns are 0) */
    uint64_t ram_len;
    /* bezout_coefficient_polynomials_coefficients(unique RAM pointers in ascending order) (ram.rs:152-207), num_ram_pointers
     * words each, when the host has them (twenty-first polynomial arithmetic); both null and num_ram_pointers 0: they are
     * computed on the device from the sorted RAM table (tvm_bezout_coefficients' algorithm) */
    const uint64_t* bezout_coefficients_0;
    const uint64_t* bezout_coefficients_1;
    uint64_t num_ram_pointers;
    const uint64_t* program_hash_trace;         /* [program_hash_len][67] */
    uint64_t program_hash_len;
    const uint64_t* sponge_trace;               /* [sponge_len][67] */
    uint64_t sponge_len;
    const uint64_t* hash_trace;                 /* [hash_len][67] */
    uint64_t hash_len;
    const uint64_t* u32_entries;                /* u32_entries in IndexMap order: [u32_len][4] = opcode (plain integer),
                                                   left operand, right operand (Montgomery words), multiplicity (plain) */
    uint64_t u32_len;
    const uint64_t* cascade_entries;            /* cascade_table_lookup_multiplicities in IndexMap order: [cascade_len][2] =
                                                   16-bit limb, multiplicity (plain integers) */
    uint64_t cascade_len;
    const uint64_t* lookup_multiplicities;      /* [256] plain integers */
} tvm_aet;
/* d_main_trace: [379][n_rows] words; columns 0..148 are written (zeros below each table's length), n_rows = the padded
 * height.  h_table_lengths_out[9]: the tables' lengths in the order of tvm_pad_main_table, which is the next call. */
int32_t tvm_fill_main_table(tvm_ctx* ctx, const tvm_aet* aet, uint64_t* d_main_trace, uint64_t n_rows,
                            uint64_t* h_table_lengths_out);

/* ---- main-table pad (the `pad` half) ----------------------------------------------
 * MasterMainTable::pad (master_table.rs:932-983) without its degree-lowering tail (tvm_fill_derived_main_columns): the
 * nine table-specific padding rules (table/program.rs:77-127, processor.rs:70-96, op_stack.rs:205-219, ram.rs:86-101,
 * jump_stack.rs:144-199, hash.rs:280-309, cascade.rs:60-67, lookup.rs:114-118, u32.rs:127-152), in place.
 * d_main_trace: [379][n_rows] words whose columns 0..148 hold the filled tables in their first h_table_lengths[t]
 * rows (t = Program, Processor, OpStack, Ram, JumpStack, Hash, Cascade, Lookup, U32; all_table_lengths,
 * master_table.rs:985-1001) and zeros below; n_rows = the padded height (a power of two). */
int32_t tvm_pad_main_table(tvm_ctx* ctx, uint64_t* d_main_trace, uint64_t n_rows, const uint64_t* h_table_lengths);

/* ---- auxiliary-table extend (first half) ----------------------------------------
 * MasterMainTable::extend (master_table.rs:1006-1075) without its degree-lowering tail and without the batch-randomizer
 * column: the 49 cross-table-argument columns of the nine tables (table/{program,processor,op_stack,ram,jump_stack,
 * hash,cascade,lookup,u32}.rs `extend`: running products, running evaluations, logarithmic-derivative sums with
 * Challenges' initial values) from the padded main table.  d_main_trace: [379][n_rows] words (columns 0..148 are
 * read); d_aux_trace: [91][n_rows][3] words, columns 0..48 are written, the others left alone; h_challenges: the 63
 * challenges incl. the 4 derived ones (host, XFE).  Call tvm_fill_derived_aux_columns afterwards. */
int32_t tvm_extend_aux_table(tvm_ctx* ctx, const uint64_t* d_main_trace, uint64_t* d_aux_trace, uint64_t n_rows,
                             const uint64_t* h_challenges);

/* ---- A1-A3: all_quotients_combined (master_table.rs:1264-1363) ------------------------------
 * Evaluates the 81 + 97 + 403 + 23 = 604 AIR constraints of the (degree-lowered) Triton VM AIR on
 * every quotient-domain row of the two extended tables, including the zerofier inverses
 * (master_table.rs:1194-1250), and combines them with the quotient weights.  The quotient domain is
 * the stride-(rows/length) view of the tables.  h_challenges: 63 XFE (Challenges, challenges.rs:54);
 * h_weights: 604 XFE (stark.rs:396-401).  d_quotient_codeword: quotient_domain.length XFE. */
#define TVM_NUM_CHALLENGES 63
#define TVM_NUM_QUOTIENT_WEIGHTS 604
#define TVM_NUM_MAIN_COLUMNS 379
#define TVM_NUM_AUX_COLUMNS 91
int32_t tvm_all_quotients_combined(tvm_ctx* ctx, const tvm_table* main_table, const tvm_table* aux_table,
                                   tvm_domain trace_domain, tvm_domain quotient_domain,
                                   const uint64_t* h_challenges, const uint64_t* h_weights,
                                   uint64_t* d_quotient_codeword);

/* ---- A3 in valid-trace mode, piecewise (for a host that distributes the evaluation over several GPUs).  The
 * constraints are generated in four classes by the length of their quotients: bit 0 = the initial / terminal quotients of degree-4
 * constraints (needed on every point of the quotient domain), bit 1 = "half" (fewer than 4N coefficients), bit 2 = "quarter"
 * (fewer than 2N), bit 3 = "three cosets" (fewer than 3N).  tvm_air_class_cosets: how many cosets of the trace domain determine
 * each class's quotient polynomial for these tables' interpolant lengths (out[class bit]; 0 = the degree bound does not hold: use
 * tvm_all_quotients_combined on every point).  tvm_air_class_values: sum over the constraints of the classes in class_mask of
 * weight * constraint * zerofier inverse on coset `coset` of table_domain (the domain the tables were extended onto; X = its
 * length / N cosets, coset k = the points k + X j) -> trace_domain.length XFE in the coset's natural order.
 * tvm_coset_values_to_coefficients: the values of a polynomial of fewer than n_cosets * N coefficients on n_cosets <= 4 cosets
 * h_offsets[j] * <w_N> -> its coefficients, ADDED to d_coeffs (n_cosets * N XFE).  On a valid trace the sum over the classes,
 * evaluated on the quotient domain, plus the class-0 values is exactly the output of tvm_all_quotients_combined. */
#define TVM_AIR_CLASS_FULL 1
#define TVM_AIR_CLASS_HALF 2
#define TVM_AIR_CLASS_QUARTER 4
#define TVM_AIR_CLASS_THREE 8
int32_t tvm_air_class_cosets(const tvm_table* main_table, const tvm_table* aux_table, tvm_domain trace_domain, uint32_t out[4]);
int32_t tvm_air_class_values(tvm_ctx* ctx, const tvm_table* main_table, const tvm_table* aux_table, tvm_domain trace_domain,
                             tvm_domain table_domain, uint32_t coset, uint32_t class_mask, const uint64_t* h_challenges,
                             const uint64_t* h_weights, uint64_t* d_out);
int32_t tvm_coset_values_to_coefficients(tvm_ctx* ctx, tvm_domain trace_domain, uint32_t n_cosets, const uint64_t* h_offsets,
                                         const uint64_t* const* d_values, uint64_t* d_coeffs);

/* ---- Q3-Q5: interpolate_quotient_segments, ldt_domain_segment_polynomials and
 * randomize_quotient_segments (stark.rs:1224-1283, 1302-1356) in one call ---------------------
 * d_quotient_codeword: quotient_domain.length XFE (the output of all_quotients_combined).
 * h_randomizer: the n_rand XFE coefficients of the quotient-segment randomizer polynomial s_4
 * (host RNG, stark.rs:1316-1322).  zeta: Stark::ZETA as a Montgomery word (stark.rs:1801).
 * out_table: the [ldt.length][5] XFE table of randomized segment codewords (device handle, hashable
 * with tvm_table_merkle_tree); d_polys: [5][poly_len] XFE coefficients, poly_len >= max(quotient/4, n_rand). */
int32_t tvm_quotient_segments(tvm_ctx* ctx, const uint64_t* d_quotient_codeword, tvm_domain quotient_domain,
                              tvm_domain ldt_domain, const uint64_t* h_randomizer, uint64_t n_rand, uint64_t zeta,
                              tvm_table** out_table, uint64_t* d_polys, uint64_t poly_len);
/* row-wise linear combination of a table's columns over its ldt-domain view: d_out[i] = sum_c w_c * cell(i, c)
 * (the P and R combinations of the randomized segments, stark.rs:520-540).  h_weights: n_cols XFE. */
int32_t tvm_table_linear_combination(tvm_ctx* ctx, const tvm_table* table, uint64_t ldt_length,
                                     const uint64_t* h_weights, uint64_t* d_out);

/* ---- D1 + D2: deep_codeword for up to 4 components and their weighted sum (stark.rs:566-625) --
 * d_out[i] = sum_k weight_k * (codeword_k[i] - value_k) / (domain.value(i) - point_k), all XFE. */
int32_t tvm_deep_codeword(tvm_ctx* ctx, uint32_t n_components, const uint64_t* const* d_codewords, tvm_domain domain,
                          const uint64_t* h_points, const uint64_t* h_values, const uint64_t* h_weights,
                          uint64_t* d_out);

/* ---- F1: ProverRound::split_and_fold (fri.rs:349-366): domain.length XFE -> domain.length/2 XFE */
int32_t tvm_fri_split_and_fold(tvm_ctx* ctx, const uint64_t* d_codeword, tvm_domain domain,
                               const uint64_t* h_challenge, uint64_t* d_out);

/* ---- F1 + H2 + the transcript in between: the COMMIT PHASE of Fri::prove (fri.rs:212-263) without the host in the loop.
 * For r = 0 .. n_rounds: the Merkle tree of codeword r (d_nodes[r], 10 * (domain.length >> r) words, as
 * tvm_codeword_merkle_tree), its root absorbed into the Fiat-Shamir sponge as ProofItem::MerkleRoot (proof_item.rs:96,
 * proof_stream.rs:54-59: the encoding [0, root] padded with 1, 0, 0, 0 is exactly one absorbed block) and, for r < n_rounds,
 * one scalar sampled from the sponge (proof_stream.rs:81-84) and codeword r + 1 = split_and_fold(codeword r, that scalar)
 * written to d_codewords[r] (3 * (domain.length >> (r + 1)) words).  The sponge runs on the device: no stream
 * synchronisation until the roots and challenges are copied out at the end.
 * h_sponge_state: the 16 words of the Tip5 sponge before the first root is enqueued (not modified: the caller replays the
 * n_rounds + 1 enqueues and n_rounds samplings on its own sponge, which must reproduce h_challenges);
 * h_roots: (n_rounds + 1) * 5 words; h_challenges: n_rounds * 3 words. */
int32_t tvm_fri_commit_phase(tvm_ctx* ctx, const uint64_t* d_codeword, tvm_domain domain, uint32_t n_rounds,
                             const uint64_t* h_sponge_state, uint64_t* const* d_codewords, uint64_t* const* d_nodes,
                             uint64_t* h_roots, uint64_t* h_challenges);

/* ---- coset-wise ("just in time") evaluation: Prover::compute_quotient_segments_with_jit_lde and the JIT branch of
 * hash_all_ldt_domain_rows (stark.rs:805-1006, master_table.rs:470-503) -------------------------------------
 * When the extended tables do not fit, the reference evaluates them coset by coset.  Here a coset group is an
 * arithmetic domain of its own (offset * generator^r, generator^R, length / R), so tvm_lde_table,
 * tvm_hash_rows and tvm_all_quotients_combined are simply called on that domain (triton_vm_amd/jit.py);
 * the only extra primitive is putting a group's results back into row order:
 *   d_dst[(i * stride + offset) * elem_words + w] = d_src[i * elem_words + w],  i < n. */
int32_t tvm_scatter_strided(tvm_ctx* ctx, const uint64_t* d_src, uint32_t elem_words, uint64_t n, uint64_t stride,
                            uint64_t offset, uint64_t* d_dst);

/* ---- S1: the STIR prover (low_degree_test/stir.rs:885-993); the round loop, sampling and inclusion proofs stay on the
 * host (triton_vm_amd/low_degree_test.py is the mirror) ---------------------------------------------------------
 * StirMerkleTree::new (stir.rs:1380-1419): leaf i = hash_varlen(codeword[i], codeword[i + d], ...), d = length /
 * stack_height; d_nodes: [2 d][5] in heap order. */
int32_t tvm_stir_merkle_tree(tvm_ctx* ctx, const uint64_t* d_xfe_codeword, uint64_t length, uint32_t stack_height,
                             uint64_t* d_nodes);
/* Stir::fold_polynomial (stir.rs:1132-1147): d_out[i] = sum_j d_poly[ff*i + j] * r^j; ceil(n_coeffs / ff) XFE out */
int32_t tvm_fold_polynomial(tvm_ctx* ctx, const uint64_t* d_poly, uint64_t n_coeffs, uint32_t folding_factor,
                            const uint64_t* h_randomness, uint64_t* d_out);
/* The witness polynomial of the next round (stir.rs:945-966):
 *   ((folded - Ans) / prod_j (X - quotient_set[j])) * sum_{e <= k} (r X)^e
 * h_answer_poly: the k coefficients of Ans (Polynomial::interpolate over the quotient set, tvm_host_xfe_interpolate);
 * work_domain: any coset of at least n_coeffs points that avoids the quotient set (the caller's choice);
 * d_out_poly: work_domain.length XFE coefficients (zero above n_coeffs - 1). */
int32_t tvm_stir_next_polynomial(tvm_ctx* ctx, const uint64_t* d_folded_poly, uint64_t n_coeffs,
                                 const uint64_t* h_quotient_set, const uint64_t* h_answer_poly, uint32_t k,
                                 const uint64_t* h_degree_correction_randomness, tvm_domain work_domain,
                                 uint64_t* d_out_poly);
/* host: coefficients of the polynomial of degree < k through k XFE points (pairwise distinct) */
int32_t tvm_host_xfe_interpolate(const uint64_t* points, const uint64_t* values, uint32_t k, uint64_t* out_coeffs);
/* the same on the device (one workgroup; k <= 256, beyond that it calls the host function): the prover's round loop does not
 * leave the GPU idle for the 0.9 ms the host form takes at k = 204 */
int32_t tvm_xfe_interpolate(tvm_ctx* ctx, const uint64_t* h_points, const uint64_t* h_values, uint32_t k, uint64_t* h_out_coeffs);

/* ---- small transfers and host-side helpers ----------------------------------------------------
 * gather n elements of elem_words words each from a device array at the given element indices
 * (Merkle authentication-structure nodes, FRI leaves, single codeword entries) into host memory */
int32_t tvm_gather_elements(tvm_ctx* ctx, const uint64_t* d_src, uint32_t elem_words, const uint64_t* h_indices,
                            uint64_t n, uint64_t* h_out);
/* The same for n_jobs gathers at once -- all the FRI responses and authentication structures of a proof (fri.rs:295-319),
 * or the three trace openings (stark.rs:672-716) -- with ONE transfer of the indices, one of the results and two stream
 * synchronisations in all instead of two per gather.  Job j reads n[j] elements of elem_words[j] words from d_src[j] at
 * h_indices[j][0..n[j]) and writes them to h_out[j]. */
int32_t tvm_gather_elements_batch(tvm_ctx* ctx, uint32_t n_jobs, const uint64_t* const* d_src, const uint32_t* elem_words,
                                  const uint64_t* const* h_indices, const uint64_t* n, uint64_t* const* h_out);
/* Host-side (CPU) field helpers for callers that do not link twenty-first (the C++/Python test
 * drivers); a Rust host uses twenty-first instead.  No device work. */
void tvm_host_tip5_permutation(uint64_t state[16]);
/* overwrite-mode absorb of n words followed by the padding 1, 0, ... (variable-length domain) */
void tvm_host_sponge_pad_and_absorb(uint64_t state[16], const uint64_t* words, uint64_t n);
void tvm_host_xfe_mul(const uint64_t a[3], const uint64_t b[3], uint64_t out[3]);
void tvm_host_xfe_inv(const uint64_t a[3], uint64_t out[3]);
void tvm_host_xfe_powers(const uint64_t x[3], uint64_t first_exponent, uint64_t n, uint64_t* out /* n XFE */);
/* out[j] = sum_i coeffs[i] * points[j]^i, or with zerofier != 0: prod_i (points[j] - coeffs[i]); n, m XFE in, m XFE out */
void tvm_host_xfe_poly_eval(const uint64_t* coeffs, uint64_t n, const uint64_t* points, uint64_t m, int32_t zerofier,
                            uint64_t* out);
/* n draws of `rng.random::<BFieldElement>()` from `StdRng::from_seed(seed)` (the prover's trace, batch and quotient
 * randomizers: master_table.rs:423-434, 1006-1024, stark.rs:1315-1322; a Rust host uses rand itself), Montgomery words */
void tvm_host_stdrng_elements(const uint8_t seed[32], uint64_t n, uint64_t* out);
/* the same n elements into device memory, generated on the device (one ChaCha block per work-item); when a draw takes the
 * range sampler's rare short path the stream is regenerated sequentially on the host -- the result is always the host's */
int32_t tvm_stdrng_elements(tvm_ctx* ctx, const uint8_t seed[32], uint64_t n, uint64_t* d_out);
/* n_streams generators at once: stream s is `StdRng::from_seed(seed + s)` (the 256-bit little-endian sum: rng_from_offset_seed,
 * master_table.rs:630-662 -- a table's trace randomizers, one stream per column, master_table.rs:423-434), per_stream draws
 * each, d_out[s * per_stream + i].  One launch; the same rare-short-path rule as tvm_stdrng_elements (then every stream is drawn
 * on the host).  Replaces 1.3 ms of sequential ChaCha on the host per proof (470 streams at 198 randomizers). */
int32_t tvm_stdrng_streams(tvm_ctx* ctx, const uint8_t seed[32], uint64_t n_streams, uint64_t per_stream, uint64_t* d_out);

/* The RAM table's Bezout coefficient polynomials: bezout_coefficient_polynomials_coefficients (table/ram.rs:152-207) for n
 * pairwise distinct roots (the unique RAM pointers, device array): a and b with a * rp + b * rp' = 1, rp = prod (X - r_i),
 * n coefficients each (device arrays).  TVM_ERR_INVALID_ARGUMENT when two roots coincide. */
int32_t tvm_bezout_coefficients(tvm_ctx* ctx, const uint64_t* d_roots, uint64_t n, uint64_t* d_a, uint64_t* d_b);

/* ---- verifier batch work --------------------------------------------------------
 * Verifier::verify's work over the num_first_round_queries revealed rows (stark.rs:1388-1763), all host data in / out:
 * the leaf digests of revealed rows (Tip5::hash_varlen of each row, stark.rs:1598-1601, 1620-1660) ... */
int32_t tvm_verifier_row_digests(tvm_ctx* ctx, const uint64_t* h_rows, uint64_t n_rows, uint64_t row_words,
                                 uint64_t* h_digests);
/* ... and, per revealed row, linearly_sum_main_and_aux_row (stark.rs:1765-1787), the quotient-segment sums, the four
 * Stark::deep_update values (stark.rs:2096-2103) and their weighted sum (stark.rs:1678-1755), which must equal the
 * value the low-degree test revealed at that index (the comparison stays with the caller).  h_main_rows [n][379],
 * h_aux_rows [n][91][3], h_quot_rows [n][5][3], h_row_indices [n] (indices into ldt_domain), h_weights_main_aux
 * [470][3], h_weights_quot [5][3], h_weights_deep [4][3]; h_ood_points / h_ood_values [4][3] in the order current row,
 * next row, alpha^4, (zeta*alpha)^4 (stark.rs:1722-1745).  h_out: [n][3]. */
int32_t tvm_verifier_deep_values(tvm_ctx* ctx, const uint64_t* h_main_rows, const uint64_t* h_aux_rows,
                                 const uint64_t* h_quot_rows, const uint64_t* h_row_indices, uint64_t n_rows,
                                 tvm_domain ldt_domain, const uint64_t* h_weights_main_aux,
                                 const uint64_t* h_weights_quot, const uint64_t* h_weights_deep,
                                 const uint64_t* h_ood_points, const uint64_t* h_ood_values, uint64_t* h_out);

/* host: the 604 AIR constraints on ONE row pair whose main rows are XFieldElements -- what Verifier::verify evaluates on the
 * out-of-domain rows (stark.rs:1466-1491; MasterAuxTable::evaluate_{initial,consistency,transition,terminal}_constraints in
 * this order).  h_main_* [379][3], h_aux_* [91][3], h_challenges [63][3]; h_out [604][3]: initial (81), consistency (97),
 * transition (403), terminal (23). */
int32_t tvm_host_air_constraints(const uint64_t* h_main_cur, const uint64_t* h_aux_cur, const uint64_t* h_main_next,
                                 const uint64_t* h_aux_next, const uint64_t* h_challenges, uint64_t* h_out);

#ifdef __cplusplus
}
#endif
#endif /* TRITON_HIP_H */
