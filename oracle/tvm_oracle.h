/*
 * tvm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithms on the hot path of
 * triton_vm::stark::Prover::prove (reference: /root/reference/triton-vm/src/stark.rs:331-719).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product library (libtriton_hip.so) never links, loads or calls it.
 *
 * The arithmetic itself lives in the third-party crate `twenty-first = "2.0.0"`
 * (/root/reference/Cargo.toml:104), which is NOT in the reference tree.  Its published algorithms
 * are restated here from the in-tree specification (specification/src/isa.md:5-8,
 * tips/tip-0005/tip-0005.md) and pinned by reference-held values: the TIP-0005 vectors and the Montgomery KAT
 * (tests/test_oracle_pins.py), the program digests of stark.rs:4828-4838 and program.rs:496-510 (multi-block
 * overwrite-mode hash_varlen), the AIR fingerprint of master_table.rs:2328-2414 (tests/test_air_fingerprint.py) and
 * the vanishing of the AIR on valid traces (tests/test_vm_tables.py; this also pins the fixed-length hashing domain
 * and merkle_step's sibling order through the Hash table's constraints), and -- end to end -- the reference's two
 * proof-digest snapshots (proof.rs:200-226, stark.rs:2434-2460): oracle/real_prover.py, which sequences these functions
 * with the oracle-side VM and the restated transcript, reproduces Tip5::hash(proof) for both (tests/test_proof_snapshot.py).
 * That pins every function below that prove() uses, including the items the tree has no separate vector for
 * (root-of-unity table, generator, Merkle node layout, Digest::from(XFE)): marked "pinned by the proof snapshots".
 *
 * Data representation: a BFieldElement is one uint64_t holding the Montgomery
 * word a*2^64 mod p (triton-constraint-builder/src/codegen.rs:926-944: 42 <-> 180388626390);
 * an XFieldElement is 3 consecutive words (c0,c1,c2) of F_p[X]/(X^3 - X + 1); a Digest is 5 words.
 */
#ifndef TVM_ORACLE_H
#define TVM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t offset;    /* Montgomery word */
    uint64_t generator; /* Montgomery word, order == length */
    uint64_t length;    /* power of two */
} orc_domain;

/* ---- base field ---- */
uint64_t orc_bfe_new(uint64_t value);        /* canonical value -> Montgomery word */
void orc_bfe_new_array(uint64_t* a, uint64_t n);            /* in place: canonical values -> Montgomery words */
void orc_bfe_value_array(uint64_t* a, uint64_t n);          /* in place: Montgomery words -> canonical values */
uint64_t orc_bfe_value(uint64_t raw);        /* Montgomery word -> canonical value */
uint64_t orc_bfe_add(uint64_t a, uint64_t b);
uint64_t orc_bfe_sub(uint64_t a, uint64_t b);
uint64_t orc_bfe_mul(uint64_t a, uint64_t b);
uint64_t orc_bfe_inv(uint64_t a);
uint64_t orc_bfe_pow(uint64_t a, uint64_t e);
uint64_t orc_bfe_generator(void);                       /* 7; pinned by the proof snapshots */
uint64_t orc_bfe_primitive_root(uint64_t order);        /* 7^((p-1)/2^32) squared down; pinned by the proof snapshots */
void orc_bfe_batch_inv(uint64_t* a, size_t n);

/* ---- extension field ---- */
void orc_xfe_add(const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_xfe_sub(const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_xfe_mul(const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_xfe_inv(const uint64_t* a, uint64_t* out);
void orc_xfe_pow(const uint64_t* a, uint64_t e, uint64_t* out);
void orc_xfe_batch_inv(uint64_t* a, size_t n);

/* ---- domains / NTT (arithmetic_domain.rs:141-296) ---- */
orc_domain orc_domain_of_length(uint64_t length);
orc_domain orc_domain_pow(orc_domain d, uint64_t exponent);
uint64_t orc_domain_value(orc_domain d, uint64_t i);
void orc_domain_values(orc_domain d, uint64_t* out);
/* in-place, natural order in and out; fk = 1 (BFE) or 3 (XFE, three interleaved transforms) */
void orc_ntt(uint64_t* a, uint64_t n, int fk);
void orc_intt(uint64_t* a, uint64_t n, int fk);
void orc_coset_evaluate(int fk, const uint64_t* coeffs, uint64_t n_coeffs, orc_domain d, uint64_t* out);
void orc_coset_interpolate(int fk, const uint64_t* values, orc_domain d, uint64_t* out_coeffs);

/* ---- master-table LDE (master_table.rs:258-322, 392-403) ----
 * trace: column-major [n_cols][n_rows][fk]; randomizers: [n_cols][h][fk];
 * out: row-major [eval.length][n_cols][fk]. */
void orc_randomized_column_interpolant(int fk, const uint64_t* column, uint64_t n_rows,
                                       const uint64_t* randomizer, uint64_t h, uint64_t* out_coeffs /* 2*n_rows*fk */);
void orc_lde_table(int fk, const uint64_t* trace, uint64_t n_rows, uint64_t n_cols,
                   const uint64_t* randomizers, uint64_t h, orc_domain eval, uint64_t* out);

/* ---- Tip5 (tips/tip-0005/tip-0005.md:31-83) ---- */
void orc_tip5_permutation(uint64_t state[16]);
void orc_hash_varlen(const uint64_t* input, size_t len, uint64_t out[5]);
void orc_tip5_trace(const uint64_t in[16], uint64_t* out /* [6][16] */);
void orc_hash_pair(const uint64_t left[5], const uint64_t right[5], uint64_t out[5]); /* order pinned by the proof snapshots */
void orc_hash_10(const uint64_t input[10], uint64_t out[5]);
void orc_hash_rows(const uint64_t* rows, uint64_t n_rows, uint64_t row_words, uint64_t* digests);
/* nodes: [2*n_leaves][5]; nodes[1] = root, nodes[n_leaves + i] = leaf i, nodes[0] = 0 */
void orc_merkle_tree(const uint64_t* leaves, uint64_t n_leaves, uint64_t* nodes);
void orc_xfe_to_digest(const uint64_t* xfe, uint64_t n, uint64_t* digests); /* fri.rs:343-347 */

/* ---- quotient plumbing (master_table.rs:1194-1250, stark.rs:1224-1356) ---- */
void orc_zerofier_inverses(orc_domain trace, orc_domain quotient,
                           uint64_t* init, uint64_t* cons, uint64_t* tran, uint64_t* term);
/* quotient codeword (XFE, quotient.length) -> 4 segment polys (each quotient.length/4 XFE coeffs) */
void orc_interpolate_quotient_segments(const uint64_t* codeword, orc_domain quotient, uint64_t* seg_polys);
/* s_4 random poly (n_rand XFE coeffs) + 4 segment polys -> 5 randomized polys of length poly_len
 * (zero padded) and their LDT-domain codewords, row-major [ldt.length][5][3] (stark.rs:1302-1356) */
void orc_randomize_quotient_segments(const uint64_t* seg_polys, uint64_t seg_len,
                                     const uint64_t* randomizer, uint64_t n_rand, orc_domain ldt,
                                     uint64_t* out_polys, uint64_t poly_len, uint64_t* out_codewords);

/* all_quotients_combined (master_table.rs:1264-1363); row-major quotient-domain tables, 63 challenges,
 * 604 weights, out: q.length XFE */
void orc_quotients_combined(const uint64_t* main_rows, uint64_t n_main, const uint64_t* aux_rows, uint64_t n_aux,
                            orc_domain trace, orc_domain q, const uint64_t* challenges, const uint64_t* weights,
                            uint64_t* out);

/* the 604 constraint values on one (current, next) row pair in the generated evaluators' order
 * (evaluate_{initial,consistency,transition,terminal}_constraints; per section base-field constraints first);
 * main_words 1 = BFieldElement main rows, 3 = XFieldElement main rows; aux rows 91 xfe; out [604][3] */
void orc_air_constraint_values(const uint64_t* main_cur, const uint64_t* main_next, const uint64_t* aux_cur,
                               const uint64_t* aux_next, const uint64_t* challenges, int main_words, uint64_t* out);

/* ---- combination / DEEP / FRI (master_table.rs:348-390,512-542; stark.rs:1360-1379,2096; fri.rs:349-366) ---- */
void orc_weighted_sum_of_columns(int fk, const uint64_t* trace, uint64_t n_rows, uint64_t n_cols,
                                 const uint64_t* randomizers, uint64_t h, const uint64_t* weights /* n_cols xfe */,
                                 uint64_t* out_poly /* 2*n_rows xfe coeffs */);
void orc_out_of_domain_row(int fk, const uint64_t* trace, uint64_t n_rows, uint64_t n_cols,
                           const uint64_t* randomizers, uint64_t h, const uint64_t point[3],
                           uint64_t* out_row /* n_cols xfe */);
void orc_poly_eval_xfe(const uint64_t* coeffs, uint64_t n, const uint64_t point[3], uint64_t out[3]);
void orc_deep_codeword(const uint64_t* codeword, orc_domain d, const uint64_t point[3],
                       const uint64_t value[3], uint64_t* out);
void orc_fri_split_and_fold(const uint64_t* codeword, orc_domain d, const uint64_t challenge[3], uint64_t* out);

/* ---- tvm_oracle_fast.c: the optimised restatement timed by bench.py's cpu_baseline leg (same results, bit for bit) ---- */
void orcf_tip5_permutation(uint64_t st[16]);
void orcf_hash_rows(const uint64_t* rows, uint64_t n_rows, uint64_t w, uint64_t* digests);
void orcf_merkle_tree(const uint64_t* leaves, uint64_t n, uint64_t* nodes);
void orcf_ntt(uint64_t* a, uint64_t n, uint64_t generator);
void orcf_lde_table(const uint64_t* trace, uint64_t n, uint64_t n_cols, const uint64_t* rnd, uint64_t h, orc_domain eval, uint64_t* out);
void orcf_quotients_combined(const uint64_t* main_rows, uint64_t n_main, const uint64_t* aux_rows, uint64_t n_aux, orc_domain trace,
                             orc_domain q, const uint64_t* challenges, const uint64_t* weights, uint64_t* out);
void orcf_deep_codeword(const uint64_t* cw, orc_domain d, const uint64_t pt[3], const uint64_t val[3], uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
