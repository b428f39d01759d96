// kernels.h -- host-callable entry points of the kernel translation units (namespace tvm).
#pragma once
#include <vector>
#include "context.h"

namespace tvm {
// ntt.hip
int ntt_columns(tvm_ctx* c, const u64* in, u64 in_len, int in_fk, u64 in_col_stride, u64* out, int out_fk,
                u64 out_col_stride, u64 out_mul, u64 out_add, int ncols, u64 n, u64 w, u64 in_scale, u64 out_scale,
                u64 out_mult);
TabLayout lde_table_layout(u64 n_rows, u64 L);   // the layout lde_table writes (incl. the successor blocks)
int lde_table(tvm_ctx* c, int fk, const u64* trace, u64 n_rows, u64 n_cols, const u64* rnd, u64 h, u64 trace_gen,
              u64 eval_offset, u64 eval_gen, u64 L, u64* table, int chunk_cols, const LdeSplit* split = nullptr);
// hash.hip
int hash_rows(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, u64 stride, u64* digests);
int merkle_tree_from_leaves(tvm_ctx* c, u64* nodes, u64 n_leaves);
// merkle_subtrees.hip: `levels` (<= 7) consecutive levels from the one of `widest` parents (a multiple of 64) upwards, in one launch
int merkle_subtrees(tvm_ctx* c, u64* nodes, u64 widest, int levels);
int xfe_leaves(tvm_ctx* c, const u64* cw, u64 plane, u64 n, u64* leaves);
int gather_rows(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, const u64* d_idx, u64 n, u64* d_out);   // d_idx: domain rows
int table_to_row_major(tvm_ctx* c, const u64* table, const TabLayout& layout, int W, u64* d_out);
int columns_to_table(tvm_ctx* c, const u64* cols, u64 col_stride, u64 L, int W, u64* table);   // natural row order
int fill_successor_blocks(tvm_ctx* c, u64* table, const TabLayout& layout, int W);
// poly.hip
int out_of_domain_rows(tvm_ctx* c, int fk, const u64* trace, u64 n, u64 n_cols, const u64* rnd, u64 h, u64 trace_gen,
                       const u64* d_points, int n_points, u64* d_rows);
int weighted_row_sum(tvm_ctx* c, int fk, const u64* trace, u64 n, u64 n_cols, const u64* d_w, int accumulate, u64* d_values);
int randomizer_contribution(tvm_ctx* c, int fk, const u64* rnd, u64 n, u64 n_cols, u64 h, const u64* d_w, u64* d_poly);
int randomized_segments(tvm_ctx* c, const u64* d_q_coeffs, u64 q_len, const u64* d_rnd, u64 n_rand, u64 zeta, u64 poly_len,
                        u64* d_polys);
int table_lincomb(tvm_ctx* c, const u64* table, const TabLayout& layout, int fk, u64 n_cols, u64 stride, const u64* d_w, u64* d_out);
int poly_eval(tvm_ctx* c, const u64* d_coeffs, u64 n, const u64* d_points, int n_points, u64* d_out);
int deep_sum(tvm_ctx* c, int n_comp, const u64* const* d_cw, const u64* h_points, const u64* h_values, const u64* h_weights,
             u64 offset, u64 gen, u64 n, u64* d_out);
int fri_fold(tvm_ctx* c, const u64* d_cw, u64 n, u64 offset, u64 gen, const u64* h_challenge, u64* d_out,
             const u64* d_challenge = nullptr);   // the challenge from the host, or three words in device memory
// Fiat-Shamir on the device: absorb ProofItem::MerkleRoot(nodes[1]) into the sponge `state` (16 words, device), then -- if
// d_challenge -- sample one scalar (3 words) from it
int sponge_absorb_root_and_sample(tvm_ctx* c, u64* d_state, const u64* d_root, u64* d_challenge);
// stir.hip
int stir_hash_stacked(tvm_ctx* c, const u64* cw, u64 n, int stack_height, u64* digests);
int xfe_interpolate(tvm_ctx* c, const u64* d_points, const u64* d_values, int k, u64* d_out, int* d_status);   // k <= 256
int stir_fold_polynomial(tvm_ctx* c, const u64* poly, u64 n, int ff, const u64* h_r, u64* out);
int stir_quotient(tvm_ctx* c, u64* vals, u64 n, u64 offset, u64 gen, const u64* d_points, const u64* d_answer,
                  const u64* d_answer_values, u32 k, u32 kb, const u64* h_r);
// fill.hip
int fill_degree_lowering(tvm_ctx* c, int table, u64* d_main, u64* d_aux, const u64* d_challenges, u64 n);
// bezout.hip
int bezout_coefficients(tvm_ctx* c, const u64* d_roots, u64 n, u64* d_a, u64* d_b);
// fill_aet.hip
int fill_main_table(tvm_ctx* c, const tvm_aet* aet, u64* d_main, u64 n, u64* h_lengths);
// pad.hip
int pad_main_table(tvm_ctx* c, u64* d_main, u64 n, const u64* lengths);
// extend.hip
int extend_aux_table(tvm_ctx* c, const u64* d_main, u64* d_aux, const u64* d_challenges, u64 n);
// verify.hip
int hash_varlen_rows(tvm_ctx* c, const u64* d_rows, u64 n, int W, u64* d_digests);
int verifier_deep_values(tvm_ctx* c, const u64* d_main_rows, int n_main, const u64* d_aux_rows, int n_aux, const u64* d_quot_rows,
                         const u64* d_row_idx, u64 q, u64 offset, u64 gen, const u64* d_w_ma, const u64* d_small, u64* d_out);
// air.hip
int all_quotients_combined(tvm_ctx* c, const u64* main_table, const TabLayout& layout, u64 main_w, const u64* aux_table,
                           u64 aux_w, u64 trace_len, u64 trace_gen, u64 q_offset, u64 q_gen, u64 q_len, const u64* d_challenges,
                           const u64* d_weights, u64* d_out, int part_select = 0, int accumulate = 0);
// a quotient domain this short leaves the chip to the parts side by side (the fork lanes): the row-by-row evaluation of all ten
// parts then costs what its longest lane does, less than valid-trace mode's six evaluations and five transforms one behind another
bool air_parts_fork(const tvm_ctx* c, u64 q_len);
}  // namespace tvm

struct tvm_table {
    u64* data = nullptr;  // row-block-major over STORAGE rows, see context.h
    u64 rows = 0;         // rows of the domain the table is defined over
    TabLayout layout;     // domain row <-> storage row
    bool has_successor_blocks = false;  // pitch = (n2 + 1) * n1: the AIR kernels read the "next" row (master_table.rs:1305-1306,
                                        // index + rows/|trace| mod rows) at storage row + n1 without a wrap-around
    u64 n_cols = 0;       // in elements of the table's field
    int fk = 1;
    int W = 0;            // base-field words per row = n_cols * fk
    u64 interpolant_len = 0;  // tables made by tvm_lde_table: every column is a polynomial with at most this many
                              // coefficients (trace length + trace randomizers); 0 = unknown
    // tables under construction by tvm_lde_table_begin / _add_columns / _end (the column split): the domains given to begin, and which
    // virtual columns have been written -- add_columns and end refuse any other handle, other domains, and an incomplete table
    bool lde_split_open = false;
    u64 lde_trace_len = 0, lde_trace_gen = 0, lde_eval_offset = 0, lde_eval_gen = 0;
    std::vector<unsigned char> lde_written;
    size_t bytes() const { return (size_t)tvm_tab_words(layout.storage_rows(), (u64)W) * sizeof(u64); }
};
