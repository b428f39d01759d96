// fill.h -- the degree-lowering fill: values of the derived ("degree lowering") columns of the master tables.
//
// Replaces the reference's generated DegreeLoweringTable::fill_derived_main_columns / fill_derived_aux_columns
// (/root/reference/triton-constraint-builder/src/substitutions.rs:128-205; called at the
// end of MasterMainTable::pad, master_table.rs:980-982, and of MasterMainTable::extend, :1066-1072): SURVEY.md 8(f) #1,
// second half.  One work-item per row of the (padded) trace; the tables are the column-major traces that
// tvm_lde_table takes -- main [379][n] words, aux [>= 90][n][3] words -- so every access of a wavefront is a run of
// consecutive words.  Sections are separate launches (a rule of a later section reads derived columns of earlier
// ones, the transition section those of the NEXT row); inside a section a work-item keeps the columns it has
// just derived in registers.  The generated kernels are csrc/fill_gen.hip (tools/air/export_fill.py).
#pragma once
#include "context.h"

namespace tvm {

struct FillArgs {
    u64* main;          // [n_main][n] base-field words (read; written by the main kernels)
    u64* aux;           // [n_aux][n][3] (written by the aux kernels), or nullptr
    const u64* ch;      // challenges [TVM_AIR_NUM_CHALLENGES][3], device (aux kernels)
    u64 n;              // rows
};

// rows: a single-row section derives every row; the transition section rows 0 .. n-2, and its columns are 0 in
// the last row (the reference's table is zero-initialised and the generated loop does not touch that row)
#define FILL_PROLOGUE(dual, first_col, n_cols, words)                                              \
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;                                     \
    if (r >= a.n) return;                                                                          \
    if ((dual) && r == a.n - 1) {                                                                  \
        u64* t_ = (words) == 1 ? a.main : a.aux;                                                   \
        for (int c_ = (first_col); c_ < (first_col) + (n_cols); c_++)                             \
            for (int k_ = 0; k_ < (words); k_++) t_[((u64)c_ * a.n + r) * (words) + k_] = 0;       \
        return;                                                                                    \
    }
#define FMC(c) (a.main[(u64)(c) * a.n + r])
#define FMN(c) (a.main[(u64)(c) * a.n + r + 1])
#define FILL_LD_X(p) xfe_make((p)[0], (p)[1], (p)[2])
#define FAC(c) FILL_LD_X(a.aux + ((u64)(c) * a.n + r) * 3)
#define FAN(c) FILL_LD_X(a.aux + ((u64)(c) * a.n + r + 1) * 3)
#define FCH(k) FILL_LD_X(a.ch + 3 * (k))
#define FILL_STORE_B(c, v) a.main[(u64)(c) * a.n + r] = (v)
#define FILL_STORE_X(c, v)                                     \
    do {                                                       \
        u64* p_ = a.aux + ((u64)(c) * a.n + r) * 3;            \
        p_[0] = (v).c0, p_[1] = (v).c1, p_[2] = (v).c2;        \
    } while (0)

}  // namespace tvm
