// fill.hip -- launches of the generated degree-lowering fill kernels (fill.h, fill_gen.hip).
#include "fill_gen.h"
#include "kernels.h"

namespace tvm {

// table 0: main (derives columns 149..378 of d_main in place); table 1: aux (derives columns 49..89 of d_aux from
// d_main, the earlier aux columns and the challenges).  Sections in the reference's order: init, cons, tran, term.
int fill_degree_lowering(tvm_ctx* c, int table, u64* d_main, u64* d_aux, const u64* d_challenges, u64 n) {
    FillArgs a;
    a.main = d_main;
    a.aux = d_aux;
    a.ch = d_challenges;
    a.n = n;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (table == 0) {
        for (int k = 0; k < TVM_FILL_MAIN_NUM_KERNELS; k++) TVM_LAUNCH(TVM_FILL_MAIN_KERNELS[k], grid, block, 0, c->stream, a);
    } else {
        for (int k = 0; k < TVM_FILL_AUX_NUM_KERNELS; k++) TVM_LAUNCH(TVM_FILL_AUX_KERNELS[k], grid, block, 0, c->stream, a);
    }
    TVM_HIP_CHECK(c, hipGetLastError());
    return TVM_OK;
}

}  // namespace tvm
