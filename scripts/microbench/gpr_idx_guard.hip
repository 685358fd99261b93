// Reduced reproducer (hipcc 7.2, gfx950) of the register-indexing miscompile met in surfdisp_mcmc_propose_kernel (r03):
// a register-resident local array written under a bounds guard with a RUN-TIME index.
//   hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only gpr_idx_guard.hip -o - | grep -n -B4 -A12 s_set_gpr_idx_on
// What comes out (scripts/microbench/gpr_idx_guard_isa.txt): the loop body copies S to a shadow register block, writes
// S[2k+1], S[2k+2] into the shadow through s_set_gpr_idx_on ... gpr_idx(DST) UNCONDITIONALLY - index 2k+2 runs to 30 for a
// 15-entry array, i.e. up to 32 registers past the block - and only then selects shadow or original with v_cndmask under
// the guard's condition.  The out-of-range indexed writes land in whatever lives behind the block (in the r03 kernel: the
// output pointer -> memory fault on the first depth-4 tree).  Guard in the product: every register array is indexed with
// compile-time constants only, and tests/test_isa_guard.py fails on any s_set_gpr_idx / v_movrel in the built library.
#include <hip/hip_runtime.h>
__global__ void tree(const double *p, double *out, int depth)
{
    const int M = depth > 1 ? (1 << depth) - 1 : 1;          // 1, 3, 7 or 15 nodes
    double S[15];
    S[0] = p[threadIdx.x];
    for (int k = 0; k < M; ++k) {                             // run-time trip count: not unrolled
        const double x = S[k];
        const double nv = x * 1.5 + 1.0;
        out[(size_t)threadIdx.x * M + k] = nv;
        if (2 * k + 2 < M) { S[2 * k + 1] = nv; S[2 * k + 2] = x; }   // guarded stores, run-time index
    }
}
