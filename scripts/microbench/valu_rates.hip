// scripts/microbench/valu_rates.hip -- development microbenchmark: what does ONE wave-level vector
// instruction of each kind cost a gfx950 SIMD when W wavefronts per SIMD issue nothing else?  The root
// search is bound by VALU issue; its roofline needs the measured price of the
// instruction kinds it is made of (plain fp32, transcendental, select/compare/bit ops, LDS reads), not a
// flat figure.  Every kernel runs ITERS x 128 independent instructions of one kind per wavefront (16
// accumulators, so dependency latency is hidden even with one wavefront) and reports
//   SIMD cycles per wave-instruction = s_memtime ticks of the longest loop among a workgroup's wavefronts
//                                      / (instructions per wave x wavefronts per SIMD), mean over workgroups.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/microbench/valu_rates.hip -o gpurun_out/valu_rates
//   gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP16_(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP16(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X)   // 128 per loop trip

enum Op { FMA = 0, MUL, ADD, EXP, RSQ, RCP, SIN, CNDMASK, BFI, CMP, MAX, MOV, XOR, CVT, FMA64, MUL64, ADD64,
          DSREAD, MIX_LAYER,
          CNDMASK_S, CNDMASK_NEG, AND, ADDU, LSHL, RNDNE, FMAC, FMAMK, MULLIT, SUB, MAXABS, FMANEG, CMP_S, DPP, DSREAD2,
          MIN, MED3, ANDOR, LSHLOR, FREXP, LDEXP, FLOOR, SQRT, MULNEG, CMPCLASS, READLANE, CNDMASK_E64V, MIX_CND_V, MIX_CND_S, MIX_EXP, MIX_CMPCND, MIX_BFI, PKFMA, PKMUL, PKADD, NOPS };
static const char *names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_exp_f32", "v_rsq_f32", "v_rcp_f32", "v_sin_f32",
                              "v_cndmask_b32", "v_bfi_b32", "v_cmp_lt_f32", "v_max_f32", "v_mov_b32", "v_xor_b32",
                              "v_cvt_i32_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "ds_read_b32",
                              "mix of one recursion layer (86 plain : 5 transcendental)",
                              "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cndmask_b32_e64 -a, a (SGPR-pair mask)", "v_and_b32",
                              "v_add_u32", "v_lshlrev_b32", "v_rndne_f32", "v_fmac_f32", "v_fmamk_f32 (literal)",
                              "v_mul_f32 (literal)", "v_sub_f32", "v_max_f32_e64 |a|", "v_fma_f32 -a", "v_cmp_lt_f32_e64 (SGPR dst)",
                              "v_mov_b32_dpp row_shr:1", "ds_read2st64_b32", "v_min_f32", "v_med3_f32", "v_and_or_b32",
                              "v_lshl_or_b32", "v_frexp_exp_i32_f32", "v_ldexp_f32", "v_floor_f32", "v_sqrt_f32",
                              "v_mul_f32_e64 -a", "v_cmp_class_f32", "v_readlane_b32", "v_cndmask_b32_e64 (vcc)",
                              "1 v_cndmask_b32 (vcc) + 7 v_fma_f32, per instruction", "1 v_cndmask_b32_e64 (SGPR pair) + 7 v_fma_f32",
                              "1 v_exp_f32 + 7 v_fma_f32", "v_cmp_lt_f32 vcc; v_cndmask vcc; 6 v_fma_f32",
                              "1 v_bfi_b32 + 7 v_fma_f32", "v_pk_fma_f32 (two fp32 per lane)", "v_pk_mul_f32", "v_pk_add_f32"};

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, long long *cyc, int iters, float seed)
{
    __shared__ float lds[1024];
    float a[16];
    double d[16];
    const float b = seed + 1.0e-7f * threadIdx.x, c = 0.999f;
    const double bd = b, cd = c;
    for (int i = 0; i < 16; ++i) { a[i] = seed * (i + 1); d[i] = a[i]; }
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = seed;
    __syncthreads();
    const unsigned addr = (threadIdx.x * 4u) & 4095u;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(b), "v"(c) : "vcc");
    asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" ::"v"(b), "v"(c) : "s20", "s21");
    const long long t0 = __builtin_readcyclecounter();       // s_memtime: shader cycles
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == RSQ) {
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == SIN) {
#define X(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
            REP16(X)
#undef X
        } else if (OP == BFI) {
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == CMP) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a[i]), "v"(c) : "vcc");
            REP16(X)
#undef X
        } else if (OP == MAX) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == MOV) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == XOR) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == CVT) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(cd), "v"(bd));
            REP16(X)
#undef X
        } else if (OP == MUL64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if (OP == ADD64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if (OP == DSREAD) {
#define X(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(addr));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == CNDMASK_S) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c) : "s20", "s21");
            REP16(X)
#undef X
        } else if (OP == CNDMASK_NEG) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, -%0, %0, s[20:21]" : "+v"(a[i]) : : "s20", "s21");
            REP16(X)
#undef X
        } else if (OP == AND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == ADDU) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == LSHL) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == RNDNE) {
#define X(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == FMAC) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == FMAMK) {
#define X(i) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (OP == MULLIT) {
#define X(i) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == SUB) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == MAXABS) {
#define X(i) asm volatile("v_max_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == FMANEG) {
#define X(i) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == CMP_S) {
#define X(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" ::"v"(a[i]), "v"(c) : "s20", "s21");
            REP16(X)
#undef X
        } else if (OP == DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == DSREAD2) {
#define X(i) asm volatile("ds_read2st64_b32 %0, %1 offset1:1" : "=v"(d[i]) : "v"(addr));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (OP == MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == MED3) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == ANDOR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP16(X)
#undef X
        } else if (OP == LSHLOR) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (OP == FREXP) {
#define X(i) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == LDEXP) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(0));
            REP16(X)
#undef X
        } else if (OP == FLOOR) {
#define X(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == SQRT) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == MULNEG) {
#define X(i) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == CMPCLASS) {
#define X(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" ::"v"(a[i]), "v"(3) : "vcc");
            REP16(X)
#undef X
        } else if (OP == READLANE) {
            int sg;
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == CNDMASK_E64V) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, -%0, %0, vcc" : "+v"(a[i]) : : );
            REP16(X)
#undef X
        } else if (OP == MIX_CND_V || OP == MIX_CND_S || OP == MIX_EXP || OP == MIX_CMPCND || OP == MIX_BFI) {
#define P(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define Q(i) if (OP == MIX_CND_V) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c)); \
             else if (OP == MIX_CND_S) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(c) : "s20", "s21"); \
             else if (OP == MIX_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); \
             else if (OP == MIX_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c), "v"(b)); \
             else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
#define R(i) if (OP == MIX_CMPCND) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a[i]), "v"(c) : "vcc"); else P(i)
#define BLK(o) R(o) P((o + 1) & 15) P((o + 2) & 15) P((o + 3) & 15) Q((o + 4) & 15) P((o + 5) & 15) P((o + 6) & 15) P((o + 7) & 15)
            BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8) BLK(0) BLK(8)
#undef BLK
#undef R
#undef Q
#undef P
        } else if (OP == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(cd), "v"(bd));
            REP16(X)
#undef X
        } else if (OP == PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if (OP == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            REP16(X)
#undef X
        } else if (OP == MIX_LAYER) {
            // the instruction mix of one layer of the Rayleigh recursion on its evanescent path
            // (ISA of surfdisp_phase_kernel<2,4,...>): 2 v_rsq + 4 v_exp among ~85 plain instructions; 96 here
            // = 6 blocks of (1 transcendental + 15 plain), all independent across the 16 accumulators
#define P(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
#define T(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            T(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
            P(0) T(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
            P(0) P(1) T(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
            P(0) P(1) P(2) T(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
            P(0) P(1) P(2) P(3) T(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
            P(0) P(1) P(2) P(3) P(4) T(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
#undef P
#undef T
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Res { double ms, cycles; };
template <int OP>
static Res run(float *out, long long *cyc, int cus, int wps, int iters)
{
    const int wpb = wps < 4 ? wps : 4;                       // wavefronts per SIMD one block brings
    const int threads = 256 * wpb, blocks = cus * (wps / wpb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);      // warm up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rate_kernel<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    const int nw = blocks * threads / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    // the SIMD arbitrates oldest-first: the oldest wavefronts of a SIMD finish early, the youngest carries the
    // whole interval, so the SIMD's busy time is the LONGEST loop among its wavefronts (mean over the launch
    // of the per-block maximum)
    const int wpblk = threads / 64;
    double sum = 0.0;
    for (int b = 0; b < blocks; ++b) {
        long long mx = 0;
        for (int w = 0; w < wpblk; ++w) mx = h[(size_t)b * wpblk + w] > mx ? h[(size_t)b * wpblk + w] : mx;
        sum += (double)mx;
    }
    return Res{ms, sum / blocks};
}

int main(int argc, char **argv)
{
    const int iters = 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;                 // Hz (nominal peak shader clock)
    const int cus = prop.multiProcessorCount;
    printf("device %s: %d CUs, nominal clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    float *out;
    long long *cyc;
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
    hipMalloc(&cyc, (size_t)cus * 8 * 4 * sizeof(long long));
    const int wps_list[] = {1, 2, 4};   // 8 would need two workgroups per CU, whose intervals do not coincide
    double clk_seen = 0.0;
    printf("%-58s", "SIMD cycles per wave-instruction at wavefronts/SIMD =");
    for (int w : wps_list) printf(" %6d", w);
    printf("\n");
    for (int op = 0; op < NOPS; ++op) {
        printf("%-58s", names[op]);
        for (int wps : wps_list) {
            Res r{0.0, 0.0};
            switch (op) {
#define C(O) case O: r = run<O>(out, cyc, cus, wps, iters); break;
                C(FMA) C(MUL) C(ADD) C(EXP) C(RSQ) C(RCP) C(SIN) C(CNDMASK) C(BFI) C(CMP) C(MAX) C(MOV) C(XOR) C(CVT)
                C(FMA64) C(MUL64) C(ADD64) C(DSREAD) C(MIX_LAYER)
                C(CNDMASK_S) C(CNDMASK_NEG) C(AND) C(ADDU) C(LSHL) C(RNDNE) C(FMAC) C(FMAMK) C(MULLIT) C(SUB) C(MAXABS) C(FMANEG)
                C(CMP_S) C(DPP) C(DSREAD2) C(MIN) C(MED3) C(ANDOR) C(LSHLOR) C(FREXP) C(LDEXP) C(FLOOR) C(SQRT) C(MULNEG)
                C(CMPCLASS) C(READLANE) C(CNDMASK_E64V) C(MIX_CND_V) C(MIX_CND_S) C(MIX_EXP) C(MIX_CMPCND) C(MIX_BFI) C(PKFMA) C(PKMUL) C(PKADD)
#undef C
            }
            const double n_inst = (double)iters * (op == MIX_LAYER ? 96.0 : 128.0);
            printf(" %6.2f", r.cycles / (n_inst * wps));
            if (op == FMA) { clk_seen = r.cycles / (r.ms * 1e-3); fprintf(stderr, "v_fma_f32 wps %d: %.4f ms wall, %.0f ticks per wave -> %.0f MHz\n", wps, r.ms, r.cycles, clk_seen / 1e6); }
        }
        printf("\n");
    }
    printf("s_memtime ticks per second of wall time (v_fma_f32, 8 wavefronts/SIMD): %.0f MHz\n", clk_seen / 1e6);
    hipFree(out); hipFree(cyc);
    return 0;
}
