// surfdisp_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// Batched surface-wave dispersion forward solver: the GPU counterpart of the reference's
// fast_surf() path (001cat/pySurfInv fast_surf_src/{fast_surf,init,calcul,flat1,surfa}.f).
// This is NOT a translation of the Fortran: the work decomposition, data layout and the
// root-refinement scheme are designed for 64-wide wavefronts; the reference is cited where a
// formula or a semantic rule is taken from it.
//
// Pipeline for one (batch, wave type):
//   K0 surfdisp_prep_kernel   : AoS model[B][5][L] -> SoA mdl[9][L][B] and / or rows[B][9][L]; per-layer earth-flattening
//                               factors computed ONCE per stack (flat1.f:33-69 recomputes them
//                               40x per solve), validation.
//   K1 surfdisp_phase_kernel  : phase velocities.  A TEAM of G lanes (G = 1..64, one wavefront holds
//                               64/G teams) owns one stack; its period-dependent working stack
//                               (b, rho, d, 1/a^2, 1/b^2 and one derived value per layer) lives in LDS; every loop
//                               iteration each lane evaluates the secular function at its own trial
//                               velocity with the 5-component (Rayleigh) or 2-component (Love)
//                               recursion state in registers.  Periods are walked in order inside the
//                               team (faithful start rule c1 = 0.9*c(k-1), mmax carry-over, failure
//                               guards, NaN semantics: calcul.f:104-220).  Template flags: INDEP (one
//                               team per (stack, period), SURFDISP_INDEPENDENT), FAST (opt-in heuristic
//                               scan, SURFDISP_FASTSCAN), EXACT (the fallback instantiation that restates
//                               DLTAR4 / DLTAR1 / NEVILL statement by statement for the stacks the
//                               production arithmetic cannot treat faithfully); phase-only calls skip
//                               the ellipticity recursions.  The teams of a wavefront run in LOCK STEP: all
//                               scan, then all refine, then all end their period (ST_WREF / ST_WEND).
//   K1b surfdisp_ellip_kernel : Rayleigh ellipticities of teams of >= 4 lanes, one lane per (stack, period),
//                               replaying the working stack's history the root search recorded.
//   K2 surfdisp_group_kernel  : group velocities, one lane per (stack, period): eigenfunction
//                               integration + energy integrals (surfa.f:714-1192 / 374-606), fp64 state
//                               for Rayleigh as in the reference (surfa.f:717-722); the KERN
//                               instantiation also writes the analytic partials dc/d(Vs, Vp, rho).
//   K3 surfdisp_finish_kernel : period-major internal results -> the caller's [B][P] arrays.
// No MFMA: the products are 5x5 / 4x4 / 2x2.  Algorithmic HBM traffic is one read of the model array and one
// write of c/U (the staged copies and period-major intermediates make it 138 MB per 65 536-stack batch against
// 24 MB); everything else is VALU + transcendental work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <atomic>
#include "surfdisp_internal.h"

#define SD_HD __host__ __device__
#ifndef SD_CERT_STRIDE
#define SD_CERT_STRIDE 6       // grid points per lane of a certified coarse pass, teams of <= 4 lanes (4: 0.516 ms, 6: 0.483, 8: 0.491 for 65 536 x L10)
#endif

namespace sd {

// ---- reference constants ---------------------------------------------------------------------
constexpr float R0      = 6371.0f;       // flat1.f:21
constexpr float PI_REF  = 3.1415927f;    // calcul.f:32, fast_surf.f:77
constexpr float DC      = 0.01f;         // init.f:25
constexpr float FACT    = 4.0f;          // init.f:25
constexpr float ACCUR   = 1.e-8f;        // surfa.f:191-192
constexpr float CLUSTER_DC = 1.0e-4f;    // spacing of clustered refine points (teams of <= 4 lanes)

// SoA field ids of mdl[NF][Lmax][B]
// (the raw thickness is not staged: nothing downstream reads it, only the flattened one, F_DFL)
enum { F_VP = 0, F_VS, F_RHO, F_QS, F_DIF, F_QQQ, F_DFL, F_HSF, F_HSR, NF = 9 };

SD_HD __forceinline__ bool fin(float x) { return fabsf(x) <= 3.402823466e38f; }   // finite, host+device
SD_HD __forceinline__ float pwr_of(int kind) { return kind == 1 ? 5.0f : 2.2750f; }  // flat1.f:27-28

// =================================================================================== K0: prep
// One lane per stack.  Flattening factors depend only on the radii, i.e. on the thickness prefix
// sums (flat1.f:33-37), not on the period or on how many layers are flattened:
//   regular layer i : dif_i, qqq_i (flat1.f:44-56), new thickness z1(i+1)-z1(i) (flat1.f:65-68)
//   layer i used as half space: hsf_i = a/r_i, hsr_i = (1/hsf_i)^pwr (flat1.f:58-62)
// The flattening factors difference nearly equal radii, which amplifies a 1-ulp difference between
// two powf/logf implementations to ~1e-3 of a thin layer's density/thickness.  Evaluate in fp64 and
// round once: that is the correctly rounded fp32 result, which is also what the reference's libm
// returns (glibc powf/logf are correctly rounded in all but vanishingly rare cases).
SD_HD __forceinline__ float powr32(float x, float y) { return (float)pow((double)x, (double)y); }
SD_HD __forceinline__ float log32(float x) { return (float)log((double)x); }

template <int KIND>
SD_HD inline void prep_stack(const PrepArgs &A, const int b)
{
    const int Lmax = A.Lmax, B = A.B;
    int n = A.nlay ? A.nlay[b] : Lmax;
    const float *src = A.model + (size_t)b * 5 * Lmax;
    float *mdl = A.mdl;
    bool ok = (n >= 2) && (n <= Lmax);
    const float pwr = pwr_of(KIND);
    const float apw = powr32(R0, pwr);
    float hmax = 0.0f, vs_prev = 0.0f, vp_prev = 0.0f;
    float hthick = 0.0f, rhomax = 0.0f, bmax = 0.0f;   // flattened: thickest layer, largest density and Vs
    bool mono = true;
    if (ok) {
        float hs = 0.0f;          // running thickness sum, fp32 in layer order (flat1.f:33-37)
        float r_i = R0;           // radius of the top of layer i
        float z_i = 0.0f;         // flattened depth of the top of layer i
        for (int i = 0; i < n; ++i) {
            const float vp = src[0 * Lmax + i], vs = src[1 * Lmax + i], rho = src[2 * Lmax + i];
            const float h = src[3 * Lmax + i], qs = src[4 * Lmax + i];
            if (!(fin(vp) && fin(vs) && fin(rho) && fin(h) && fin(qs)) ||
                !(vp > 0.0f) || !(rho > 0.0f) || (vs < 0.0f) || (h < 0.0f))
                ok = false;
            if (i < n - 1 && h > hmax) hmax = h;
            if ((i > 0 && (vs < vs_prev || vp < vp_prev))) mono = false;
            vs_prev = vs; vp_prev = vp;
            hs = hs + h;
            const float r_n = R0 - hs;                       // radius of the bottom of layer i
            float dif = 0.0f, qqq = 0.0f, dfl = 0.0f;
            if (i < n - 1) {
                const float fltd = log32(r_i / r_n);
                dif = (1.0f / r_n - 1.0f / r_i) * R0 / fltd;
                const float difr = powr32(r_i, pwr) - powr32(r_n, pwr);
                qqq = difr / (fltd * apw * pwr);
                const float z_n = R0 * log32(R0 / r_n);
                dfl = z_n - z_i;
                z_i = z_n;
                if (!(r_n > 0.0f) || !fin(dif) || !fin(qqq)) ok = false;
            }
            const float hsf = R0 / r_i;
            const float hsr = powr32(1.0f / hsf, pwr);
            // 6 % on the velocities for the attenuation correction (1 + qs ln(1/T)/pi, calcul.f:122-126)
            hthick = fmaxf(hthick, dfl);
            if (i > 0 && !(fabsf(vs) > ACCUR)) hthick = 1.0e30f;
            rhomax = fmaxf(rhomax, rho * fmaxf(qqq, hsr));
            bmax = fmaxf(bmax, 1.06f * vs * fmaxf(dif, hsf));
            const size_t o = (size_t)i * B + b;
            const size_t fs = (size_t)Lmax * B;
            const float v9[NF] = {vp, vs, rho, qs, dif, qqq, dfl, hsf, hsr};      // F_VP .. F_HSR
            for (int f = 0; f < NF; ++f) {
                if (A.write_soa) mdl[f * fs + o] = v9[f];
                if (A.rows) A.rows[((size_t)b * NF + f) * Lmax + i] = v9[f];
            }
            r_i = r_n;
        }
    }
    A.nl[b] = ok ? n : 0;          // 0 => BADMODEL: K1/K2 write zeros
    if (A.fsafe) A.fsafe[b] = (ok && mono) ? hmax : 1.0e30f;
    if (A.ovf) {                                               // inputs of the phase kernel's entry_overflow_risk
        A.ovf[b] = hthick;
        A.ovf[(size_t)B + b] = 2.0f * logf(fmaxf(rhomax, 1.0e-30f));
        A.ovf[2 * (size_t)B + b] = 4.0f * logf(fmaxf(2.0f * bmax * bmax, 1.0e-30f));
    }
    if (A.fb_count && b == 0) { A.fb_count[0] = 0; A.fb_count[1] = 0; A.fb_count[2] = 0; }   // empty list of stacks for the exact fallback; [1], [2]: scan trials / ellipticities re-evaluated
    if (A.nsolved_init) A.nsolved_init[b] = ok ? A.P : 0;      // independent mode: reduced with atomicMin
}

// Device version: TL lanes per stack (TL = 1 .. 64, a power of two), lane j owns layers j, j + TL, ...  The fp64
// pow / log of the flattening factors are most of the work and one lane walking 96 layers takes 0.22 ms whatever the
// batch size, so small batches spread a stack over a whole wavefront.  Bit-identical to prep_stack (which stays as the
// host reference of tests/hostcheck): the running thickness sum is formed in layer order by every lane for itself
// (fp32 adds only), and a layer's radii, depths and factors are the same expressions of that sum.
template <int KIND, int TL>
__global__ __launch_bounds__(256) void surfdisp_prep_kernel(PrepArgs A)
{
    const int Lmax = A.Lmax, B = A.B;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(tid / TL), j = (int)(tid % TL);
    const bool valid = b < B;
    int n = valid ? (A.nlay ? A.nlay[b] : Lmax) : 0;
    const float *src = A.model + (size_t)(valid ? b : 0) * 5 * Lmax;
    float *mdl = A.mdl;
    bool ok = (n >= 2) && (n <= Lmax);
    if (!ok) n = 0;
    const float pwr = pwr_of(KIND);
    const float apw = powr32(R0, pwr);
    const size_t fs = (size_t)Lmax * B;
    float hmax = 0.0f, hthick = 0.0f, rhomax = 0.0f, bmax = 0.0f;
    bool mono = true;
    float hs = 0.0f;                // running thickness sum over layers < i0 (fp32, layer order: flat1.f:33-37)
    int i0 = 0;
    for (int i = j; i < n; i += TL) {
        for (; i0 < i; ++i0) hs = hs + src[3 * Lmax + i0];
        const float vp = src[0 * Lmax + i], vs = src[1 * Lmax + i], rho = src[2 * Lmax + i];
        const float h = src[3 * Lmax + i], qs = src[4 * Lmax + i];
        if (!(fin(vp) && fin(vs) && fin(rho) && fin(h) && fin(qs)) ||
            !(vp > 0.0f) || !(rho > 0.0f) || (vs < 0.0f) || (h < 0.0f))
            ok = false;
        if (i < n - 1 && h > hmax) hmax = h;
        if (i > 0 && (vs < src[1 * Lmax + i - 1] || vp < src[0 * Lmax + i - 1])) mono = false;
        const float r_i = R0 - hs;                           // radius of the top of layer i
        const float r_n = R0 - (hs + h);                     // ... and of its bottom
        const float z_i = (i == 0) ? 0.0f : R0 * log32(R0 / r_i);
        float dif = 0.0f, qqq = 0.0f, dfl = 0.0f;
        if (i < n - 1) {
            const float fltd = log32(r_i / r_n);
            dif = (1.0f / r_n - 1.0f / r_i) * R0 / fltd;
            const float difr = powr32(r_i, pwr) - powr32(r_n, pwr);
            qqq = difr / (fltd * apw * pwr);
            dfl = R0 * log32(R0 / r_n) - z_i;
            if (!(r_n > 0.0f) || !fin(dif) || !fin(qqq)) ok = false;
        }
        const float hsf = R0 / r_i;
        const float hsr = powr32(1.0f / hsf, pwr);
        hthick = fmaxf(hthick, dfl);
        if (i > 0 && !(fabsf(vs) > ACCUR)) hthick = 1.0e30f;   // a liquid layer below the top: exact kernel (entry_overflow_risk)
        rhomax = fmaxf(rhomax, rho * fmaxf(qqq, hsr));
        bmax = fmaxf(bmax, 1.06f * vs * fmaxf(dif, hsf));
        const size_t o = (size_t)i * B + b;
        const float v9[NF] = {vp, vs, rho, qs, dif, qqq, dfl, hsf, hsr};          // F_VP .. F_HSR
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (A.write_soa) mdl[f * fs + o] = v9[f];
            if (A.rows) A.rows[((size_t)b * NF + f) * Lmax + i] = v9[f];          // consecutive lanes: consecutive words
        }
    }
    // the stack's verdict and statistics over the lanes of its team
#pragma unroll
    for (int d = TL >> 1; d > 0; d >>= 1) {
        const int ok_o = __shfl_xor((int)ok, d), mono_o = __shfl_xor((int)mono, d);   // every lane shuffles
        ok = ok && (ok_o != 0);
        mono = mono && (mono_o != 0);
        hmax = fmaxf(hmax, __shfl_xor(hmax, d));
        hthick = fmaxf(hthick, __shfl_xor(hthick, d));
        rhomax = fmaxf(rhomax, __shfl_xor(rhomax, d));
        bmax = fmaxf(bmax, __shfl_xor(bmax, d));
    }
    if (valid && j == 0) {
        A.nl[b] = ok ? n : 0;          // 0 => BADMODEL: K1/K2 write zeros
        if (A.fsafe) A.fsafe[b] = (ok && mono) ? hmax : 1.0e30f;
        if (A.nsolved_init) A.nsolved_init[b] = ok ? A.P : 0;
        if (A.ovf) {
            A.ovf[b] = hthick;
            A.ovf[(size_t)B + b] = 2.0f * logf(fmaxf(rhomax, 1.0e-30f));
            A.ovf[2 * (size_t)B + b] = 4.0f * logf(fmaxf(2.0f * bmax * bmax, 1.0e-30f));
        }
        if (A.fb_count && b == 0) { A.fb_count[0] = 0; A.fb_count[1] = 0; A.fb_count[2] = 0; }
    }
}

// per-period, per-layer working values (calcul.f:112-131 then flat1 with n_flat layers)
struct LayerV { float a, b, rho, d; };

struct LayerRaw { float a_ref, b_ref, rho_ref, qs, dif, qqq, dfl, hsf, hsr; };

SD_HD __forceinline__ LayerRaw layer_load(const float *__restrict__ mdl, size_t fs, size_t o)
{
    LayerRaw r;
    r.a_ref = mdl[F_VP * fs + o];  r.b_ref = mdl[F_VS * fs + o];  r.rho_ref = mdl[F_RHO * fs + o];
    r.qs = mdl[F_QS * fs + o];     r.dif = mdl[F_DIF * fs + o];   r.qqq = mdl[F_QQQ * fs + o];
    r.dfl = mdl[F_DFL * fs + o];   r.hsf = mdl[F_HSF * fs + o];   r.hsr = mdl[F_HSR * fs + o];
    return r;
}

SD_HD __forceinline__ LayerV layer_derive(const LayerRaw &r, float lnT, bool is_halfspace)
{
#pragma clang fp contract(off)   // bit-identical at every call site (K2 integrates twice)
    const float qsq = r.qs * lnT / PI_REF;                                        // calcul.f:122
    const float qpq = qsq * 1.33333333f * (r.b_ref * r.b_ref) / (r.a_ref * r.a_ref); // calcul.f:123
    const float bb = r.b_ref * (1.0f + qsq);
    const float aa = r.a_ref * (1.0f + qpq);
    LayerV v;
    if (!is_halfspace) {
        v.a = aa * r.dif; v.b = bb * r.dif;
        v.rho = r.rho_ref * r.qqq;
        v.d = r.dfl;
    } else {
        v.a = aa * r.hsf; v.b = bb * r.hsf;
        v.rho = r.rho_ref * r.hsr;
        v.d = 0.0f;
    }
    return v;
}

SD_HD __forceinline__ LayerV layer_at(const float *__restrict__ mdl, size_t fs, size_t o,
                                           float lnT, bool is_halfspace)
{
    return layer_derive(layer_load(mdl, fs, o), lnT, is_halfspace);
}

// ================================================================ secular functions (registers)
// LDS working stack of one team: w[m*LS + f*S + slot], f = 0..5 (derived slot W_IR, b, rho, d, 1/a^2, 1/b^2):
// the six values of a layer sit at compile-time offsets f*S from one address (ds_read immediates)
constexpr int NFW = 6;
// Layer stride LS (words) for S = SD_PHASE_BLOCK / G slots: NFW * S padded so that the lanes of a team - which take
// consecutive LAYERS when they (re)build the stack, snapshot it or sum its thicknesses - fall into different LDS banks
// together with the neighbouring teams of their 32-lane group.  NFW * S is a multiple of 32 words for every team size
// up to 16 lanes: unpadded, all lanes of a team hit ONE bank (16-way conflicts on each of the 36 writes of a rebuild;
// SQ_LDS_BANK_CONFLICT was 29-55 % of the LDS cycles, profiles/r03a).  Wanted: lane j, neighbouring slot t -> bank
// (j * LS + t) mod 32 all different, i.e. LS = q * odd (mod 32) with q = 32 / G slots per group (1 for G >= 32).
// Love needs four of the six fields (1/(rho b^2), b, rho, d: 1/b^2 = 1/(rho b^2) x rho is one multiplication in the
// recursion): its working stacks take two thirds of the LDS - what decides how many Love workgroups fit beside the
// Rayleigh ones of a joint solve.
constexpr int NFW_LOVE = 4;
SD_HD constexpr int lds_ls(int S, int NF = NFW)
{
    const int G = SD_PHASE_BLOCK / S;
    const int q = (G >= 32) ? 1 : 32 / G;
    int ls = NF * S;
    while (ls % (2 * q) != q) ++ls;
    return ls;
}
// LS: the layer stride of the working stack at hand (a variable or parameter of the code that uses the macros)
#define W_AT(m, f) wq[(m) * LS + (f) * S]
#define W_IR(m) W_AT(m, 0)    // Rayleigh: layer 0 1/rho, layer m >= 1 rho(m-1)/rho(m) (rescale factor of the carried state); Love: 1/(rho b^2)
#define W_B(m) W_AT(m, 1)
#define W_R(m) W_AT(m, 2)
#define W_D(m) W_AT(m, 3)
#define W_IA2(m) W_AT(m, 4)   // 1/a^2  (c-independent; saves a division per layer per trial)
#define W_IB2(m) W_AT(m, 5)   // 1/b^2  (0 for a liquid layer)

// ---- fast-but-tight fp32 helpers for the inner recursion --------------------------------------
// The reference evaluates ~9 IEEE divisions, 2 sqrt and 2-4 libm calls per layer per trial
// velocity.  Here: reciprocal = v_rcp_f32 + one Newton step (<= 1 ulp), sqrt = v_sqrt_f32 (1 ulp),
// sinh/cosh = two v_exp_f32 (~1 ulp), sincos = 3-constant Cody-Waite reduction + minimax
// polynomials (~1 ulp for |x| < 1e4).  Every one of these perturbs a matrix entry by ~1e-7
// relative, i.e. like a 1e-7 relative change of a layer's thickness or velocity -- physically
// nothing; measured end-to-end parity is unchanged.
__device__ __forceinline__ float rcp_nr(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }
// sinh(x), cosh(x): 0.5 e^x and 0.5 e^-x straight from v_exp_f32 (the 0.5 folded into the exponent)
__device__ __forceinline__ void sinhcosh_sp(float x, float *sh, float *ch)
{
    // no low-order correction of x log2(e): e^x and e^-x then carry relative errors +-|x| 9e-8, which for |x| > 1 is a
    // common scale factor of sinh and cosh (the secular function's root does not move) and for |x| < 1 is below one
    // ulp anyway - golden-case parity unchanged, root search 6 % faster (profiles/r02d/ab_recursion_variants.txt).
    // Tried and dropped (r02e): the pair scaled by e^-|x| from ONE exponential ((1 +- e^-2|x|)/2; a transcendental costs
    // four plain instructions).  The secular function then comes out times s(c) = exp(-sum |x|): same roots and signs,
    // but s has a square-root kink wherever c crosses a layer velocity and bends the function between them, and the
    // refine step interpolates - the unscaled function is what the reference's NEVILL sees.  Measured: 4.7 % faster, c
    // against the exact kernel 5.4e-6 -> 1.2e-5, one sediment fixture entry 5.5e-5; made safe (kink test, unscaled
    // values for NEVILL, overflow cue from the product of the factors) the gain was gone (profiles/r02e/ab_experiments.txt).
    const float t = x * 1.44269502e+00f;
    const float p = __builtin_amdgcn_exp2f(t - 1.0f), q = __builtin_amdgcn_exp2f(-t - 1.0f);
    *sh = p - q;
    *ch = p + q;
}
__device__ __forceinline__ void sincos_cw(float x, float *sn, float *cs)
{
    const float TWO_OVER_PI = 6.36619747e-01f;
    const float P1 = 1.57079601e+00f, P2 = 3.13916473e-07f, P3 = 5.39030253e-15f;   // pi/2 split (Cody-Waite)
    const float q = __builtin_rintf(x * TWO_OVER_PI);
    float r = fmaf(-q, P1, x);
    r = fmaf(-q, P2, r);
    r = fmaf(-q, P3, r);
    const int n = (int)q;
    const float r2 = r * r;
    // minimax on [-pi/4, pi/4] (Cephes sinf/cosf coefficients)
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                    r2 * r2, fmaf(-0.5f, r2, 1.0f));
    const float s0 = (n & 1) ? pc : ps;
    const float c0 = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}

// Rayleigh, production kernel: Dunkin's compound-matrix recursion (surfa.f:193-357) in a form built for the VALU.
// start = 1 -> dispersion (returns -bb1, surfa.f:357); start = 2 / 3 -> the two ellipticity passes (bb1, surfa.f:360-363).
//
// The layer's compound matrix is never formed.  Its fifteen entries (surfa.f:289-320) are linear in the nine products
// (cosp, rsinp, sinpr) x (cosq, rsinq, sinqr); collecting the update by product instead of by entry gives, exactly
// (algebraic identity, checked in fp64 to 3e-14; same fp32 error against an fp64 evaluation as the entry-by-entry form),
//   h = (b2, b3, b4)/rhoc, h5 = b5/rhoc^2,  u1 = g^2 b1 + 2g h3 - h5,  u2 = g1^2 b1 + 2g1 h3 - h5,
//   E1 = rsinp (rsinq u1 + cosq h2) - cosp rsinq h4 + (1 - cosp cosq) u2,
//   E2 = sinpr (sinqr u2 - cosq h4) + cosp sinqr h2 + (1 - cosp cosq) u1,
//   b1' = b1 - E1 - E2,  h3' = h3 + g E1 + g1 E2,  h5' = h5 + g^2 E1 + g1^2 E2,
//   h2' = cosp (cosq h2 + rsinq u1) + sinpr (rsinq h4 + cosq u2),
//   h4' = rsinp (sinqr h2 - cosq u1) - cosp (sinqr u2 - cosq h4).
// The state is carried as (b1, h2..h5): going from layer m to m+1 only rescales it by rho_m/rho_(m+1) (rhoc = rho c^2
// and c is the same), and the half-space row is applied to rhoc h.  ~40 operations per layer instead of ~100.
// Vertical wavenumbers ra, rb and their reciprocals come from ONE v_rsq_f32 each; sinh/cosh from two v_exp_f32.
// (The reference's own arithmetic, statement by statement, is delta_rayleigh_ref below: the exact fallback kernel.)
// One trial velocity's constants, the carried state and a layer's five values (rat: rho(m-1)/rho(m), formed when the
// working stack is built).  ray_step / ray_close are shared by the root search (delta_rayleigh: layer values from the LDS
// working stack) and by the ellipticity kernel (surfdisp_ellip_kernel: layer values rebuilt in registers) - the same
// expressions, so a (c, stack state) pair gives the same value whichever kernel evaluates it.
struct RTrial { float wvno, csq, icsq; };
struct RState { float b1, h2, h3, h4, h5; };
struct RLyr { float sv, d, ia2, ib2, rat; };
__device__ __forceinline__ RTrial ray_trial(const float c, const float T)
{
    RTrial t;
    t.wvno = 6.28318531f * rcp_nr(c * T);                 // <= 1 ulp: like a 6e-8 change of the period
    t.csq = c * c;
    t.icsq = rcp_nr(t.csq);
    return t;
}
// start vector (1,0,0,0,0) / e2 / e3 in the scale of the top layer (irho0 = 1 / rho of layer 0)
__device__ __forceinline__ RState ray_start(const RTrial &t, const int start, const float irho0)
{
    const float irhoc0 = (start == 1) ? 0.0f : irho0 * t.icsq;
    return RState{(start == 1) ? 1.0f : 0.0f, (start == 2) ? irhoc0 : 0.0f, (start == 3) ? irhoc0 : 0.0f, 0.0f, 0.0f};
}
// FIRST: only the TOP layer may be liquid in the production kernel (a stack with a liquid layer further down is handed
// to the exact fallback kernel by the prep kernel's statistics), so the test is made once per evaluation
// CERT (r04, the COUNT-GUIDED coarse scan for Rayleigh - the opt-in SURFDISP_FASTSCAN; the default scan walks every grid point):
// the state's first component b1 is det U_s of the two solutions that satisfy the free-surface condition (start: U = I, tractions
// 0).  At a trial (k, omega) the number of mode branches below it is
//      N = #{zeros of det U_s(z) over the stack} + #{positive eigenvalues of Z_h - Z_s at the top of the half space}
// (Morse index of the Neumann problem + the half space's boundary index; Z = T U^-1 with the tractions ordered (tr, tz), which
// makes the motion-stress system Hamiltonian and Z symmetric).  The zeros are counted LAYER BY LAYER, inside each layer, by
// Wittrick and Williams' rule (below) - exact in exact arithmetic while every oscillatory layer's S phase stays below pi
// (scripts/analysis/rayleigh_count_ww*.py: equal to the brute-force count on every such trial of 50 random stacks).  The first
// attempt (SD_RCERT == 2) looked at the sign of b1 at the interfaces only and missed pairs of zeros inside one layer
// (profiles/r04a/rayleigh_count.txt: 25 of 1.3e8 soak stacks on another root).
// What the count is NOT: a proof that an interval between two trials with equal counts holds no root.  Along the scan's line
// omega = const the count rises where the line crosses a branch whose group velocity is positive and FALLS where it is negative:
// a branch with a zero-group-velocity point (soft sediments with Vp/Vs near 3 over rock) can be crossed twice between two coarse
// points, +1 then -1.  Love branches cannot (their group velocity is an energy ratio of one sign; the Love certificate is a
// theorem); Rayleigh ones can, and two soak stacks in 1.2e9 did (profiles/r04b/rayleigh_count_ww.txt).  Hence opt-in.
// kc accumulates the count, kunc flags a trial whose count is not safe (a sign within rounding, an S phase beyond the bound, a
// liquid layer).
#ifndef SD_RCERT_PHASE
#define SD_RCERT_PHASE 3.0f
#endif
#ifndef SD_RCERT_SPHASE
#define SD_RCERT_SPHASE 3.0f    // S phase (rad) of an oscillatory layer up to which its in-layer count is taken (the theorem's bound: pi)
#endif
#ifndef SD_RCERT
#define SD_RCERT 1              // 1: the Rayleigh FAST instantiations (opt-in, SURFDISP_FASTSCAN) are the count-guided scan; 0: the r01
#endif                          // heuristic scan; 2: the first attempt (interface-only count; profiles/r04a/rayleigh_count.txt)
template <bool FIRST, bool CERT = false>
__device__ __forceinline__ void ray_step(RState &s, const RTrial &t, const RLyr &y, const int start, float &phi, int *kc = nullptr,
                                         bool *kunc = nullptr)
{
    float b1 = s.b1, h2 = s.h2, h3 = s.h3, h4 = s.h4, h5 = s.h5;
    const float wvno = t.wvno, csq = t.csq, icsq = t.icsq;
    const float sv = y.sv, d = y.d, ia2 = y.ia2, ib2 = y.ib2;
    if (!FIRST) {                                      // into this layer's scale: rhoc_prev / rhoc = rho_prev / rho
        const float rat = y.rat;
        h2 *= rat; h3 *= rat; h4 *= rat; h5 *= rat * rat;
    }
    const float arga = fmaf(-csq, ia2, 1.0f);                    // 1 - c^2/a^2, surfa.f:211
    // |arga| is clamped away from zero: c == a to the last bit then runs through the oscillatory formulas with
    // ra = 1e-15, which give the reference's degenerate values (rsinp = 0, sinpr = k d, cosp = 1; surfa.f:263-266)
    // to 1e-15 - no separate branch
    const float xa = fmaxf(fabsf(arga), 1.0e-30f), ya = __builtin_amdgcn_rsqf(xa);
    const float ra = copysignf(xa * ya, -arga), ira = copysignf(ya, -arga);   // ra = x rsq(x) to ~1.5 ulp; < 0: evanescent
    const float wd = wvno * d;
    if (FIRST && !(fabsf(sv) > ACCUR)) {
        // liquid surface layer, surfa.f:216-251 (skipped entirely in the ellipticity passes): only a11 = cosp and
        // a21 = rhoc sinpr are non-zero (surfa.f:236-250)
        if (start != 1) return;
        if (CERT) *kunc = true;                            // (a liquid layer: no certificate)
        const float pm = wd * ra;
        float sinpr, cosp;
        if (fabsf(ra) < ACCUR) { sinpr = wd; cosp = 1.0f; }
        else if (ra < 0.0f) {
            float sh, ch; sinhcosh_sp(pm, &sh, &ch);
            sinpr = sh / ra;
            cosp = ch;
        } else {
            float sn, cs; sincos_cw(pm, &sn, &cs);
            sinpr = sn / ra; cosp = cs;
            phi += pm;
        }
        const float n1 = cosp * b1;
        const float n2 = sinpr * b1;
        const float n5 = cosp * h5 - sinpr * h4;
        s.b1 = n1; s.h2 = n2; s.h3 = 0.0f; s.h4 = 0.0f; s.h5 = n5;
        return;
    }
    const float argb = fmaf(-csq, ib2, 1.0f);
    const float xb = fmaxf(fabsf(argb), 1.0e-30f), yb = __builtin_amdgcn_rsqf(xb);    // same for c == b (surfa.f:275-279)
    const float rb = copysignf(xb * yb, -argb), irb = copysignf(yb, -argb);
    const float g = 2.0f * (sv * sv) * icsq;
    const float g1 = g - 1.0f;
    const float pm = wd * ra;
    const float qm = wd * rb;
    float rsinp, sinpr, cosp, rsinq, sinqr, cosq;
    if (arga > 0.0f) {                                 // evanescent P (ra < 0), surfa.f:267-269
        float sh, ch; sinhcosh_sp(pm, &sh, &ch);
        rsinp = -ra * sh; sinpr = sh * ira; cosp = ch;
    } else {                                           // oscillatory P, surfa.f:271-273
        float sn, cs; sincos_cw(pm, &sn, &cs);
        rsinp = ra * sn; sinpr = sn * ira; cosp = cs;
        phi += pm;
    }
    if (!(argb > 0.0f)) {
        float sn, cs; sincos_cw(qm, &sn, &cs);
        rsinq = rb * sn; sinqr = sn * irb; cosq = cs;
        phi += qm;
    } else {
        float sh, ch; sinhcosh_sp(qm, &sh, &ch);
        rsinq = -rb * sh; sinqr = sh * irb; cosq = ch;
    }
    const float g2 = g * g, g12 = g1 * g1;
    const float u1 = fmaf(g2, b1, fmaf(g + g, h3, -h5));
    const float u2 = fmaf(g12, b1, fmaf(g1 + g1, h3, -h5));
    const float D = fmaf(-cosp, cosq, 1.0f);
    const float t1 = fmaf(rsinq, u1, cosq * h2);            // rsinq u1 + cosq h2
    const float t2 = fmaf(sinqr, u2, -(cosq * h4));         // sinqr u2 - cosq h4
    const float Cx = cosp * rsinq, Cy = cosp * sinqr;
    const float E1 = fmaf(rsinp, t1, fmaf(-Cx, h4, D * u2));
    const float E2 = fmaf(sinpr, t2, fmaf(Cy, h2, D * u1));
    const float n1 = (b1 - E1) - E2;
    if (CERT) {
#if SD_RCERT == 2
        // (first attempt, kept for the record: sign changes of b1 at the interfaces only - misses pairs of zeros inside a layer)
        *kc += ((n1 < 0.0f) != (b1 < 0.0f)) ? 1 : 0;
        *kunc = *kunc || !(fabsf(pm) + fabsf(qm) < SD_RCERT_PHASE) ||
                !(fabsf(n1) > 1.0e-4f * (fabsf(b1) + fabsf(E1) + fabsf(E2)));
#else
        // Zeros of det U_s INSIDE this layer (Wittrick-Williams).  With the surface pair's impedance Z_t at the layer's top and the
        // layer's propagator in blocks, det U_s at depth z below the top vanishes where M(z) = Z_t - K11(z) is singular, K11 = the
        // impedance of the slab clamped at z seen from its top; M(0+) is positive definite and - as long as no clamped-clamped
        // mode of the slab lies below the frequency, which Korn's inequality guarantees while the S phase k d r_beta < pi (and
        // always where S is evanescent) - its eigenvalues only move down: the zeros inside the layer = the negative eigenvalues
        // of M(d), 0, 1 or 2.  n1 is linear in the state, n1 = c1 b1 + c2 h2 + c3 h3 + c4 h4 + c5 h5, and for a state that is the
        // minors of [I; Z] (b1 = 1, h2 = -z22, h3 = -z12, h4 = z11, h5 = det Z) that is c5 det(Z - K11): K11's entries are ratios
        // of the coefficients, and
        //      det M = n1 / (b1 c5),      M11 = (h4 c5 - c2 b1) / (b1 c5),
        //      c5 = rsinp rsinq + sinpr sinqr + 2 (1 - cosp cosq),   c2 = -(rsinp cosq + cosp sinqr)
        // (scripts/analysis/rayleigh_count_ww*.py: per layer and in total equal to the finely stepped / brute-force counts on
        // every trial with all S phases below pi).  Unsafe: an oscillatory layer beyond SD_RCERT_SPHASE, or any of the three
        // signs within rounding (n1, c5, and - where it decides - M11's numerator against their own terms).
        const float p1 = rsinp * rsinq, p2 = sinpr * sinqr;
        const float c5 = p1 + p2 + (D + D);
        const float c2 = -fmaf(rsinp, cosq, Cy);
        const float bc = b1 * c5;
        const float m11a = h4 * c5, m11b = c2 * b1;
        const float m11n = m11a - m11b;
        const bool dneg = (n1 < 0.0f) != (bc < 0.0f);
        const bool mneg = (m11n < 0.0f) != (bc < 0.0f);
        *kc += dneg ? 1 : (mneg ? 2 : 0);
        *kunc = *kunc || (!(argb > 0.0f) && !(fabsf(qm) < SD_RCERT_SPHASE)) ||
                !(fabsf(n1) > 1.0e-4f * (fabsf(b1) + fabsf(E1) + fabsf(E2))) ||
                !(fabsf(c5) > 1.0e-4f * (fabsf(p1) + fabsf(p2) + 2.0f * fabsf(D))) ||
                (!dneg && !(fabsf(m11n) > 1.0e-4f * (fabsf(m11a) + fabsf(m11b)))) || !(fabsf(bc) > 0.0f);
#ifdef SD_DEBUG_COUNT
        if (csq > SD_DEBUG_COUNT * SD_DEBUG_COUNT && csq < (SD_DEBUG_COUNT + 0.07f) * (SD_DEBUG_COUNT + 0.07f))
            printf("   layer c %.6f: b1 % .5e n1 % .5e c5 % .5e (p1 % .3e p2 % .3e 2D % .3e) m11n % .5e (%.3e, %.3e) det<0 %d m11<0 %d count %d qm %.3f pm %.3f unsafe %d\n",
                   sqrtf(csq), b1, n1, c5, p1, p2, D + D, m11n, m11a, m11b, (int)dneg, (int)mneg, *kc, qm, pm, (int)*kunc);
#endif
#endif
    }
    const float n3 = fmaf(g, E1, fmaf(g1, E2, h3));
    const float n5 = fmaf(g2, E1, fmaf(g12, E2, h5));
    const float n2 = fmaf(cosp, t1, sinpr * fmaf(rsinq, h4, cosq * u2));
    const float n4 = fmaf(rsinp, fmaf(sinqr, h2, -(cosq * u1)), -(cosp * t2));
    s.b1 = n1; s.h2 = n2; s.h3 = n3; s.h4 = n4; s.h5 = n5;
}
// half-space closure, surfa.f:340-354, on (b1, rhoc h2..h4, rhoc^2 h5) with rhoc of the last layer gone through
// (a itself is not needed: every occurrence is a^2, available as 1/ia2); A = the values of layer mmax - 1, rho_last its
// density, rho_prev the density of layer mmax - 2 (0 when there is none: the state is in that layer's scale)
// *mag (if given): the sum of the magnitudes of the closure's five terms - |value| far below it means the value is the
// remainder of a cancellation, i.e. its SIGN is within the rounding of this arithmetic (the scan's ambiguity test)
__device__ __forceinline__ float ray_close(const RState &s, const RTrial &t, const RLyr &A, const float rho_last,
                                           const float rho_prev, const int start, float *mag = nullptr, const bool want_mag = true,
                                           int *kc = nullptr, bool *kunc = nullptr)
{
    const float csq = t.csq, icsq = t.icsq;
    const float sv = A.sv, ia2 = A.ia2;
    const float irho = rcp_nr(rho_last);
    const float rhoc = rho_prev * csq;
    const float arga = fmaf(-csq, ia2, 1.0f);
    float ra = sqrt_hw(fabsf(arga));
    if (arga > 0.0f) ra = -ra;
    const float argb = fmaf(-csq, A.ib2, 1.0f);
    float rb = sqrt_hw(fabsf(argb));
    if (argb > 0.0f) rb = -rb;
    const float sss = sv * sv;
    const float g = 2.0f * sss * icsq;
    const float g1 = g - 1.0f;
    const float gra = g * ra, g1s = g1 * g1;
    const float ira = rcp_nr(ra), igra = rcp_nr(gra), ig = rcp_nr(g);
    const float it12 = ia2 * irho;                                   // 1/(rho a^2)
    const float rba = rb - ira;
    const float h11 = -2.0f * rb * sss * ia2 + csq * g1s * ia2 * igra;
    const float h13 = -rb * it12 + g1 * it12 * igra;
    const float h14 = rb * it12 * igra;
    const float h15 = rba * (irho * irho) * ia2 * icsq * ig;         // rba/(rho a)^2/c^2/g
    const float h12 = -ig * it12;
    const float bb1 = h11 * s.b1 + rhoc * (h12 * s.h2 + 2.0f * h13 * s.h3 + h14 * s.h4 + rhoc * (h15 * s.h5));
    if (kc) {
        // boundary index at the top of the half space: positive eigenvalues of S = Z_h - Z_s (both divided by k > 0),
        //   Z_s = (rhoc / b1) [[h4, -h3], [-h3, -h2]]   (b2..b4 = -minors / k: rhoc h2..h4 of this state),
        //   Z_h = rho c^2 / (1 - ra rb) [[-ra, g1 - g ra rb], [g1 - g ra rb, -rb]]   (ra, rb > 0: the half space's decaying pair)
        const float rap = fabsf(ra), rbp = fabsf(rb);
        const float zf = rho_last * csq * rcp_nr(1.0f - rap * rbp), zo = g1 - g * rap * rbp;
        const float sf = rhoc * rcp_nr(s.b1);
        const float S11 = -zf * rap - sf * s.h4, S12 = zf * zo + sf * s.h3, S22 = -zf * rbp + sf * s.h2;
        const float dq = S11 * S22 - S12 * S12, tp = S11 + S22;
        *kc += (dq < 0.0f) ? 1 : ((tp > 0.0f) ? 2 : 0);
#ifdef SD_DEBUG_COUNT
        if (csq > SD_DEBUG_COUNT * SD_DEBUG_COUNT && csq < (SD_DEBUG_COUNT + 0.07f) * (SD_DEBUG_COUNT + 0.07f))
            printf("   half space c %.6f: S11 % .5e (%.3e - %.3e) S12 % .5e S22 % .5e (%.3e + %.3e) det % .5e trace % .5e count %d\n",
                   sqrtf(csq), S11, -zf * rap, sf * s.h4, S12, S22, -zf * rbp, sf * s.h2, dq, tp, *kc);
#endif
        // unsafe: the half space not evanescent in P and S, a determinant or - where it matters - a trace within rounding, not finite
        *kunc = *kunc || !(arga > 0.0f) || !(argb > 0.0f) || !(fabsf(dq) > 1.0e-4f * (fabsf(S11 * S22) + S12 * S12)) ||
                (dq > 0.0f && !(fabsf(tp) > 1.0e-4f * (fabsf(S11) + fabsf(S22)))) || !fin(dq) || !fin(s.b1) || s.b1 == 0.0f;
    }
    if (mag && want_mag) *mag = fabsf(h11 * s.b1) + fabsf(rhoc) * (fabsf(h12 * s.h2) + 2.0f * fabsf(h13 * s.h3) + fabsf(h14 * s.h4) +
                                                        fabsf(rhoc * (h15 * s.h5)));
    return (start == 1) ? -bb1 : bb1;
}

template <bool PIPE2 = true, bool CERT = false>
__device__ __forceinline__ float delta_rayleigh(const float *wq, const int LS, const int S,
                                                const int mmax, const float c, const float T,
                                                const int start, float &phi, float *mag = nullptr, const bool want_mag = true,
                                                int *kcp = nullptr, bool *kuncp = nullptr, const bool count = false)
{
    int kc_ = 0; bool kunc_ = !count;
    int *const kc = &kc_; bool *const kunc = &kunc_;
    // phi: vertical phase sum_i k d_i sqrt(c^2/v_i^2 - 1) over the layers (and wave types) that are oscillatory
    // at c -- the WKB mode counter the opt-in fast scan bounds between two coarse points (free: pm and qm are
    // the recursion's own arguments)
    phi = 0.0f;
    const RTrial t = ray_trial(c, T);
    RState s = ray_start(t, start, (start == 1) ? 0.0f : W_IR(0));
    // software pipeline: layer m+1's LDS values are in flight while layer m is computed.  The loop
    // is unrolled by two over alternating register sets (no rotation moves between iterations).
    auto load = [&](int m) -> RLyr { return {W_B(m), W_D(m), W_IA2(m), W_IB2(m), W_IR(m)}; };
    const int last = mmax - 1;                                       // the half space
    RLyr A = load(0);
    int m = 0;
    if (last >= 1) {                                                 // layer 0: the one that may be water
        const RLyr Bq = load(1);
        if (CERT && count) ray_step<true, true>(s, t, A, start, phi, kc, kunc); else ray_step<true>(s, t, A, start, phi);
        A = Bq;
        m = 1;
    }
    if (PIPE2) {
        while (m + 2 <= last) {
            const RLyr Bq = load(m + 1);
            if (CERT && count) ray_step<false, true>(s, t, A, start, phi, kc, kunc); else ray_step<false>(s, t, A, start, phi);
            A = load(m + 2);
            if (CERT && count) ray_step<false, true>(s, t, Bq, start, phi, kc, kunc); else ray_step<false>(s, t, Bq, start, phi);
            m += 2;
        }
    } else {
        // (also the opt-in fast scan's instantiations: their extra scan state put teams of four at 133 VGPRs = three
        // wavefronts per SIMD, 27 M solves/s with one batch in flight; single-buffered 122 = four, 36 M.)
        // Teams of two lanes - what a caller with several batches in flight gets (SURFDISP_PIPELINED) - keep ONE register
        // set in flight: their wavefronts share SIMDs with the group-velocity kernel's (168 VGPRs), and three of them
        // fit beside one of those only while 3 x VGPRs + 168 <= 512 (101 this way; measured at 120: the three-batch
        // headline drops 7 %, profiles/r02e/ab_experiments.txt)
        while (m + 1 <= last) {
            const RLyr Bq = load(m + 1);
            if (CERT && count) ray_step<false, true>(s, t, A, start, phi, kc, kunc); else ray_step<false>(s, t, A, start, phi);
            A = Bq;
            m += 1;
        }
    }
    if (m < last) {
        const RLyr Bq = load(m + 1);
        if (CERT && count) ray_step<false, true>(s, t, A, start, phi, kc, kunc); else ray_step<false>(s, t, A, start, phi);
        A = Bq;
    }
    // A holds layer mmax-1 here; the state is in the scale of the last layer stepped through
    const float v = ray_close(s, t, A, W_R(last), last >= 1 ? W_R(last - 1) : 0.0f, start, mag, want_mag,
                              (CERT && count) ? kc : nullptr, kunc);
    if (CERT) { *kcp = kc_; *kuncp = kunc_ || (last < 1); }
    return v;
}

// The exact fallback kernel's Rayleigh secular function: DLTAR4 restated statement by statement (surfa.f:193-357),
// IEEE division and square root, libm-grade exp / sin / cos, no fused multiply-adds - so that every intermediate
// overflows to inf, and every inf - inf turns NaN, exactly where the reference's does (which decides the "roots" the
// reference returns next to the overflowed region).  Working stack of that kernel: a in the W_IA2 slot.
// start = 1 -> dispersion (-bb1); 2 / 3 -> the two ellipticity passes (bb1), combined by the caller (surfa.f:360-363).
// AINV: called on the PRODUCTION kernel's working stack, whose W_IA2 slot holds 1/a^2 (to 1 ulp) instead of a: a is taken
// as 1/sqrt of it (IEEE) - the reference's arithmetic on a P velocity that may differ from its own in the last bit, which
// moves the value ten times less than the production recursion's rounding does (the scan's ambiguity re-evaluation).
// (The production kernel inlines the body: a call would cost its wavefronts a fourth of their registers - the calling
// convention's - and with them the fourth wavefront per SIMD.)
struct RefLyr { float a, b, rho, d; };
// GET: m -> the working stack's values of layer m (called for m = 0 .. mmax - 1, once each, in order)
template <class GET>
__device__ __forceinline__ float delta_rayleigh_ref_gen(GET get, const int mmax, const float c, const float t, const int start)
{
#pragma clang fp contract(off)
    const float accur = 1.e-8f, accurs = 1.e-8f;
    const float wvno = 6.28318531f / (c * t);
    const float csq = c * c;
    float b1 = (start == 1) ? 1.0f : 0.0f, b2 = (start == 2) ? 1.0f : 0.0f, b3 = (start == 3) ? 1.0f : 0.0f,
          b4 = 0.0f, b5 = 0.0f;
    float ra = 0.0f, rb = 0.0f, g = 0.0f, g1 = 0.0f;
    RefLyr yl{1.0f, 1.0f, 1.0f, 0.0f};
    for (int m = 0; m < mmax; ++m) {
        yl = get(m);
        const float pm_ = yl.a, sm = yl.b, rho = yl.rho, d = yl.d;             // p(m), s(m), rho(m), d(m)
        const float arga = 1.0f - csq / (pm_ * pm_);
        ra = sqrtf(fabsf(arga));
        if (arga > 0.0f) ra = -ra;
        float a11, a12, a13, a14, a15, a21, a22, a23, a24, a31, a32, a33, a41, a42, a51;
        if (!(fabsf(sm) > accurs)) {                       // liquid surface layer, surfa.f:216-251
            const float pm = wvno * ra * d;
            if (start > 1) continue;
            const float rhoc = rho * csq;
            float sinpr, cosp;
            if (fabsf(ra) < accur || ra == 0.0f) { sinpr = wvno * d; cosp = 1.0f; }
            else if (ra < 0.0f) { sinpr = (expf(pm) - expf(-pm)) / (2.0f * ra); cosp = 0.5f * (expf(pm) + expf(-pm)); }
            else { sinpr = sinf(pm) / ra; cosp = cosf(pm); }
            a11 = cosp; a21 = rhoc * sinpr;
            a31 = a41 = a51 = a12 = a22 = a32 = a42 = a13 = a23 = a33 = a14 = a24 = a15 = 0.0f;
        } else {
            const float argb = 1.0f - csq / (sm * sm);
            rb = sqrtf(fabsf(argb));
            if (argb > 0.0f) rb = -rb;
            g = 2.0f * (sm * sm) / csq;
            g1 = g - 1.0f;
            if (m == mmax - 1) break;                      // if(mmax-m) 40,52,40
            const float rhoc = rho * csq;
            const float pm = wvno * ra * d, qm = wvno * rb * d;
            float rsinp, sinpr, cosp, rsinq, sinqr, cosq;
            if (ra < 0.0f) { rsinp = -ra * 0.5f * (expf(pm) - expf(-pm)); sinpr = -rsinp / (ra * ra); cosp = 0.5f * (expf(pm) + expf(-pm)); }
            else if (ra == 0.0f) { rsinp = 0.0f; sinpr = wvno * d; cosp = 1.0f; }
            else { rsinp = ra * sinf(pm); sinpr = rsinp / (ra * ra); cosp = cosf(pm); }
            if (fabsf(rb) < accur) { rsinq = 0.0f; sinqr = wvno * d; cosq = 1.0f; }
            else if (rb > 0.0f) { rsinq = rb * sinf(qm); sinqr = rsinq / (rb * rb); cosq = cosf(qm); }
            else { rsinq = -rb * 0.5f * (expf(qm) - expf(-qm)); sinqr = -rsinq / (rb * rb); cosq = 0.5f * (expf(qm) + expf(-qm)); }
            const float rr = rsinp * rsinq, ss = sinpr * sinqr, cc = cosp * cosq;
            const float rs1 = rsinp * cosq, rs2 = sinqr * cosp, rs3 = sinpr * cosq, rs4 = rsinq * cosp;
            const float gm = 2.0f * g - 1.0f, gs = g * g, g1s = g1 * g1, ccm = 1.0f - cc, gg1 = g * g1;
            const float rhocs = rhoc * rhoc;
            const float suu = gs * rr + g1s * ss;
            a11 = 2.0f * gs - gm;
            a11 = a11 * cc - suu - 2.0f * gg1;
            a12 = -(rs1 + rs2) / rhoc;
            a13 = gm * ccm + g1 * ss + g * rr;
            a13 = -2.0f * a13 / rhoc;
            a14 = (rs3 + rs4) / rhoc;
            a15 = 2.0f * ccm + rr + ss;
            a15 = a15 / rhocs;
            a21 = rhoc * (g1s * rs3 + gs * rs4);
            a22 = cc;
            a23 = 2.0f * (g * rs4 + g1 * rs3);
            a24 = sinpr * rsinq;
            a31 = rhoc * (gg1 * gm * ccm + g1s * g1 * ss + gs * g * rr);
            a32 = g1 * rs2 + g * rs1;
            a33 = 1.0f + 2.0f * (2.0f * gg1 * ccm + suu);
            a41 = -rhoc * (g1s * rs2 + gs * rs1);
            a42 = rsinp * sinqr;
            a51 = rhocs * (2.0f * gs * g1s * ccm + gs * gs * rr + g1s * g1s * ss);
        }
        const float bb1 = a11 * b1 + a12 * b2 + a13 * b3 + a14 * b4 + a15 * b5;
        const float bb2 = a21 * b1 + a22 * b2 + a23 * b3 + a24 * b4 - a14 * b5;
        const float bb3 = a31 * b1 + a32 * b2 + a33 * b3 - 0.5f * a23 * b4 + 0.5f * a13 * b5;
        const float bb4 = a41 * b1 + a42 * b2 - 2.0f * a32 * b3 + a22 * b4 - a12 * b5;
        const float bb5 = a51 * b1 - a41 * b2 + 2.0f * a31 * b3 - a21 * b4 + a11 * b5;
        b1 = bb1; b2 = bb2; b3 = bb3; b4 = bb4; b5 = bb5;
    }
    // label 52: the half space (ra, rb, g, g1 of layer mmax from the last trip of the loop)
    const float pp = yl.a, sss = yl.b * yl.b, ppp = pp * pp;
    const float rhp = yl.rho * pp;
    const float gra = g * ra, g1s = g1 * g1, rba = rb - 1.0f / ra;
    const float h11 = -2.0f * rb * sss / ppp + csq * g1s / ppp / gra;
    float h12 = rhp * pp;
    const float h13 = -rb / h12 + g1 / h12 / gra;
    const float h14 = rb / h12 / gra;
    const float h15 = rba / rhp / rhp / csq / g;
    h12 = -1.0f / g / h12;
    const float bb1 = h11 * b1 + h12 * b2 + 2.0f * h13 * b3 + h14 * b4 + h15 * b5;
    return (start == 1) ? -bb1 : bb1;
}
template <bool AINV>
__device__ __forceinline__ float delta_rayleigh_ref_body(const float *wq, const int LS, const int S,
                                                         const int mmax, const float c, const float t, const int start)
{
    return delta_rayleigh_ref_gen([&](int m) { return RefLyr{AINV ? 1.0f / sqrtf(W_IA2(m)) : W_IA2(m), W_B(m), W_R(m), W_D(m)}; },
                                  mmax, c, t, start);
}
__device__ __noinline__ float delta_rayleigh_ref(const float *wq, const int LS, const int S,
                                                 const int mmax, const float c, const float t, const int start)
{
    return delta_rayleigh_ref_body<false>(wq, LS, S, mmax, c, t, start);
}

// Love: Thomson-Haskell 2-vector from the half space up, surfa.f:143-179 (production kernel; the reference's own
// arithmetic is delta_love_ref below).  Per layer: ONE v_rsq_f32 gives rb = sqrt|c^2/b^2 - 1| and 1/rb (|.| clamped away
// from zero: c == b then runs through the oscillatory formulas with rb = 1e-15, which give the reference's degenerate
// values y = -k d, z = 0, cosq = 1 of surfa.f:163-165 to 1e-15); 1/(rho b^2) comes from the working stack (the slot the
// Rayleigh recursion keeps its density ratios in).  Three transcendentals per evanescent layer instead of five.
// CERT: also the Sturm count of the trial - kc = net number of times the (displacement, stress) vector has crossed
// "stress = 0" clockwise on the way up from the half space (see the certified scan in phase_body), kunc = the count is not
// safe (a sign or a multiple of pi within rounding).  In a layer where c > b the pair (ut, tt / (h rb)) ROTATES by
// q = -k d rb: floor(|q| / pi) or one more crossings, the parity being whether tt changed sign; where c < b it moves along
// a hyperbola towards the diagonal and can cross at most once, counter-clockwise.
template <bool CERT = false>
__device__ __forceinline__ float delta_love(const float *wq, const int LS, const int S,
                                            const int mmax, const float c, const float T, float &phi, int &kc, bool &kunc,
                                            const bool count = true, float *mag = nullptr)
{
    float lmag = 0.0f;                                     // |h z ut| + |cosq tt| of the last step (see ray_close's mag)
    kc = 0; kunc = !count;                                 // (count = false: no certificate from this trial)
    phi = 0.0f;                                            // see delta_rayleigh
    const float wvno = 6.2831853f * rcp_nr(c * T);
    const float csq = c * c;
    const int mh = mmax - 1;
    float bm = W_B(mh);
    float h = W_R(mh) * bm * bm;
    float rb = sqrt_hw(fabsf(fmaf(csq, W_IR(mh) * W_R(mh), -1.0f)));   // sqrt|c^2/b^2 - 1|, 1/b^2 = 1/(rho b^2) x rho
    float ut = 1.0f, tt = h * rb;
    // layer m-1's five LDS values are in flight while layer m is computed; unrolled by two over
    // alternating register sets
    struct Lyr { float b, d, r, ih; };
    auto load = [&](int m) -> Lyr { const int q = m > 0 ? m : 0; return {W_B(q), W_D(q), W_R(q), W_IR(q)}; };
    auto step = [&](const Lyr &y) {
        bm = y.b;
        const float d = y.d, rho = y.r, ih = y.ih;
        if (bm == 0.0f) return;                            // water, surfa.f:152
        const float arg = fmaf(csq, ih * rho, -1.0f);      // c^2/b^2 - 1: < 0 evanescent
        const float x = fmaxf(fabsf(arg), 1.0e-30f), irb = __builtin_amdgcn_rsqf(x);
        rb = x * irb;
        h = rho * bm * bm;
        const float q = -wvno * d * rb;
        float yv, z, cosq;
        if (arg < 0.0f) {
            float sh, ch; sinhcosh_sp(q, &sh, &ch);
            yv = sh * irb;
            z = -rb * sh;                                  // -rb^2 y
            cosq = ch;
        } else {
            float sn, cs; sincos_cw(q, &sn, &cs);
            yv = sn * irb; z = rb * sn; cosq = cs;
        }
        phi -= (arg < 0.0f) ? 0.0f : q;
        const float eut = cosq * ut - yv * tt * ih;
        const float ett = h * z * ut + cosq * tt;
        lmag = fabsf(h * z * ut) + fabsf(cosq * tt);
        if (CERT && count) {
            const bool flip = (tt < 0.0f) != (ett < 0.0f);
            if (arg < 0.0f) {
                kc -= flip ? 1 : 0;
                // a pair that enters an evanescent layer close to its DECAYING direction (a mode trapped in a low-velocity
                // channel below a lid) leaves it as the difference of two e^{|q|} terms: below 1e-3 of them the sign of the
                // stress - here and in the point-by-point scan's secular function next to this trial - is rounding noise
                kunc = kunc || !(fabsf(ett) > 1.0e-3f * (fabsf(h * z * ut) + fabsf(cosq * tt)));
            } else {
                const float t = -q * 0.318309886f;         // |q| / pi
                const float nf = floorf(t), r = t - nf;
                const int nfull = (int)nf;
                kc += nfull + (((nfull & 1) != 0) != flip ? 1 : 0);
                kunc = kunc || (r > 1.0f - 4.0e-4f) || (nfull >= 1 && r < 4.0e-4f);      // |q| within ~1e-3 rad of a multiple of pi
            }
            kunc = kunc || !(fabsf(ett) > 1.0e-5f * (fabsf(ett) + fabsf(h * rb * eut)));  // stress ~ 0 at an interface (or not finite)
        }
        ut = eut; tt = ett;
    };
    int m = mh - 1;
    Lyr A = load(m);
    while (m >= 1) {
        const Lyr Bq = load(m - 1);
        step(A);
        A = load(m - 2);
        step(Bq);
        m -= 2;
    }
    if (m == 0) step(A);
    if (mag) *mag = lmag;
    return -tt;
}

// ... and its Love secular function: DLTAR1 statement by statement (surfa.f:143-179), same rules.
__device__ __forceinline__ float delta_love_ref_body(const float *wq, const int LS, const int S,
                                                     const int mmax, const float c, const float t)
{
#pragma clang fp contract(off)
    const float wvno = 6.2831853f / (c * t);
    float covb = c / W_B(mmax - 1);
    float h = W_R(mmax - 1) * W_B(mmax - 1) * W_B(mmax - 1);
    float rb = sqrtf(fabsf(covb * covb - 1.0f));
    float ut = 1.0f, tt = h * rb;
    for (int m = mmax - 2; m >= 0; --m) {
        const float bm = W_B(m);
        if (bm == 0.0f) continue;
        covb = c / bm;
        rb = sqrtf(fabsf(covb * covb - 1.0f));
        h = W_R(m) * bm * bm;
        const float d = W_D(m);
        const float q = -wvno * d * rb;
        float y, z, cosq;
        if (rb < 0.1e-20f || c - bm == 0.0f) { y = -wvno * d; z = 0.0f; cosq = 1.0f; }
        else if (c - bm < 0.0f) {
            const float exqp = expf(q), exqm = 1.0f / exqp;
            y = (exqp - exqm) / (2.0f * rb);
            z = -rb * rb * y;
            cosq = (exqp + exqm) / 2.0f;
        } else {
            const float sinq = sinf(q);
            y = sinq / rb; z = rb * sinq; cosq = cosf(q);
        }
        const float eut = cosq * ut - y * tt / h;
        const float ett = h * z * ut + cosq * tt;
        ut = eut; tt = ett;
    }
    return -tt;
}
__device__ __noinline__ float delta_love_ref(const float *wq, const int LS, const int S,
                                             const int mmax, const float c, const float t)
{
    return delta_love_ref_body(wq, LS, S, mmax, c, t);
}

// layer dropping for one trial velocity, surfa.f:94-105.  No early exit: every load is independent
// of the running sum, so the LDS reads pipeline instead of costing one round trip per layer
// (adding 0.0f for a skipped layer is exact, and nothing after the first crossing can change mm).
__device__ __forceinline__ int drop_layers(const float *wq, const int LS, const int S,
                                           const int n, const float c, const float T)
{
    // mmax = ii + 1 at the first layer ii whose running sum exceeds dmax.  The thicknesses are >= 0, so the running sums
    // never decrease and that layer is found by COUNTING the sums that do not exceed dmax: no "found" flag, no select
    // per layer (five vector instructions per layer and none on the scalar unit; the flag version took six + five)
    const float dmax = FACT * c * T;
    int cnt = 0;
    float sum = 0.0f;
#pragma unroll 8
    for (int ii = 0; ii < n; ++ii) {
        const float bi = W_B(ii), di = W_D(ii);
        sum = sum + ((c < bi) ? di : 0.0f);
        cnt += (sum > dmax) ? 0 : 1;
    }
    const int mm = (cnt < n) ? cnt + 1 : n;
    return mm < 2 ? 2 : mm;
}

// ================================================================================== K1: phase
// ST_WREF / ST_WEND (lock step, see PhaseArgs::lockstep): a team that has its bracket / its root waits for the other teams
// of its wavefront, so that the refine pass and the end-of-period block run ONCE per period for all of them
// ST_NEVILL0 (Love, production kernel): the pass before NEVILL's first in which the bracket's two end values are evaluated again with
// the reference's arithmetic (see the evaluation block)
enum { ST_SCAN = 0, ST_REFINE = 1, ST_ELLIP = 2, ST_DONE = 3, ST_NEVILL = 4, ST_WREF = 5, ST_WEND = 6, ST_NEVILL0 = 7 };


// INDEP = false: "faithful" - a team owns a stack and walks its periods in order (reference
//   semantics: start rule 0.9*c(k-1), mmax carry-over, failure cascade).
// INDEP = true : "independent" - a team owns ONE (stack, period) root search (BASELINE north_star's
//   work unit): every period starts from the k=1 rule of fast_surf.f:157-171 evaluated at its own
//   period, on a freshly built full stack.  P times more teams, P times shorter dependency chain:
//   the mode for small batches; equal to the faithful mode to ~1e-6 on well-behaved (monotone)
//   stacks, NOT on rough ones (defects 2 and 9) - the caller opts in.
// FAST = true: opt-in heuristic coarse-to-fine scan (SURFDISP_FASTSCAN; instantiated for teams of 2, 4 and 8 lanes
// only); FAST = false: every grid point, as the reference - the default, and all larger teams.
// EXACT = false: the production root search - factorised Rayleigh recursion, team subdivision + interpolation
//   instead of NEVILL.  Two things those cannot reproduce faithfully:
//   * a bracket with more than one visible sign change - which of several roots NEVILL lands on depends on its
//     evaluation sequence: the team then runs a statement-by-statement NEVILL (surfa.f:2-83, state ST_NEVILL: one
//     evaluation per pass, every lane of the team at the same trial velocity) for that period;
//   * a secular function that leaves the fp32 range - the reference's overflow points depend on how it forms
//     its matrix entries: the team appends its stack to A.fb_list and stops.
// EXACT = true : the fallback that re-solves the listed stacks from their first period with the reference's own
//   matrix-entry arithmetic (delta_rayleigh<false>) and NEVILL for every root.  Only stacks that overflow fp32 get
//   here (physical models never do), so its speed does not matter; launched after the production kernel with a
//   grid for the worst case, idle blocks exit at once.
template <int KIND, int G, bool INDEP, bool FAST = false, bool EXACT = false>
__device__ __forceinline__ void phase_body(const PhaseArgs &A)
{
    extern __shared__ float w_lds[];
    constexpr int S = SD_PHASE_BLOCK / G;                 // stacks (teams) per workgroup
    constexpr int NFK = (KIND == 1) ? NFW_LOVE : NFW;     // fields per layer of the working stack
    constexpr int LS = lds_ls(S, NFK);                    // ... and its layer stride (words)
    const int tid = threadIdx.x;
#ifdef SD_WAVECLOCK
    const unsigned long long wclk0 = __builtin_amdgcn_s_memrealtime();
    // shader cycles (s_memtime) of this wavefront: whole loop, evaluations of the secular function, stack rebuilds
    const unsigned long long wcyc0 = __builtin_readcyclecounter();
    unsigned long long wcyc_eval = 0, wcyc_build = 0, wcyc_pre = 0, wpasses = 0;
    // team-passes by state, layers stepped per pass (the wavefront's trip count = max over lanes) and summed over lanes
    unsigned long long wn_scan = 0, wn_refine = 0, wn_nevill = 0, wn_ellip = 0, wn_idle = 0, wtrip = 0, wlanelayers = 0, wlanes = 0;
#endif
    const int lane = tid & 63;
    const int slot = tid / G;
    const int j = tid % G;                     // lane index inside the team
    const int tbase = lane - j;                // first lane of my team within the wavefront
    const unsigned long long tmask =
        (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << tbase);
    const int Lcap = A.Lmax, B = A.B, P = A.P;
    long tg = (long)blockIdx.x * S + slot;                 // team index
    bool team_valid = INDEP ? (tg < (long)B * P) : (tg < B);
    if (EXACT) {                                           // the units listed by the production kernel
        team_valid = tg < (long)(*A.fb_count);
        tg = team_valid ? (long)A.fb_list[tg] : 0;
    }
    const int b = INDEP ? (int)(tg % B) : (int)tg;         // consecutive teams = consecutive stacks
    const int k_own = INDEP ? (int)(tg / B) : 0;           // INDEP: the one period this team solves
    float *wq = w_lds + slot;
    // second slot: a snapshot of the layers the ellipticity recursion of period k still needs while
    // the main slot already holds period k+1 (only for teams of >= 4 lanes, see OVERLAP below)
    // the ellipticity (two more recursions per period, surfa.f:360-363) only feeds the group-velocity
    // kernel: a phase-only call (A.ratio == nullptr) skips it altogether
    // Production teams of >= 4 lanes never compute them: surfdisp_ellip_kernel does (one lane per (stack, period)), and
    // their instantiations carry no ellipticity state at all.  In-kernel: the exact fallback and two-lane teams (an
    // ellipticity pass with both lanes busy).  -DSD_ELL_INKERNEL_WIDE restores the r02 arrangement for A/B builds (two
    // lanes of a wide team riding in the next period's first scan pass, reading a snapshot in a second LDS slot).
#ifdef SD_ELL_INKERNEL_WIDE
    const bool want_ratio = (KIND == 2) && (A.ratio != nullptr);
    const bool OVERLAP = !EXACT && want_ratio && (G >= 4) && !INDEP && (A.overlap != 0);
#else
    constexpr bool ELL_HERE = EXACT || (G < 4);
    const bool want_ratio = ELL_HERE && (KIND == 2) && (A.ratio != nullptr);
    constexpr bool OVERLAP = false;
#endif
    float *wq2 = w_lds + (size_t)LS * Lcap + slot;
    // NEVILL's interpolation table x(1..11), y(1..11) (surfa.f:8) of this team, behind the working stacks
    float *nvx = w_lds + (size_t)((!EXACT && G >= 4 && A.overlap != 0) ? 2 : 1) * LS * Lcap + (size_t)slot * 24, *nvy = nvx + 12;
    // staged fields of this team's stack (SoA copy or rows, see PhaseArgs): field f of layer i at M_AT(f, i)
    const float *__restrict__ mrow = A.msrc + (size_t)b * A.ms_b;
    const size_t msf = (size_t)A.ms_f, msi = (size_t)A.ms_i;
#define M_AT(f, i) mrow[(size_t)(f) * msf + (size_t)(i) * msi]

    int n = 0, st = ST_DONE;
    if (team_valid) { n = A.nl[b]; if (n >= 2) st = ST_SCAN; }

    // team-uniform state
    int k = k_own, nsolved = 0, mm_carry = n, mm_frozen = n, sub = 0, passes = 0;
    int nflat_cur = n;                 // layers the current period's rebuild refreshed (recorded for the ellipticity kernel)
    float T = 1.0f, b1top = 0.0f;
    float p0c = 0.0f, p0d = 0.0f;      // "previous point" of lane 0: scan carry or bracket low end
    float cb = 0.0f, db = 0.0f;        // bracket high end (refine)
    float croot = 0.0f, r12 = 0.0f;
    int p0mm = 0;                      // effective half space the value p0d was computed with
    bool p0ok = false;                 // p0d was computed with mm_frozen (usable for interpolation)
    bool first = true;
    int status = SURFDISP_OK;
    // opt-in fast scan (teams of 2..8 lanes): after the first pass of a period the scan advances
    // FSTRIDE grid points per lane; an interval between two coarse points is skipped only if it is
    // judged free of sign changes (see below), otherwise its fine points are scanned as usual
    constexpr int FSTRIDE = (FAST && (KIND == 1 || SD_RCERT) && G <= 4) ? SD_CERT_STRIDE : 4;
    constexpr bool fastok = FAST && (G >= 2) && (G <= 8);
    // CERT (Love): the coarse scan's certificate is a theorem IN EXACT ARITHMETIC (instead of the heuristics below) behind fp32
    // guards - the "unsafe count" tests - whose margins are soaked, not proved.  At fixed frequency the angle of
    // the pair (displacement, stress) at the surface, followed continuously up from the half space, falls monotonically as
    // the trial velocity rises (Sturm / Pruefer), and a mode sits wherever it passes a multiple of pi: the number of modes
    // between two trial velocities is the difference of their crossing counts (delta_love<true>; checked against brute-force
    // root counts on 120 000 random pairs).  Two coarse points with EQUAL counts, the same effective half space (layer
    // dropping is monotone in c) and both below the half-space velocity have no root - hence no sign change at any grid
    // point - between them: the points in between need not be evaluated, and the bracket the point-by-point scan finds is
    // the one this scan finds.  An unsafe count (a sign or a multiple of pi within rounding) fails the certificate.
    // Teams of up to 8 lanes (stacks of up to ~30 layers): 65 536 x L10 0.71 -> 0.48 ms, 65 536 x L30 1.61 -> 1.05 ms; with 16
    // lanes (deep stacks) one plain pass already covers 16 grid points and the coarse pass's dearer evaluations ate the
    // gain (16 384 x L64: 1.12 -> 1.21 ms), so they keep the plain scan.  Checked against the point-by-point scan bit for
    // bit on 5.7e8 random stacks (scripts/soak_cert.py).
    // (Rayleigh: the FAST instantiations are opt-in - SURFDISP_FASTSCAN - and run the same coarse scan on the count of ray_step /
    // ray_close, which is exact but not monotone along the scan line where a branch has a zero-group-velocity point: see there.)
    constexpr bool CERT = fastok && (KIND == 1 || SD_RCERT) && !EXACT;
    int p0Kp = 0x40000000;             // Sturm count at p0 (CERT), packed: count + 4096, bit 30 = unsafe
    // ... and only on stacks where two modes cannot sit within one coarse interval: velocities that never
    // decrease with depth (no channel waves) and no layer thicker than three wavelengths of the period at
    // hand (overtones of a thick layer crowd together as (c T / 2h)^2)
    const float fsafe = (fastok && team_valid) ? A.fsafe[b] : 1.0e30f;
    bool coarse = false;               // this pass scans on the coarse grid
    int fine_left = 1;                 // fine points still to scan before going (back) to coarse
    // q0ok: the coarse point before p0 (for lane 0); !q0ok (first coarse pass after fine ones): the fine point p0 - dc
    float q0c = 0.0f, q0d = 0.0f; int q0mm = 0; bool q0ok = false;
    float p0phi = 0.0f;                // vertical phase (delta_rayleigh) at p0c, valid whenever coarse is set
    // pending ellipticity of the previous period (OVERLAP): evaluated by lanes 0-1 of the first
    // scan pass of the next period instead of costing a pass of its own
    bool ell_pend = false;
    int ell_k = 0, ell_mm = 2;
    float ell_c = 1.0f, ell_T = 1.0f;
    // NEVILL's state between two evaluations (c1, del1 = p0c, p0d; c2, del2 = cb, db; c3 = croot)
    int nv_nev = 1, nv_m = 1, nv_ic = 0;
    bool defer = false;                // !EXACT: this stack goes to the exact fallback kernel
    bool ell_flag = false;             // in-kernel ellipticity of the period at hand: to be redone by the ellipticity kernel
    // ... whenever c^2 is below this: 2 b^2 / c^2 of the stack's fastest layer beyond A.ell_gmax (read once: the prep kernel's statistic)
    const float ell_c2min = (want_ratio && !EXACT && team_valid && A.ovf != nullptr && A.ell_ambig != 0.0f)
                                ? __expf(0.25f * A.ovf[2 * (size_t)B + b]) / A.ell_gmax : 0.0f;

    // (re)build the working stack for period k over the first nflat layers only -- the reference
    // refreshes just the layers inside the previous period's effective half space and leaves the
    // deeper ones stale (calcul.f:112,133); LDS keeps them across periods exactly like COMMON /d/.
    auto build = [&](int nflat) {
        const float lnT = logf(1.0f / T);                      // alog(t_base/t1), calcul.f:122
        for (int i = j; i < nflat; i += G) {
            // only the fields this layer's role needs (regular: dif, qqq, dflat; half space: hsf, hsr);
            // this block runs whenever ANY team of the wavefront changes period, i.e. almost every
            // pass and with one or two teams active, so its loads and divisions are worth counting
            const bool hs = (i == nflat - 1);
            LayerRaw r;
            r.a_ref = M_AT(F_VP, i); r.b_ref = M_AT(F_VS, i); r.rho_ref = M_AT(F_RHO, i);
            r.qs = M_AT(F_QS, i);
            r.dif = hs ? 0.0f : M_AT(F_DIF, i); r.qqq = hs ? 0.0f : M_AT(F_QQQ, i);
            r.dfl = hs ? 0.0f : M_AT(F_DFL, i);
            r.hsf = hs ? M_AT(F_HSF, i) : 0.0f; r.hsr = hs ? M_AT(F_HSR, i) : 0.0f;
            const LayerV v = layer_derive(r, lnT, hs);
            // the reciprocals are this kernel's own helper values (not the reference's): v_rcp + one Newton
            // step (<= 1 ulp) instead of three IEEE divisions
            W_B(i) = v.b; W_R(i) = v.rho; W_D(i) = v.d;
            if (KIND == 1 && !EXACT) W_IR(i) = rcp_nr(v.rho * v.b * v.b);       // Love, production: 1/(rho b^2)
            else if (i == 0 || EXACT) W_IR(i) = rcp_nr(v.rho);
            if (KIND == 2) {
                W_IA2(i) = EXACT ? v.a : rcp_nr(v.a * v.a);                      // exact kernel: a itself (delta_rayleigh_ref)
                W_IB2(i) = (v.b > 0.0f) ? rcp_nr(v.b * v.b) : 0.0f;
            }
        }
        if (KIND == 2 && !EXACT) {
            // the carried state of delta_rayleigh is rescaled by rho(m-1)/rho(m) on entering layer m: formed here, once
            // per period, from the densities now in the slot - the first stale layer below the refreshed ones included
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int top = nflat < n ? nflat : n - 1;
            for (int i = 1 + j; i <= top; i += G) W_IR(i) = W_R(i - 1) * rcp_nr(W_R(i));
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // growth of the vertical phase between two trial velocities over the first mm - 1 layers of the working stack (see the
    // bracket branch of the scan); every lane of the team calls it.  Returns 0 where two bounds already show it below
    // A.phimulti: no layer's share exceeds omega d sqrt(1/c1^2 - 1/c2^2) per wave type (sqrt(x + e) - sqrt(x) <= sqrt(e)) and only
    // layers that are oscillatory at c2 have one - first with the thickness of the whole working stack (dtot, summed once per
    // period for the layer-dropping shortcut), then with the thickness of the oscillatory layers (a compare and an add per
    // layer); the sum of square roots itself is left to the few brackets that pass both (Love, ten layers: it cost 5 % of the
    // root search when every bracket formed it).
    float dtot = 0.0f;
    auto bracket_phase = [&](float c1, float c2, int mm) -> float {
        const float om = 6.2831853f * __builtin_amdgcn_rcpf(T);
        const float i1 = __builtin_amdgcn_rcpf(c1 * c1), i2 = __builtin_amdgcn_rcpf(c2 * c2);
        const float sq = 1.001f * om * sqrt_hw(fmaxf(i1 - i2, 0.0f));
        if (sq * dtot * ((KIND == 2) ? 2.0f : 1.0f) <= A.phimulti) return 0.0f;
        float dosc = 0.0f;
        for (int i = j; i < mm - 1; i += G) {
            const float d = W_D(i);
            if (KIND == 2) dosc += ((W_IB2(i) > i2) ? d : 0.0f) + ((W_IA2(i) > i2) ? d : 0.0f);
            else           dosc += (W_B(i) < c2) ? d : 0.0f;
        }
#pragma unroll
        for (int d = G >> 1; d > 0; d >>= 1) dosc += __shfl_xor(dosc, d);
        if (sq * dosc <= A.phimulti) return 0.0f;
        float sum = 0.0f;
        for (int i = j; i < mm - 1; i += G) {
            const float od = om * W_D(i);
            const float ib2 = (KIND == 2) ? W_IB2(i) : W_IR(i) * W_R(i);       // 1/b^2 (0: liquid)
            sum += od * (sqrt_hw(fmaxf(ib2 - i2, 0.0f)) - sqrt_hw(fmaxf(ib2 - i1, 0.0f)));
            if (KIND == 2) {
                const float ia2 = W_IA2(i);
                sum += od * (sqrt_hw(fmaxf(ia2 - i2, 0.0f)) - sqrt_hw(fmaxf(ia2 - i1, 0.0f)));
            }
        }
#pragma unroll
        for (int d = G >> 1; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        return fabsf(sum);
    };

    // Production kernel: can the reference's arithmetic overflow fp32 in this period although the production recursion stays
    // finite?  Rayleigh (DLTAR4 normalises per layer): a matrix ENTRY.  The largest entry is a51 ~ rhoc^2 g^4 rsinp rsinq <= rhoc^2 g^4 e^(pm+qm)
    // with pm + qm <= 2 k d.  Bound it per stack (thickest layer, largest rho and Vs: prep kernel) at the lowest
    // trial velocity of the period; beyond e^84 the stack goes to the exact fallback.
    // Love: DLTAR1 carries (ut, tt) through the layers WITHOUT normalisation (surfa.f:143-179): across evanescent layers the pair
    // grows like exp(sum of k d sqrt(1 - c^2/b^2)) and overflows fp32 beyond e^88 (or a layer's exp(q) underflows and its
    // reciprocal is inf) - the reference then scans NaNs and returns the edge of the overflowed region as the root, while the
    // production recursion stays finite and finds the true one.  Layer dropping bounds that sum: the layer at which the
    // evanescent thickness passes 4 c T becomes the half space, so the layers above it hold at most k x 4 c T = 8 pi of
    // exponent - EXCEPT when the TOP layer alone passes it (mmax is at least 2, surfa.f:105: the top layer stays a layer
    // whatever its thickness).  That one product, at the period's lowest trial velocity, decides: from e^76 on (the rest of the
    // margin: the 8 pi of further layers are not possible then, rho b^2 rb factors are) the stack goes to the exact fallback.
    // 200-km top layers at T < 3 s; no stack of the bench workloads.
    auto entry_overflow_risk = [&](float c_lo) -> bool {
        if (EXACT || !team_valid) return false;
        if (KIND == 1) {
            const float bb = W_B(0);                                        // (water on top, b = 0: never counted, never kept)
            if (!(bb > c_lo)) return !(c_lo > 0.0f && T > 0.0f);
            const float q0 = 6.2831853f * __builtin_amdgcn_rcpf(c_lo * T) * W_D(0) *
                             sqrt_hw(fmaxf(1.0f - c_lo * c_lo * __builtin_amdgcn_rcpf(bb * bb), 0.0f));
            return !(q0 < 76.0f);                                           // also when c_lo or T is not positive
        }
        const float hthick = A.ovf[b], lnrho2 = A.ovf[(size_t)B + b], lng4 = A.ovf[2 * (size_t)B + b];
        const float c_hi = W_B(mm_carry - 1) + 0.31f;                      // upper guard of the scan, calcul.f:166
        const float lnmag = 12.566371f * hthick / (c_lo * T) + lnrho2 + 4.0f * __logf(c_hi) +
                            fmaxf(lng4 - 8.0f * __logf(c_lo), 0.0f);
        return !(lnmag < 84.0f);                                           // also when c_lo or T is not positive
    };

    // Layer dropping (surfa.f:94-105) picks the first layer at which the thickness summed over the layers with
    // c < b exceeds 4 c T.  When the thickness of ALL n layers in the working stack (stale deep ones included) stays
    // below 4 c T at the period's lowest trial velocity, no trial of the period can drop anything: mmax = n without
    // walking the stack (three quarters of the bench workload's periods).  1e-5 covers the rounding of the partial sums.
    bool nodrop = false;
    auto no_drop_possible = [&](float c_lo) -> bool {
        float dsum = 0.0f;
        for (int i = j; i < n; i += G) dsum += W_D(i);
#pragma unroll
        for (int d = G >> 1; d > 0; d >>= 1) dsum += __shfl_xor(dsum, d);
        dtot = dsum;                                                       // (bracket_phase's first bound)
        return dsum * 1.00001f <= FACT * c_lo * T;
    };

    if (st != ST_DONE) {
        T = A.per[k];
        // clear the slot (a new process sees zeroed COMMON /d/)
        for (int i = j; i < Lcap; i += G) { W_IR(i) = 0.0f; W_B(i) = 0.0f; W_R(i) = 0.0f; W_D(i) = 0.0f; if (KIND == 2) { W_IA2(i) = 0.0f; W_IB2(i) = 0.0f; } }
        build(n);
        b1top = W_B(0);
        // first guess, fast_surf.f:157-171
        const bool water = M_AT(F_VS, 0) < 0.1f;
        const int il = water ? 1 : 0;
        const float b_corr = M_AT(F_QS, il) * logf(1.0f / T) / PI_REF;
        float qq = M_AT(F_VS, il);
        if (KIND == 2) qq = 0.9f * qq;
        p0c = qq * (1.0f + b_corr);
        if (water) p0c = 0.5f;
        first = true;
        nodrop = no_drop_possible(p0c);
        defer = (!EXACT && A.strict != 0) || entry_overflow_risk(p0c);    // SURFDISP_STRICT: everything to the exact kernel
        // CERT: a period in which no trial can drop layers (one eigenproblem for every trial velocity) starts on the coarse grid
        if (CERT) coarse = nodrop;
    }

    int wprio = -1;
    const bool LOCK = !EXACT && (INDEP ? A.lockstep >= 2 : A.lockstep != 0);
    // NEVILL's prologue, surfa.f:12-16, from the scan's bracket (del1 is the scan's value, whatever layer dropping it was computed
    // with - as in the reference)
    auto nevill_start = [&]() {
        nv_ic = 0; nv_nev = 1; nv_m = 1;
        croot = (p0c + cb) / 2.0f;                             // c3, evaluated by the next pass
        st = (KIND == 1 && !EXACT) ? ST_NEVILL0 : ST_NEVILL;   // (Love: first the end values in the reference's arithmetic)
        sub = 0;
        if (!EXACT && j == 0 && A.amb_count) atomicAdd(A.amb_count, 1);     // (statistics)
    };
    while (__any(st != ST_DONE)) {
#ifdef SD_WAVECLOCK
        const unsigned long long wp0 = __builtin_readcyclecounter();
#endif
        // The SIMD arbitrates between its wavefronts by priority, then age: left alone, the four wavefronts of a
        // SIMD finish one after the other and the last one runs alone for a seventh of the kernel.  Let a wavefront
        // that is behind (period index of its first team) go first: 2.39 -> 2.15 ms for one bench batch (profiles/r02c).
        // Not for narrow teams (< 8 lanes) when the caller keeps a second batch in flight (SURFDISP_PIPELINED): there the age
        // order lets the older batch drain while the younger one fills the machine (three bench batches in flight: 38.4 M
        // solves/s without, 37.7 M with).  Wide teams keep it also then: the two root searches of a joint Rayleigh + Love
        // solve of 64-layer stacks, 5.59 -> 5.23 ms (r03).  SURFDISP_BALANCE=0 / 1 forces it.
        if (!INDEP && A.balance) {
            const int kw = __builtin_amdgcn_readfirstlane(k);
            const int pr = 3 - min(3, (4 * kw) / max(P, 1));
            if (pr != wprio) {
                wprio = pr;
                if (pr == 3) __builtin_amdgcn_s_setprio(3);
                else if (pr == 2) __builtin_amdgcn_s_setprio(2);
                else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        }
        // ---------------------------------------------------------------- choose the trial point
        float cj = 1.0f;
        int mmj = 2, start = 1;
        bool eval = (st != ST_DONE) && (st != ST_WREF) && (st != ST_WEND);
        const float *wl = wq;                                  // the stack this lane's recursion reads
        float Tl = T;
        const bool ell_lane = OVERLAP && ell_pend && (st == ST_SCAN) && (j < 2);
        const int js = (OVERLAP && ell_pend && st == ST_SCAN) ? j - 2 : j;   // scan-lane index in the team
        if (ell_lane) {
            cj = ell_c; mmj = ell_mm; start = 2 + j; wl = wq2; Tl = ell_T;
        } else if (st == ST_SCAN) {
            // exact fp32 grid of the reference: c2 = c1 + dc repeatedly (calcul.f:157,161)
            const int nadd = (fastok && coarse) ? FSTRIDE * (first ? js : js + 1) : (first ? js : js + 1);   // (CERT may start a period on the coarse grid)
            cj = p0c;
            if (fastok) {
#pragma unroll
                for (int i = 0; i < FSTRIDE * G; ++i) if (i < nadd) cj = cj + DC;
            } else {
#pragma unroll
                for (int i = 0; i < G; ++i) if (i < nadd) cj = cj + DC;
            }
            mmj = nodrop ? (n < 2 ? 2 : n) : drop_layers(wq, LS, S, n, cj, T);   // idrop=0 before every scan trial
        } else if (st == ST_NEVILL0) {
            // the scan's two end values again (del1 with the layer dropping of ITS trial, del2's is the frozen one)
            const bool hi = (G == 1) ? (sub != 0) : ((j & 1) != 0);
            cj = hi ? cb : p0c; mmj = hi ? mm_frozen : p0mm;
        } else if (st == ST_NEVILL) {
            cj = croot; mmj = mm_frozen;                       // NEVILL's c3, idrop = 1
        } else if (st == ST_REFINE) {
            const float w = cb - p0c;
            cj = p0c + (float)(j + 1) * (w * (1.0f / (float)(G + 1)));
            if (G <= 4 && G > 1 && p0ok && w > 16.0f * CLUSTER_DC) {
                // small teams: instead of G equidistant points, cluster them around the secant
                // estimate (spacing CLUSTER_DC).  Delta(c) is smooth across a 0.01 bracket, so the
                // root normally falls between two neighbours and the next acceptance test passes
                // (one refine pass instead of two); if it does not, the sign pattern still shrinks
                // the bracket and the next pass clusters around a better estimate.
                // (estimates of where to put trial points and of the root inside a bracket <= 0.01 km/s wide: v_rcp_f32
                // quotients, 1 ulp - an IEEE division is ten instructions, and a dozen of them sat on every pass)
                float ts = -p0d * w * __builtin_amdgcn_rcpf(db - p0d);
                // offsets in units of CLUSTER_DC: tight around the estimate, wider outside, so that a
                // poorer estimate still lands between two points
                const float off = (G == 4) ? ((j == 0) ? -4.0f : (j == 1) ? -1.0f : (j == 2) ? 1.0f : 4.0f)
                                           : ((j == 0) ? -1.5f : 1.5f);
                const float half = ((G == 4) ? 4.0f : 1.5f) * CLUSTER_DC;
                ts = fminf(fmaxf(ts, half + CLUSTER_DC), w - half - CLUSTER_DC);
                if (ts == ts) cj = p0c + (ts + off * CLUSTER_DC);
            }
            mmj = mm_frozen;                                   // frozen as NEVILL sees it
        } else if (st == ST_ELLIP) {
            cj = croot; mmj = mm_frozen;
            start = (G == 1) ? 2 + sub : 2 + j;
            eval = (G == 1) || (j < 2);
        }
        float val = 0.0f, phj = 0.0f;
        int kcj = 0; bool kuncj = false;                       // Sturm count of this lane's trial (CERT)
#ifdef SD_WAVECLOCK
        const unsigned long long we0 = __builtin_readcyclecounter();
        wcyc_pre += we0 - wp0;                                 // priority, trial velocities, layer dropping
        ++wpasses;
        {
            const unsigned long long lead = __ballot(j == 0);
            wn_scan += __popcll(__ballot(st == ST_SCAN) & lead); wn_refine += __popcll(__ballot(st == ST_REFINE) & lead);
            wn_nevill += __popcll(__ballot(st == ST_NEVILL) & lead); wn_ellip += __popcll(__ballot(st == ST_ELLIP) & lead);
            wn_idle += __popcll(__ballot(st == ST_DONE || st == ST_WREF || st == ST_WEND) & lead);
            int mx = eval ? mmj : 0, sm = eval ? mmj : 0;
            for (int d = 32; d > 0; d >>= 1) { mx = max(mx, __shfl_xor(mx, d)); sm += __shfl_xor(sm, d); }
            wtrip += mx; wlanelayers += sm; wlanes += __popcll(__ballot(eval));
        }
#endif
        float vmag = 0.0f;                                     // magnitude of the terms the value is the sum of (production recursion)
#if defined(SD_AMBIG) && SD_AMBIG == 2
        bool amb_defer = false;
#endif
        if (eval) {
#ifdef SD_AMBIG
            const bool want_mag = true;
#else
            const bool want_mag = want_ratio && st == ST_ELLIP;   // (the in-kernel ellipticity passes' cancellation test)
#endif
            if (KIND == 2) val = EXACT ? delta_rayleigh_ref(wl, LS, S, mmj, cj, Tl, start)
                                       : delta_rayleigh<(G != 2) && !FAST, CERT>(wl, LS, S, mmj, cj, Tl, start, phj, &vmag, want_mag, &kcj, &kuncj, coarse);
            // Love, NEVILL passes of the production kernel: DLTAR1 statement by statement on the production working stack (it holds
            // b, rho, d as the exact kernel's does).  A bracket goes to NEVILL because it may hold several roots, and which of them
            // NEVILL lands on depends on the VALUES it sees (its 10 x rule, its interpolation) - with e^{kd} of hundreds of km of
            // layer the production recursion's values differ from the reference's by whole orders of magnitude, or are inf where
            // those are finite (r04 soak, thick-layer family, Love: 2e-4 of the stacks on another overtone with the production
            // values, 1.4e-2 before there was a NEVILL for such brackets at all).
#ifndef SD_NO_LOVE_REFNEV
            else if (!EXACT && (st == ST_NEVILL || st == ST_NEVILL0)) val = delta_love_ref_body(wl, LS, S, mmj, cj, Tl);
#endif
            else           val = EXACT ? delta_love_ref(wl, LS, S, mmj, cj, Tl) : delta_love<CERT>(wl, LS, S, mmj, cj, Tl, phj, kcj, kuncj, coarse,
#ifdef SD_AMBIG
                                                                                                      &vmag
#else
                                                                                                      nullptr
#endif
                                                                                                      );   // counts only where they are compared: coarse passes
        }
        // (-DSD_AMBIG builds only, see DESIGN.md.)  A scan trial whose value is the remainder of a cancellation - |value| below A.ambig of the terms it is the sum of -
        // has a SIGN within the rounding of the production recursion, and the scan decides on signs (calcul.f:157-167): a
        // flipped sign moves the bracket by one grid step onto a neighbouring root or past a pair of close roots (the zero
        // pattern mismatches of the soaks, 2e-5 of random stacks).  Such a trial is evaluated again with the reference's own
        // arithmetic (DLTAR4 / DLTAR1 statement by statement) and that value's sign is used.  Rare by construction (counted in
        // A.amb_count); the wavefront's other lanes wait for it.
#ifdef SD_AMBIG
        if (!EXACT && eval && st == ST_SCAN && !ell_lane && A.ambig > 0.0f && fabsf(val) < A.ambig * vmag) {
#if SD_AMBIG == 2   // (second experiment: the whole stack to the exact fallback kernel - the reference's arithmetic on the reference's own inputs)
            amb_defer = true;
#else
            val = (KIND == 2) ? delta_rayleigh_ref_body<true>(wl, LS, S, mmj, cj, Tl, 1) : delta_love_ref_body(wl, LS, S, mmj, cj, Tl);
#endif
            if (A.amb_count) atomicAdd(A.amb_count, 1);
        }
#if SD_AMBIG == 2
        if ((__ballot(amb_defer) & tmask) != 0ull) defer = true;
        amb_defer = false;
#endif
#endif
        // ... and the in-kernel ellipticity passes (two-lane teams) likewise: a closure that is the remainder of a cancellation
        // marks the (stack, period) for the ellipticity kernel, which evaluates both passes with the reference's arithmetic
        // on the replayed working stack (see there)
#ifdef SD_NO_ELLG
        const bool ell_amb = false;
#else
        const bool ell_amb = !EXACT && want_ratio && eval && st == ST_ELLIP && A.ell_ambig != 0.0f &&
                             (A.ell_ambig < 0.0f || fabsf(val) < A.ell_ambig * vmag ||
                              cj * cj < ell_c2min);                                       // see surfdisp_ellip_kernel
#endif
#ifdef SD_WAVECLOCK
        wcyc_eval += __builtin_readcyclecounter() - we0;
#endif
        // ---------------------------------------------------------------- team-level decisions
        const int lm1 = (lane + 63) & 63;
        const float sc = __shfl(cj, lm1), sv_ = __shfl(val, lm1);
        const float pc = (j == 0) ? p0c : sc;
        const float pd = (j == 0) ? p0d : sv_;
        const int smm = __shfl(mmj, lm1);
        const int pmm = (j == 0) ? p0mm : smm;
        const int kpk = CERT ? ((kcj + 4096) | (kuncj ? 0x40000000 : 0)) : 0;          // count | unsafe flag, packed
        const int sKp = CERT ? __shfl(kpk, lm1) : 0;                                   // ... of the previous lane
        const int pKp = (j == 0) ? p0Kp : sKp;
        const bool searching = ((st == ST_SCAN) || (st == ST_REFINE)) && !ell_lane;
        const bool has_prev = !((st == ST_SCAN) && first && (js == 0));
        auto negnan = [](float x) { return signbit(x) && !(x != x); };   // a NaN compares as positive, see below
        const bool cross = has_prev && (negnan(val) != negnan(pd));
#ifdef SD_DEBUG_TRIALS   // (developer build: every evaluated trial of a one-stack call)
        if (A.B == 1 && eval && k == SD_DEBUG_TRIALS && cj > SD_DEBUG_CMIN)
            printf("k %d pass %d st %d lane %d c %.7f val % .6e mm %d count %d unsafe %d coarse %d | p0c %.7f p0d % .4e cb %.7f db % .4e\n", k, passes, st, j, cj, val, mmj, kcj, (int)kuncj, (int)coarse, p0c, p0d, cb, db);
#endif
        bool guard = false;
        if (st == ST_SCAN && has_prev && !cross)               // calcul.f:165-166
            guard = (cj < 0.8f * b1top) || !(cj < W_B(mmj - 1) + 0.3f);
        // coarse pass: the interval (previous coarse point, this one) may be skipped only if the
        // secular function has the same sign at both ends, was evaluated with the same effective
        // half space at the coarse points around it, and ln|Delta| bends so little at BOTH ends of the
        // interval that no pair of roots can hide in it.  Anything else is rescanned point by point.
        bool uncert = false, back0 = false;
        if (CERT) {
            if (coarse && st == ST_SCAN) {
                // below the half-space velocity by two coarse steps (there the problem stops being an eigenvalue problem)
                const bool near_hs = !(cj < W_B(mmj - 1) - 2.0f * (float)FSTRIDE * DC);
                // ... and the scan's lower guard (calcul.f:165: a trial below 0.8 b(1) ends the search) looks at every grid
                // point: the first one a skip would pass over is pc + dc
                const bool low_guard = (pc + DC) < 0.8f * b1top;
                uncert = has_prev && (near_hs || low_guard || !((pmm == mmj) && fin(pd) && fin(val) && !kuncj && (pKp == kpk)));
            }
        } else if (fastok) {
        const int ln1 = (lane + 1) & 63, lm2 = (lane + 62) & 63;
        const float nx_d = __shfl(val, ln1), sp_d = __shfl(val, lm2);
        const int nx_mm = __shfl(mmj, ln1), sp_mm = __shfl(mmj, lm2);
        const float sphi = __shfl(phj, lm1), nphi = __shfl(phj, ln1);
        // ln(p n / m^2) < 1 for three same-sign values: the second difference of ln|Delta| at the middle point.
        // A pair of roots between two coarse points lifts it to >= 2.2 at one of them whatever exponential
        // envelope multiplies the function (soft layers: e^{k d} factors change Delta by orders of magnitude per
        // coarse step; a test on the second difference of Delta itself, r01i-r01l, is blind there and fires
        // before most simple roots instead)
        auto logsd_ok = [](float p, float m, float n) {
            const float rm = __builtin_amdgcn_rcpf(m);
            const float a = p * rm, b = n * rm;
            return (a > 0.0f) && (b > 0.0f) && (a * b < 2.7182818f);
        };
        if (coarse && st == ST_SCAN) {
            const bool has_next = (j < G - 1);
            const bool has_pp = (j >= 1) || q0ok;
            const float pp_d = (j >= 2) ? sp_d : ((j == 1) ? p0d : q0d);
            const int pp_mm = (j >= 2) ? sp_mm : ((j == 1) ? p0mm : q0mm);
            // right end: with the next point (a last lane's right end is looked at by lane 0 of the next pass, back0
            // below).  A next point of the other sign means a root in the NEXT interval; two more in this one
            // would be three modes within two intervals, which the phase rule excludes.
            const bool nx_same = (negnan(nx_d) == negnan(val));
            const bool okf = !has_next || ((pmm == mmj) && (mmj == nx_mm) && fin(nx_d) &&
                                           (nx_same ? logsd_ok(pd, val, nx_d) : (fabsf(nphi - phj) < A.phimax)));
            // left end: with the coarse point before ...
            const bool oklog_b = has_pp && logsd_ok(pp_d, pd, val);
            const bool okb = !has_pp || ((pp_mm == pmm) && (pmm == mmj) && fin(pp_d) && oklog_b);
            // ... or, in the first coarse pass after fine ones, with the fine point before p0: the slope of
            // ln|Delta| over the last fine step against the slope over this interval (a pair close behind p0
            // bends it by 1.2-2.3 per fine step; 0.5 is allowed)
            bool oke = true;
            if (j == 0 && !q0ok) {
                const float rf = pd * __builtin_amdgcn_rcpf(q0d), rc = val * __builtin_amdgcn_rcpf(pd);
                const float rf2 = rf * rf;
                const float x = rc * __builtin_amdgcn_rcpf(rf2 * rf2);
                oke = (q0mm == pmm) && (pmm == mmj) && (rf > 0.0f) && (rc > 0.0f) && (x > 0.135f) && (x < 7.39f);
            }
            // the secular function is analytic in c except at the half-space velocity (its closure is
            // linear in sqrt|c^2/b^2 - 1|): within two coarse steps of that branch point nothing is skipped
            const bool near_hs = !(cj < W_B(mmj - 1) - 2.0f * (float)FSTRIDE * DC);
            // ... and the interval must be too short for two modes: consecutive modes are ~pi apart in the
            // vertical phase summed over the oscillatory layers, however the stack is built (thick or slow
            // layers, short periods); the interval may add at most A.phimax (pi/4 by default) to it
            const float pphi = (j == 0) ? p0phi : sphi;
            const bool okphi = fabsf(phj - pphi) < A.phimax;
            uncert = near_hs || !(okf && okb && oke && okphi && fin(pd) && fin(val));
            // lane 0 also holds the right end of the previous pass's last interval (q0, p0): if that fails the
            // rescan starts at q0
            back0 = (j == 0) && q0ok && (cross ? !okphi : !oklog_b);
        }
        }
        const bool ev = searching && (cross || guard || uncert);
        const unsigned long long em = __ballot(ev) & tmask;
        const int fl = em ? (__ffsll((long long)em) - 1) : -1;
        const int src = (fl < 0) ? tbase : fl;
        const float e_c = __shfl(cj, src), e_d = __shfl(val, src);
        const float e_pc = __shfl(pc, src), e_pd = __shfl(pd, src);
        const int e_mm = __shfl(mmj, src);
        const int e_pmm = __shfl(pmm, src);
        const int l_mm = __shfl(mmj, tbase + G - 1);
        const int e_cross = __shfl((int)cross, src);
        const int t_back0 = fastok ? __shfl((int)back0, tbase) : 0;
        const int lastl = tbase + G - 1;
        const float l_c = __shfl(cj, lastl), l_d = __shfl(val, lastl);
        const float l_phi = fastok ? __shfl(phj, lastl) : 0.0f;
        const int l_Kp = CERT ? __shfl(kpk, lastl) : 0;
        const int e_pKp = CERT ? __shfl(pKp, src) : 0;
        // the values of the team's first two lanes: the two ellipticity recursions, and NEVILL's del3 (every lane of the team
        // evaluated the same c3) - only where some team of the wavefront needs them
        float v0 = 0.0f, v1 = 0.0f;
        if (__any(want_ratio || st == ST_NEVILL || st == ST_NEVILL0)) { v0 = __shfl(val, tbase); v1 = __shfl(val, (G > 1) ? tbase + 1 : tbase); }
        // what only a REFINE pass reads (wavefront-uniform test: in lock step most passes have no refining team):
        // the right neighbour of the crossing lane, the lane before the last one, and - the fourth point of the second
        // three-point estimate - two lanes below / above the crossing lane and two before the last one
        float e_nc = 0.0f, e_nd = 0.0f, pl_c = 0.0f, pl_d = 0.0f, e_ppc = 0.0f, e_ppd = 0.0f, e_n2c = 0.0f, e_n2d = 0.0f,
              pl2_c = 0.0f, pl2_d = 0.0f;
        int pl_mm = 0;
        if (fastok || __any(st == ST_REFINE)) {
            const int nxt = (src < lastl) ? src + 1 : lastl;
            e_nc = __shfl(cj, nxt); e_nd = __shfl(val, nxt);
            const int pl = (G > 1) ? lastl - 1 : lastl;
            pl_c = __shfl(cj, pl); pl_d = __shfl(val, pl);
            if (fastok) pl_mm = __shfl(mmj, pl);
            const int lm2s = (src - 2 >= tbase) ? src - 2 : tbase;
            e_ppc = __shfl(cj, lm2s); e_ppd = __shfl(val, lm2s);
            const int nx2 = (src + 2 <= lastl) ? src + 2 : lastl;
            e_n2c = __shfl(cj, nx2); e_n2d = __shfl(val, nx2);
            const int pl2 = (G > 2) ? lastl - 2 : lastl;
            pl2_c = __shfl(cj, pl2); pl2_d = __shfl(val, pl2);
        }
        const bool had_ell = OVERLAP && ell_pend && (st == ST_SCAN);

        bool solved = false, failed = false;
        // A secular function that left the fp32 range (NaN / inf: products of e^{k d} terms beyond 3e38 in very
        // thick layers at short periods).  The reference's scan compares SIGN(1., del): the NaNs its arithmetic
        // produces carry the sign bit (x86 default NaN) and both secular functions return the NEGATED recursion
        // result (surfa.f:179,357), so a NaN passes as a positive value (negnan above).  In NEVILL a NaN end value
        // makes the Neville step return a NaN abscissa, and the arithmetic IFs of surfa.f:32-34 send a NaN to their
        // third label, i.e. to a BISECTION step (flang and gfortran lower `if (x) l1,l2,l3` to x<0, x==0, else):
        // the reference keeps halving on signs alone and returns the edge of the overflowed region as that
        // period's root (pinned bit for bit by the fixtures tests/golden/ref_families.npz).  Same here: the
        // subdivision below decides on signs with NaN = positive, and interpolation is only trusted on finite
        // values (a non-finite estimate falls back to the bracket's low end once it is 1e-6 wide).
        // What NEVILL cannot do is separate two fp32 numbers above 16 km/s (spacing 1.9e-6 > its 1e-6 tolerance,
        // surfa.f:10,44): its 50-cycle limit trips (surfa.f:17-27), calcul.f:172-189 jumps to 9999 and the whole
        // call returns nothing, also the periods already solved - SURFDISP_NUMERIC (see `fatal` in REFINE below).
        bool fatal = false, multi = false;
        if (!EXACT) {
            // what this kernel does not reproduce faithfully (see the template flags)
            const bool nonfin = ((__ballot(eval && !fin(val)) & tmask) != 0ull);
            if (st != ST_DONE && nonfin) defer = true;         // -> exact fallback kernel
            if (st == ST_REFINE && passes == 0) {              // first refine pass: sign changes across the bracket
                const int ncross = __popcll(__ballot(searching && cross) & tmask) + ((negnan(l_d) != negnan(db)) ? 1 : 0);
                multi = ncross >= 2;                           // -> NEVILL for this period
            }
        }
        if (OVERLAP && ell_pend && st == ST_SCAN) {
            if (j == 0) A.ratio[(size_t)ell_k * B + b] = 0.5f * v1 / v0;   // surfa.f:363
            ell_pend = false;
        }
        if (fastok && st == ST_SCAN && coarse) {
            ++passes;
            if (fl >= 0) {                                     // rescan this interval point by point
                p0c = e_pc; p0d = e_pd; p0mm = e_pmm;
                if (CERT) p0Kp = e_pKp;
                const bool e_back = (fl == tbase) && (t_back0 != 0);
                if (e_back) { p0c = q0c; p0d = q0d; p0mm = q0mm; }
                // after a failed certificate stay on the fine grid for the next interval too (a restart at q0
                // has two intervals to cover; a change of the layer dropping usually sits close to the root)
                coarse = false; fine_left = (e_cross && !e_back) ? FSTRIDE : 2 * FSTRIDE; q0ok = false;
                if (CERT) { fine_left = FSTRIDE; first = false; }      // (the certified scan rescans the one interval)
            } else {
                q0c = pl_c; q0d = pl_d; q0mm = pl_mm; q0ok = true;
                p0c = l_c; p0d = l_d; p0mm = l_mm; p0phi = l_phi;
                if (CERT) { p0Kp = l_Kp; first = false; }
            }
        } else if (st == ST_SCAN) {
            ++passes;
            if (fl >= 0 && e_cross) {                          // bracket found -> refine
                p0c = e_pc; p0d = e_pd; cb = e_c; db = e_d; mm_frozen = e_mm; p0mm = e_pmm;
                // the low end was evaluated with ITS OWN layer dropping (idrop=0 per scan trial); if
                // that differs from the frozen one its magnitude belongs to a different function
                // and only its sign may be used (NEVILL's 10x guard, surfa.f:47-51, covers this)
                p0ok = (e_pmm == e_mm);
                passes = 0;
                st = LOCK ? ST_WREF : ST_REFINE;
                // EXACT: always NEVILL.  Production: NEVILL (statement by statement, from the scan's bracket) when the
                // bracket may hold SEVERAL roots: consecutive modes are ~pi apart in the vertical phase (sum of omega d
                // sqrt(1/v^2 - 1/c^2) over the oscillatory layers), so a bracket across which it grows by more than A.phimulti
                // (1 rad; an ordinary bracket's growth is 1e-2) - thick soft layers at short periods, overtones 1e-3 km/s
                // apart - may hold more than one, which of them NEVILL lands on depends on its evaluation sequence, and a
                // team of a few lanes cannot see them: the subdivision's own test (`multi`: more than one sign change among
                // the G points of the first refine pass) only sees roots further apart than its points.  r04 soak: 4e-3 of
                // the soft-sediment family's Love stacks came back on another overtone (1 .. 15 % off) from teams of <= 8
                // lanes.  The phase is summed by the team, once per bracket (a layer per lane and turn).
                // (Lock step: the teams of a wavefront find their brackets in different passes, and a block run for one team
                // costs the wavefront as much as for all - the phase is summed when all of them leave ST_WREF together, below.)
#ifdef SD_NO_PHASEMULTI
                if (EXACT) nevill_start();
#else
                if (EXACT || (!LOCK && bracket_phase(p0c, cb, mm_frozen) > A.phimulti)) nevill_start();
#endif
            } else if (fl >= 0) {
                failed = true;                                 // label 250
            } else {
                p0c = l_c; p0d = l_d; p0mm = l_mm; first = false;
                if (CERT) p0Kp = l_Kp;
                if (fastok) { p0phi = l_phi; q0d = pl_d; q0mm = pl_mm; }    // pl: the fine point p0 - dc
                if (passes > 100000) failed = true;            // cannot happen: c grows by dc/pass
                if (fastok) {
                    fine_left -= had_ell ? G - 2 : G;
                    if (fine_left <= 0 && (CERT ? nodrop : (fsafe <= 3.0f * p0c * T))) { coarse = true; q0ok = false; }
                    // (CERT: fine passes carry no counts, so the coarse pass starts AT p0 - its first lane evaluates it again)
                    if (CERT && coarse) first = true;
                }
            }
        } else if (!EXACT && st == ST_REFINE && multi) {
            // more than one sign change inside the bracket: hand this period to NEVILL, from the scan's bracket
            // (p0c, cb and their values are still the scan's: this is the first refine pass).  (Sending the whole stack
            // to the exact fallback kernel instead changes nothing measurable: soak mismatch rates 1.70e-5 / 1.39e-5 ->
            // 1.71e-5 / 1.36e-5, profiles/r03a/ab_multi_defer.txt.)
            nv_ic = 0; nv_nev = 1; nv_m = 1;
            croot = (p0c + cb) / 2.0f;
            st = (KIND == 1) ? ST_NEVILL0 : ST_NEVILL;
            sub = 0;
        } else if (st == ST_NEVILL0) {
            if (G == 1) { if (sub == 0) { p0d = val; sub = 1; } else { db = val; st = ST_NEVILL; } }
            else { p0d = v0; db = v1; st = ST_NEVILL; }
        } else if (st == ST_NEVILL) {
            // NEVILL, statement by statement (surfa.f:17-83).  One evaluation per pass: del3 = Delta(c3) has just
            // been computed by every lane of the team (v0); what follows runs up to the next evaluation.
            // SIGN(1., x): the reference's NaNs are positive when they come out of the secular function (negnan
            // above) or are passed on by a subtraction, negative when a subtraction creates them (inf - inf: the
            // x86 default NaN).  No fused multiply-adds: the reference is built without contraction.
#pragma clang fp contract(off)
            auto sg = [](float x) { return (signbit(x) && !(x != x)) ? -1.0f : 1.0f; };
            auto sgsub = [&](float a, float bq) {
                const float r = a - bq;
                if (r != r) return ((a != a) || (bq != bq)) ? 1.0f : -1.0f;
                return signbit(r) ? -1.0f : 1.0f;
            };
            float c1 = p0c, c2 = cb, d1 = p0d, d2 = db, c3 = croot;
            const float d3 = v0;
            nv_ic = nv_ic + 1;
            bool fin_ = false;
            if (!(nv_ic < 50)) fatal = true;                   // TOO MANY CYCLES: lstop, calcul.f:172-189 -> 9999
            else {
                bool bis;
                if (c1 - c3 <= 0.0f) bis = (c2 - c3 <= 0.0f);  // 777 / 1320 / 1330: arithmetic IFs, a NaN
                else bis = !(c2 - c3 < 0.0f);                  // expression takes the third label
                if (!bis) {
                    const float s13s = sgsub(d1, d3), s32s = sgsub(d3, d2);            // label 1000
                    if (sg(d3) * sg(d1) <= 0.0f) { c2 = c3; d2 = d3; } else { c1 = c3; d1 = d3; }
                    if (fabsf(c1 - c2) - 0.1e-5f <= 0.0f) fin_ = true;                 // 1444: accur1
                    else {
                        if (s13s != s32s) nv_nev = 0;
                        const float ss1 = fabsf(d1), s1 = 0.1f * ss1, ss2 = fabsf(d2), s2 = 0.1f * ss2;
                        if (s1 > ss2 || s2 > ss1) bis = true;
                        else if (nv_nev == 0) bis = true;
                        else {
                            int m = nv_m;
                            if (nv_nev == 2) { nvx[m + 1] = c3; nvy[m + 1] = d3; }     // 1350
                            else { nvx[1] = c1; nvy[1] = d1; nvx[2] = c2; nvy[2] = d2; m = 1; }
                            const float ym1 = nvy[m + 1];
                            for (int kk = 1; kk <= m; ++kk) {                           // 1355-1360
                                const int jn = m - kk + 1;
                                const float yj = nvy[jn];
                                if (fabsf(ym1 - yj) <= 0.1e-7f) { bis = true; break; }
                                nvx[jn] = (-yj * nvx[jn + 1] + ym1 * nvx[jn]) / (ym1 - yj);
                            }
                            if (!bis) { c3 = nvx[1]; nv_nev = 2; nv_m = (m + 1 > 10) ? 10 : m + 1; }   // 21
                        }
                    }
                }
                if (!fin_ && bis) { c3 = (c1 + c2) / 2.0f; nv_nev = 1; nv_m = 1; }      // 1344
                p0c = c1; p0d = d1; cb = c2; db = d2; croot = c3;
            }
            if (fin_) {                                        // label 20: cc = c3
                if (croot <= W_B(mm_frozen - 1)) {             // calcul.f:191
                    if (want_ratio) { st = ST_ELLIP; sub = 0; } else solved = true;
                } else failed = true;
            }
        } else if (st == ST_REFINE) {
            // new bracket + one more known point next to it (for the final 3-point step)
            float tc, td;
            bool tok = true;
            float uc = 0.0f, ud = 0.0f;                        // a fourth point, on the other side where there is one
            bool uok = false;
            const float oa = p0c, oda = p0d, ob = cb, odb = db;
            const bool oaok = p0ok;
            if (fl >= 0) {
                if (fl != tbase) p0ok = true;                  // low end replaced by a frozen-mmax point
                p0c = e_pc; p0d = e_pd; cb = e_c; db = e_d;
                if (fl < lastl) { tc = e_nc; td = e_nd; } else { tc = ob; td = odb; }
                if (fl >= tbase + 2) { uc = e_ppc; ud = e_ppd; uok = true; }              // the point below the low end
                else if (fl == tbase + 1) { uc = oa; ud = oda; uok = oaok; }              // ... which may be the old low end
                else if (fl + 2 <= lastl) { uc = e_n2c; ud = e_n2d; uok = true; }         // first interval: two points above
                else if (fl + 1 == lastl) { uc = ob; ud = odb; uok = true; }
            } else {
                p0c = l_c; p0d = l_d; p0ok = true;
                if (G > 1) { tc = pl_c; td = pl_d; } else { tc = oa; td = oda; tok = oaok; }
                if (G > 2) { uc = pl2_c; ud = pl2_d; uok = true; }
            }
            {
                // Candidate root by inverse quadratic interpolation through the bracket ends and the
                // neighbouring point (offsets from the low end keep fp32 exact enough) and by the
                // secant.  Accept when the bracket is narrow AND the two agree (the secular function
                // is locally smooth, so the 3-point estimate is far better than their difference), or
                // when the bracket has shrunk to NEVILL's own tolerance (surfa.f:10,44).  Otherwise
                // subdivide again: near osculating modes Delta(c) is strongly curved and only a tight
                // bracket pins the root the reference finds.
                const float w = cb - p0c, sx = tc - p0c;
                // The estimates below are ratios of PRODUCTS of function values, formed with v_rcp_f32 - which returns 0 for
                // arguments beyond 2^126 and flushes results below 2^-126: with |Delta| ~ 1e19 .. 1e20 (water layer over soft
                // sediments at periods of a few seconds) the denominators (f1 - f0)(f1 - f2) ~ 1e38 .. 1e40 turned BOTH
                // three-point estimates into exactly 0, they "agreed", and the bracket's low end - a subdivision point up to
                // 1e-3 km/s from the root - was returned as the root (r04 soak: 1e-5 of random stacks, all team sizes below 16
                // lanes).  The values are brought to order one by a common power of two first (exact).
                const float fm = fmaxf(fabsf(p0d), fabsf(db));
                const int fex = (fm > 0.0f && fin(fm)) ? __builtin_amdgcn_frexp_expf(fm) : 0;
#ifdef SD_NO_FSCALE
                const float f0 = p0d, f1 = db, f2 = td;
#else
                const float f0 = ldexpf(p0d, -fex), f1 = ldexpf(db, -fex), f2 = ldexpf(td, -fex);
                ud = ldexpf(ud, -fex);
#endif
                auto qt = [](float a, float bq) { return a * __builtin_amdgcn_rcpf(bq); };
                float ts = qt(-f0 * w, f1 - f0);
                float t = qt(w * (f0 * f2), (f1 - f0) * (f1 - f2)) + qt(sx * (f0 * f1), (f2 - f0) * (f2 - f1));
                if (!(ts >= 0.0f)) ts = 0.0f;
                if (!(ts <= w)) ts = w;
                if (!p0ok) ts = 0.5f * w;                         // magnitudes not comparable: bisect
                const bool inside = p0ok && tok && (t >= 0.0f) && (t <= w);
                // Next to osculating modes the secular function bends sharply inside a bracket of a few 1e-4 km/s, and
                // inverse interpolation - every three-point estimate alike - then misses the root by several 1e-6 while
                // the estimates agree with each other (ragged fixture, 60 s: c off by 1.8e-6, which |dlnU/dlnc| = 560
                // turns into 1e-3 of U).  Such a bracket is subdivided again: the slope towards a neighbouring point must
                // be within a quarter of the slope across the bracket (ordinary brackets: a fraction of a percent).
                auto bends = [&](float xs, float fs) {                 // slopes compared without forming them
                    const float dm = f1 - f0;
                    const float dn = (xs > w) ? fs - f1 : f0 - fs, hn = (xs > w) ? xs - w : 0.0f - xs;
                    return !(fabsf(dn * w - dm * hn) <= 0.25f * fabsf(dm * hn));
                };
                const bool smooth = inside && !bends(sx, f2);
                bool agree = smooth && (fabsf(t - ts) <= A.atol);
                if (smooth && !agree && uok && !bends(uc - p0c, ud)) {
                    // The secant is only a second-order check: with a dozen evaluated points across the bracket a
                    // SECOND three-point estimate (same bracket ends, the neighbour on the other side) is the sharper
                    // witness - two cubically accurate estimates that agree to atol pin the root as well as another
                    // pass of subdivision would (deep stacks: half of the periods took that extra pass for the
                    // secant's sake, 28 of a stack's 95 passes instead of 19).
                    const float sx2 = uc - p0c, f3 = ud;
                    const float t2 = qt(w * (f0 * f3), (f1 - f0) * (f1 - f3)) + qt(sx2 * (f0 * f1), (f3 - f0) * (f3 - f1));
                    agree = (t2 >= 0.0f) && (t2 <= w) && (fabsf(t - t2) <= A.atol);
                }
                ++passes;                                          // hard bound: fp32 cannot resolve <1 ulp
                // ... and never across the KINK at the cut-off: a layer of finite thickness enters the secular function through
                // even functions of its vertical wavenumbers (cos, sin(x)/x, x sin(x)) and is smooth where c passes its P or S
                // velocity, but the half-space closure is LINEAR in sqrt|1 - c^2/b^2|: at the S velocity of the working stack's
                // half space the function behaves like sqrt|x|.  With that velocity among the points an estimate is built
                // from, all three-point estimates miss alike and still agree (ragged fixture, 60 s: root 5e-6 km/s below the
                // half space's S velocity; teams of 16 lanes accepted a 3.5e-5 bracket and came out 8e-6 off, 1e-3 of U at
                // |dlnU/dlnc| = 485; NEVILL - and teams of any other size - within 6e-7).  A bracket that reaches up to the
                // cut-off is subdivided until it no longer does (roots lie below it, calcul.f:191).
                bool accept = !(w > 1.0e-6f) || passes > 64;
                if (!accept && !(w > A.wtol) && agree) {
#ifndef SD_NO_KINK
                    accept = !(W_B(mm_frozen - 1) <= cb * 1.000001f);
#ifdef SD_COUNT_KINK
                    if (j == 0 && A.amb_count && !accept) atomicAdd(A.amb_count + 1, 1);   // (developer statistics: refine passes added by the kink test)
#endif
#else
                    accept = true;
#endif
                }
                if (accept) {
                    croot = p0c + (inside ? t : ts);
                    if (p0c >= 16.0f) fatal = true;                // NEVILL's 50 cycles, see above
                    else if (croot <= W_B(mm_frozen - 1)) {        // calcul.f:191
                        if (want_ratio) {
                            if (OVERLAP && k + 1 < P) {
                                // snapshot the layers the ellipticity recursion reads, then move on:
                                // the next build overwrites exactly these (first mm_frozen) layers
                                for (int i = j; i < mm_frozen; i += G) {
#pragma unroll
                                    for (int f = 0; f < NFK; ++f)
                                        wq2[i * LS + f * S] = wq[i * LS + f * S];
                                }
                                ell_pend = true; ell_k = k; ell_mm = mm_frozen; ell_c = croot; ell_T = T;
                                solved = true;
                            } else { st = ST_ELLIP; sub = 0; }
                        }
                        else solved = true;
                    } else failed = true;
                }
            }
        } else if (st == ST_ELLIP) {
            if ((__ballot(ell_amb) & tmask) != 0ull) ell_flag = true;
            if (G == 1) {
                if (sub == 0) { r12 = val; sub = 1; }
                else { r12 = 0.5f * val / r12; solved = true; }
            } else {
                r12 = 0.5f * v1 / v0;                          // surfa.f:363
                solved = true;
            }
        }
        if (!EXACT && defer) {
            if (j == 0) A.fb_list[atomicAdd(A.fb_count, 1)] = (int)tg;
            defer = false; st = ST_DONE; ell_pend = false; solved = false; failed = false; fatal = false;
        }
        if (fatal) {
            nsolved = 0; k = 0; status = SURFDISP_NUMERIC; st = ST_DONE; ell_pend = false; solved = false; failed = false;
        }
        if (failed) {
            status = (k == 0) ? SURFDISP_NOROOT : SURFDISP_PARTIAL;
            st = ST_DONE;
        }
        if (LOCK) {
            // Lock step of the teams of a wavefront: with many teams per wavefront (16 of four lanes) SOME team is at its
            // refine pass or at the end of a period in nearly every pass, and the wavefront then runs those blocks for one
            // team in sixteen - team decisions and end-of-period block were 35 % of a ten-layer wavefront's time.  Here a team
            // with its bracket waits until no team of the wavefront scans any more, all refine together, and a team with its
            // root waits until all have theirs: one refine pass and one end-of-period block per period and wavefront.  A
            // waiting team evaluates nothing; each team's own sequence of evaluations - and so every result - is unchanged.
            if (solved) { st = ST_WEND; solved = false; }
            if (!__any(st == ST_SCAN) && st == ST_WREF) {
                st = ST_REFINE;
#ifndef SD_NO_PHASEMULTI
                if (bracket_phase(p0c, cb, mm_frozen) > A.phimulti) nevill_start();      // (see the bracket branch of the scan)
#endif
            }
            if (!__any(st == ST_SCAN || st == ST_WREF || st == ST_REFINE || st == ST_NEVILL || st == ST_NEVILL0 || st == ST_ELLIP) && st == ST_WEND)
                solved = true;
        }
#ifdef SD_WAVECLOCK
        const unsigned long long wb0 = __builtin_readcyclecounter();
#endif
        if (solved) {
            if (j == 0) {
                A.c[(size_t)k * B + b] = croot;                // period-major: coalesced across teams
                if (want_ratio && !ell_pend) A.ratio[(size_t)k * B + b] = r12;
                if (A.hist) A.hist[(size_t)k * B + b] = EXACT ? -1 : (nflat_cur | (mm_frozen << 16) | (ell_flag ? 0x40000000 : 0));
            }
            ell_flag = false;
            nsolved = ++k;
            if (INDEP || k >= P) { st = ST_DONE; }
            else {
                T = A.per[k];
                mm_carry = mm_frozen;                          // mmax left by the last idrop=0 trial
#ifdef SD_DEBUG_TRIALS
                if (A.B == 1 && j == 0) printf("period %d starts: previous root %.7f, layers rebuilt for T = %.4f: %d of %d\n", k, croot, T, mm_carry, n);
#endif
                build(mm_carry);
                nflat_cur = mm_carry;
                b1top = W_B(0);
                p0c = 0.90f * croot;                           // calcul.f:143
                p0d = 0.0f; p0mm = 0; p0ok = false; first = true; passes = 0;
                coarse = false; fine_left = 1; q0ok = false;
                st = ST_SCAN;
                nodrop = no_drop_possible(p0c);
                defer = entry_overflow_risk(p0c);              // acted on at the end of the next pass
                if (CERT) coarse = nodrop;
            }
        }
#ifdef SD_WAVECLOCK
        wcyc_build += __builtin_readcyclecounter() - wb0;      // store of the root, next period's set-up, stack rebuild
#endif
    }
#ifdef SD_WAVECLOCK
    if (!EXACT && A.wclk && (threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (SD_PHASE_BLOCK / 64) + (threadIdx.x >> 6);
        unsigned long long *o = A.wclk + 16 * w;
        o[0] = wclk0; o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = __builtin_readcyclecounter() - wcyc0; o[3] = wcyc_eval;
        o[4] = wcyc_build; o[5] = wpasses; o[6] = wcyc_pre;
        o[7] = wn_scan; o[8] = wn_refine; o[9] = wn_nevill; o[10] = wn_ellip; o[11] = wn_idle;
        o[12] = wtrip; o[13] = wlanelayers; o[14] = wlanes; o[15] = 0;
    }
#endif
    if (INDEP) {
        // a failed period zeroes itself and every later one (calcul.f:203-219): the first failing
        // period index is reduced over the stack's teams; the finish kernel applies it
        if (team_valid && j == 0 && n >= 2 && status != SURFDISP_OK) {
            A.c[(size_t)k_own * B + b] = 0.0f;
            // -1: the secular function left the fp32 range somewhere in this stack (the finish kernel then
            // reports SURFDISP_NUMERIC, as the faithful mode does)
            atomicMin(&A.nsolved[b], status == SURFDISP_NUMERIC ? -1 : k_own);
        }
        return;
    }
    if (team_valid && j == 0) {
        if (n < 2) status = SURFDISP_BADMODEL;
        for (int q = nsolved; q < P; ++q) A.c[(size_t)q * B + b] = 0.0f;
        A.nsolved[b] = nsolved;
        if (A.status) A.status[b] = status;
    }
#undef M_AT
}

template <int KIND, int G, bool INDEP, bool FAST = false, bool EXACT = false>
__global__ __launch_bounds__(SD_PHASE_BLOCK) void surfdisp_phase_kernel(PhaseArgs A)
{
    phase_body<KIND, G, INDEP, FAST, EXACT>(A);
}

// ============================================================================= K1b: ellipticity
// The Rayleigh ellipticity of every solved (stack, period): 0.5 bb1(e3) / bb1(e2), the two extra recursions of
// DLTAR4(mup = 2) at the root (calcul.f:195, surfa.f:202-208,360-363), one lane per (stack, period) - full lanes, where the
// root search could give them two lanes of a team riding in the next period's scan pass and a second LDS working stack to
// read from (which cost deep stacks a workgroup per CU: 16 384 x L64, teams of 16: 51 KB -> 26 KB of LDS, three -> four
// workgroups per CU = the launch's 4 096 wavefronts in ONE round).
// The reference evaluates them on the working stack as phase 1 left it (COMMON /d/): only the first nflat layers are
// refreshed per period, the layer nflat - 1 in half-space form, deeper ones keep an earlier period's values - and the
// frozen mmax may reach into those.  The root search records (nflat, mmax) per solved period; this kernel replays the
// history: layer i carries the values of the LAST period k' <= k whose rebuild covered it (k' falls monotonically as i
// grows), with the same expressions the rebuild uses (layer_derive, rcp_nr), then steps both start vectors through
// ray_step / ray_close - the root search's own functions.
// r04: where the closure of either pass is the remainder of a cancellation (|value| below A.ell_ambig of its terms: soft
// sediments at c < 0.5 km/s, where the production recursion's ellipticity was good to 3e-4 .. 3e-3 only and cost 1e-4 of the
// group velocity) BOTH passes are evaluated again with the reference's own arithmetic (DLTAR4 statement by statement on the
// replayed working stack) - the value calcul.f:195 itself forms.  only_flagged: the root search computed the ellipticities
// itself (two-lane teams) and marked the (stack, period) pairs that need this (bit 30 of the history word).
__global__ __launch_bounds__(256) void surfdisp_ellip_kernel(EllipArgs A)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int B = A.B, P = A.P;
    if (idx >= (size_t)B * P) return;
    const int b = (int)(idx % B), k = (int)(idx / B);       // a wavefront = 64 stacks, one period: coalesced SoA reads
    const int n = A.nl[b];
    if (n < 2 || k >= A.nsolved[b]) return;                 // (the finish kernel writes 0 for unsolved periods)
    const int hk = A.hist[idx];
    if (hk < 0) return;                                      // computed by the exact fallback kernel itself
    if (A.only_flagged && !(hk & 0x40000000)) return;
    const int mmf = (hk >> 16) & 0x3fff;
    const size_t fs = (size_t)A.Lmax * B;
    const float c = A.c[idx], T = A.per[k];
    // the working stack as phase 1 left it, layer by layer (i = 0, 1, ... in order): the period whose rebuild last refreshed
    // layer i, and that rebuild's extent
    struct Replay { int kk, nflat; float lnT; };
    const Replay r0{k, hk & 0xffff, logf(1.0f / T)};
    auto layer = [&](int i, Replay &r) -> LayerV {
        while (r.nflat <= i && r.kk > 0) {                   // an earlier rebuild: period 0 refreshed all n layers
            --r.kk;
            const int h = A.hist[(size_t)r.kk * B + b];
            r.nflat = (h < 0) ? n : (h & 0xffff);
            r.lnT = logf(1.0f / A.per[r.kk]);
        }
        return layer_derive(layer_load(A.mdl, fs, (size_t)i * B + b), r.lnT, i == r.nflat - 1);
    };
    const int last = mmf - 1;
    float v2 = 0.0f, v3 = 0.0f;
    bool exact = A.only_flagged != 0;
    if (!exact) {
        const RTrial t = ray_trial(c, T);
        int kk = k;                                          // period whose rebuild last refreshed the layer at hand
        int nflat = hk & 0xffff;
        float lnT = r0.lnT;
        RState s2{}, s3{};
        float phi = 0.0f, rho_prev = 0.0f;
        RLyr y{};
        float rho_i = 0.0f;
        for (int i = 0; i <= last; ++i) {
            while (nflat <= i && kk > 0) {                   // an earlier rebuild: period 0 refreshed all n layers
                --kk;
                const int h = A.hist[(size_t)kk * B + b];
                nflat = (h < 0) ? n : (h & 0xffff);
                lnT = logf(1.0f / A.per[kk]);
            }
            const LayerV v = layer_derive(layer_load(A.mdl, fs, (size_t)i * B + b), lnT, i == nflat - 1);
            rho_prev = rho_i;
            rho_i = v.rho;
            y.sv = v.b; y.d = v.d;
            y.ia2 = rcp_nr(v.a * v.a);
            y.ib2 = (v.b > 0.0f) ? rcp_nr(v.b * v.b) : 0.0f;
            y.rat = (i > 0) ? rho_prev * rcp_nr(rho_i) : 0.0f;
            if (i == 0) {
                const float irho0 = rcp_nr(rho_i);
                s2 = ray_start(t, 2, irho0);
                s3 = ray_start(t, 3, irho0);
                if (last >= 1) { ray_step<true>(s2, t, y, 2, phi); ray_step<true>(s3, t, y, 3, phi); }
            } else if (i < last) {
                ray_step<false>(s2, t, y, 2, phi);
                ray_step<false>(s3, t, y, 3, phi);
            }
        }
        const float rp = (last >= 1) ? rho_prev : 0.0f;
        float m2 = 0.0f, m3 = 0.0f;
        v2 = ray_close(s2, t, y, rho_i, rp, 2, &m2); v3 = ray_close(s3, t, y, rho_i, rp, 3, &m3);
        // ... or where c is far below a layer's S velocity: g = 2 b^2 / c^2 of that layer is in the hundreds and the recursion's
        // u1 = g^2 b1 + 2 g h3 - h5 (as the reference's a11 .. a51, differently arranged) cancels inside the LAYER steps - soft
        // sediments over rock, c = 0.2 .. 0.4 km/s: the production ellipticity came out 3e-4 .. 4e-3 off the reference's (1e-4
        // .. 4e-4 of U) with an unremarkable closure.  The stack's largest flattened S velocity is the prep kernel's statistic.
        const float b2max = A.ovf ? 0.5f * __expf(0.25f * A.ovf[2 * (size_t)B + b]) : 0.0f;
        exact = (A.ell_ambig < 0.0f) ||
                ((A.ell_ambig > 0.0f) && (fabsf(v2) < A.ell_ambig * m2 || fabsf(v3) < A.ell_ambig * m3 || 2.0f * b2max > A.ell_gmax * c * c));
    }
    if (exact) {
        Replay r = r0;
        v2 = delta_rayleigh_ref_gen([&](int m) { const LayerV v = layer(m, r); return RefLyr{v.a, v.b, v.rho, v.d}; }, mmf, c, T, 2);
        r = r0;
        v3 = delta_rayleigh_ref_gen([&](int m) { const LayerV v = layer(m, r); return RefLyr{v.a, v.b, v.rho, v.d}; }, mmf, c, T, 3);
        if (A.amb_count) atomicAdd(A.amb_count + 1, 1);
    }
    A.ratio[idx] = 0.5f * v3 / v2;                           // surfa.f:363
}

// ================================================================================== K2: group
// sublayer bookkeeping shared by Rayleigh and Love (surfa.f:781-822 / 412-446): layers are split
// into ndiv equal sublayers with identical properties, so nothing is materialised -- a layer is
// visited with a repeat count.
struct Drop { int hs_layer; int nreg_hs; };

template <int KIND>
SD_HD __forceinline__ Drop drop_group(const float *__restrict__ mdl, size_t fs, int B, int b,
                                           int n, float lnT, float c, float T, int ndiv, bool water,
                                           float div)
{
    // surfa.f:853-866 (Rayleigh: compares a then b across the cut) / 475-487 (Love: b only).
    // Cuts can only happen at layer boundaries because sublayers of one layer are identical.
    const float dmax = FACT * T * c;
    float sum = 0.0f;
    Drop r; r.hs_layer = n - 1; r.nreg_hs = 0;
    LayerRaw nraw = layer_load(mdl, fs, (size_t)b);
    for (int jl = 0; jl < n; ++jl) {
        const LayerRaw raw = nraw;
        if (jl + 1 < n) nraw = layer_load(mdl, fs, (size_t)(jl + 1) * B + b);
        const LayerV v = layer_derive(raw, lnT, jl == n - 1);
        if (!(c < v.b)) continue;
        if (jl == n - 1) break;                                   // ii == mmax: keep the true half space
        const int nsub = (jl == 0 && water) ? 1 : ndiv;
        const float dsub = (ndiv > 1 && !(jl == 0 && water)) ? v.d / div : v.d;
        for (int s = 0; s < nsub; ++s) sum = sum + dsub;
        if (!(sum > dmax)) continue;
        const LayerV nx = layer_derive(nraw, lnT, jl + 1 == n - 1);
        bool lower, equal;
        if (KIND == 2) {
            lower = (nx.a < v.a) || (nx.a == v.a && nx.b < v.b);
            equal = (nx.a == v.a) && (nx.b == v.b);
        } else {
            lower = nx.b < v.b;
            equal = nx.b == v.b;
        }
        if (lower) { r.hs_layer = jl; r.nreg_hs = nsub - 1; break; }   // label 902: mmax = ii
        if (equal) continue;
        r.hs_layer = jl + 1; r.nreg_hs = 0; break;                     // label 90009: mmax = ii+1
    }
    return r;
}

// Analytic partial derivatives of the phase velocity (the quantities REIGEN / LEIGEN form from their
// energy integrals and leave in COMMON /rar1/: surfa.f:1130-1135, 1180-1184, 1204-1207 / 509-512, 561-565, 581-583).
// The eigenproblem is solved for the attenuation-dispersed, earth-flattened layer values; the
// caller's derivatives are with respect to its own Vs, Vp, rho, so each layer carries the chain
// factors of calcul.f:122-126 and flat1.f:44-62:
//   b = b_ref (1 + qsq) f,  a = a_ref (1 + qsq 4/3 b_ref^2/a_ref^2) f,  rho = rho_ref r
// (KOut::raw, SURFDISP_KERN_REFCOORD: unit chain factors - the partials with respect to the flattened, attenuated layer
// values themselves, i.e. the reference's own numbers summed over a layer's sublayers; the water layer's share, which the
// reference does not form, is left out: what tests/golden/ref_partials.npz pins.)
//
// Where they go (r04).  A unit = one (stack, period) = one lane.  Every layer's share is written ONCE, unscaled, as soon as
// the sweep has it (a plain store: no zero fill, no read-modify-write; the share of the layer that also holds the half
// space is kept in three registers until the half-space terms are known); the common factor 1 / (dL/dk) - known only after
// the last layer - and the index of the deepest layer written go to two per-unit words, and the consumer applies them:
//   * scratch route (kernels workspace): layer-major scratch [3][Lmax][P*B] - the lanes of a wavefront (consecutive stacks,
//     one period) store consecutive words; surfdisp_kern_transpose_kernel reads it back through an LDS tile, multiplies by
//     the unit's factor, writes zeros for layers below the unit's half space / unsolved units, and stores whole rows of the
//     caller's [B][P][Lmax] arrays.  HBM traffic: scratch written once, read once, rows written once (r03: zero fill +
//     accumulate + scale, each a read-modify-write of the scratch, + the transposition: 7.2 x the rows).
//   * direct route (a workspace without the scratch): the same stores go to the caller's rows (stride 1) and the lane
//     itself scales / zero-fills its rows at the end - the same products, so both routes agree bit for bit.
struct KOut {
    float *b;                 // this unit's entry of layer 0 in the dc/dVs array; nullptr: no partials
    size_t stride;            // words between consecutive layers (uniform)
    ptrdiff_t off_a, off_r;   // BYTE offsets of the dc/dVp and dc/drho entries from the dc/dVs entry (uniform; 0: not wanted)
    int raw;                  // reference coordinates (see above)
    SD_HD void put(int i, float vb, float va, float vr) const
    {
        float *q = b + (size_t)i * stride;
        *q = vb;
        if (off_a) *reinterpret_cast<float *>(reinterpret_cast<char *>(q) + off_a) = va;
        if (off_r) *reinterpret_cast<float *>(reinterpret_cast<char *>(q) + off_r) = vr;
    }
};
struct K3 { float b, a, r; };
struct Chain { float dbdb, dadb, dada, rfac; };
SD_HD __forceinline__ Chain chain_of(const LayerRaw &r, float lnT, bool is_halfspace, bool raw)
{
    if (raw) return Chain{1.0f, 0.0f, 1.0f, 1.0f};
    const float qsq = r.qs * lnT / PI_REF;
    const float qpq = qsq * 1.33333333f * (r.b_ref * r.b_ref) / (r.a_ref * r.a_ref);
    const float vfac = is_halfspace ? r.hsf : r.dif;
    Chain ch;
    ch.dbdb = (1.0f + qsq) * vfac;
    ch.dadb = 2.66666667f * qsq * (r.b_ref / r.a_ref) * vfac;
    ch.dada = (1.0f - qpq) * vfac;
    ch.rfac = is_halfspace ? r.hsr : r.qqq;
    return ch;
}
// One layer's share from its six energy integrals (surfa.f:1110-1121, summed over the layer's sublayers) -> dL/d(lambda,
// mu, rho) (surfa.f:1130-1132) -> dc/d(b, a, rho) of the flattened layer, still to be divided by dL/dk (surfa.f:1133-1135)
// -> the caller's layer.  Everything that depends on the layer alone - 2 rho b c/k and the chain factors - is folded into
// four coefficients when the layer is entered (KC: the layer's raw values need not stay in registers across its RK4 steps).
// fp32: the integrals are fp32 Boole sums; 1e-7 of their largest term is 1e-6 of a period's largest partial.
struct KC { float b1, b2, a, r; };
SD_HD __forceinline__ KC kern_coef(const LayerRaw &raw, const LayerV &v, float lnT, bool is_halfspace, float cw, bool rawc)
{
    const Chain ch = chain_of(raw, lnT, is_halfspace, rawc);
    const float rc2 = 2.0f * v.rho * cw;                     // 2 rho c / k
    return KC{rc2 * v.b * ch.dbdb, rc2 * v.a * ch.dadb, rc2 * v.a * ch.dada, cw * ch.rfac / v.rho};
}
// KR: the type the six sums are carried and combined in.  dc/drho is the difference of the kinetic and the two elastic
// terms, which cancel to ~1e-3 of their size (equipartition): combined in fp32 it came out 6e-4 of a period's peak off
// the reference's value (tests/golden/ref_partials.npz) - the reference carries them DOUBLE PRECISION (surfa.f:717-722).
#ifndef SD_KERN_REAL
#define SD_KERN_REAL double
#endif
typedef SD_KERN_REAL KR;
SD_HD __forceinline__ K3 kern_layer_rayleigh(const KC &kc, float rho, float xlamb, float xmu, float wvno, float wvnosq,
                                             float omegsq, KR dmmr, KR dmmz, KR drsz, KR dzsr, KR smmz, KR smmr)
{
    const KR two = 2, k2 = wvnosq, tk = 2.0f * wvno;
    const KR dldl = -k2 * dmmr + tk * drsz - smmz;
    const KR dldm = -k2 * (two * dmmr + dmmz) - tk * dzsr - (two * smmz + smmr);
    const KR dldr = (KR)omegsq * (dmmr + dmmz);
    return K3{(float)((KR)kc.b1 * (dldm - two * dldl) + (KR)kc.b2 * dldl), (float)((KR)kc.a * dldl),
              (float)((KR)kc.r * ((KR)rho * dldr + (KR)xlamb * dldl + (KR)xmu * dldm))};
}

// ---- Rayleigh, surfa.f:714-1192 ---------------------------------------------------------------
struct RCoef { float a12, a13, a21, a24, a31, a34, a42, a43, ddz; };

struct RInt {                       // energy integrals, fp64 accumulators (reference: fp32 sumi*)
    double i0, i1, i2;
};

// One RK4 step of a constant-coefficient linear system is a fixed linear map.  With the state split
// as x = (ur, tz), w = (uz, tr) the system matrix is block anti-diagonal (x' = M1 w, w' = M2 x), so
//   P = I + k1 A + k2 A^2 + k3 A^3 + k4 A^4 = [[I + k2 N1 + k4 N1^2,  k1 M1 + k3 N1 M1],
//                                             [k1 M2 + k3 N2 M2,  I + k2 N2 + k4 N2^2]],
// N1 = M1 M2, N2 = M2 M1.  k1..k4 are formed from the reference's fp32 weights exactly as its stage
// recursion combines them (surfa.f:764-771, 955-968), so P reproduces the reference's step to fp64
// rounding at a quarter of the arithmetic; all sublayers of a layer share it.
struct M2x2 { double a, b, c, d; };     // [[a b],[c d]]
SD_HD __forceinline__ M2x2 mm(const M2x2 &x, const M2x2 &y)
{
#pragma clang fp contract(off)
    M2x2 r;
    r.a = fma(x.a, y.a, x.b * y.c); r.b = fma(x.a, y.b, x.b * y.d);
    r.c = fma(x.c, y.a, x.d * y.c); r.d = fma(x.c, y.b, x.d * y.d);
    return r;
}
// x y + u v, one multiply and three fma per element
SD_HD __forceinline__ M2x2 mm2(const M2x2 &x, const M2x2 &y, const M2x2 &u, const M2x2 &v)
{
#pragma clang fp contract(off)
    M2x2 r;
    r.a = fma(x.a, y.a, fma(x.b, y.c, fma(u.a, v.a, u.b * v.c))); r.b = fma(x.a, y.b, fma(x.b, y.d, fma(u.a, v.b, u.b * v.d)));
    r.c = fma(x.c, y.a, fma(x.d, y.c, fma(u.c, v.a, u.d * v.c))); r.d = fma(x.c, y.b, fma(x.d, y.d, fma(u.c, v.b, u.d * v.d)));
    return r;
}
SD_HD __forceinline__ M2x2 madd(const M2x2 &x, const M2x2 &y)
{
    M2x2 r; r.a = x.a + y.a; r.b = x.b + y.b; r.c = x.c + y.c; r.d = x.d + y.d; return r;
}
struct RProp { M2x2 p11, p12, p21, p22; };

SD_HD __forceinline__ RProp make_prop(const RCoef &q)
{
#pragma clang fp contract(off)
    const double wh = (double)(0.5f * q.ddz), w1 = (double)(1.0f * q.ddz);
    const double t6 = (double)((1.0f / 6.0f) * q.ddz), t3 = (double)((1.0f / 3.0f) * q.ddz);
    const double k1 = (t6 + t3) + (t3 + t6);
    const double k2 = (t3 * wh + t3 * wh) + t6 * w1;
    const double k3 = t3 * wh * wh + t6 * w1 * wh;
    const double k4 = t6 * w1 * wh * wh;
    const M2x2 m1 = {(double)q.a31, (double)q.a34, (double)q.a21, (double)q.a24};
    const M2x2 m2 = {(double)q.a13, (double)q.a12, (double)q.a43, (double)q.a42};
    // One block of P at a time (register pressure: this kernel sits at the 168-VGPR boundary of three
    // wavefronts per SIMD).  N^2 of a 2x2 matrix is tr(N) N - det(N) I (Cayley-Hamilton), so
    //   I + k2 N + k4 N^2 = (1 - k4 det) I + (k2 + k4 tr) N      and      k1 M + k3 N M = (k1 I + k3 N) M
    // need no explicit N^2 and N M products (fp64 runs at half rate here).
    RProp P;
    auto diag_block = [&](const M2x2 &nn) -> M2x2 {
        const double tr = nn.a + nn.d;
        const double det = fma(nn.a, nn.d, -(nn.b * nn.c));
        const double al = fma(k4, tr, k2), be = fma(-k4, det, 1.0);
        return M2x2{fma(al, nn.a, be), al * nn.b, al * nn.c, fma(al, nn.d, be)};
    };
    auto off_block = [&](const M2x2 &nn, const M2x2 &m) -> M2x2 {
        const M2x2 q = {fma(k3, nn.a, k1), k3 * nn.b, k3 * nn.c, fma(k3, nn.d, k1)};
        return mm(q, m);
    };
    {
        const M2x2 n1 = mm(m1, m2);
        P.p11 = diag_block(n1);
        P.p12 = off_block(n1, m1);
    }
    {
        const M2x2 n2 = mm(m2, m1);
        P.p22 = diag_block(n2);
        P.p21 = off_block(n2, m2);
    }
    return P;
}
SD_HD __forceinline__ RProp prop_sq(const RProp &P)
{
    RProp Q;
    Q.p11 = mm2(P.p11, P.p11, P.p12, P.p21);
    Q.p12 = mm2(P.p11, P.p12, P.p12, P.p22);
    Q.p21 = mm2(P.p21, P.p11, P.p22, P.p21);
    Q.p22 = mm2(P.p21, P.p12, P.p22, P.p22);
    return Q;
}
// v = (ur, uz, tz, tr)
SD_HD __forceinline__ void prop_apply(const RProp &P, double v[4])
{
#pragma clang fp contract(off)
    const double ur = v[0], uz = v[1], tz = v[2], tr = v[3];
    // one multiply and three fused multiply-adds per row (fp64 runs at half rate: the applications of P
    // are most of this kernel's time); every sweep uses this same routine, so they stay bit-consistent
    v[0] = fma(P.p11.a, ur, fma(P.p11.b, tz, fma(P.p12.a, uz, P.p12.b * tr)));
    v[2] = fma(P.p11.c, ur, fma(P.p11.d, tz, fma(P.p12.c, uz, P.p12.d * tr)));
    v[1] = fma(P.p21.a, ur, fma(P.p21.b, tz, fma(P.p22.a, uz, P.p22.b * tr)));
    v[3] = fma(P.p21.c, ur, fma(P.p21.d, tz, fma(P.p22.c, uz, P.p22.d * tr)));
}

// integrate both solutions from the half space to the surface.  INTEG = false: only the surface
// values are wanted, so each sublayer is one application of P^4.  INTEG = true: step by step, and
// accumulate the Boole energy integrals of the combined solution (xnorm*y + z)/bb
// (surfa.f:1087-1129).  The two sweeps agree to fp64 rounding (~1e-15 relative), far inside what the
// ~1e6 cancellation of the combination needs.
// MODE 0: surface values only, one application of P^4 per sublayer (fast path);
// MODE 1: surface values only, four applications of P per sublayer - bit-identical to what MODE 2
//         computes, which the robust path needs (see group_rayleigh);
// MODE 2: energy integrals.  two_vec = false: z[] is the combined solution itself (fast path);
//         two_vec = true: y[] and z[] are stepped separately and combined at every knot with the
//         fitted xnorm / bb, exactly like the reference's stored knots (surfa.f:1092-1095).
template <int MODE, bool KERN = false>
SD_HD __forceinline__ void rayleigh_sweep(const float *__restrict__ mdl, size_t fs, int B, int b,
                                               int n, float lnT, int ndiv, bool water, float div,
                                               const Drop dr, float wvno, float wvnosq, float omegsq,
                                               double y[4], double z[4], bool do_y,
                                               double xnorm, double bbn, RInt &acc,
                                               const KOut ko = KOut{nullptr, 1, 0, 0, 0}, float cw = 0.0f, K3 *hold = nullptr)
{
#pragma clang fp contract(off)   // both sweeps must see identical coefficients
    LayerRaw nraw = layer_load(mdl, fs, (size_t)dr.hs_layer * B + b);
    for (int jl = dr.hs_layer; jl >= 0; --jl) {
        const LayerRaw raw = nraw;
        if (jl > 0) nraw = layer_load(mdl, fs, (size_t)(jl - 1) * B + b);   // in flight during this layer
        const int nsub = (jl == 0 && water) ? 1 : ndiv;
        const int nreg = (jl == dr.hs_layer) ? dr.nreg_hs : nsub;
        if (nreg <= 0) continue;
        const LayerV v = layer_derive(raw, lnT, jl == n - 1);
        if (v.b <= 0.0f) {                                           // water: surfa.f:930
            if (MODE == 2 && KERN && jl > 0) ko.put(jl, 0.0f, 0.0f, 0.0f);   // (the top layer's entry: group_rayleigh)
            continue;
        }
        const float dsub = (ndiv > 1 && !(jl == 0 && water)) ? v.d / div : v.d;
        const float xmu = v.rho * v.b * v.b;                         // surfa.f:831-832
        const float xlamb = v.rho * (v.a * v.a - 2.0f * v.b * v.b);
        KC kc{};
        if (MODE == 2 && KERN) kc = kern_coef(raw, v, lnT, jl == n - 1, cw, ko.raw != 0);
        RCoef q;
        q.ddz = -dsub / (4.0f * 1.0f);
        q.a12 = 1.0f / (xlamb + 2.0f * xmu);
        q.a13 = wvno * xlamb * q.a12;
        q.a21 = -omegsq * v.rho;
        q.a24 = wvno; q.a31 = -wvno;
        q.a34 = 1.0f / xmu;
        q.a42 = -q.a13;
        q.a43 = q.a21 + 4.0f * wvnosq * xmu * (xlamb + xmu) * q.a12;
        const RProp P = make_prop(q);
        if (MODE == 0) {
            const RProp P4 = prop_sq(prop_sq(P));
            for (int s = 0; s < nreg; ++s) {
                if (do_y) prop_apply(P4, y);
                prop_apply(P4, z);
            }
            continue;
        }
        if (MODE == 1) {
            for (int s = 0; s < nreg; ++s) {
#pragma unroll
                for (int kk = 3; kk >= 0; --kk) {
                    if (do_y) prop_apply(P, y);
                    prop_apply(P, z);
                }
            }
            continue;
        }
        const bool two_vec = do_y;
        const float dz = dsub / 4.0f;
        const float l2m = xlamb + 2.0f * xmu;
        const float ixmu = q.a34, il2m = q.a12;       // 1/mu, 1/(lambda+2mu): already formed above
        float f_mr[5], f_mz[5], f_rz[5], f_zr[5];
        constexpr bool kern = KERN;
        KR k_mr = 0, k_mz = 0, k_rz = 0, k_zr = 0, k_sz = 0, k_sr = 0;   // this layer's sums
#ifdef SD_KERN_BOOLE_INCR
        // the two strain integrals only the partials need: Boole sums formed knot by knot (weights 7 32 12 32 7), the
        // knot shared with the next sublayer kept - no five-entry arrays alive across the RK4 steps
        float w_sz = 0.0f, w_sr = 0.0f, t_sz = 0.0f, t_sr = 0.0f;
#else
        float f_sz[5], f_sr[5];
#endif
        const double ibb = 1.0 / bbn;
        auto knot = [&](int kk) {
            // fast path: z[] is the combined, normalised solution (xnorm*y + z)/bb itself;
            // robust path: combine the separately integrated solutions at the knot
            float aur, auz, atz, atr;
            if (two_vec) {
                aur = (float)((xnorm * y[0] + z[0]) * ibb); auz = (float)((xnorm * y[1] + z[1]) * ibb);
                atz = (float)((xnorm * y[2] + z[2]) * ibb); atr = (float)((xnorm * y[3] + z[3]) * ibb);
            } else {
                aur = (float)z[0]; auz = (float)z[1]; atz = (float)z[2]; atr = (float)z[3];
            }
            const float durdz = atr * ixmu - wvno * auz;
            const float duzdz = (atz + wvno * xlamb * aur) * il2m;
            f_mr[kk] = aur * aur; f_mz[kk] = auz * auz;
            f_rz[kk] = aur * duzdz; f_zr[kk] = auz * durdz;
            if (kern) {
#ifdef SD_KERN_BOOLE_INCR
                t_sz = duzdz * duzdz; t_sr = durdz * durdz;
                const float wk = (kk == 2) ? 12.0f : ((kk & 1) ? 32.0f : 7.0f);
                w_sz = fmaf(wk, t_sz, w_sz); w_sr = fmaf(wk, t_sr, w_sr);
#else
                f_sz[kk] = duzdz * duzdz; f_sr[kk] = durdz * durdz;
#endif
            }
        };
        for (int s = 0; s < nreg; ++s) {
            // bottom knot: the top knot of the sublayer below when it belongs to the same layer
            if (s == 0) knot(4);
            else { f_mr[4] = f_mr[0]; f_mz[4] = f_mz[0]; f_rz[4] = f_rz[0]; f_zr[4] = f_zr[0];
#ifdef SD_KERN_BOOLE_INCR
                   if (kern) { w_sz = 7.0f * t_sz; w_sr = 7.0f * t_sr; }
#else
                   if (kern) { f_sz[4] = f_sz[0]; f_sr[4] = f_sr[0]; }
#endif
            }
#pragma unroll
            for (int kk = 3; kk >= 0; --kk) {
                if (two_vec) prop_apply(P, y);
                prop_apply(P, z);
                knot(kk);
            }
            const float hq = dz / 22.5f;
#define SD_BOOLE(v) (hq * (7.0f * (v[0] + v[4]) + 32.0f * (v[1] + v[3]) + 12.0f * v[2]))
            const float dmmr = SD_BOOLE(f_mr), dmmz = SD_BOOLE(f_mz);
            const float drsz = SD_BOOLE(f_rz), dzsr = SD_BOOLE(f_zr);
            if (kern) {
                k_mr += dmmr; k_mz += dmmz; k_rz += drsz; k_zr += dzsr;
#ifdef SD_KERN_BOOLE_INCR
                k_sz += hq * w_sz; k_sr += hq * w_sr;
#else
                k_sz += SD_BOOLE(f_sz); k_sr += SD_BOOLE(f_sr);
#endif
            }
#undef SD_BOOLE
            // the sublayer's contributions in fp32 as in the reference (surfa.f:1126-1128); only the
            // running sums are fp64 (fp64 arithmetic runs at half rate)
            acc.i0 += (double)(v.rho * (dmmr + dmmz));
            acc.i1 += (double)(l2m * dmmr + xmu * dmmz);
            acc.i2 += (double)(xmu * dzsr - xlamb * drsz);
        }
        if (kern) {
            const K3 sh = kern_layer_rayleigh(kc, v.rho, xlamb, xmu, wvno, wvnosq, omegsq, k_mr, k_mz, k_rz, k_zr, k_sz, k_sr);
            if (jl == dr.hs_layer) *hold = sh;                       // the half-space terms join it (group_rayleigh)
            else ko.put(jl, sh.b, sh.a, sh.r);
        }
    }
}

template <bool KERN = false>
SD_HD float group_rayleigh(const float *__restrict__ mdl, size_t fs, int B, int b, int n,
                                float T, float c, float ratio, double *dbg = nullptr,
                                const KOut ko = KOut{nullptr, 1, 0, 0, 0}, float *kscale = nullptr, int *khs = nullptr)
{
#pragma clang fp contract(off)
    const float lnT = logf(1.0f / T);
    int ndiv = 5;
    const int ivre = 99 / (n - 1);                                    // surfa.f:783-784
    if (ndiv > ivre) ndiv = ivre;
    if (ndiv < 1) ndiv = 1;
    const float div = (float)ndiv;
    const LayerV top = layer_at(mdl, fs, (size_t)b, lnT, false);
    const bool water = (ndiv > 1) ? (top.b <= 0.1e-10f) : false;      // jj=2 only when splitting
    const bool wet = !(top.b > 0.0f);
    const Drop dr = drop_group<2>(mdl, fs, B, b, n, lnT, c, T, ndiv, water, div);
    const float wvno = 6.2831853072f / (c * T);
    const float wvnosq = wvno * wvno;
    const float omega = 6.2831853072f / T;
    const float omegsq = omega * omega;
    RInt acc; acc.i0 = 0.0; acc.i1 = 0.0; acc.i2 = 0.0;
    float tzz = 0.0f;
    const float cw = c / wvno;                                        // (KERN) c / k
    K3 hold{0.0f, 0.0f, 0.0f};                                        // (KERN) share of the layer that holds the half space
    if (KERN) { *kscale = 0.0f; *khs = dr.hs_layer; }                 // no factor yet: the consumer writes zeros
    float wat_a = 0.0f, wat_r = 0.0f;                                 // (KERN) the water layer's own share
    if (wet) {                                                        // surfa.f:879-910
        const float d1 = top.d;
        const float xl1 = top.rho * (top.a * top.a - 2.0f * top.b * top.b);
        const float ra = c / top.a;
        const float cr1 = ra * ra - 1.0f;
        const float mag = wvno * sqrtf(fabsf(cr1));
        if (mag <= 1.0e-35f) {
            acc.i0 = top.rho * d1;
        } else {
            float sin2ra, cosra, rab1, sinra_over;
            if (cr1 >= 0.0f) {
                sin2ra = sinf(2.0f * mag * d1) / (4.0f * mag);
                cosra = cosf(mag * d1);
                rab1 = mag * mag;
                sinra_over = sinf(mag * d1) / mag;
            } else {
                sin2ra = sinhf(2.0f * mag * d1) / (4.0f * mag);
                cosra = coshf(mag * d1);
                rab1 = -(mag * mag);
                sinra_over = sinhf(mag * d1) / mag;
            }
            const float cos2rm = 1.0f / (cosra * cosra);
            const float fac1 = (0.5f * d1 + sin2ra) * cos2rm;
            const float fac3 = wvno * (0.5f * d1 - sin2ra) * cos2rm;
            const float fac2 = wvno * fac3 / rab1;
            acc.i0 = top.rho * (fac1 + fac2);
            acc.i1 = xl1 * fac2;
            acc.i2 = xl1 * fac3;
            tzz = -top.rho * omegsq * sinra_over / cosra;
            if (KERN && !ko.raw) {
                // the water layer's own partials (the reference skips liquid layers, surfa.f:1088): in a
                // fluid the dilatation is tau_zz/lambda, so int theta^2 dz = k^2 (c/a)^4 int ur^2 dz
                const LayerRaw wraw = layer_load(mdl, fs, (size_t)b);
                const Chain ch = chain_of(wraw, lnT, false, false);
                const float dldl = -wvnosq * (ra * ra) * (ra * ra) * fac2;
                const float dldr = omegsq * (fac1 + fac2);
                wat_a = 2.0f * top.rho * top.a * cw * dldl * ch.dada;
                wat_r = cw * (dldr + top.a * top.a * dldl) * ch.rfac;
            }
        }
        if (KERN) ko.put(0, 0.0f, wat_a, wat_r);                      // (the sweep skips liquid layers)
    }
    // half-space start vectors, surfa.f:913-926, 986-989
    const LayerV hsv = layer_at(mdl, fs, (size_t)dr.hs_layer * B + b, lnT, dr.hs_layer == n - 1);
    const float cova = c / hsv.a, covb = c / hsv.b;
    const float gam = 2.0f / (covb * covb);
    const float gamm1 = gam - 1.0f;
    const float ra = wvno * sqrtf(fabsf(cova * cova - 1.0f));
    const float rb = wvno * sqrtf(fabsf(covb * covb - 1.0f));
    const float det = wvnosq - ra * rb;
    const float h = hsv.rho * omegsq;
    const float brkt = -gamm1 * wvno + gam * ra * rb / wvno;
    const double y0[4] = {1.0, 0.0, (double)(-h * brkt / det), (double)(-h * ra / det)};   // ur,uz,tz,tr
    double z0[4] = {0.0, 1.0, (double)(-h * rb / det), (double)(-h * brkt / det)};
    double y[4], z[4];
    double xnorm = 0.0, bbn = 1.0;
    // Surface fit (surfa.f:1056-1069) + one refinement (restart solution 2 from the combined start
    // vector, surfa.f:990-998).  STEP = 0: P^4 per sublayer; STEP = 1: step by step.
    auto fit = [&](auto step_tag) -> double {
        constexpr int STEP = decltype(step_tag)::value;
        for (int i = 0; i < 4; ++i) { y[i] = y0[i]; z[i] = z0[i]; }
        rayleigh_sweep<STEP>(mdl, fs, B, b, n, lnT, ndiv, water, div, dr, wvno, wvnosq, omegsq,
                             y, z, true, 0.0, 1.0, acc);
        const double yt0 = y[0], yt1 = y[1];
        double aa = z[0] - ratio * z[1];
        double bb = ratio * yt1 - yt0;
        if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
        xnorm = aa / bb;
        bb = xnorm * yt1 + z[1];
        if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
        bbn = bb;
        const float ampur = (float)((xnorm * yt0 + z[0]) / bb);
        const float xtest = fabsf(ampur / ratio - 1.0f);
        if (xtest >= 0.00001f) {
            for (int i = 0; i < 4; ++i) { z0[i] = z0[i] + xnorm * y0[i]; z[i] = z0[i]; }
            rayleigh_sweep<STEP>(mdl, fs, B, b, n, lnT, ndiv, water, div, dr, wvno, wvnosq, omegsq,
                                 y, z, false, 0.0, 1.0, acc);
            aa = z[0] - ratio * z[1];
            bb = ratio * yt1 - yt0;
            if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
            xnorm = aa / bb;
            bb = xnorm * yt1 + z[1];
            if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
            bbn = bb;
        }
        // growth of the fastest solution relative to the normalisation: rounding noise injected at
        // the bottom of a direct integration of the combined solution reaches eps * this at the top
        return fmax(fmax(fabs(yt0), fabs(yt1)), fmax(fabs(z[0]), fabs(z[1]))) / fabs(bbn);
    };
    const double z0_orig[4] = {z0[0], z0[1], z0[2], z0[3]};
    const double cancel = fit(std::integral_constant<int, 0>{});
    if (dbg) { for (int i = 0; i < 4; ++i) { dbg[i] = y[i]; dbg[4 + i] = z[i]; } dbg[12] = dr.hs_layer; dbg[13] = dr.nreg_hs; dbg[14] = ndiv; dbg[8] = xnorm; dbg[9] = bbn; }
    // half-space analytic terms use the combined vector at the top of the half space
    float aur, auz;
    const bool any_solid = (dr.hs_layer > (wet ? 1 : 0)) || (dr.nreg_hs > 0);
    (void)tzz;
    if (cancel <= 1.0e7) {
        // Fast path.  Integrate the COMBINED solution w = (xnorm*y + z)/bb itself (one 4-vector): the
        // RK4 map is linear, so this equals combining separately integrated y and z up to the
        // rounding noise the fastest-growing solution picks up on the way (<= 1e7 * 1e-16).
        for (int i = 0; i < 4; ++i) { z[i] = (xnorm * y0[i] + z0[i]) / bbn; y[i] = 0.0; }
        aur = (float)z[0]; auz = (float)z[1];
        rayleigh_sweep<2, KERN>(mdl, fs, B, b, n, lnT, ndiv, water, div, dr, wvno, wvnosq, omegsq,
                                y, z, false, xnorm, bbn, acc, ko, cw, &hold);
    } else {
        // Robust path (thick structure / short period: the solutions grow by up to ~1e27 and the
        // rounding noise excited on the way up is far larger than the answer).  The reference stays
        // accurate here because it combines the SAME stored knot values its fit was made with, so
        // the excited noise cancels exactly.  Do the same without storing: redo the fit step by
        // step, then step y and z again with bit-identical arithmetic (make_prop / prop_apply are
        // built with contraction off and explicit fma) and combine at every knot.
        for (int i = 0; i < 4; ++i) z0[i] = z0_orig[i];
        (void)fit(std::integral_constant<int, 1>{});
        for (int i = 0; i < 4; ++i) { y[i] = y0[i]; z[i] = z0[i]; }
        aur = (float)((xnorm * y0[0] + z0[0]) / bbn);
        auz = (float)((xnorm * y0[1] + z0[1]) / bbn);
        rayleigh_sweep<2, KERN>(mdl, fs, B, b, n, lnT, ndiv, water, div, dr, wvno, wvnosq, omegsq,
                                y, z, true, xnorm, bbn, acc, ko, cw, &hold);
    }
    if (wet && !any_solid) { aur = ratio; auz = 1.0f; }              // label 77777, surfa.f:1140-1144
    {   // label 7002, surfa.f:1145-1186
        const float xmu = hsv.rho * hsv.b * hsv.b;
        const float xlamb = hsv.rho * (hsv.a * hsv.a - 2.0f * hsv.b * hsv.b);
        const float ap = -hsv.rho * (wvno * aur + rb * auz) / det;
        const float bp = -hsv.rho * (-ra * aur / wvno - auz) / det;
        const float a1 = -wvno * ap / hsv.rho;
        const float a2 = -wvno * rb * bp / hsv.rho;
        const float a3 = ra * ap / hsv.rho;
        const float a4 = wvnosq * bp / hsv.rho;
        if (rb == 0.0f) return hsv.b;                                 // label 7006
        const double dmmr = a1 * a1 / (2.0f * ra) + 2.0f * a1 * a2 / (ra + rb) + a2 * a2 / (2.0f * rb);
        const double dmmz = a3 * a3 / (2.0f * ra) + 2.0f * a3 * a4 / (ra + rb) + a4 * a4 / (2.0f * rb);
        const double drsz = -a1 * a3 / 2.0f - (a1 * a4 * rb + a2 * a3 * ra) / (ra + rb) - a2 * a4 / 2.0f;
        const double dzsr = -a1 * a3 / 2.0f - (a1 * a4 * ra + a2 * a3 * rb) / (ra + rb) - a2 * a4 / 2.0f;
        acc.i0 += hsv.rho * (dmmr + dmmz);
        acc.i1 += (xlamb + 2.0f * xmu) * dmmr + xmu * dmmz;
        acc.i2 += xmu * dzsr - xlamb * drsz;
        if (KERN) {                                                   // surfa.f:1163-1164, 1178-1184
            const float smmz = ra * a3 * a3 / 2.0f + 2.0f * ra * rb * a3 * a4 / (ra + rb) + rb * a4 * a4 / 2.0f;
            const float smmr = ra * a1 * a1 / 2.0f + 2.0f * ra * rb * a1 * a2 / (ra + rb) + rb * a2 * a2 / 2.0f;
            const LayerRaw hraw = layer_load(mdl, fs, (size_t)dr.hs_layer * B + b);
            const K3 sh = kern_layer_rayleigh(kern_coef(hraw, hsv, lnT, dr.hs_layer == n - 1, cw, ko.raw != 0), hsv.rho, xlamb, xmu,
                                              wvno, wvnosq, omegsq, (KR)dmmr, (KR)dmmz, (KR)drsz, (KR)dzsr, (KR)smmz, (KR)smmr);
            ko.put(dr.hs_layer, hold.b + sh.b, hold.a + sh.a, hold.r + sh.r);
        }
    }
    if (dbg) { dbg[10] = acc.i0; dbg[11] = acc.i1; dbg[15] = acc.i2; }
    const float s0 = (float)acc.i0, s1 = (float)acc.i1, s2 = (float)acc.i2;
    if (KERN) *kscale = 1.0f / (-2.0f * (wvno * s1 + s2));           // 1 / (dL/dk), surfa.f:1203-1207: applied by the consumer
    return (wvno * s1 + s2) / (omega * s0);                           // surfa.f:1186
}

// ---- Love, surfa.f:374-606 (all fp32, as the reference) --------------------------------------
template <bool KERN = false>
SD_HD float group_love(const float *__restrict__ mdl, size_t fs, int B, int b, int n,
                            float T, float c, const KOut ko = KOut{nullptr, 1, 0, 0, 0}, float *kscale = nullptr, int *khs = nullptr)
{
    const float lnT = logf(1.0f / T);
    int ndiv = 5;
    const int ivre = 999 / (n - 1);                                   // surfa.f:414-415
    if (ndiv > ivre) ndiv = ivre;
    if (ndiv < 1) ndiv = 1;
    const float div = (float)ndiv;
    const LayerV top = layer_at(mdl, fs, (size_t)b, lnT, false);
    const bool water = (ndiv > 1) ? (top.b <= 0.1e-10f) : false;
    const Drop dr = drop_group<1>(mdl, fs, B, b, n, lnT, c, T, ndiv, water, div);
    const float wvno = 6.2831853f / (c * T);
    const LayerV hsv = layer_at(mdl, fs, (size_t)dr.hs_layer * B + b, lnT, dr.hs_layer == n - 1);
    constexpr bool kern = KERN;
    const float wvnosq = wvno * wvno;
    const float omega = 6.2831853f / T, omegsq = omega * omega;
    const float cw = c / wvno;
    // one layer's share of dc/db, dc/drho (surfa.f:509-512, 561-565), still to be divided by dL/dk
    auto kern_layer = [&](int jl, const LayerRaw &raw, const LayerV &v, float dm, float sm) -> K3 {
        const Chain ch = chain_of(raw, lnT, jl == n - 1, ko.raw != 0);
        const float dldm = -(wvnosq * dm + sm);
        const float dldr = omegsq * dm;
        return K3{2.0f * v.rho * v.b * cw * dldm * ch.dbdb, 0.0f, cw * (dldr + v.b * v.b * dldm) * ch.rfac};
    };
    if (kern) { *kscale = 0.0f; *khs = dr.hs_layer; }
    float ut0 = 1.0f;
    for (int attempt = 0; attempt < 16; ++attempt) {
        float ut = ut0;
        K3 hold{0.0f, 0.0f, 0.0f};                                    // (KERN) share of the layer that holds the half space
        const float covb = c / hsv.b;
        const float hh = hsv.rho * hsv.b * hsv.b;
        const float rbh = wvno * sqrtf(fabsf(covb * covb - 1.0f));
        float tq = -hh * rbh * ut0;
        const float dm0 = (rbh == 0.0f) ? 1.0e25f : 0.5f / rbh;
        float sumi0 = hsv.rho * dm0;
        float sumi1 = hh * dm0;
        bool overflow = false;
        LayerRaw nraw = layer_load(mdl, fs, (size_t)dr.hs_layer * B + b);
        // the half space itself (surfa.f:502-512): int u^2 = 1/(2 rb), int (du/dz)^2 = rb/2 - like the reference's sumi0 /
        // sumi1 start values not scaled with ut0^2
        if (kern && rbh > 0.0f) hold = kern_layer(dr.hs_layer, nraw, hsv, dm0, 0.5f * rbh);
        for (int jl = dr.hs_layer; jl >= 0 && !overflow; --jl) {
            const LayerRaw raw = nraw;
            if (jl > 0) nraw = layer_load(mdl, fs, (size_t)(jl - 1) * B + b);
            const int nsub = (jl == 0 && water) ? 1 : ndiv;
            const int nreg = (jl == dr.hs_layer) ? dr.nreg_hs : nsub;
            if (nreg <= 0) continue;
            const LayerV v = layer_derive(raw, lnT, jl == n - 1);
            if (v.b == 0.0f) {                                        // surfa.f:524 (still tests |ut|)
                if (fabsf(ut) > 1.0e10f) overflow = true;
                if (kern && jl != dr.hs_layer) ko.put(jl, 0.0f, 0.0f, 0.0f);
                continue;
            }
            const float dsub = (ndiv > 1 && !(jl == 0 && water)) ? v.d / div : v.d;
            const float cv = c / v.b;
            const float rb = wvno * sqrtf(fabsf(cv * cv - 1.0f));
            const float h = v.rho * v.b * v.b;
            const float dz = dsub / 4.0f;
            // the four quarter-step propagators are the same for every sublayer of this layer
            float yk[4], zk[4], ck[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float xkk = (float)(kk + 1);
                const float q = rb * dz * xkk;
                if (c < v.b) {
                    const float ep = expf(q), emq = 1.0f / ep;
                    yk[kk] = (ep - emq) / (2.0f * rb);
                    zk[kk] = rb * rb * yk[kk];
                    ck[kk] = (ep + emq) / 2.0f;
                } else if (c == v.b) {
                    yk[kk] = dz * xkk; zk[kk] = 0.0f; ck[kk] = 1.0f;
                } else {
                    float sn, cs; sincosf(q, &sn, &cs);
                    yk[kk] = sn / rb; zk[kk] = -rb * sn; ck[kk] = cs;
                }
            }
            float k_dm = 0.0f, k_sm = 0.0f;
            for (int s = 0; s < nreg; ++s) {
                if (fabsf(ut) > 1.0e10f) { overflow = true; break; } // surfa.f:519-522
                float dmm[5], smm[5];
                dmm[0] = ut * ut;
                if (kern) smm[0] = (tq / h) * (tq / h);
                float eut = ut, ett = tq;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    eut = ck[kk] * ut - yk[kk] * tq / h;
                    ett = -h * zk[kk] * ut + ck[kk] * tq;
                    dmm[kk + 1] = eut * eut;
                    if (kern) smm[kk + 1] = (ett * ett) / (h * h);
                }
                ut = eut; tq = ett;
                const float dm = (dz / 22.5f) * (7.0f * (dmm[0] + dmm[4]) + 32.0f * (dmm[1] + dmm[3]) + 12.0f * dmm[2]);
                sumi0 = sumi0 + v.rho * dm;
                sumi1 = sumi1 + h * dm;
                if (kern) {
                    k_dm += dm;
                    k_sm += (dz / 22.5f) * (7.0f * (smm[0] + smm[4]) + 32.0f * (smm[1] + smm[3]) + 12.0f * smm[2]);
                }
            }
            if (kern && !overflow) {
                const K3 sh = kern_layer(jl, raw, v, k_dm, k_sm);
                if (jl == dr.hs_layer) { hold.b += sh.b; hold.r += sh.r; }
                else ko.put(jl, sh.b, 0.0f, sh.r);
            }
        }
        if (overflow) { ut0 = ut0 / 1.0e5f; continue; }
        if (kern) {                                                   // surfa.f:581-585
            ko.put(dr.hs_layer, hold.b, 0.0f, hold.r);
            *kscale = 1.0f / (-2.0f * wvno * sumi1);                  // 1 / (dL/dk): applied by the consumer
        }
        sumi0 = sumi0 / (ut * ut);
        sumi1 = sumi1 / (ut * ut);
        return sumi1 / (c * sumi0);                                   // surfa.f:606
    }
    return 0.0f;
}

// KERN: also write the analytic partials (A.kb/ka/kr).  The plain variant is held to 168 VGPRs (three
// wavefronts per SIMD instead of two).
template <int KIND, bool KERN>
#ifndef SD_GROUP_WAVES
#define SD_GROUP_WAVES 3
#endif
#ifndef SD_KERN_WAVES
#define SD_KERN_WAVES 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KERN ? SD_KERN_WAVES : SD_GROUP_WAVES, 8)))
void surfdisp_group_kernel(GroupArgs A)
{
    const int B = A.B, P = A.P;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int b, k;                                               // a wavefront = 64 stacks, one period
    if (A.xcd_order) {
        // Workgroups go to the 8 XCDs in turn (blockIdx % 8), each with its own L2.  In plain period-major order the P
        // workgroups that read the same 256 stacks are B/256 launches apart and every one of them fetches the stacks from HBM
        // again, once per sweep (16 384 x L64 x P20: 0.5 .. 0.8 GB per launch for 17 MB of stacks).  Here XCD x takes the stack
        // blocks x, x + 8, ... g at a time, period by period: the P readers of a stack block run on one XCD, close in time.
        const int g = A.xcd_order;                          // stack blocks an XCD works on at a time, period by period
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int r = q % (g * P);
        k = r / g;
        b = (((q / (g * P)) * g + r % g) * 8 + x) * 256 + (int)threadIdx.x;
        if (b >= B) return;
        if (A.krev) k = P - 1 - k;
        idx = (size_t)k * B + b;
    } else {
        if (idx >= (size_t)B * P) return;
        b = (int)(idx % B); k = (int)(idx / B);
        if (A.krev) { k = P - 1 - k; idx = (size_t)k * B + b; }
    }
    const size_t o = idx;                                   // period-major [P][B]: coalesced
    const int n = A.nl[b];
    KOut ko{nullptr, 1, 0, 0, 0};
    float *ra_ = nullptr, *rr_ = nullptr;                   // direct route: this unit's dc/dVp, dc/drho rows
    if (KERN) {
        ko.raw = A.kraw;
        if (A.kscr) {                                       // layer-major scratch [3][Lmax][P][B]: coalesced
            const size_t arr = (size_t)A.Lmax * P * B;
            ko.stride = (size_t)P * B;
            ko.b = A.kscr + idx;
            ko.off_a = (KIND == 2 && A.ka) ? (ptrdiff_t)(arr * sizeof(float)) : 0;
            ko.off_r = A.kr ? (ptrdiff_t)(2 * arr * sizeof(float)) : 0;
        } else {                                            // the caller's [B][P][Lmax] rows (small workspace)
            const size_t ro = ((size_t)b * P + k) * A.Lmax;
            ko.stride = 1;
            ko.b = A.kb + ro;
            ra_ = A.ka ? A.ka + ro : nullptr;
            rr_ = A.kr ? A.kr + ro : nullptr;
            // (uniform byte distances between the caller's arrays; Love has no dc/dVp: its rows are zero-filled below)
            ko.off_a = (KIND == 2 && A.ka) ? (ptrdiff_t)(reinterpret_cast<const char *>(A.ka) - reinterpret_cast<const char *>(A.kb)) : 0;
            ko.off_r = A.kr ? (ptrdiff_t)(reinterpret_cast<const char *>(A.kr) - reinterpret_cast<const char *>(A.kb)) : 0;
        }
    }
    float kscale = 0.0f;
    int khs = -1;
    float ugr = 0.0f;
    if (n >= 2 && k < A.nsolved[b]) {
        const size_t fs = (size_t)A.Lmax * B;
        const float T = A.per[k];
        const float c = A.c[o];
        if (KIND == 2) ugr = group_rayleigh<KERN>(A.mdl, fs, B, b, n, T, c, A.ratio[(size_t)k * B + b],
                                                  A.dbg ? A.dbg + 16 * o : nullptr, ko, &kscale, &khs);
        else           ugr = group_love<KERN>(A.mdl, fs, B, b, n, T, c, ko, &kscale, &khs);
    }
    if (KERN) {
        if (!(fabsf(kscale) <= 3.0e38f)) kscale = 0.0f;     // (no finite factor: zeros, as for an unsolved unit)
        if (A.kscr) { A.kscale[o] = kscale; A.khs[o] = (kscale != 0.0f) ? khs : -1; }
        else {
            // direct route: the lane finishes its own rows - the products the transposition kernel forms on the scratch route
            const int top = (kscale != 0.0f) ? khs : -1;
            for (int i = 0; i < A.Lmax; ++i) {
                ko.b[i] = (i <= top) ? ko.b[i] * kscale : 0.0f;
                if (ra_) ra_[i] = (KIND == 2 && i <= top) ? ra_[i] * kscale : 0.0f;
                if (rr_) rr_[i] = (i <= top) ? rr_[i] * kscale : 0.0f;
            }
        }
    }
    A.u[o] = ugr;
}

// K3: period-major internal results -> the caller's [B][P] arrays, through an LDS tile so that both
// the reads (along b) and the writes (whole rows of consecutive stacks) are coalesced.
__global__ __launch_bounds__(256) void surfdisp_finish_kernel(FinishArgs A)
{
    extern __shared__ float tile[];                 // [64][P+1]
    const int B = A.B, P = A.P, PS = P + 1;
    const int b0 = blockIdx.x * 64;
    const int nb = min(64, B - b0);
    if (A.nsolved && threadIdx.x < nb) {
        // independent mode: status word from the reduced first-failing period
        const int bb = b0 + threadIdx.x;
        const int ns = A.nsolved[bb];
        if (A.status) A.status[bb] = (A.nl[bb] < 2) ? SURFDISP_BADMODEL
                                     : (ns >= P ? SURFDISP_OK
                                     : (ns < 0 ? SURFDISP_NUMERIC : (ns == 0 ? SURFDISP_NOROOT : SURFDISP_PARTIAL)));
    }
    for (int pass = 0; pass < 3; ++pass) {
        const float *src = (pass == 0) ? A.ct : (pass == 1 ? A.ut : A.rt);
        float *dst = (pass == 0) ? A.c : (pass == 1 ? A.u : A.ratio);
        if (!src || !dst) continue;                 // phase-only call: no group velocities (block-uniform)
        for (int i = threadIdx.x; i < 64 * P; i += 256) {
            const int k = i / 64, bl = i % 64;
            if (bl < nb) {
                float v = src[(size_t)k * B + b0 + bl];
                if (A.nsolved && k >= A.nsolved[b0 + bl]) v = 0.0f;   // independent mode: failure cascade
                // the ellipticity slot of an unsolved period was never written
                if (pass == 2 && (k >= A.nsolved_all[b0 + bl] || A.nl[b0 + bl] < 2)) v = 0.0f;
                tile[bl * PS + k] = v;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nb * P; i += 256) {
            const int bl = i / P, k = i % P;
            dst[(size_t)b0 * P + i] = tile[bl * PS + k];
        }
        __syncthreads();
    }
}

// K2b: the analytic partials from the layer-major scratch [3][Lmax][P*B] (unit index u = k*B + b) to the caller's
// [B][P][Lmax] rows, 64 units x 64 layers per workgroup through an LDS tile: reads run along the units, writes along
// the layers.  Applies the unit's factor 1 / (dL/dk) and writes zeros below the unit's deepest layer (entries the
// group-velocity kernel never wrote are not read).  blockIdx.z: 0 dc/dVs, 1 dc/dVp, 2 dc/drho.
__global__ __launch_bounds__(256) void surfdisp_kern_transpose_kernel(KernTransposeArgs A)
{
    __shared__ float tile[64][65];
    const int B = A.B, P = A.P, Lmax = A.Lmax;
    float *__restrict__ out = (blockIdx.z == 0) ? A.kb : ((blockIdx.z == 1) ? A.ka : A.kr);
    if (!out) return;                                        // (block-uniform)
    const float *__restrict__ scr = A.kscr + (size_t)blockIdx.z * Lmax * P * B;
    const bool zero_only = (blockIdx.z == 1) && (A.kind != 2);   // Love has no dc/dVp
    const int nbb = (B + 63) / 64;
    const int k = blockIdx.x / nbb, b0 = (blockIdx.x % nbb) * 64, i0 = blockIdx.y * 64;
    const size_t PB = (size_t)P * B;
    const int bl = threadIdx.x % 64;
    float sc = 0.0f; int hs = -1;
    if (b0 + bl < B && !zero_only) { sc = A.kscale[(size_t)k * B + b0 + bl]; hs = A.khs[(size_t)k * B + b0 + bl]; }
    for (int il = threadIdx.x / 64; il < 64; il += 4)
        tile[il][bl] = (i0 + il <= hs) ? scr[(size_t)(i0 + il) * PB + (size_t)k * B + b0 + bl] * sc : 0.0f;
    __syncthreads();
    for (int t = threadIdx.x; t < 64 * 64; t += 256) {
        const int bq = t / 64, il = t % 64;
        if (i0 + il < Lmax && b0 + bq < B) out[((size_t)(b0 + bq) * P + k) * Lmax + i0 + il] = tile[il][bq];
    }
}

}  // namespace sd

// ======================================================================================= launch
namespace {

constexpr int SD_MAX_DEVICES = 64;

template <int KIND, int G, bool INDEP, bool FAST = false, bool EXACT = false>
hipError_t launch_phase_g(hipStream_t s, const sd::PhaseArgs &a)
{
    constexpr int S = SD_PHASE_BLOCK / G;
    const size_t lds = EXACT ? sd::phase_exact_lds_bytes(a.Lmax, G, KIND) : sd::phase_lds_bytes(a.Lmax, G, a.overlap != 0, KIND);
    auto kern = sd::surfdisp_phase_kernel<KIND, G, INDEP, FAST, EXACT>;
    // raise the dynamic-LDS limit of this instantiation only when a launch needs more than any before it (per
    // device): the attribute call costs ~10 us, visible in launch-bound Metropolis loops
    static std::atomic<size_t> lds_set[SD_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = (dev >= 0 && dev < SD_MAX_DEVICES) ? dev : 0;
    if (lds > lds_set[di].load(std::memory_order_acquire) || dev != di) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        size_t cur = lds_set[di].load(std::memory_order_relaxed);
        while (lds > cur && !lds_set[di].compare_exchange_weak(cur, lds, std::memory_order_release)) {}
    }
    const long teams = INDEP ? (long)a.B * a.P : (long)a.B;
    const int grid = (int)((teams + S - 1) / S);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SD_PHASE_BLOCK), lds, s, a);
    return hipGetLastError();
}

// the exact fallback: teams of 16 lanes, or as many as it takes for the working stacks to fit 64 KB of LDS
template <int KIND, bool INDEP>
hipError_t launch_phase_x(hipStream_t s, const sd::PhaseArgs &a)
{
    switch (sd::phase_exact_team(a.Lmax, KIND)) {
        case 16: return launch_phase_g<KIND, 16, INDEP, false, true>(s, a);
        case 32: return launch_phase_g<KIND, 32, INDEP, false, true>(s, a);
        default: return launch_phase_g<KIND, 64, INDEP, false, true>(s, a);
    }
}

template <int KIND, bool INDEP>
hipError_t launch_phase_k(hipStream_t s, const sd::PhaseArgs &a, int G)
{
    if (a.fast) {                                  // opt-in fast scan: teams of 2..8 lanes only
        switch (G) {
            case 2:  return launch_phase_g<KIND, 2, INDEP, true>(s, a);
            case 4:  return launch_phase_g<KIND, 4, INDEP, true>(s, a);
            case 8:  return launch_phase_g<KIND, 8, INDEP, true>(s, a);
            default: break;
        }
    }
    switch (G) {
        case 1:  return launch_phase_g<KIND, 1, INDEP>(s, a);
        case 2:  return launch_phase_g<KIND, 2, INDEP>(s, a);
        case 4:  return launch_phase_g<KIND, 4, INDEP>(s, a);
        case 8:  return launch_phase_g<KIND, 8, INDEP>(s, a);
        case 16: return launch_phase_g<KIND, 16, INDEP>(s, a);
        case 32: return launch_phase_g<KIND, 32, INDEP>(s, a);
        case 64: return launch_phase_g<KIND, 64, INDEP>(s, a);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

namespace sd {

// working stack per team (+ the ellipticity snapshot slot for teams of >= 4 lanes; allocated for
// Love too so that one number describes a launch)
// NEVILL's table x(12), y(12) per team behind the working stack(s)
size_t phase_lds_bytes(int Lmax, int G, bool overlap, int kind) { const int S = SD_PHASE_BLOCK / G; return ((size_t)((G >= 4 && overlap) ? 2 : 1) * lds_ls(S, kind == 1 ? NFW_LOVE : NFW) * Lmax + (size_t)24 * S) * sizeof(float); }
// exact fallback: one working stack per team + NEVILL's table x(12), y(12)
size_t phase_exact_lds_bytes(int Lmax, int G, int kind) { const int S = SD_PHASE_BLOCK / G; return ((size_t)lds_ls(S, kind == 1 ? NFW_LOVE : NFW) * Lmax + (size_t)24 * S) * sizeof(float); }
int phase_exact_team(int Lmax, int kind)
{
    int G = 16;
    while (G < 64 && phase_exact_lds_bytes(Lmax, G, kind) > 64u * 1024u) G *= 2;
    return G;
}

hipError_t launch_phase_exact(hipStream_t s, int kind, bool independent, const PhaseArgs &a)
{
    if (independent) return kind == 2 ? launch_phase_x<2, true>(s, a) : launch_phase_x<1, true>(s, a);
    return kind == 2 ? launch_phase_x<2, false>(s, a) : launch_phase_x<1, false>(s, a);
}

template <int KIND>
static void launch_prep_k(hipStream_t s, int TL, const PrepArgs &a)
{
    const long threads = (long)a.B * TL;
    const int grid = (int)((threads + 255) / 256);
    switch (TL) {
        case 1:  hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 1>), dim3(grid), dim3(256), 0, s, a); break;
        case 2:  hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 2>), dim3(grid), dim3(256), 0, s, a); break;
        case 4:  hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 4>), dim3(grid), dim3(256), 0, s, a); break;
        case 8:  hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 8>), dim3(grid), dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 16>), dim3(grid), dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 32>), dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((surfdisp_prep_kernel<KIND, 64>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

hipError_t launch_prep(hipStream_t s, int kind, const PrepArgs &a)
{
    // lanes per stack: enough wavefronts to give every SIMD about four (262 144 lanes), never more lanes than layers
    int TL = 1;
    while (TL < 64 && (long)a.B * TL < 262144L && TL < a.Lmax) TL *= 2;
    if (kind == 2) launch_prep_k<2>(s, TL, a); else launch_prep_k<1>(s, TL, a);
    return hipGetLastError();
}

hipError_t launch_phase(hipStream_t s, int kind, int G, bool independent, const PhaseArgs &a)
{
    if (independent) return kind == 2 ? launch_phase_k<2, true>(s, a, G) : launch_phase_k<1, true>(s, a, G);
    return kind == 2 ? launch_phase_k<2, false>(s, a, G) : launch_phase_k<1, false>(s, a, G);
}

hipError_t launch_kern_transpose(hipStream_t s, const KernTransposeArgs &a)
{
    const dim3 grid((unsigned)(a.P * ((a.B + 63) / 64)), (unsigned)((a.Lmax + 63) / 64), 3u);
    hipLaunchKernelGGL(surfdisp_kern_transpose_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ellip(hipStream_t s, const EllipArgs &a)
{
    const size_t total = (size_t)a.B * a.P;
    hipLaunchKernelGGL(surfdisp_ellip_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_finish(hipStream_t s, const FinishArgs &a)
{
    const int grid = (a.B + 63) / 64;
    const size_t lds = (size_t)64 * (a.P + 1) * sizeof(float);
    hipLaunchKernelGGL(surfdisp_finish_kernel, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_group(hipStream_t s, int kind, const GroupArgs &a_in)
{
    GroupArgs a = a_in;
    const size_t total = (size_t)a.B * a.P;
    int grid = (int)((total + 255) / 256);
    const bool kern = a.kb != nullptr;
    // Workgroup order (see the kernel).  group_order < 0 (default): the XCD-aware order, four stack blocks per XCD at a time, for the
    // launches that also write the partials when every XCD gets the same number (<= 8) of stack blocks; plain period-major order
    // otherwise.  Measured (16 384 x L64 / 25 600 x L96, P20, Rayleigh; profiles/r04b/group_order.txt): with the partials
    // 535 us + 808 MB fetched in plain order, 533 us + 216 MB with four blocks at a time, 565 us + 75 MB with one; without them
    // 471 / 502 / 499 us (the plain launch is not worth it); 25 600 stacks (12.5 blocks per XCD: uneven shares) 1.25 ms plain,
    // 1.39 .. 1.55 ms in every grouped order.
    const int nblk = (a.B + 255) / 256;
    int g = a.group_order;
    if (g < 0) g = (kern && a.B % 2048 == 0 && nblk / 8 <= 8) ? (nblk / 8 < 4 ? nblk / 8 : 4) : 0;
    a.krev = g >= 100 ? 1 : 0;
    g %= 100;
    a.xcd_order = (nblk >= 8 && a.B % 256 == 0 && g > 0) ? g : 0;
    if (a.xcd_order) grid = (((nblk + 7) / 8 + a.xcd_order - 1) / a.xcd_order) * a.xcd_order * 8 * a.P;
    if (kind == 2 && kern)  hipLaunchKernelGGL((surfdisp_group_kernel<2, true>), dim3(grid), dim3(256), 0, s, a);
    else if (kind == 2)     hipLaunchKernelGGL((surfdisp_group_kernel<2, false>), dim3(grid), dim3(256), 0, s, a);
    else if (kern)          hipLaunchKernelGGL((surfdisp_group_kernel<1, true>), dim3(grid), dim3(256), 0, s, a);
    else                    hipLaunchKernelGGL((surfdisp_group_kernel<1, false>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace sd

