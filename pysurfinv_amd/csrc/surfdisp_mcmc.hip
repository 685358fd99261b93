// surfdisp_mcmc.hip -- the Metropolis glue of one lock step on the device: what the reference does per
// step and chain in Python - BrownianVar.move for every random-walk scalar (brownian.py:20-27), Point.misfit
// (point.py:15-31) and the accept rule (point.py:34-37) - as two kernels around the forward solve, so that a lock step is
// propose -> parameters->stacks -> prep / root search / finish -> accept: six launches instead of ~45 torch kernels and
// two host synchronisations (25 600 chains x 96 layers: 0.38 ms of a 7.1 ms lock step).
// Random numbers: Philox4x32-10 (Salmon et al. 2011), keyed by the caller's seed, counter = (call counter, element,
// draw index): reproducible for a given (seed, counter), independent of the launch geometry.  Statistical parity with
// the reference's Mersenne Twister stream, like the torch generator it replaces (pysurfinv_amd.brownian).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "surfdisp_internal.h"

namespace sd {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, uint32_t k0, uint32_t k1)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = U4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
        k0 += W0; k1 += W1;
    }
    return ctr;
}
// uniform in (0, 1) from 53 random bits (never 0 nor 1)
__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

// Bounded Gaussian step around x (brownian.py:20-27 BrownianVar.move): redrawn while it falls outside (lo, hi), at most 1000
// tries (two normals per Philox call), then a uniform draw ("No valid perturb, uniform reset instead!"); reset: the uniform
// branch at once (MCinv.reset).  node: position in the speculative tree (0 for the plain lock step).
__device__ __forceinline__ double draw_bounded(double x, double lo, double hi, double s, uint32_t k0, uint32_t k1,
                                               unsigned long long counter, uint32_t node, long gidx, bool reset)
{
    const uint32_t chi32 = (uint32_t)(counter >> 32) ^ (node << 20);
    double nv = 0.0;
    bool ok = false;
    for (uint32_t t = 0; t < 500 && !ok && !reset; ++t) {
        const U4 r = philox4x32_10(U4{(uint32_t)counter, chi32 ^ (t << 8), (uint32_t)gidx, (uint32_t)(gidx >> 32)}, k0, k1);
        const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        nv = x + s * rad * cs;
        ok = (nv < hi) && (nv > lo);
        if (!ok) { nv = x + s * rad * sn; ok = (nv < hi) && (nv > lo); }
    }
    if (!ok) {
        const U4 r = philox4x32_10(U4{(uint32_t)counter, chi32 ^ 0x00ffff00u, (uint32_t)gidx, (uint32_t)(gidx >> 32)}, k0, k1);
        nv = lo + (hi - lo) * u53(r.x, r.y);
    }
    return nv;
}

// One thread per (chain, parameter).  depth <= 1: one proposal, out [C][N].  depth d > 1 (speculative sampler): the binary
// tree of the next d accept / reject outcomes - node k's proposal is drawn from the state its branch would be in (node 0:
// the chain's state; child 2k+1 "accepted": the proposal of k; child 2k+2 "rejected": the state of k), out [C][2^d - 1][N].
// Every scalar moves independently (brownian.py), so a thread builds the whole tree of its own parameter.
__global__ __launch_bounds__(256) void surfdisp_mcmc_propose_kernel(McmcProposeArgs A)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)A.C * A.N) return;
    const int n = (int)(idx % A.N);
    const long c = idx / A.N;
    const double lo = A.vmin[n], hi = A.vmax[n], s = A.step[n];
    const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
    const long gidx = idx + A.chain0 * A.N;                             // (chain, parameter) of the whole sampler: the random stream's index
    if (A.redo) {
        // masked redraw (a sampler with prior rules, MetropolisBatch): only the chains whose last proposal broke a rule draw
        // again - try number A.attempt of this step (its own random numbers); reset 2: the chain's state itself (no proposal)
        if (A.redo[c] != (unsigned char)A.redo_tag) return;
        A.out[idx] = (A.reset == 2) ? A.p[idx]
                                    : draw_bounded(A.p[idx], lo, hi, s, k0, k1, A.counter, 64u + (uint32_t)A.attempt, gidx, A.reset != 0);
        return;
    }
    const int M = A.depth > 1 ? (1 << A.depth) - 1 : 1;
    double S[SD_MCMC_MAX_NODES];
    S[0] = A.p[idx];
    // fully unrolled: every index of S is a compile-time constant.  (With a run-time index the register array is written
    // through s_set_gpr_idx BEFORE the bounds test selects the result - hipcc 7.2 speculates the guarded stores - and the
    // out-of-range writes of the nodes 7..14 of a depth-4 tree land in the neighbouring live registers, the output pointer
    // among them: a memory fault, found on the first depth-4 run.)
#pragma unroll
    for (int k = 0; k < SD_MCMC_MAX_NODES; ++k) {
        if (k < M) {
            const double x = S[k];
            const double nv = draw_bounded(x, lo, hi, s, k0, k1, A.counter, (uint32_t)k, gidx, A.reset != 0);
            A.out[((size_t)c * M + k) * A.N + n] = nv;
            if (2 * k + 2 < SD_MCMC_MAX_NODES) { S[2 * k + 1] = nv; S[2 * k + 2] = x; }
        }
    }
}

// misfit of one predicted curve against a chain's observations: (misfit, chi-square with the reference's clamp, L)
__device__ __forceinline__ void misfit_of(const McmcAcceptArgs &A, const float *cp, bool failed, size_t ob,
                                          double &mis, double &chi, double &L)
{
    chi = 0.0;
    int cnt = 0;
    for (int k = 0; k < A.P; ++k) {
        const double v = (double)cp[k];
        if (v < 0.01) failed = true;                                   // models.py:29-33
        if (A.mask[ob + k]) {
            const double r = (A.c_obs[ob + k] - v) / A.uncer[ob + k];
            chi += r * r;
            ++cnt;
        }
    }
    mis = sqrt(chi / (double)cnt);                                     // point.py:27-31
    if (!(chi < 50.0)) chi = sqrt(chi * 50.0);
    L = exp(-0.5 * chi);
    if (failed) { mis = 88888.0; chi = 88888.0; L = 0.0; }             // point.py:20-21
}

// One thread per chain: misfit of the proposal against the chain's observations, accept rule, state update, mcTrack row.
// depth d > 1: the chain walks the tree of surfdisp_mcmc_propose_kernel for nsteps <= d steps - at node k the usual test of
// proposal k against the current state, then child 2k+1 (accepted) or 2k+2 - one mcTrack row per step, step_stride doubles
// apart.  Every proposal was drawn from the state the chain is in when it is tested, so the chain is the plain sampler's.
__global__ __launch_bounds__(256) void surfdisp_mcmc_accept_kernel(McmcAcceptArgs A)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.C) return;
    const int P = A.P, N = A.N;
    const int M = A.depth > 1 ? (1 << A.depth) - 1 : 1;
    const int nsteps = A.depth > 1 ? A.nsteps : 1;
    const size_t ob = A.obs_per_chain ? (size_t)c * P : 0;
    const unsigned long long gc = (unsigned long long)(A.chain0 + c);
    double *p0 = A.p0 + (size_t)c * N;
    double chi0 = A.chi0[c];
    int node = 0;
    for (int s = 0; s < nsteps; ++s) {
        const size_t q = (size_t)c * M + node;                         // this step's proposal: stack q of the batched solve
        double mis, chi, L;
        misfit_of(A, A.c + q * P, A.status && A.status[q] != 0, ob, mis, chi, L);
        bool acc;
        if (A.first) acc = true;                                       // a chain's first row: the start model itself
        else if (chi < chi0) acc = true;                               // point.py:34-37
        else {
            const U4 r = philox4x32_10(U4{(uint32_t)A.counter, (uint32_t)(A.counter >> 32) ^ 0x00aaaa00u ^ ((uint32_t)s << 28),
                                          (uint32_t)gc, (uint32_t)(gc >> 32)}, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
            const double u = u53(r.x, r.y);
            acc = u > 1.0 - exp(-(chi - chi0) / 2.0);
        }
        const double *p1 = A.p1 + q * N;
        if (A.row) {
            double *row = A.row + (size_t)c * A.row_stride + (size_t)s * A.step_stride;
            row[0] = mis; row[1] = L; row[2] = acc ? 1.0 : 0.0;
            for (int n = 0; n < N; ++n) row[3 + n] = p1[n];
        }
        if (acc) {
            for (int n = 0; n < N; ++n) p0[n] = p1[n];
            chi0 = chi;
        }
        node = acc ? 2 * node + 1 : 2 * node + 2;
    }
    A.chi0[c] = chi0;
}

hipError_t launch_mcmc_propose(hipStream_t s, const McmcProposeArgs &a)
{
    const long total = (long)a.C * a.N;
    hipLaunchKernelGGL(surfdisp_mcmc_propose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_mcmc_accept(hipStream_t s, const McmcAcceptArgs &a)
{
    hipLaunchKernelGGL(surfdisp_mcmc_accept_kernel, dim3((unsigned)((a.C + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace sd
