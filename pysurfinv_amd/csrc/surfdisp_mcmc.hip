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

// One thread per (chain, parameter): Gaussian step around the current value, redrawn while it falls outside
// (vmin, vmax), at most 1000 tries, then a uniform draw (brownian.py:20-27 BrownianVar.move; reset = the uniform branch).
__global__ __launch_bounds__(256) void surfdisp_mcmc_propose_kernel(McmcProposeArgs A)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)A.C * A.N) return;
    const int n = (int)(idx % A.N);
    const double x = A.p[idx], lo = A.vmin[n], hi = A.vmax[n], s = A.step[n];
    const uint32_t k0 = (uint32_t)A.seed, k1 = (uint32_t)(A.seed >> 32);
    const long gidx = idx + A.chain0 * A.N;                             // (chain, parameter) of the whole sampler: the random stream's index
    double nv = 0.0;
    bool ok = false;
    for (uint32_t t = 0; t < 500 && !ok; ++t) {                        // two normals per Philox call: 1000 tries
        const U4 r = philox4x32_10(U4{(uint32_t)A.counter, (uint32_t)(A.counter >> 32) ^ (t << 8), (uint32_t)gidx, (uint32_t)(gidx >> 32)}, k0, k1);
        const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        nv = x + s * rad * cs;
        ok = (nv < hi) && (nv > lo);
        if (!ok) { nv = x + s * rad * sn; ok = (nv < hi) && (nv > lo); }
    }
    if (!ok || A.reset) {                                              // "No valid perturb, uniform reset instead!" / MCinv.reset
        const U4 r = philox4x32_10(U4{(uint32_t)A.counter, (uint32_t)(A.counter >> 32) ^ 0x00ffff00u, (uint32_t)gidx, (uint32_t)(gidx >> 32)}, k0, k1);
        nv = lo + (hi - lo) * u53(r.x, r.y);
    }
    A.out[idx] = nv;
}

// One thread per chain: misfit of the proposal against the chain's observations, accept rule, state update, mcTrack row.
__global__ __launch_bounds__(256) void surfdisp_mcmc_accept_kernel(McmcAcceptArgs A)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.C) return;
    const int P = A.P, N = A.N;
    const float *cp = A.c + (size_t)c * P;
    const size_t ob = A.obs_per_chain ? (size_t)c * P : 0;
    bool failed = A.status && A.status[c] != 0;
    double chi = 0.0;
    int cnt = 0;
    for (int k = 0; k < P; ++k) {
        const double v = (double)cp[k];
        if (v < 0.01) failed = true;                                   // models.py:29-33
        if (A.mask[ob + k]) {
            const double r = (A.c_obs[ob + k] - v) / A.uncer[ob + k];
            chi += r * r;
            ++cnt;
        }
    }
    double mis = sqrt(chi / (double)cnt);                              // point.py:27-31
    if (!(chi < 50.0)) chi = sqrt(chi * 50.0);
    double L = exp(-0.5 * chi);
    if (failed) { mis = 88888.0; chi = 88888.0; L = 0.0; }             // point.py:20-21
    const double chi0 = A.chi0[c];
    bool acc;
    if (A.first) acc = true;                                           // a chain's first row: the start model itself
    else if (chi < chi0) acc = true;                                   // point.py:34-37
    else {
        const U4 r = philox4x32_10(U4{(uint32_t)A.counter, (uint32_t)(A.counter >> 32) ^ 0x00aaaa00u, (uint32_t)(A.chain0 + c), (uint32_t)((unsigned long long)(A.chain0 + c) >> 32)}, (uint32_t)A.seed, (uint32_t)(A.seed >> 32));
        const double u = u53(r.x, r.y);
        acc = u > 1.0 - exp(-(chi - chi0) / 2.0);
    }
    const double *p1 = A.p1 + (size_t)c * N;
    double *p0 = A.p0 + (size_t)c * N;
    if (A.row) {
        double *row = A.row + (size_t)c * A.row_stride;
        row[0] = mis; row[1] = L; row[2] = acc ? 1.0 : 0.0;
        for (int n = 0; n < N; ++n) row[3 + n] = p1[n];
    }
    if (acc) {
        for (int n = 0; n < N; ++n) p0[n] = p1[n];
        A.chi0[c] = chi;
    }
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
