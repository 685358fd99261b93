// surfdisp_internal.h -- kernel argument blocks and launch prototypes shared by
// surfdisp_kernels.hip and surfdisp_capi.hip.  Not part of the public ABI (include/surfdisp.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "surfdisp.h"

namespace sd {

struct PrepArgs {
    int B, Lmax;
    const int *nlay;      // [B] or nullptr
    const float *model;   // [B][5][Lmax] (vp, vs, rho, h, qsinv) as handed over by the caller
    float *mdl;           // [9][Lmax][B] SoA: vp, vs, rho, 1/Qs + five flattening factors
    int *nl;              // [B] validated layer count, 0 = bad model
    int P;
    int *nsolved_init;    // nullptr, or [B]: set to P (independent mode reduces it with atomicMin)
    float *fsafe;         // [B]: thickest layer (km) of a stack whose Vs and Vp never decrease with depth,
                          // 1e30 otherwise (the scan then skips nothing on that stack)
    float *ovf;           // [3][B]: thickest flattened layer, 2 ln(max rho), 4 ln(2 max Vs^2): entry_overflow_risk
    int *fb_count;        // [1]: zeroed here; stacks the production root search hands to the exact fallback
    float *rows;          // nullptr, or [B][9][Lmax]: the same nine fields one ROW per stack and field - what the root
                          // search's per-period rebuild reads when a stack has >= 8 lanes (consecutive lanes = consecutive
                          // layers: coalesced; the SoA copy costs a cache line per value there)
    int write_soa;        // 0: only the rows are needed (phase-only call with wide teams: no group-velocity kernel)
};

struct PhaseArgs {
    int B, Lmax, P;
    const float *mdl;
    const int *nl;
    const float *per;     // [P]
    float *c;             // [P][B] period-major (internal; transposed by the finish kernel)
    float *ratio;         // [P][B] ellipticity (Rayleigh), input of the group-velocity kernel
    int *nsolved;         // [B]
    int *status;          // [B] or nullptr
    float wtol;           // bracket width below which the root may be read off by interpolation
    float atol;           // ... provided secant and 3-point estimates agree to this (km/s)
    int fast;             // opt-in heuristic coarse-to-fine scan (SURFDISP_FASTSCAN); 0 = every grid point, the default
    const float *fsafe;   // [B], see PrepArgs
    int overlap;          // second LDS slot: the ellipticity passes ride in the next period's first scan pass
    float phimax;         // fast scan: largest vertical-phase increment (rad) of an interval that may be skipped
    const float *ovf;     // [3][B], see PrepArgs
    int *fb_count;        // [1] number of entries of fb_list
    int *fb_list;         // [teams]: team indices (stack, or period * B + stack in independent mode) for the exact kernel
    int balance;          // wavefront priority by progress (one batch in flight), see the kernel's main loop
    int strict;           // SURFDISP_STRICT: every team hands its stack to the exact kernel
    // where the root search reads the staged fields: value (field f, layer i) of stack b at
    // msrc[b * ms_b + f * ms_f + i * ms_i] - the SoA copy (1, Lmax * B, B) or the rows (9 * Lmax, Lmax, 1)
    const float *msrc;
    long ms_b, ms_f, ms_i;
    // nullptr, or [P][B]: per solved period the state the ellipticity kernel replays the working stack from - (number of
    // layers the period's rebuild refreshed) | (frozen effective half space << 16); -1: this period's ellipticity was
    // computed in the kernel itself (exact fallback)
    int *hist;
    int lockstep;         // the teams of a wavefront refine and end their periods together (see the root search's main loop)
    float ambig;          // a scan trial with |value| below this fraction of its terms' magnitude is evaluated again with the
                          // reference's own arithmetic (0: never)
    float phimulti;       // a bracket across which the vertical phase grows by more than this (rad) goes to NEVILL
    int *amb_count;       // nullptr, or [2]: number of scan trials / of ellipticities evaluated again (statistics)
    float ell_ambig;      // in-kernel ellipticity passes: a closure below this fraction of its terms marks the pair for the ellipticity kernel
    float ell_gmax;       // ... and so does g = 2 b^2 / c^2 of the stack's fastest layer beyond this
#ifdef SD_WAVECLOCK
    unsigned long long *wclk;   // developer build: [waves][2] s_memrealtime at wavefront start / end
#endif
};

struct GroupArgs {
    int B, Lmax, P;
    const float *mdl;
    const int *nl;
    const float *per;
    const float *c;       // [P][B]
    const float *ratio;
    const int *nsolved;
    float *u;             // [P][B]
    double *dbg;          // nullptr, or [B][P][16] intermediate values (developer builds)
    float *kb, *ka, *kr;  // nullptr, or [B][P][Lmax] analytic partials dc/dVs, dc/dVp, dc/drho
    float *kscr;          // nullptr, or the layer-major scratch [3][Lmax][P][B] the unscaled shares are stored in (coalesced)
    float *kscale;        // with kscr: [P][B] the unit's factor 1 / (dL/dk), 0 = no partials (unsolved / invalid unit)
    int *khs;             // with kscr: [P][B] deepest layer the unit wrote (-1: none)
    int kraw;             // SURFDISP_KERN_REFCOORD: partials in the reference's coordinates (unit chain factors)
    int xcd_order;        // set by launch_group: workgroup -> (stack block, period) order that keeps a stack block's periods on one XCD
    int krev;             // (developer knob, SURFDISP_GROUP_ORDER + 100: periods in descending order)
    int group_order;      // SURFDISP_GROUP_ORDER: < 0 the library's rule (launch_group), 0 plain period-major order, g > 0: an XCD takes g stack blocks at a time
};
struct KernTransposeArgs {
    int B, P, Lmax, kind;
    const float *kscr;    // [3][Lmax][P][B]
    const float *kscale;  // [P][B]
    const int *khs;       // [P][B]
    float *kb, *ka, *kr;  // the caller's [B][P][Lmax] rows (ka, kr may be nullptr)
};

struct EllipArgs {
    int B, Lmax, P;
    const float *mdl;     // SoA staged fields
    const int *nl;
    const float *per;
    const float *c;       // [P][B] roots
    const int *hist;      // [P][B], see PhaseArgs
    const int *nsolved;   // [B]
    float *ratio;         // [P][B]
    float ell_ambig;      // a closure below this fraction of its terms: both passes again with the reference's arithmetic (0: never; < 0: always)
    int only_flagged;     // 1: the root search wrote the ellipticities itself; redo only the pairs it marked (bit 30 of hist)
    int *amb_count;       // nullptr, or [2] statistics, see PhaseArgs
    float ell_gmax;       // ... and where g = 2 b^2 / c^2 of the stack's fastest layer exceeds this
    const float *ovf;     // [3][B] prep statistics (entry 2: 4 ln(2 bmax^2))
};
hipError_t launch_ellip(hipStream_t s, const EllipArgs &a);

struct FinishArgs {
    int B, P;
    const float *ct, *ut; // [P][B]
    float *c, *u;         // [B][P] caller's arrays
    const int *nsolved;   // nullptr (faithful) or [B] first failing period (independent mode)
    const int *nl;
    int *status;
    const float *rt;      // nullptr, or [P][B] ellipticity (period-major, Rayleigh) ...
    float *ratio;         // ... and the caller's [B][P] array it goes to (ABI 3)
    const int *nsolved_all; // [B] periods solved (either mode): unsolved periods of `ratio` are written as 0
};

struct LayersArgs {
    int C, N;
    const double *params;  // [C][N]
    const int *idesc;      // see surfdisp_layers.hip
    const double *fdesc;
    float *model;          // [C][5][L]
    double *scratch;       // [C][64][2] (vs, qs) of the thermal layer's grid points, or nullptr
};
constexpr int SD_MCMC_MAX_DEPTH = 4, SD_MCMC_MAX_NODES = (1 << SD_MCMC_MAX_DEPTH) - 1;
struct McmcProposeArgs {
    int C, N;
    const double *p;        // [C][N] current parameters
    const double *vmin, *vmax, *step;   // [N]
    unsigned long long seed, counter;
    int reset;              // 1: uniform prior draw for every entry (MCinv.reset), 0: bounded Gaussian step
    double *out;            // [C][N]
    long chain0;            // global index of chain 0 of this launch (keys the random streams: a sampler split into chain groups draws what the unsplit one draws)
    int depth;              // <= 1: one proposal per chain; d > 1: the speculative tree of 2^d - 1 proposals (out [C][2^d-1][N])
    const unsigned char *redo;   // nullptr, or [C]: only chains with redo[c] == redo_tag draw (masked redraw, depth 1)
    int attempt;            // ... try number of this step: its own random numbers
    int redo_tag;
};
struct McmcAcceptArgs {
    int C, N, P;
    const float *c;         // [C][P] predicted phase velocities of the proposals
    const int *status;      // [C] or nullptr
    const double *c_obs, *uncer;   // [P], or [C][P] when obs_per_chain
    const unsigned char *mask;     // same shape: 1 = the period counts
    int obs_per_chain;
    const double *p1;       // [C][N] proposals
    double *p0;             // [C][N] chain states, updated in place
    double *chi0;           // [C] chi-square of the states, updated in place
    double *row;            // nullptr, or row c at row + c * row_stride: [misfit, L, accepted, params of the proposal]
    long row_stride;        // in doubles
    unsigned long long seed, counter;
    int first;              // 1: first row of a chain (always accepted: the start model)
    long chain0;            // global index of chain 0 of this launch
    int depth, nsteps;      // speculative tree: walk nsteps <= depth steps (depth <= 1: the plain single test)
    long step_stride;       // doubles between the mcTrack rows of consecutive steps of one chain
};
hipError_t launch_mcmc_propose(hipStream_t s, const McmcProposeArgs &a);
hipError_t launch_mcmc_accept(hipStream_t s, const McmcAcceptArgs &a);
hipError_t launch_layers(hipStream_t s, const LayersArgs &a, int L);
hipError_t launch_prior(hipStream_t s, const LayersArgs &a, int L, const int *flags, double vs_max, int only_tag, int mark_tag, unsigned char *tags);
hipError_t launch_thermal(hipStream_t s, const LayersArgs &a);

// lanes per workgroup of the root search.  Nothing in it synchronises across wavefronts, so any multiple of 64
// works; measured with two batches in flight (M solves/s): 64 lanes 28.0, 128 30.8, 256 32.3, 512 32.7 - smaller
// workgroups do NOT help the second batch in, they slow the pair down
#ifndef SD_PHASE_BLOCK
#define SD_PHASE_BLOCK 256
#endif
size_t phase_lds_bytes(int Lmax, int G, bool overlap, int kind);   // per workgroup of SD_PHASE_BLOCK lanes (kind: 1 Love, 2 Rayleigh)
size_t phase_exact_lds_bytes(int Lmax, int G, int kind);
int phase_exact_team(int Lmax, int kind);                // lanes per stack of the exact fallback kernel
hipError_t launch_phase_exact(hipStream_t s, int kind, bool independent, const PhaseArgs &a);
hipError_t launch_finish(hipStream_t s, const FinishArgs &a);
hipError_t launch_kern_transpose(hipStream_t s, const KernTransposeArgs &a);
hipError_t launch_prep(hipStream_t s, int kind, const PrepArgs &a);
hipError_t launch_phase(hipStream_t s, int kind, int G, bool independent, const PhaseArgs &a);
hipError_t launch_group(hipStream_t s, int kind, const GroupArgs &a);

}  // namespace sd
