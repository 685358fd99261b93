/*
 * surfdisp.h -- C ABI of libsurfdisp_hip.so: MI355X (gfx950) batched surface-wave
 * dispersion forward solver, drop-in for pySurfInv's fast_surf() path.
 *
 * Boundary replaced (reference 001cat/pySurfInv):
 *   - Fortran entry   SUBROUTINE FAST_SURF(n_layer0,kind0,a_ref0,b_ref0,rho_ref0,d_ref0,
 *                     qs_ref0,cvper,ncvper,uR0,uL0,cR0,cL0)      fast_surf_src/fast_surf.f:2-5
 *   - f2py signature  fast_surf(nlay,ilvry,Vp,Vs,rho,h,qsinv,per,nper)->(ur0,ul0,cr0,cl0)
 *                                                                 fast_surf_src/fast_surf.pyf:6-19
 *   - call sites      models.py:27 (_calForward), senskernel.py:188 (SensKernelPert._forward)
 *
 * Plain pointers and sizes only; no torch / C++ types.  The solve entry points keep no state between
 * calls (the reference does: COMMON blocks, fast_surf.f:48-71) and may be called from several threads;
 * surfdisp_set_team and the environment knobs are process-wide tuning state.
 *
 * Conventions kept from the reference:
 *   - argument order (Vp, Vs, rho, h, 1/Qs)                       fast_surf.f:2-5,44-45
 *   - kind / ilvry: 1 = Love, 2 = Rayleigh                        models.py:13-16
 *   - last layer is the half-space, its thickness is ignored       flat1.f:69
 *   - a top layer with Vs < 0.1 is water                          fast_surf.f:158,171
 *   - no exceptions: unsolved periods are 0 in c and U             fast_surf.f:197, calcul.f:203-219
 *   - periods must be ascending; outputs depend on the period LIST (mmax carry-over,
 *     start rule c1 = 0.9*c(k-1): calcul.f:112,133,143)
 *   - "fresh process" state for every solve (ndiv = 5, init.f:25)
 */
#ifndef SURFDISP_H
#define SURFDISP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURFDISP_ABI_VERSION 4      /* 4 (r04): + SURFDISP_KERN_REFCOORD, surfdisp_workspace_counters, surfdisp_prior_device, surfdisp_mcmc_propose_masked_device; every ABI-3 symbol kept */
#define SURFDISP_NPER_MAX 200      /* fast_surf.pyf:14-19: cvper and outputs are real*4[200] */
#define SURFDISP_NLAY_MAX 200      /* layers per stack accepted by this library */

#define SURFDISP_KIND_LOVE     1   /* == reference kind0 / ilvry */
#define SURFDISP_KIND_RAYLEIGH 2
#define SURFDISP_INDEPENDENT   0x20 /* OR into `kind`: one team per (stack, period) root search, every
                                     * period started from the first-period rule (fast_surf.f:157-171) on
                                     * a freshly built stack.  P x more parallelism / P x lower latency
                                     * for small batches; equals the default "faithful" mode to ~1e-6 on
                                     * monotone stacks but NOT on rough ones (low-velocity zones), where
                                     * the reference's sequential start rule picks roots and failures
                                     * (SURVEY.md section 4 defects 2, 9).  Caller opts in. */
#define SURFDISP_PIPELINED     0x40 /* OR into `kind`: a launch hint, results are unchanged.  The caller keeps a
                                     * second batch of this size in flight on another stream, so the lanes per
                                     * stack are chosen for twice the stacks (fewer lanes per stack waste fewer
                                     * trial velocities; with one batch alone they would leave SIMDs idle) */
#define SURFDISP_EXACTSCAN     0x80 /* OR into `kind`: every grid point of the scan is evaluated.  Rayleigh: that is the default
                                     * anyway (the flag wins over SURFDISP_FASTSCAN).  Love: the default skips grid points
                                     * between two coarse points that a counting theorem (exact arithmetic) shows free of roots,
                                     * behind fp32 guards whose margins are soaked, not proved: same brackets and results, bit for
                                     * bit, on every random stack tried (scripts/soak_cert.py); this flag walks them all. */
#define SURFDISP_FASTSCAN      0x100 /* OR into `kind`: OPT-IN count-guided scan (Rayleigh; Love's certified scan is on by default).
                                     * By default the secular function is evaluated at EVERY 0.01 km/s grid point from
                                     * 0.9 c(k-1) up to the first sign change, as the reference does (calcul.f:143-166).
                                     * With this flag teams of 2..8 lanes evaluate every 4th / 6th grid point together with the
                                     * number of mode branches below the trial (a Wittrick-Williams count carried by the
                                     * recursion: exact in exact arithmetic) and skip the points between two trials with EQUAL
                                     * counts; any other interval is rescanned point by point: ~45 % fewer evaluations.  Equal
                                     * counts exclude a root between the two trials unless a branch with a zero-group-velocity
                                     * point is crossed twice there (Rayleigh branches of soft sediments with Vp/Vs near 3 can
                                     * have one; Love branches cannot): bit-identical to the default scan on all but two of
                                     * 1.2e9 random stacks.  Callers who need the reference's root
                                     * selection on every input leave it off.  Also switched on for every call of the process
                                     * by the environment variable SURFDISP_FASTSCAN=1 (read once). */
#define SURFDISP_STRICT        0x200 /* OR into `kind`: verification mode.  EVERY stack is solved by the kernel that restates
                                     * DLTAR4 / DLTAR1 / NEVILL statement by statement (the one the default mode keeps for
                                     * stacks whose secular function leaves the fp32 range): the reference's own matrix-entry
                                     * arithmetic, overflow points and evaluation sequence, several times slower.  The default
                                     * mode agrees with it to ~1e-6 on c; use it to check that on your own models
                                     * (tests/test_gpu_parity.py does, at the bench size), not in production. */
#define SURFDISP_KERN_REFCOORD 0x400 /* OR into `kind` of surfdisp_forward_kernels_device: the analytic partials in the REFERENCE's
                                     * coordinates - with respect to the earth-flattened, attenuation-corrected layer values the
                                     * eigenproblem is solved for (no chain factors of calcul.f:122-126 / flat1.f:44-62), i.e. what
                                     * REIGEN / LEIGEN leave in COMMON /rar1/ (surfa.f:1133-1135, 1182-1184, 1204-1207; Love 511-512,
                                     * 564-565, 582-583) summed over each layer's sublayers; a water layer's own share (which the
                                     * reference does not form) is left out.  The verification mode the reference-generated fixture
                                     * tests/golden/ref_partials.npz is compared in; ignored by the other entries. */
#define SURFDISP_PHASE_ONLY    0x10 /* OR into `kind` of the batched entries: phase velocities only
                                     * (what Point.misfit consumes, point.py:18); u is not written
                                     * and may be NULL */

/* per-model status word (the explicit form of the reference's "zeros in c") */
enum {
    SURFDISP_OK         = 0,   /* all P periods solved */
    SURFDISP_PARTIAL    = 1,   /* bracketing failed at period k>1: c,U of periods k..P are 0 (calcul.f:203,218-219) */
    SURFDISP_NOROOT     = 2,   /* bracketing failed at the first period: everything 0 (calcul.f:203-212) */
    SURFDISP_BADMODEL   = 4,   /* nlay < 2, nlay > Lmax, or non-finite input: everything 0 */
    SURFDISP_NUMERIC    = 8    /* a root at or above 16 km/s: one fp32 ulp (1.9e-6) exceeds NEVILL's 1e-6 bracket
                                * tolerance, the reference exhausts its 50 cycles and the call returns nothing
                                * (surfa.f:17-27, calcul.f:172-189): everything 0, also the periods already solved.
                                * (A secular function that overflows fp32 is not an error - the reference returns
                                * the edge of the overflowed region as a root and so does this library.) */
};

/* return codes */
enum {
    SURFDISP_SUCCESS        = 0,
    SURFDISP_ERR_INVALID    = -1,  /* bad argument (NULL pointer, B<1, P<1 or >200, kind, L range) */
    SURFDISP_ERR_NO_DEVICE  = -2,  /* no gfx950 device / HIP runtime unavailable -- there is NO CPU fallback */
    SURFDISP_ERR_HIP        = -3,  /* a HIP call failed; see surfdisp_last_error() */
    SURFDISP_ERR_WORKSPACE  = -4   /* workspace too small */
};

/* ---- (1) Fortran-ABI drop-in: same symbol name and by-reference convention as the object the
 *          reference builds from fast_surf.f:2-5 (gfortran / flang lower-case + underscore).
 *          Runs ONE stack through the GPU path.  uR/uL/cR/cL are float[200]; only the pair that
 *          matches *kind is written, and only its first imax entries (fast_surf.f:197-208) --
 *          the caller pre-zeroes them exactly as f2py does (intent(out) arrays are zero-filled). */
void fast_surf_(const int *n_layer, const int *kind,
                const float *vp, const float *vs, const float *rho,
                const float *h, const float *qsinv,
                const float *per /*[200]*/, const int *nper,
                float *uR, float *uL, float *cR, float *cL);

/* ---- (2) batched solve, host buffers.
 *   model  [B][5][Lmax]  rows = vp, vs, rho, h, qsinv (the five fast_surf.f:2-5 layer arrays)
 *   nlay   [B] or NULL (every stack has Lmax layers); unused tail entries of a row are ignored
 *   per    [P] ascending periods (s);   kind = 1 Love | 2 Rayleigh
 *   c, u   [B][P] phase / group velocity (km/s), 0 where unsolved;  status [B] or NULL
 * Copies in, runs the kernels on `device`, copies out.  Returns SURFDISP_SUCCESS or an error.
 * The calling THREAD keeps what the call needed on the device for its next call (fast_surf_ too): a small arena + pinned
 * staging buffer for small calls, a grow-only buffer (given back at once beyond 3 GiB) and three streams for large ones,
 * which - stacks of up to 20 layers - go through the device in chunks of ~32 768 stacks taking turns on those streams
 * (copies beside kernels).  surfdisp_thread_release() frees the calling thread's share (a thread that is about to exit
 * calls it; a thread pool need not). */
int surfdisp_forward_batch(int device, int B, int Lmax, const int *nlay, const float *model,
                           int P, const float *per, int kind,
                           float *c, float *u, int *status);
void surfdisp_thread_release(void);

/* ---- (3) batched solve, DEVICE pointers, stream-ordered, no allocation, no host sync:
 *          safe to capture in a hipGraph.  `stream` is a hipStream_t (NULL = default stream).
 *          `workspace` = device buffer of at least surfdisp_workspace_bytes(B, Lmax, P) bytes.
 *          All pointers (nlay, model, per, c, u, status, workspace) are device pointers on the
 *          current device.  nlay and status may be NULL. */
size_t surfdisp_workspace_bytes(int B, int Lmax, int P);
int surfdisp_forward_batch_device(void *stream, int B, int Lmax, const int *nlay,
                                  const float *model, int P, const float *per, int kind,
                                  float *c, float *u, int *status,
                                  void *workspace, size_t workspace_bytes);

/* ---- (3b) ABI 3: the same solve, which also returns the Rayleigh ELLIPTICITY the reference computes and keeps in
 *          COMMON /o/ ratio(k, 1) (calcul.f:195: ratio = dltar(c1, t1, 3), i.e. DLTAR4 with mup = 2, surfa.f:360-363;
 *          f2py exposes the block as a module attribute, fast_surf.pyf:126-140): ratio [B][P], 0 where unsolved.
 *          `ratio` may be NULL (then identical to (3)); with SURFDISP_PHASE_ONLY the ellipticity recursions are still
 *          run when `ratio` is given.  Love: zeros.  surfdisp_forward_batch_device stays for ABI 2 callers. */
int surfdisp_forward_batch_device2(void *stream, int B, int Lmax, const int *nlay,
                                   const float *model, int P, const float *per, int kind,
                                   float *c, float *u, float *ratio, int *status,
                                   void *workspace, size_t workspace_bytes);

/* ---- (4) measurement variant of (3): identical launches bracketed by HIP events recorded on
 *          `stream`; blocks until done; kernel_ms[3] = durations of the prep, phase (root search)
 *          and group-velocity kernels in milliseconds.  Used by bench.py's roofline figures. */
int surfdisp_forward_batch_device_timed(void *stream, int B, int Lmax, const int *nlay,
                                        const float *model, int P, const float *per, int kind,
                                        float *c, float *u, int *status,
                                        void *workspace, size_t workspace_bytes, float *kernel_ms);

/* ---- (5) non-blocking measurement: the caller creates events (surfdisp_events_create), passes
 *          four per call; they are recorded on `stream` before prep, after prep, after the root
 *          search (and its idle fallback launch) and after ellipticity kernel + group + finish.  Read with surfdisp_events_elapsed_ms once the caller has
 *          synchronised.  bench.py records them inside its timed region. */
int surfdisp_forward_batch_device_events(void *stream, int B, int Lmax, const int *nlay,
                                         const float *model, int P, const float *per, int kind,
                                         float *c, float *u, int *status,
                                         void *workspace, size_t workspace_bytes, void *const *events4);
int surfdisp_events_create(int n, void **events);
int surfdisp_events_destroy(int n, void **events);
int surfdisp_events_elapsed_ms(void *start, void *stop, float *ms);
/*          surfdisp_stream_wait_event: `stream` waits for one of those events (recorded on another stream) - a caller with two
 *          solves on two streams orders their kernels with it (pysurfinv_amd.forward.JointPlan). */
int surfdisp_stream_wait_event(void *stream, void *event);

/* ---- (5b) forward solve + analytic sensitivity kernels.  REIGEN / LEIGEN form the
 *          partial derivatives of the phase velocity from their energy integrals and never return
 *          them (surfa.f:1130-1135, 1180-1183, 1204-1207; Love 561-565, 584-585); senskernel.py
 *          re-derives them by 2L+1 perturbed solves (senskernel.py:129-158).  Here they come out of
 *          the same group-velocity kernel launch: dcdb / dcda / dcdr [B][P][Lmax] = d c(period) /
 *          d (Vs | Vp | rho) of input layer i in (km/s)/(km/s) resp. (km/s)/(g/cm^3), with respect
 *          to the CALLER's layer values (the chain factors of the attenuation correction
 *          calcul.f:122-126 and of the earth flattening flat1.f:44-62 are applied); zero for water
 *          layers, layers below the effective half space and unsolved periods.  dcda, dcdr may be
 *          NULL; Love has no dcda (written as zeros if given).  Workspace: surfdisp_kernels_workspace_bytes
 *          (every layer's share is stored once, unscaled, in a layer-major scratch inside it - coalesced - and a
 *          transposition kernel applies the common factor 1 / (dL/dk) and writes whole rows); a workspace of only
 *          surfdisp_workspace_bytes is accepted and takes the direct, slower route (same values, bit for bit). */
size_t surfdisp_kernels_workspace_bytes(int B, int Lmax, int P);
int surfdisp_forward_kernels_device(void *stream, int B, int Lmax, const int *nlay,
                                    const float *model, int P, const float *per, int kind,
                                    float *c, float *u, int *status,
                                    float *dcdb, float *dcda, float *dcdr,
                                    void *workspace, size_t workspace_bytes);

/* ---- (6) parameters -> layer stacks on the device (the row next to the hot path, SURVEY.md 8f-2:
 *          Model1D.seisPropLayers, models.py:72-102 + layers.py:139-284) for models with a static
 *          layer structure.  params [C][N] fp64, model [C][5][L] fp32 (rows vp, vs, rho, h, 1/Qs);
 *          idesc / fdesc: descriptor built by pysurfinv_amd.layers_batch (layout in
 *          csrc/surfdisp_layers.hip).  All device pointers; stream-ordered; graph-capturable. */
int surfdisp_params_to_model_device(void *stream, int C, int N, int L, const double *params,
                                    const int *idesc, const double *fdesc, float *model);
/*          The same for a model whose mantle is the thermal OceanMantleHybrid layer (SURVEY.md 8f-4:
 *          layers.py:297-363 over ThermSeis.py HSCM / OceanSeisRitz / OceanSeisRuan): one more kernel
 *          (csrc/surfdisp_thermal.hip) evaluates the half-space cooling models, the mineral physics /
 *          anelasticity and the merge spline, one wavefront per chain, into the caller-owned scratch
 *          (surfdisp_thermal_scratch_bytes(C) bytes of device memory). */
size_t surfdisp_thermal_scratch_bytes(int C);
int surfdisp_params_to_model_thermal_device(void *stream, int C, int N, int L, const double *params,
                                            const int *idesc, const double *fdesc,
                                            void *scratch, size_t scratch_bytes, float *model);

/* ---- (6b) the Metropolis glue of one lock step on the device (csrc/surfdisp_mcmc.hip).
 *   propose: every random-walk scalar of every chain moves by a bounded Gaussian step, redrawn while it falls outside
 *            (vmin, vmax), at most 1000 tries, then a uniform draw (BrownianVar.move, brownian.py:20-27); reset != 0: a
 *            uniform prior draw for every entry (MCinv.reset, models.py:206-219).  p, out [C][N]; vmin, vmax, step [N].
 *   accept : misfit of the proposals' predicted curves c[C][P] (fp32, as the solver returns them) against the
 *            observations (chi2 = sum(((cO-cP)/uncer)^2) over the masked-in periods, misfit = sqrt(chi2/N), chi2 :=
 *            sqrt(50 chi2) when >= 50, L = exp(-chi2/2); a failed solve = (88888, 88888, 0): point.py:15-31), accept rule
 *            chi1 < chi0 or u > 1 - exp(-(chi1-chi0)/2) (point.py:34-37), p0 / chi0 updated in place, and the mcTrack row
 *            [misfit, L, accepted, *proposal] (models.py:254-256) written to row + chain * row_stride (doubles) if row.
 *            c_obs / uncer / mask are [P], or [C][P] with obs_per_chain.  first != 0: a chain's first row (accepted).
 *   Random numbers: Philox4x32-10 keyed by `seed`; the caller passes a fresh `counter` per call.  chain0: the index, in
 *   the whole sampler, of this call's chain 0 - the random streams are indexed by chain0 + c, so a sampler that advances
 *   its chains in several groups (one call per group, e.g. on several streams) draws exactly what one call over all
 *   chains draws.  Device pointers, stream-ordered, no host synchronisation, graph-capturable. */
int surfdisp_mcmc_propose_device(void *stream, int C, int N, const double *p, const double *vmin, const double *vmax,
                                 const double *step, unsigned long long seed, unsigned long long counter, int reset, double *out,
                                 long chain0);
int surfdisp_mcmc_accept_device(void *stream, int C, int N, int P, const float *c, const int *status,
                                const double *c_obs, const double *uncer, const unsigned char *mask, int obs_per_chain,
                                const double *p1, double *p0, double *chi0, double *row, long row_stride,
                                unsigned long long seed, unsigned long long counter, int first, long chain0);
/* ---- (6c) prior predicates on the device, so that a sampler with a prior keeps its lock step there.  The reference redraws a
 *   proposal until `isgood(model)` holds (MCinv.perturb / reset, models.py:192-219: up to 1000 Gaussian tries, then uniform
 *   draws); its model classes build `isgood` from a few GENERIC tests on the grid points of seisPropGrids (models.py:294-320):
 *   Vs increasing inside a layer group (monoIncrease), no drop of Vs across a group boundary, Vs below a cap.
 *   surfdisp_prior_device evaluates those for every chain: flags [L] per OUTPUT layer of the params->stack descriptor (bit 0: Vs
 *   must increase across the layer; bit 1: ... and on to the top of the next layer; bit 2: must not drop to the top of the next
 *   layer; bit 3: both its points at most vs_max).  tags [C] (unsigned char): a chain that breaks a rule gets tags[c] = mark_tag (1..255); with
 *   only_tag >= 0 only the chains with tags[c] >= only_tag are looked at - the rounds of one step use rising tags (the caller
 *   clears tags once per step), so nothing needs clearing between rounds.  Models with a static layer structure, no thermal layer.
 *   surfdisp_mcmc_propose_masked_device redraws the chains with tags[c] == tag only (try number `attempt` of this step: its own
 *   random numbers; mode 0 bounded Gaussian step, 1 uniform prior draw, 2 the chain's state itself), leaving the other rows of
 *   `out` as they are.  pysurfinv_amd.mcmc.MetropolisBatch(isgood=PriorRules(...)) strings them into masked redraw rounds. */
int surfdisp_prior_device(void *stream, int C, int N, int L, const double *params, const int *idesc, const double *fdesc,
                          const int *flags, double vs_max, int only_tag, int mark_tag, unsigned char *tags);
int surfdisp_mcmc_propose_masked_device(void *stream, int C, int N, const double *p, const double *vmin, const double *vmax,
                                        const double *step, unsigned long long seed, unsigned long long counter, int attempt, int mode,
                                        const unsigned char *tags, int tag, double *out, long chain0);
/* The SPECULATIVE lock step for few chains (a lock step of 100 chains leaves the chip idle, so it costs no more to solve
 * several proposals per chain): propose_tree draws the binary tree of the next `depth` (1..4) accept / reject outcomes -
 * node k's proposal from the state its branch would be in (node 0: the chain's state; child 2k+1 "accepted": proposal k;
 * child 2k+2 "rejected": the state of k), out [C][2^depth - 1][N] = the C * (2^depth - 1) stacks of ONE batched solve;
 * accept_tree then walks nsteps <= depth steps per chain with the usual test at every node and writes one mcTrack row
 * per step, step_stride doubles apart.  Every proposal is drawn from, and tested against, the state the chain is in at
 * that step, so the chain is distributed exactly as the plain sampler's.  depth = 1 is the plain pair of calls above. */
int surfdisp_mcmc_propose_tree_device(void *stream, int C, int N, int depth, const double *p, const double *vmin,
                                      const double *vmax, const double *step, unsigned long long seed,
                                      unsigned long long counter, double *out, long chain0);
int surfdisp_mcmc_accept_tree_device(void *stream, int C, int N, int P, int depth, int nsteps, const float *c,
                                     const int *status, const double *c_obs, const double *uncer, const unsigned char *mask,
                                     int obs_per_chain, const double *q, double *p0, double *chi0, double *row,
                                     long row_stride, long step_stride, unsigned long long seed, unsigned long long counter,
                                     long chain0);

/* ---- (7) introspection of the two-tier root search.  The production kernel hands the stacks it cannot treat
 *          faithfully to an exact fallback kernel that runs right behind it inside the same call: a secular
 *          function that leaves the fp32 range (the reference's overflow points depend on how it forms its matrix
 *          entries, surfa.f:289-330) and brackets with more than one visible sign change (which root NEVILL,
 *          surfa.f:2-83, lands on depends on its evaluation sequence).  Returns in *count how many stacks (in
 *          SURFDISP_INDEPENDENT mode: (stack, period) units) of the last solve on `workspace` took that path.
 *          Waits for `stream`. */
int surfdisp_workspace_fallback_count(void *stream, const void *workspace, int B, int Lmax, int P, int *count);
/*          ... and counts3[3] = {that count; brackets the root search refined with NEVILL because they may hold several roots
 *          (vertical phase growing by more than 1 rad across the bracket); Rayleigh ellipticities evaluated again with the
 *          reference's own arithmetic (closure cancelling, or c far below the stack's fastest S velocity)}. */
int surfdisp_workspace_counters(void *stream, const void *workspace, int B, int Lmax, int P, int *counts3);

/* ---- tuning / introspection ------------------------------------------------------------- */
/* lanes of one wavefront that cooperate on one stack's root search (1,2,4,...,64); 0 = choose
 * from (B, Lmax).  Also settable through the environment variable SURFDISP_TEAM. */
int  surfdisp_set_team(int lanes);
int  surfdisp_get_team(int B, int Lmax);           /* what a Rayleigh c+U launch with (B, Lmax) would use */
int  surfdisp_get_team2(int B, int Lmax, int P, int kind);   /* ... a launch with these kind flags (wave type, PHASE_ONLY, PIPELINED, INDEPENDENT) */
int  surfdisp_device_count(void);                  /* gfx950 devices visible; <=0: none */
int  surfdisp_abi_version(void);
const char *surfdisp_last_error(void);             /* thread-local, never NULL */
const char *surfdisp_kernel_name(int which);       /* 0 prep, 1 phase (root search), 2 group, 3 finish */

#ifdef __cplusplus
}
#endif
#endif /* SURFDISP_H */
