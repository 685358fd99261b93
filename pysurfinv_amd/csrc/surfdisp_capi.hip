// surfdisp_capi.hip -- the C ABI of libsurfdisp_hip.so (declared in include/surfdisp.h).
// Host-side only: argument checking, workspace carving, stream-ordered launches.  There is no
// CPU fallback: without a HIP device every entry point fails with SURFDISP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "surfdisp_internal.h"

namespace {

thread_local char g_err[512] = "";
// process-wide tuning state (surfdisp_set_team / surfdisp_debug_buffer): atomics, so concurrent callers race
// benignly; a launch reads each value once
std::atomic<int> g_team_override{0};
std::atomic<double *> g_dbg{nullptr};    // developer hook, see surfdisp_debug_buffer

// environment knobs, read ONCE per process (a getenv per launch is visible in launch-bound Metropolis loops)
struct EnvKnobs {
    int team = 0;                 // SURFDISP_TEAM
    int team_love = 0;            // SURFDISP_TEAM_LOVE (developer knob): lanes per stack of Love root searches only
    size_t overlap_max = 64u * 1024u;   // SURFDISP_OVERLAP_MAX
    size_t lds_budget = 44u * 1024u;    // SURFDISP_LDS_BUDGET (developer knob): root-search LDS per 256 lanes
    size_t lds_budget_pipelined = 44u * 1024u;   // SURFDISP_LDS_BUDGET_PIPELINED (developer knob): ... of a SURFDISP_PIPELINED launch (64 KB until r03)
    float refine_wtol = 1.2e-3f;  // SURFDISP_WTOL
    float refine_atol = 1.0e-6f;  // SURFDISP_ATOL
    float phimax = 0.7853982f;    // SURFDISP_SCAN_PHASE (fast scan only; developer knob)
    float ambig = 3.0e-5f;        // SURFDISP_AMBIG (developer knob): scan trials below this fraction of their terms' magnitude are
                                  // evaluated again with the reference's arithmetic (0 = off)
    float ell_ambig = 3.0e-3f;    // SURFDISP_ELL_AMBIG (developer knob): ellipticity closures below this fraction of their terms' magnitude are
                                  // evaluated again with the reference's arithmetic (0 = off)
    float ell_gmax = 25.0f;       // SURFDISP_ELL_GMAX (developer knob): ... and where 2 b^2 / c^2 of the stack's fastest layer exceeds this
    int group_order = -1;         // SURFDISP_GROUP_ORDER (developer knob): workgroup order of the group-velocity kernel (-1: the library's rule; 0: plain period-major; g: XCD-aware, g stack blocks at a time)
    float phimulti = 1.0f;        // SURFDISP_PHIMULTI (developer knob): vertical-phase growth (rad) across a bracket beyond which NEVILL refines it
    bool fastscan = false;        // SURFDISP_FASTSCAN=1: opt every call of the process into the heuristic scan
    int device = 0;               // SURFDISP_DEVICE (fast_surf_)
    int balance = -1;             // SURFDISP_BALANCE (developer knob): wavefront priority by progress, -1 = automatic
    int lockstep = -1;            // SURFDISP_LOCKSTEP (developer knob): -1 = automatic (on), 0 / 1, 2 = also for (stack, period) units
    int certscan = 1;             // SURFDISP_CERTSCAN (developer knob): 0 = Love root searches walk every grid point (no certified skipping)
    int host_slots = 3;           // SURFDISP_HOST_SLOTS (developer knob): chunks in flight of a large host-buffer call (2 or 3)
    long host_chunk_layers = 327680;   // SURFDISP_HOST_CHUNK (developer knob): layers' worth of stacks per chunk
    int host_pipeline = 1;        // SURFDISP_HOST_PIPELINE (developer knob): 0 = large host-buffer calls as one chunk
    int rows_min_team = 8;        // SURFDISP_ROWS_MIN_TEAM (developer knob): teams of at least this many lanes rebuild from the row copy
#ifdef SD_ELL_INKERNEL_WIDE
    int ell_kernel = 0;           // A/B build: the ellipticity recursions inside the root search for every team size (r02)
#else
    int ell_kernel = 1;           // the ellipticity kernel for teams of >= 4 lanes
#endif
    EnvKnobs()
    {
        if (const char *e = getenv("SURFDISP_TEAM")) team = atoi(e);
        if (const char *e = getenv("SURFDISP_TEAM_LOVE")) team_love = atoi(e);
        if (const char *e = getenv("SURFDISP_OVERLAP_MAX")) overlap_max = (size_t)atol(e);
        if (const char *e = getenv("SURFDISP_LDS_BUDGET")) lds_budget = lds_budget_pipelined = (size_t)atol(e);
        if (const char *e = getenv("SURFDISP_LDS_BUDGET_PIPELINED")) lds_budget_pipelined = (size_t)atol(e);
        if (const char *e = getenv("SURFDISP_WTOL")) refine_wtol = (float)atof(e);
        if (const char *e = getenv("SURFDISP_ATOL")) refine_atol = (float)atof(e);
        if (const char *e = getenv("SURFDISP_SCAN_PHASE")) phimax = (float)atof(e);
        if (const char *e = getenv("SURFDISP_AMBIG")) ambig = (float)atof(e);
        if (const char *e = getenv("SURFDISP_PHIMULTI")) phimulti = (float)atof(e);
        if (const char *e = getenv("SURFDISP_GROUP_ORDER")) group_order = atoi(e);
        if (const char *e = getenv("SURFDISP_ELL_AMBIG")) ell_ambig = (float)atof(e);
        if (const char *e = getenv("SURFDISP_ELL_GMAX")) ell_gmax = (float)atof(e);
        if (const char *e = getenv("SURFDISP_FASTSCAN")) fastscan = atoi(e) != 0;
        if (const char *e = getenv("SURFDISP_DEVICE")) device = atoi(e);
        if (const char *e = getenv("SURFDISP_BALANCE")) balance = atoi(e);
        if (const char *e = getenv("SURFDISP_ROWS_MIN_TEAM")) rows_min_team = atoi(e);
        if (const char *e = getenv("SURFDISP_LOCKSTEP")) lockstep = atoi(e);
        if (const char *e = getenv("SURFDISP_HOST_PIPELINE")) host_pipeline = atoi(e);
        if (const char *e = getenv("SURFDISP_CERTSCAN")) certscan = atoi(e);
        if (const char *e = getenv("SURFDISP_HOST_SLOTS")) host_slots = atoi(e);
        if (const char *e = getenv("SURFDISP_HOST_CHUNK")) { host_chunk_layers = atol(e); if (host_chunk_layers < 1024) host_chunk_layers = 1024; }
    }
};
const EnvKnobs &knobs() { static const EnvKnobs k; return k; }

void set_err(const char *fmt, const char *a = "", const char *b = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a, b);
}

#define SD_HIP(call)                                                         \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) {                                              \
            set_err("%s failed: %s", #call, hipGetErrorString(e_));          \
            return SURFDISP_ERR_HIP;                                         \
        }                                                                    \
    } while (0)

constexpr int SD_KIND_FLAGS = SURFDISP_PHASE_ONLY | SURFDISP_INDEPENDENT | SURFDISP_PIPELINED | SURFDISP_EXACTSCAN |
                             SURFDISP_FASTSCAN | SURFDISP_STRICT | SURFDISP_KERN_REFCOORD;

size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Carve {
    float *mdl, *rows, *ratio, *ct, *ut;
    int *nl, *nsolved, *hist;
    float *fsafe, *ovf;
    int *fb_count, *fb_list;
    int *amb_count;       // [1] scan trials evaluated again with the reference's arithmetic (behind fb_count: zeroed with it)
    size_t total;
};

Carve carve(void *base, int B, int Lmax, int P)
{
    char *p = static_cast<char *>(base);
    size_t off = 0;
    Carve c;
    c.mdl = reinterpret_cast<float *>(p + off);     off += align_up((size_t)9 * Lmax * B * sizeof(float));      // sd::NF staged fields, SoA
    c.rows = reinterpret_cast<float *>(p + off);    off += align_up((size_t)9 * Lmax * B * sizeof(float));      // ... and one row per stack and field
    c.ratio = reinterpret_cast<float *>(p + off);   off += align_up((size_t)P * B * sizeof(float));
    c.ct = reinterpret_cast<float *>(p + off);      off += align_up((size_t)P * B * sizeof(float));
    c.ut = reinterpret_cast<float *>(p + off);      off += align_up((size_t)P * B * sizeof(float));
    c.hist = reinterpret_cast<int *>(p + off);      off += align_up((size_t)P * B * sizeof(int));
    c.nl = reinterpret_cast<int *>(p + off);        off += align_up((size_t)B * sizeof(int));
    c.nsolved = reinterpret_cast<int *>(p + off);   off += align_up((size_t)B * sizeof(int));
    c.fsafe = reinterpret_cast<float *>(p + off);   off += align_up((size_t)B * sizeof(float));
    c.ovf = reinterpret_cast<float *>(p + off);     off += align_up((size_t)3 * B * sizeof(float));
    c.fb_count = reinterpret_cast<int *>(p + off);  c.amb_count = c.fb_count + 1;  off += align_up(3 * sizeof(int));
    c.fb_list = reinterpret_cast<int *>(p + off);   off += align_up((size_t)P * B * sizeof(int));
    c.total = off;
    return c;
}

int pick_team(int B, int Lmax, bool need_ratio = true, bool pipelined = false, int kind = SURFDISP_KIND_RAYLEIGH)
{
    int G = g_team_override.load(std::memory_order_relaxed);
    if (G == 0 && kind == SURFDISP_KIND_LOVE) G = knobs().team_love;
    if (G == 0) G = knobs().team;
    if (G == 0) {
        // measured on MI355X (scripts/sweep_team.py, profiles/r02e/sweep_team.txt): at least three wavefronts per
        // SIMD (196 608 lanes on 256 CUs x 4 SIMDs) whatever the batch size, tiny batches a whole wavefront per
        // stack (latency); never fewer than 2 lanes per stack
        const long target = 196608L;
        G = 2;
        while (G < 64 && (long)B * G < target) G *= 2;
        // (r03, with the teams of a wavefront in lock step - profiles/r03b/sweep_team_lockstep.txt: for Rayleigh stacks of
        // >= 24 layers teams of more than 16 lanes cost more in wasted evaluations than the third and fourth wavefront per
        // SIMD give back - 8 192 x L64: 1.49 ms with 16 lanes against 1.68 with 32, 8 192 x L30: 0.76 against 0.84 - so
        // beyond 16 lanes they aim at 131 072 lanes only; and a Rayleigh launch that is alone on the chip never takes
        // two-lane teams - 131 072 x L10: 2.17 against 3.08 ms)
        if (kind != SURFDISP_KIND_LOVE && Lmax >= 24 && G > 16) {
            G = 16;
            while (G < 64 && (long)B * G < 131072L) G *= 2;
        }
        if (kind != SURFDISP_KIND_LOVE && !pipelined && G < 4) G = 4;
    }
    // Love evaluations are cheap (a 2-vector recursion, ~35 instructions per layer against Rayleigh's ~95), so a pass's team
    // bookkeeping weighs more: never fewer than 4 lanes (8 until the teams of a wavefront went in lock step, which took most of
    // that bookkeeping away: 65 536 x L10 now 0.65 ms with 4 lanes against 0.70 with 8, 65 536 x L30 1.46 against 1.55), and
    // the caller sizes a Love launch for its own stacks even beside another stream's kernels (forward_device_impl: 16 384 x L64
    // beside the Rayleigh root search of a joint solve, teams of 16 instead of 8: 6.59 -> 6.24 ms); measured with the four-field
    // Love working stack, scripts/sweep_team.py, profiles/r03a/love_team.txt, profiles/r03b/sweep_team_lockstep.txt
    // (with the certified coarse scan of teams of <= 8 lanes two-lane teams win again on very large batches - 131 072 x L10:
    // 0.73 ms against 0.81 with four lanes - so the floor only holds where that scan is switched off)
    if (kind == SURFDISP_KIND_LOVE && !g_team_override.load(std::memory_order_relaxed) && !knobs().team && !knobs().team_love && G < 4 &&
        knobs().certscan == 0)
        G = 4;
    if (G < 1) G = 1;
    if (G > 64) G = 64;
    int p2 = 1;
    while (p2 * 2 <= G) p2 *= 2;
    G = p2;
    // The working stacks of a workgroup's 256/G teams are what limits the workgroups per CU (one slot each: since r03 the
    // ellipticities of teams of >= 4 lanes come from their own kernel; need_ratio only sizes the second slot of an A/B build
    // with -DSD_ELL_INKERNEL_WIDE): keep them within 44 KB per 256 lanes, i.e. at least three workgroups = wavefronts per
    // SIMD, normally four - a wider team wastes fewer evaluations than a half-empty SIMD costs (scripts/sweep_team.py,
    // profiles/r02e/sweep_team.txt; re-checked on the r03 kernels, every auto choice within 1 % of the best forced size).
    const size_t per256 = 256 / SD_PHASE_BLOCK;        // the budget is per 256 lanes
    // (until r03 a caller that keeps another batch in flight - SURFDISP_PIPELINED, the joint Rayleigh + Love plan - got 64 KB =
    // two workgroups: the other stream's wavefronts fill the SIMDs, and the narrower team's fewer evaluations won)
    // (r03: with the teams of a wavefront in lock step the wider team wins there too - joint solve of 16 384 x L64, Rayleigh
    // teams of 16 instead of 8: 5.59 -> 5.32 ms - so the budget is the same 44 KB; SURFDISP_LDS_BUDGET_PIPELINED restores 64 KB)
    const size_t budget = pipelined ? knobs().lds_budget_pipelined : knobs().lds_budget;
    auto lds_of = [&](int g) { return sd::phase_lds_bytes(Lmax, g, need_ratio && g >= 4, kind) * per256; };
    while (G < 64 && lds_of(G) > budget) G *= 2;
    return G;
}

// A/B builds with -DSD_ELL_INKERNEL_WIDE only (the r02 arrangement): the second LDS slot (ellipticity of period k
// evaluated inside the first scan pass of period k+1) saves one pass per period but doubles the workgroup's LDS
// (measured in r02: on at 49 KB is 7-19 % faster than off, on at 74 KB / 147 KB is 9 % / 38 % slower).
static bool use_overlap(int Lmax, int G)
{
    const size_t cap = knobs().overlap_max;
    return G >= 4 && sd::phase_lds_bytes(Lmax, G, true, SURFDISP_KIND_RAYLEIGH) * (256 / SD_PHASE_BLOCK) <= cap;
}

int check_args(int B, int Lmax, int P, int kind, const void *model, const void *per,
               const void *c, const void *u)
{
    const int wave = kind & ~SD_KIND_FLAGS;
    if (B < 1 || Lmax < 2 || Lmax > SURFDISP_NLAY_MAX || P < 1 || P > SURFDISP_NPER_MAX ||
        (wave != SURFDISP_KIND_LOVE && wave != SURFDISP_KIND_RAYLEIGH) || !model || !per || !c ||
        (!u && !(kind & SURFDISP_PHASE_ONLY))) {
        set_err("invalid argument (B>=1, 2<=Lmax<=200, 1<=P<=200, kind 1|2, non-null buffers)");
        return SURFDISP_ERR_INVALID;
    }
    return SURFDISP_SUCCESS;
}

}  // namespace

extern "C" {

int surfdisp_abi_version(void) { return SURFDISP_ABI_VERSION; }

const char *surfdisp_last_error(void) { return g_err; }

const char *surfdisp_kernel_name(int which)
{
    switch (which) {
        case 0: return "surfdisp_prep_kernel";
        case 1: return "surfdisp_phase_kernel";
        case 2: return "surfdisp_group_kernel";
        case 3: return "surfdisp_finish_kernel";
        default: return "";
    }
}

int surfdisp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int surfdisp_set_team(int lanes)
{
    if (lanes < 0 || lanes > 64 || (lanes & (lanes - 1))) {
        set_err("team must be 0 or a power of two <= 64");
        return SURFDISP_ERR_INVALID;
    }
    g_team_override.store(lanes, std::memory_order_relaxed);
    return SURFDISP_SUCCESS;
}

int surfdisp_get_team(int B, int Lmax) { return surfdisp_get_team2(B, Lmax, 1, SURFDISP_KIND_RAYLEIGH); }   // a Rayleigh c+U call, no flags

// lanes per stack a launch with these flags would use (kind: 1 | 2 with SURFDISP_PHASE_ONLY / _PIPELINED / _INDEPENDENT
// OR'd in; P: periods, used by the independent decomposition) - what forward_device_impl computes
int surfdisp_get_team2(int B, int Lmax, int P, int kind)
{
    const bool phase_only = (kind & SURFDISP_PHASE_ONLY) != 0, indep = (kind & SURFDISP_INDEPENDENT) != 0;
    const bool pipelined = (kind & SURFDISP_PIPELINED) != 0, strict = (kind & SURFDISP_STRICT) != 0;
    const int wave = kind & ~SD_KIND_FLAGS;
    const bool want_ell = (wave == SURFDISP_KIND_RAYLEIGH) && !phase_only;
    const long units = (indep ? (long)B * (P > 0 ? P : 1) : (long)B) * ((pipelined && wave != SURFDISP_KIND_LOVE) ? 2 : 1);
    const bool ell_k_ok = want_ell && knobs().ell_kernel != 0 && !strict;
    return pick_team((int)(units > 0x3fffffff ? 0x3fffffff : units), Lmax, want_ell && !ell_k_ok, pipelined, wave);
}

// developer hook (not in include/surfdisp.h): device buffer of B*P*16 doubles receiving
// group-velocity intermediates of the next launches; nullptr switches it off.
void surfdisp_debug_buffer(double *dev) { g_dbg.store(dev, std::memory_order_relaxed); }

size_t surfdisp_workspace_bytes(int B, int Lmax, int P)
{
    if (B < 1 || Lmax < 2 || P < 1) return 0;
    return carve(nullptr, B, Lmax, P).total;
}

// workspace of surfdisp_forward_kernels_device with room for the layer-major scratch [3][Lmax][P][B] the analytic
// partials are accumulated in (the lanes of a wavefront - consecutive stacks, one period - then touch consecutive
// words instead of rows 4 P Lmax bytes apart); a workspace of only surfdisp_workspace_bytes still works, slower
size_t surfdisp_kernels_workspace_bytes(int B, int Lmax, int P)
{
    if (B < 1 || Lmax < 2 || P < 1) return 0;
    // + per unit its factor 1 / (dL/dk) and the deepest layer it wrote
    return align_up(carve(nullptr, B, Lmax, P).total) + align_up((size_t)3 * Lmax * P * B * sizeof(float)) +
           2 * align_up((size_t)P * B * sizeof(float));
}

// introspection: how many stacks (or (stack, period) units in independent mode) the last solve that used this
// workspace handed to the exact fallback kernel.  Waits for `stream`.
int surfdisp_workspace_fallback_count(void *stream, const void *workspace, int B, int Lmax, int P, int *count)
{
    if (!workspace || !count || B < 1 || Lmax < 2 || P < 1) { set_err("bad argument"); return SURFDISP_ERR_INVALID; }
    const Carve w = carve(const_cast<void *>(workspace), B, Lmax, P);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SD_HIP(hipMemcpyAsync(count, w.fb_count, sizeof(int), hipMemcpyDeviceToHost, s));
    SD_HIP(hipStreamSynchronize(s));
    return SURFDISP_SUCCESS;
}

// introspection: {stacks handed to the exact fallback kernel, brackets sent to NEVILL because the vertical phase grows by more
// than SURFDISP_PHIMULTI across them (-DSD_AMBIG builds: + scan trials evaluated again), ellipticities evaluated again with the
// reference's arithmetic} of the last solve on `workspace`.  Waits for `stream`.
int surfdisp_workspace_counters(void *stream, const void *workspace, int B, int Lmax, int P, int *counts3)
{
    if (!workspace || !counts3 || B < 1 || Lmax < 2 || P < 1) { set_err("bad argument"); return SURFDISP_ERR_INVALID; }
    const Carve w = carve(const_cast<void *>(workspace), B, Lmax, P);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SD_HIP(hipMemcpyAsync(counts3, w.fb_count, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    SD_HIP(hipStreamSynchronize(s));
    return SURFDISP_SUCCESS;
}

static int forward_device_impl(void *stream, int B, int Lmax, const int *nlay,
                               const float *model, int P, const float *per, int kind,
                               float *c, float *u, int *status,
                               void *workspace, size_t workspace_bytes, hipEvent_t *ev,
                               float *kb = nullptr, float *ka = nullptr, float *kr = nullptr, float *ratio = nullptr)
{
    int rc = check_args(B, Lmax, P, kind, model, per, c, u);
    if (rc) return rc;
    if (!workspace || workspace_bytes < surfdisp_workspace_bytes(B, Lmax, P)) {
        set_err("workspace too small");
        return SURFDISP_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool phase_only = (kind & SURFDISP_PHASE_ONLY) != 0;
    const bool indep = (kind & SURFDISP_INDEPENDENT) != 0;
    const bool pipelined = (kind & SURFDISP_PIPELINED) != 0;
    // the reference's point-by-point scan unless the caller (flag) or the process (environment) opted into the
    // heuristic one; SURFDISP_EXACTSCAN (ABI 1) is accepted and wins over both
    const EnvKnobs &kn = knobs();
    const bool strict = (kind & SURFDISP_STRICT) != 0;
    const bool exactscan = (kind & SURFDISP_EXACTSCAN) != 0;
    const bool kern_raw = (kind & SURFDISP_KERN_REFCOORD) != 0;
    bool fastscan = ((kind & SURFDISP_FASTSCAN) != 0 || kn.fastscan) && !exactscan && !strict;
    kind &= ~SD_KIND_FLAGS;
    // Love: the coarse scan with its Sturm-count certificate (phase_body, CERT) is the DEFAULT - it returns the bracket the
    // point-by-point scan returns, by a theorem, not a heuristic; SURFDISP_EXACTSCAN / SURFDISP_CERTSCAN=0 walk every point
#if defined(SD_RCERT) && SD_RCERT
    fastscan = !exactscan && !strict && (kn.certscan != 0 || fastscan);      // (experimental build: the Rayleigh count drives the scan too)
#else
    if (kind == SURFDISP_KIND_LOVE) fastscan = !exactscan && !strict && (kn.certscan != 0 || fastscan);
#endif
    // the ellipticity recursions (two more evaluations per period) feed the group-velocity kernel - and the caller who
    // asked for the ratio itself (ABI 3), also in a phase-only call
    const bool want_ell = (kind == SURFDISP_KIND_RAYLEIGH) && (!phase_only || ratio != nullptr);
    const Carve w = carve(workspace, B, Lmax, P);
    // independent mode has B*P root searches in flight: size the teams for that many; a caller that
    // keeps a second batch in flight (SURFDISP_PIPELINED) has twice the stacks on the chip
    const long units = (indep ? (long)B * P : (long)B) * ((pipelined && kind != SURFDISP_KIND_LOVE) ? 2 : 1);
    // The ellipticities come from their own kernel (one lane per (stack, period), replaying the working stack's history):
    // the root search then runs as in a phase-only call - no riders, no second LDS slot.  The exact fallback kernel
    // (and SURFDISP_STRICT) keeps them in-kernel.
    // Teams of two lanes keep them too: their ellipticity pass has both lanes busy, one start vector each (three batches in
    // flight, 65 536 x L10: 34.0 M solves/s in-kernel against 33.6 M with the extra kernel).
    const bool ell_k_ok = want_ell && kn.ell_kernel != 0 && !strict;
    const int G = pick_team((int)(units > 0x3fffffff ? 0x3fffffff : units), Lmax, want_ell && !ell_k_ok, pipelined, kind);
    const bool ell_k = ell_k_ok && G >= 4;
    const bool ell_in = want_ell && !ell_k;

    // Staged copy of the model the root search rebuilds its working stack from, once per period: with >= 8 lanes per
    // stack consecutive lanes take consecutive layers, so the fields are laid out one row per stack (coalesced; from the
    // SoA copy every value costs its own cache line: 12.5 GB of fabric traffic per 25 600 x 96-layer lock step,
    // profiles/r03a).  Narrow teams read the SoA copy (consecutive stacks share a line).  The group-velocity kernel - one
    // lane per (stack, period) - always reads the SoA copy; a phase-only call with rows skips writing it (the exact
    // fallback launch reads whatever the production launch read).
    const bool use_rows = (G >= kn.rows_min_team);
    const bool need_soa = !use_rows || !phase_only || want_ell;
    sd::PrepArgs pa{B, Lmax, nlay, model, w.mdl, w.nl, P, indep ? w.nsolved : nullptr, w.fsafe, w.ovf, w.fb_count,
                    use_rows ? w.rows : nullptr, need_soa ? 1 : 0};
    if (ev) SD_HIP(hipEventRecord(ev[0], s));
    SD_HIP(sd::launch_prep(s, kind, pa));
    if (ev) SD_HIP(hipEventRecord(ev[1], s));
    const float wtol = kn.refine_wtol, atol = kn.refine_atol, phimax = kn.phimax;
    sd::PhaseArgs ph{B, Lmax, P, w.mdl, w.nl, per, w.ct, ell_in ? w.ratio : nullptr, w.nsolved, status, wtol, atol,
                     fastscan ? 1 : 0, w.fsafe, (ell_in && use_overlap(Lmax, G)) ? 1 : 0, phimax,
                     w.ovf, w.fb_count, w.fb_list, kn.balance >= 0 ? kn.balance : ((pipelined && G < 8) ? 0 : 1), strict ? 1 : 0};
#ifdef SD_WAVECLOCK
    ph.wclk = reinterpret_cast<unsigned long long *>(g_dbg.load(std::memory_order_relaxed));
#endif
    if (use_rows) { ph.msrc = w.rows; ph.ms_b = 9L * Lmax; ph.ms_f = Lmax; ph.ms_i = 1; }
    else          { ph.msrc = w.mdl;  ph.ms_b = 1;         ph.ms_f = (long)Lmax * B; ph.ms_i = B; }
    // (two-lane teams compute their ellipticities themselves but record the history too: the pairs whose closure cancels
    // are redone by the ellipticity kernel)
    const bool ell_fix = ell_in && !strict && kn.ell_kernel != 0 && kn.ell_ambig != 0.0f;
    ph.hist = (ell_k || ell_fix) ? w.hist : nullptr;
    ph.lockstep = kn.lockstep >= 0 ? kn.lockstep : 1;
    ph.ambig = kn.ambig; ph.phimulti = kn.phimulti; ph.amb_count = w.amb_count; ph.ell_ambig = ell_fix ? kn.ell_ambig : 0.0f; ph.ell_gmax = kn.ell_gmax;
    SD_HIP(sd::launch_phase(s, kind, G, indep, ph));
    // the exact fallback re-solves what the production kernel listed (normally nothing: idle blocks exit at once)
    ph.overlap = 0; ph.fast = 0;
    if (ell_k) ph.ratio = w.ratio;                         // ... with its ellipticities in-kernel (marked -1 in hist)
    SD_HIP(sd::launch_phase_exact(s, kind, indep, ph));
    if (ev) SD_HIP(hipEventRecord(ev[2], s));              // [ev1, ev2] = the root search (+ its idle fallback launch)
    if (ell_k || ell_fix) {
        sd::EllipArgs ea{B, Lmax, P, w.mdl, w.nl, per, w.ct, w.hist, w.nsolved, w.ratio, kn.ell_ambig, ell_k ? 0 : 1, w.amb_count, kn.ell_gmax, w.ovf};
        SD_HIP(sd::launch_ellip(s, ea));
    }
#ifdef SD_WAVECLOCK
    double *gdbg = nullptr;                                // the debug buffer holds wavefront clocks in this build
#else
    double *gdbg = g_dbg.load(std::memory_order_relaxed);
#endif
    // analytic partials: through the layer-major scratch when the caller's workspace has room for it
    float *kscr = nullptr, *kscale = nullptr;
    int *khs = nullptr;
    if (kb && workspace_bytes >= surfdisp_kernels_workspace_bytes(B, Lmax, P)) {
        char *q = static_cast<char *>(workspace) + align_up(w.total);
        kscr = reinterpret_cast<float *>(q);    q += align_up((size_t)3 * Lmax * P * B * sizeof(float));
        kscale = reinterpret_cast<float *>(q);  q += align_up((size_t)P * B * sizeof(float));
        khs = reinterpret_cast<int *>(q);
    }
    sd::GroupArgs ga{B, Lmax, P, w.mdl, w.nl, per, w.ct, w.ratio, w.nsolved, w.ut, gdbg, kb, ka, kr, kscr, kscale, khs,
                     kern_raw ? 1 : 0, 0, 0, kn.group_order};
    if (!phase_only) SD_HIP(sd::launch_group(s, kind, ga));
    if (!phase_only && kscr) {
        // one launch for the three arrays: factor 1 / (dL/dk), zeros below each unit's half space, whole rows
        sd::KernTransposeArgs ta{B, P, Lmax, kind, kscr, kscale, khs, kb, ka, kr};
        SD_HIP(sd::launch_kern_transpose(s, ta));
    }
    if (ratio && kind != SURFDISP_KIND_RAYLEIGH) SD_HIP(hipMemsetAsync(ratio, 0, (size_t)B * P * sizeof(float), s));   // Love: zeros
    sd::FinishArgs fa{B, P, w.ct, phase_only ? nullptr : w.ut, c, phase_only ? nullptr : u,
                      indep ? w.nsolved : nullptr, w.nl, status,
                      (ratio && want_ell) ? w.ratio : nullptr, (ratio && want_ell) ? ratio : nullptr, w.nsolved};
    SD_HIP(sd::launch_finish(s, fa));
    if (ev) SD_HIP(hipEventRecord(ev[3], s));
    return SURFDISP_SUCCESS;
}

int surfdisp_forward_batch_device(void *stream, int B, int Lmax, const int *nlay,
                                  const float *model, int P, const float *per, int kind,
                                  float *c, float *u, int *status,
                                  void *workspace, size_t workspace_bytes)
{
    return forward_device_impl(stream, B, Lmax, nlay, model, P, per, kind, c, u, status,
                               workspace, workspace_bytes, nullptr);
}

// ABI 3: the same solve, also returning the Rayleigh ellipticity the reference keeps in COMMON /o/ (calcul.f:195).
int surfdisp_forward_batch_device2(void *stream, int B, int Lmax, const int *nlay,
                                   const float *model, int P, const float *per, int kind,
                                   float *c, float *u, float *ratio, int *status,
                                   void *workspace, size_t workspace_bytes)
{
    return forward_device_impl(stream, B, Lmax, nlay, model, P, per, kind, c, u, status,
                               workspace, workspace_bytes, nullptr, nullptr, nullptr, nullptr, ratio);
}

// Forward solve + analytic partial derivatives of the phase velocity with respect to each layer's
// Vs, Vp, rho (what REIGEN/LEIGEN compute and discard, surfa.f:1130-1135,1204-1207 /
// 561-565,584-585), from the energy integrals the group-velocity kernel forms anyway.
int surfdisp_forward_kernels_device(void *stream, int B, int Lmax, const int *nlay,
                                    const float *model, int P, const float *per, int kind,
                                    float *c, float *u, int *status,
                                    float *dcdb, float *dcda, float *dcdr,
                                    void *workspace, size_t workspace_bytes)
{
    if (!dcdb) { set_err("surfdisp_forward_kernels_device: dcdb is NULL"); return SURFDISP_ERR_INVALID; }
    if (kind & SURFDISP_PHASE_ONLY) {
        set_err("surfdisp_forward_kernels_device: the partials come from the group-velocity kernel (no PHASE_ONLY)");
        return SURFDISP_ERR_INVALID;
    }
    return forward_device_impl(stream, B, Lmax, nlay, model, P, per, kind, c, u, status,
                               workspace, workspace_bytes, nullptr, dcdb, dcda, dcdr);
}

// Parameters -> layer stacks on the device for a static-structure model; the
// descriptor layout is documented in surfdisp_layers.hip and built by pysurfinv_amd.layers_batch.
int surfdisp_params_to_model_device(void *stream, int C, int N, int L, const double *params,
                                    const int *idesc, const double *fdesc, float *model)
{
    if (C < 1 || N < 0 || L < 2 || L > SURFDISP_NLAY_MAX || !params || !idesc || !fdesc || !model) {
        set_err("surfdisp_params_to_model_device: bad argument");
        return SURFDISP_ERR_INVALID;
    }
    sd::LayersArgs a{C, N, params, idesc, fdesc, model, nullptr};
    SD_HIP(sd::launch_layers(static_cast<hipStream_t>(stream), a, L));
    return SURFDISP_SUCCESS;
}

// Same for a model with one thermal mantle layer (OceanMantleHybrid): the thermal
// kernel fills scratch ([C][64][2] doubles, caller-owned: no allocation, graph-capturable), the layer
// kernel assembles the stack.
size_t surfdisp_thermal_scratch_bytes(int C)
{
    return C < 1 ? 0 : (size_t)C * 64 * 2 * sizeof(double);
}

int surfdisp_params_to_model_thermal_device(void *stream, int C, int N, int L, const double *params,
                                            const int *idesc, const double *fdesc,
                                            void *scratch, size_t scratch_bytes, float *model)
{
    if (C < 1 || N < 0 || L < 2 || L > SURFDISP_NLAY_MAX || !params || !idesc || !fdesc || !model || !scratch) {
        set_err("surfdisp_params_to_model_thermal_device: bad argument");
        return SURFDISP_ERR_INVALID;
    }
    if (scratch_bytes < surfdisp_thermal_scratch_bytes(C)) {
        set_err("surfdisp_params_to_model_thermal_device: scratch too small (surfdisp_thermal_scratch_bytes)");
        return SURFDISP_ERR_INVALID;
    }
    sd::LayersArgs a{C, N, params, idesc, fdesc, model, static_cast<double *>(scratch)};
    SD_HIP(sd::launch_thermal(static_cast<hipStream_t>(stream), a));
    SD_HIP(sd::launch_layers(static_cast<hipStream_t>(stream), a, L));
    return SURFDISP_SUCCESS;
}

// Metropolis glue on the device (surfdisp_mcmc.hip): proposal and misfit / accept / state update of one lock step.
int surfdisp_mcmc_propose_device(void *stream, int C, int N, const double *p, const double *vmin, const double *vmax,
                                 const double *step, unsigned long long seed, unsigned long long counter, int reset, double *out, long chain0)
{
    if (C < 1 || N < 1 || !p || !vmin || !vmax || !step || !out || chain0 < 0) { set_err("surfdisp_mcmc_propose_device: bad argument"); return SURFDISP_ERR_INVALID; }
    sd::McmcProposeArgs a{C, N, p, vmin, vmax, step, seed, counter, reset ? 1 : 0, out, chain0, 1, nullptr, 0, 0};
    SD_HIP(sd::launch_mcmc_propose(static_cast<hipStream_t>(stream), a));
    return SURFDISP_SUCCESS;
}

// masked redraw of the chains the prior kernel tagged (tags[c] == tag; see surfdisp_prior_device); mode 0: bounded Gaussian step,
// 1: uniform prior draw, 2: the chain's state itself
int surfdisp_mcmc_propose_masked_device(void *stream, int C, int N, const double *p, const double *vmin, const double *vmax,
                                        const double *step, unsigned long long seed, unsigned long long counter, int attempt, int mode,
                                        const unsigned char *tags, int tag, double *out, long chain0)
{
    if (C < 1 || N < 1 || !p || !vmin || !vmax || !step || !out || !tags || chain0 < 0 || attempt < 0 || attempt > 1023 || mode < 0 || mode > 2 ||
        tag < 1 || tag > 255) {
        set_err("surfdisp_mcmc_propose_masked_device: bad argument"); return SURFDISP_ERR_INVALID;
    }
    sd::McmcProposeArgs a{C, N, p, vmin, vmax, step, seed, counter, mode, out, chain0, 1, tags, attempt, tag};
    SD_HIP(sd::launch_mcmc_propose(static_cast<hipStream_t>(stream), a));
    return SURFDISP_SUCCESS;
}

// The generic prior predicates on the device (csrc/surfdisp_layers.hip, surfdisp_prior_kernel): tags[c] = mark_tag where chain c's
// model breaks a rule; only_tag >= 0: only chains with tags[c] >= only_tag are looked at.  Static-structure models, no thermal layer.
int surfdisp_prior_device(void *stream, int C, int N, int L, const double *params, const int *idesc, const double *fdesc,
                          const int *flags, double vs_max, int only_tag, int mark_tag, unsigned char *tags)
{
    if (C < 1 || N < 0 || L < 1 || L > SURFDISP_NLAY_MAX || !params || !idesc || !fdesc || !flags || !tags || mark_tag < 1 || mark_tag > 255 ||
        only_tag >= mark_tag) {
        set_err("surfdisp_prior_device: bad argument"); return SURFDISP_ERR_INVALID;
    }
    sd::LayersArgs a{C, N, params, idesc, fdesc, nullptr, nullptr};
    SD_HIP(sd::launch_prior(static_cast<hipStream_t>(stream), a, L, flags, vs_max, only_tag, mark_tag, tags));
    return SURFDISP_SUCCESS;
}

int surfdisp_mcmc_propose_tree_device(void *stream, int C, int N, int depth, const double *p, const double *vmin, const double *vmax,
                                      const double *step, unsigned long long seed, unsigned long long counter, double *out, long chain0)
{
    if (C < 1 || N < 1 || depth < 1 || depth > sd::SD_MCMC_MAX_DEPTH || !p || !vmin || !vmax || !step || !out || chain0 < 0) {
        set_err("surfdisp_mcmc_propose_tree_device: bad argument (1 <= depth <= 4)"); return SURFDISP_ERR_INVALID;
    }
    sd::McmcProposeArgs a{C, N, p, vmin, vmax, step, seed, counter, 0, out, chain0, depth, nullptr, 0, 0};
    SD_HIP(sd::launch_mcmc_propose(static_cast<hipStream_t>(stream), a));
    return SURFDISP_SUCCESS;
}

int surfdisp_mcmc_accept_device(void *stream, int C, int N, int P, const float *c, const int *status,
                                const double *c_obs, const double *uncer, const unsigned char *mask, int obs_per_chain,
                                const double *p1, double *p0, double *chi0, double *row, long row_stride,
                                unsigned long long seed, unsigned long long counter, int first, long chain0)
{
    if (C < 1 || N < 1 || P < 1 || !c || !c_obs || !uncer || !mask || !p1 || !p0 || !chi0) {
        set_err("surfdisp_mcmc_accept_device: bad argument"); return SURFDISP_ERR_INVALID;
    }
    sd::McmcAcceptArgs a{C, N, P, c, status, c_obs, uncer, mask, obs_per_chain ? 1 : 0, p1, p0, chi0, row, row_stride, seed, counter, first ? 1 : 0, chain0, 1, 1, 0};
    SD_HIP(sd::launch_mcmc_accept(static_cast<hipStream_t>(stream), a));
    return SURFDISP_SUCCESS;
}

int surfdisp_mcmc_accept_tree_device(void *stream, int C, int N, int P, int depth, int nsteps, const float *c, const int *status,
                                     const double *c_obs, const double *uncer, const unsigned char *mask, int obs_per_chain,
                                     const double *q, double *p0, double *chi0, double *row, long row_stride, long step_stride,
                                     unsigned long long seed, unsigned long long counter, long chain0)
{
    if (C < 1 || N < 1 || P < 1 || depth < 1 || depth > sd::SD_MCMC_MAX_DEPTH || nsteps < 1 || nsteps > depth ||
        !c || !c_obs || !uncer || !mask || !q || !p0 || !chi0 || chain0 < 0) {
        set_err("surfdisp_mcmc_accept_tree_device: bad argument (1 <= nsteps <= depth <= 4)"); return SURFDISP_ERR_INVALID;
    }
    sd::McmcAcceptArgs a{C, N, P, c, status, c_obs, uncer, mask, obs_per_chain ? 1 : 0, q, p0, chi0, row, row_stride, seed, counter, 0,
                         chain0, depth, nsteps, step_stride};
    SD_HIP(sd::launch_mcmc_accept(static_cast<hipStream_t>(stream), a));
    return SURFDISP_SUCCESS;
}

// Measurement variant that does NOT synchronise: the caller owns four events per call
// (surfdisp_events_create) which are recorded on the launch stream before prep, between the
// kernels and after finish; durations are read later with surfdisp_events_elapsed_ms, after the
// caller's own synchronisation.  This is what bench.py uses inside its timed region.
int surfdisp_forward_batch_device_events(void *stream, int B, int Lmax, const int *nlay,
                                         const float *model, int P, const float *per, int kind,
                                         float *c, float *u, int *status,
                                         void *workspace, size_t workspace_bytes, void *const *events4)
{
    if (!events4) { set_err("events4 is NULL"); return SURFDISP_ERR_INVALID; }
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) ev[i] = static_cast<hipEvent_t>(events4[i]);
    return forward_device_impl(stream, B, Lmax, nlay, model, P, per, kind, c, u, status,
                               workspace, workspace_bytes, ev);
}

int surfdisp_events_create(int n, void **events)
{
    if (n < 0 || !events) { set_err("bad events array"); return SURFDISP_ERR_INVALID; }
    for (int i = 0; i < n; ++i) {
        hipEvent_t e;
        SD_HIP(hipEventCreate(&e));
        events[i] = e;
    }
    return SURFDISP_SUCCESS;
}

int surfdisp_events_destroy(int n, void **events)
{
    if (!events) return SURFDISP_ERR_INVALID;
    for (int i = 0; i < n; ++i) if (events[i]) (void)hipEventDestroy(static_cast<hipEvent_t>(events[i]));
    return SURFDISP_SUCCESS;
}

// make `stream` wait for `event` (recorded on another stream by surfdisp_forward_batch_device_events): lets a caller order
// the kernels of two solves on two streams, e.g. the Love root search behind the Rayleigh one while the Rayleigh
// group-velocity kernel fills the rest of the chip (forward.JointPlan)
int surfdisp_stream_wait_event(void *stream, void *event)
{
    if (!event) { set_err("NULL event"); return SURFDISP_ERR_INVALID; }
    SD_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return SURFDISP_SUCCESS;
}

int surfdisp_events_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) { set_err("NULL event"); return SURFDISP_ERR_INVALID; }
    SD_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return SURFDISP_SUCCESS;
}

// Measurement variant: same launches, bracketed by HIP events recorded ON THE LAUNCH STREAM;
// waits for completion and returns the three kernel durations in milliseconds
// (kernel_ms[0..2] = prep, phase, group).  Not capturable in a graph (it synchronises).
int surfdisp_forward_batch_device_timed(void *stream, int B, int Lmax, const int *nlay,
                                        const float *model, int P, const float *per, int kind,
                                        float *c, float *u, int *status,
                                        void *workspace, size_t workspace_bytes, float *kernel_ms)
{
    if (!kernel_ms) { set_err("kernel_ms is NULL"); return SURFDISP_ERR_INVALID; }
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) SD_HIP(hipEventCreate(&ev[i]));
    int rc = forward_device_impl(stream, B, Lmax, nlay, model, P, per, kind, c, u, status,
                                 workspace, workspace_bytes, ev);
    if (rc == SURFDISP_SUCCESS) {
        hipError_t e = hipEventSynchronize(ev[3]);
        if (e != hipSuccess) { set_err("hipEventSynchronize failed: %s", hipGetErrorString(e)); rc = SURFDISP_ERR_HIP; }
        for (int i = 0; i < 3 && rc == SURFDISP_SUCCESS; ++i)
            if (hipEventElapsedTime(&kernel_ms[i], ev[i], ev[i + 1]) != hipSuccess) {
                set_err("hipEventElapsedTime failed"); rc = SURFDISP_ERR_HIP;
            }
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// Small calls (the one-stack drop-in fast_surf_ above all) reuse a per-thread device arena and a pinned
// staging buffer: one host->device and one device->host copy per call instead of six, and no
// hipMalloc/hipFree.  Kept until the thread exits.
namespace {
constexpr size_t ARENA_MAX = (size_t)4 << 20;         // calls needing more go through hipMalloc
struct HostArena {
    int dev = -1;
    char *d = nullptr, *h = nullptr;
    size_t dcap = 0, hcap = 0;
    bool reserve(int device, size_t dbytes, size_t hbytes)
    {
        if (dev != device) { release(); dev = device; }
        if (dbytes > dcap) {
            if (d) (void)hipFree(d);
            d = nullptr; dcap = 0;
            if (hipMalloc(reinterpret_cast<void **>(&d), dbytes) != hipSuccess) return false;
            dcap = dbytes;
        }
        if (hbytes > hcap) {
            if (h) (void)hipHostFree(h);
            h = nullptr; hcap = 0;
            if (hipHostMalloc(reinterpret_cast<void **>(&h), hbytes, hipHostMallocDefault) != hipSuccess) return false;
            hcap = hbytes;
        }
        return true;
    }
    void release()
    {
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        d = h = nullptr; dcap = hcap = 0;
    }
    // no destructor on purpose: at thread/process exit the HIP runtime may already be gone
};
thread_local HostArena g_arena;
}  // namespace

// Large host-buffer calls: a grow-only device buffer and three streams per calling thread (no hipMalloc / hipFree per
// call), the batch cut into chunks that take turns on the streams - chunk k+1's copy-in and chunk k-1's copy-out run
// beside chunk k's kernels, and consecutive chunks' kernels overlap as independent batches in flight do.  Measured
// (scripts/time_host_api.py, C call, host buffers in and out): 65 536 x L10 3.07 -> 2.26 ms (21 -> 29 M solves/s),
// 262 144 x L10 11.4 -> 7.7 ms (23 -> 34 M); slots 2 / 3 and chunks of 16 384 / 32 768 / 65 536 ten-layer stacks swept.
struct HostPipe {
    int dev = -1;
    char *d = nullptr;
    size_t cap = 0;
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    bool ensure(int device, size_t bytes)
    {
        if (dev != device) { release(); dev = device; }
        for (int i = 0; i < 3; ++i)
            if (!s[i] && hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) return false;
        if (bytes > cap) {
            if (d) (void)hipFree(d);
            d = nullptr; cap = 0;
            if (hipMalloc(reinterpret_cast<void **>(&d), bytes) != hipSuccess) return false;
            cap = bytes;
        }
        return true;
    }
    void release()
    {
        if (d) (void)hipFree(d);
        for (int i = 0; i < 3; ++i) { if (s[i]) (void)hipStreamDestroy(s[i]); s[i] = nullptr; }
        d = nullptr; cap = 0;
    }
    // no destructor on purpose (see HostArena)
};
thread_local HostPipe g_pipe;
constexpr size_t PIPE_KEEP_MAX = (size_t)3 << 30;      // a larger buffer is given back after the call

// chunks of a large host-buffer call: about 327 680 layers' worth of stacks each (32 768 ten-layer stacks), at least two;
// stacks of more than 20 layers go through in one piece (16 384 x L64: 3.56 ms either way, 25 600 x L96: 8.65 against 9.35 ms)
static int host_chunks(int B, int Lmax)
{
    long bc = (long)knobs().host_chunk_layers / Lmax;
    if (bc < 16384) return 1;          // deep stacks: the copies are a tenth of the call and smaller launches cost more than they hide
    if (bc > 65536) bc = 65536;
    if (B < bc) return 1;
    const long n = (B + bc - 1) / bc;
    return (int)(n < 2 ? 2 : n);
}

static int forward_batch_pipelined(int device, int nch, int B, int Lmax, const int *nlay, const float *model,
                                   int P, const float *per, int kind, float *c, float *u, int *status)
{
    const int Bc = (int)((((long)B + nch - 1) / nch + 63) / 64 * 64);          // stacks per chunk (the last one may be shorter)
    const size_t nm = (size_t)Bc * 5 * Lmax * sizeof(float), np = (size_t)P * sizeof(float);
    const size_t ni = (size_t)Bc * sizeof(int), no = (size_t)Bc * P * sizeof(float);
    const size_t ws = surfdisp_workspace_bytes(Bc, Lmax, P);
    const size_t o_model = 0, o_per = o_model + align_up(nm), o_nl = o_per + align_up(np), o_c = o_nl + align_up(ni);
    const size_t o_u = o_c + align_up(no), o_st = o_u + align_up(no), o_ws = o_st + align_up(ni), slot = align_up(o_ws + ws);
    const int NS = knobs().host_slots >= 3 ? 3 : 2;
    if (!g_pipe.ensure(device, NS * slot)) { set_err("host-buffer pipeline: allocation failed"); return SURFDISP_ERR_HIP; }
    int ret = SURFDISP_SUCCESS;
    auto copy_out = [&](int k) -> bool {                   // chunk k's results to the caller's arrays (stream-ordered behind its kernels)
        char *d = g_pipe.d + (size_t)(k % NS) * slot;
        hipStream_t s = g_pipe.s[k % NS];
        const long b0 = (long)k * Bc;
        const size_t bc = (size_t)((B - b0) < Bc ? (B - b0) : Bc);
        return hipMemcpyAsync(c + b0 * P, d + o_c, bc * P * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess &&
               (!u || hipMemcpyAsync(u + b0 * P, d + o_u, bc * P * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess) &&
               (!status || hipMemcpyAsync(status + b0, d + o_st, bc * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess);
    };
    int k = 0;
    for (; k < nch && ret == SURFDISP_SUCCESS; ++k) {
        char *d = g_pipe.d + (size_t)(k % NS) * slot;
        hipStream_t s = g_pipe.s[k % NS];
        const long b0 = (long)k * Bc;
        const int bc = (int)((B - b0) < Bc ? (B - b0) : Bc);
        if (bc <= 0) break;
        // the slot's previous chunk leaves first (a copy to pageable memory holds the calling thread until that chunk is
        // done - the other slot's chunk is computing meanwhile)
        if (k >= NS && !copy_out(k - NS)) { set_err("device->host copy failed"); ret = SURFDISP_ERR_HIP; break; }
        const bool ok = hipMemcpyAsync(d + o_model, model + (size_t)b0 * 5 * Lmax, (size_t)bc * 5 * Lmax * sizeof(float), hipMemcpyHostToDevice, s) == hipSuccess &&
                        (k >= NS || hipMemcpyAsync(d + o_per, per, np, hipMemcpyHostToDevice, s) == hipSuccess) &&
                        (!nlay || hipMemcpyAsync(d + o_nl, nlay + b0, (size_t)bc * sizeof(int), hipMemcpyHostToDevice, s) == hipSuccess);
        if (!ok) { set_err("host->device copy failed"); ret = SURFDISP_ERR_HIP; break; }
        ret = surfdisp_forward_batch_device(s, bc, Lmax, nlay ? reinterpret_cast<int *>(d + o_nl) : nullptr,
                                            reinterpret_cast<float *>(d + o_model), P, reinterpret_cast<float *>(d + o_per),
                                            kind | SURFDISP_PIPELINED, reinterpret_cast<float *>(d + o_c),
                                            u ? reinterpret_cast<float *>(d + o_u) : nullptr,
                                            reinterpret_cast<int *>(d + o_st), d + o_ws, ws);
    }
    if (ret == SURFDISP_SUCCESS) {
        for (int j = (k >= NS ? k - NS : 0); j < k; ++j)
            if (!copy_out(j)) { set_err("device->host copy failed"); ret = SURFDISP_ERR_HIP; break; }
    }
    for (int i = 0; i < NS; ++i) {
        hipError_t e = hipStreamSynchronize(g_pipe.s[i]);
        if (e != hipSuccess && ret == SURFDISP_SUCCESS) { set_err("kernel execution failed: %s", hipGetErrorString(e)); ret = SURFDISP_ERR_HIP; }
    }
    if (g_pipe.cap > PIPE_KEEP_MAX) g_pipe.release();
    return ret;
}

int surfdisp_forward_batch(int device, int B, int Lmax, const int *nlay, const float *model,
                           int P, const float *per, int kind,
                           float *c, float *u, int *status)
{
    int rc = check_args(B, Lmax, P, kind, model, per, c, u);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_err("no HIP device: libsurfdisp_hip has no CPU fallback");
        return SURFDISP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_err("bad device ordinal"); return SURFDISP_ERR_INVALID; }
    // run on `device`, then hand the calling thread back the device it had selected
    struct DeviceScope {
        int prev = -1;
        ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    } scope;
    if (hipGetDevice(&scope.prev) != hipSuccess) scope.prev = -1;
    if (scope.prev == device) scope.prev = -1;                 // nothing to restore
    else SD_HIP(hipSetDevice(device));
    const size_t nm = (size_t)B * 5 * Lmax * sizeof(float);
    const size_t np = (size_t)P * sizeof(float);
    const size_t ni = (size_t)B * sizeof(int);
    const size_t no = (size_t)B * P * sizeof(float);
    const size_t ws = surfdisp_workspace_bytes(B, Lmax, P);
    // inputs [model | per | nlay] and outputs [c | u | status] are contiguous, then the workspace
    const size_t o_model = 0;
    const size_t o_per = o_model + align_up(nm);
    const size_t o_nl = o_per + align_up(np);
    const size_t o_c = o_nl + align_up(ni);
    const size_t o_u = o_c + align_up(no);
    const size_t o_st = o_u + align_up(no);
    const size_t o_ws = o_st + align_up(ni);
    const bool small = (o_ws + ws) <= ARENA_MAX;
    if (!small) {
        const int nch = knobs().host_pipeline ? host_chunks(B, Lmax) : 1;
        if (nch > 1) return forward_batch_pipelined(device, nch, B, Lmax, nlay, model, P, per, kind, c, u, status);
    }
    char *d = nullptr, *h = nullptr;
    hipStream_t s = nullptr;
    if (small) {
        if (!g_arena.reserve(device, o_ws + ws, o_ws)) { set_err("arena allocation failed"); return SURFDISP_ERR_HIP; }
        d = g_arena.d; h = g_arena.h;
    } else {
        if (!g_pipe.ensure(device, o_ws + ws)) { set_err("host-buffer call: allocation failed"); return SURFDISP_ERR_HIP; }
        d = g_pipe.d; s = g_pipe.s[0];
    }
    int ret = SURFDISP_SUCCESS;
    do {
        bool ok;
        if (small) {
            memcpy(h + o_model, model, nm);
            memcpy(h + o_per, per, np);
            if (nlay) memcpy(h + o_nl, nlay, ni);
            ok = hipMemcpyAsync(d, h, o_c, hipMemcpyHostToDevice, s) == hipSuccess;
        } else {
            ok = hipMemcpyAsync(d + o_model, model, nm, hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(d + o_per, per, np, hipMemcpyHostToDevice, s) == hipSuccess &&
                 (!nlay || hipMemcpyAsync(d + o_nl, nlay, ni, hipMemcpyHostToDevice, s) == hipSuccess);
        }
        if (!ok) { set_err("host->device copy failed"); ret = SURFDISP_ERR_HIP; break; }
        ret = surfdisp_forward_batch_device(s, B, Lmax, nlay ? reinterpret_cast<int *>(d + o_nl) : nullptr,
                                            reinterpret_cast<float *>(d + o_model), P,
                                            reinterpret_cast<float *>(d + o_per), kind,
                                            reinterpret_cast<float *>(d + o_c),
                                            reinterpret_cast<float *>(d + o_u),
                                            reinterpret_cast<int *>(d + o_st), d + o_ws, ws);
        if (ret) break;
        if (small) {
            ok = hipMemcpyAsync(h + o_c, d + o_c, o_ws - o_c, hipMemcpyDeviceToHost, s) == hipSuccess;
        } else {
            ok = hipMemcpyAsync(c, d + o_c, no, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 (!u || hipMemcpyAsync(u, d + o_u, no, hipMemcpyDeviceToHost, s) == hipSuccess) &&
                 (!status || hipMemcpyAsync(status, d + o_st, ni, hipMemcpyDeviceToHost, s) == hipSuccess);
        }
        if (!ok) { set_err("device->host copy failed"); ret = SURFDISP_ERR_HIP; break; }
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_err("kernel execution failed: %s", hipGetErrorString(e)); ret = SURFDISP_ERR_HIP; break; }
        if (small) {
            memcpy(c, h + o_c, no);
            if (u) memcpy(u, h + o_u, no);
            if (status) memcpy(status, h + o_st, ni);
        }
    } while (0);
    // an error exit may leave copies / kernels queued on the thread's cached stream that still write into the caller's pageable
    // arrays or into the buffer the next call reuses: drain it before returning (the result is the error already recorded)
    if (ret != SURFDISP_SUCCESS) (void)hipStreamSynchronize(s);
    if (!small && g_pipe.cap > PIPE_KEEP_MAX) g_pipe.release();
    return ret;
}

// frees what the host-buffer entries cached for the calling thread (device arena, pinned staging, pipeline buffer, streams)
void surfdisp_thread_release(void)
{
    g_arena.release();
    g_arena.dev = -1;
    g_pipe.release();
    g_pipe.dev = -1;
}

// Fortran-ABI drop-in for the reference object's symbol (fast_surf.f:2-5).
void fast_surf_(const int *n_layer, const int *kind,
                const float *vp, const float *vs, const float *rho,
                const float *h, const float *qsinv,
                const float *per, const int *nper,
                float *uR, float *uL, float *cR, float *cL)
{
    if (!n_layer || !kind || !nper || !vp || !vs || !rho || !h || !qsinv || !per) return;
    const int n = *n_layer, P = *nper > SURFDISP_NPER_MAX ? SURFDISP_NPER_MAX : *nper;  // init.f:63-66
    if (n < 2 || n > SURFDISP_NLAY_MAX || P < 1) { set_err("fast_surf_: bad n_layer/nper"); return; }
    float *model = static_cast<float *>(malloc((size_t)5 * n * sizeof(float)));
    float c[SURFDISP_NPER_MAX], u[SURFDISP_NPER_MAX];
    if (!model) return;
    memcpy(model + 0 * n, vp, n * sizeof(float));
    memcpy(model + 1 * n, vs, n * sizeof(float));
    memcpy(model + 2 * n, rho, n * sizeof(float));
    memcpy(model + 3 * n, h, n * sizeof(float));
    memcpy(model + 4 * n, qsinv, n * sizeof(float));
    const int dev = knobs().device;
    const int rc = surfdisp_forward_batch(dev, 1, n, nullptr, model, P, per, *kind, c, u, nullptr);
    free(model);
    if (rc != SURFDISP_SUCCESS) {
        // no silent fallback: report loudly and leave the outputs untouched (= all zeros = failure
        // in the reference's convention, models.py:29-33)
        fprintf(stderr, "surfdisp fast_surf_: %s\n", g_err);
        return;
    }
    for (int i = 0; i < P; ++i) {                       // fast_surf.f:197-208
        if (c[i] == 0.0f) break;
        if (*kind == SURFDISP_KIND_LOVE) { if (cL) cL[i] = c[i]; if (uL) uL[i] = u[i]; }
        else                             { if (cR) cR[i] = c[i]; if (uR) uR[i] = u[i]; }
    }
}

}  // extern "C"
