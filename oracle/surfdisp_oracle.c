/*
 * surfdisp_oracle.c  --  CPU ORACLE.  TEST INFRASTRUCTURE, NOT A PRODUCT PATH.
 *
 * A reentrant plain-C restatement of the reference's fast_surf() algorithm
 * (001cat/pySurfInv, fast_surf_src, all .f files).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this file; the shipped
 * library (pysurfinv_amd/csrc) never does.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (i)  the reference's own known-answer data senskernel-1.0/TEST1 (committed
 *        as tests/golden/test1_eus.npz), and
 *   (ii) outputs of the reference Fortran itself, compiled unmodified by
 *        oracle/build_ref.sh and captured in tests/golden (npz files) by
 *        tests/golden/make_golden.py.
 *
 * Semantics chosen where the reference is history dependent:
 *   - "fresh process per call": ndiv starts at 5 for every solve (defect 1),
 *     COMMON /dispe/ starts zeroed (stale-output defect), so a NEVILL
 *     non-convergence yields all-zero outputs.
 *   - everything else, including the cross-period carry-over of mmax (defect 2)
 *     and the sequential start rule c1 = 0.9*c(k-1) (defect 9), is reproduced.
 *
 * Each function cites the reference file:line it follows.  Arithmetic is fp32
 * where the reference is REAL*4 and fp64 where it is DOUBLE PRECISION; build with
 * -ffp-contract=off so that the operation order below is what executes.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "surfdisp_oracle.h"

#define NSZ 1000 /* fast_surf.f:9  nsize */
#define NKN 208  /* >= max Rayleigh sublayers: max(100, SURFDISP_NLAY_MAX)+1 */

typedef struct {
    /* COMMON /ref/ (fast_surf.f:63) */
    int n;                 /* n_layer = nmax */
    int kind;              /* 1 Love, 2 Rayleigh */
    float a_ref[NSZ], b_ref[NSZ], rho_ref[NSZ], d_ref[NSZ], qs_ref[NSZ];
    /* COMMON /d/ working stack (fast_surf.f:48) */
    float a[NSZ], b[NSZ], rho[NSZ], d[NSZ], qs[NSZ];
    /* COMMON /c/ */
    int nmax, mmax, idrop, ndiv;
    float fact;
    /* REIGEN knot storage yy1..4, yz1..4 (surfa.f:737-738), fp64; [sublayer][knot] */
    double yy1[NKN][5], yy2[NKN][5], yy3[NKN][5], yy4[NKN][5];
    double yz1[NKN][5], yz2[NKN][5], yz3[NKN][5], yz4[NKN][5];
    /* COMMON /rar1/ dcda, dcdb, dcdr, dwx (surfa.f:722,729 / 390,396: DOUBLE PRECISION) and COMMON /rar/ mmax as the
     * last REIGEN / LEIGEN call left them: the analytic partials per SUBLAYER of the flattened, attenuated stack */
    double dcda[NSZ], dcdb[NSZ], dcdr[NSZ], dwx[NSZ];
    int rar_mmax;
    /* counters for the work model (not in the reference) */
    long n_delta;
    /* developer trace of the scan of one period (surfdisp_oracle_scan_trace; not in the reference) */
    int tr_k, tr_cap, tr_n, tr_extra;
    float *tr_c, *tr_d; int *tr_mm;
    /* developer aid (surfdisp_oracle_bracket_profile): nsub > 0 - once period tr_k's bracket is found, the secular function
     * at nsub + 1 equidistant points across it with the layer dropping frozen (idrop = 1), as NEVILL sees it */
    int tr_nsub;
    /* test hook (surfdisp_oracle_forward_at; not in the reference): phase velocities at which the ellipticity
     * and the group velocity of each period are evaluated instead of the oracle's own roots */
    const float *c_at;
} ctx_t;

static inline float sgn1(float x) { return copysignf(1.0f, x); } /* SIGN(1.,x) */

/* NOT the reference: "equally valid" fp32 evaluations of the reference's own formulas, used only by scripts/soak.py to tell
 * whether a root or a sign is DEFINED by the formulas or by the rounding of one particular libm (a result that moves by more
 * than the parity bar under such a variant cannot be reproduced by any other arithmetic than the reference's own, bit for bit):
 *   bit 0: e^x of the secular functions as exp2f(x log2 e) instead of expf (the hyperbolic functions of thin evanescent layers
 *          are differences of two such values: (e^x - e^-x)/2 for |x| << 1 is cancellation noise of the exponential's last bit)
 *   bit 1: the earth-flattening factors from double-precision log / pow rounded once (flat1.f:44-56 differences nearly equal powf
 *          values: one ulp of powf moves a sub-kilometre layer's density by 4e-4) */
static int g_variant = 0;
void surfdisp_oracle_set_variant(int v) { g_variant = v; }
static inline float exp_v(float x) { return (g_variant & 1) ? exp2f(x * 1.44269504f) : expf(x); }
static inline float log_v(float x) { return (g_variant & 2) ? (float)log((double)x) : logf(x); }
static inline float pow_v(float x, float y) { return (g_variant & 2) ? (float)pow((double)x, (double)y) : powf(x, y); }

/* ---------------------------------------------------------------- flat1.f:2-73
 * Earth-flattening (Biswas 1972), in place on the first n layers. */
static void flat1(float *h, float *ro, float *vp, float *vs, int n, int kind)
{
    float hh[NSZ];
    const float a = 6371.0f;
    for (int i = 0; i < n; ++i) hh[i] = h[i];
    float pwr = 2.2750f;
    if (kind == 1) pwr = 5.0f;
    int nm = n - 1;
    float hs = 0.0f;
    for (int i = 0; i < n; ++i) {           /* flat1.f:33-37 radii */
        float ht = hs;
        hs = hs + hh[i];
        hh[i] = a - ht;
    }
    for (int i = 0; i < nm; ++i) {          /* flat1.f:41-56 */
        int ii = i + 1;
        float fltd = log_v(hh[i] / hh[ii]);
        float dif = (1.0f / hh[ii] - 1.0f / hh[i]) * a / fltd;
        float difr = pow_v(hh[i], pwr) - pow_v(hh[ii], pwr);
        float qqq = difr / (fltd * pow_v(a, pwr) * pwr);
        ro[i] = ro[i] * qqq;
        vp[i] = vp[i] * dif;
        vs[i] = vs[i] * dif;
    }
    float fact = a / hh[n - 1];             /* flat1.f:58-62 half space */
    vp[n - 1] = vp[n - 1] * fact;
    vs[n - 1] = vs[n - 1] * fact;
    ro[n - 1] = ro[n - 1] * pow_v(1.0f / fact, pwr);
    float z0 = 0.0f;
    for (int i = 1; i < n; ++i) {           /* flat1.f:65-68 */
        float z1 = a * log_v(a / hh[i]);
        h[i - 1] = z1 - z0;
        z0 = z1;
    }
    h[n - 1] = 0.0f;
}

/* attenuation-dispersed model for period t1 over the first m layers
 * (calcul.f:112-131; phase 2 copies :239-249 and :325-335 are identical) */
static void build_model(ctx_t *s, float t1, int m)
{
    const float pi = 3.1415927f, t_base = 1.0f;
    for (int i = 0; i < m; ++i) {
        s->rho[i] = s->rho_ref[i];
        s->d[i] = s->d_ref[i];
        float qsq = s->qs_ref[i] * logf(t_base / t1) / pi;
        float qpq = qsq * 1.33333333f * (s->b_ref[i] * s->b_ref[i]) / (s->a_ref[i] * s->a_ref[i]);
        s->b[i] = s->b_ref[i] * (1.0f + qsq);
        s->a[i] = s->a_ref[i] * (1.0f + qpq);
    }
}

/* ------------------------------------------------------------ surfa.f:135-183
 * Love secular function, Thomson-Haskell from the half space up. */
static float dltar1(const ctx_t *s, float c, float t)
{
    const float *b = s->b, *rho = s->rho, *d = s->d;
    int mmax = s->mmax;
    float wvno = 6.2831853f / (c * t);
    float covb = c / b[mmax - 1];
    float h = rho[mmax - 1] * b[mmax - 1] * b[mmax - 1];
    float rb = sqrtf(fabsf(covb * covb - 1.0f));
    float ut = 1.0f, tt = h * rb, ett = tt;
    for (int k = 1; k <= mmax - 1; ++k) {
        int m = mmax - k - 1; /* 0-based */
        if (b[m] == 0.0f) continue;
        covb = c / b[m];
        rb = sqrtf(fabsf(covb * covb - 1.0f));
        h = rho[m] * b[m] * b[m];
        float q = -wvno * d[m] * rb;
        float y, z, cosq;
        if (rb < 0.1e-20f || c == b[m]) {          /* label 1221 */
            y = -wvno * d[m]; z = 0.0f; cosq = 1.0f;
        } else if (c < b[m]) {                     /* label 1209 */
            float exqp = exp_v(q), exqm = 1.0f / exqp;
            y = (exqp - exqm) / (2.0f * rb);
            z = -rb * rb * y;
            cosq = (exqp + exqm) / 2.0f;
        } else {                                   /* label 1231 */
            float sinq = sinf(q);
            y = sinq / rb; z = rb * sinq; cosq = cosf(q);
        }
        float eut = cosq * ut - y * tt / h;
        ett = h * z * ut + cosq * tt;
        ut = eut; tt = ett;
    }
    return -ett;
}

/* ------------------------------------------------------------ surfa.f:185-372
 * Rayleigh secular function: Dunkin 5-component compound-matrix recursion.
 * mup=1 dispersion (-bb1); mup=2 ellipticity (two passes, 0.5*bb1/r12). */
static float dltar4(const ctx_t *s, float c, float t, int mup)
{
    const float *p = s->a, *sv = s->b, *rho = s->rho, *d = s->d;
    const int mmax = s->mmax;
    const float accur = 1.e-8f, accurs = 1.e-8f;
    float wvno = 6.28318531f / (c * t);
    float csq = c * c;
    int jump = (mup > 1) ? 2 : 1;
    float r12 = 0.0f, bb1 = 0.0f;
    for (;;) {
        float b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0;
        if (jump == 1) b1 = 1.0f; else if (jump == 2) b2 = 1.0f; else b3 = 1.0f;
        float ra = 0, rb = 0, g = 0, g1 = 0;
        int m;
        for (m = 0; m < mmax; ++m) {
            float arga = 1.0f - csq / (p[m] * p[m]);
            ra = sqrtf(fabsf(arga));
            if (arga > 0.0f) ra = -ra;
            float a11, a12, a13, a14, a15, a21, a22, a23, a24, a31, a32, a33, a41, a42, a51;
            if (!(fabsf(sv[m]) > accurs)) {
                /* liquid surface layer, surfa.f:216-251 */
                float pm = wvno * ra * d[m];
                if (mup > 1) continue;
                float rhoc = rho[m] * csq, sinpr, cosp;
                if (fabsf(ra) < accur || ra == 0.0f) {
                    sinpr = wvno * d[m]; cosp = 1.0f;
                } else if (ra < 0.0f) {
                    sinpr = (exp_v(pm) - exp_v(-pm)) / (2.0f * ra);
                    cosp = 0.5f * (exp_v(pm) + exp_v(-pm));
                } else {
                    sinpr = sinf(pm) / ra;
                    cosp = cosf(pm);
                }
                a11 = cosp; a21 = rhoc * sinpr;
                a31 = a41 = a51 = a12 = a22 = a32 = a42 = a13 = a23 = a33 = a14 = a24 = a15 = 0.0f;
            } else {
                float argb = 1.0f - csq / (sv[m] * sv[m]);
                rb = sqrtf(fabsf(argb));
                if (argb > 0.0f) rb = -rb;
                g = 2.0f * (sv[m] * sv[m]) / csq;
                g1 = g - 1.0f;
                if (m == mmax - 1) break;          /* half space: label 52 */
                float rhoc = rho[m] * csq;
                float pm = wvno * ra * d[m];
                float qm = wvno * rb * d[m];
                float rsinp, sinpr, cosp, rsinq, sinqr, cosq;
                if (ra < 0.0f) {
                    rsinp = -ra * 0.5f * (exp_v(pm) - exp_v(-pm));
                    sinpr = -rsinp / (ra * ra);
                    cosp = 0.5f * (exp_v(pm) + exp_v(-pm));
                } else if (ra == 0.0f) {
                    rsinp = 0.0f; sinpr = wvno * d[m]; cosp = 1.0f;
                } else {
                    rsinp = ra * sinf(pm);
                    sinpr = rsinp / (ra * ra);
                    cosp = cosf(pm);
                }
                if (fabsf(rb) < accur) {
                    rsinq = 0.0f; sinqr = wvno * d[m]; cosq = 1.0f;
                } else if (rb > 0.0f) {
                    rsinq = rb * sinf(qm);
                    sinqr = rsinq / (rb * rb);
                    cosq = cosf(qm);
                } else {
                    rsinq = -rb * 0.5f * (exp_v(qm) - exp_v(-qm));
                    sinqr = -rsinq / (rb * rb);
                    cosq = 0.5f * (exp_v(qm) + exp_v(-qm));
                }
                float rr = rsinp * rsinq, ss = sinpr * sinqr, cc = cosp * cosq;
                float rs1 = rsinp * cosq, rs2 = sinqr * cosp, rs3 = sinpr * cosq, rs4 = rsinq * cosp;
                float gm = 2.0f * g - 1.0f;
                float gs = g * g, g1s = g1 * g1;
                float ccm = 1.0f - cc;
                float gg1 = g * g1;
                float rhocs = rhoc * rhoc;
                float suu = gs * rr + g1s * ss;
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
            /* surfa.f:326-335 */
            float n1 = a11 * b1 + a12 * b2 + a13 * b3 + a14 * b4 + a15 * b5;
            float n2 = a21 * b1 + a22 * b2 + a23 * b3 + a24 * b4 - a14 * b5;
            float n3 = a31 * b1 + a32 * b2 + a33 * b3 - 0.5f * a23 * b4 + 0.5f * a13 * b5;
            float n4 = a41 * b1 + a42 * b2 - 2.0f * a32 * b3 + a22 * b4 - a12 * b5;
            float n5 = a51 * b1 - a41 * b2 + 2.0f * a31 * b3 - a21 * b4 + a11 * b5;
            b1 = n1; b2 = n2; b3 = n3; b4 = n4; b5 = n5;
            bb1 = n1;
        }
        if (m >= mmax) m = mmax - 1; /* DO ran to completion: Fortran leaves m=mmax+1; only a
                                        liquid bottom layer gets here - outside reference use */
        {   /* half-space closure, surfa.f:340-354 */
            float pp = p[m];
            float sss = sv[m] * sv[m];
            float ppp = pp * pp;
            float rhp = rho[m] * pp;
            float gra = g * ra;
            float g1s = g1 * g1;
            float rba = rb - 1.0f / ra;
            float a11 = -2.0f * rb * sss / ppp + csq * g1s / ppp / gra;
            float a12 = rhp * pp;
            float a13 = -rb / a12 + g1 / a12 / gra;
            float a14 = rb / a12 / gra;
            float a15 = rba / rhp / rhp / csq / g;
            a12 = -1.0f / g / a12;
            bb1 = a11 * b1 + a12 * b2 + 2.0f * a13 * b3 + a14 * b4 + a15 * b5;
        }
        if (mup == 1) return -bb1;
        if (mup == 2) {
            if (jump == 2) r12 = bb1;
            jump = jump + 1;
            if (jump == 3) continue;
            return 0.5f * bb1 / r12;
        }
        return fabsf(bb1);
    }
}

/* ------------------------------------------------------------- surfa.f:85-133
 * layer dropping (when idrop==0) + dispatch. kk: 1 Love, 2 Rayleigh, 3 ellipticity */
static float dltar(ctx_t *s, float cc, float tt, int kk)
{
    if (s->idrop <= 0) {
        float dmax = s->fact * cc * tt;
        s->mmax = s->nmax;
        float sum = 0.0f;
        for (int ii = 0; ii < s->nmax; ++ii) {
            if (cc < s->b[ii]) {
                sum = sum + s->d[ii];
                if (sum > dmax) { s->mmax = ii + 1; break; }
            }
        }
        s->idrop = 1;
        if (s->mmax < 2) s->mmax = 2;
    }
    s->n_delta++;
    if (kk == 1) return dltar1(s, cc, tt);
    if (kk == 2) return dltar4(s, cc, tt, 1);
    return dltar4(s, cc, tt, 2);
}

/* --------------------------------------------------------------- surfa.f:2-83
 * bracket refinement: bisection mixed with Neville interpolation.
 * returns 0 and *cc on success, 1 on "too many cycles" (lstop). */
static int nevill(ctx_t *s, float t, float c1, float c2, float del1, float del2,
                  int ifunc, float *cc)
{
    const float accur1 = 0.1e-5f, accur2 = 0.1e-7f;
    float x[21], y[21];
    int ic = 0, nev, m = 1;
    float c3 = (c1 + c2) / 2.0f;
    float del3 = dltar(s, c3, t, ifunc);
    nev = 1;
    for (;;) {
        ic = ic + 1;
        if (!(ic < 50)) return 1;
        int bisect = 0;
        /* surfa.f:32-34 : c3 must lie strictly inside (c1,c2).  These are arithmetic IFs: a NaN expression is
         * neither negative nor zero and takes the THIRD label (flang, like gfortran, lowers `if (x) l1,l2,l3` to
         * x<0 -> l1, x==0 -> l2, else l3).  A NaN c3 (Neville step through a NaN end value) therefore goes
         * 777 -> 1330 -> 1344: the reference falls back to BISECTION and keeps going - pinned bit for bit by
         * tests/golden/ref_families.npz (overflow_R, wild_*), where the reference returns roots next to the
         * edge of the overflowed region instead of exhausting its 50 cycles. */
        if (c1 - c3 <= 0.0f) { if (c2 - c3 <= 0.0f) bisect = 1; }        /* 1320: if(c2-c3) 1344,1344,1000 */
        else                 { if (!(c2 - c3 < 0.0f)) bisect = 1; }      /* 1330: if(c2-c3) 1000,1344,1344 */
        if (!bisect) {
            float s13 = del1 - del3;
            float s32 = del3 - del2;
            if (sgn1(del3) * sgn1(del1) <= 0.0f) { c2 = c3; del2 = del3; }
            else { c1 = c3; del1 = del3; }
            if (fabsf(c1 - c2) <= accur1) { *cc = c3; return 0; }
            if (sgn1(s13) != sgn1(s32)) nev = 0;
            float ss1 = fabsf(del1), s1 = 0.1f * ss1;
            float ss2 = fabsf(del2), s2 = 0.1f * ss2;
            if (s1 > ss2 || s2 > ss1) bisect = 1;
            else if (nev == 0) bisect = 1;
            else {
                if (nev == 2) { x[m + 1] = c3; y[m + 1] = del3; }      /* label 1350 */
                else { x[1] = c1; y[1] = del1; x[2] = c2; y[2] = del2; m = 1; }
                for (int kk = 1; kk <= m; ++kk) {                        /* label 1355 */
                    int j = m - kk + 1;
                    if (fabsf(y[m + 1] - y[j]) <= accur2) { bisect = 1; break; }
                    x[j] = (-y[j] * x[j + 1] + y[m + 1] * x[j]) / (y[m + 1] - y[j]);
                }
                if (!bisect) {
                    c3 = x[1];
                    del3 = dltar(s, c3, t, ifunc);
                    nev = 2;
                    m = m + 1;
                    if (m > 10) m = 10;
                    continue;
                }
            }
        }
        /* label 1344 */
        c3 = (c1 + c2) / 2.0f;
        del3 = dltar(s, c3, t, ifunc);
        nev = 1;
        m = 1;
    }
}

/* ----------------------------------------------------------- surfa.f:374-712
 * Love group velocity (only what feeds ugr). t, c of this period. */
static float leigen(ctx_t *s, float t, float c)
{
    float *b = s->b, *rho = s->rho, *d = s->d, *a = s->a, *qs = s->qs;
    static const float const_lim = 1.E+10f, const_lim1 = 1.E+5f;
    int mmax = s->n, nmax = mmax;
    int mm1 = mmax - 1;
    int ivre = (NSZ - 1) / mm1;                          /* surfa.f:414 */
    if (s->ndiv > ivre) s->ndiv = ivre;
    int ndiv = s->ndiv;
    float div = (float)ndiv;
    if (ndiv > 1) {                                      /* surfa.f:418-445 */
        float pr_d[NSZ], pr_b[NSZ], pr_r[NSZ], pr_q[NSZ];
        int jj = 1;
        if (b[0] <= 0.1e-10f) jj = 2;
        for (int j = jj; j <= mm1; ++j) {
            int ldiv = (j - jj) * ndiv;
            for (int i = 1; i <= ndiv; ++i) {
                pr_d[ldiv + i - 1] = d[j - 1] / div;
                pr_b[ldiv + i - 1] = b[j - 1];
                pr_r[ldiv + i - 1] = rho[j - 1];
                pr_q[ldiv + i - 1] = qs[j - 1];
            }
        }
        mmax = (mm1 - jj + 1) * ndiv + jj;
        d[mmax - 1] = 0.0f;
        a[mmax - 1] = a[nmax - 1];
        b[mmax - 1] = b[nmax - 1];
        rho[mmax - 1] = rho[nmax - 1];
        qs[mmax - 1] = qs[nmax - 1];
        nmax = mmax;
        mm1 = mmax - 1;
        for (int j = jj; j <= mm1; ++j) {
            d[j - 1] = pr_d[j - jj];
            b[j - 1] = pr_b[j - jj];
            rho[j - 1] = pr_r[j - jj];
            qs[j - 1] = pr_q[j - jj];
        }
    }
    /* layer dropping, surfa.f:475-487 */
    mmax = nmax;
    float dmax = s->fact * c * t;
    {
        float sum = 0.0f;
        int max = 0, ii, cut = 0;
        for (ii = 1; ii <= mmax; ++ii) {
            max = ii;
            if (c < b[ii - 1]) {
                sum = sum + d[ii - 1];
                if (ii == mmax) { cut = 1; break; }
                if (sum <= dmax) continue;
                if (b[ii] < b[ii - 1]) { cut = 1; break; }
                if (b[ii] == b[ii - 1]) continue;
                max = max + 1; cut = 1; break;           /* label 90009 */
            }
        }
        if (!cut) max = max + 1;                         /* fall through into 90009 */
        mmax = max;
    }
    float wvno = 6.2831853f / (c * t);
    float wvnosq = wvno * wvno;
    float tmp = 6.2831853f / t;
    float omegsq = tmp * tmp;
    const float xxmin = 1.0e-20f;                        /* surfa.f:404 */
    float amp[NSZ];                                      /* displacement at the middle knot of each sublayer, surfa.f:553 */
    for (int j = 0; j < NSZ; ++j) { amp[j] = 0.0f; s->dcdr[j] = 0.0; s->dcdb[j] = 0.0; }   /* surfa.f:461-466 */
    float ut0 = 1.0f;
    float ut, tq, sumi0, sumi1;
restart:
    ut = ut0;
    {
        float covb = c / b[mmax - 1];
        float h = rho[mmax - 1] * b[mmax - 1] * b[mmax - 1];
        float rb = wvno * sqrtf(fabsf(covb * covb - 1.0f));
        tq = -h * rb * ut0;
        amp[mmax - 1] = ut;                              /* surfa.f:497 */
        float dm, sm;
        if (rb == 0.0f) { dm = 1.0e25f; sm = 0.0f; }
        else { dm = 0.5f / rb; sm = 0.5f * rb; }
        float dldr = omegsq * dm;                        /* surfa.f:509-512 (REAL*4 expressions stored to REAL*8) */
        float dldm = -(wvnosq * dm + sm);
        s->dcdb[mmax - 1] = 2.0f * rho[mmax - 1] * b[mmax - 1] * c * dldm / wvno;
        s->dcdr[mmax - 1] = (c / wvno) * (dldr + b[mmax - 1] * b[mmax - 1] * dldm);
        sumi0 = rho[mmax - 1] * dm;
        sumi1 = h * dm;
    }
    for (int k = 1; k <= mmax - 1; ++k) {
        if (fabsf(ut) > const_lim) { ut0 = ut0 / const_lim1; goto restart; }
        int m = mmax - k; /* 1-based */
        if (b[m - 1] == 0.0f) continue;
        float covb = c / b[m - 1];
        float rb = wvno * sqrtf(fabsf(covb * covb - 1.0f));
        float h = rho[m - 1] * b[m - 1] * b[m - 1];
        float dz = d[m - 1] / 4.0f;
        float dmm[5], smm[5];
        dmm[0] = ut * ut;
        smm[0] = (tq / h) * (tq / h);                    /* surfa.f:530: (tq/h)**2 */
        float eut = ut, ett = tq;
        for (int kk = 2; kk <= 5; ++kk) {
            float xkk = (float)(kk - 1);
            float q = rb * dz * xkk;
            float y, z, cosq;
            if (c < b[m - 1]) {
                float exqp = expf(q), exqm = 1.0f / exqp;
                y = (exqp - exqm) / (2.0f * rb);
                z = rb * rb * y;
                cosq = (exqp + exqm) / 2.0f;
            } else if (c == b[m - 1]) {
                y = dz * xkk; z = 0.0f; cosq = 1.0f;
            } else {
                float sinq = sinf(q);
                y = sinq / rb; z = -rb * sinq; cosq = cosf(q);
            }
            eut = cosq * ut - y * tq / h;
            ett = -h * z * ut + cosq * tq;
            dmm[kk - 1] = eut * eut;
            smm[kk - 1] = (ett * ett) / (h * h);         /* surfa.f:552 */
            if (kk == 3) amp[m - 1] = eut;               /* surfa.f:553-554 */
        }
        ut = eut; tq = ett;
        float dm = (dz / 22.5f) * (7.0f * (dmm[0] + dmm[4]) + 32.0f * (dmm[1] + dmm[3]) + 12.0f * dmm[2]);
        float sm = (dz / 22.5f) * (7.0f * (smm[0] + smm[4]) + 32.0f * (smm[1] + smm[3]) + 12.0f * smm[2]);
        float dldm = -(wvnosq * dm + sm);                /* surfa.f:562-565 */
        float dldr = omegsq * dm;
        s->dcdb[m - 1] = 2.0f * rho[m - 1] * b[m - 1] * c * dldm / wvno;
        s->dcdr[m - 1] = (c / wvno) * (dldr + b[m - 1] * b[m - 1] * dldm);
        sumi0 = sumi0 + rho[m - 1] * dm;
        sumi1 = sumi1 + h * dm;
    }
    {   /* surfa.f:573-597: scale by dL/dk; "exclusion of low amplitudes" */
        s->dcdb[NSZ - 1] = 0.0; s->dcdr[NSZ - 1] = 0.0;
        float dldk = -2.0f * wvno * sumi1;
        for (int l = 1; l <= mmax; ++l) {
            if (b[l - 1] == 0.0f) continue;
            amp[l - 1] = amp[l - 1] / ut;
            s->dcdb[l - 1] = s->dcdb[l - 1] / dldk;
            s->dcdr[l - 1] = s->dcdr[l - 1] / dldk;
            if (!(b[l - 1] < b[mmax - 1]) && (fabsf(amp[l - 1]) - xxmin < 0.0f)) { s->dcdb[l - 1] = 0.0; s->dcdr[l - 1] = 0.0; }
        }
    }
    s->rar_mmax = mmax;
    if (!(b[0] <= 0.0f)) {                               /* surfa.f:609-631: one entry down, entry 1 = the surface */
        for (int l = 1; l <= mmax; ++l) {
            int i = mmax - l + 1, j = i + 1;
            s->dcdb[j - 1] = s->dcdb[i - 1];
            s->dcdr[j - 1] = s->dcdr[i - 1];
        }
        s->rar_mmax = mmax + 1;
    }
    s->dcdb[0] = 0.0; s->dcdr[0] = 0.0;
    sumi0 = sumi0 / (ut * ut);
    sumi1 = sumi1 / (ut * ut);
    return sumi1 / (c * sumi0);                          /* surfa.f:606 */
}

/* ---------------------------------------------------------- surfa.f:714-1431
 * Rayleigh group velocity.  Only the part that feeds ugr (surfa.f:1186) is
 * restated: sublayering, layer dropping, water layer, two RK4 solutions
 * (fp64 state), surface combination, Boole energy integrals, half-space terms. */
static float reigen(ctx_t *s, float t, float c, float ratio)
{
    float *a = s->a, *b = s->b, *rho = s->rho, *d = s->d, *qs = s->qs;
    double (*yy1)[5] = s->yy1, (*yy2)[5] = s->yy2, (*yy3)[5] = s->yy3, (*yy4)[5] = s->yy4;
    double (*yz1)[5] = s->yz1, (*yz2)[5] = s->yz2, (*yz3)[5] = s->yz3, (*yz4)[5] = s->yz4;
    float xlamb[NSZ], xmu[NSZ];
    const float wwt[4] = {0.0f, 0.5f, 0.5f, 1.0f};
    const float wt[4] = {1.0f / 6.0f, 1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 6.0f};
    const float xxmin = 1.0e-15f;
    int mmax = s->n, nmax = mmax;
    int mm1 = mmax - 1;
    int ivre = 99 / mm1;                                 /* surfa.f:783 */
    if (s->ndiv > ivre) s->ndiv = ivre;
    int ndiv = s->ndiv;
    float div = (float)ndiv;
    if (ndiv > 1) {                                      /* surfa.f:789-820 */
        float pr_d[NSZ], pr_a[NSZ], pr_b[NSZ], pr_r[NSZ], pr_q[NSZ];
        int jj = 1;
        if (b[0] <= 0.1e-10f) jj = 2;
        for (int j = jj; j <= mm1; ++j) {
            int ldiv = (j - jj) * ndiv;
            for (int i = 1; i <= ndiv; ++i) {
                pr_d[ldiv + i - 1] = d[j - 1] / div;
                pr_a[ldiv + i - 1] = a[j - 1];
                pr_b[ldiv + i - 1] = b[j - 1];
                pr_r[ldiv + i - 1] = rho[j - 1];
                pr_q[ldiv + i - 1] = qs[j - 1];
            }
        }
        mmax = (mm1 - jj + 1) * ndiv + jj;
        d[mmax - 1] = 0.0f;
        a[mmax - 1] = a[nmax - 1];
        b[mmax - 1] = b[nmax - 1];
        rho[mmax - 1] = rho[nmax - 1];
        qs[mmax - 1] = qs[nmax - 1];
        nmax = mmax;
        mm1 = mmax - 1;
        for (int j = jj; j <= mm1; ++j) {
            d[j - 1] = pr_d[j - jj];
            a[j - 1] = pr_a[j - jj];
            b[j - 1] = pr_b[j - jj];
            rho[j - 1] = pr_r[j - jj];
            qs[j - 1] = pr_q[j - jj];
        }
    }
    mmax = nmax;
    for (int i = 0; i < mmax; ++i) {                     /* surfa.f:828-834 */
        xmu[i] = rho[i] * b[i] * b[i];
        xlamb[i] = rho[i] * (a[i] * a[i] - 2.0f * b[i] * b[i]);
    }
    for (int j = 0; j < nmax; ++j) { s->dcda[j] = 0.0; s->dcdb[j] = 0.0; s->dcdr[j] = 0.0; }   /* surfa.f:839-842 */
    /* layer dropping, surfa.f:853-866 */
    float dmax = s->fact * t * c;
    {
        float sum = 0.0f;
        int max = 0, ii, cut = 0;
        for (ii = 1; ii <= mmax; ++ii) {
            max = ii;
            if (c < b[ii - 1]) {
                sum = sum + d[ii - 1];
                if (ii == mmax) { cut = 1; break; }
                if (sum <= dmax) continue;
                if (a[ii] < a[ii - 1]) { cut = 1; break; }
                if (a[ii] == a[ii - 1]) {
                    if (b[ii] < b[ii - 1]) { cut = 1; break; }
                    if (b[ii] == b[ii - 1]) continue;
                }
                max = max + 1; cut = 1; break;           /* label 90009 */
            }
        }
        if (!cut) max = max + 1;
        mmax = max;
    }
    float sumi0 = 0.0f, sumi1 = 0.0f, sumi2 = 0.0f, sumi3 = 0.0f;
    float wvno = 6.2831853072f / (c * t);
    float wvnosq = wvno * wvno;
    float omega = 6.2831853072f / t;
    float omegsq = omega * omega;
    float tzz = 0.0f;
    if (!(b[0] > 0.0f)) {
        /* water layer, surfa.f:879-910 (complex arithmetic restated on the two
         * real branches: cra real for c>a(1), pure imaginary for c<a(1)) */
        float ra = c / a[0];
        float cr1 = ra * ra - 1.0f;
        float mag = wvno * sqrtf(fabsf(cr1));
        if (mag <= 1.0e-35f) {
            sumi0 = rho[0] * d[0];
        } else {
            float sin2ra, cosra, rab1, sinra_over;
            if (cr1 >= 0.0f) {             /* cra = mag (real) */
                sin2ra = sinf(2.0f * mag * d[0]) / (4.0f * mag);
                cosra = cosf(mag * d[0]);
                rab1 = mag * mag;
                sinra_over = sinf(mag * d[0]) / mag;
            } else {                       /* cra = i*mag */
                sin2ra = sinhf(2.0f * mag * d[0]) / (4.0f * mag);
                cosra = coshf(mag * d[0]);
                rab1 = -(mag * mag);
                sinra_over = sinhf(mag * d[0]) / mag;
            }
            float cos2rm = 1.0f / (cosra * cosra);
            float fac1 = (0.5f * d[0] + sin2ra) * cos2rm;
            float fac3 = wvno * (0.5f * d[0] - sin2ra) * cos2rm;
            float fac2 = wvno * fac3 / rab1;
            float fac4 = rab1 * fac3 / wvno;
            sumi0 = rho[0] * (fac1 + fac2);
            sumi1 = xlamb[0] * fac2;
            sumi2 = xlamb[0] * fac3;
            sumi3 = xlamb[0] * fac4;
            tzz = -rho[0] * omegsq * sinra_over / cosra;
        }
    }
    /* half-space start vectors, surfa.f:913-926 */
    float cova = c / a[mmax - 1];
    float covb = c / b[mmax - 1];
    float gam = 2.0f / (covb * covb);
    float gamm1 = gam - 1.0f;
    float ra = wvno * sqrtf(fabsf(cova * cova - 1.0f));
    float rb = wvno * sqrtf(fabsf(covb * covb - 1.0f));
    float det = wvnosq - ra * rb;
    float h = rho[mmax - 1] * omegsq;
    float brkt = -gamm1 * wvno + gam * ra * rb / wvno;
    int iter = 0;
    int mmm1 = mmax - 1;
    double aur1 = 1.0, auz1 = 0.0, atz1 = -h * brkt / det, atr1 = -h * ra / det;
    /* solution 1, surfa.f:928-979 */
    for (int mm = 1; mm <= mmm1; ++mm) {
        int m = mmax - mm - 1; /* 0-based */
        if (b[m] <= 0.0f) continue;
        float ddz = -d[m] / (4.0f * 1.0f);
        float a12 = 1.0f / (xlamb[m] + 2.0f * xmu[m]);
        float a13 = wvno * xlamb[m] * a12;
        float a21 = -omegsq * rho[m];
        float a24 = wvno, a31 = -wvno;
        float a34 = 1.0f / xmu[m];
        float a42 = -a13;
        float a43 = a21 + 4.0f * wvnosq * xmu[m] * (xlamb[m] + xmu[m]) * a12;
        yy3[m][4] = aur1; yy1[m][4] = auz1; yy2[m][4] = atz1; yy4[m][4] = atr1;
        for (int kk = 2; kk <= 5; ++kk) {
            int k = 6 - kk;
            double e1 = aur1, e2 = auz1, e3 = atz1, e4 = atr1;
            double d1 = 0, d2 = 0, d3 = 0, d4 = 0;
            for (int ll = 0; ll < 4; ++ll) {
                double s1 = aur1 + wwt[ll] * ddz * d1;
                double s2 = auz1 + wwt[ll] * ddz * d2;
                double s3 = atz1 + wwt[ll] * ddz * d3;
                double s4 = atr1 + wwt[ll] * ddz * d4;
                d1 = a31 * s2 + a34 * s4;
                d2 = a12 * s3 + a13 * s1;
                d3 = a21 * s2 + a24 * s4;
                d4 = a42 * s3 + a43 * s1;
                e1 = e1 + wt[ll] * ddz * d1;
                e2 = e2 + wt[ll] * ddz * d2;
                e3 = e3 + wt[ll] * ddz * d3;
                e4 = e4 + wt[ll] * ddz * d4;
            }
            aur1 = e1; auz1 = e2; atz1 = e3; atr1 = e4;
            yy1[m][k - 1] = auz1; yy2[m][k - 1] = atz1; yy3[m][k - 1] = aur1; yy4[m][k - 1] = atr1;
        }
    }
    if (!(b[0] > 0.0f)) {
        yy1[0][0] = yy1[1][0]; yy2[0][0] = yy2[1][0]; yy3[0][0] = yy3[1][0]; yy4[0][0] = yy4[1][0];
    }
    double xnorm = 0.0, bb = 1.0;
    for (;;) {
        /* solution 2, surfa.f:986-1055 */
        double aur2 = 0.0, auz2 = 1.0, atz2 = -h * rb / det, atr2 = -h * brkt / det;
        if (iter != 0) {
            double u1 = 1.0, z1 = 0.0, tz1 = -h * brkt / det, tr1 = -h * ra / det;
            aur2 = aur2 + xnorm * u1;
            auz2 = auz2 + xnorm * z1;
            atz2 = atz2 + xnorm * tz1;
            atr2 = atr2 + xnorm * tr1;
        }
        for (int mm = 1; mm <= mmm1; ++mm) {
            int m = mmax - mm - 1;
            if (b[m] <= 0.0f) continue;
            float ddz = -d[m] / (4.0f * 1.0f);
            float a12 = 1.0f / (xlamb[m] + 2.0f * xmu[m]);
            float a13 = wvno * xlamb[m] * a12;
            float a21 = -omegsq * rho[m];
            float a24 = wvno, a31 = -wvno;
            float a34 = 1.0f / xmu[m];
            float a42 = -a13;
            float a43 = a21 + 4.0f * wvnosq * xmu[m] * (xlamb[m] + xmu[m]) * a12;
            yz3[m][4] = aur2; yz1[m][4] = auz2; yz2[m][4] = atz2; yz4[m][4] = atr2;
            for (int kk = 2; kk <= 5; ++kk) {
                int k = 6 - kk;
                double e1 = aur2, e2 = auz2, e3 = atz2, e4 = atr2;
                double d1 = 0, d2 = 0, d3 = 0, d4 = 0;
                for (int ll = 0; ll < 4; ++ll) {
                    double s1 = aur2 + wwt[ll] * ddz * d1;
                    double s2 = auz2 + wwt[ll] * ddz * d2;
                    double s3 = atz2 + wwt[ll] * ddz * d3;
                    double s4 = atr2 + wwt[ll] * ddz * d4;
                    d1 = a31 * s2 + a34 * s4;
                    d2 = a12 * s3 + a13 * s1;
                    d3 = a21 * s2 + a24 * s4;
                    d4 = a42 * s3 + a43 * s1;
                    e1 = e1 + wt[ll] * ddz * d1;
                    e2 = e2 + wt[ll] * ddz * d2;
                    e3 = e3 + wt[ll] * ddz * d3;
                    e4 = e4 + wt[ll] * ddz * d4;
                }
                aur2 = e1; auz2 = e2; atz2 = e3; atr2 = e4;
                yz1[m][k - 1] = auz2; yz2[m][k - 1] = atz2; yz3[m][k - 1] = aur2; yz4[m][k - 1] = atr2;
            }
        }
        if (!(b[0] > 0.0f)) {
            yz1[0][0] = yz1[1][0]; yz2[0][0] = yz2[1][0]; yz3[0][0] = yz3[1][0]; yz4[0][0] = yz4[1][0];
        }
        /* surface combination, surfa.f:1056-1069 */
        double aa = yz3[0][0] - ratio * yz1[0][0];
        bb = ratio * yy1[0][0] - yy3[0][0];
        if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
        xnorm = aa / bb;
        bb = xnorm * yy1[0][0] + yz1[0][0];
        if (fabs(bb) < 1.e-10) bb = copysign(1.e-10, bb);
        float ampur_top = (float)((xnorm * yy3[0][0] + yz3[0][0]) / bb);
        iter = iter + 1;
        if (iter > 1) break;
        float xtest = fabsf(ampur_top / ratio - 1.0f);
        if (xtest >= 0.00001f) continue;
        break;
    }
    /* energy integrals, surfa.f:1087-1138 */
    float aur = 0, auz = 0, atz = 0, atr = 0;
    int m;
    for (m = 1; m <= mmax; ++m) {
        if (b[m - 1] <= 0.0f) continue;
        if (m >= mmax) break;                            /* label 77777 */
        float dz = d[m - 1] / 4.0f;
        float dmr[5], dmz[5], smr[5], smz[5], dmrsmz[5], dmzsmr[5];
        int i0 = m - 1;
        for (int kk = 0; kk < 5; ++kk) {
            aur = (float)((xnorm * yy3[i0][kk] + yz3[i0][kk]) / bb);
            auz = (float)((xnorm * yy1[i0][kk] + yz1[i0][kk]) / bb);
            atz = (float)((xnorm * yy2[i0][kk] + yz2[i0][kk]) / bb);
            atr = (float)((xnorm * yy4[i0][kk] + yz4[i0][kk]) / bb);
            float durdz = atr / xmu[i0] - wvno * auz;
            float duzdz = (atz + wvno * xlamb[i0] * aur) / (xlamb[i0] + 2.0f * xmu[i0]);
            dmr[kk] = aur * aur;
            dmz[kk] = auz * auz;
            smr[kk] = durdz * durdz;
            smz[kk] = duzdz * duzdz;
            dmrsmz[kk] = aur * duzdz;
            dmzsmr[kk] = auz * durdz;
        }
#define BOOLE(v) ((dz / 22.5f) * (7.0f * (v[0] + v[4]) + 32.0f * (v[1] + v[3]) + 12.0f * v[2]))
        double dmmr = BOOLE(dmr), dmmz = BOOLE(dmz), smmz = BOOLE(smz), smmr = BOOLE(smr);
        double drsz = BOOLE(dmrsmz), dzsr = BOOLE(dmzsmr);
#undef BOOLE
        sumi0 = (float)(sumi0 + rho[i0] * (dmmr + dmmz));
        sumi1 = (float)((xlamb[i0] + 2.0f * xmu[i0]) * dmmr + xmu[i0] * dmmz + sumi1);
        sumi2 = (float)(xmu[i0] * dzsr - xlamb[i0] * drsz + sumi2);
        sumi3 = (float)((xlamb[i0] + 2.0f * xmu[i0]) * smmz + xmu[i0] * smmr + sumi3);
        {   /* surfa.f:1130-1135 (dldl, dldm, dldr DOUBLE PRECISION; the REAL*4 factors are formed first, left to right) */
            double dldl = -wvnosq * dmmr + 2.0f * wvno * drsz - smmz;
            double dldm = -wvnosq * (2.0 * dmmr + dmmz) - 2.0f * wvno * dzsr - (2.0 * smmz + smmr);
            double dldr = omegsq * (dmmr + dmmz);
            s->dcdb[i0] = 2.0f * rho[i0] * b[i0] * c * (dldm - 2.0 * dldl) / wvno;
            s->dcda[i0] = 2.0f * rho[i0] * a[i0] * c * dldl / wvno;
            s->dcdr[i0] = (c / wvno) * (dldr + xlamb[i0] * dldl / rho[i0] + xmu[i0] * dldm / rho[i0]);
        }
        if (fabsf(auz) + fabsf(aur) - xxmin <= 0.0f) goto halfspace;   /* -> 7002 */
    }
    if (m > mmax) m = mmax + 1; /* DO exhausted (cannot happen: m==mmax breaks) */
    /* label 77777 */
    if (!((b[0] > 0.1e-10f) || m != 2)) {
        aur = ratio; auz = 1.0f; atr = 0.0f; atz = tzz;
    }
halfspace:
    {   /* label 7002, surfa.f:1145-1186 */
        int i0 = m - 1;
        (void)atr; (void)atz;
        float ap = -rho[i0] * (wvno * aur + rb * auz) / det;
        float bp = -rho[i0] * (-ra * aur / wvno - auz) / det;
        float a1 = -wvno * ap / rho[i0];
        float a2 = -wvno * rb * bp / rho[i0];
        float a3 = ra * ap / rho[i0];
        float a4 = wvnosq * bp / rho[i0];
        float ugr;
        if (rb == 0.0f) {                                /* label 7006, surfa.f:1165-1173 */
            ugr = b[i0];
            sumi0 = rho[i0] * 1.0e25f; sumi1 = xmu[i0] * 1.0e25f; sumi2 = 0.0f;
            s->dcdb[i0] = -2.0f * wvno * 1.0e25f;
        } else {
            double dmmr = a1 * a1 / (2.0f * ra) + 2.0f * a1 * a2 / (ra + rb) + a2 * a2 / (2.0f * rb);
            double dmmz = a3 * a3 / (2.0f * ra) + 2.0f * a3 * a4 / (ra + rb) + a4 * a4 / (2.0f * rb);
            double smmz = ra * a3 * a3 / 2.0f + 2.0f * ra * rb * a3 * a4 / (ra + rb) + rb * a4 * a4 / 2.0f;
            double smmr = ra * a1 * a1 / 2.0f + 2.0f * ra * rb * a1 * a2 / (ra + rb) + rb * a2 * a2 / 2.0f;
            double drsz = -a1 * a3 / 2.0f - (a1 * a4 * rb + a2 * a3 * ra) / (ra + rb) - a2 * a4 / 2.0f;
            double dzsr = -a1 * a3 / 2.0f - (a1 * a4 * ra + a2 * a3 * rb) / (ra + rb) - a2 * a4 / 2.0f;
            sumi0 = (float)(sumi0 + rho[i0] * (dmmr + dmmz));
            sumi1 = (float)((xlamb[i0] + 2.0f * xmu[i0]) * dmmr + xmu[i0] * dmmz + sumi1);
            sumi2 = (float)(xmu[i0] * dzsr - xlamb[i0] * drsz + sumi2);
            (void)sumi3;
            double dldr = omegsq * (dmmr + dmmz);        /* surfa.f:1178-1184 */
            double dldm = -wvnosq * (2.0 * dmmr + dmmz) - 2.0f * wvno * dzsr - (2.0 * smmz + smmr);
            double dldl = -wvnosq * dmmr + 2.0f * wvno * drsz - smmz;
            s->dcda[i0] = 2.0f * rho[i0] * a[i0] * c * dldl / wvno;
            s->dcdb[i0] = 2.0f * rho[i0] * b[i0] * c * (dldm - 2.0 * dldl) / wvno;
            s->dcdr[i0] = (c / wvno) * (dldr + xlamb[i0] * dldl / rho[i0] + xmu[i0] * dldm / rho[i0]);
            ugr = (wvno * sumi1 + sumi2) / (omega * sumi0);   /* surfa.f:1186 */
        }
        {   /* surfa.f:1200-1249: divide by dL/dk, dwx, shift one entry down (solid top), entry 1 = the surface */
            const int mx = mmax;                         /* (m, the entry just written, may lie above it: early exit to 7002) */
            for (int q = (b[0] <= 0.0f) ? 2 : 1; q <= mx; ++q) {
                double dldk = -2.0f * (wvno * sumi1 + sumi2);
                s->dcdr[q - 1] = s->dcdr[q - 1] / dldk;
                s->dcda[q - 1] = s->dcda[q - 1] / dldk;
                s->dcdb[q - 1] = s->dcdb[q - 1] / dldk;
                s->dwx[q - 1] = (s->dcda[q - 1] * 4.0f / 3.0f * b[q - 1] / a[q - 1] + s->dcdb[q - 1]) * b[q - 1];
            }
            s->rar_mmax = mx;
            if (!(b[0] <= 0.0f)) {
                for (int q = 1; q <= mx; ++q) {
                    int i = mx - q + 1, j = i + 1;
                    s->dcda[j - 1] = s->dcda[i - 1]; s->dcdb[j - 1] = s->dcdb[i - 1];
                    s->dcdr[j - 1] = s->dcdr[i - 1]; s->dwx[j - 1] = s->dwx[i - 1];
                }
                s->rar_mmax = mx + 1;
            }
            s->dcda[0] = 0.0; s->dcdb[0] = 0.0; s->dcdr[0] = 0.0; s->dwx[0] = 0.0;
        }
        return ugr;
    }
}

/* ------------------------------------------- fast_surf.f:2-211 + calcul.f:2-420 */
static int forward_ctx(ctx_t *s, int nlay, int kind,
                       const float *vp, const float *vs, const float *rho,
                       const float *h, const float *qsinv,
                       const float *per, int nper,
                       float *c_out, float *u_out, int *nsolved, long *n_delta_out, float *ratio_out)
{
    if (nsolved) *nsolved = 0;
    if (nlay < 2 || nlay > SURFDISP_NLAY_MAX || nper < 1 || nper > SURFDISP_NPER_MAX ||
        (kind != 1 && kind != 2))
        return SURFDISP_ORACLE_EINVAL;
    const float pi = 3.1415927f, t_base = 1.0f, dc = 0.01f;
    float c[SURFDISP_NPER_MAX], ratio[SURFDISP_NPER_MAX], cgrp[SURFDISP_NPER_MAX];
    int status = SURFDISP_ORACLE_OK;
    /* a new process sees zeroed COMMON blocks */
    memset(s->a, 0, 5 * NSZ * sizeof(float));
    s->n_delta = 0; s->tr_n = 0;
    s->n = nlay; s->kind = kind;
    for (int i = 0; i < nlay; ++i) {                      /* fast_surf.f:89-99 */
        s->a_ref[i] = vp[i]; s->b_ref[i] = vs[i]; s->rho_ref[i] = rho[i];
        s->qs_ref[i] = qsinv[i]; s->d_ref[i] = h[i];
    }
    for (int k = 0; k < nper; ++k) { c_out[k] = 0.0f; u_out[k] = 0.0f; c[k] = 0.0f; ratio[k] = 0.0f; }
    s->ndiv = 5; s->fact = 4.0f;                          /* init.f:25 */
    s->nmax = nlay; s->mmax = nlay; s->idrop = 0;
    /* first guess, fast_surf.f:157-171 */
    int ilay = 0;
    if (s->b_ref[0] < 0.1f) ilay = 1;
    float b_corr = s->qs_ref[ilay] * logf(t_base / per[0]) / pi;
    float qq = s->b_ref[ilay];
    if (kind == 2) qq = 0.9f * qq;
    float c1 = qq * (1.0f + b_corr);
    if (s->b_ref[0] < 0.1f) c1 = 0.5f;

    int imax = 0, fatal = 0;
    const int ifunc = kind;
    /* phase 1: calcul.f:104-220 */
    for (int k = 0; k < nper; ++k) {
        float t1 = per[k];
        build_model(s, t1, s->mmax);                      /* first mmax layers only! */
        flat1(s->d, s->rho, s->a, s->b, s->mmax, kind);
        if (k > 0) c1 = 0.90f * c[k - 1];
        s->idrop = 0;
        float del1 = dltar(s, c1, t1, ifunc);
        const int tracing = (s->tr_c != NULL) && (s->tr_k == k);
        if (tracing && s->tr_n < s->tr_cap) { s->tr_c[s->tr_n] = c1; s->tr_d[s->tr_n] = del1; s->tr_mm[s->tr_n++] = s->mmax; }
        float c2, del2;
        int found = 0, failed = 0;
        for (;;) {
            c2 = c1 + dc;
            s->idrop = 0;
            del2 = dltar(s, c2, t1, ifunc);
            if (tracing && s->tr_n < s->tr_cap) { s->tr_c[s->tr_n] = c2; s->tr_d[s->tr_n] = del2; s->tr_mm[s->tr_n++] = s->mmax; }
            if (tracing && s->tr_nsub > 0 && sgn1(del1) != sgn1(del2)) {
                s->tr_n = 0;
                s->idrop = 1;
                for (int e = 0; e <= s->tr_nsub && s->tr_n < s->tr_cap; ++e) {
                    const float cx = c1 + (c2 - c1) * ((float)e / (float)s->tr_nsub);
                    s->tr_c[s->tr_n] = cx; s->tr_d[s->tr_n] = dltar(s, cx, t1, ifunc); s->tr_mm[s->tr_n++] = s->mmax;
                }
                return SURFDISP_ORACLE_OK;
            }
            if (tracing && sgn1(del1) != sgn1(del2)) {          /* keep scanning tr_extra points past the bracket */
                float cx = c2;
                for (int e = 0; e < s->tr_extra && s->tr_n < s->tr_cap; ++e) {
                    cx = cx + dc; s->idrop = 0;
                    const float dx = dltar(s, cx, t1, ifunc);
                    s->tr_c[s->tr_n] = cx; s->tr_d[s->tr_n] = dx; s->tr_mm[s->tr_n++] = s->mmax;
                }
                return SURFDISP_ORACLE_OK;
            }
            if (sgn1(del1) != sgn1(del2)) { found = 1; break; }
            c1 = c2; del1 = del2;
            if (c1 - 0.8f * s->b[0] < 0.0f) { failed = 1; break; }
            if (!(c1 - (s->b[s->mmax - 1] + 0.3f) < 0.0f)) { failed = 1; break; }
        }
        if (found) {
            float cn = 0.0f;
            if (nevill(s, t1, c1, c2, del1, del2, ifunc, &cn)) {
                fatal = 1; status = SURFDISP_ORACLE_NEVILL;   /* goto 9999: no phase 2 */
                break;
            }
            c1 = cn;
            if (c1 - s->b[s->mmax - 1] <= 0.0f) {
                c[k] = c1;
                cgrp[k] = (s->c_at && s->c_at[k] > 0.0f) ? s->c_at[k] : c1;     /* test hook, see ctx_t */
                if (ifunc == 2) ratio[k] = dltar(s, cgrp[k], t1, 3);
                imax = k + 1;
                continue;
            }
            failed = 1;
        }
        if (failed) {                                     /* label 250 */
            status = (k == 0) ? SURFDISP_ORACLE_NOROOT : SURFDISP_ORACLE_PARTIAL;
            if (k == 0) fatal = 1;
            break;
        }
    }
    /* phase 2: calcul.f:224-365 */
    if (!fatal) {
        s->mmax = s->nmax;
        for (int lip = 0; lip < imax; ++lip) {
            float t = per[lip];
            build_model(s, t, nlay);
            for (int i = 0; i < nlay; ++i) s->qs[i] = s->qs_ref[i];
            flat1(s->d, s->rho, s->a, s->b, nlay, kind);
            float ugr = (kind == 2) ? reigen(s, t, cgrp[lip], ratio[lip]) : leigen(s, t, cgrp[lip]);
            u_out[lip] = ugr;
            c_out[lip] = c[lip];
            if (ratio_out) ratio_out[lip] = ratio[lip];
        }
        if (nsolved) *nsolved = imax;
    }
    if (n_delta_out) *n_delta_out = s->n_delta;
    return status;
}

int surfdisp_oracle_forward(int nlay, int kind,
                            const float *vp, const float *vs, const float *rho,
                            const float *h, const float *qsinv,
                            const float *per, int nper,
                            float *c_out, float *u_out, int *nsolved, long *n_delta_out)
{
    ctx_t *s = (ctx_t *)malloc(sizeof(ctx_t));
    if (!s) return SURFDISP_ORACLE_EINVAL;
    s->tr_c = NULL; s->c_at = NULL;
    int st = forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, per, nper, c_out, u_out,
                         nsolved, n_delta_out, NULL);
    free(s);
    return st;
}

/* developer aid: the secular function across the bracket of period k (nsub + 1 equidistant points, frozen layer dropping);
 * returns the number of points written (0: no bracket) */
int surfdisp_oracle_bracket_profile(int nlay, int kind,
                                    const float *vp, const float *vs, const float *rho,
                                    const float *h, const float *qsinv,
                                    const float *per, int nper, int k, int nsub,
                                    float *c_tr, float *d_tr, int *mm_tr, int cap)
{
    ctx_t *s = (ctx_t *)calloc(1, sizeof(ctx_t));
    if (!s) return 0;
    float cbuf[SURFDISP_NPER_MAX], ubuf[SURFDISP_NPER_MAX];
    s->tr_k = k; s->tr_cap = cap; s->tr_extra = 0; s->tr_nsub = nsub; s->tr_c = c_tr; s->tr_d = d_tr; s->tr_mm = mm_tr;
    forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, per, nper, cbuf, ubuf, NULL, NULL, NULL);
    const int n = (s->tr_nsub > 0 && s->tr_n == nsub + 1) ? s->tr_n : 0;
    free(s);
    return n;
}

/* The analytic partials of ONE period: a one-period solve (the reference overwrites COMMON /rar1/ at every period, so a
 * one-period call is how a period's values are read from it, oracle/refso.py::last_partials), then dcda, dcdb, dcdr, dwx
 * [1000] as REIGEN / LEIGEN leave them (per sublayer of the flattened, attenuated stack, shifted one entry down when the
 * top layer is solid), *mmax = COMMON /rar/ mmax, *ndiv = the clamped COMMON /c/ ndiv.  Love leaves dcda, dwx untouched
 * (written as zeros here).  Returns the solve's status; *c_out, *u_out = the period's phase and group velocity. */
int surfdisp_oracle_partials(int nlay, int kind,
                             const float *vp, const float *vs, const float *rho,
                             const float *h, const float *qsinv, float period,
                             double *dcda, double *dcdb, double *dcdr, double *dwx,
                             int *mmax, int *ndiv, float *c_out, float *u_out)
{
    ctx_t *s = (ctx_t *)calloc(1, sizeof(ctx_t));
    if (!s) return SURFDISP_ORACLE_EINVAL;
    float c1 = 0.0f, u1 = 0.0f;
    int st = forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, &period, 1, &c1, &u1, NULL, NULL, NULL);
    for (int i = 0; i < NSZ; ++i) {
        dcda[i] = (kind == 2) ? s->dcda[i] : 0.0; dcdb[i] = s->dcdb[i]; dcdr[i] = s->dcdr[i];
        dwx[i] = (kind == 2) ? s->dwx[i] : 0.0;
    }
    if (mmax) *mmax = s->rar_mmax;
    if (ndiv) *ndiv = s->ndiv;
    if (c_out) *c_out = c1;
    if (u_out) *u_out = u1;
    free(s);
    return st;
}

/* developer aid: the secular-function values the scan of period k evaluates (grid point, value, effective half
 * space), continued `extra` grid points past the first sign change; returns the number of points written */
int surfdisp_oracle_scan_trace(int nlay, int kind,
                               const float *vp, const float *vs, const float *rho,
                               const float *h, const float *qsinv,
                               const float *per, int nper, int k, int extra,
                               float *c_tr, float *d_tr, int *mm_tr, int cap)
{
    ctx_t *s = (ctx_t *)calloc(1, sizeof(ctx_t));
    if (!s) return 0;
    float cbuf[SURFDISP_NPER_MAX], ubuf[SURFDISP_NPER_MAX];
    s->tr_k = k; s->tr_cap = cap; s->tr_extra = extra; s->tr_c = c_tr; s->tr_d = d_tr; s->tr_mm = mm_tr;
    forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, per, nper, cbuf, ubuf, NULL, NULL, NULL);
    const int n = s->tr_n;
    free(s);
    return n;
}

/* debug variant: also returns the Rayleigh ellipticity ratio(k) of calcul.f:195 */
int surfdisp_oracle_forward_dbg(int nlay, int kind,
                                const float *vp, const float *vs, const float *rho,
                                const float *h, const float *qsinv,
                                const float *per, int nper,
                                float *c_out, float *u_out, float *ratio_out)
{
    ctx_t *s = (ctx_t *)malloc(sizeof(ctx_t));
    if (!s) return SURFDISP_ORACLE_EINVAL;
    s->tr_c = NULL; s->c_at = NULL;
    int st = forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, per, nper, c_out, u_out,
                         NULL, NULL, ratio_out);
    free(s);
    return st;
}

/* Test hook: the oracle's own root search, but ellipticity (calcul.f:195) and group velocity (REIGEN / LEIGEN) of
 * period k evaluated at the GIVEN phase velocity c_at[k] (> 0; else at the oracle's root).  Answers "what does the
 * reference's group-velocity computation return at the phase velocity another implementation found" - near osculating
 * modes U changes by >1000x the relative change of c, so comparing U at slightly different c says nothing about the
 * group-velocity arithmetic.  c_out = the oracle's own roots. */
int surfdisp_oracle_forward_at(int nlay, int kind,
                               const float *vp, const float *vs, const float *rho,
                               const float *h, const float *qsinv,
                               const float *per, int nper, const float *c_at,
                               float *c_out, float *u_out)
{
    ctx_t *s = (ctx_t *)malloc(sizeof(ctx_t));
    if (!s) return SURFDISP_ORACLE_EINVAL;
    s->tr_c = NULL; s->c_at = c_at;
    int st = forward_ctx(s, nlay, kind, vp, vs, rho, h, qsinv, per, nper, c_out, u_out, NULL, NULL, NULL);
    free(s);
    return st;
}

/* Fortran-ABI shim with the reference symbol's signature (fast_surf.f:2-5);
 * lets tests drive reference and oracle through the same ctypes code. */
void surfdisp_oracle_fast_surf_(const int *n_layer, const int *kind,
                                const float *vp, const float *vs, const float *rho,
                                const float *h, const float *qsinv,
                                const float *per, const int *nper,
                                float *uR, float *uL, float *cR, float *cL)
{
    float c[SURFDISP_NPER_MAX], u[SURFDISP_NPER_MAX];
    int np_ = *nper, ns = 0;
    if (np_ > SURFDISP_NPER_MAX) np_ = SURFDISP_NPER_MAX;
    int st = surfdisp_oracle_forward(*n_layer, *kind, vp, vs, rho, h, qsinv, per, np_, c, u, &ns, NULL);
    if (st == SURFDISP_ORACLE_EINVAL) return;
    for (int i = 0; i < ns; ++i) {                        /* fast_surf.f:197-208 */
        if (*kind == 1) { cL[i] = c[i]; uL[i] = u[i]; }
        else            { cR[i] = c[i]; uR[i] = u[i]; }
    }
}

/* batched driver: model[B][5][Lmax] = (vp, vs, rho, h, qsinv); OpenMP over models */
int surfdisp_oracle_forward_batch(int B, int Lmax, const int *nlay, const float *model,
                                  int P, const float *per, int kind,
                                  float *c, float *u, int *status, int nthreads)
{
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+:bad)
#endif
    {
        ctx_t *s = (ctx_t *)malloc(sizeof(ctx_t));
        if (s) { s->tr_c = NULL; s->c_at = NULL; }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < B; ++i) {
            const float *m = model + (size_t)i * 5 * Lmax;
            int n = nlay ? nlay[i] : Lmax;
            int ns = 0;
            int st = s ? forward_ctx(s, n, kind, m, m + Lmax, m + 2 * Lmax, m + 3 * Lmax,
                                     m + 4 * Lmax, per, P, c + (size_t)i * P, u + (size_t)i * P,
                                     &ns, NULL, NULL)
                       : SURFDISP_ORACLE_EINVAL;
            if (status) status[i] = st;
            if (st != SURFDISP_ORACLE_OK) bad++;
        }
        free(s);
    }
    return bad;
}
