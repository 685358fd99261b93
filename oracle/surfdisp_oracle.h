/* surfdisp_oracle.h -- CPU ORACLE (test infrastructure, not a product path).
 * See surfdisp_oracle.c for the parity statement. */
#ifndef SURFDISP_ORACLE_H
#define SURFDISP_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define SURFDISP_NPER_MAX 200   /* fast_surf.f:9  nper  (cvper/outputs are float[200]) */
#define SURFDISP_NLAY_MAX 200   /* oracle limit; the reference's COMMON arrays hold 1000 */

enum {
    SURFDISP_ORACLE_OK      = 0, /* every period solved */
    SURFDISP_ORACLE_PARTIAL = 1, /* bracketing failed at period k>1: periods k..P are 0 (calcul.f:203,218-219) */
    SURFDISP_ORACLE_NOROOT  = 2, /* bracketing failed at the first period: all 0 (calcul.f:203-212) */
    SURFDISP_ORACLE_NEVILL  = 3, /* NEVILL ran 50 cycles (surfa.f:17-28): all 0 under fresh-process semantics */
    SURFDISP_ORACLE_EINVAL  = -1
};

/* One solve: (layer stack, wave type) -> c[P], U[P]; unsolved periods are 0.
 * kind: 1 Love, 2 Rayleigh.  qsinv = 1/Qs.  Periods ascending.
 * nsolved (optional) = imax(1); n_delta (optional) = secular-function evaluations. */
int surfdisp_oracle_forward(int nlay, int kind,
                            const float *vp, const float *vs, const float *rho,
                            const float *h, const float *qsinv,
                            const float *per, int nper,
                            float *c_out, float *u_out, int *nsolved, long *n_delta);

/* developer aid: secular-function values of the scan of period k (see the .c file) */
int surfdisp_oracle_scan_trace(int nlay, int kind,
                               const float *vp, const float *vs, const float *rho,
                               const float *h, const float *qsinv,
                               const float *per, int nper, int k, int extra,
                               float *c_tr, float *d_tr, int *mm_tr, int cap);

int surfdisp_oracle_bracket_profile(int nlay, int kind,
                                    const float *vp, const float *vs, const float *rho,
                                    const float *h, const float *qsinv,
                                    const float *per, int nper, int k, int nsub,
                                    float *c_tr, float *d_tr, int *mm_tr, int cap);

int surfdisp_oracle_forward_dbg(int nlay, int kind,
                                const float *vp, const float *vs, const float *rho,
                                const float *h, const float *qsinv,
                                const float *per, int nper,
                                float *c_out, float *u_out, float *ratio_out);

/* test hook: group velocity (and ellipticity) evaluated at given phase velocities c_at[nper] (see the .c file) */
int surfdisp_oracle_forward_at(int nlay, int kind,
                               const float *vp, const float *vs, const float *rho,
                               const float *h, const float *qsinv,
                               const float *per, int nper, const float *c_at,
                               float *c_out, float *u_out);

/* analytic partials dc/d(a, b, rho) of ONE period as REIGEN / LEIGEN leave them in COMMON /rar1/ (double[1000] each; see the .c file) */
int surfdisp_oracle_partials(int nlay, int kind,
                             const float *vp, const float *vs, const float *rho,
                             const float *h, const float *qsinv, float period,
                             double *dcda, double *dcdb, double *dcdr, double *dwx,
                             int *mmax, int *ndiv, float *c_out, float *u_out);

/* NOT the reference: switches the whole library to an "equally valid" fp32 evaluation of the reference's formulas (bit 0:
 * exponentials of the secular functions through exp2f; bit 1: flattening factors from double-precision log / pow); 0 = the
 * reference's arithmetic.  For scripts/soak.py's classification of mismatches only; never set by the tests that pin parity. */
void surfdisp_oracle_set_variant(int v);

/* Same signature as the reference's Fortran symbol fast_surf_ (fast_surf.f:2-5). */
void surfdisp_oracle_fast_surf_(const int *n_layer, const int *kind,
                                const float *vp, const float *vs, const float *rho,
                                const float *h, const float *qsinv,
                                const float *per, const int *nper,
                                float *uR, float *uL, float *cR, float *cL);

/* model[B][5][Lmax] rows = vp, vs, rho, h, qsinv; nlay may be NULL (= Lmax). Returns #non-OK. */
int surfdisp_oracle_forward_batch(int B, int Lmax, const int *nlay, const float *model,
                                  int P, const float *per, int kind,
                                  float *c, float *u, int *status, int nthreads);
#ifdef __cplusplus
}
#endif
#endif
