// surfdisp_thermal.hip -- the thermal mantle layer of a parameters -> layer stack evaluation on the
// device: OceanMantleHybrid (layers.py:297-363) over HSCM, OceanSeisRitz and
// OceanSeisRuan (ThermSeis.py:56-173, 325-448).  One 64-lane workgroup per chain, lane j = grid
// point j of the layer (at most 61 points); the result (vs, qs per grid point) goes to a scratch
// array that surfdisp_layers_kernel reads for layers of kind 6.
//
// Arithmetic is fp64 with contraction off and the operation order of pysurfinv_amd/thermseis.py
// (the torch host mirror, itself pinned to the reference at 1e-14), so the two agree to the last
// bits of erf/exp/pow/log.  The ragged not-a-knot spline system is solved serially (Thomas) by every
// lane of the wave - 61 unknowns - where the torch mirror uses cyclic reduction.
//
// Descriptor tail (after the arrays documented in surfdisp_layers.hip):
//   idesc tail (10 ints): layer index, ThermAge slot, Tp slot, conversion (0 Ritzwoller, 1 Yamauchi),
//                        Q age is constant (lithoAgeQ), bit mask of the crust layers above, npts,
//                        slot + 1 of a per-row Info.lithoAge, slot + 1 of a per-row Info.period (0 = the constants), 0
//   fdesc tail (4 doubles): ThermAge const, Tp const, Q age const, Info.period
#include <hip/hip_runtime.h>
#include "surfdisp_internal.h"

#pragma clang fp contract(off)

namespace sd {

namespace {

constexpr double C2K = 273.15;
constexpr double YEAR = 365.0 * 24 * 3600;

struct Tm2 { double Tm, z_adia; };

// calTm of HSCM._calT, ThermSeis.py:64-79 (thermseis.hscm_mantle_temperature)
__device__ Tm2 mantle_temperature(double age, double Tp)
{
    const double T0 = 0.0, Da = 0.4;
    const double scale = 1e3 / (2 * sqrt(age * YEAR * 1 * 1.0));
    auto f = [&](double z) { return erf(z * scale); };
    auto g = [&](double z) {
        const double dz = 0.001;
        const double fz = f(z);
        const double dfz = (f(z + dz) - fz) / dz + 1e-10;
        return fz / dfz - z - (Tp - T0) / Da;
    };
    double z0 = 0.0, z1 = 400.0;
    for (int it = 0; it < 16; ++it) {
        const double z2 = (z1 + z0) / 2;
        if (g(z2) < 0) z0 = z2; else z1 = z2;
    }
    return {(Da * z1 + Tp - T0) / f(z1) + T0, z0};
}

// torch.linspace(0, 200, 200)[k]: start + step k in the lower half, end - step (199 - k) (one rounding)
// in the upper half
__device__ double default_depth(int k)
{
    const double step = (200.0 - 0.0) / 199;
    return k < 100 ? step * k : fma(-step, (double)(199 - k), 200.0);
}

struct TPR { double T, P, rho; };

// HSCM fields at depth zd km (ThermSeis.py:22-35, 80-101; thermseis.hscm)
__device__ TPR thermal_at(double zd, double age, double Tp, Tm2 m)
{
    TPR r;
    r.P = 3.4e3 * 9.8 * zd * 1000;
    const double theta = erf(zd * 1e3 / (2 * sqrt(age * YEAR * 1 * 1.0)));
    double T = (m.Tm - 0.0) * theta + 0.0;
    if (zd > m.z_adia) T = Tp + zd * 0.4;
    r.T = T + C2K;
    r.rho = 3.43e3 * (1 - 4.4e-5 * (r.T - (500 + C2K))) * (1 + 6.12e-12 * (r.P - 0.6e9));
    return r;
}

// OceanSeisRitz, RhoType 'raw', X = 0.1, default mineral fractions (ThermSeis.py:103-173)
__device__ double ritz_vs(const TPR &t)
{
    //                     rho0     rho_X    K0   K_T    K_P  K_X  mu0  mu_T   mu_P mu_X  a0         a1          a2          a3
    const double M[5][14] = {
        {3.222e3, 1.182e3, 129, -16e-3, 4.2, 0,   82,  -14e-3, 1.4, -30, 0.2010e-4, 0.1390e-7,  0.1627e-2,  -0.3380},
        {3.198e3, 0.804e3, 111, -12e-3, 6.0, -10, 81,  -11e-3, 2.0, -29, 0.3871e-4, 0.0446e-7,  0.0343e-2,  -1.7278},
        {3.280e3, 0.377e3, 105, -13e-3, 6.2, 13,  67,  -10e-3, 1.7, -6,  0.3206e-4, 0.0811e-7,  0.1347e-2,  -1.8167},
        {3.578e3, 0.702e3, 198, -28e-3, 5.7, 12,  108, -12e-3, 0.8, -24, 0.6969e-4, -0.0108e-7, -3.0799e-2, 5.0395},
        {3.565e3, 0.758e3, 173, -21e-3, 4.9, 7,   92,  -10e-3, 1.4, -7,  0.0991e-4, 0.1165e-7,  1.0624e-2,  -2.5000}};
    const double W[5] = {0.75, 0.21, 0.035, 0.0, 0.005};
    const double X = 0.1, Tref = 273.15, Pref = 101.325e-6;
    const double T = t.T, P = t.P / 1e9;
    double rho_a = 0, mu_v = 0, mu_r = 0;
    for (int i = 0; i < 5; ++i) {
        const double *d = M[i];
        const double alpha = d[10] + d[11] * T + d[12] * (1.0 / T) + d[13] * (1.0 / (T * T));
        const double rho0X = d[0] * d[1] / 1e3;
        const double mu = d[6] + (T - Tref) * d[7] + (P - Pref) * d[8] + X * d[9];
        const double K = d[2] + (T - Tref) * d[3] + (P - Pref) * d[4] + X * d[5];
        const double rho = rho0X * (1 - alpha * (T - Tref) + (P - Pref) / K);
        rho_a = rho_a + W[i] * rho;
        mu_v = mu_v + W[i] * mu;
        mu_r = mu_r + W[i] / mu;
    }
    const double mu = 0.5 * (mu_v + 1 / mu_r) * 1e9;
    return sqrt(mu / rho_a) / 1000;
}

// OceanSeisRuan (ThermSeis.py:433-448) over OceanSeisYaTa._anel with the damp solidus (:325-412)
__device__ void ruan(const TPR &t, double period, double &vs, double &qs)
{
    const double T = t.T, P = t.P;
    const double Ju = 1 / (72.45 - 0.01094 * (T - C2K) + 1.75 * P / 1e9) * 1e-9;
    const double Pg = P / 1e9;
    const double sol = -5.1 * (Pg * Pg) + 92.5 * Pg + 1120.6 + C2K;
    const double Tn = T / sol;
    const double safe = Tn > 0 ? Tn : 1.0;
    double a_eta, a_p, sig_p;
    if (Tn < 0.94) a_eta = 1.0;
    else if (Tn < 1) a_eta = exp(-(Tn - 0.94) / (safe - safe * 0.94) * 1.6094379124341003 /* log 5 */);
    else a_eta = 1.0 / 5;
    if (Tn < 0.91) a_p = 0.01; else if (Tn < 0.96) a_p = 0.01 + 0.4 * (Tn - 0.91); else a_p = 0.03;
    if (Tn < 0.92) sig_p = 4.0; else if (Tn < 1) sig_p = 4 + 37.5 * (Tn - 0.92); else sig_p = 7.0;
    const double E = 4.625e5, R = 8.314, V = 7.913e-6, etaR = 6.22e21, TR = 1200 + C2K, PR = 1.5e9;
    const double mu_U = (72.45 - 0.01094 * (T - C2K) + 1.75 * P * 1e-9) * 1e9;
    const double eta = etaR * exp(E / R * (1 / T - 1 / TR)) * exp(V / R * (P / T - PR / TR)) * a_eta;
    const double tau_M = eta / mu_U;
    const double tau_ns = period / (2 * 3.141592653589793 * tau_M);
    const double lg = log(6e-5 / tau_ns) / (1.4142135623730951 * sig_p);
    const double pw = pow(tau_ns, 0.38);
    const double J1b = 0.664 * pw / 0.38;
    const double J1p = 2.5066282746310002 / 2 * a_p * sig_p * (1 - erf(lg));
    const double J2b = 3.141592653589793 / 2 * 0.664 * pw;
    const double J2p = 3.141592653589793 / 2 * (a_p * exp(-(lg * lg)));
    const double J1 = 1 + J1b + J1p, J2 = J2b + J2p + tau_ns;
    vs = 1 / sqrt(t.rho * Ju * J1) / 1000;
    qs = J1 / J2;
}

}  // namespace

__global__ __launch_bounds__(64) void surfdisp_thermal_kernel(LayersArgs A)
{
    __shared__ double xk[64], yk[64], sk[64], cp[64], dp[64];
    const int c = blockIdx.x, j = threadIdx.x;
    const int nin = A.idesc[0], ngrid = A.idesc[1], L = A.idesc[2];
    const int *lay_i = A.idesc + 4;
    const int *coef_i = lay_i + 8 * nin;
    const int *hyb_i = coef_i + 8 * nin + L;
    const double *lay_f = A.fdesc + 1;
    const double *grid_f = lay_f + 9 * nin;
    const double *hyb_f = grid_f + 9 * ngrid;
    const double *p = A.params + (size_t)c * A.N;
    const int lh = hyb_i[0], npts = hyb_i[6];

    // thicknesses above the thermal layer (as in surfdisp_layers_kernel)
    double z = (lay_i[6] > 0) ? -fmax(p[lay_i[6] - 1], 0.0) : A.fdesc[0], crust_h = 0.0, H = 0.0, z0 = 0.0;
    for (int l = 0; l <= lh; ++l) {
        const int hs = lay_i[8 * l + 1];
        double Hl = (hs >= 0) ? p[hs] : lay_f[9 * l];
        if (lay_i[8 * l + 2]) Hl = Hl - z;
        if (l == lh) { z0 = z; H = Hl; }
        else if ((hyb_i[5] >> l) & 1) crust_h = crust_h + Hl;
        z += Hl;
    }
    const double age_p = hyb_i[1] >= 0 ? p[hyb_i[1]] : hyb_f[0];
    const double age = fmax(age_p, 1e-3);
    const double Tp = hyb_i[2] >= 0 ? p[hyb_i[2]] : hyb_f[1];
    const double q_age = fmax(hyb_i[4] ? (hyb_i[7] > 0 ? p[hyb_i[7] - 1] : hyb_f[2]) : age_p, 1e-3);
    const double period = hyb_i[8] > 0 ? p[hyb_i[8] - 1] : hyb_f[3];

    const Tm2 m_vs = mantle_temperature(age, Tp);
    const Tm2 m_def = (Tp == 1325.0) ? m_vs : mantle_temperature(age, 1325.0);
    const Tm2 m_q = (q_age == age) ? m_def : mantle_temperature(q_age, 1325.0);

    // meltStart (layers.py:312-320): first of the 200 default depths hotter than 0.92 x damp solidus
    int first = 1 << 30;
    for (int k = j; k < 200; k += 64) {
        const TPR t = thermal_at(default_depth(k), age, 1325.0, m_def);
        const double Pg = t.P / 1e9;
        const double sol = -5.1 * (Pg * Pg) + 92.5 * Pg + 1120.6 + C2K;
        if (t.T > 0.92 * sol) first = min(first, k);
    }
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    if (first == (1 << 30)) first = 199;
    const double z_melt = default_depth(first) - crust_h;

    // this lane's grid point
    const bool on = j < npts;
    const int g = lay_i[8 * lh + 4] + (on ? j : 0);
    const double *gf = grid_f + 9 * g;
    const double zj = gf[0] * H;
    const TPR t = thermal_at(crust_h + zj, age, Tp, m_vs);
    double vs_th, dummy;
    if (hyb_i[3] == 1) ruan(t, 1.0, vs_th, dummy); else vs_th = ritz_vs(t);
    double pert = 0.0;
    const int nc = lay_i[8 * lh + 3];
    for (int k = 0; k < nc; ++k) {
        const int sl = coef_i[8 * lh + k];
        pert += gf[1 + k] * ((sl >= 0) ? p[sl] : lay_f[9 * lh + 1 + k]);
    }
    const double x_hi = (z_melt + crust_h) * 1.7 - crust_h;
    const bool upper = zj < z_melt, lower = zj > x_hi;
    const bool keep = on && (upper || lower);
    const double yj = upper ? vs_th : pert + vs_th;

    // knots to the front (merge2, layers.py:321-325)
    const unsigned long long mask = __ballot(keep);
    const int n = max(__popcll(mask), 2);
    const int ki = __popcll(mask & ((1ull << j) - 1ull));
    if (keep) { xk[ki] = zj; yk[ki] = yj; }
    __syncthreads();

    // knot derivatives of the not-a-knot cubic spline (thermseis.cubic_spline_through), solved by
    // plain elimination - the same serial work in every lane, lane 0 stores
    if (j == 0) {
        auto dx = [&](int i) { return xk[i + 1] - xk[i]; };
        auto sl = [&](int i) { return (yk[i + 1] - yk[i]) / dx(i); };
        if (n == 2) {
            sk[0] = sk[1] = sl(0);
        } else {
            auto row = [&](int i, double &a, double &b, double &cc, double &r) {
                if (n == 3) {
                    if (i == 0) { a = 0; b = 1; cc = 1; r = 2 * sl(0); }
                    else if (i == 1) { a = dx(1); b = 2 * (dx(0) + dx(1)); cc = dx(0); r = 3 * (dx(0) * sl(1) + dx(1) * sl(0)); }
                    else { a = 1; b = 1; cc = 0; r = 2 * sl(1); }
                } else if (i == 0) {
                    const double d = xk[2] - xk[0];
                    a = 0; b = dx(1); cc = d;
                    r = ((dx(0) + 2 * d) * dx(1) * sl(0) + dx(0) * dx(0) * sl(1)) / d;
                } else if (i == n - 1) {
                    const double d = xk[n - 1] - xk[n - 3];
                    const double dxa = dx(n - 2), dxb = dx(n - 3);
                    a = d; b = dxb; cc = 0;
                    r = (dxa * dxa * sl(n - 3) + (2 * d + dxa) * dxb * sl(n - 2)) / d;
                } else {
                    a = dx(i); b = 2 * (dx(i - 1) + dx(i)); cc = dx(i - 1);
                    r = 3 * (dx(i) * sl(i - 1) + dx(i - 1) * sl(i));
                }
            };
            double a, b, cc, r;
            row(0, a, b, cc, r);
            cp[0] = cc / b; dp[0] = r / b;
            for (int i = 1; i < n; ++i) {
                row(i, a, b, cc, r);
                const double den = b - a * cp[i - 1];
                cp[i] = cc / den;
                dp[i] = (r - a * dp[i - 1]) / den;
            }
            sk[n - 1] = dp[n - 1];
            for (int i = n - 2; i >= 0; --i) sk[i] = dp[i] - cp[i] * sk[i + 1];
        }
    }
    __syncthreads();

    // evaluate at this lane's depth: piece = (number of knots <= x) - 1, clamped to [0, n-2]
    int piece = -1;
    for (int i = 0; i < n; ++i) piece += (xk[i] <= zj) ? 1 : 0;
    piece = min(max(piece, 0), n - 2);
    const double dxp = xk[piece + 1] - xk[piece];
    const double slope = (yk[piece + 1] - yk[piece]) / dxp;
    const double tq = (sk[piece] + sk[piece + 1] - 2 * slope) / dxp;
    const double c3 = tq / dxp;
    const double c2 = (slope - sk[piece]) / dxp - tq;
    const double tt = zj - xk[piece];
    const double vs = yk[piece] + tt * (sk[piece] + tt * (c2 + tt * c3));

    // Qs of the pre-melting model at the lithospheric age (layers.py:350-363)
    const TPR tq_ = thermal_at(z0 + zj, q_age, 1325.0, m_q);
    double vq, qs;
    ruan(tq_, period, vq, qs);
    qs = fmin(qs, 5000.0);
    if (on) {
        A.scratch[((size_t)c * 64 + j) * 2 + 0] = vs;
        A.scratch[((size_t)c * 64 + j) * 2 + 1] = qs;
    }
}

hipError_t launch_thermal(hipStream_t s, const LayersArgs &a)
{
    hipLaunchKernelGGL(surfdisp_thermal_kernel, dim3(a.C), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace sd
