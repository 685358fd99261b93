// tests/hostcheck/hostcheck.hip -- DEBUG HARNESS, test infrastructure only.
// Compiles the group-velocity device math of surfdisp_kernels.hip for the HOST so that it can be
// stepped through / compared with the oracle in the CPU-only development container.  It is not
// linked into libsurfdisp_hip.so and nothing in pysurfinv_amd/ loads it.
#include "../../pysurfinv_amd/csrc/surfdisp_kernels.hip"
#include <vector>
#include <cstdlib>

extern "C" int sd_hostcheck_group(int B, int Lmax, const int *nlay, const float *model, int P,
                                  const float *per, int kind, const float *c, const float *ratio,
                                  float *u, double *dbg)
{
    std::vector<float> mdl((size_t)10 * Lmax * B);
    std::vector<int> nl(B);
    sd::PrepArgs pa{B, Lmax, nlay, model, mdl.data(), nl.data()};
    pa.write_soa = 1;
    for (int b = 0; b < B; ++b) {
        if (kind == 2) sd::prep_stack<2>(pa, b); else sd::prep_stack<1>(pa, b);
    }
    const size_t fs = (size_t)Lmax * B;
    if (const char *e = getenv("SD_PERT_FIELD")) {       // sensitivity probe: scale one SoA field
        const int f = atoi(e); const float eps = (float)atof(getenv("SD_PERT_EPS"));
        for (size_t i = 0; i < fs; ++i) mdl[f * fs + i] *= (1.0f + eps);
    }
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < P; ++k) {
            const float cc = c[(size_t)b * P + k];
            float ug = 0.0f;
            if (nl[b] >= 2 && cc > 0.0f) {
                if (kind == 2) ug = sd::group_rayleigh(mdl.data(), fs, B, b, nl[b], per[k], cc, ratio[(size_t)b * P + k], dbg ? dbg + 16 * ((size_t)b * P + k) : nullptr);
                else           ug = sd::group_love(mdl.data(), fs, B, b, nl[b], per[k], cc);
            }
            u[(size_t)b * P + k] = ug;
        }
    return 0;
}
