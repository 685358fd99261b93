// scripts/microbench/issue_rate.hip -- development microbenchmark: what VALU issue rate does the
// Rayleigh layer recursion (delta_rayleigh of surfdisp_kernels.hip, the production form, unchanged) sustain on gfx950 when
// nothing else is in the way -- no state machine, no root search, every wavefront busy for the whole
// launch?  Modes: 0 = every lane the same trial velocity (no divergence), 1 = lanes spread over
// 3.0..4.4 km/s like the teams of the real kernel (evanescent / oscillatory S mixes inside a wave),
// 2 = mode 1 with two independent evaluations per lane and iteration.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I include -I pysurfinv_amd/csrc \
//         scripts/microbench/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate
// Read the instruction count with:  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE -- gpurun_out/issue_rate
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pysurfinv_amd/csrc/surfdisp_kernels.hip"

using namespace sd;

template <int MODE>
__global__ __launch_bounds__(256) void issue_kernel(float *out, int iters, int L, float T)
{
    extern __shared__ float w_lds[];
    constexpr int S = 64;
    const int tid = threadIdx.x, slot = tid / 4, j = tid % 4;
    float *wq = w_lds + slot;
    const int LS = lds_ls(64);
    for (int i = j; i < L; i += 4) {                       // a 10-layer crust/mantle stack, Vs 3.0 -> 4.6
        const float z = (float)i / (float)(L - 1);
        const float b = 3.0f + 1.6f * z, a = 1.76f * b, rho = 0.541f + 0.3601f * a, d = 200.0f / L;
        const float zp = (float)(i > 0 ? i - 1 : 0) / (float)(L - 1), rho_up = 0.541f + 0.3601f * 1.76f * (3.0f + 1.6f * zp);
        W_IR(i) = (i == 0) ? 1.0f / rho : rho_up / rho; W_B(i) = b; W_R(i) = rho; W_D(i) = d; W_IA2(i) = 1.0f / (a * a); W_IB2(i) = 1.0f / (b * b);
    }
    __syncthreads();
    float c = (MODE == 0) ? 3.456f : 3.0f + 1.4f * (float)(slot % 16) / 16.0f + 0.01f * j;
    float acc = 0.0f, phi_ = 0.0f;
    if (MODE == 2) {
        // two independent evaluations per lane and iteration (instruction-level parallelism 2)
        float c2 = c + 0.005f;
        for (int it = 0; it < iters / 2; ++it) {
            const float v = delta_rayleigh(wq, LS, S, L, c, T, 1, phi_);
            const float w = delta_rayleigh(wq, LS, S, L, c2, T, 1, phi_);
            acc += v + w;
            c += (v > 1e30f) ? 1e-3f : 0.0f;
            c2 += (w > 1e30f) ? 1e-3f : 0.0f;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const float v = delta_rayleigh(wq, LS, S, L, c, T, 1, phi_);
            acc += v;
            c += (v > 1e30f) ? 1e-3f : 0.0f;                  // keeps the loop from being hoisted
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024;      // 1024 = 4 wavefronts per SIMD
    const int iters = 2000, L = 10;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)lds_ls(64) * L * sizeof(float);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(issue_kernel<0>, dim3(blocks), dim3(256), lds, 0, out, iters, L, 20.0f);
            else if (mode == 1) hipLaunchKernelGGL(issue_kernel<1>, dim3(blocks), dim3(256), lds, 0, out, iters, L, 20.0f);
            else           hipLaunchKernelGGL(issue_kernel<2>, dim3(blocks), dim3(256), lds, 0, out, iters, L, 20.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double evals = (double)blocks * 256 * iters;
            printf("mode %d rep %d: %.3f ms, %.3e lane-evaluations/s, %.3e layer-steps/s/lane-slot\n", mode, rep, ms,
                   evals / (ms * 1e-3), evals * L / (ms * 1e-3));
        }
    }
    std::vector<float> h(16);
    hipMemcpy(h.data(), out, 16 * sizeof(float), hipMemcpyDeviceToHost);
    printf("check %g %g\n", h[0], h[5]);
    return 0;
}
