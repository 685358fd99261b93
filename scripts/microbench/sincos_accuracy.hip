// scripts/microbench/sincos_accuracy.hip -- development: absolute error of the hardware v_sin_f32 / v_cos_f32 (argument in
// revolutions: x / 2pi, reduced with v_fract_f32) against the root search's 3-constant Cody-Waite sincos_cw, both against
// fp64 sin / cos, over the arguments the recursion sees (k d r: 0 .. 40 rad).
//   hipcc -O3 --offload-arch=gfx950 -I include -I pysurfinv_amd/csrc scripts/microbench/sincos_accuracy.hip -o scripts/microbench/bin/sincos_accuracy
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../pysurfinv_amd/csrc/surfdisp_kernels.hip"
using namespace sd;
__global__ void k(const float *x, float *o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    sincos_cw(x[i], &s, &c);
    const float r = x[i] * 0.15915494309189535f;
    const float f = r - floorf(r);
    o[4 * i] = s; o[4 * i + 1] = c; o[4 * i + 2] = __builtin_amdgcn_sinf(f); o[4 * i + 3] = __builtin_amdgcn_cosf(f);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n), o(4 * (size_t)n);
    for (int i = 0; i < n; ++i) h[i] = 40.0f * (float)i / n * ((i & 1) ? -1.0f : 1.0f);
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)n * 16);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    for (float lim : {3.2f, 10.0f, 40.0f}) {
        double e[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            if (fabsf(h[i]) > lim) continue;
            const double s = sin((double)h[i]), c = cos((double)h[i]);
            e[0] = fmax(e[0], fabs(o[4 * (size_t)i] - s)); e[1] = fmax(e[1], fabs(o[4 * (size_t)i + 1] - c));
            e[2] = fmax(e[2], fabs(o[4 * (size_t)i + 2] - s)); e[3] = fmax(e[3], fabs(o[4 * (size_t)i + 3] - c));
        }
        printf("|x| <= %4.1f: max abs error  sincos_cw sin %.2e cos %.2e | v_sin_f32 %.2e v_cos_f32 %.2e\n", lim, e[0], e[1], e[2], e[3]);
    }
    return 0;
}
