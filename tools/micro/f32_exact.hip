// Diagnostic: are the float32 helpers of mrs_device.hpp (div, sqrt, mul, add, fma) correctly rounded on gfx950?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "../../mrs-gym_amd/csrc/mrs_device.hpp"
using namespace mrs;
__global__ void k(const float *a, const float *b, float *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[i] = f32div(a[i], b[i]);
    o[n + i] = f32sqrt(fabsf(a[i]));
    o[2 * n + i] = f32add(f32mul(a[i], b[i]), a[i]);
    o[3 * n + i] = f32div(a[i], f32mul(f32mul(b[i], b[i]), b[i]));
}
int main()
{
    const int n = 1 << 20;
    float *ha = (float *)malloc(n * 4), *hb = (float *)malloc(n * 4), *ho = (float *)malloc(4 * n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) { ha[i] = (rand() / (float)RAND_MAX - 0.5f) * 8.f; hb[i] = (rand() / (float)RAND_MAX) * 5.f + 1e-3f; }
    float *a, *b, *o; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, 4 * n * 4);
    hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, o, n);
    hipMemcpy(ho, o, 4 * n * 4, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        volatile float m = ha[i] * hb[i];
        volatile float c = hb[i] * hb[i]; volatile float c3 = c * hb[i];
        bad[0] += ho[i] != ha[i] / hb[i];
        bad[1] += ho[n + i] != sqrtf(fabsf(ha[i]));
        bad[2] += ho[2 * n + i] != (float)(m + ha[i]);
        bad[3] += ho[3 * n + i] != ha[i] / c3;
    }
    printf("mismatches of %d: div %d, sqrt %d, mul+add %d, div by cube %d\n", n, bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
