// Diagnostic: cost of one dependent kernel launch in a stream (the floor under every kernel of mrs_step).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double pad[96]; int *count; };
__global__ void k_empty(int *count) { if (threadIdx.x == 0 && blockIdx.x == 0 && count == nullptr) printf("x"); }
__global__ void k_count(Big A) { const int c = *A.count; if ((int)(blockIdx.x * blockDim.x + threadIdx.x) < c) A.count[1] = 1; }
__global__ void k_touch(double *x, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) x[i] += 1.0; }
template <class F> static float timeit(F f, int n, hipStream_t s)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / n;
}
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    int *cnt; hipMalloc(&cnt, 64); hipMemset(cnt, 0, 64);
    double *x; size_t n = 262144; hipMalloc(&x, n * 13 * 8); hipMemset(x, 0, n * 13 * 8);
    Big B; B.count = cnt;
    int grids[] = {1, 64, 256, 1024, 4096};
    for (int g : grids) {
        printf("empty   <<<%4d,256>>> %.2f us\n", g, timeit([&] { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s, cnt); }, 2000, s));
        printf("count   <<<%4d,256>>> %.2f us (800-byte args, one global read, exit)\n", g, timeit([&] { hipLaunchKernelGGL(k_count, dim3(g), dim3(256), 0, s, B); }, 2000, s));
    }
    printf("touch 1 plane  <<<1024,256>>> %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, x, n); }, 2000, s));
    printf("touch 13 planes<<<13312,256>>> %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_touch, dim3(13312), dim3(256), 0, s, x, n * 13); }, 2000, s));
    return 0;
}
