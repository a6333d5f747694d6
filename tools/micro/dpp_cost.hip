// Diagnostic: SIMD cost of the DPP lane-movement forms inside a chain of float32 adds (the N = 64 pair loops hand a travelling
// accumulator to the next lane with wave_ror:1 once per pair): cycles per instruction at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dpp_cost.hip -o build/micro/dpp_cost && build/micro/dpp_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 2000
template <int CTRL>
__device__ __forceinline__ float mv(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
// MODE 0: plain adds; 1: wave_ror:1 (0x13C); 2: row_ror:1 (0x121); 3: quad_perm [1,2,3,0] (0x39); 4: wave_shr:1 (0x138); 5: row_bcast15 (0x142)
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, float seed)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
    const float y = seed * 0.5f;
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] + y;
            else if (MODE == 1) a[i] = mv<0x13C>(a[i]) + y;
            else if (MODE == 2) a[i] = mv<0x121>(a[i]) + y;
            else if (MODE == 3) a[i] = mv<0x39>(a[i]) + y;
            else if (MODE == 4) a[i] = mv<0x138>(a[i]) + y;
            else a[i] = mv<0x142>(a[i]) + y;
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int MODE>
void run(const char *name, long long *d, float *sink)
{
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {           // waves per SIMD: 256 CUs x 4 SIMDs
        const int blocks = 256 * wps;       // 256-thread workgroups = 4 waves = one per SIMD of a CU
        k<MODE><<<blocks, 256>>>(d, sink, 1.0f);
        CK(hipDeviceSynchronize());
        k<MODE><<<blocks, 256>>>(d, sink, 1.0f);
        CK(hipDeviceSynchronize());
        static long long h[4096];
        CK(hipMemcpy(h, d, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
        double m = 0;
        for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
        m /= blocks * 4;
        printf("  %d w/SIMD: %6.2f cyc/instr/wave -> %5.2f per SIMD", wps, m / (ITER * 8.0), m / (ITER * 8.0) / wps);
    }
    printf("\n");
}
int main()
{
    long long *d; float *sink;
    CK(hipMalloc(&d, sizeof(long long) * 4096)); CK(hipMalloc(&sink, sizeof(float) * 256 * 1024));
    run<0>("v_add_f32", d, sink);
    run<1>("v_add_f32 dpp wave_ror:1", d, sink);
    run<2>("v_add_f32 dpp row_ror:1", d, sink);
    run<3>("v_add_f32 dpp quad_perm", d, sink);
    run<4>("v_add_f32 dpp wave_shr:1", d, sink);
    run<5>("v_add_f32 dpp row_bcast15", d, sink);
    return 0;
}
