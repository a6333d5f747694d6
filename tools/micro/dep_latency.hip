// Diagnostic: what a DEPENDENT chain costs a lone wave against the same instructions issued independently (8 accumulators), one wave
// per SIMD and four (cycles per instruction per wave).  The step kernel is bound by its waves' dependency chains, not by issue
// slots (tools/probes/timeline_simd.py at E = 1024: a lone wave needs 12 us for its 3000 instructions); this says what a link costs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dep_latency.hip -o build/micro/dep_latency && build/micro/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 1000
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float mv(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
template <int KIND> struct Op;
template <> struct Op<0> { typedef float T; static __device__ T f(T a, T y, T z) { return __builtin_fmaf(a, z, y); } };
template <> struct Op<1> { typedef double T; static __device__ T f(T a, T y, T z) { return __builtin_fma(a, z, y); } };
template <> struct Op<2> { typedef f2 T; static __device__ T f(T a, T y, T z) { return __builtin_elementwise_fma(a, z, y); } };
template <> struct Op<3> { typedef float T; static __device__ T f(T a, T y, T z) { return __builtin_amdgcn_rcpf(a); } };
template <> struct Op<4> { typedef float T; static __device__ T f(T a, T y, T z) { return __builtin_amdgcn_exp2f(a); } };
template <> struct Op<5> { typedef float T; static __device__ T f(T a, T y, T z) { return __builtin_fmaxf(a, y); } };
template <> struct Op<6> { typedef float T; static __device__ T f(T a, T y, T z) { return mv<0x13C>(a) + y; } };
template <> struct Op<7> { typedef float T; static __device__ T f(T a, T y, T z) { return a > y ? z : a; } };
template <> struct Op<8> { typedef double T; static __device__ T f(T a, T y, T z) { return (double)(float)a; } };
template <> struct Op<9> { typedef double T; static __device__ T f(T a, T y, T z) { return a + y; } };
template <class T> __device__ T mk(float s) { return (T)s; }
template <> __device__ f2 mk<f2>(float s) { return f2{s, s + 1}; }
template <class T> __device__ float sum(T v) { return (float)v; }
template <> __device__ float sum<f2>(f2 v) { return v.x + v.y; }
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(long long *out, float *sink, float seed)
{
    typedef typename Op<KIND>::T T;
    T a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = mk<T>(seed * (i + 1) + threadIdx.x * 1e-3f);
    const T y = mk<T>(seed * 0.5f), z = mk<T>(0.999f);
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) a[i] = Op<KIND>::f(a[i], y, z);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += sum(a[i]);
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND, int NACC>
void run(const char *name, long long *d, float *sink)
{
    printf("%-40s", name);
    for (int wps : {1, 4}) {
        const int blocks = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) { k<KIND, NACC><<<blocks, 256>>>(d, sink, 1.0f); CK(hipDeviceSynchronize()); }
        static long long h[4096];
        CK(hipMemcpy(h, d, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
        double m = 0;
        for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
        m /= blocks * 4;
        printf("  %d w/SIMD: %6.2f cyc/op/wave", wps, m / (ITER * 8.0));
    }
    printf("\n");
}
#define BOTH(K, NAME) run<K, 1>(NAME ", dependent chain", d, sink); run<K, 8>(NAME ", 8 independent", d, sink);
int main()
{
    long long *d; float *sink;
    CK(hipMalloc(&d, sizeof(long long) * 4096)); CK(hipMalloc(&sink, sizeof(float) * 256 * 1024));
    BOTH(0, "v_fma_f32") BOTH(5, "v_max_f32") BOTH(7, "v_cmp + v_cndmask") BOTH(2, "v_pk_fma_f32") BOTH(6, "v_add_f32 dpp wave_ror:1") BOTH(3, "v_rcp_f32") BOTH(4, "v_exp_f32")
    BOTH(1, "v_fma_f64") BOTH(9, "v_add_f64") BOTH(8, "cvt f64 -> f32 -> f64")
    return 0;
}
