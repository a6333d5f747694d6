// Diagnostic: does the memory skeleton of one fused step (tools/micro/skeleton.hip) get faster with 16-byte per-lane
// accesses?  Same bytes, same grid (N = 64 x 4096 envs, 512-thread workgroups), same lock-step order; only the width of
// the per-lane access differs:
//   narrow : the library's round-2 layout -- 13 float64 planes (8 B per lane and instruction), 15 float32 planes (4 B),
//            a 12-byte action as three 4-byte loads, the 24-byte observation row as three 8-byte stores
//   wide   : the same words paired into 16-byte records -- 6 double2 planes + 1 double plane, 3 float4 planes + 1 float3
//            (as float4: 64 B per agent instead of 60), the action as one 12-byte load, the observation row through LDS as
//            16-byte stores
//   hipcc --offload-arch=gfx950 -O3 tools/micro/skeleton_wide.hip -o build/skeleton_wide && build/skeleton_wide
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    double *st; float *pid; const float *act; float *obs; unsigned long long *adj;
    size_t T; int spin; int mode; // bit0 loads, bit1 stores
};
constexpr int BLOCK = 512;

__global__ __launch_bounds__(BLOCK) void k_narrow(const Args A)
{
    const size_t a = (size_t)blockIdx.x * BLOCK + threadIdx.x, T = A.T;
    double s[13]; float g[15]; float ac[3] = {0, 0, 0};
    if (A.mode & 1) {
        ac[0] = A.act[a * 3]; ac[1] = A.act[a * 3 + 1]; ac[2] = A.act[a * 3 + 2];
#pragma unroll
        for (int k = 0; k < 13; ++k) s[k] = A.st[k * T + a];
#pragma unroll
        for (int k = 0; k < 15; ++k) g[k] = A.pid[k * T + a];
    } else {
#pragma unroll
        for (int k = 0; k < 13; ++k) s[k] = (double)a;
#pragma unroll
        for (int k = 0; k < 15; ++k) g[k] = (float)a;
    }
    double acc = ac[0] + ac[1] + ac[2];
#pragma unroll
    for (int k = 0; k < 13; ++k) acc += s[k];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc += g[k];
    for (int it = 0; it < A.spin; ++it) acc = __builtin_fma(acc, 0.999999, 1e-9);
    if (A.mode & 2) {
#pragma unroll
        for (int k = 0; k < 13; ++k) A.st[k * T + a] = s[k] + acc * 1e-30;
#pragma unroll
        for (int k = 0; k < 15; ++k) A.pid[k * T + a] = g[k] + (float)acc * 1e-30f;
        float2 *o2 = reinterpret_cast<float2 *>(A.obs + a * 6);
        o2[0] = make_float2((float)s[0], (float)s[1]); o2[1] = make_float2((float)s[2], (float)s[3]); o2[2] = make_float2((float)s[4], (float)s[5]);
        A.adj[a] = (unsigned long long)acc;
    } else if (acc == 1.2345) A.adj[a] = 1;
}

// OBS: 0 = three 8-byte stores per lane, 1 = through LDS as 16-byte stores (the wave's 64 rows of 24 B = 1536 B = 96 x 16 B)
template <int OBS>
__global__ __launch_bounds__(BLOCK) void k_wide(const Args A)
{
    const size_t a = (size_t)blockIdx.x * BLOCK + threadIdx.x, T = A.T;
    const size_t lane = a & 63;
    double2 s2[6]; double s1; float4 g4[4]; float ac[3] = {0, 0, 0};
    const double2 *st2 = reinterpret_cast<const double2 *>(A.st);          // [6][T] double2, then [T] double
    const double *st1 = A.st + 12 * T;
    const float4 *p4 = reinterpret_cast<const float4 *>(A.pid);            // [4][T] float4
    if (A.mode & 1) {
        const float *ap = A.act + a * 3;
        ac[0] = ap[0]; ac[1] = ap[1]; ac[2] = ap[2];                        // the compiler merges these into one dwordx3
#pragma unroll
        for (int k = 0; k < 6; ++k) s2[k] = st2[k * T + a];
        s1 = st1[a];
#pragma unroll
        for (int k = 0; k < 4; ++k) g4[k] = p4[k * T + a];
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) s2[k] = make_double2((double)a, (double)a);
        s1 = (double)a;
#pragma unroll
        for (int k = 0; k < 4; ++k) g4[k] = make_float4((float)a, (float)a, (float)a, (float)a);
    }
    double acc = ac[0] + ac[1] + ac[2] + s1;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += s2[k].x + s2[k].y;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += g4[k].x + g4[k].y + g4[k].z + g4[k].w;
    for (int it = 0; it < A.spin; ++it) acc = __builtin_fma(acc, 0.999999, 1e-9);
    if (A.mode & 2) {
        double2 *w2 = reinterpret_cast<double2 *>(A.st);
        const double e = acc * 1e-30;
#pragma unroll
        for (int k = 0; k < 6; ++k) w2[k * T + a] = make_double2(s2[k].x + e, s2[k].y + e);
        (A.st + 12 * T)[a] = s1 + e;
        float4 *q4 = reinterpret_cast<float4 *>(A.pid);
        const float f = (float)acc * 1e-30f;
#pragma unroll
        for (int k = 0; k < 4; ++k) q4[k * T + a] = make_float4(g4[k].x + f, g4[k].y + f, g4[k].z + f, g4[k].w + f);
        if (OBS == 0) {
            float2 *o2 = reinterpret_cast<float2 *>(A.obs + a * 6);
            o2[0] = make_float2((float)s2[0].x, (float)s2[0].y); o2[1] = make_float2((float)s2[1].x, (float)s2[1].y); o2[2] = make_float2((float)s2[2].x, (float)s2[2].y);
        } else {
            __shared__ float stage[BLOCK * 6];
            float *w = stage + (threadIdx.x & ~63) * 6;
            float2 *wl = reinterpret_cast<float2 *>(w + lane * 6);
            wl[0] = make_float2((float)s2[0].x, (float)s2[0].y); wl[1] = make_float2((float)s2[1].x, (float)s2[1].y); wl[2] = make_float2((float)s2[2].x, (float)s2[2].y);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float4 *dst = reinterpret_cast<float4 *>(A.obs + (a - lane) * 6);
            const float4 *src = reinterpret_cast<const float4 *>(w);
            dst[lane] = src[lane];
            if (lane < 32) dst[64 + lane] = src[64 + lane];
        }
        A.adj[a] = (unsigned long long)acc;
    } else if (acc == 1.2345) A.adj[a] = 1;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t T = 4096 * 64;
    Args A{};
    A.T = T;
    CK(hipMalloc(&A.st, 13 * T * 8)); CK(hipMalloc(&A.pid, 16 * T * 4)); CK(hipMalloc((void **)&A.act, T * 12));
    CK(hipMalloc(&A.obs, T * 24)); CK(hipMalloc(&A.adj, T * 8));
    CK(hipMemset(A.st, 0, 13 * T * 8)); CK(hipMemset(A.pid, 0, 16 * T * 4)); CK(hipMemset((void *)A.act, 0, T * 12));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char *name; int kern, mode, spin; };
    const V vs[] = {{"narrow loads+stores       ", 0, 3, 0}, {"narrow loads only         ", 0, 1, 0}, {"narrow stores only        ", 0, 2, 0},
                    {"wide   loads+stores       ", 1, 3, 0}, {"wide   loads only         ", 1, 1, 0}, {"wide   stores only        ", 1, 2, 0},
                    {"wide   l+s, obs via LDS   ", 2, 3, 0}, {"wide   stores, obs via LDS", 2, 2, 0},
                    {"no memory, spin 500       ", 0, 0, 500},
                    {"narrow l+s + spin 500     ", 0, 3, 500}, {"wide   l+s + spin 500     ", 1, 3, 500}, {"wide/LDS l+s + spin 500   ", 2, 3, 500},
                    {"narrow l+s + spin 1000    ", 0, 3, 1000}, {"wide   l+s + spin 1000    ", 1, 3, 1000}};
    const int grid = (int)(T / BLOCK);
    for (const V &v : vs) {
        A.mode = v.mode; A.spin = v.spin;
        auto launch = [&]() {
            if (v.kern == 0) k_narrow<<<grid, BLOCK>>>(A);
            else if (v.kern == 1) k_wide<0><<<grid, BLOCK>>>(A);
            else k_wide<1><<<grid, BLOCK>>>(A);
        };
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 50; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 500; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double us = best / 500 * 1e3;
        const double pidb = v.kern == 0 ? 60 : 64;
        const double bytes = ((v.mode & 1) ? 104 + pidb + 12 : 0) + ((v.mode & 2) ? 104 + pidb + 24 + 8 : 0);
        printf("%s %7.2f us per launch  %6.2f TB/s\n", v.name, us, bytes * T / us / 1e6);
    }
    return 0;
}
