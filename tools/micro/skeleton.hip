// Diagnostic: the memory skeleton of one fused step at the bench size (N = 64 x 4096 envs), no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/skeleton.hip -o build/skeleton && build/skeleton
// Every lane reads what k_step<set_target_vel> reads (13 float64 state planes, 15 float32 controller planes, a 12-byte
// action) and writes what it writes (the same planes, 4 rotor planes, a 48-byte observation row, one adjacency word), in
// the same lock-step order: all loads, a dependent reduction, all stores.  Variants: plane-major (the library's
// layout), tile-major (all fields of 64 agents contiguous), loads only, stores only, and a spin of dependent VALU work
// between the loads and the stores (does the memory time hide behind compute of other waves?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NS = 13, NP = 15;
struct Args {
    double *st; float *pid; const float *act; float *rpm; float *obs; unsigned long long *adj;
    size_t T; int spin; int mode; // mode bit0: loads, bit1: stores, bit2: tile-major
    int what; // stores/loads to include: 1 state, 2 controller planes, 4 rotor planes, 8 observation rows, 16 adjacency word, 32 action; 64: observation rows through LDS (coalesced 16-byte stores)
};
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_skel(const Args A)
{
    const size_t a = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t T = A.T;
    const bool tile = A.mode & 4;
    const size_t wave = a >> 6, lane = a & 63;
    double s[NS]; float g[NP]; float ac[3] = {0, 0, 0};
    if (A.mode & 1) {
        if (A.what & 32) { ac[0] = A.act[a * 3]; ac[1] = A.act[a * 3 + 1]; ac[2] = A.act[a * 3 + 2]; }
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = !(A.what & 1) ? 0.0 : tile ? A.st[(wave * NS + k) * 64 + lane] : A.st[k * T + a];
#pragma unroll
        for (int k = 0; k < NP; ++k) g[k] = !(A.what & 2) ? 0.f : tile ? A.pid[(wave * NP + k) * 64 + lane] : A.pid[k * T + a];
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = (double)a;
#pragma unroll
        for (int k = 0; k < NP; ++k) g[k] = (float)a;
    }
    double acc = ac[0] + ac[1] + ac[2];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc += s[k];
#pragma unroll
    for (int k = 0; k < NP; ++k) acc += g[k];
    for (int it = 0; it < A.spin; ++it) acc = __builtin_fma(acc, 0.999999, 1e-9); // dependent float64 chain: ~spin x 8 cycles alone
    if (A.mode & 2) {
        if (A.what & 1) {
#pragma unroll
            for (int k = 0; k < NS; ++k) { const double v = s[k] + acc * 1e-30; if (tile) A.st[(wave * NS + k) * 64 + lane] = v; else A.st[k * T + a] = v; }
        }
        if (A.what & 2) {
#pragma unroll
            for (int k = 0; k < NP; ++k) { const float v = g[k] + (float)acc * 1e-30f; if (tile) A.pid[(wave * NP + k) * 64 + lane] = v; else A.pid[k * T + a] = v; }
        }
        if (A.what & 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) A.rpm[k * T + a] = (float)acc;
        }
        if (A.what & 8) {
#pragma unroll
            for (int k = 0; k < 12; ++k) A.obs[a * 12 + k] = (float)s[k];
        }
        if (A.what & 64) { // the wave's 64 rows of 12 floats = 3 KB contiguous: through LDS, three 16-byte stores per lane
            __shared__ float stage[BLOCK * 12];
            float *w = stage + (threadIdx.x & ~63) * 12;
#pragma unroll
            for (int k = 0; k < 12; ++k) w[lane * 12 + k] = (float)s[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            float4 *dst = reinterpret_cast<float4 *>(A.obs + (a - lane) * 12);
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[k * 64 + lane] = reinterpret_cast<float4 *>(w)[k * 64 + lane];
        }
        if (A.what & 16) A.adj[a] = (unsigned long long)acc;
    } else if (acc == 1.2345) A.adj[a] = 1;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t T = 4096 * 64;
    Args A{};
    A.T = T;
    CK(hipMalloc(&A.st, NS * T * 8)); CK(hipMalloc(&A.pid, NP * T * 4)); CK(hipMalloc((void **)&A.act, T * 12));
    CK(hipMalloc(&A.rpm, 4 * T * 4)); CK(hipMalloc(&A.obs, T * 48)); CK(hipMalloc(&A.adj, T * 8));
    CK(hipMemset(A.st, 0, NS * T * 8)); CK(hipMemset(A.pid, 0, NP * T * 4)); CK(hipMemset((void *)A.act, 0, T * 12));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double rb = NS * 8 + NP * 4 + 12, wbytes = NS * 8 + NP * 4 + 16 + 48 + 8;
    printf("per agent: read %.0f B, write %.0f B; total %.1f MB per launch\n", rb, wbytes, (rb + wbytes) * T / 1e6);
    struct V { const char *name; int mode, spin, block, what; };
    const V vs[] = {{"plane-major loads+stores      ", 3, 0, 512, 63}, {"plane-major loads only        ", 1, 0, 512, 63}, {"plane-major stores only       ", 2, 0, 512, 63},
                    {"tile-major  loads+stores      ", 7, 0, 512, 63}, {"tile-major  loads only        ", 5, 0, 512, 63}, {"tile-major  stores only       ", 6, 0, 512, 63},
                    {"plane-major l+s, 256 threads  ", 3, 0, 256, 63}, {"no memory, spin 500           ", 0, 500, 512, 63}, {"no memory, spin 2000          ", 0, 2000, 512, 63},
                    {"plane-major l+s + spin 500    ", 3, 500, 512, 63}, {"plane-major l+s + spin 2000   ", 3, 2000, 512, 63}, {"tile-major  l+s + spin 2000   ", 7, 2000, 512, 63},
                    {"stores: state 104 B           ", 2, 0, 512, 1}, {"stores: controller 60 B       ", 2, 0, 512, 2}, {"stores: rotor 16 B            ", 2, 0, 512, 4},
                    {"stores: obs rows 48 B         ", 2, 0, 512, 8}, {"stores: obs rows via LDS 48 B ", 2, 0, 512, 64}, {"stores: adjacency 8 B         ", 2, 0, 512, 16},
                    {"stores: all but rotor         ", 2, 0, 512, 27}, {"stores: all, obs via LDS      ", 2, 0, 512, 87},
                    {"loads: state 104 B            ", 1, 0, 512, 1}, {"loads: controller 60 B        ", 1, 0, 512, 2}, {"loads: action 12 B            ", 1, 0, 512, 32},
                    {"l+s, obs via LDS              ", 3, 0, 512, 119}, {"l+s, obs via LDS, no rotor    ", 3, 0, 512, 115}};
    for (const V &v : vs) {
        A.mode = v.mode; A.spin = v.spin; A.what = v.what;
        const int grid = (int)(T / v.block);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 50; ++i) { if (v.block == 512) k_skel<512><<<grid, 512>>>(A); else k_skel<256><<<grid, 256>>>(A); }
            CK(hipEventRecord(e0));
            for (int i = 0; i < 500; ++i) { if (v.block == 512) k_skel<512><<<grid, 512>>>(A); else k_skel<256><<<grid, 256>>>(A); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double us = best / 500 * 1e3;
        const int w = v.what;
        const double rbytes = ((w & 1) ? 104 : 0) + ((w & 2) ? 60 : 0) + ((w & 32) ? 12 : 0);
        const double wb2 = ((w & 1) ? 104 : 0) + ((w & 2) ? 60 : 0) + ((w & 4) ? 16 : 0) + ((w & 72) ? 48 : 0) + ((w & 16) ? 8 : 0);
        const double bytes = ((v.mode & 1) ? rbytes : 0) + ((v.mode & 2) ? wb2 : 0);
        printf("%s %7.2f us per launch  %6.2f TB/s\n", v.name, us, bytes * T / us / 1e6);
    }
    return 0;
}
