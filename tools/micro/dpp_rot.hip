// Diagnostic: full-wave lane rotation by one through DPP (wave_ror:1 / wave_rol:1, a VALU instruction) against ds_bpermute_b32
// (an LDS-pipe instruction): which lane receives from which, and the cost of a chain of 32 hand-overs per wave with every
// CU holding 16 waves (the step kernel's occupancy).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dpp_rot.hip -o build/dpp_rot && build/dpp_rot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_map(int *ror, int *rol)
{
    const int v = threadIdx.x;
    ror[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);
    rol[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);
}

// MODE 0: 32 x (compare-ish VALU work + v_or_b32 with the rotated word); MODE 1: the same with ds_bpermute hand-overs to lane + k
template <int MODE>
__global__ __launch_bounds__(512) void k_chain(unsigned *out, const float *in, int reps)
{
    const int lane = threadIdx.x & 63;
    float x = in[blockIdx.x * 512 + threadIdx.x];
    unsigned w = 0, lo = 0;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int k = 31; k >= 1; --k) {
            x = __builtin_fmaf(x, 1.0001f, 0.25f);
            const unsigned bit = x > 3.0f ? (1u << k) : 0u;
            lo |= bit;
            if (MODE == 0) w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x13C, 0xf, 0xf, false) | bit;
            else w |= (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 64 - k) & 63) << 2, (int)bit);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = w ^ lo;
}

int main()
{
    int *d; CK(hipMalloc(&d, 128 * 4));
    k_map<<<1, 64>>>(d, d + 64);
    int h[128]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("wave_ror:1  lane 0 <- %d, lane 1 <- %d, lane 16 <- %d, lane 32 <- %d, lane 63 <- %d\n", h[0], h[1], h[16], h[32], h[63]);
    printf("wave_rol:1  lane 0 <- %d, lane 1 <- %d, lane 15 <- %d, lane 31 <- %d, lane 63 <- %d\n", h[64], h[65], h[64 + 15], h[64 + 31], h[127]);
    const int G = 512;
    float *in; unsigned *out; CK(hipMalloc(&in, G * 512 * 4)); CK(hipMalloc(&out, G * 512 * 4)); CK(hipMemset(in, 0, G * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int reps : {1, 8}) {
            float best = 1e9f;
            for (int t = 0; t < 3; ++t) {
                for (int i = 0; i < 20; ++i) { if (mode == 0) k_chain<0><<<G, 512>>>(out, in, reps); else k_chain<1><<<G, 512>>>(out, in, reps); }
                CK(hipEventRecord(e0));
                for (int i = 0; i < 200; ++i) { if (mode == 0) k_chain<0><<<G, 512>>>(out, in, reps); else k_chain<1><<<G, 512>>>(out, in, reps); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%s  %d x 31 hand-overs per wave: %.2f us per launch\n", mode == 0 ? "DPP wave_ror " : "ds_bpermute  ", reps, best / 200 * 1e3);
        }
    }
    return 0;
}
