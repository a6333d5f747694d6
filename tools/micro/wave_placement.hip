// Diagnostic: which SIMD does wave k of a 256-thread workgroup land on, and which workgroups share a CU,
// for a 1024-workgroup launch shaped like the fused step kernel (36 KB LDS, 4 workgroups per CU)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 4) void k_probe(unsigned *out, int spin)
{
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    double acc = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 1e-9; // keep the workgroup resident for a while
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (acc == 12345.678) out[0] = 0;
}
int main()
{
    const int G = 1024;
    unsigned *d; hipMalloc(&d, G * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 36872, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    std::map<unsigned, std::vector<int>> per_cu;
    int hist[4][4] = {};
    for (int b = 0; b < G; ++b) {
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            if (w == 0) per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        }
    }
    printf("wave index -> SIMD histogram over %d workgroups:\n", G);
    for (int w = 0; w < 4; ++w) printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct CUs used: %zu\n", per_cu.size());
    int shown = 0;
    for (auto &kv : per_cu) {
        if (shown++ >= 6) break;
        printf("  cu key %05x: workgroups", kv.first);
        for (int b : kv.second) printf(" %d(w0 simd %u)", b, (h[(b * 4) * 2] >> 4) & 3);
        printf("\n");
    }
    return 0;
}
