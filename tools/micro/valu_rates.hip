// Diagnostic: SIMD issue cost of the vector instructions k_step is made of, at 1 / 2 / 4 resident waves per SIMD.
// Each kernel runs ITER x 16 independent copies of one instruction per wave (8 destination registers, no
// dependency closer than 8 instructions) and reports shader cycles (s_memtime) per wave-instruction per SIMD:
//   cycles = (wave lifetime in ticks) / (ITER * 16 * waves_per_simd)
// i.e. what one more instruction of that kind costs a SIMD that is kept busy by `waves_per_simd` waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// KERNEL(name, decl, body8): body8(k) emits one instruction on accumulator k
#define KERNEL(name, SETUP, BODY)                                                                   \
    __global__ __launch_bounds__(256) void name(long long *out, float seed)                          \
    {                                                                                                \
        SETUP                                                                                        \
        const long long t0 = clock64();                                                              \
        for (int it = 0; it < ITER; ++it) {                                                          \
            REP8(BODY) REP8(BODY)                                                                    \
        }                                                                                            \
        const long long t1 = clock64();                                                              \
        FINISH                                                                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

#define F32SETUP float a[8], b = seed + threadIdx.x * 1e-7f, c = 0.999f; for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1);
#define F64SETUP double a[8], b = seed + threadIdx.x * 1e-7, c = 0.999; for (int k = 0; k < 8; ++k) a[k] = seed * (k + 1);
#define FINISH { float s = 0; for (int k = 0; k < 8; ++k) s += (float)a[k]; if (s == 1234.5f) out[0] = 0; }

#define I_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
#define I_MUL32(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_ADD32(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define I_EXP32(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
#define I_SQRT32(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define I_MOV32(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b));
#define I_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
#define I_CMP32(k) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define I_CMP32S(k) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" : : "v"(a[k]), "v"(b) : "s20", "s21");
#define I_MED3(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
#define I_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
#define I_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_MAX64(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define I_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define I_RSQ64(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
#define I_CMP64(k) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");

KERNEL(k_fma32, F32SETUP, I_FMA32)
KERNEL(k_mul32, F32SETUP, I_MUL32)
KERNEL(k_add32, F32SETUP, I_ADD32)
KERNEL(k_rcp32, F32SETUP, I_RCP32)
KERNEL(k_exp32, F32SETUP, I_EXP32)
KERNEL(k_sqrt32, F32SETUP, I_SQRT32)
KERNEL(k_mov32, F32SETUP, I_MOV32)
KERNEL(k_cndmask, F32SETUP, I_CNDMASK)
KERNEL(k_cmp32, F32SETUP, I_CMP32)
KERNEL(k_cmp32s, F32SETUP, I_CMP32S)
KERNEL(k_med3, F32SETUP, I_MED3)
KERNEL(k_fma64, F64SETUP, I_FMA64)
KERNEL(k_mul64, F64SETUP, I_MUL64)
KERNEL(k_add64, F64SETUP, I_ADD64)
KERNEL(k_max64, F64SETUP, I_MAX64)
KERNEL(k_rcp64, F64SETUP, I_RCP64)
KERNEL(k_rsq64, F64SETUP, I_RSQ64)
KERNEL(k_cmp64, F64SETUP, I_CMP64)

// conversions and packed forms need mixed register classes
#undef FINISH
#define FINISH { float s = 0; for (int k = 0; k < 8; ++k) s += (float)a[k] + (float)d[k]; if (s == 1234.5f) out[0] = 0; }
#define CVTSETUP double a[8]; float d[8]; for (int k = 0; k < 8; ++k) { a[k] = seed * (k + 1); d[k] = seed * k; }
#define I_CVT3264(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d[k]) : "v"(a[k]));
#define I_CVT6432(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(d[k]));
KERNEL(k_cvt3264, CVTSETUP, I_CVT3264)
KERNEL(k_cvt6432, CVTSETUP, I_CVT6432)
// v_pk_fma_f32 on 64-bit register pairs (two float32 lanes-worth of work per instruction)
#define PKSETUP double a[8], d[8]; double b = seed, c = 0.999; for (int k = 0; k < 8; ++k) { a[k] = seed * (k + 1); d[k] = 0; }
#define I_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(b));
#define I_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#define I_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
KERNEL(k_pkfma, PKSETUP, I_PKFMA)
KERNEL(k_pkmul, PKSETUP, I_PKMUL)
KERNEL(k_pkadd, PKSETUP, I_PKADD)
// cross-lane / LDS
#define BPSETUP int a[8], d[8]; const int addr = ((threadIdx.x + 1) & 63) << 2; for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x + k; d[k] = 0; }
#define I_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(6)" : "+v"(a[k]) : "v"(addr));
#define I_DPPROT(k) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define I_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[k]) : : "vcc");
#define I_LSHLOR(k) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(addr));
KERNEL(k_bperm, BPSETUP, I_BPERM)
KERNEL(k_dpprot, BPSETUP, I_DPPROT)
KERNEL(k_addc, BPSETUP, I_ADDC)
KERNEL(k_lshlor, BPSETUP, I_LSHLOR)
// scalar ALU beside nothing (for the SALU-offload estimate)
#define SSETUP int a[8], d[8]; for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x + k; d[k] = 0; }
#define I_SALU(k) asm volatile("s_lshl_b64 s[20:21], s[20:21], 1" : : : "s20", "s21", "scc");
KERNEL(k_salu, SSETUP, I_SALU)
// LDS broadcast read of 16 bytes (what the pair loops do per neighbour)
__global__ __launch_bounds__(256) void k_ldsread(long long *out, float seed)
{
    __shared__ float4 tile[512];
    tile[threadIdx.x] = make_float4(seed, seed, seed, seed);
    tile[256 + threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    const float4 *nb = tile + (threadIdx.x & 63);
    const long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)nb), "n"(k * 16));
            asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1" : "+v"(acc.x) : "v"(v.x));
        }
    }
    const long long t1 = clock64();
    if (acc.x == 1234.5f) out[0] = 0;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(long long *, float);
struct Entry { const char *name; kern_t k; };

int main()
{
    Entry tab[] = {
        {"v_fma_f32", k_fma32}, {"v_mul_f32", k_mul32}, {"v_add_f32", k_add32}, {"v_rcp_f32", k_rcp32}, {"v_exp_f32", k_exp32},
        {"v_sqrt_f32", k_sqrt32}, {"v_mov_b32", k_mov32}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_f32 vcc", k_cmp32},
        {"v_cmp_gt_f32 sgpr", k_cmp32s}, {"v_med3_f32", k_med3},
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_max_f64", k_max64}, {"v_rcp_f64", k_rcp64},
        {"v_rsq_f64", k_rsq64}, {"v_cmp_gt_f64", k_cmp64}, {"v_cvt_f32_f64", k_cvt3264}, {"v_cvt_f64_f32", k_cvt6432},
        {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
        {"ds_bpermute_b32", k_bperm}, {"v_mov_b32_dpp ror", k_dpprot}, {"v_addc_co_u32", k_addc}, {"v_lshl_or_b32", k_lshlor},
        {"s_lshl_b64", k_salu}, {"ds_read_b128+v_add", k_ldsread},
    };
    setvbuf(stdout, nullptr, _IONBF, 0);
    long long *out;
    hipMalloc(&out, 8 * 4096 * 8);
    std::vector<long long> h(4096 * 8);
    printf("%-22s %10s %10s %10s   (shader cycles per wave-instruction per SIMD)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (auto &e : tab) {
        printf("%-22s", e.name);
        for (int wps : {1, 2, 4}) {
            const int grid = 256 * wps; // 256-thread workgroups: one wave per SIMD each; wps workgroups per CU
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 1.0f);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 1.0f);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, grid * 4 * 8, hipMemcpyDeviceToHost);
            std::vector<long long> v(h.begin(), h.begin() + grid * 4);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            printf(" %10.2f", med / (ITER * 16.0 * wps));
        }
        printf("\n");
    }
    return 0;
}
