// mrs_kernels.hip -- gfx950 kernels + C-ABI host entry points (include/mrs_hip.h).
//
// Mapping: one wavefront lane per quadcopter, agent-major.  A 256-thread workgroup owns
// floor(256/N) whole envs (N=64: 4 envs, one wave each; N=256: one env over 4 waves); N in
// (256,1024] takes one env per workgroup of roundup64(N) threads.  The env's float32 positions are
// staged once in LDS and serve both O(N^2) loops (pre-step: downwash, Quadcopter.py:99-115;
// post-step: COMM_RANGE adjacency, MRS.py:117-124) as conflict-free broadcast reads.
// Nothing here is a dense contraction: MFMA is deliberately unused; the kernel streams the SoA
// state planes once in and once out per step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "mrs_device.hpp"

using namespace mrs;

// Diagnostic knock-outs (tools/abl_build.sh "-DMRS_KO=<bits>"; results are WRONG, only the duration is looked at): what does
// the launch cost without 1: the state stores of the tail, 2: the observation + adjacency stores, 4: the controller-memory
// stores, 8: the downwash pair loop, 16: the adjacency pass.  The code stays (the store is taken iff E < 0).
#ifndef MRS_KO
#define MRS_KO 0
#endif
#if MRS_KO == -1 // chosen at run time through the environment variable MRS_KO, read by every mrs_step call (tools/steady_bench.py sets it after the roll-in)
#define KO_KEEP(bit) (!(A.ko & (bit)))
#else
#define KO_KEEP(bit) (!(MRS_KO & (bit)) || A.E < 0)
#endif

// ------------------------------------------------------------------------------------------ args
// Member order = order of first use.  The kernel reads its arguments through the scalar cache, which is cold when a
// launch starts: every 64-byte line of this struct is a memory round trip that all waves of the launch wait for before
// they can issue their first load.  The sizes and the pointers of the first loads share the first line; the physics
// constants (360 B, first needed by the controller) come last.
struct StepArgs {
    int E, N, T, epb, D, W;
#if MRS_KO == -1
    int ko;
#endif
    const float *actions;
    const uint8_t *mask;
    MrsBuffers b;
    int n_obs;
    int obs_fields[MRS_OBS_MAX_FIELDS];
    unsigned obs_code; // the same field list, 4 bits per field (one scalar instead of an array in the argument segment)
    int do_adj, comm_inf;
    float d2_thresh;
    double hclip;
    double park_z;      // bodies with their centre at or below this height go through the contact solve (needs_contact)
    // quad-quad contact (MrsParams.pair_contact, mrs_device.hpp pair_contact_term): an env's flag says that the adjacency
    // pass over the positions it is starting from saw a pair within contact range; only flagged envs look for partners
    int *pair_flag;     // [E] library workspace, or null when the model is off
    unsigned long long *pair_rows; // [T][W] library workspace: who is within contact range of whom (valid where the flag says so)
    float pair_rc2;     // (2 r + contact_threshold)^2
    float pair_r2, pair_inv_dt, pair_erp_dt; // 2 r, 1 / dt, erp / dt
    Recips rc;
    int *contact_count, *contact_count_next, *contact_list; // library workspace (MrsHandle)
    double *contact_state;                                  // [13][T] parked states, indexed by list slot
    DownwashConst dc;   // host-computed once per call
    MrsParams P;
};

typedef double vel_t; // velocity planes: float64 like Bullet's state (float32 planes were measured: -0.3 us, not worth the parity loss)
#define VELP(ptr) (ptr)
__device__ __forceinline__ void load_state(const MrsBuffers &b, size_t a, size_t T, double p[3], double q[4], double v[3], double w[3])
{
    p[0] = b.pos[a]; p[1] = b.pos[T + a]; p[2] = b.pos[2 * T + a];
    q[0] = b.quat[a]; q[1] = b.quat[T + a]; q[2] = b.quat[2 * T + a]; q[3] = b.quat[3 * T + a];
    v[0] = VELP(b.vel)[a]; v[1] = VELP(b.vel)[T + a]; v[2] = VELP(b.vel)[2 * T + a];
    w[0] = VELP(b.angvel)[a]; w[1] = VELP(b.angvel)[T + a]; w[2] = VELP(b.angvel)[2 * T + a];
}
__device__ __forceinline__ void store_state(const MrsBuffers &b, size_t a, size_t T, const double p[3], const double q[4], const double v[3], const double w[3])
{
    b.pos[a] = p[0]; b.pos[T + a] = p[1]; b.pos[2 * T + a] = p[2];
    b.quat[a] = q[0]; b.quat[T + a] = q[1]; b.quat[2 * T + a] = q[2]; b.quat[3 * T + a] = q[3];
    VELP(b.vel)[a] = (vel_t)v[0]; VELP(b.vel)[T + a] = (vel_t)v[1]; VELP(b.vel)[2 * T + a] = (vel_t)v[2];
    VELP(b.angvel)[a] = (vel_t)w[0]; VELP(b.angvel)[T + a] = (vel_t)w[1]; VELP(b.angvel)[2 * T + a] = (vel_t)w[2];
}

// (Round 3 experiment, removed: global_store ... sc1, agent-scope write-through, so that the step's stores do not sit dirty in
// the XCD's L2 until the end-of-kernel release.  State planes: -0.3 ... +0.2 us, inside the noise; the 8-byte observation /
// adjacency stores: +2 us, each becomes a fabric write; controller memory: +-0.)
template <int BIT, class T>
__device__ __forceinline__ void st(T *ptr, T v) { *ptr = v; }

// Workgroup-relative addressing.  Agent a = wg_base + tid for every live lane of k_step / k_observe_adj, so a
// plane access is (uniform pointer, SGPR pair) + (small 32-bit per-lane offset): the saddr form of global_load /
// global_store, with the plane arithmetic on the scalar unit -- instead of one 64-bit VALU add per lane for
// each of the ~70 plane accesses of a step.
struct WgBuffers {
    double *pos, *quat;
    vel_t *vel, *angvel;
    float4 *pid;
    float *obs, *rpm;
    uint64_t *adj;
};
__device__ __forceinline__ WgBuffers wg_buffers(const StepArgs &A, size_t wg_base)
{
    WgBuffers w;
    w.pos = A.b.pos + wg_base; w.quat = A.b.quat + wg_base; w.vel = VELP(A.b.vel) + wg_base; w.angvel = VELP(A.b.angvel) + wg_base;
    // (no null tests: a base formed from a null pointer is never dereferenced -- every use sits behind a test of the MrsBuffers
    // member itself -- and without the tests the scalar loads of the four pointers go out together instead of one round trip each)
    w.pid = reinterpret_cast<float4 *>(A.b.pid) + wg_base;
    w.rpm = A.b.rpm + wg_base;
    w.obs = A.b.obs + wg_base * (size_t)A.D;
    w.adj = A.b.adj + wg_base * (size_t)A.W;
    return w;
}
__device__ __forceinline__ void load_state(const WgBuffers &b, unsigned t, size_t T, double p[3], double q[4], double v[3], double w[3])
{
    p[0] = b.pos[t]; p[1] = (b.pos + T)[t]; p[2] = (b.pos + 2 * T)[t];
    q[0] = b.quat[t]; q[1] = (b.quat + T)[t]; q[2] = (b.quat + 2 * T)[t]; q[3] = (b.quat + 3 * T)[t];
    v[0] = b.vel[t]; v[1] = (b.vel + T)[t]; v[2] = (b.vel + 2 * T)[t];
    w[0] = b.angvel[t]; w[1] = (b.angvel + T)[t]; w[2] = (b.angvel + 2 * T)[t];
}
__device__ __forceinline__ void store_state(const WgBuffers &b, unsigned t, size_t T, const double p[3], const double q[4], const double v[3], const double w[3])
{
    st<1>(b.pos + t, p[0]); st<1>(b.pos + T + t, p[1]); st<1>(b.pos + 2 * T + t, p[2]);
    st<1>(b.quat + t, q[0]); st<1>(b.quat + T + t, q[1]); st<1>(b.quat + 2 * T + t, q[2]); st<1>(b.quat + 3 * T + t, q[3]);
    st<1>(b.vel + t, (vel_t)v[0]); st<1>(b.vel + T + t, (vel_t)v[1]); st<1>(b.vel + 2 * T + t, (vel_t)v[2]);
    st<1>(b.angvel + t, (vel_t)w[0]); st<1>(b.angvel + T + t, (vel_t)w[1]); st<1>(b.angvel + 2 * T + t, (vel_t)w[2]);
}

// newest observation slice, (E,N,D) row-major: Environment.get_X of a concatenating state_fn
__device__ __forceinline__ void write_obs(unsigned code, int n_obs, float *o, const double p[3], const double q[4], const double v[3], const double w[3])
{
    // state_fn = cat(pos, vel) (README.md:28-29, the bench's): the 24-byte row as three 8-byte stores instead of six
    if (n_obs == 2 && code == (MRS_OBS_POS | (MRS_OBS_VEL << 4)) && (reinterpret_cast<uintptr_t>(o) & 7u) == 0) {
        float2 *o2 = reinterpret_cast<float2 *>(o);
        o2[0] = make_float2((float)p[0], (float)p[1]); o2[1] = make_float2((float)p[2], (float)v[0]); o2[2] = make_float2((float)v[1], (float)v[2]);
        return;
    }
    bool want_euler = false;
    for (int f = 0; f < n_obs; ++f) want_euler |= (((code >> (4 * f)) & 15u) == MRS_OBS_EULER);
    Observed ob;
    if (want_euler) observe<true, false>(p, q, v, w, ob);
    else observe<false, false>(p, q, v, w, ob);
    int off = 0;
    for (int f = 0; f < n_obs; ++f) {
        switch ((code >> (4 * f)) & 15u) {
        case MRS_OBS_POS: o[off] = ob.px; o[off + 1] = ob.py; o[off + 2] = ob.pz; off += 3; break;
        case MRS_OBS_VEL: o[off] = ob.vx; o[off + 1] = ob.vy; o[off + 2] = ob.vz; off += 3; break;
        case MRS_OBS_EULER: o[off] = ob.roll; o[off + 1] = ob.pitch; o[off + 2] = ob.yaw; off += 3; break;
        case MRS_OBS_ANGVEL: o[off] = ob.wx; o[off + 1] = ob.wy; o[off + 2] = ob.wz; off += 3; break;
        case MRS_OBS_QUAT: o[off] = (float)q[0]; o[off + 1] = (float)q[1]; o[off + 2] = (float)q[2]; o[off + 3] = (float)q[3]; off += 4; break;
        default: break;
        }
    }
}
__device__ __forceinline__ void write_obs(const StepArgs &A, float *o, const double p[3], const double q[4], const double v[3], const double w[3])
{
    write_obs(A.obs_code, A.n_obs, o, p, q, v, w);
}

// two float32 in an aligned register pair: the operand form of the packed instructions (v_pk_add/mul/fma_f32: two IEEE
// operations for the issue cost of one, tools/micro/valu_rates2.hip)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b)
{
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float a) { return f2{a, a}; }

// The value of lane - 1 (lane 0: of lane 63): a full-wave rotation by one lane on the vector unit (DPP wave_ror:1), no LDS
// round trip.  The N = 64 pair loops hand a result to "lane + k" by letting an accumulator TRAVEL: it takes the k = 31
// result on board first and moves one lane per step, so that what lane i put in at step k has moved k lanes when the loop
// is through (tools/micro/dpp_rot.hip: 31 hand-overs 0.96 us against 1.42 us through ds_bpermute_b32 at 16 waves per CU).
__device__ __forceinline__ uint32_t wave_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C, 0xf, 0xf, false); }
__device__ __forceinline__ float wave_ror1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, false)); }

// bit-packed adjacency row of agent i from the env's LDS position tile (MRS.calc_A, MRS.py:117-124):
// float32, d2 = fma(dz,dz,fma(dy,dy,dx*dx)) as torch's norm kernel evaluates it, sqrt(d2) <= R
// folded into d2 <= d2_thresh (largest float whose correctly rounded sqrt is <= R; host-computed).
// The tile is three arrays (tx, ty, tz = the env's first agent): one ds_read2_b32 -- a broadcast, every lane of the
// env reads the same j -- puts agents j and j+1 into a register pair, the differences and the squared distance of
// both come from five packed instructions, and each verdict lands at a compile-time bit position (one select and one
// OR; the row's own bit is cleared once at the end).  Was: one float4 read, six scalar operations, two compares and a
// 64-bit variable shift per agent -- 13 vector instructions per pair against 6 (N = 256: 68 -> see docs/experiments.md section 6).
__device__ __forceinline__ bool adjacency_row(const StepArgs &A, float thr_s, const float *tx, const float *ty, const float *tz, int i, float4 me, uint64_t *row,
                                              unsigned long long *hrow)
{
    // returns whether another agent of the env is within quad-quad contact range (A.pair_rc2), and notes which in hrow
    // (same bit layout as row; null when the contact model is off)
    bool hit = false;
    const bool want_hit = hrow != nullptr;
    float rc2 = A.pair_rc2;
    asm volatile("" : "+v"(rc2));
    float thr = thr_s;
    asm volatile("" : "+v"(thr)); // not re-read from the argument segment inside the loop (see adjacency_phase)
    const bool inf = A.comm_inf != 0;
    const f2 mx = splat(me.x), my = splat(me.y), mz = splat(me.z);
    // the next float above each threshold (finite and positive; a negative range sentinel only moves further from every d2 >= 0)
    f2 thr_up2 = splat(__uint_as_float(__float_as_uint(thr) + (thr >= 0.f ? 1u : 0u))), rc2_up2 = splat(__uint_as_float(__float_as_uint(rc2) + 1u));
    asm volatile("" : "+v"(thr_up2), "+v"(rc2_up2));
    for (int wd = 0; wd < A.W; ++wd) {
        const int j0 = wd * 64, jn = min(64, A.N - j0);
        uint32_t half[2] = {0u, 0u}, hhalf[2] = {0u, 0u};
        if (inf && !want_hit) {
            const uint64_t all = jn == 64 ? ~0ull : ((1ull << jn) - 1ull);
            half[0] = (uint32_t)all; half[1] = (uint32_t)(all >> 32);
        } else if (jn == 64) {
            // A verdict costs 1.5 instructions (round 2: three -- compare, select, or): d2 <= T  <=>  d2 < T' with T' the next
            // float above T  <=>  the SIGN BIT of d2 - T' (a subtraction of two different floats never rounds to zero, so the
            // sign is exact); one packed subtraction gives two signs and v_alignbit_b32 shifts each into the word,
            // ({word, sign} >> 31).  The first verdict ends up in the top bit: one v_bfrev per word puts them in column order.
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                uint32_t bits = 0, hb = 0;
#pragma unroll
                for (int jj = 0; jj < 32; jj += 2) {
                    const int j = j0 + hh * 32 + jj;
                    const f2 dx = pk_sub(mx, f2{tx[j], tx[j + 1]}), dy = pk_sub(my, f2{ty[j], ty[j + 1]}), dz = pk_sub(mz, f2{tz[j], tz[j + 1]});
                    const f2 d2 = pk_fma(dz, dz, pk_fma(dy, dy, pk_mul(dx, dx)));
                    const f2 st = pk_sub(d2, thr_up2);
                    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(st.x), 31);
                    bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(st.y), 31);
                    if (want_hit) {
                        const f2 sc = pk_sub(d2, rc2_up2);
                        hb = __builtin_amdgcn_alignbit(hb, __float_as_uint(sc.x), 31);
                        hb = __builtin_amdgcn_alignbit(hb, __float_as_uint(sc.y), 31);
                    }
                }
                half[hh] = __builtin_bitreverse32(bits); hhalf[hh] = __builtin_bitreverse32(hb);
            }
        } else { // the last, partial word of an N that is not a multiple of 64
            for (int jj = 0; jj < jn; ++jj) {
                const int j = j0 + jj;
                const float dx = f32sub(me.x, tx[j]), dy = f32sub(me.y, ty[j]), dz = f32sub(me.z, tz[j]);
                const float d2 = f32fma(dz, dz, f32fma(dy, dy, f32mul(dx, dx)));
                if (d2 <= thr) half[jj >> 5] |= 1u << (jj & 31);
                if (d2 <= rc2) hhalf[jj >> 5] |= 1u << (jj & 31);
            }
        }
        if (inf) { // ones - eye whatever the positions are (the loops above ran for the contact range only)
            const uint64_t all = jn == 64 ? ~0ull : ((1ull << jn) - 1ull);
            half[0] = (uint32_t)all; half[1] = (uint32_t)(all >> 32);
        }
        uint64_t bits = ((uint64_t)half[1] << 32) | half[0], hbits = ((uint64_t)hhalf[1] << 32) | hhalf[0];
        if ((unsigned)(i - j0) < 64u) { bits &= ~(1ull << (i - j0)); hbits &= ~(1ull << (i - j0)); } // ones - eye (MRS.py:118-119, :123)
        if (row) row[wd] = bits;
        if (want_hit) { hrow[wd] = hbits; hit |= hbits != 0; }
    }
    return hit;
}


// N = 64: an env is exactly one wave, so its LDS position tile is written and read by the same wave and needs
// no workgroup barrier -- LDS operations of one wave execute in order; this only pins the compiler's ordering.
// (Every workgroup barrier couples four waves that sit on four different SIMDs and progress at different rates:
// measured with clock64 stamps, waves spent ~10 % of their lifetime in the two barriers this replaces.)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// N = 64 position tile: an env is one wave, its 64 float32 positions sit in LDS as three arrays x[128], y[128], z[128]
// (each STORED TWICE back to back so that "neighbour (lane + k) mod 64" is the un-wrapped slot lane + k: a constant
// offset per unrolled k) inside the env's 128-float4 slot of lds_tile.  Separate arrays, not float4 records, so that one
// ds_read2_b32 puts the coordinates of neighbours k and k+1 into an aligned register pair -- the operand form of the
// packed float32 instructions (v_pk_add/mul/fma_f32: two IEEE operations for the issue cost of one, measured in
// tools/micro/valu_rates2.hip), which the pair loops below use for everything that is not a transcendental or a compare.
__device__ __forceinline__ float *tile64(float4 *lds_tile, int el) { return reinterpret_cast<float *>(lds_tile + el * 128); }
__device__ __forceinline__ void tile64_write(float4 *lds_tile, int el, int i, float x, float y, float z)
{
    float *t = tile64(lds_tile, el);
    t[i] = t[64 + i] = x; t[128 + i] = t[192 + i] = y; t[256 + i] = t[320 + i] = z;
}
// the differences to neighbours k and k+1 of the lane whose tile pointer (already offset by the lane) is t
// The z array sits 256 dwords behind x, one more than ds_read2_b32's offset field reaches: "t[256 + k]" costs an address
// addition per pass.  tz = the LDS address of t + 256 in a register of its own (kept an LDS pointer: laundering a generic
// pointer through asm would turn the reads into flat loads -- measured, +4 us; as it stands -46 instructions, -0.1 ... -0.2 us).
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ lds_cfloat *tile64_z(const float *t)
{
    unsigned off = (unsigned)(uintptr_t)(lds_cfloat *)(t + 256);
    asm volatile("" : "+v"(off));
    return (lds_cfloat *)(uintptr_t)off;
}
#define TILE64_Z(t) lds_cfloat *const tz = tile64_z(t)
template <class PZ>
__device__ __forceinline__ void tile64_rel2z(const float *t, PZ tz, int k, f2 mx, f2 my, f2 mz, f2 &rx, f2 &ry, f2 &rz)
{
    rx = pk_sub(f2{t[k], t[k + 1]}, mx); ry = pk_sub(f2{t[128 + k], t[128 + k + 1]}, my); rz = pk_sub(f2{tz[k], tz[k + 1]}, mz);
}
#define tile64_rel2(t, k, mx, my, mz, rx, ry, rz) tile64_rel2z(t, tz, k, mx, my, mz, rx, ry, rz)

// StepArgs.pair_flag[e], N = 64: exactly ONE pair of the env is within contact range, agents (flag & 63) and (flag >> 8 & 63).
// By far the common case among the flagged envs, and the step kernel's duration is that of its slowest wave: a wave that scans
// its env (63 neighbours per lane, alone on its SIMD by then) lengthened every launch by ~1.5 us; with the pair named it
// evaluates one term.  (Not a bit of 0x01010101, the "unknown" pattern that hipMemset(.., 1, ..) leaves.)
#define MRS_PAIR_SINGLE 0x40000000
// several pairs: StepArgs.pair_rows holds, for every agent of the env, the agents within contact range of it (bit j of word
// j / 64), written by the same adjacency pass.  Any other non-zero value (1, the memset pattern): unknown, the step looks
// at every pair of the env.
#define MRS_PAIR_ROWS 2

// ---- N = 64: an env is one wave.  The relation is symmetric with bit-identical arithmetic in both directions
// ((-dx)^2 == dx^2), so each unordered pair is tested once: lane i tests (i, i+k), k = 1..31, notes it at RELATIVE bit k
// and hands the verdict to lane i+k, where it is relative bit 64-k; k = 32 is tested by both ends.  The hand-over is a
// travelling word (wave_ror1), k descending; one 64-bit rotate by the lane index at the end turns relative into
// absolute columns.  t = the env's tile + lane (tile64_write), (mex, mey, mez) = this lane's own float32 position.
//   rows = false: COMM_RANGE = inf (MRS.py:118-119: ones - eye whatever the positions are) or no rows wanted
//   want_hit: also the smallest squared distance this lane has seen (quad-quad contact range: one v_min3_f32 per two pairs;
//   round 2 built a bit per pair, three instructions each, for an event that concerns 0.1-0.3 % of the envs)
// (Round 2 experiment, removed: evaluating the NEXT step's downwash in this loop and carrying the force to the next launch:
// +1.2 us per step, the work moves from the start of the kernel, where issue slots are idle, to its end, where none are.
// Round 5, built again on the present kernel -- the pair term riding in this loop on the shared differences and dxy^2, the sum and a
// 64-bit signature of the position in a 16-byte record per agent, taken by the next step when all 64 signatures of the env match;
// bit-identical results, no scratch, the start's loop, tile and 96 LDS reads gone for 99.9 % of the envs: +4.0 us per step.  The
// tail is what the launch's latest workgroups run ALONE, behind their contact solve; an instruction there costs three of the start's.)
__device__ __forceinline__ void adj64_pass(const float *t, float mex, float mey, float mez, float thr_s, bool rows, bool want_hit, int lane,
                                           uint64_t &row, float &dmin)
{
    // own verdicts at bit k of `lo`; the verdicts from below collect in `hr` at bit k too and are mirrored into place
    // (relative bit 64-k) by one v_bfrev at the end: one select and two ORs per pair, no per-pair shifts
    uint32_t lo = 0, hr = 0, top = 0;
    // The range threshold lives in a VECTOR register for the loop (comm_range = inf arrives as thr = +inf from the host).
    // As a kernel argument it is re-read from the argument segment inside every pair once scalar registers are short --
    // s_load + s_waitcnt lgkmcnt(0), which also drains the pair's LDS read: the loop then runs one pair per memory round trip.
    float thr = thr_s;
    asm volatile("" : "+v"(thr));
    dmin = __builtin_huge_valf();
    const f2 mx = splat(mex), my = splat(mey), mz = splat(mez);
    TILE64_Z(t);
    if (rows) {
        auto verdict = [&](int k, float d2) {
            const uint32_t bit = d2 <= thr ? (1u << k) : 0u;
            lo |= bit;
            hr = (k == 31 ? 0u : wave_ror1(hr)) | bit;
        };
        {   // k = 31, and k = 32: relative bit 32 is tested by both ends
            f2 rx, ry, rz;
            tile64_rel2(t, 31, mx, my, mz, rx, ry, rz);
            const f2 d2 = pk_fma(rz, rz, pk_fma(ry, ry, pk_mul(rx, rx)));
            verdict(31, d2.x);
            top = d2.y <= thr ? 1u : 0u;
            dmin = __builtin_fminf(__builtin_fminf(dmin, d2.x), d2.y);
        }
#pragma unroll
        for (int k = 29; k >= 1; k -= 2) { // two neighbours per pass: d2 = fma(dz,dz,fma(dy,dy,dx*dx)) as torch's norm kernel evaluates it
            f2 rx, ry, rz;
            tile64_rel2(t, k, mx, my, mz, rx, ry, rz);
            const f2 d2 = pk_fma(rz, rz, pk_fma(ry, ry, pk_mul(rx, rx)));
            verdict(k + 1, d2.y);
            verdict(k, d2.x);
            dmin = __builtin_fminf(__builtin_fminf(dmin, d2.x), d2.y);
        }
        hr = wave_ror1(hr); // the k = 1 verdict's one step
    } else if (want_hit) { // the contact range only, nothing to hand over
#pragma unroll
        for (int k = 1; k < 33; k += 2) {
            f2 rx, ry, rz;
            tile64_rel2(t, k, mx, my, mz, rx, ry, rz);
            const f2 d2 = pk_fma(rz, rz, pk_fma(ry, ry, pk_mul(rx, rx)));
            dmin = __builtin_fminf(__builtin_fminf(dmin, d2.x), d2.y);
        }
    }
    const uint32_t hi = (__builtin_bitreverse32(hr) << 1) | top; // bit k -> bit 32-k
    const uint64_t rel = rows ? (((uint64_t)hi << 32) | lo) : ~1ull;
    row = lane ? ((rel << lane) | (rel >> (64 - lane))) : rel;
}

// StepArgs.pair_flag[e] / pair_rows of an N = 64 env whose tile (t = tile + lane) holds the positions the NEXT step starts
// from.  near = this lane may have a pair within contact range (exactly so, or conservatively: a wave in which no lane says
// so has none, a wave in which one does walks its 32 pairs per lane again -- rare, a fraction of a percent of the envs).
__device__ __forceinline__ void adj64_flag(const StepArgs &A, const float *t, float mex, float mey, float mez, int lane, int e, bool near)
{
    float rc2 = A.pair_rc2;
    asm volatile("" : "+v"(rc2));
    int flag = 0;
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
        // which of its pairs: neighbours k = 1..31 within range at bit k of hm, the antipode k = 32 in ht
        // (the same float32 operations as the pass, hence the same distances)
        // Unrolled, two neighbours per pass like the pass itself: the wave is alone on its SIMD by now and the launch waits for
        // it -- as a rolled loop of 32 dependent LDS round trips this walk was 1.9 us (tools/probes/timeline_simd.py: the seven latest
        // workgroups of a bench launch were the seven with a flagged env), now the reads are in flight together.
        uint32_t hm = 0;
        bool ht = false;
        {
            const f2 mx = splat(mex), my = splat(mey), mz = splat(mez);
            TILE64_Z(t);
#pragma unroll
            for (int k = 1; k < 33; k += 2) {
                f2 rx, ry, rz;
                tile64_rel2(t, k, mx, my, mz, rx, ry, rz);
                const f2 d2 = pk_fma(rz, rz, pk_fma(ry, ry, pk_mul(rx, rx)));
                hm |= d2.x <= rc2 ? (1u << k) : 0u;
                if (k + 1 < 32) hm |= d2.y <= rc2 ? (1u << (k + 1)) : 0u; else ht = d2.y <= rc2;
            }
        }
        const int cnt = __builtin_popcount(hm) + (ht ? 1 : 0);
        const uint64_t testers = __builtin_amdgcn_ballot_w64(cnt != 0);
        if (testers != 0) {
            flag = MRS_PAIR_ROWS; // several pairs: every lane notes its partners (below)
            const int a = __builtin_ctzll(testers), nb = __builtin_popcountll(testers);
            if (__builtin_amdgcn_ballot_w64(cnt > 1) == 0) {
                const uint32_t hma = (uint32_t)__builtin_amdgcn_readlane((int)hm, a);
                if (nb == 1 && hma != 0) {
                    flag = MRS_PAIR_SINGLE | a | (((a + __builtin_ctz(hma)) & 63) << 8);
                } else if (nb == 2 && a < 32 && __builtin_amdgcn_ballot_w64(ht && hm == 0) == testers && (testers >> (a + 32)) == 1ull) {
                    flag = MRS_PAIR_SINGLE | a | ((a + 32) << 8);
                }
            }
            if (flag == MRS_PAIR_ROWS) {
                // the partners of a lane: the ones it tested itself (hm, relative bit k) and the ones that tested it
                // (handed to lane + k, mirrored into relative bit 64-k), rotated into absolute columns
                uint32_t hrh = 0;
                const int lane4 = lane << 2;
#pragma unroll
                for (int k = 1; k < 32; ++k) hrh |= (uint32_t)__builtin_amdgcn_ds_bpermute(lane4 + 4 * (64 - k), (int)(hm & (1u << k)));
                const uint64_t hrel = ((uint64_t)((__builtin_bitreverse32(hrh) << 1) | (ht ? 1u : 0u)) << 32) | hm;
                A.pair_rows[(size_t)e * 64 + lane] = lane ? ((hrel << lane) | (hrel >> (64 - lane))) : hrel;
            }
        }
    }
    if (lane == 0) A.pair_flag[e] = flag;
}

// ---- N = 128, 192, 256 (256-thread workgroups): an env is nb = N/64 waves, each wave one 64-agent block of it with an
// N = 64 tile of its own, and every unordered pair is tested ONCE -- own block: adj64_pass; blocks b and b+1: the wave of b
// (adj_cross<64>); blocks b and b + nb/2 (nb even): each of the two waves half of the pairs (adj_cross<32>, see
// downwash_cross for the split).  What the loop costs is LDS bandwidth before arithmetic (three ds_read2_b32 = 12 LDS cycles
// per two pairs, sixteen waves per CU on one LDS): 128 tests per lane instead of the 256 of adjacency_row.
// Lane l meets the other block's lanes (l + k) mod 64, k descending (tp = that block's tile + l + o).  Its own verdicts are
// shifted into own[] in that order (bit k - 32 of own[1], bit k of own[0]: see adjacency_row for the sign-bit verdict);
// the other lane's copy travels (wave_ror1 ahead of every shift): after the k = 32 (k = 0) pass, the word in lane x holds,
// at the same bit positions, the verdicts of the other block's lane x + 32 (x), and is handed to that wave through LDS.
template <int K, bool ROWS>
__device__ __forceinline__ void adj_cross(const float *tp, float mex, float mey, float mez, f2 thr_up2, uint32_t own[2], uint32_t trav[2], float &dmin)
{
    const f2 mx = splat(mex), my = splat(mey), mz = splat(mez);
    const float *t = tp + (K - 8);
#pragma unroll
    for (int ph = K / 32 - 1; ph >= 0; --ph) {
        uint32_t ow = 0, tr = 0;
#pragma unroll 1
        for (int it = 0; it < 4; ++it, t -= 8) {
            TILE64_Z(t);
#pragma unroll
            for (int k = 6; k >= 0; k -= 2) {
                f2 rx, ry, rz;
                tile64_rel2(t, k, mx, my, mz, rx, ry, rz);
                const f2 d2 = pk_fma(rz, rz, pk_fma(ry, ry, pk_mul(rx, rx)));
                if (ROWS) {
                    const f2 sg = pk_sub(d2, thr_up2);
                    ow = __builtin_amdgcn_alignbit(ow, __float_as_uint(sg.y), 31);
                    tr = __builtin_amdgcn_alignbit(wave_ror1(tr), __float_as_uint(sg.y), 31);
                    ow = __builtin_amdgcn_alignbit(ow, __float_as_uint(sg.x), 31);
                    tr = __builtin_amdgcn_alignbit(wave_ror1(tr), __float_as_uint(sg.x), 31);
                }
                dmin = __builtin_fminf(__builtin_fminf(dmin, d2.x), d2.y);
            }
        }
        if (ROWS) { own[ph] = ow; trav[ph] = tr; }
    }
}
// verdicts received "about lane m - k at bit k"  ->  relative column bits of lane m (bit (64 - k) mod 64)
__device__ __forceinline__ uint64_t adj_mirror64(uint64_t x) { return (__builtin_bitreverse64(x) << 1) | (x & 1ull); }
__device__ __forceinline__ uint64_t adj_rotl64(uint64_t x, int n) { return n ? ((x << n) | (x >> (64 - n))) : x; }

// MrsBuffers.adj_dense: the rows of one wave's 64 agents as the float32 0/1 matrix rows the reference returns (MRS.py:117-124),
// written by the wave that has just built them instead of by a second kernel that reads the packed rows back
// (mrs_adjacency_expand): the words go through LDS (rl: 64 * WPR words of the wave's own, dead, tile) so that every store
// instruction is 64 lanes x 16 bytes of one contiguous kilobyte -- lane l of pass q holds the float4 number 64 q + l of the
// wave's 64 x N chunk.  WPR = words per row = N / 64; dst = the chunk (16-byte aligned, checked by the host).
template <int WPR>
__device__ __forceinline__ void dense_rows_store(uint64_t *rl, const uint64_t *word, int lane, float *dst)
{
#pragma unroll
    for (int c = 0; c < WPR; ++c) rl[lane * WPR + c] = word[c];
    wave_lds_sync();
    float4 *const d4 = reinterpret_cast<float4 *>(dst);
    constexpr int C4 = 16 * WPR; // float4 per row
#pragma unroll 4
    for (int q = 0; q < C4; ++q) {
        const int f = 64 * q + lane, r = f / C4, c4 = f - r * C4;
        const uint32_t nib = (uint32_t)(rl[r * WPR + (c4 >> 4)] >> ((c4 & 15) * 4));
        d4[f] = make_float4((nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f);
    }
}

// One pass over the env's pairs for one squared-distance threshold: the lane's nb row words (word c = the agents of block c),
// the other waves' verdicts through the exchange words.  Contains a workgroup barrier when rows is set (uniform).
// rows = false: ones - eye, only the minima are computed (if wanted).  The tiles must hold the positions.
// dm[0..2] = the smallest squared distance among the lane's pairs in its own block / with the next block / with the block
// opposite; run (wave-uniform) = which of the three parts to walk at all (bit 0, 1, 2) -- a part left out has no pair within
// the threshold, its verdicts are zeros.
struct BlockIds { int nb, b, next, prev, opp, o; };
template <int BLOCK>
__device__ __forceinline__ void adj_blocks_pass(float4 *lds_tile, int tid, bool live, const BlockIds &B, float4 mine, float thr_s, bool rows, bool want_min,
                                                int run, uint64_t word[4], float dm[3])
{
    // x3 = lds_tile + 2 * BLOCK is k_step's `nanflag[256]` region (dead by the tail: the NaN vote is read once, before the pair
    // loop) and ends exactly where `ncontact` starts: 4 waves x 64 words.  Only true for 256-thread workgroups.
    static_assert(BLOCK == 256, "adjacency_blocks: the third exchange array overlays nanflag[256] of a 256-thread workgroup");
    const int lane = tid & 63, wt = tid >> 6, nb = B.nb, b = B.b, o = B.o; // wt - b = the tile of the env's first block
    uint32_t *const my_x = reinterpret_cast<uint32_t *>(tile64(lds_tile, wt)); // exchange words 384 + lane, 448 + lane: the tile's spare quarter
    uint32_t *const x3 = reinterpret_cast<uint32_t *>(lds_tile + 2 * BLOCK);    // the third one: behind the tiles
    uint64_t r_own = ~(1ull << lane), rel_next = ~0ull, rel_prev = ~0ull, rel_opp = ~0ull; // COMM_RANGE = inf: ones - eye
    if (live && (rows || want_min)) {
        if (run & 1) adj64_pass(tile64(lds_tile, wt) + lane, mine.x, mine.y, mine.z, thr_s, rows, want_min, lane, r_own, dm[0]);
        else r_own = 0;
        float thr = thr_s;
        asm volatile("" : "+v"(thr));
        f2 thr_up2 = splat(__uint_as_float(__float_as_uint(thr) + (thr >= 0.f ? 1u : 0u)));
        asm volatile("" : "+v"(thr_up2));
        if (nb >= 3) {
            float *const ot = tile64(lds_tile, wt - b + B.next);
            uint32_t own[2] = {~0u, ~0u}, trav[2] = {0u, 0u};
            if (!(run & 2)) own[0] = own[1] = 0u;
            else if (rows) adj_cross<64, true>(ot + lane, mine.x, mine.y, mine.z, thr_up2, own, trav, dm[1]);
            else adj_cross<64, false>(ot + lane, mine.x, mine.y, mine.z, thr_up2, own, trav, dm[1]);
            rel_next = ((uint64_t)own[1] << 32) | own[0];
            if (rows) {
                uint32_t *const ox = reinterpret_cast<uint32_t *>(ot);
                ox[384 + ((lane + 32) & 63)] = trav[1]; ox[448 + lane] = trav[0];
            }
        }
        if (!(nb & 1)) {
            float *const ot = tile64(lds_tile, wt - b + B.opp);
            uint32_t own[2] = {~0u, ~0u}, trav[2] = {0u, 0u};
            if (!(run & 4)) own[0] = 0u;
            else if (rows) adj_cross<32, true>(ot + lane + o, mine.x, mine.y, mine.z, thr_up2, own, trav, dm[2]);
            else adj_cross<32, false>(ot + lane + o, mine.x, mine.y, mine.z, thr_up2, own, trav, dm[2]);
            if (rows) {
                rel_opp = (uint64_t)own[0] << o;
                x3[(wt - b + B.opp) * 64 + ((lane + o) & 63)] = trav[0];
            }
        }
    }
    if (rows) { // uniform: the threshold is the launch's
        __syncthreads();
        if (live) {
            if (nb >= 3) rel_prev = adj_mirror64(((uint64_t)my_x[384 + lane] << 32) | my_x[448 + lane]);
            if (!(nb & 1)) rel_opp |= adj_mirror64((uint64_t)x3[wt * 64 + lane] << (1 - o));
        }
    }
    word[b] = r_own;
    if (nb >= 3) { word[B.next] = adj_rotl64(rel_next, lane); word[B.prev] = adj_rotl64(rel_prev, lane); }
    if (!(nb & 1)) word[B.opp] = adj_rotl64(rel_opp, lane);
}

template <int BLOCK>
__device__ __forceinline__ void adjacency_blocks(const StepArgs &A, float thr_s, bool comm_inf, float4 *lds_tile, int tid, int el, int i, bool live,
                                                 uint64_t *row, float4 mine, int e)
{
    const bool want_hit = A.pair_flag != nullptr;
    BlockIds B;
    B.nb = A.N >> 6; B.b = i >> 6;
    const int hb = B.nb >> 1;
    B.o = B.b >= hb ? 1 : 0; B.opp = B.b >= hb ? B.b - hb : B.b + hb; B.next = B.b + 1 == B.nb ? 0 : B.b + 1; B.prev = B.b ? B.b - 1 : B.nb - 1;
    tile64_write(lds_tile, tid >> 6, tid & 63, mine.x, mine.y, mine.z);
    __syncthreads();
    float dm[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    uint64_t word[4];
    adj_blocks_pass<BLOCK>(lds_tile, tid, live, B, mine, thr_s, !comm_inf, want_hit, 7, word, dm);
    if (row && live && KO_KEEP(2)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < B.nb) st<2>(row + c, word[c]);
    }
    if (want_hit) {
        // quad-quad contact range: a pair is seen by one of its two lanes.  An env with a hit (a few per cent of them once bodies
        // lie on the ground side by side; a launch lasts as long as its slowest workgroup) notes every agent's partners: the
        // same pass with the contact range as the threshold, but only the parts of it in which a lane of the wave has seen a
        // hit -- one of a dozen, typically.  The tiles still hold the positions; the vote's barrier separates the two uses of
        // the exchange words.
        float rc2 = A.pair_rc2;
        asm volatile("" : "+v"(rc2));
        const int run = (__builtin_amdgcn_ballot_w64(live && dm[0] <= rc2) ? 1 : 0) | (__builtin_amdgcn_ballot_w64(live && dm[1] <= rc2) ? 2 : 0) |
                        (__builtin_amdgcn_ballot_w64(live && dm[2] <= rc2) ? 4 : 0);
        if (__syncthreads_or(run)) {
            float unused[3];
            uint64_t hw[4];
            adj_blocks_pass<BLOCK>(lds_tile, tid, live, B, mine, rc2, true, false, run, hw, unused);
            bool hit = false;
            if (live) {
                unsigned long long *const hrow = A.pair_rows + ((size_t)e * A.N + i) * A.W;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < B.nb) { hrow[c] = hw[c]; hit |= hw[c] != 0; }
            }
            // the env's flag: its waves vote through the first exchange word of their tiles (dead after the pass's barrier)
            __syncthreads();
            uint32_t *const vote = reinterpret_cast<uint32_t *>(tile64(lds_tile, tid >> 6)) + 384;
            const bool wave_hit = __builtin_amdgcn_ballot_w64(hit) != 0;
            if ((tid & 63) == 0) vote[0] = wave_hit ? 1u : 0u;
            __syncthreads();
            if (live && i == 0) {
                bool any = false;
                for (int c = 0; c < B.nb; ++c) any |= reinterpret_cast<uint32_t *>(tile64(lds_tile, (tid >> 6) + c))[384] != 0;
                A.pair_flag[e] = any ? MRS_PAIR_ROWS : 0;
            }
        } else if (live && i == 0) A.pair_flag[e] = 0;
    }
    if (A.b.adj_dense && row) { // uniform.  The tiles are dead once every wave is through the passes above
        __syncthreads();
        if (__builtin_amdgcn_ballot_w64(live) != 0) { // whole waves are live or not (N is a multiple of 64)
            uint64_t *const rl = reinterpret_cast<uint64_t *>(tile64(lds_tile, tid >> 6));
            float *const dst = A.b.adj_dense + ((size_t)e * A.N + (size_t)(i & ~63)) * A.N;
            if (B.nb == 4) dense_rows_store<4>(rl, word, tid & 63, dst);
            else if (B.nb == 3) dense_rows_store<3>(rl, word, tid & 63, dst);
            else dense_rows_store<2>(rl, word, tid & 63, dst);
        }
    }
}

// COMM_RANGE adjacency of the workgroup's envs from their CURRENT positions (`mine` per lane), staged through
// the LDS position tile.  Contains a workgroup barrier: every thread of the workgroup must call it.
// Also (A.pair_flag): notes per env whether any pair of it is within quad-quad contact range of these positions -- the
// squared distances are formed here anyway -- for the step that starts from them.  row may be null (flag only).
template <int BLOCK, bool N64 = false>
__device__ __forceinline__ void adjacency_phase(const StepArgs &A, float thr_s, bool comm_inf, float4 *lds_tile, int tid, int el, int i, bool live,
                                                uint64_t *row, float4 mine, int e)
{
    const bool n64 = N64 || A.N == 64; // N64: the N = 64 instantiation of k_step (the other branches fold away)
    // A NaN (or infinite) position -- a diverged or mis-set body -- is adjacent to nobody: torch's `dist <= R` is False for a
    // NaN distance (MRS.py:121).  The sign-bit verdicts below read the sign of d^2 - T', and a NaN carries a sign of its own
    // through the arithmetic (x86 hands out negative ones, and the negate modifiers of the packed subtractions flip them), so no
    // NaN enters the pair loops at all: such a coordinate is replaced by a finite one far outside every range and different
    // for every agent of the env (1e18 m times one plus its index: two such agents are 1e18 m apart as well).
    {
        const float far = 1e18f * (float)(1 + i);
        mine.x = fabsf(mine.x) <= 1e17f ? mine.x : far; mine.y = fabsf(mine.y) <= 1e17f ? mine.y : far; mine.z = fabsf(mine.z) <= 1e17f ? mine.z : far;
    }
    if constexpr (BLOCK == 256 && !N64) {
        if (A.N > 64 && (A.N & 63) == 0) { // 128, 192, 256
            adjacency_blocks<BLOCK>(A, thr_s, comm_inf, lds_tile, tid, el, i, live, row, mine, e);
            return;
        }
    }
    const bool want_hit = A.pair_flag != nullptr;
    int *const hit_flag = reinterpret_cast<int *>(lds_tile) + 3 * BLOCK; // generic N: one word per env slot, behind the three arrays
    if (!n64 && want_hit && tid < A.epb) hit_flag[tid] = 0;
    float *const gx = reinterpret_cast<float *>(lds_tile); // any other N: three arrays of BLOCK floats (see adjacency_row)
    if (n64) tile64_write(lds_tile, el, i, mine.x, mine.y, mine.z);
    else { gx[tid] = mine.x; gx[BLOCK + tid] = mine.y; gx[2 * BLOCK + tid] = mine.z; }
    if (n64) wave_lds_sync(); else __syncthreads();
    if (n64) {
        if (live) {
            const int lane = tid & 63;
            const float *t = tile64(lds_tile, el) + lane;
            uint64_t r64;
            float dmin;
            adj64_pass(t, mine.x, mine.y, mine.z, thr_s, !comm_inf, want_hit, lane, r64, dmin);
            if (want_hit) adj64_flag(A, t, mine.x, mine.y, mine.z, lane, e, dmin <= A.pair_rc2);
            if (row && KO_KEEP(2)) st<2>(row, r64);
            if (row && A.b.adj_dense) // (uniform) the env's matrix, by the env's wave
                dense_rows_store<1>(reinterpret_cast<uint64_t *>(tile64(lds_tile, el)), &r64, lane, A.b.adj_dense + (size_t)e * 64 * 64);
        }
    } else {
        bool hit = false;
        if (live) hit = adjacency_row(A, thr_s, gx + el * A.N, gx + BLOCK + el * A.N, gx + 2 * BLOCK + el * A.N, i, mine, row,
                                      want_hit ? A.pair_rows + ((size_t)e * A.N + i) * A.W : nullptr);
        if (want_hit) { // workgroup-uniform: every thread takes the barrier
            if (hit) atomicOr(&hit_flag[el], 1);
            __syncthreads();
            if (live && i == 0) A.pair_flag[e] = hit_flag[el] ? MRS_PAIR_ROWS : 0; // every lane of the env has written its row
        }
    }
}

// |pair term| for two neighbours at once: d2 = dxy^2, rz = dz of both.  |dz| enters only through the abs operand modifier
// of v_rcp_f32 / v_fma_f32 (the packed forms have none); operation for operation the arithmetic of downwash_mag2.
__device__ __forceinline__ f2 downwash_mag2_pk(f2 d2, f2 rz, const DownwashRegs &dr)
{
    const f2 rdz = {__builtin_amdgcn_rcpf(fabsf(rz.x)), __builtin_amdgcn_rcpf(fabsf(rz.y))};
    const f2 rb = {__builtin_amdgcn_rcpf(__builtin_fmaf(dr.dw2, fabsf(rz.x), dr.dw3)), __builtin_amdgcn_rcpf(__builtin_fmaf(dr.dw2, fabsf(rz.y), dr.dw3))};
    const f2 arg = pk_fma(pk_mul(pk_mul(d2, rb), rb), splat(-0.5f * 1.44269504088896341f), splat(dr.lg));
    const f2 ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    return pk_mul(pk_mul(rdz, rdz), ex);
}

// Round 4, built, bit-identical (same checksums), and measured WITHOUT gain -- left off: N = 256 x 1024 envs,
// set_control, steady state: 47.8 / 48.1 us per step with the vote against 47.7 / 47.4 without (six sweeps: 45.8 / 45.5 against
// 44.5 / 44.5).  The vote can only drop a pass in which all 64 lanes' pairs are exact zeros; that holds on the spawn grid (lane
// offset k = a fixed grid displacement), but 700 steps of open-loop thrust scatter the swarm over tens of metres before it
// comes down, the lane index no longer says anything about where an agent is, and with ~2 % of the pairs live a pass of 128
// pairs has a live one nine times out of ten.  What would help there is a per-lane list of the live partners (a scan at ~16
// instructions per pass either way); not built.
#ifndef MRS_DW_SKIP
#define MRS_DW_SKIP 0 // A/B switch (tools/abl_build.sh): 1 = vote out the passes of the multi-wave pair loops whose terms are all exact zeros
#endif
// Does any lane of the wave have a pair in this pass (dxy^2 = d2, dz = rz, two neighbours) whose term is not an exact zero?
// (DownwashConst.zero_c; a pair outside the 10 m cylinder or with a non-finite distance is deselected anyway.)  On the 16 m
// grid of BASELINE config 4 nine passes in ten have none: six instructions and a scalar branch instead of the two
// reciprocals, the exponential and the selects of two neighbours.
__device__ __forceinline__ bool downwash_pass_live(f2 d2, f2 rz, const DownwashRegs &dr)
{
    const f2 beta = {__builtin_fmaf(dr.dw2, fabsf(rz.x), dr.dw3), __builtin_fmaf(dr.dw2, fabsf(rz.y), dr.dw3)};
    const f2 t = pk_fma(pk_mul(beta, beta), splat(dr.nzc), d2); // > 0: dxy^2 > zero_c beta^2, the term is exactly 0
    const bool live = (t.x <= 0.f && d2.x < 100.f) || (t.y <= 0.f && d2.y < 100.f);
    return __builtin_amdgcn_ballot_w64(live) != 0;
}

// The downwash of an N = 64 env from its LDS tile (see tile64_write), evaluated once per unordered pair: see k_step.
// t = the env's tile + lane.  Two neighbours per pass through the packed float32 instructions; the reciprocals, the
// exponential and the selects stay per neighbour.  Per pair operation for operation the arithmetic of downwash_mag2;
// the terms a lane keeps and the terms handed to it are summed separately (float32, flushed to float64 every 8 pairs).
// hook(j), j = 0..14, runs ahead of the j-th pass: k_step spreads the state loads the loop does not need over the passes.
// SKIP (the blocks of multi-wave envs): passes without a live pair are voted out (downwash_pass_live); bit-identical sums.
template <bool SKIP, class Hook>
__device__ __forceinline__ double downwash_ring64(const float *t, float mx_, float my_, float mz_, int lane4, const DownwashConst &dc, Hook &&hook)
{
    const DownwashRegs dr = downwash_regs(dc);
    const f2 mx = splat(mx_), my = splat(my_), mz = splat(mz_);
    TILE64_Z(t);
    // Magnitudes are summed and the sign is applied once at the end (-(a + b) == (-a) + (-b) bit for bit).  The terms a lane
    // keeps: float32 partial sums of two per pass, flushed to float64 every four passes.  The terms for the lower quadcopter
    // of a pair travel to it in `trav` (see wave_ror1), k descending; a lane receives 31 of them as one float32 sum.
    f2 keep = {0.f, 0.f};
    float trav = 0.f;
    double dkeep = 0.;
    const auto mag2 = [&](f2 d2, f2 rz) { return downwash_mag2_pk(d2, rz, dr); };
    // (Round 5 experiment, removed: the two selects per pair as integer masks -- (0 - bits) >> 31, ands, v_bitop3 -- instead of
    // v_cmp + s_and + v_cndmask: bit-identical, 7 compares left of 164 in the loop, and 0.9 us per step SLOWER at 4096 envs.)
    const auto mine_of = [](float m, float rz, float d2) { return (rz > 0.f && d2 < 100.f) ? m : 0.f; };  // the neighbour is above: this lane's term
    const auto theirs_of = [](float m, float rz, float d2) { return (rz < 0.f && d2 < 100.f) ? m : 0.f; }; // below: the neighbour's (0 when dz == 0)
    // The tile reads run ONE PASS AHEAD of the arithmetic (round 5): the `asm volatile` below ends a scheduling region, and the
    // compiler issued a pass's three ds_read2_b32 and waited for them on the spot.  Reads return in order; the wait for a pass's
    // operands now falls a whole pass after their issue (inside the noise on its own: 22.18 against 22.26 us per step; together
    // with regions of four passes, MRS_DW_REGION, -0.25 us).
    const auto rd = [&](int k, f2 &x, f2 &y, f2 &z) { x = f2{t[k], t[k + 1]}; y = f2{t[128 + k], t[128 + k + 1]}; z = f2{tz[k], tz[k + 1]}; };
    f2 nx, ny, nz;
    {   // k = 31, and the antipode k = 32, which both ends evaluate (each keeps its own term)
        f2 cx, cy, cz;
        rd(31, cx, cy, cz);
        rd(29, nx, ny, nz);
        const f2 rx = pk_sub(cx, mx), ry = pk_sub(cy, my), rz = pk_sub(cz, mz);
        const f2 d2 = pk_fma(ry, ry, pk_mul(rx, rx));
        const f2 m = mag2(d2, rz);
        keep = f2{mine_of(m.x, rz.x, d2.x), mine_of(m.y, rz.y, d2.y)};
        trav = theirs_of(m.x, rz.x, d2.x);
    }
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const int k = 29 - 2 * j;
        const f2 cx = nx, cy = ny, cz = nz;
        if (j < 14) rd(k - 2, nx, ny, nz);
        hook(j);
        const f2 rx = pk_sub(cx, mx), ry = pk_sub(cy, my), rz = pk_sub(cz, mz);
        const f2 d2 = pk_fma(ry, ry, pk_mul(rx, rx));
        if (!SKIP || downwash_pass_live(d2, rz, dr)) {
            const f2 m = mag2(d2, rz);
            trav = f32add(wave_ror1(trav), theirs_of(m.y, rz.y, d2.y)); // k + 1
            trav = f32add(wave_ror1(trav), theirs_of(m.x, rz.x, d2.x)); // k
            keep = keep + f2{mine_of(m.x, rz.x, d2.x), mine_of(m.y, rz.y, d2.y)};
        } else trav = wave_ror1(wave_ror1(trav)); // two exact zeros on board: the word still travels its two lanes
#ifndef MRS_DW_REGION
#define MRS_DW_REGION 4 // passes per scheduling region of the pair loop (A/B): the asm statement below ends a region
#endif
        if (MRS_DW_REGION > 0 && (j % MRS_DW_REGION) == MRS_DW_REGION - 1) asm volatile("" : "+v"(keep)); // formed here, not sunk to the end of the loop
        if ((j & 3) == 3) { dkeep += (double)f32add(keep.x, keep.y); keep = f2{0.f, 0.f}; }
    }
    dkeep += (double)f32add(keep.x, keep.y);
    return -(dkeep + (double)wave_ror1(trav));
}

// Envs of N = 128, 192, 256 (B = N/64 waves, one 64-agent BLOCK of the env per wave, each block with an N = 64 tile of its
// own): a wave runs downwash_ring64 on its own block and takes the pairs between blocks from here.  Lane l meets the
// other block's agents (l + k) mod 64, k = K-1 .. 0 (tp = that block's tile + l + o, o = the lane offset of k = 0): its own
// terms are summed as in downwash_ring64, the other block's terms travel (wave_ror1) and end up, after the k = 0 pass, in
// the lane whose number is the receiving agent's (less o) -- ONE LDS word per lane handed to the other wave after the
// loop, where the ring over all N agents (below, any other N) hands over one word per pair.
//   K = 64: every pair of the two blocks (blocks b and b+1; the wave of b does them all)
//   K = 32: half of them (blocks b and b + B/2, B even: the lower block takes k = 0..31, the upper one, with o = 1, k = 1..32
//           of ITS lanes = the other half)
// Returns the sum of the terms this lane keeps (magnitudes), trav_out = the sum for agent (lane + o) mod 64 of the other block.
template <int K>
__device__ __forceinline__ double downwash_cross(const float *tp, float mx_, float my_, float mz_, const DownwashRegs &dr, float &trav_out)
{
    const f2 mx = splat(mx_), my = splat(my_), mz = splat(mz_);
    const auto mine_of = [](float m, float rz, float d2) { return (rz > 0.f && d2 < 100.f) ? m : 0.f; };
    const auto theirs_of = [](float m, float rz, float d2) { return (rz < 0.f && d2 < 100.f) ? m : 0.f; };
    float trav = 0.f;
    double dkeep = 0.;
    const float *t = tp + (K - 8);
#pragma unroll 1
    for (int it = 0; it < K / 8; ++it, t -= 8) { // eight lane distances = four packed passes per trip, k descending
        TILE64_Z(t);
        f2 keep = {0.f, 0.f};
#pragma unroll
        for (int k = 6; k >= 0; k -= 2) {
            f2 rx, ry, rz;
            tile64_rel2(t, k, mx, my, mz, rx, ry, rz);
            const f2 d2 = pk_fma(ry, ry, pk_mul(rx, rx));
            if (!MRS_DW_SKIP || downwash_pass_live(d2, rz, dr)) {
                const f2 m = downwash_mag2_pk(d2, rz, dr);
                trav = f32add(wave_ror1(trav), theirs_of(m.y, rz.y, d2.y)); // k + 1
                trav = f32add(wave_ror1(trav), theirs_of(m.x, rz.x, d2.x)); // k
                keep = keep + f2{mine_of(m.x, rz.x, d2.x), mine_of(m.y, rz.y, d2.y)};
            } else trav = wave_ror1(wave_ror1(trav));
            asm volatile("" : "+v"(keep));
        }
        dkeep += (double)f32add(keep.x, keep.y);
    }
    trav_out = trav;
    return dkeep;
}

// Quad-quad contact of lane i with one other agent (oracle/mrs_oracle.c:pair_contact, the same float32 operations): the
// centres' difference r = p_i - p_j, both unconstrained velocities as float32; adds this lane's half of the correction.
__device__ __forceinline__ void pair_contact_term(const StepArgs &A, float rx, float ry, float rz, float ux, float uy, float uz, float dv[3])
{
    const float d2 = f32fma(rz, rz, f32fma(ry, ry, f32mul(rx, rx)));
    if (!(d2 <= A.pair_rc2) || !(d2 > 0.f)) return;
    const float d = f32sqrt(d2), rd = f32div(1.0f, d);
    const float nx = f32mul(rx, rd), ny = f32mul(ry, rd), nz = f32mul(rz, rd);
    const float vn = f32fma(uz, nz, f32fma(uy, ny, f32mul(ux, nx)));
    const float gap = f32sub(d, A.pair_r2);
    const float rhs = f32sub(-vn, f32mul(gap, gap > 0.f ? A.pair_inv_dt : A.pair_erp_dt));
    if (rhs > 0.f) {
        const float h = f32mul(0.5f, rhs);
        dv[0] = f32fma(h, nx, dv[0]); dv[1] = f32fma(h, ny, dv[1]); dv[2] = f32fma(h, nz, dv[2]);
    }
}

// ------------------------------------------------------------------------------------ step kernel
#ifndef MRS_DEFER_LOADS
#define MRS_DEFER_LOADS 1 // measured at 512-thread workgroups: 27.5 -> 27.2 us per step (time to the first position 7.8k -> 6.7k ticks)
#endif
#ifndef MRS_LATE_LOADS
#define MRS_LATE_LOADS 1
#endif
#ifndef MRS_MIN_WAVES
#define MRS_MIN_WAVES 1 // __launch_bounds__ 2nd argument = minimum waves per SIMD (caps VGPRs at 512/this)
#endif
// Wave priorities by phase, fused kernel: LAGGARDS FIRST.  The four waves of a SIMD do identical work; at equal
// priority the arbiter lets the oldest run ahead, the waves finish one after another and the SIMD spends the
// second half of the kernel with one or two waves left -- too few to hide latency (clock64 stamps: waves ended
// between 42k and 79k ticks of a 79k-tick kernel).  Priority falling with progress (pair loop 2, controller 1,
// pose/outputs 0) keeps them abreast: 34.3 -> 31.8 us per step.  The contact solve, the one serial stretch that
// three waves of its workgroup wait for, runs at 3.  Round 2: all four levels in use (pair loop 3, controller 2,
// forces 1, outputs 0): 29.2 -> 28.7 us; equal priorities cost +2.5 us, leaders-first +2.8 us (tools/abl_run.sh).
// End of round 3, envs of at most one wave (N <= 64): the ladder compressed to 3 / 3 / 2 / 1 / 1 (pair loop, controller, forces,
// pose + stores, adjacency) -- 22.18 -> 21.68 us per step over twelve windows on one box, -0.1 ... -0.3 on three others
// (tools/abl_run.sh; 3/3/1/0, 3/2/2/0, 3/2/1/1 no better than before, 3/3/3/0 +1.8, rising priorities +2.6).  Envs of several
// waves keep 3 / 2 / 1 / 0 / 0: N = 256 x 1024 envs 45.8 against 47.2 us with the compressed ladder.
#ifndef MRS_P_DW1
#define MRS_P_DW1 3
#endif
#ifndef MRS_P_CTRL
#define MRS_P_CTRL 2
#endif
#ifndef MRS_P_FORCE
#define MRS_P_FORCE 1 // rotor forces, aerodynamics, velocity integration (after the controller)
#endif
#ifndef MRS_P_TAIL
#define MRS_P_TAIL 0
#endif
#ifndef MRS_P_ADJ
#define MRS_P_ADJ 0
#endif
// Round 4: with the sweep cap back at 10 the solve is a third longer, and the steep ladder wins for one-wave envs too: 3 / 2 / 1 / 0 / 0
// against round 3's 3 / 3 / 2 / 1 / 1 over five interleaved rounds of three 400-step windows on two boxes: 22.48 against 22.79 us per
// step (means; best windows 21.73 against 22.15), 3 / 3 / 2 / 0 / 0: 22.81, 3 / 2 / 2 / 0 / 0: 23.22, 3 / 3 / 1 / 0 / 0: 22.88, 3 / 2 / 1 / 1 / 1: 22.92,
// 3 / 2 / 0 / 0 / 0: 22.90, 3 / 1 / 1 / 0 / 0: 22.98, 3 / 3 / 3 / 1 / 1: 24.3, 3 / 3 / 2 / 2 / 2: 23.0 (profiles/r04_ab.txt).  The macros stay apart for A/B builds.
#ifndef MRS_P1_CTRL
#define MRS_P1_CTRL 2 // N <= 64
#endif
#ifndef MRS_P1_FORCE
#define MRS_P1_FORCE 1
#endif
#ifndef MRS_P1_TAIL
#define MRS_P1_TAIL 0
#endif
#ifndef MRS_P1_ADJ
#define MRS_P1_ADJ 0
#endif
// s_setprio takes an immediate: the choice between the two ladders is a (uniform) branch
#define MRS_SETPRIO(one_wave, p1, pn) do { if (one_wave) __builtin_amdgcn_s_setprio(p1); else __builtin_amdgcn_s_setprio(pn); } while (0)
#ifndef MRS_P_SOLVE
#define MRS_P_SOLVE 3
#endif
#ifndef MRS_REST_SHORTCUT
#define MRS_REST_SHORTCUT 1 // A/B switch of round 3 (tools/abl_build.sh): 0 sends every grounded body through the sweeps
#endif
#ifndef MRS_FUSED_WAVES
// the fused kernels are held to 128 VGPRs = 4 resident waves per SIMD = all 1024 workgroups of the bench swarm
// resident at once (set_target_vel / _pos would take 132 / 134 uncapped and drop to 3: measured 43.2 vs 37.5 us)
#define MRS_FUSED_WAVES 4
#endif
// FUSED = true (256-thread workgroups): the whole of MRS.step in ONE launch -- the workgroup resolves its own
// grounded bodies (compacted through LDS onto its first waves), then integrates poses and emits the newest
// observation slice and adjacency rows from registers.  A dependent launch costs ~3 us on this stack and the
// three-launch form pays it three times (measured: tools/micro/launch_floor.hip).
// FUSED = false: velocities only; grounded bodies are queued for k_contact, observation/adjacency follow in
// k_observe_adj (N_AGENTS > 256, or MRS_STEP_SPLIT=1).
// Kernel-argument preload (round 5): the first 14 dwords of the argument segment -- the sizes and the pointers of the first loads,
// handed over as scalar parameters of their own AHEAD of the struct that holds them too -- arrive in SGPRs with the wave
// (-mllvm -amdgpu-kernarg-preload-count=14, gfx950 user SGPRs: 2 for the segment pointer + 14), so that the action and position
// loads are issued without a scalar-cache round trip first: every wave of the launch starts on a cold scalar cache, and the
// timeline stamps put "arguments arrived" 0.75 us after the wave's start.  The struct's own copies serve everything later.
#define MRS_STEP_PRE int pE, int pN, int pT, int pEPB, const float *pact, const double *ppos, const double *pquat, const vel_t *pvel, const vel_t *pangvel
#define MRS_STEP_PRE_ARGS(S) (S).E, (S).N, (S).T, (S).epb, (S).actions, (S).b.pos, (S).b.quat, VELP((S).b.vel), VELP((S).b.angvel)
// N64 (round 5): the instantiation for N_AGENTS = 64 (one env per wave, BLOCK / 64 envs per workgroup): N, the envs per workgroup and
// the row width are compile-time constants, the index division is a shift and the branches of the other layouts (generic N, envs
// of several waves, the ring) are not in the kernel's text at all -- 14 000 lines of assembly with them, the path of an N = 64
// wave threaded through it.
template <int ACT, int BLOCK, bool FUSED, bool N64 = false>
__global__ __launch_bounds__(BLOCK, (BLOCK <= 256 ? (FUSED ? MRS_FUSED_WAVES : MRS_MIN_WAVES) : (FUSED ? 4 : 1))) void k_step(MRS_STEP_PRE, const StepArgs A)
{
    const int AN = N64 ? 64 : pN, AEPB = N64 ? BLOCK / 64 : pEPB, AW = N64 ? 1 : A.W;
    extern __shared__ float4 lds_tile[]; // BLOCK positions (doubled for N = 64), then one int flag per env slot
    int *nanflag = reinterpret_cast<int *>(lds_tile + 2 * BLOCK); // [256] read once (the NaN vote below) and DEAD afterwards:
                                                                  // adjacency_blocks reuses it in the tail as its third exchange array
    int *ncontact = nanflag + 256; // bodies of this workgroup that need the contact solve
    // N = 64 layout: each env's 64 positions are stored TWICE back to back (128 slots per env) so that
    // "neighbour (lane + k) mod 64" is the un-wrapped slot lane + k: a constant LDS offset per unrolled k
    const bool n64 = N64 || (AN == 64);

    const int tid = threadIdx.x;
#ifdef MRS_TIMELINE // diagnostic build (tools/probes/timeline_probe.py): lane 0 of every wave stamps clock64() at the phase
                    // boundaries into the rpm buffer, 16 floats per wave, instead of the rotor speeds
    const long long t_start = clock64();
    float *tl = A.b.rpm ? A.b.rpm + ((size_t)blockIdx.x * (BLOCK / 64) + (tid >> 6)) * 16 : nullptr;
#define TL(k) do { if (tl && (tid & 63) == 0) tl[k] = (float)(clock64() - t_start); } while (0)
    if (tl && (tid & 63) == 0) tl[11] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFF); // wave start on the 100 MHz chip-wide clock
    if (tl && (tid & 63) == 0) { // where the wave runs: HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID (tools/probes/timeline_simd.py)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        tl[13] = (float)(hw & 0xFFFF); tl[14] = (float)(hw >> 16); tl[15] = (float)(xcc & 0xF);
    }
#elif defined(MRS_MARKS) // analysis build: phase boundaries as comments in the assembly (tools/probes/isa_sections.py)
#define TL(k) asm volatile("; MRS_MARK " #k)
#else
#define TL(k) do { } while (0)
#endif
    const int el = tid / AN;
    const int i = tid - el * AN;
    const int e = blockIdx.x * AEPB + el;
    const bool live = (el < AEPB) && (e < pE);
    const size_t T = (size_t)pT;
    const size_t a = live ? (size_t)e * AN + i : 0;
    const unsigned la = live ? (unsigned)tid : 0u;                                        // a == wg_base + la
    const size_t wb_base = (size_t)blockIdx.x * (size_t)AEPB * (size_t)AN;
    WgBuffers wb;
    wb.pos = const_cast<double *>(ppos) + wb_base; wb.quat = const_cast<double *>(pquat) + wb_base; // the preloaded copies
    wb.vel = const_cast<vel_t *>(pvel) + wb_base; wb.angvel = const_cast<vel_t *>(pangvel) + wb_base;
    constexpr int ADIM = (ACT == MRS_ACT_SET_SPEEDS || ACT == MRS_ACT_SET_CONTROL) ? 4 : 3;

    double p[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, v[3] = {0, 0, 0}, w[3] = {0, 0, 0};
    float act[4] = {0, 0, 0, 0};
    if (tid < AEPB) nanflag[tid] = 0;    // N != 64: read after the tile barrier below
    if (!FUSED) {                         // LDS slot counter of the three-launch path's compaction
        if (tid == 0) *ncontact = 0;
        if (n64) __syncthreads();         // visible before any wave runs ahead; taken here, before a load is in flight
    }
    // MRS_LATE_ACT (round 5, N = 64 one-launch kernel): the launch opens with every wave of the chip asking for its actions and
    // positions at once -- 9.4 MB that nothing can be computed beside (the wave's first arithmetic needs the positions).  The
    // action is only needed by the NaN vote and the controller: its load rides in the pair loop like the rest of the state
    // (MRS_LATE_LOADS) and the vote follows the loop (a NaN env has walked its pairs for nothing; its state is untouched).
#ifndef MRS_LATE_ACT
#define MRS_LATE_ACT 1
#endif
    constexpr bool late_act = MRS_LATE_ACT && MRS_LATE_LOADS && N64 && FUSED && !MRS_EXACT_F32 && ACT != MRS_ACT_NONE;
    const float *const ap = pact + (size_t)blockIdx.x * (size_t)AEPB * (size_t)AN * ADIM;
    if (live) {
        // Issue order = the order the data is needed in (loads return in order and the waits are counted): the
        // action (NaN vote) and the position (LDS tile, pair loop) first, so that the quaternion and the velocities
        // are still in flight while the pair loop runs instead of being waited for up front.
        if (ACT != MRS_ACT_NONE && !late_act) {
#pragma unroll
            for (int k = 0; k < ADIM; ++k) act[k] = ap[la * ADIM + k];
        }
#if MRS_DEFER_LOADS
        p[0] = wb.pos[la]; p[1] = (wb.pos + T)[la]; p[2] = (wb.pos + 2 * T)[la];
#else
        load_state(wb, la, T, p, q, v, w);
#endif
    }
    TL(9); // kernel arguments arrived, first loads issued
#if MRS_DEFER_LOADS
    __builtin_amdgcn_sched_barrier(0);
#endif
    {   // the other buffers' workgroup bases come from the struct (scalar loads): formed behind the first loads, not ahead of them
        const WgBuffers w2 = wg_buffers(A, wb_base);
        wb.pid = w2.pid; wb.obs = w2.obs; wb.rpm = w2.rpm; wb.adj = w2.adj;
    }
    double downwash_acc = 0;
    // N = 64 (one env per wave): the three-array tile of the packed pair loop (tile64_write); any other N, and the
    // MRS_EXACT_F32 build's all-pairs loop: one float4 per agent
    const bool tile_soa = n64 && !MRS_EXACT_F32;
    // An env that spans several waves (64 < N <= 256, fused kernel):
    //  - N = 128, 192, 256: one N = 64 tile per wave (= per 64-agent block of the env), see downwash_cross;
    //  - any other N: the symmetric ring below on the same kind of tile for the whole env -- three arrays of 2N floats, every
    //    coordinate twice, so that neighbours k and k+1 of a lane are one ds_read2_b32 at an un-wrapped index (6N floats per
    //    env; epb * N <= BLOCK, and the tile region holds 8 * BLOCK).
    const bool multi = !MRS_EXACT_F32 && FUSED && !n64 && ACT != MRS_ACT_NONE && AN > 64;
    const bool blk = multi && (AN & 63) == 0;
    const bool ring = multi && !blk;
    float *const ring_x = reinterpret_cast<float *>(lds_tile) + el * 6 * AN;
    if (tile_soa || blk) tile64_write(lds_tile, tid >> 6, tid & 63, (float)p[0], (float)p[1], (float)p[2]);
    else if (ring) {
        if (live) {
            ring_x[i] = ring_x[AN + i] = (float)p[0]; ring_x[2 * AN + i] = ring_x[3 * AN + i] = (float)p[1];
            ring_x[4 * AN + i] = ring_x[5 * AN + i] = (float)p[2];
        }
    } else lds_tile[tid] = make_float4((float)p[0], (float)p[1], (float)p[2], 0.f);
    TL(10); // positions arrived and staged
    // MRS_LATE_LOADS (N = 64): the ten remaining state words are not even issued here but one per pass of the pair loop
    // (downwash_ring64's hook).  All 16 waves of a CU reach this point together and the CU's address unit takes ~16
    // cycles per 512-byte load instruction: issued in one go, the last wave waits ~2.5k cycles (clock64 stamps) for
    // its tenth load to be ACCEPTED before it can start the loop; spread out, the loads ride along with the arithmetic.
    const bool late_loads = MRS_LATE_LOADS && FUSED && tile_soa && ACT != MRS_ACT_NONE;
#if MRS_DEFER_LOADS
    // The rest of the state is first needed by the controller: issued only now, behind the positions of EVERY wave
    // (all 4096 waves issue at once; issued up front, these 21 MB would be served before the last wave's position)
    __builtin_amdgcn_sched_barrier(0);
    if (live && !late_loads) {
        q[0] = wb.quat[la]; q[1] = (wb.quat + T)[la]; q[2] = (wb.quat + 2 * T)[la]; q[3] = (wb.quat + 3 * T)[la];
        v[0] = wb.vel[la]; v[1] = (wb.vel + T)[la]; v[2] = (wb.vel + 2 * T)[la];
        w[0] = wb.angvel[la]; w[1] = (wb.angvel + T)[la]; w[2] = (wb.angvel + 2 * T)[la];
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (n64) wave_lds_sync(); else __syncthreads();
    // MRS.py:247-248: any NaN in the env's action aborts that env's step
    bool env_nan = false;
    if (ACT != MRS_ACT_NONE && !late_act) {
        bool bad = false;
#pragma unroll
        for (int k = 0; k < ADIM; ++k) bad |= isnan(act[k]);
        if (n64) {
            env_nan = __builtin_amdgcn_ballot_w64(live && bad) != 0; // the wave is the env
        } else {
            if (live && bad) nanflag[el] = 1;
            __syncthreads();
            env_nan = (el < AEPB) && nanflag[el];
        }
    }
    TL(0); // loads, tile, NaN vote
    const bool masked = live && A.mask && !A.mask[e];
    bool doit = live && !masked && !env_nan;
    if (!late_act && live && i == 0 && env_nan && A.b.status) atomicOr(&A.b.status[e], MRS_STATUS_NAN_ACTION);

    // (Round 3 experiment, removed: the waves of an env that the previous step flagged for quad-quad contact -- pair terms and a
    // second adjacency pass on top, and a launch lasts as long as its slowest workgroup -- at the top priority through every
    // phase: N = 256 x 1024 envs 48.5 against 47.7 us, N = 64 x 4096: 21.7 against 21.4.)
    if (FUSED) __builtin_amdgcn_s_setprio(MRS_P_DW1);
    int my_slot = -1;
    bool parked = false;
#if !MRS_EXACT_F32
    // ---- downwash, envs that span several waves (64 < N <= 256, fused kernel): the pair term is symmetric (see the N = 64
    // loop below), so every unordered pair is evaluated once and the term handed to the lower quadcopter.
    // Envs of whole waves (blk): per 64-agent block, the other agent's terms travelling in a register and ONE LDS word per
    // lane handed to the other wave after the loop (downwash_cross; N = 256 x 1024 envs: 63.0 -> 54.9 us against the ring).
    if (blk && KO_KEEP(8)) {
        const int lane = tid & 63, wt = tid >> 6, nb = AN >> 6, b = i >> 6; // wt - b = the tile of the env's first block
        float *const my_tile = tile64(lds_tile, wt);
        // the terms from the other waves arrive in the 128 floats a tile leaves free behind its three arrays
        if (doit) {
            const float mx = (float)p[0], my = (float)p[1], mz = (float)p[2];
            downwash_acc = downwash_ring64<(MRS_DW_SKIP != 0)>(my_tile + lane, mx, my, mz, lane << 2, A.dc, [](int) {});
            const DownwashRegs dr = downwash_regs(A.dc);
            if (nb >= 3) { // every pair with the next block
                float *const ot = tile64(lds_tile, wt - b + (b + 1 == nb ? 0 : b + 1));
                float trav;
                downwash_acc -= downwash_cross<64>(ot + lane, mx, my, mz, dr, trav);
                ot[384 + lane] = trav;
            }
            if (!(nb & 1)) { // half of the pairs with the block opposite, which takes the other half
                const int hb = nb >> 1, o = b >= hb ? 1 : 0;
                float *const ot = tile64(lds_tile, wt - b + (b >= hb ? b - hb : b + hb));
                float trav;
                downwash_acc -= downwash_cross<32>(ot + lane + o, mx, my, mz, dr, trav);
                ot[448 + ((lane + o) & 63)] = trav;
            }
        }
        __syncthreads();
        if (doit) {
            if (nb >= 3) downwash_acc -= (double)my_tile[384 + lane];
            if (!(nb & 1)) downwash_acc -= (double)my_tile[448 + lane];
        }
    }
    // Any other N (ring): lane i evaluates the pairs (i, i+k) at ring distance k = 1..(N-1)/2 -- across waves through an LDS
    // exchange buffer laid over the not yet used state stash: R distances per workgroup barrier, double-buffered, received
    // terms added in distance order (deterministic).  Half the transcendental work of the all-pairs loop (N = 256 x 1024
    // envs: 83.4 -> 77.2 us, round 2).  Every thread runs the loop: the barriers are workgroup-wide and the trip count depends
    // on N only.  (Round 3 experiment, removed: the handed-over terms ADDED into one LDS word per receiving agent -- 64-bit
    // fixed point, so that the sum does not depend on the order of arrival -- instead of laid out per ring distance and
    // summed after a barrier per eight distances: N = 256 x 1024 envs 66.9 against 61.8 us per step, N = 128: 40.8 against 38.1.)
    constexpr int RING_R = 8;
    if (ring && KO_KEEP(8)) {
        float *xb = reinterpret_cast<float *>(ncontact + 2 + BLOCK); // [2][RING_R][BLOCK] floats inside sp[13][BLOCK] doubles
        const float *tx = ring_x + i, *ty = ring_x + 2 * AN + i, *tz = ring_x + 4 * AN + i; // neighbour i + k at offset k, no wrap
        const DownwashRegs dr = downwash_regs(A.dc);
        const f2 mx = splat((float)p[0]), my = splat((float)p[1]), mz = splat((float)p[2]);
        const int half = (AN - 1) / 2;
        for (int k0 = 1; k0 <= half; k0 += RING_R) {
            float *buf = xb + (((k0 - 1) / RING_R) & 1) * (RING_R * BLOCK);
            float acc32 = 0.f;
#pragma unroll
            for (int r = 0; r < RING_R; r += 2) { // two ring distances per pass on the packed float32 instructions
                const int k = k0 + r;
                if (k <= half && doit) {
                    const f2 rx = pk_sub(f2{tx[k], tx[k + 1]}, mx), ry = pk_sub(f2{ty[k], ty[k + 1]}, my), rz = pk_sub(f2{tz[k], tz[k + 1]}, mz);
                    const f2 d2 = pk_fma(ry, ry, pk_mul(rx, rx));
                    const f2 m = downwash_mag2_pk(d2, rz, dr);
                    int j = i + k;
                    j = j >= AN ? j - AN : j;
                    {   // the neighbour is above: the term is this lane's; below: the neighbour's (0 when dz == 0)
                        const bool near = d2.x < 100.f;
                        acc32 += (rz.x > 0.f && near) ? -m.x : 0.f;
                        buf[r * BLOCK + el * AN + j] = (rz.x < 0.f && near) ? -m.x : 0.f;
                    }
                    if (k + 1 <= half) {
                        const int j1 = (j + 1 == AN) ? 0 : j + 1;
                        const bool near = d2.y < 100.f;
                        acc32 += (rz.y > 0.f && near) ? -m.y : 0.f;
                        buf[(r + 1) * BLOCK + el * AN + j1] = (rz.y < 0.f && near) ? -m.y : 0.f;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RING_R; ++r)
                if (k0 + r <= half && doit) acc32 += buf[r * BLOCK + tid];
            downwash_acc += (double)acc32;
        }
        if (!(AN & 1) && doit) { // antipodal pair: evaluated by both ends, each keeping its own term
            const int k = AN / 2;
            const float rx = tx[k] - (float)p[0], ry = ty[k] - (float)p[1], dz = tz[k] - (float)p[2];
            downwash_acc += (double)(dz > 0.f ? downwash_mag2(__builtin_fmaf(ry, ry, rx * rx), dz, dr.dw2, dr.dw3, dr.lg) : 0.f);
        }
        __syncthreads(); // the exchange buffer is the state stash of the contact phase
    }
#endif
    if (late_loads && live && !doit) { // a masked or NaN-action env keeps its state: loaded here for the outputs
        q[0] = wb.quat[la]; q[1] = (wb.quat + T)[la]; q[2] = (wb.quat + 2 * T)[la]; q[3] = (wb.quat + 3 * T)[la];
        v[0] = wb.vel[la]; v[1] = (wb.vel + T)[la]; v[2] = (wb.vel + 2 * T)[la];
        w[0] = wb.angvel[la]; w[1] = (wb.angvel + T)[la]; w[2] = (wb.angvel + 2 * T)[la];
    }
    if (late_act && doit) { // (no MRS_KO bit 8 here) the pair loop ahead of the NaN vote (`doit` is uniform per wave: every lane is live)
        const int lane = tid & 63;
        downwash_acc = downwash_ring64<false>(tile64(lds_tile, tid >> 6) + lane, (float)p[0], (float)p[1], (float)p[2], lane << 2, A.dc, [&](int j) {
            switch (j) {
            case 0: q[0] = wb.quat[la]; break;
            case 1: q[1] = (wb.quat + T)[la]; break;
            case 2: q[2] = (wb.quat + 2 * T)[la]; break;
            case 3: q[3] = (wb.quat + 3 * T)[la]; break;
            case 4: v[0] = wb.vel[la]; break;
            case 5: v[1] = (wb.vel + T)[la]; break;
            case 6: v[2] = (wb.vel + 2 * T)[la]; break;
            case 7: w[0] = wb.angvel[la]; break;
            case 8: w[1] = (wb.angvel + T)[la]; break;
            case 9: w[2] = (wb.angvel + 2 * T)[la]; break;
            case 10:
#pragma unroll
                for (int k = 0; k < ADIM; ++k) act[k] = ap[la * ADIM + k];
                break;
            default: break;
            }
        });
        bool bad = false;
#pragma unroll
        for (int k = 0; k < ADIM; ++k) bad |= isnan(act[k]);
        env_nan = __builtin_amdgcn_ballot_w64(bad) != 0; // MRS.py:247-248; the wave is the env
        if (i == 0 && env_nan && A.b.status) atomicOr(&A.b.status[e], MRS_STATUS_NAN_ACTION);
        doit = !env_nan;
    }
    if (doit) {
        V3 fb = v3(0., 0., 0.), tb = v3(0., 0., 0.);
        M3 Rb; // quat_to_matrix_bullet(q): prop heights of the ground effect, then the velocity integration
        // ---- downwash (Quadcopter.py:99-115): O(N) broadcast reads of the env's LDS tile per lane.
        // Runs FIRST, while only the 13 state words are live: the controller's registers (PID memory,
        // rotation matrices) do not have to survive the 64-iteration loop, which is what keeps the
        // kernel at 4 resident waves per SIMD.
        if (ACT != MRS_ACT_NONE && !multi && KO_KEEP(8)) {
            const float4 *tile_env = lds_tile + el * AN;
            const DownwashConst &dc = A.dc;
            const float mx = (float)p[0], my = (float)p[1], mz = (float)p[2];
#if !MRS_EXACT_F32
            if (AN == 64) {
                // N = 64: the env is exactly this wave.  The pair term depends only on (|dz|, dxy^2) and lands
                // on the LOWER quadcopter of the pair, so each unordered pair is evaluated once: lane i takes
                // the pairs (i, i+k), k = 1..31, keeps the term if the other is above, and hands it to lane
                // i+k (one ds_bpermute) if the other is below; k = 32 pairs lanes with their antipode, each
                // side evaluating its own.  32 evaluations per lane instead of 64.  (`doit` is uniform per
                // env, so the whole wave is here.)
                const int lane = tid & 63;
                // tile + lane: neighbour (lane + k) mod 64 at offset k, no wrap; lane << 2 = ds_bpermute byte address of this lane
                if (!late_act) downwash_acc = downwash_ring64<false>(tile64(lds_tile, el) + lane, mx, my, mz, lane << 2, A.dc, [&](int j) {
#ifdef MRS_P_DW2 // A/B: the second half of the pair loop one level down
                    if (FUSED && j == 7) __builtin_amdgcn_s_setprio(MRS_P_DW2);
#endif
                    if (!late_loads) return;
                    // (the controller memory the outer loop of the cascade wants right after this loop was tried here too,
                    // passes 10..14: no gain, 27.2 against 27.2 us per step, and spills in set_target_pos)
                    switch (j) { // `doit` is uniform per env = per wave, so every lane here is live: la == tid
                    case 0: q[0] = wb.quat[la]; break;
                    case 1: q[1] = (wb.quat + T)[la]; break;
                    case 2: q[2] = (wb.quat + 2 * T)[la]; break;
                    case 3: q[3] = (wb.quat + 3 * T)[la]; break;
                    case 4: v[0] = wb.vel[la]; break;
                    case 5: v[1] = (wb.vel + T)[la]; break;
                    case 6: v[2] = (wb.vel + 2 * T)[la]; break;
                    case 7: w[0] = wb.angvel[la]; break;
                    case 8: w[1] = (wb.angvel + T)[la]; break;
                    case 9: w[2] = (wb.angvel + 2 * T)[la]; break;
                    default: break;
                    }
                });
            } else
#endif
            {
#pragma unroll 4
                for (int j = 0; j < AN; ++j) {
                    const float4 pj = tile_env[j];
#if MRS_EXACT_F32
                    const float f = downwash_pair(f32sub(pj.x, mx), f32sub(pj.y, my), f32sub(pj.z, mz), dc.pr32, dc.dw1, dc.dw2, dc.dw3);
#else
                    const float f = downwash_pair_fast(pj.x - mx, pj.y - my, pj.z - mz, dc);
#endif
                    downwash_acc += (double)f;
                }
            }
        }
        TL(1); // pair loop
        if (FUSED) MRS_SETPRIO(AN <= 64, MRS_P1_CTRL, MRS_P_CTRL);
        if (ACT != MRS_ACT_NONE) {
            const MrsParams &P = A.P;
            double rpm[4];
            constexpr bool NEEDS_PID = (ACT >= MRS_ACT_TARGET_ACCEL);
            // Outer loops of the cascade first (QuadControl.pos_control / vel_control up to target_accel):
            // they need only the float32 velocity/position read-back, so their 6..12 PID planes are loaded,
            // used and stored BEFORE the register-hungry attitude part -- this is what lets
            // k_step<set_target_vel> fit the 128 VGPRs of 4 resident waves per SIMD.
            Pid s = {};
            V3 ta = v3(0., 0., 0.);
            if (ACT == MRS_ACT_TARGET_VEL || ACT == MRS_ACT_TARGET_POS) {
                Observed o0;
                observe<false, false>(p, q, v, w, o0);
                // controller memory: 16-byte records (MrsBuffers.pid), one load / store instruction per record
                float4 *g = wb.pid + la;
                if (ACT == MRS_ACT_TARGET_POS) {
                    const float4 r0 = g[0];
                    s.ipx = r0.x; s.ipy = r0.y; s.ipz = r0.z;
                    ta = pos_control_accel(P, s, o0, act[0], act[1], act[2]);
                    g[0] = make_float4((float)s.ipx, (float)s.ipy, (float)s.ipz, 0.f);
                } else {
                    const float4 r1 = g[T], r2 = g[2 * T], r3 = g[3 * T];
                    s.dvx = r1.x; s.dvy = r1.y; s.dvz = r1.z;
                    s.ivx = r1.w; s.ivy = r2.x; s.ivz = r2.y;
                    s.lvx = r2.z; s.lvy = r2.w; s.lvz = r3.x;
                    s.ltx = r3.y; s.lty = r3.z; s.ltz = r3.w;
                    ta = vel_control_accel(P, A.rc, s, o0, act[0], act[1], act[2]);
                    if (KO_KEEP(4)) {
                        g[T] = make_float4((float)s.dvx, (float)s.dvy, (float)s.dvz, (float)s.ivx);
                        g[2 * T] = make_float4((float)s.ivy, (float)s.ivz, s.lvx, s.lvy);
                        g[3 * T] = make_float4(s.lvz, s.ltx, s.lty, s.ltz);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TL(20); // outer loop of the cascade (pos/vel control)
            Observed ob;
            M3 R; // from_euler(float32 euler read-back): only the PID modes use it
            if (NEEDS_PID) {
                observe_ctrl(p, q, v, w, ob, R, A.P.round_euler_readback != 0);
            } else observe<true, true>(p, q, v, w, ob);
            TL(21); // read-back + rotation matrices
            if (NEEDS_PID) {
                float4 *g = wb.pid + la;
                const float4 r4 = g[4 * T];
                s.iox = r4.x; s.ioy = r4.y; s.ioz = r4.z;
                if (ACT == MRS_ACT_TARGET_ORI) { // Quadcopter.py:63-65
                    const M3 Rt = euler_to_matrix((double)act[0], (double)act[1], (double)act[2]);
                    attitude_control(P, A.rc, s, Rt, R, ob, v3(0., 0., 9.81), 9.81, 1.0 / 9.81, rpm);
                } else {
                    if (ACT == MRS_ACT_TARGET_ACCEL) ta = v3((double)act[0], (double)act[1], (double)act[2]);
                    accel_control(P, A.rc, s, ta, R, ob, rpm);
                }
                if (KO_KEEP(4)) g[4 * T] = make_float4((float)s.iox, (float)s.ioy, (float)s.ioz, 0.f);
            } else if (ACT == MRS_ACT_SET_CONTROL) {
                set_control(P, act[0], act[1], act[2], act[3], rpm);
            } else {
                rpm[0] = act[0]; rpm[1] = act[1]; rpm[2] = act[2]; rpm[3] = act[3];
            }
#ifndef MRS_TIMELINE
            if (A.b.rpm) {
#pragma unroll
                for (int k = 0; k < 4; ++k) (wb.rpm + k * T)[la] = (float)rpm[k];
            }
#endif
            TL(2); // controller
#ifdef MRS_P_FORCE
            if (FUSED) MRS_SETPRIO(AN <= 64, MRS_P1_FORCE, MRS_P_FORCE);
#endif
            // ---- Quadcopter.set_speeds (Quadcopter.py:38-45): rotor thrusts + yaw reaction torque.
            // With ACTION_TYPE=set_speeds the reference's arithmetic is float32 (float32 action tensor).
            double F[4], sumw, zt;
            float s32[4] = {act[0], act[1], act[2], act[3]};
            if (ACT == MRS_ACT_SET_SPEEDS) {
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sq = f32mul(s32[k], s32[k]);
                    F[k] = (double)f32mul(sq, (float)P.kf);
                    t[k] = f32mul(sq, (float)P.km);
                }
                zt = (double)f32add(f32sub(f32add(-t[0], t[1]), t[2]), t[3]);
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = f32add(acc, f32div(f32mul((float)(2 * kPi), s32[k]), 60.f));
                sumw = (double)acc;
            } else {
                double t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double sq = rpm[k] * rpm[k];
                    F[k] = sq * P.kf; t[k] = sq * P.km;
                }
                // sum_k 2 pi rpm_k / 60 (Quadcopter.py:90) with the constant factored out: one multiply instead
                // of four float64 divisions (~56 VALU); differs from the term-by-term sum by < 2 ulp
                sumw = (((rpm[0] + rpm[1]) + rpm[2]) + rpm[3]) * (2 * kPi / 60);
                zt = ((-t[0] + t[1]) - t[2]) + t[3];
            }
            // ---- Quadcopter.dynamics ground effect (Quadcopter.py:70-87)
            Rb = quat_to_matrix_bullet(q[0], q[1], q[2], q[3]);
            const bool gnd_on = (ob.roll < (float)(kPi / 2)) && (ob.pitch < (float)(kPi / 2)); // :80 np.abs(bool)
            const double gcoef = gnd_on ? P.gnd_eff_coeff : 0.0; // switched once, not per rotor
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double h = p[2] + (Rb.m20 * P.prop_x[k] + Rb.m21 * P.prop_y[k] + Rb.m22 * P.prop_z[k]);
                h = h < A.hclip ? A.hclip : h; // :77
                const double ratio = P.prop_radius * rcp64_coarse(4 * h); // h >= hclip > 0; 1e-14 relative on a term that is ~1e-4 of the thrust
                double g;
                if (ACT == MRS_ACT_SET_SPEEDS) {
                    const float sq = f32mul(s32[k], s32[k]);
                    g = (double)f32mul(f32mul(sq, (float)P.kf), (float)gcoef) * (ratio * ratio);
                } else {
                    g = ((rpm[k] * rpm[k]) * P.kf) * gcoef * (ratio * ratio);
                }
                const double f = F[k] + g;
                fb.z += f;
                tb.x += P.prop_y[k] * f;   // r x (0,0,f) at the prop link COM (cf2x.urdf:42-78)
                tb.y += -P.prop_x[k] * f;
            }
            tb.z += zt;
            // ---- drag (Quadcopter.py:88-98): R32 (c .* v32), applied in LINK_FRAME => used as a body-frame force
            {
                const double t0 = (-1 * P.drag_xy * sumw) * (double)ob.vx, t1 = (-1 * P.drag_xy * sumw) * (double)ob.vy,
                             t2 = (-1 * P.drag_z * sumw) * (double)ob.vz;
                fb.x += (double)ob.r00 * t0 + (double)ob.r01 * t1 + (double)ob.r02 * t2;
                fb.y += (double)ob.r10 * t0 + (double)ob.r11 * t1 + (double)ob.r12 * t2;
                fb.z += (double)ob.r20 * t0 + (double)ob.r21 * t1 + (double)ob.r22 * t2;
            }
            fb.z += downwash_acc;
        }
        TL(22); // rotor forces, ground effect, drag
        if (ACT == MRS_ACT_NONE) Rb = quat_to_matrix_bullet(q[0], q[1], q[2], q[3]);
        integrate_velocity(A.P, A.rc, Rb, v, w, fb, tb);
    }
    // ---- quad-quad contact on the unconstrained velocities (only envs whose flag is set; see StepArgs.pair_flag).  The
    // agents' float32 velocities go through the position tile's spare space: N = 64 and the multi-wave ring keep every
    // coordinate twice (the second copies are free once the downwash loop is through), the float4 tile has its
    // second half.  An env is a wave at N = 64 (wave-local sync); otherwise the branch is made workgroup-uniform.
    if (A.pair_flag != nullptr) {
        // did the adjacency pass over the positions this step started from see a pair of this env within contact range?
        // (loading the flag earlier -- after the downwash loop, after the controller -- was measured: no difference)
        const int pf = doit ? A.pair_flag[e] : 0;
        const bool mine = doit && pf != 0;
        const bool any = n64 ? (__builtin_amdgcn_ballot_w64(mine) != 0) : (__syncthreads_or(mine) != 0);
        if (any) {
            float *const t64 = tile64(lds_tile, tid >> 6); // N = 64: the env's tile; blk: this wave's block of it
            if (tile_soa || blk) { const int ln = tid & 63; t64[64 + ln] = (float)v[0]; t64[192 + ln] = (float)v[1]; t64[320 + ln] = (float)v[2]; }
            else if (ring) { if (live) { ring_x[AN + i] = (float)v[0]; ring_x[3 * AN + i] = (float)v[1]; ring_x[5 * AN + i] = (float)v[2]; } }
            else lds_tile[BLOCK + tid] = make_float4((float)v[0], (float)v[1], (float)v[2], 0.f);
            if (n64) wave_lds_sync(); else __syncthreads();
            const float px = (float)p[0], py = (float)p[1], pz = (float)p[2];
            const float vx = (float)v[0], vy = (float)v[1], vz = (float)v[2];
            float dv[3] = {0.f, 0.f, 0.f};
            if (tile_soa && (pf & MRS_PAIR_SINGLE)) { // one pair in the env (the flag is the wave's): its two agents, one term each
                const int i0 = pf & 63, j0 = (pf >> 8) & 63;
                if (i == i0 || i == j0) {
                    const int j = i == i0 ? j0 : i0;
                    pair_contact_term(A, f32sub(px, t64[j]), f32sub(py, t64[128 + j]), f32sub(pz, t64[256 + j]),
                                      f32sub(vx, t64[64 + j]), f32sub(vy, t64[192 + j]), f32sub(vz, t64[320 + j]), dv);
                }
            } else if (mine) {
                // the terms of a lane are added in ascending order of the partner's index, as the oracle's loop does
                auto partner = [&](int j) {
                    float qx, qy, qz, wx, wy, wz;
                    if (tile_soa) { qx = t64[j]; qy = t64[128 + j]; qz = t64[256 + j]; wx = t64[64 + j]; wy = t64[192 + j]; wz = t64[320 + j]; }
                    else if (blk) { const float *tj = t64 + ((j >> 6) - (i >> 6)) * 512 + (j & 63); qx = tj[0]; qy = tj[128]; qz = tj[256]; wx = tj[64]; wy = tj[192]; wz = tj[320]; }
                    else if (ring) { qx = ring_x[j]; qy = ring_x[2 * AN + j]; qz = ring_x[4 * AN + j]; wx = ring_x[AN + j]; wy = ring_x[3 * AN + j]; wz = ring_x[5 * AN + j]; }
                    else { const float4 a = lds_tile[el * AN + j], b = lds_tile[BLOCK + el * AN + j]; qx = a.x; qy = a.y; qz = a.z; wx = b.x; wy = b.y; wz = b.z; }
                    pair_contact_term(A, f32sub(px, qx), f32sub(py, qy), f32sub(pz, qz), f32sub(vx, wx), f32sub(vy, wy), f32sub(vz, wz), dv);
                };
                if (pf == MRS_PAIR_ROWS) { // the adjacency pass noted every agent's partners: one term per set bit
                    const unsigned long long *rw = A.pair_rows + (wb_base + la) * (size_t)AW;
                    for (int wd = 0; wd < AW; ++wd) {
                        unsigned long long bits = rw[wd];
                        while (bits) {
                            partner(wd * 64 + __builtin_ctzll(bits));
                            bits &= bits - 1;
                        }
                    }
                } else { // positions set from outside since the last adjacency pass: every other agent of the env
                    for (int j = 0; j < AN; ++j)
                        if (j != i) partner(j);
                }
            }
            v[0] += (double)dv[0]; v[1] += (double)dv[1]; v[2] += (double)dv[2];
            if (!n64) __syncthreads(); // the tile's second half / second copies are somebody else's scratch from here on
        }
    }
    if (doit) {
        TL(3); // forces + velocity integration
        // near the ground; a body lying flat at rest is finished here, in its own lane (contact_at_rest)
        const bool grounded = needs_contact(A.P.enable_contact, A.park_z, p[2]) && !(MRS_REST_SHORTCUT && A.P.rest_shortcut && contact_at_rest(A.P, A.rc, p[2], q, v, w));
        if (FUSED) {
            parked = grounded;
        } else if (grounded) {
            // near the ground: queue the body for k_contact (compacted: the solver's cost scales with the
            // number of grounded bodies, and its registers stay out of this kernel).  Its pre-step pose and
            // unconstrained velocities travel in the slot-indexed planes of contact_state, so k_contact
            // reads them coalesced without first chasing the list entry.
            my_slot = atomicAdd(ncontact, 1); // LDS counter: slot inside this workgroup's reservation
        } else {
            integrate_pose(A.P, p, q, v, w);
            store_state(wb, la, T, p, q, v, w);
        }
    }
    if (FUSED) {
        // ---- contact: every lane stashes its state in LDS planes; the grounded bodies' lane ids are compacted
        // into clist and solved by the first ceil(n/64) waves reading/writing those planes.  Nothing but the
        // stash is live across the solve, so its registers and the controller's never coexist.
        // Compaction without atomics or a zeroed counter: each wave ranks its own grounded lanes (ballot + mbcnt)
        // into its own 64-entry segment of clist and publishes the count; the solver walks the segments in wave
        // order, so the list -- and with it the order the bodies are solved in -- is the same on every run.
        constexpr int NW = BLOCK / 64;
        int *clist = ncontact + 2;
        double *sp = reinterpret_cast<double *>(clist + BLOCK); // [13][BLOCK]
        int *wcnt = reinterpret_cast<int *>(sp + 13 * BLOCK);   // [NW] per-wave counts
        sp[tid] = p[0]; sp[BLOCK + tid] = p[1]; sp[2 * BLOCK + tid] = p[2];
        sp[3 * BLOCK + tid] = q[0]; sp[4 * BLOCK + tid] = q[1]; sp[5 * BLOCK + tid] = q[2]; sp[6 * BLOCK + tid] = q[3];
        sp[7 * BLOCK + tid] = v[0]; sp[8 * BLOCK + tid] = v[1]; sp[9 * BLOCK + tid] = v[2];
        sp[10 * BLOCK + tid] = w[0]; sp[11 * BLOCK + tid] = w[1]; sp[12 * BLOCK + tid] = w[2];
        {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(parked);
            const int lane = tid & 63;
            if (parked) clist[(tid & ~63) + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = tid;
            if (lane == 0) wcnt[tid >> 6] = __builtin_popcountll(m);
        }
        __syncthreads();
        TL(4); // barrier 1
        int wend[NW]; // running end of each wave's segment in the concatenated list
        int n = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { n += wcnt[k]; wend[k] = n; }
        // (Round 2 experiment, removed: a wave without a share of the solve finishing its free-flying lanes -- pose, store,
        // observation -- instead of waiting at the barrier below.  No gain, 28.0 against 27.9 us: it then runs the
        // tail twice, once for the free-flying and once for the grounded lanes.)
        if (n > 0) { // uniform over the workgroup
            // The solving wave is the workgroup's critical path (the other waves wait for it at the barrier below) but
            // shares its SIMD with three waves of other workgroups that are still in their issue-bound forces
            // phase: at equal priority its dependent chain advances one instruction per ~17 cycles.  Raised
            // priority lets it issue whenever it is ready.
            __builtin_amdgcn_s_setprio(MRS_P_SOLVE);
            // the solver's view of a listed body: its float64 velocities in the LDS stash, read whenever they are wanted and changed in
            // place (nothing float64 is live across the sweeps)
            struct StashBody {
                double *sp; int b;
                __device__ __forceinline__ void load(V3 &vv, V3 &ww) const
                {
                    vv = v3(sp[7 * BLOCK + b], sp[8 * BLOCK + b], sp[9 * BLOCK + b]); ww = v3(sp[10 * BLOCK + b], sp[11 * BLOCK + b], sp[12 * BLOCK + b]);
                }
                __device__ __forceinline__ void add(const V3 &dv, const V3 &dw) const
                {
                    sp[7 * BLOCK + b] += dv.x; sp[8 * BLOCK + b] += dv.y; sp[9 * BLOCK + b] += dv.z;
                    sp[10 * BLOCK + b] += dw.x; sp[11 * BLOCK + b] += dw.y; sp[12 * BLOCK + b] += dw.z;
                }
            };
            if (const int sl = tid; sl < n) { // n <= BLOCK: every lane lists at most itself
                int seg = 0;
#pragma unroll
                for (int k = 0; k + 1 < NW; ++k) seg += (sl >= wend[k]);
                int start = 0;
#pragma unroll
                for (int k = 0; k + 1 < NW; ++k) start = (seg == k + 1) ? wend[k] : start;
                const int b = clist[seg * 64 + (sl - start)];
                const double qq[4] = {sp[3 * BLOCK + b], sp[4 * BLOCK + b], sp[5 * BLOCK + b], sp[6 * BLOCK + b]};
                float *dgp = nullptr;
#ifdef MRS_TIMELINE // per-body sweep diagnostics in the pid planes 0/1 (tools/probes/sweeps_probe.py); the run's physics is void
                float dg[2] = {0.f, 0.f};
                dgp = dg;
#endif
                contact_solve_rows<float>(A.P, A.rc, sp[2 * BLOCK + b], qq, StashBody{sp, b}, dgp);
#ifdef MRS_TIMELINE
                if (A.b.pid) wb.pid[b] = make_float4(dg[0], dg[1], 0.f, 0.f);
#endif
            }
            __builtin_amdgcn_s_setprio(0);
            TL(5); // own share of the contact solve
            __syncthreads();
        }
        auto reload = [&]() {
            p[0] = sp[tid]; p[1] = sp[BLOCK + tid]; p[2] = sp[2 * BLOCK + tid];
            q[0] = sp[3 * BLOCK + tid]; q[1] = sp[4 * BLOCK + tid]; q[2] = sp[5 * BLOCK + tid]; q[3] = sp[6 * BLOCK + tid];
            v[0] = sp[7 * BLOCK + tid]; v[1] = sp[8 * BLOCK + tid]; v[2] = sp[9 * BLOCK + tid];
            w[0] = sp[10 * BLOCK + tid]; w[1] = sp[11 * BLOCK + tid]; w[2] = sp[12 * BLOCK + tid];
        };
        auto finish = [&]() { // pose, state planes, newest observation slice (MRS.py:255-256) of this lane
            if (doit) {
                integrate_pose(A.P, p, q, v, w);
                if (KO_KEEP(1)) store_state(wb, la, T, p, q, v, w);
            }
            if (A.b.obs && live && A.n_obs > 0 && KO_KEEP(2)) {
                // (Round 3 experiment, removed: the wave's 64 observation rows of 24 bytes staged through the free position tile and
                // stored as 16-byte whole-line pieces instead of three 8-byte stores at a 24-byte stride: 22.4 against 22.4 us.)
                write_obs(A, wb.obs + la * (unsigned)A.D, p, q, v, w);
            }
        };
        // (Round 5 experiment, removed: at N = 64 every wave solving its own listed bodies in their own lanes through the stash -- no
        // list, no barrier, no wave waiting for another one's bodies; bit-identical results.  24.9 against 22.3 us per step in a
        // same-process A/B: 2.5 solving waves per workgroup instead of one, and the vector slots they take are not idle ones.)
        // (Round 3 experiment, removed: the lanes that are not listed for the solve -- 19 out of 20 -- finishing their step
        // between the two barriers, under the solver wave's serial chain, and the listed ones behind it.  25.9 against 23.4 us
        // per step: two exec-masked passes over the 13 state stores write partial cache lines instead of whole ones.)
        TL(6); // barrier 2
        MRS_SETPRIO(AN <= 64, MRS_P1_TAIL, MRS_P_TAIL); // last phase, lowest priority: see MRS_P_* above
        reload();
        // (the mirror image of MRS_LATE_LOADS -- the 13 state stores spread over the passes of the adjacency pair loop
        // instead of one burst ahead of it -- was measured: no gain, 27.0 against 27.0 us per step)
        finish();
        const float fpx = (float)p[0], fpy = (float)p[1], fpz = (float)p[2];
        // ---- newest observation slice + adjacency rows of the post-step state (MRS.py:255-257)
        TL(7); // pose + store
        if (MRS_P_ADJ != MRS_P_TAIL || MRS_P1_ADJ != MRS_P1_TAIL) MRS_SETPRIO(AN <= 64, MRS_P1_ADJ, MRS_P_ADJ);
        // (fetching the last phase's scalar arguments ahead of the second barrier was measured: no gain, 29.4 vs 29.6 us)
        // (Round 5 experiment, removed: the N = 64 adjacency pass AHEAD of the hand-off, on p + dt v -- the final position of every body
        // the contact rows do not touch -- with the rows and columns of the listed bodies put right in the tail from their final
        // positions (one ballot per listed body); bit-identical rows, no scratch: 24.6 against 22.2 us per step.  The phases between
        // the pair loop and the hand-off are where four waves per SIMD compete for issue slots; 300 more instructions there cost
        // more than the tail saves.)
        // (Round 3 experiment, removed: the N = 64 adjacency pass taken out of this tail and run between the two barriers of the
        // hand-off on predicted positions -- final for every lane that is not listed -- with the listed agents' rows and columns
        // put right afterwards, and the solver waves' envs looked after by the waves next in line.  Bit-identical rows; 23.8 us
        // per step against 23.3 with eight envs per workgroup, 23.0 with four: the "idle" slots of a workgroup in its hand-off
        // are the other resident workgroup's, there was nothing to fill.)
        if ((A.do_adj || A.pair_flag != nullptr) && KO_KEEP(16)) {
            // (the env index is formed again from an opaque copy of the thread index: kept live from the top of the kernel
            // it was the one register too many across the contact sweeps)
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            const int e2 = blockIdx.x * AEPB + (n64 ? (t2 >> 6) : el);
            adjacency_phase<BLOCK, N64>(A, A.d2_thresh, A.comm_inf != 0 || !A.do_adj, lds_tile, tid, el, i, live,
                                         A.do_adj ? wb.adj + la * (unsigned)AW : nullptr,
                                         make_float4(fpx, fpy, fpz, 0.f), e2);
        }
        TL(8); // observation + adjacency
#ifdef MRS_TIMELINE
        if (tl && (tid & 63) == 0) tl[12] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFF);
#endif
        return;
    }
    // Two-level compaction into one global list: lanes take slots from an LDS counter, ONE lane per
    // workgroup reserves the range with a single device-scope atomic (a per-wave atomic on one word
    // serialises 4096 returning atomics at ~12 ns each and dominated the kernel once a third of the
    // swarm was on the ground).
    __syncthreads();
    if (tid == 0) {
        const int n = *ncontact;
        ncontact[1] = n ? atomicAdd(A.contact_count, n) : 0;
        if (blockIdx.x == 0) *A.contact_count_next = 0; // next step's counter (this one is read by k_contact)
    }
    __syncthreads();
    if (my_slot >= 0) {
        const size_t slot = (size_t)(ncontact[1] + my_slot);
        A.contact_list[slot] = (int)a;
        double *cs = A.contact_state + slot;
        cs[0] = p[0]; cs[T] = p[1]; cs[2 * T] = p[2];
        cs[3 * T] = q[0]; cs[4 * T] = q[1]; cs[5 * T] = q[2]; cs[6 * T] = q[3];
        cs[7 * T] = v[0]; cs[8 * T] = v[1]; cs[9 * T] = v[2];
        cs[10 * T] = w[0]; cs[11 * T] = w[1]; cs[12 * T] = w[2];
    }
}

// Contact pass over the compacted list written by k_step: ground contact + pose integration of the
// queued bodies (BulletSim.step_sim's constraint solve + integrateTransforms for those bodies).
#ifndef MRS_CONTACT_WAVES
#define MRS_CONTACT_WAVES 2 // measured: capping VGPRs for 4+ resident waves spills the unrolled solver (4x slower)
#endif
__global__ __launch_bounds__(256, MRS_CONTACT_WAVES) void k_contact(const StepArgs A)
{
    const int count = *A.contact_count;
    const size_t T = (size_t)A.T;
    // fixed grid + stride: an empty list costs one quick wave per workgroup, a full one fills the chip
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += gridDim.x * blockDim.x) {
        const size_t a = (size_t)A.contact_list[idx];
        const double *cs = A.contact_state + idx;
        double p[3] = {cs[0], cs[T], cs[2 * T]}, q[4] = {cs[3 * T], cs[4 * T], cs[5 * T], cs[6 * T]};
        double v[3] = {cs[7 * T], cs[8 * T], cs[9 * T]}, w[3] = {cs[10 * T], cs[11 * T], cs[12 * T]};
        contact_stage(A.P, A.rc, p, q, v, w);
        integrate_pose(A.P, p, q, v, w);
        store_state(A.b, a, T, p, q, v, w);
    }
}

// standalone observe / adjacency (reset()/set() -> calc_Xk, DataGenerator's calc_Ak after reset)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_observe_adj(const StepArgs A)
{
    extern __shared__ float4 lds_tile[];
    const int tid = threadIdx.x;
    const int el = tid / A.N;
    const int i = tid - el * A.N;
    const int e = blockIdx.x * A.epb + el;
    const bool live = (el < A.epb) && (e < A.E);
    const unsigned la = live ? (unsigned)tid : 0u;
    const WgBuffers wb = wg_buffers(A, (size_t)blockIdx.x * (size_t)A.epb * (size_t)A.N);
    double p[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, v[3] = {0, 0, 0}, w[3] = {0, 0, 0};
    if (live) { // only the planes the observation spec names are moved (cat(pos, vel): 48 of the 104 bytes)
        const size_t T = (size_t)A.T;
        bool nq = false, nv = false, nw = false;
        for (int f = 0; f < A.n_obs; ++f) {
            nq |= (A.obs_fields[f] == MRS_OBS_EULER) || (A.obs_fields[f] == MRS_OBS_QUAT);
            nv |= (A.obs_fields[f] == MRS_OBS_VEL);
            nw |= (A.obs_fields[f] == MRS_OBS_ANGVEL);
        }
        p[0] = wb.pos[la]; p[1] = (wb.pos + T)[la]; p[2] = (wb.pos + 2 * T)[la];
        if (nq) { q[0] = wb.quat[la]; q[1] = (wb.quat + T)[la]; q[2] = (wb.quat + 2 * T)[la]; q[3] = (wb.quat + 3 * T)[la]; }
        if (nv) { v[0] = wb.vel[la]; v[1] = (wb.vel + T)[la]; v[2] = (wb.vel + 2 * T)[la]; }
        if (nw) { w[0] = wb.angvel[la]; w[1] = (wb.angvel + T)[la]; w[2] = (wb.angvel + 2 * T)[la]; }
    }
    if (A.b.obs && live && A.n_obs > 0) write_obs(A, wb.obs + la * (unsigned)A.D, p, q, v, w);
    if (A.do_adj || A.pair_flag != nullptr)
        adjacency_phase<BLOCK>(A, A.d2_thresh, A.comm_inf != 0 || !A.do_adj, lds_tile, tid, el, i, live, A.do_adj ? wb.adj + la * (unsigned)A.W : nullptr,
                               make_float4((float)p[0], (float)p[1], (float)p[2], 0.f), e);
}

// packed (M,N,W) -> dense float32 (M,N,N).  N % 4 == 0: one thread per four consecutive columns (a 16-byte store;
// a wave writes 1 KB contiguous), grid-stride so that the launch stays a few thousand workgroups; otherwise one
// thread per element.  Pure streaming: 4 N^2 bytes written per matrix.
__global__ void k_adj_expand4(const uint64_t *__restrict__ packed, float4 *__restrict__ dense4, int N, int W, size_t total4)
{
    const int n4 = N >> 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / n4;
        const int j = (int)(idx - row * n4) << 2;
        const unsigned nib = (unsigned)(packed[row * W + (j >> 6)] >> (j & 63)) & 15u; // j % 4 == 0: the four bits share a word
        dense4[idx] = make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)(nib >> 3));
    }
}
__global__ void k_adj_expand(const uint64_t *__restrict__ packed, float *__restrict__ dense, int N, int W, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t row = idx / N;
    const int j = (int)(idx - row * N);
    const uint64_t bits = packed[row * W + (j >> 6)];
    dense[idx] = (float)((bits >> (j & 63)) & 1ull);
}

// Environment.set_state / Object.set_state (Environment.py:97-103, Object.py:42-65)
struct SetArgs {
    MrsBuffers b;
    const float *pos, *ori, *vel, *angvel;
    const double *pos64, *quat64, *vel64, *angvel64;
    const uint8_t *mask;
    int ori_kind, N;
    size_t T;
};
__global__ void k_set_state(const SetArgs S)
{
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= S.T) return;
    const size_t T = S.T;
    if (S.mask && !S.mask[a / S.N]) return;
    if (S.pos) for (int k = 0; k < 3; ++k) S.b.pos[k * T + a] = (double)S.pos[a * 3 + k];
    if (S.vel) for (int k = 0; k < 3; ++k) VELP(S.b.vel)[k * T + a] = (vel_t)S.vel[a * 3 + k];
    if (S.angvel) for (int k = 0; k < 3; ++k) VELP(S.b.angvel)[k * T + a] = (vel_t)S.angvel[a * 3 + k];
    if (S.pos64) for (int k = 0; k < 3; ++k) S.b.pos[k * T + a] = S.pos64[a * 3 + k];
    if (S.vel64) for (int k = 0; k < 3; ++k) VELP(S.b.vel)[k * T + a] = (vel_t)S.vel64[a * 3 + k];
    if (S.angvel64) for (int k = 0; k < 3; ++k) VELP(S.b.angvel)[k * T + a] = (vel_t)S.angvel64[a * 3 + k];
    if (S.quat64) for (int k = 0; k < 4; ++k) S.b.quat[k * T + a] = S.quat64[a * 4 + k];
    if (S.ori) {
        double q[4];
        if (S.ori_kind == MRS_ORI_EULER) { // Object.py:54-56
            euler_to_quat((double)S.ori[a * 3], (double)S.ori[a * 3 + 1], (double)S.ori[a * 3 + 2], q);
        } else if (S.ori_kind == MRS_ORI_QUAT) { // passthrough
            for (int k = 0; k < 4; ++k) q[k] = (double)S.ori[a * 4 + k];
        } else { // Object.py:51-53 R.from_matrix(ori).as_quat(): Shepperd's method on the given matrix
            const float *m = S.ori + a * 9;
            const double m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
            const double tr = m00 + m11 + m22;
            if (tr >= m00 && tr >= m11 && tr >= m22) {
                q[3] = 1 + tr; q[0] = m21 - m12; q[1] = m02 - m20; q[2] = m10 - m01;
            } else if (m00 >= m11 && m00 >= m22) {
                q[0] = 1 - tr + 2 * m00; q[1] = m10 + m01; q[2] = m20 + m02; q[3] = m21 - m12;
            } else if (m11 >= m22) {
                q[1] = 1 - tr + 2 * m11; q[0] = m10 + m01; q[2] = m21 + m12; q[3] = m02 - m20;
            } else {
                q[2] = 1 - tr + 2 * m22; q[0] = m20 + m02; q[1] = m21 + m12; q[3] = m10 - m01;
            }
            const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (int k = 0; k < 4; ++k) q[k] /= n;
        }
        for (int k = 0; k < 4; ++k) S.b.quat[k * T + a] = q[k];
    }
}

__global__ void k_pid_reset(MrsBuffers b, const uint8_t *mask, int N, size_t T)
{
    const size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= T) return;
    if (mask && !mask[a / N]) return;
    float4 *g = reinterpret_cast<float4 *>(b.pid) + a;
    const float nan = __builtin_nanf(""); // last_vel_e / last_target_vel: "attribute not created yet" (QuadControl.py:55-59)
    g[0] = make_float4(0.f, 0.f, 0.f, 0.f); g[T] = make_float4(0.f, 0.f, 0.f, 0.f); g[2 * T] = make_float4(0.f, 0.f, nan, nan);
    g[3 * T] = make_float4(nan, nan, nan, nan); g[4 * T] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------ spawn
// Counter-based RNG: splitmix64 over (seed, global env, agent, draw#) -> independent of sharding.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t seed, uint64_t env, uint32_t agent, uint32_t draw)
{
    const uint64_t h = splitmix64(splitmix64(seed ^ (env * 0xD1342543DE82EF95ull)) ^ ((uint64_t)agent << 32 | draw));
    return ((h >> 40) + 0.5f) * (1.0f / 16777216.0f); // (0,1)
}

struct SpawnArgs {
    MrsBuffers b;
    uint64_t seed;
    int64_t env_base;
    const uint8_t *mask;
    int E, N, max_rounds;
    float min_dist;
    float ori_lo[3], ori_hi[3];
    size_t T;
    // user START_POS distributions (MRS.py:127-154 with self.START_POS a torch distribution): the caller draws the
    // samples, (E, cand_rounds, N, 3) float32 -- round r holds what every agent would get if it were re-sampled in
    // round r -- and the greedy rejection runs here.  resume: start from the positions already in the state buffers.
    const float *cand;
    int cand_rounds, resume;
};

// torch's float32 norm over the last axis of get_relative_position (MRS.py:135, :152): sqrt(fma(dz,dz,fma(dy,dy,dx*dx))), the
// chain tests/golden/F3 pins for calc_A; the layouts of tests/golden/F5b are reproduced bit for bit only with it.
__device__ __forceinline__ float spawn_dist(float4 a, float4 b)
{
    const float dx = f32sub(a.x, b.x), dy = f32sub(a.y, b.y), dz = f32sub(a.z, b.z);
    return f32sqrt(f32fma(dz, dz, f32fma(dy, dy, f32mul(dx, dx))));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

// MRS.generate_start_pos (MRS.py:127-154) for the default spawn distribution (MRS.py:69-78) or caller-drawn candidates:
// one workgroup per env, one lane per agent; the layout lives in LDS, every lane keeps its own collision count.
// The greedy pick of MRS.py:140-144 -- torch.mode over the row indices of the remaining collisions = the agent with the most
// of them, the LOWEST index among equals -- is a workgroup-wide maximum of (count << 12 | 4095 - index) per pick (wave
// shuffles + one LDS word per wave), after which every lane that still collides with the pick drops one from its count:
// "collisions[idx,:] = 0; collisions[:,idx] = 0".  (Round 3 ran the picks on ONE thread: 210 us per call at E = 4096, N = 12.)
__global__ void k_spawn(const SpawnArgs S)
{
    extern __shared__ float4 sp[]; // N positions
    __shared__ unsigned s_wmax[2][16];
    const int e = blockIdx.x;
    if (S.mask && !S.mask[e]) return;
    const uint64_t ge = (uint64_t)(S.env_base + e);
    const int tid = threadIdx.x, nw = (blockDim.x + 63) >> 6;
    const bool live = tid < S.N;
    uint32_t draws = 0;
    auto sample = [&](int i) {
        // xy ~ Normal(0,1) pushed through SphereTransform(radius=1, within=True) (Util.py:176-195):
        // points outside the unit disc are pulled onto its rim; z ~ U[1,3]
        const float u1 = u01(S.seed, ge, i, draws), u2 = u01(S.seed, ge, i, draws + 1), u3 = u01(S.seed, ge, i, draws + 2);
        const float r = sqrtf(-2.f * logf(u1));
        float x = r * cosf(6.2831853f * u2), y = r * sinf(6.2831853f * u2);
        const float mag = sqrtf(x * x + y * y);
        if (mag > 1.f) { x /= mag; y /= mag; }
        sp[i] = make_float4(x, y, 1.f + 2.f * u3, 0.f);
    };
    int cand_next = 0; // next unused round of candidates (uniform over the workgroup)
    if (S.cand) {
        if (live) {
            if (S.resume) sp[tid] = make_float4((float)S.b.pos[(size_t)e * S.N + tid], (float)S.b.pos[S.T + (size_t)e * S.N + tid],
                                                (float)S.b.pos[2 * S.T + (size_t)e * S.N + tid], 0.f);
            else { const float *c = S.cand + (((size_t)e * S.cand_rounds) * S.N + tid) * 3; sp[tid] = make_float4(c[0], c[1], c[2], 0.f); }
        }
        cand_next = S.resume ? 0 : 1;
    } else if (live) sample(tid);
    draws += 3;
    __syncthreads();
    int round = 0;
    bool need_more = false, failed = false;
    unsigned it = 0; // picks so far: the parity selects the exchange row, one barrier per pick
    for (;; ++round) {
        int c = 0; // codist < 2*AGENT_RADIUS (MRS.py:135-138): this agent's row of `collisions`
        const float4 me = live ? sp[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int j = 0; j < S.N; ++j) c += (j != tid) && (spawn_dist(me, sp[j]) < S.min_dist);
        if (!__syncthreads_or(c > 0)) break;                    // while torch.any(codist < 2 R), MRS.py:137
        if (round >= S.max_rounds) { failed = true; break; }
        if (S.cand && cand_next >= S.cand_rounds) { failed = need_more = true; break; } // out of samples: the caller draws more and resumes
        bool flagged = false;
        for (;; ++it) { // MRS.py:140-144
            unsigned key = (live && !flagged && c > 0) ? (((unsigned)c << 12) | (unsigned)(4095 - tid)) : 0u;
            key = wave_max_u32(key);
            if (nw > 1) {
                if ((tid & 63) == 0) s_wmax[it & 1][tid >> 6] = key;
                __syncthreads();
                key = 0u;
                for (int k = 0; k < nw; ++k) key = max(key, s_wmax[it & 1][k]);
            }
            if (key == 0u) break;
            const int best = 4095 - (int)(key & 4095u);
            if (tid == best) { flagged = true; c = 0; }
            else if (live && !flagged && c > 0 && spawn_dist(me, sp[best]) < S.min_dist) c--;
        }
        ++it;
        __syncthreads(); // every lane is through its reads of the layout
        if (flagged) { // MRS.py:145-151
            if (S.cand) {
                const float *cd = S.cand + (((size_t)e * S.cand_rounds + cand_next) * S.N + tid) * 3;
                sp[tid] = make_float4(cd[0], cd[1], cd[2], 0.f);
            } else sample(tid);
        }
        cand_next++;
        draws += 3;
        __syncthreads();
    }
    if (tid == 0 && failed && S.b.status) atomicOr(&S.b.status[e], need_more ? MRS_STATUS_SPAWN_MORE : MRS_STATUS_SPAWN_FAIL);
    if (live) {
        const size_t a = (size_t)e * S.N + tid, T = S.T;
        const float4 me = sp[tid];
        S.b.pos[a] = (double)me.x; S.b.pos[T + a] = (double)me.y; S.b.pos[2 * T + a] = (double)me.z;
        if (S.cand) return; // positions only: orientation / velocities follow through mrs_set_state (MRS.reset)
        // MRS.generate_start_ori (MRS.py:157-161): randrange(lo, hi) per axis
        float eul[3];
        for (int k = 0; k < 3; ++k) {
            const float u = u01(S.seed ^ 0xA5A5A5A5ull, ge, tid, 0x40000000u + k);
            eul[k] = S.ori_lo[k] + u * (S.ori_hi[k] - S.ori_lo[k]);
        }
        double q[4];
        euler_to_quat((double)eul[0], (double)eul[1], (double)eul[2], q);
        for (int k = 0; k < 4; ++k) S.b.quat[k * T + a] = q[k];
        for (int k = 0; k < 3; ++k) { VELP(S.b.vel)[k * T + a] = 0; VELP(S.b.angvel)[k * T + a] = 0; }
    }
}

// ------------------------------------------------------------------------------- Reynolds expert
// Reynolds.forward_batch (examples/simulating_data/helper/Reynolds.py:80-110) + Reynolds_Controller.forward_batch
// (Reynolds_Node.py:26-38) for K = 1: all-to-all neighbourhood of the previous step's states.  One lane per
// agent, the env's positions and velocities staged in LDS, float32 with the operation order of
// oracle/mrs_oracle.c:orc_reynolds (bit-identical to it; the reference's torch reductions sum in their own order).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reynolds(const float *__restrict__ x_prev, float *__restrict__ actions, int E, int N, int D, int epb)
{
    extern __shared__ float4 lds_tile[]; // [BLOCK] positions, [BLOCK] velocities
    const int tid = threadIdx.x;
    const int el = tid / N;
    const int i = tid - el * N;
    const int e = blockIdx.x * epb + el;
    const bool live = (el < epb) && (e < E);
    float xi[6] = {0, 0, 0, 0, 0, 0};
    const size_t a = live ? (size_t)e * N + i : 0;
    if (live) {
#pragma unroll
        for (int k = 0; k < 6; ++k) xi[k] = x_prev[a * D + k];
    }
    lds_tile[tid] = make_float4(xi[0], xi[1], xi[2], 0.f);
    lds_tile[BLOCK + tid] = make_float4(xi[3], xi[4], xi[5], 0.f);
    __syncthreads();
    if (!live) return;
    const float4 *P = lds_tile + el * N, *V = lds_tile + BLOCK + el * N;
    float rp[3] = {0, 0, 0}, inv[3] = {0, 0, 0}, rv[3] = {0, 0, 0};
    for (int j = 0; j < N; ++j) {
        if (j == i) continue; // A_ii = 0: every term of the diagonal is an exact zero
        const float4 pj = P[j], vj = V[j];
        const float np0 = f32sub(pj.x, xi[0]), np1 = f32sub(pj.y, xi[1]), np2 = f32sub(pj.z, xi[2]);
        const float nv0 = f32sub(vj.x, xi[3]), nv1 = f32sub(vj.y, xi[4]), nv2 = f32sub(vj.z, xi[5]);
        const float dp = f32sqrt(f32fma(np2, np2, f32fma(np1, np1, f32mul(np0, np0))));
        const float dv = f32sqrt(f32fma(nv2, nv2, f32fma(nv1, nv1, f32mul(nv0, nv0))));
        const float den = f32add(0.0f, f32mul(f32mul(dp, dp), dp)); // (A - 1) + |n|^3 with A = 1
        rp[0] = f32add(rp[0], f32mul(np0, dp)); rp[1] = f32add(rp[1], f32mul(np1, dp)); rp[2] = f32add(rp[2], f32mul(np2, dp));
        inv[0] = f32add(inv[0], f32div(np0, den)); inv[1] = f32add(inv[1], f32div(np1, den)); inv[2] = f32add(inv[2], f32div(np2, den));
        rv[0] = f32add(rv[0], f32mul(nv0, dv)); rv[1] = f32add(rv[1], f32mul(nv1, dv)); rv[2] = f32add(rv[2], f32mul(nv2, dv));
    }
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = f32mul(0.5f, f32add(f32add(rp[k], f32mul(3.0f, -inv[k])), f32mul(3.0f, rv[k])));
    const float no = f32sqrt(f32fma(o[2], o[2], f32fma(o[1], o[1], f32mul(o[0], o[0]))));
    const float mag = f32div(no > 1.0f ? 1.0f : no, no); // torch.clamp(norm, max=1) / norm
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = f32mul(o[k], mag);
        actions[a * 3 + k] = (v != v) ? 0.0f : v; // actions[isnan] = 0 (Reynolds.py:105)
    }
}

// ------------------------------------------------------------------------------------------- host
struct MrsHandle {
    MrsParams P;
    int E, N, device;
    int block, epb, W;
    int sblock;         // workgroup size of the fused step for N = 64: 512 = eight envs (waves) per workgroup (MRS_STEP_BLOCK overrides)
    double hclip;
    int *ws;            // device workspace: [0..1] two alternating contact counters, [2..2+T) contact list
    double *cs;         // device workspace: [13][T] parked states of the listed bodies
    bool fused;         // one-launch step (256-thread workgroups; MRS_STEP_SPLIT=1 keeps the three-launch form)
    unsigned step_parity;
    int *pair_flag;     // device workspace [E]: quad-quad contact candidates per env (StepArgs.pair_flag); all ones = "look"
    unsigned long long *pair_rows; // device workspace [T][W]: the candidates per agent (StepArgs.pair_rows)
    bool big_lds[MRS_ACT_TARGET_ORI + 1]; // hipFuncAttributeMaxDynamicSharedMemorySize raised for this handle's device, per ACTION_TYPE
    bool raycast_big_lds = false; // k_raycast's dynamic-LDS attribute raised on this device (N >= 683)
    // The part of StepArgs that only depends on the parameters, the observation spec and the range (fill_common): kept from one
    // call to the next -- a step of a small swarm (C2: 12 us of kernel) is host-bound, and the threshold search, the logarithm and
    // the divisions of fill_common were ~0.4 us of every call.  Invalidated by mrs_set_params.
    struct StepArgs *cached;
    double cached_range;
    int cached_fields[MRS_OBS_MAX_FIELDS], cached_n_obs;
};

static thread_local char g_err[256] = "";
static int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
static int hipfail(hipError_t e, const char *where)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

// A handle is bound to one device (mrs_create); the caller's current device may be another one.  Every entry
// point that launches selects the handle's device for the duration of the call and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

extern "C" int mrs_abi_version(void) { return MRS_ABI_VERSION; }
extern "C" const char *mrs_last_error(void) { return g_err; }

extern "C" int mrs_params_default(MrsParams *p)
{
    if (!p) return fail(MRS_E_ARG, "mrs_params_default: NULL");
    memset(p, 0, sizeof(*p));
    // cf2x.urdf:5,11,12,34,42-78 (Quadcopter.read_attributes, Quadcopter.py:119-150)
    p->arm = 0.0397; p->kf = 3.16e-10; p->km = 7.94e-12; p->thrust2weight = 2.25;
    p->gnd_eff_coeff = 11.36859; p->prop_radius = 2.31348e-2; p->drag_xy = 9.1785e-7; p->drag_z = 10.311e-7;
    p->dw1 = 2267.18; p->dw2 = .16; p->dw3 = -.11;
    p->mass = 0.027; p->ixx_file = 1.4e-5; p->iyy_file = 1.4e-5; p->izz_file = 2.17e-5;
    const double px[4] = {0.028, -0.028, -0.028, 0.028}, py[4] = {0.028, 0.028, -0.028, -0.028};
    for (int k = 0; k < 4; ++k) { p->prop_x[k] = px[k]; p->prop_y[k] = py[k]; p->prop_z[k] = 0.0; }
    p->coll_radius = 0.06; p->coll_half_len = 0.0125;
    p->gravity = 9.81; p->dt = 0.01; p->ctrl_gravity = 9.81; p->ctrl_dt = 0.01; // BulletSim.py:13-14, :52-57
    // [BULLET-KNOWLEDGE] inertia of the collision hull's AABB (+2 x 0.001 margin), damping 0.04f, clamp 100
    const double lx = 2 * (p->coll_radius + 0.002), lz = 2 * (p->coll_half_len + 0.002);
    p->inertia[0] = p->inertia[1] = p->mass / 12.0 * (lx * lx + lz * lz);
    p->inertia[2] = p->mass / 12.0 * (lx * lx + lx * lx);
    p->lin_damp = (double)0.04f; p->ang_damp = (double)0.04f; p->max_coord_vel = 100.0; p->use_gyro = 1;
    p->ground_z = 0.5; p->friction = 1.5 * 0.5; p->erp = 0.2; p->contact_threshold = 0.02; // plane.urdf:5,24
    p->solver_iters = 10; p->enable_contact = 1; p->pair_contact = 1; p->rest_shortcut = 1;
    return 0;
}

extern "C" int mrs_params_derived(const MrsParams *p, double out[7])
{
    if (!p || !out) return fail(MRS_E_ARG, "mrs_params_derived: NULL");
    const double gf = p->gravity * p->mass; // Quadcopter.py:156-162
    const double hover = sqrt(gf / (4 * p->kf));
    const double maxrpm = sqrt((p->thrust2weight * gf) / (4 * p->kf));
    const double maxthrust = 4. * p->kf * maxrpm * maxrpm;
    out[0] = gf; out[1] = hover; out[2] = maxrpm; out[3] = maxthrust;
    out[4] = sqrt(2.0) * p->arm * p->kf * maxrpm * maxrpm;
    out[5] = 2. * p->km * maxrpm * maxrpm;
    out[6] = 0.25 * p->prop_radius * sqrt((15 * maxrpm * maxrpm * p->kf * p->gnd_eff_coeff) / maxthrust);
    return 0;
}

extern "C" int mrs_adj_words(int n_agents) { return n_agents > 0 ? (n_agents + 63) / 64 : MRS_E_ARG; }

static int obs_width(int f)
{
    switch (f) {
    case MRS_OBS_POS: case MRS_OBS_VEL: case MRS_OBS_EULER: case MRS_OBS_ANGVEL: return 3;
    case MRS_OBS_QUAT: return 4;
    default: return -1;
    }
}
extern "C" int mrs_obs_dim(const int32_t *fields, int n)
{
    if (n < 0 || n > MRS_OBS_MAX_FIELDS || (n > 0 && !fields)) return MRS_E_ARG;
    int d = 0;
    for (int i = 0; i < n; ++i) {
        const int w = obs_width(fields[i]);
        if (w < 0) return MRS_E_ARG;
        d += w;
    }
    return d;
}

extern "C" int mrs_set_params(MrsHandle *h, const MrsParams *params)
{
    if (!h || !params) return fail(MRS_E_ARG, "mrs_set_params: NULL");
    h->P = *params;
    if (h->cached) { delete h->cached; h->cached = nullptr; }
    double d[7];
    mrs_params_derived(params, d);
    h->hclip = d[6];
    return 0;
}

extern "C" int mrs_create(const MrsParams *params, int n_envs, int n_agents, int device, MrsHandle **out)
{
    if (!params || !out) return fail(MRS_E_ARG, "mrs_create: NULL argument");
    if (n_envs < 1 || n_agents < 1) return fail(MRS_E_ARG, "mrs_create: n_envs and n_agents must be >= 1");
    if (n_agents > 1024) return fail(MRS_E_ARG, "mrs_create: N_AGENTS > 1024 not supported (one env per workgroup)");
    if ((size_t)n_envs * (size_t)n_agents > 0x7fffffffull) return fail(MRS_E_ARG, "mrs_create: E*N overflows int32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(MRS_E_NO_DEVICE, "mrs_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MRS_E_ARG, "mrs_create: bad device index");
    MrsHandle *h = new (std::nothrow) MrsHandle();
    if (!h) return fail(MRS_E_ARG, "mrs_create: out of host memory");
    h->E = n_envs; h->N = n_agents; h->device = device;
    if (n_agents <= 256) { h->block = 256; h->epb = 256 / n_agents; }
    else { h->block = 1024; h->epb = 1; }
    h->W = (n_agents + 63) / 64;
    const char *split = getenv("MRS_STEP_SPLIT");
    h->fused = (h->block == 256) && !(split && split[0] == '1');
    // N = 64: eight envs (waves) per workgroup.  Measured at the bench size (tools/probes/abl_sblock.sh, same build, us per step):
    // 64 threads 32.1, 128: 30.5, 256: 29.0, 512: 27.5, 1024: 29.6 -- two workgroups of eight waves per CU pool their
    // grounded bodies over more envs (fewer, fuller solver waves) and put two waves of the same hand-off group on each SIMD
    // ... as long as that still gives every CU a workgroup: a swarm of fewer than 8 x CUs envs takes the largest workgroup
    // that does (1024 envs on 256 CUs: 256 threads 13.9 us per step, 512: 17.5; 512 envs: 128 threads 13.7, 512: 16.0)
    // Round 3: with the bodies at rest finished in their own lanes (contact_at_rest) a workgroup lists a dozen bodies, not a
    // hundred, for its hand-off, and four envs per workgroup (four workgroups per CU) overtake eight: 512 threads 24.7, 256: 24.1,
    // 128: 24.6 us per step in the benchmark's window (tools/steady_bench.py, first repetition).
    h->sblock = 256;
    if (n_agents == 64) {
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        // Round 5, by resident waves per SIMD (one env = one wave): up to one 64 threads, up to two 128, more 256 -- us per step
        // (tools/steady_bench.py, bench action type) at 1024 envs: 64: 14.95, 128: 15.37, 256: 15.09; 2048 envs: 17.35, 16.35, 17.07;
        // 3072 envs: 20.91, 21.31, 18.99; BASELINE config 2 (set_speeds, 1024 envs, no A): 13.60, 14.21, 14.15.  (Before: the largest
        // size that still gave every CU a workgroup, i.e. 256 from 1024 envs on.)
        h->sblock = n_envs <= 4 * ncu ? 64 : (n_envs <= 8 * ncu ? 128 : 256);
    }
    if (const char *sb = getenv("MRS_STEP_BLOCK")) {
        const int v = atoi(sb);
        if ((v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) && n_agents == 64) h->sblock = v; // (the other workgroup sizes exist as N = 64 instantiations only)
    }
    h->cached = nullptr;
    mrs_set_params(h, params);
    // internal workspace (never user-visible): contact counters + compacted contact list
    h->ws = nullptr; h->cs = nullptr; h->step_parity = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if ((e = hipSetDevice(device)) != hipSuccess) { delete h; return hipfail(e, "mrs_create hipSetDevice"); }
    const size_t ws_bytes = (2 + (size_t)n_envs * n_agents) * sizeof(int);
    e = hipSuccess;
    if (!h->fused) { // the one-launch step keeps its contact list and parked states in LDS
        e = hipMalloc((void **)&h->ws, ws_bytes);
        if (e == hipSuccess) e = hipMemset(h->ws, 0, ws_bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize(); // null-stream memset: ordered before any caller stream's first step
        if (e == hipSuccess) e = hipMalloc((void **)&h->cs, 13 * (size_t)n_envs * n_agents * sizeof(double));
    }
    h->pair_flag = nullptr; h->pair_rows = nullptr;
    if (e == hipSuccess) {
        e = hipMalloc((void **)&h->pair_flag, (size_t)n_envs * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&h->pair_rows, (size_t)n_envs * n_agents * (size_t)((n_agents + 63) / 64) * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(h->pair_flag, 1, (size_t)n_envs * sizeof(int)); // nothing known yet: every env looks
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        if (h->ws) (void)hipFree(h->ws);
        if (h->cs) (void)hipFree(h->cs);
        if (h->pair_flag) (void)hipFree(h->pair_flag);
        if (h->pair_rows) (void)hipFree(h->pair_rows);
        (void)hipSetDevice(cur);
        delete h;
        return hipfail(e, "mrs_create workspace");
    }
    (void)hipSetDevice(cur);
    *out = h;
    return 0;
}

extern "C" void mrs_destroy(MrsHandle *h)
{
    if (!h) return;
    if (h->cached) { delete h->cached; h->cached = nullptr; }
    if (h->ws || h->cs || h->pair_flag || h->pair_rows) {
        DeviceGuard dg(h->device);
        if (h->pair_flag) (void)hipFree(h->pair_flag);
        if (h->pair_rows) (void)hipFree(h->pair_rows);
        if (h->ws) (void)hipFree(h->ws);
        if (h->cs) (void)hipFree(h->cs);
    }
    delete h;
}

static float d2_threshold(double comm_range)
{
    // largest float t with sqrtf(t) <= (float)comm_range, so that "sqrt(d2) <= R" == "d2 <= t" exactly
    float r = (float)comm_range;
    if (!(r >= 0.f)) return -1.f;
    // adjacency_phase moves non-finite (diverged) positions to finite ones >= 1e18 m away so that no NaN enters the sign-bit
    // verdicts; a finite range at that scale would reach them (T' = inf, inf - inf = NaN).  Ranges above 1e17 m count as
    // 1e17 m: positions beyond 1e17 m are treated as diverged anyway (same function), and a NaN agent stays adjacent to nobody.
    if (r > 1e17f) r = 1e17f;
    float t = r * r;
    while (sqrtf(nextafterf(t, INFINITY)) <= r) t = nextafterf(t, INFINITY);
    while (t > 0.f && sqrtf(t) > r) t = nextafterf(t, -INFINITY);
    return t;
}

static int fill_common_uncached(MrsHandle *h, const MrsBuffers *b, const int32_t *obs_fields, int n_obs, double comm_range, StepArgs &A);
static int fill_common(MrsHandle *h, const MrsBuffers *b, const int32_t *obs_fields, int n_obs, double comm_range, StepArgs &A)
{
    if (n_obs < 0 || n_obs > MRS_OBS_MAX_FIELDS || (n_obs > 0 && !obs_fields)) return fail(MRS_E_ARG, "bad observation field list");
    bool hit = h->cached != nullptr && h->cached_n_obs == n_obs && (h->cached_range == comm_range || (std::isnan(h->cached_range) && std::isnan(comm_range)));
    for (int i = 0; hit && i < n_obs; ++i) hit = h->cached_fields[i] == obs_fields[i];
    if (!hit) {
        if (!h->cached) h->cached = new (std::nothrow) StepArgs;
        if (!h->cached) return fail(MRS_E_ARG, "out of host memory");
        MrsBuffers none;
        memset(&none, 0, sizeof(none));
        const int rc = fill_common_uncached(h, &none, obs_fields, n_obs, comm_range, *h->cached);
        if (rc) { delete h->cached; h->cached = nullptr; return rc; }
        h->cached_range = comm_range; h->cached_n_obs = n_obs;
        for (int i = 0; i < n_obs; ++i) h->cached_fields[i] = obs_fields[i];
    }
    A = *h->cached;
    A.b = *b;
    // the two members that depend on the call's buffers
    A.n_obs = (n_obs > 0 && !b->obs) ? 0 : n_obs;
    A.do_adj = (b->adj != nullptr) && !std::isnan(comm_range);
    return 0;
}
static int fill_common_uncached(MrsHandle *h, const MrsBuffers *b, const int32_t *obs_fields, int n_obs, double comm_range, StepArgs &A)
{
    memset(&A, 0, sizeof(A));
    A.P = h->P; A.b = *b; A.E = h->E; A.N = h->N; A.T = h->E * h->N; A.epb = h->epb; A.W = h->W; A.hclip = h->hclip;
    A.park_z = h->P.ground_z + std::sqrt(h->P.coll_radius * h->P.coll_radius + h->P.coll_half_len * h->P.coll_half_len) + h->P.contact_threshold;
    A.pair_flag = (h->P.enable_contact && h->P.pair_contact && h->N > 1) ? h->pair_flag : nullptr;
    A.pair_rows = h->pair_rows;
    A.pair_r2 = 2.0f * (float)h->P.coll_radius;
    A.pair_rc2 = (A.pair_r2 + (float)h->P.contact_threshold) * (A.pair_r2 + (float)h->P.contact_threshold);
    A.pair_inv_dt = (float)(1.0 / h->P.dt); A.pair_erp_dt = (float)(h->P.erp / h->P.dt);
    { // Quadcopter.py:103-110 constants of the pair term, float32 like the reference's arithmetic (see DownwashConst)
        DownwashConst &c = A.dc;
        c.pr32 = (float)h->P.prop_radius; c.dw1 = (float)h->P.dw1; c.dw2 = (float)h->P.dw2; c.dw3 = (float)h->P.dw3;
        c.c_alpha = c.dw1 * (0.25f * c.pr32) * (0.25f * c.pr32);
        c.lg_alpha = (float)std::log2((double)c.c_alpha);
        c.zero_c = (150.0f + (c.lg_alpha > 0.f ? c.lg_alpha : 0.f)) * (1.0f / 0.72134752f) * 1.00001f; // downwash_zero_c
    }
    A.rc.inv_mass = 1.0 / h->P.mass; A.rc.inv_i0 = 1.0 / h->P.inertia[0]; A.rc.inv_i1 = 1.0 / h->P.inertia[1];
    A.rc.inv_i2 = 1.0 / h->P.inertia[2]; A.rc.inv_4kf = 1.0 / (4 * h->P.kf); A.rc.inv_dt = 1.0 / h->P.dt;
    {
        const float dt32 = (float)h->P.ctrl_dt;
        uint32_t bits;
        memcpy(&bits, &dt32, sizeof(bits));
        A.rc.inv_ctrl_dt32 = (float)(1.0 / (double)dt32); // RN32(RN64(1 / dt32)) = RN32(1 / dt32): 53 >= 2 * 24 + 2
        A.rc.ctrl_div_fast = std::isnormal(dt32) && (bits & 0x7FFFFFu) != 0x7FFFFFu && std::isnormal(A.rc.inv_ctrl_dt32);
    }
    const int D = mrs_obs_dim(obs_fields, n_obs);
    if (D < 0) return fail(MRS_E_ARG, "bad observation field list");
    A.n_obs = n_obs; A.D = D;
    for (int i = 0; i < n_obs; ++i) { A.obs_fields[i] = obs_fields[i]; A.obs_code |= (unsigned)obs_fields[i] << (4 * i); }
    if (n_obs > 0 && !b->obs) A.n_obs = 0;
    A.do_adj = (b->adj != nullptr) && !std::isnan(comm_range);
    A.comm_inf = std::isinf(comm_range) && comm_range > 0;
    A.d2_thresh = A.comm_inf ? INFINITY : d2_threshold(comm_range);
    return 0;
}

static int launch_observe_adj(MrsHandle *h, const StepArgs &A, hipStream_t st);

template <int ACT>
static hipError_t launch_step(MrsHandle *h, const StepArgs &A, hipStream_t st, bool fused)
{
    const int grid = (h->E + h->epb - 1) / h->epb;
    const size_t lds = 2 * (size_t)h->block * sizeof(float4) + 258 * sizeof(int);
    // fused: + compacted lane list + 13 float64 state planes + per-wave counts (36 888 B; 4 workgroups per CU fit the 160 KB LDS)
    if (fused && h->sblock != 256) { // N = 64 with fewer envs (waves) per workgroup: same kernel, smaller hand-off group
        StepArgs B = A;
        B.epb = std::min(h->sblock / h->N, 256); // whole envs per workgroup (the NaN-flag array holds 256)
        const int g = (h->E + B.epb - 1) / B.epb;
        const size_t l = 2 * (size_t)h->sblock * sizeof(float4) + 258 * sizeof(int) + (size_t)h->sblock * sizeof(int) + 13 * (size_t)h->sblock * sizeof(double) + (size_t)(h->sblock / 64) * sizeof(int);
        if (h->sblock == 128) hipLaunchKernelGGL((k_step<ACT, 128, true, true>), dim3(g), dim3(128), l, st, MRS_STEP_PRE_ARGS(B), B);
        else if (h->sblock == 512) { // 74 KB of LDS per workgroup: above the 64 KB default limit of a launch
            // The attribute belongs to the (function, device) pair and a process may hold handles on several devices: noted per
            // handle (a handle is bound to one device and is not thread-safe, include/mrs_hip.h) and per ACTION_TYPE instantiation.
            if (!h->big_lds[ACT]) {
                const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<ACT, 512, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
                if (ea != hipSuccess) return ea;
                h->big_lds[ACT] = true;
            }
            hipLaunchKernelGGL((k_step<ACT, 512, true, true>), dim3(g), dim3(512), l, st, MRS_STEP_PRE_ARGS(B), B);
        } else if (h->sblock == 1024) {
            if (!h->big_lds[ACT]) {
                const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_step<ACT, 1024, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l);
                if (ea != hipSuccess) return ea;
                h->big_lds[ACT] = true;
            }
            hipLaunchKernelGGL((k_step<ACT, 1024, true, true>), dim3(g), dim3(1024), l, st, MRS_STEP_PRE_ARGS(B), B);
        }
        else hipLaunchKernelGGL((k_step<ACT, 64, true, true>), dim3(g), dim3(64), l, st, MRS_STEP_PRE_ARGS(B), B);
    } else
    if (fused && h->N == 64) hipLaunchKernelGGL((k_step<ACT, 256, true, true>), dim3(grid), dim3(256), lds + 256 * sizeof(int) + 13 * 256 * sizeof(double) + 4 * sizeof(int), st, MRS_STEP_PRE_ARGS(A), A);
    else if (fused) hipLaunchKernelGGL((k_step<ACT, 256, true>), dim3(grid), dim3(256), lds + 256 * sizeof(int) + 13 * 256 * sizeof(double) + 4 * sizeof(int), st, MRS_STEP_PRE_ARGS(A), A);
    else if (h->block == 256) hipLaunchKernelGGL((k_step<ACT, 256, false>), dim3(grid), dim3(256), lds, st, MRS_STEP_PRE_ARGS(A), A);
    else hipLaunchKernelGGL((k_step<ACT, 1024, false>), dim3(grid), dim3(1024), lds, st, MRS_STEP_PRE_ARGS(A), A);
    return hipGetLastError();
}

extern "C" int mrs_step(MrsHandle *h, const MrsBuffers *b, const float *actions, int action_type,
                        const int32_t *obs_fields, int n_obs_fields, double comm_range, void *stream);
extern "C" int mrs_adjacency_expand(MrsHandle *h, const uint64_t *packed, float *dense, int n_matrices, void *stream);

// MrsBuffers.adj_dense: written by the kernel that builds the rows where an env is whole waves (N = 64; N = 128, 192, 256 in
// 256-thread workgroups) and the matrix is 16-byte aligned; anywhere else by mrs_adjacency_expand behind it.
static bool dense_in_kernel(const MrsHandle *h, const MrsBuffers *b, bool step_kernel)
{
    if (!b->adj_dense || !b->adj || ((uintptr_t)b->adj_dense & 15)) return false;
    if (h->N == 64) return true;
    if (h->N > 64 && h->N <= 256 && (h->N & 63) == 0) return ((step_kernel && h->fused) ? h->sblock : h->block) == 256;
    return false;
}
static int dense_behind(MrsHandle *h, const MrsBuffers *b, bool in_kernel, void *stream)
{
    if (!b->adj_dense || in_kernel) return 0;
    if (!b->adj) return fail(MRS_E_ARG, "adj_dense needs the packed rows (adj) as well");
    return mrs_adjacency_expand(h, b->adj, b->adj_dense, h->E, stream);
}

// n_substeps consecutive steps from ONE host call: the launches are queued back to back on the stream (the GPU runs
// them without a gap; what is saved is the caller's per-step work between them).  Keeping the state in registers
// across substeps inside one launch was built and measured in round 2: wrapped in a loop, the fused kernel no longer
// fits its 128 vector / 104 scalar registers (the compiler hoists ~50 per-lane plane addresses and the argument loads
// out of the loop: 350-860 bytes of scratch per lane), so the substeps stay separate launches.
extern "C" int mrs_step_n(MrsHandle *h, const MrsBuffers *b, const float *actions, int action_type, int n_substeps, int64_t action_stride,
                          const int32_t *obs_fields, int n_obs_fields, double comm_range, int64_t obs_stride, int64_t adj_stride, void *stream)
{
    if (!h || !b) return fail(MRS_E_ARG, "mrs_step_n: NULL handle/buffers");
    if (n_substeps < 1) return fail(MRS_E_ARG, "mrs_step_n: n_substeps must be >= 1");
    MrsBuffers bb = *b;
    for (int s = 0; s < n_substeps; ++s) {
        bb.adj_dense = s == n_substeps - 1 ? b->adj_dense : nullptr; // the dense matrices of the last substep only (with its packed rows)
        const int rc = mrs_step(h, &bb, actions ? actions + (long long)s * action_stride : nullptr, action_type, obs_fields, n_obs_fields, comm_range, stream);
        if (rc) return rc;
        if (bb.obs) bb.obs += obs_stride;
        if (bb.adj) bb.adj += adj_stride;
    }
    return 0;
}

extern "C" int mrs_step(MrsHandle *h, const MrsBuffers *b, const float *actions, int action_type,
                        const int32_t *obs_fields, int n_obs_fields, double comm_range, void *stream)
{
    if (!h || !b) return fail(MRS_E_ARG, "mrs_step: NULL handle/buffers");
    DeviceGuard dg(h->device);
    if (!b->pos || !b->quat || !b->vel || !b->angvel) return fail(MRS_E_ARG, "mrs_step: state buffers missing");
    if (action_type < MRS_ACT_NONE || action_type > MRS_ACT_TARGET_ORI)
        return fail(MRS_E_ACTION_TYPE, "mrs_step: unknown ACTION_TYPE (the reference raises AttributeError, Environment.py:92)");
    if (action_type != MRS_ACT_NONE && !actions) return fail(MRS_E_ARG, "mrs_step: actions is NULL");
    if (action_type >= MRS_ACT_TARGET_ACCEL && !b->pid) return fail(MRS_E_ARG, "mrs_step: PID buffer missing");
    StepArgs A;
    int rc = fill_common(h, b, obs_fields, n_obs_fields, comm_range, A);
    if (rc) return rc;
    A.actions = actions;
    const bool want_dense = b->adj_dense && !std::isnan(comm_range);
    const bool dense_k = want_dense && dense_in_kernel(h, b, true);
    if (!dense_k) A.b.adj_dense = nullptr;
#if MRS_KO == -1
    { const char *ko = getenv("MRS_KO"); A.ko = ko ? atoi(ko) : 0; }
#endif
    // alternating counters: this step's k_step adds to ws[parity] (read by this step's k_contact) and
    // zeroes ws[parity^1] for the next step -- stream order makes that safe without a memset node
    if (h->ws) {
        A.contact_count = h->ws + (h->step_parity & 1);
        A.contact_count_next = h->ws + ((h->step_parity & 1) ^ 1);
        A.contact_list = h->ws + 2;
        A.contact_state = h->cs;
    }
    h->step_parity++;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    const bool fused = h->fused;
    switch (action_type) {
    case MRS_ACT_NONE: e = launch_step<MRS_ACT_NONE>(h, A, st, fused); break;
    case MRS_ACT_SET_SPEEDS: e = launch_step<MRS_ACT_SET_SPEEDS>(h, A, st, fused); break;
    case MRS_ACT_SET_CONTROL: e = launch_step<MRS_ACT_SET_CONTROL>(h, A, st, fused); break;
    case MRS_ACT_TARGET_ACCEL: e = launch_step<MRS_ACT_TARGET_ACCEL>(h, A, st, fused); break;
    case MRS_ACT_TARGET_VEL: e = launch_step<MRS_ACT_TARGET_VEL>(h, A, st, fused); break;
    case MRS_ACT_TARGET_POS: e = launch_step<MRS_ACT_TARGET_POS>(h, A, st, fused); break;
    default: e = launch_step<MRS_ACT_TARGET_ORI>(h, A, st, fused); break;
    }
    if (e != hipSuccess) return hipfail(e, "mrs_step launch");
    if (fused) return want_dense ? dense_behind(h, b, dense_k, stream) : 0;
    if (h->P.enable_contact) {
        // worst-case grid; blocks beyond the device-side count return at once
        const int T = h->E * h->N;
        const int blocks = (T + 255) / 256 < 1024 ? (T + 255) / 256 : 1024;
        hipLaunchKernelGGL(k_contact, dim3(blocks), dim3(256), 0, st, A);
        e = hipGetLastError();
        if (e != hipSuccess) return hipfail(e, "mrs_step contact launch");
    }
    if ((A.n_obs > 0 && A.b.obs) || A.do_adj) {
        rc = launch_observe_adj(h, A, st);
        if (rc) return rc;
    }
    return want_dense ? dense_behind(h, b, dense_k, stream) : 0;
}

static int launch_observe_adj(MrsHandle *h, const StepArgs &A, hipStream_t st)
{
    const int grid = (h->E + h->epb - 1) / h->epb;
    const size_t lds = 2 * (size_t)h->block * sizeof(float4) + 256 * sizeof(int); // tile + adjacency_blocks' third exchange word
    if (h->block == 256) hipLaunchKernelGGL((k_observe_adj<256>), dim3(grid), dim3(256), lds, st, A);
    else hipLaunchKernelGGL((k_observe_adj<1024>), dim3(grid), dim3(1024), lds, st, A);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "observe/adjacency launch");
}

extern "C" int mrs_observe(MrsHandle *h, const MrsBuffers *b, const int32_t *obs_fields, int n_obs_fields, void *stream)
{
    if (!h || !b || !b->obs) return fail(MRS_E_ARG, "mrs_observe: NULL handle/buffers/obs");
    DeviceGuard dg(h->device);
    StepArgs A;
    int rc = fill_common(h, b, obs_fields, n_obs_fields, NAN, A);
    if (rc) return rc;
    return launch_observe_adj(h, A, (hipStream_t)stream);
}

extern "C" int mrs_adjacency(MrsHandle *h, const MrsBuffers *b, double comm_range, void *stream)
{
    if (!h || !b || !b->adj) return fail(MRS_E_ARG, "mrs_adjacency: NULL handle/buffers/adj");
    DeviceGuard dg(h->device);
    if (std::isnan(comm_range)) return fail(MRS_E_ARG, "mrs_adjacency: comm_range is NaN");
    StepArgs A;
    int rc = fill_common(h, b, nullptr, 0, comm_range, A);
    if (rc) return rc;
    const bool dense_k = dense_in_kernel(h, b, false);
    if (!dense_k) A.b.adj_dense = nullptr;
    rc = launch_observe_adj(h, A, (hipStream_t)stream);
    return rc ? rc : dense_behind(h, b, dense_k, stream);
}

extern "C" int mrs_adjacency_expand(MrsHandle *h, const uint64_t *packed, float *dense, int n_matrices, void *stream)
{
    if (!h || !packed || !dense || n_matrices < 0) return fail(MRS_E_ARG, "mrs_adjacency_expand: bad argument");
    DeviceGuard dg(h->device);
    const size_t total = (size_t)n_matrices * h->N * h->N;
    if (total == 0) return 0;
    const int block = 256;
    const size_t grid = (total + block - 1) / block;
    if (grid > 0x7fffffffull) return fail(MRS_E_ARG, "mrs_adjacency_expand: too large");
    if ((h->N & 3) == 0 && ((uintptr_t)dense & 15) == 0) {
        const size_t total4 = total >> 2;
        const size_t g4 = (total4 + block - 1) / block;
        hipLaunchKernelGGL(k_adj_expand4, dim3((unsigned)(g4 < 16384 ? g4 : 16384)), dim3(block), 0, (hipStream_t)stream, packed,
                           reinterpret_cast<float4 *>(dense), h->N, h->W, total4);
    } else {
        hipLaunchKernelGGL(k_adj_expand, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream, packed, dense, h->N, h->W, total);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_adjacency_expand launch");
}

static int launch_set(MrsHandle *, const SetArgs &S, hipStream_t st)
{
    const int block = 256;
    const unsigned grid = (unsigned)((S.T + block - 1) / block);
    hipLaunchKernelGGL(k_set_state, dim3(grid), dim3(block), 0, st, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_set_state launch");
}

// Positions were rewritten from outside the step (set_state, spawn): the envs concerned (all, or those of env_mask) look
// for quad-quad contact partners in their next step (StepArgs.pair_flag); the adjacency pass of that step -- or of
// mrs_observe / mrs_adjacency -- makes the flags exact again.  (Per env, so that a loop that resets a few envs every step
// -- MRS(AUTO_RESET=True) -- does not send the whole swarm through the scan.)
__global__ void k_pairs_unknown(int *flag, const uint8_t *mask, int E)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E && mask[e]) flag[e] = 1;
}
static int pairs_unknown(MrsHandle *h, const uint8_t *env_mask, hipStream_t st, const char *where)
{
    if (!h->pair_flag) return 0;
    hipError_t e;
    if (env_mask) {
        hipLaunchKernelGGL(k_pairs_unknown, dim3((unsigned)((h->E + 255) / 256)), dim3(256), 0, st, h->pair_flag, env_mask, h->E);
        e = hipGetLastError();
    } else e = hipMemsetAsync(h->pair_flag, 1, (size_t)h->E * sizeof(int), st);
    return e == hipSuccess ? 0 : hipfail(e, where);
}

extern "C" int mrs_set_state(MrsHandle *h, const MrsBuffers *b, const float *pos, const float *ori, int ori_kind,
                             const float *vel, const float *angvel, const uint8_t *env_mask, void *stream)
{
    if (!h || !b) return fail(MRS_E_ARG, "mrs_set_state: NULL handle/buffers");
    DeviceGuard dg(h->device);
    if (ori && (ori_kind < MRS_ORI_EULER || ori_kind > MRS_ORI_MATRIX)) return fail(MRS_E_ARG, "mrs_set_state: bad ori_kind");
    SetArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.pos = pos; S.ori = ori; S.vel = vel; S.angvel = angvel; S.mask = env_mask; S.ori_kind = ori_kind;
    S.N = h->N; S.T = (size_t)h->E * h->N;
    const int rc = launch_set(h, S, (hipStream_t)stream);
    return rc ? rc : pairs_unknown(h, env_mask, (hipStream_t)stream, "mrs_set_state");
}

extern "C" int mrs_set_state_f64(MrsHandle *h, const MrsBuffers *b, const double *pos, const double *quat,
                                 const double *vel, const double *angvel, const uint8_t *env_mask, void *stream)
{
    if (!h || !b) return fail(MRS_E_ARG, "mrs_set_state_f64: NULL handle/buffers");
    DeviceGuard dg(h->device);
    SetArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.pos64 = pos; S.quat64 = quat; S.vel64 = vel; S.angvel64 = angvel; S.mask = env_mask;
    S.N = h->N; S.T = (size_t)h->E * h->N;
    const int rc = launch_set(h, S, (hipStream_t)stream);
    return rc ? rc : pairs_unknown(h, env_mask, (hipStream_t)stream, "mrs_set_state_f64");
}

extern "C" int mrs_pid_reset(MrsHandle *h, const MrsBuffers *b, const uint8_t *env_mask, void *stream)
{
    if (!h || !b || !b->pid) return fail(MRS_E_ARG, "mrs_pid_reset: NULL handle/buffers");
    DeviceGuard dg(h->device);
    const size_t T = (size_t)h->E * h->N;
    const int block = 256;
    hipLaunchKernelGGL(k_pid_reset, dim3((unsigned)((T + block - 1) / block)), dim3(block), 0, (hipStream_t)stream, *b, env_mask, h->N, T);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_pid_reset launch");
}

extern "C" int mrs_reynolds(MrsHandle *h, const float *x_prev, int D, float *actions, void *stream)
{
    if (!h || !x_prev || !actions) return fail(MRS_E_ARG, "mrs_reynolds: NULL argument");
    DeviceGuard dg(h->device);
    if (D < 6) return fail(MRS_E_ARG, "mrs_reynolds: D must be >= 6 (pos, vel lead the state vector)");
    const int grid = (h->E + h->epb - 1) / h->epb;
    const size_t lds = 2 * (size_t)h->block * sizeof(float4);
    if (h->block == 256) hipLaunchKernelGGL((k_reynolds<256>), dim3(grid), dim3(256), lds, (hipStream_t)stream, x_prev, actions, h->E, h->N, D, h->epb);
    else hipLaunchKernelGGL((k_reynolds<1024>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, x_prev, actions, h->E, h->N, D, h->epb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_reynolds launch");
}

extern "C" int mrs_spawn(MrsHandle *h, const MrsBuffers *b, uint64_t seed, int64_t env_index_base, double agent_radius,
                         const float ori_lo[3], const float ori_hi[3], int max_rounds, const uint8_t *env_mask, void *stream)
{
    if (!h || !b || !ori_lo || !ori_hi) return fail(MRS_E_ARG, "mrs_spawn: NULL argument");
    DeviceGuard dg(h->device);
    if (max_rounds < 1) return fail(MRS_E_ARG, "mrs_spawn: max_rounds must be >= 1");
    SpawnArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.seed = seed; S.env_base = env_index_base; S.mask = env_mask; S.E = h->E; S.N = h->N;
    S.max_rounds = max_rounds; S.min_dist = (float)(2 * agent_radius); S.T = (size_t)h->E * h->N;
    for (int k = 0; k < 3; ++k) { S.ori_lo[k] = ori_lo[k]; S.ori_hi[k] = ori_hi[k]; }
    const int block = ((h->N + 63) / 64) * 64;
    const size_t lds = (size_t)h->N * sizeof(float4);
    hipLaunchKernelGGL(k_spawn, dim3(h->E), dim3(block), lds, (hipStream_t)stream, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? pairs_unknown(h, env_mask, (hipStream_t)stream, "mrs_spawn") : hipfail(e, "mrs_spawn launch");
}

extern "C" int mrs_spawn_from(MrsHandle *h, const MrsBuffers *b, const float *candidates, int n_rounds, int resume, double agent_radius,
                              const uint8_t *env_mask, void *stream)
{
    if (!h || !b || !candidates) return fail(MRS_E_ARG, "mrs_spawn_from: NULL argument");
    if (n_rounds < 1) return fail(MRS_E_ARG, "mrs_spawn_from: n_rounds must be >= 1");
    DeviceGuard dg(h->device);
    SpawnArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.mask = env_mask; S.E = h->E; S.N = h->N; S.max_rounds = 1 << 30; S.min_dist = (float)(2 * agent_radius);
    S.T = (size_t)h->E * h->N; S.cand = candidates; S.cand_rounds = n_rounds; S.resume = resume;
    const int block = ((h->N + 63) / 64) * 64;
    const size_t lds = (size_t)h->N * sizeof(float4);
    hipLaunchKernelGGL(k_spawn, dim3(h->E), dim3(block), lds, (hipStream_t)stream, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? pairs_unknown(h, env_mask, (hipStream_t)stream, "mrs_spawn_from") : hipfail(e, "mrs_spawn_from launch");
}

// ---------------------------------------------------------------------------------------------------- sensors
#include "mrs_sensors.hpp"
