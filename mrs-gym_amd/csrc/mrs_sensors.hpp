// mrs_sensors.hpp -- the reference's geometry sensors (mrsgym/Object.py:100-174), batched over every quadcopter of
// every env, against the analytic scene of env_generator('simple') (EnvCreator.py:7-13): the ground box of
// plane.urdf:24 (30 x 30 x 1 m at the origin, top face z = ground_z) and one collision cylinder per quadcopter
// (cf2x.urdf:34, axis = body z).  Included at the end of mrs_kernels.hip (same translation unit: MrsHandle, fail()).
//
// Not on the step() hot path: float64 geometry, one lane per ray / per pair, the env's cylinders staged in LDS.
// Parity with pybullet's rayTestBatch / getClosestPoints is unpinned (pybullet absent); the checker is
// oracle/mrs_sensors.c, itself certified by brute force (tests/test_oracle_sensors.py).
#pragma once

namespace mrs_sense {

struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 mk(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return D3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

constexpr double kGroundHalfXY = 15.0, kGroundThick = 1.0; // plane.urdf:24

// centre and unit axis (third column of Bullet's matrix of the state quaternion) of every cylinder of the env -> LDS
__device__ __forceinline__ void stage_cylinders(const MrsBuffers &b, size_t T, size_t a0, int N, double *lc)
{
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const size_t a = a0 + j;
        const double qx = b.quat[a], qy = b.quat[T + a], qz = b.quat[2 * T + a], qw = b.quat[3 * T + a];
        const double s = 2.0 / (qx * qx + qy * qy + qz * qz + qw * qw);
        double ax = (qx * qz + qw * qy) * s, ay = (qy * qz - qw * qx) * s, az = 1.0 - (qx * qx + qy * qy) * s;
        const double n = 1.0 / sqrt(ax * ax + ay * ay + az * az);
        lc[6 * j] = b.pos[a]; lc[6 * j + 1] = b.pos[T + a]; lc[6 * j + 2] = b.pos[2 * T + a];
        lc[6 * j + 3] = ax * n; lc[6 * j + 4] = ay * n; lc[6 * j + 5] = az * n;
    }
}

// Reciprocal and reciprocal square root for the divisions of the GJK and of the ray tests: the hardware's approximations (v_rcp_f64 / v_rsq_f64, ~2^-27) and two
// Newton steps each, without the scaling and fix-up passes of the correctly rounded forms (their operands here are lengths and
// volumes of centimetre-sized bodies metres apart, nowhere near the ends of the exponent range): <= 2 ulp.
#ifndef MRS_GJK_FAST_DIV
#define MRS_GJK_FAST_DIV 1
#endif
__device__ __forceinline__ double g_rcp(double x)
{
#if MRS_GJK_FAST_DIV
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}
__device__ __forceinline__ double g_rsqrt(double x) // x > 0
{
#if MRS_GJK_FAST_DIV
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    r = __builtin_fma(__builtin_fma(-hx * r, r, 0.5), r, r);
    r = __builtin_fma(__builtin_fma(-hx * r, r, 0.5), r, r);
    return r;
#else
    return 1.0 / sqrt(x);
#endif
}
// segment o + t d, t in [0,1], against the capped cylinder (centre c, unit axis a); entering parameter or -1.
// A segment that starts inside reports nothing (Bullet's convex cast from inside a convex shape).
__device__ __forceinline__ double ray_cylinder(D3 o, D3 d, D3 c, D3 a, double rc, double hl)
{
    const D3 oc = o - c;
    const double oz = dot(oc, a), dz = dot(d, a);
    const D3 orad = oc - oz * a, drad = d - dz * a;
    const double cc = dot(orad, orad) - rc * rc;
    if (cc <= 0 && fabs(oz) <= hl) return -1.0;
    double best = -1.0;
    const double aa = dot(drad, drad);
    if (aa > 0) {
        const double bb = dot(orad, drad), disc = bb * bb - aa * cc;
        if (disc >= 0) {
            const double t = (-bb - (disc > 0 ? disc * g_rsqrt(disc) : 0.0)) * g_rcp(aa);
            if (t >= 0 && t <= 1 && fabs(oz + t * dz) <= hl) best = t;
        }
    }
    if (dz != 0) {
        const double t = ((dz > 0 ? -hl : hl) - oz) * g_rcp(dz);
        if (t >= 0 && t <= 1) {
            const D3 r = orad + t * drad;
            if (dot(r, r) <= rc * rc && (best < 0 || t < best)) best = t;
        }
    }
    return best;
}

__device__ __forceinline__ double ray_box(D3 o, D3 d, D3 lo, D3 hi)
{
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) inside &= (oo[k] >= l[k] && oo[k] <= h[k]);
    if (inside) return -1.0;
    double t0 = 0, t1 = 1;
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0) {
            miss |= (oo[k] < l[k] || oo[k] > h[k]);
        } else {
            const double rd = g_rcp(dd[k]);
            double x = (l[k] - oo[k]) * rd, y = (h[k] - oo[k]) * rd;
            if (x > y) { const double s = x; x = y; y = s; }
            t0 = fmax(t0, x); t1 = fmin(t1, y);
        }
    }
    return (miss || t0 > t1) ? -1.0 : t0;
}

struct RayArgs {
    MrsBuffers b;
    const float *offset, *dirs; // [R][3]
    int32_t *hit;               // (E,N,R): -1 none, 0..N-1 quadcopter, N ground
    float *pos_world, *pos_body, *dist;
    int E, N, R, body;
    float range;
    double rc, hl, ground_z;
    size_t T;
};

// Object.raycast (Object.py:150-174): one workgroup per env, lanes stride over (agent, ray)
__global__ __launch_bounds__(256) void k_raycast(const RayArgs S)
{
    extern __shared__ double lc[]; // [N][6] centre, axis; then [N][12] floats: the casting agent's float32 read-back (matrix, position);
                                   // then [3][NP] floats: the centres once more, float32, one plane per coordinate (the cull below)
    float *rb = reinterpret_cast<float *>(lc + 6 * S.N);
    const int NP = (S.N + 63) & ~63; // whole blocks of 64: the pad slots hold a centre nothing comes near
    float *cf = rb + 12 * S.N;
    const int e = blockIdx.x;
    const size_t a0 = (size_t)e * S.N, T = S.T;
    stage_cylinders(S.b, T, a0, S.N, lc);
    // what get_ori(mat=True) / get_pos() hand to raycast (Object.py:86-97): float32 of the float32-truncated state -- once per
    // agent (round 4; every one of an agent's rays used to redo the float64 quaternion -> matrix read-back)
    for (int i = threadIdx.x; i < S.N; i += blockDim.x) {
        const size_t a = a0 + i;
        const double pd[3] = {S.b.pos[a], S.b.pos[T + a], S.b.pos[2 * T + a]};
        const double qd[4] = {S.b.quat[a], S.b.quat[T + a], S.b.quat[2 * T + a], S.b.quat[3 * T + a]};
        const double zero[3] = {0, 0, 0};
        mrs::Observed ob;
        mrs::observe<false, true>(pd, qd, zero, zero, ob);
        float *o = rb + 12 * i;
        o[0] = ob.r00; o[1] = ob.r01; o[2] = ob.r02; o[3] = ob.r10; o[4] = ob.r11; o[5] = ob.r12; o[6] = ob.r20; o[7] = ob.r21; o[8] = ob.r22;
        o[9] = ob.px; o[10] = ob.py; o[11] = ob.pz;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NP; j += blockDim.x) { // (an odd N's pad slot: a centre nothing comes near)
        cf[j] = j < S.N ? (float)lc[6 * j] : 1e15f; cf[NP + j] = j < S.N ? (float)lc[6 * j + 1] : 1e15f; cf[2 * NP + j] = j < S.N ? (float)lc[6 * j + 2] : 1e15f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S.N * S.R; idx += blockDim.x) {
        const int i = idx / S.R, r = idx - i * S.R;
        const size_t a = a0 + i;
        const float *ro = rb + 12 * i;
        const float Rm[9] = {ro[0], ro[1], ro[2], ro[3], ro[4], ro[5], ro[6], ro[7], ro[8]};
        const float opos[3] = {ro[9], ro[10], ro[11]};
        // :151 directions *= RANGE, :159-163 rotate (body=True), start = offset + pos, end = directions + start: float32
        float dl[3], of[3], dw[3], ofw[3], st[3], en[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { dl[k] = mrs::f32mul(S.dirs[3 * r + k], S.range); of[k] = S.offset[3 * r + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (S.body) {
                ofw[k] = mrs::f32add(mrs::f32add(mrs::f32mul(Rm[3 * k], of[0]), mrs::f32mul(Rm[3 * k + 1], of[1])), mrs::f32mul(Rm[3 * k + 2], of[2]));
                dw[k] = mrs::f32add(mrs::f32add(mrs::f32mul(Rm[3 * k], dl[0]), mrs::f32mul(Rm[3 * k + 1], dl[1])), mrs::f32mul(Rm[3 * k + 2], dl[2]));
            } else { ofw[k] = of[k]; dw[k] = dl[k]; }
            st[k] = mrs::f32add(ofw[k], opos[k]);
            en[k] = mrs::f32add(dw[k], st[k]);
        }
        const D3 o = mk(st[0], st[1], st[2]), d = mk((double)en[0] - st[0], (double)en[1] - st[1], (double)en[2] - st[2]);
        double best = ray_box(o, d, mk(-kGroundHalfXY, -kGroundHalfXY, S.ground_z - kGroundThick), mk(kGroundHalfXY, kGroundHalfXY, S.ground_z));
        int obj = best >= 0 ? S.N : -1;
        // Round 4: a segment that misses a cylinder's bounding sphere misses the cylinder -- ten flops (no division, no root) in
        // front of the sixty-odd of the exact test, which only the few cylinders near the ray then reach: same hits, same
        // parameters (the exact test is untouched): 466 -> 197 us per call at N = 64 x 4096, 8 rays (profiles/r04_sensor_bench.txt).
        // The survivors are few per ray (two or three of 64) but different for every lane: tested where they turn up, nearly every
        // pass of a wave would run the exact test for somebody.  So each lane first NOTES its survivors of 64 cylinders in a bit mask
        // (the cheap loop), then walks its own bits, ascending as before (the lowest index wins a tie): the wave runs the exact test as
        // often as its busiest lane has survivors.
        // Round 5: the cull runs in float32, two cylinders per packed instruction, and is made CONSERVATIVE instead of exact: the
        // bounding sphere is taken 2 % larger in r^2 and every comparison is given a margin of 4e-6 of the magnitudes that entered it
        // (float32 rounds the ten operations of a test to < 1e-6 of them; the float32 copies of the centres are < 1e-6 m off) --
        // a segment it calls a miss is a miss in float64 as well, the survivors are a few more than before, and the exact float64 test
        // that decides is untouched: same hits.  With the divisions and the root of the exact tests as reciprocals with two Newton steps
        // (g_rcp, g_rsqrt): 190 -> 164 us per call at N = 64 x 4096, 8 rays (same box).
        // Then: ONE sign per cylinder -- the smallest of three slacks of the line's distance and closest-point parameter, a superset of the
        // segment test -- and the candidate mask shifted together two bits per pass: 165 -> 126 us.  (Measured and not kept: the
        // survivors of a round of 256 rays compacted into an LDS list and tested densely, nearest hit by 64-bit atomic minima -- same hits,
        // 182 us: the exact tests are not where the time is, the cull's 64 tests per ray are.)
        typedef float F2 __attribute__((ext_vector_type(2)));
        const float ox = st[0], oy = st[1], oz = st[2], dxf = (float)d.x, dyf = (float)d.y, dzf = (float)d.z;
        const float ddf = dxf * dxf + dyf * dyf + dzf * dzf;
        const float rho2f = (float)((S.rc * S.rc + S.hl * S.hl) * 1.02), kap = 4e-6f;
        // candidate <=> the LINE comes within rho of the centre, at a parameter within [-rho/|d|, 1 + rho/|d|] (a superset of the
        // segment's bounding-sphere test: one sign to look at per cylinder instead of a three-way case)
        const float sd = sqrtf(rho2f * ddf) * 1.01f + 1e-6f * ddf, sdd = ddf + sd, kdd = kap * ddf; // (margins: b carries ~1e-6 of |oc| |d|)
        for (int j0 = 0; j0 < S.N; j0 += 64) {
            unsigned wlo = 0, whi = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned wbits = 0;
#pragma unroll 8
                for (int jj = 30; jj >= 0; jj -= 2) { // descending: the word is shifted up two bits per pass
                    const int j = j0 + 32 * h + jj;
                    const F2 cx = *reinterpret_cast<const F2 *>(cf + j), cy = *reinterpret_cast<const F2 *>(cf + NP + j), cz = *reinterpret_cast<const F2 *>(cf + 2 * NP + j);
                    const F2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                    const F2 q = ocx * ocx + ocy * ocy + ocz * ocz, b = ocx * dxf + ocy * dyf + ocz * dzf;
                    // dd (q - rho^2) - b^2 <= margin, b <= sd, -b <= dd + sd: the smallest of the three slacks is >= 0
                    const F2 s0 = (b * b - (q - rho2f) * ddf) + q * kdd, s1 = sd - b, s2 = b + sdd;
                    const F2 sl = __builtin_elementwise_min(__builtin_elementwise_min(s0, s1), s2);
                    // (a NaN slack -- a non-finite centre -- compares false with "< 0": the cylinder goes to the exact test, as before)
                    wbits = (wbits << 2) | (sl.y < 0.f ? 0u : 2u) | (sl.x < 0.f ? 0u : 1u);
                }
                if (h == 0) wlo = wbits; else whi = wbits;
            }
            unsigned long long cand = ((unsigned long long)whi << 32) | wlo;
            while (cand) {
                const int j = j0 + __builtin_ctzll(cand);
                cand &= cand - 1;
                const double t = ray_cylinder(o, d, mk(lc[6 * j], lc[6 * j + 1], lc[6 * j + 2]), mk(lc[6 * j + 3], lc[6 * j + 4], lc[6 * j + 5]), S.rc, S.hl);
                if (t >= 0 && (best < 0 || t < best)) { best = t; obj = j; }
            }
        }
        const size_t o3 = ((size_t)a * S.R + r) * 3, o1 = (size_t)a * S.R + r;
        S.hit[o1] = obj;
        float pw[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
        if (obj >= 0) {
            const double hit[3] = {o.x + best * d.x, o.y + best * d.y, o.z + best * d.z};
#pragma unroll
            for (int k = 0; k < 3; ++k) pw[k] = mrs::f32sub((float)hit[k], ofw[k]); // :166 "pos world" = hit - rotated offset
#pragma unroll
            for (int k = 0; k < 3; ++k) {                                           // :167 pos = R^T pos_world - R^T pos
                const float x = mrs::f32add(mrs::f32add(mrs::f32mul(Rm[k], pw[0]), mrs::f32mul(Rm[3 + k], pw[1])), mrs::f32mul(Rm[6 + k], pw[2]));
                const float y = mrs::f32add(mrs::f32add(mrs::f32mul(Rm[k], opos[0]), mrs::f32mul(Rm[3 + k], opos[1])), mrs::f32mul(Rm[6 + k], opos[2]));
                pb[k] = mrs::f32sub(x, y);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { S.pos_world[o3 + k] = pw[k]; S.pos_body[o3 + k] = pb[k]; }
        S.dist[o1] = sqrtf(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]); // :171 (zeros on a miss, :168-170)
    }
}

// ---- closest points of two convex cylinders: GJK with Ericson's closest-point-on-simplex cases (same algorithm as
// the oracle's gjk_cyl_cyl; the simplex lives in registers / scratch, this is not a hot kernel)
// Stopping rule of the GJK iteration: the gap v.v - v.w (an upper bound of |v|^2 - d^2) relative to v.v.  The oracle iterates to
// 1e-14; the outputs here are float32, whose half-ulp is 6e-8 relative: 1e-9 leaves the float64 distance 60 times closer than
// that and saves the last iteration or two of four -- k_proximity at N = 64 x 4096 with points: 1380 -> 830 us, 750 with the
// reciprocals below (round 5).
// (Built and measured in round 5, not kept: every lane advancing its pair one iteration per pass and taking the next pair of the
// list as soon as its own is finished.  The pairs of an env need 2 - 4 iterations nearly all and up to 41 a few, a wave working
// through "its" 64 pairs runs at 41 % of its lanes; but lanes at different iterations are in different simplex cases, every pass
// then executes all four: 1021 us against 829 at equal tolerance.)
#ifndef MRS_GJK_TOL
#define MRS_GJK_TOL 1e-9
#endif
struct Cyl {
    D3 c, a;
};
__device__ __forceinline__ D3 cyl_support(const Cyl &s, D3 d, double rc, double hl)
{
    const double da = dot(d, s.a);
    const D3 rad = d - da * s.a;
    const double n2 = dot(rad, rad);
    D3 r = s.c + (da >= 0 ? hl : -hl) * s.a;
    if (n2 > 1e-280) r = r + (rc * g_rsqrt(n2)) * rad;
    return r;
}
// The simplex lives in REGISTERS (round 3; round 2 indexed its arrays with run-time indices -- "add the point at s.n",
// "keep points idx[k]" -- which forced the whole structure, and its three working copies, into 1 312 bytes of scratch per
// lane): every index below is a compile-time constant after inlining and unrolling, the one run-time position (where the
// new point goes) is a select per slot, the support point on B is not kept (b = a - w), the roll-back copy of the previous
// simplex is replaced by the previous closest points, and the tetrahedron case chooses its face in a first pass over the
// four faces and builds it in a second instead of keeping the best candidate simplex.
struct Simplex {
    D3 w[4], a[4]; // w = a - b (Minkowski difference), a = support point on A
    double l[4];
    int n;
};
template <int I0, int I1, int I2, int NK>
__device__ __forceinline__ void sx_keep(Simplex &s, double l0, double l1, double l2)
{
    const D3 w0 = s.w[I0], w1 = s.w[I1], w2 = s.w[I2], a0 = s.a[I0], a1 = s.a[I1], a2 = s.a[I2];
    s.w[0] = w0; s.a[0] = a0; s.l[0] = l0;
    if (NK > 1) { s.w[1] = w1; s.a[1] = a1; s.l[1] = l1; }
    if (NK > 2) { s.w[2] = w2; s.a[2] = a2; s.l[2] = l2; }
    s.n = NK;
}
__device__ __forceinline__ void sx_segment(Simplex &s)
{
    const D3 A = s.w[0], ab = s.w[1] - A;
    const double t = -dot(A, ab), dn = dot(ab, ab);
    if (t <= 0 || dn <= 0) { s.n = 1; s.l[0] = 1; return; }
    if (t >= dn) { sx_keep<1, 0, 0, 1>(s, 1, 0, 0); return; }
    s.l[1] = t * g_rcp(dn); s.l[0] = 1 - s.l[1];
}
__device__ __forceinline__ void sx_triangle(Simplex &s)
{
    const D3 a = s.w[0], b = s.w[1], c = s.w[2], ab = b - a, ac = c - a, ap = -1.0 * a;
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { sx_keep<0, 0, 0, 1>(s, 1, 0, 0); return; }
    const D3 bp = -1.0 * b;
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { sx_keep<1, 0, 0, 1>(s, 1, 0, 0); return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 * g_rcp(d1 - d3); sx_keep<0, 1, 0, 2>(s, 1 - v, v, 0); return; }
    const D3 cp = -1.0 * c;
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { sx_keep<2, 0, 0, 1>(s, 1, 0, 0); return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 * g_rcp(d2 - d6); sx_keep<0, 2, 0, 2>(s, 1 - w, w, 0); return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) * g_rcp((d4 - d3) + (d5 - d6)); sx_keep<1, 2, 0, 2>(s, 1 - w, w, 0); return; }
    const double den = g_rcp(va + vb + vc), v = vb * den, w = vc * den;
    s.l[0] = 1 - v - w; s.l[1] = v; s.l[2] = w;
}
__device__ __forceinline__ D3 sx_point(const Simplex &s)
{
    D3 p = mk(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < s.n) p = p + s.l[k] * s.w[k];
    return p;
}
// face F of the tetrahedron (points I0, I1, I2; the fourth is IO): is the origin on its outer side, and if so how far is
// the closest point of the face (squared)?  build = true: also leave that face's reduced simplex in s.
template <int I0, int I1, int I2, int IO, bool BUILD>
__device__ __forceinline__ bool sx_face(Simplex &s, double &dd)
{
    const D3 a = s.w[I0], b = s.w[I1], c = s.w[I2], d = s.w[IO];
    const D3 n = cross(b - a, c - a);
    const double so = dot(-1.0 * a, n), sd = dot(d - a, n);
    if (!(so * sd < 0 || sd == 0)) return false;
    Simplex t = s;
    sx_keep<I0, I1, I2, 3>(t, 0, 0, 0);
    sx_triangle(t);
    const D3 q = sx_point(t);
    dd = dot(q, q);
    if (BUILD) s = t;
    return true;
}
__device__ __forceinline__ bool sx_tetra(Simplex &s)
{
    // faces {0,1,2}, {0,2,3}, {0,3,1}, {1,3,2} with their opposite points 3, 1, 2, 0; the first face with the smallest
    // distance wins (strict <, as the oracle's loop)
    double dd[4] = {0, 0, 0, 0};
    bool out[4];
    out[0] = sx_face<0, 1, 2, 3, false>(s, dd[0]);
    out[1] = sx_face<0, 2, 3, 1, false>(s, dd[1]);
    out[2] = sx_face<0, 3, 1, 2, false>(s, dd[2]);
    out[3] = sx_face<1, 3, 2, 0, false>(s, dd[3]);
    int best = -1;
    double bd = -1;
#pragma unroll
    for (int f = 0; f < 4; ++f)
        if (out[f] && (bd < 0 || dd[f] < bd)) { bd = dd[f]; best = f; }
    if (best < 0) return true; // the origin is inside
    double unused;
    if (best == 0) sx_face<0, 1, 2, 3, true>(s, unused);
    else if (best == 1) sx_face<0, 2, 3, 1, true>(s, unused);
    else if (best == 2) sx_face<0, 3, 1, 2, true>(s, unused);
    else sx_face<1, 3, 2, 0, true>(s, unused);
    return false;
}
__device__ inline double gjk_cyl_cyl(const Cyl &A, const Cyl &B, double rc, double hl, D3 &pa, D3 &pb)
{
    Simplex s;
    s.n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.w[k] = mk(0, 0, 0); s.a[k] = mk(0, 0, 0); s.l[k] = 0; }
    D3 v = A.c - B.c;
    if (dot(v, v) < 1e-24) v = mk(1, 0, 0);
    D3 ca = mk(0, 0, 0), cb = mk(0, 0, 0); // closest points of the current simplex (what a roll-back returns to)
    for (int it = 0; it < 64; ++it) {
        const D3 sa = cyl_support(A, -1.0 * v, rc, hl), sb = cyl_support(B, v, rc, hl), w = sa - sb;
        const double vv = dot(v, v);
        if (s.n > 0 && vv - dot(v, w) <= MRS_GJK_TOL * vv + 1e-30) break;
        bool dup = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) dup |= (k < s.n) && (s.w[k].x == w.x && s.w[k].y == w.y && s.w[k].z == w.z);
        if (dup) break;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == s.n) { s.w[k] = w; s.a[k] = sa; }
        s.n++;
        bool inside = false;
        if (s.n == 1) s.l[0] = 1;
        else if (s.n == 2) sx_segment(s);
        else if (s.n == 3) sx_triangle(s);
        else inside = sx_tetra(s);
        if (inside) {
            const D3 e1 = s.w[1] - s.w[0], e2 = s.w[2] - s.w[0], e3 = s.w[3] - s.w[0], o = -1.0 * s.w[0];
            const double vol = dot(e1, cross(e2, e3));
            double l1 = dot(o, cross(e2, e3)) / vol, l2 = dot(e1, cross(o, e3)) / vol, l3 = dot(e1, cross(e2, o)) / vol;
            if (!(fabs(vol) > 0)) { l1 = l2 = l3 = 0.25; }
            pa = pb = (1 - l1 - l2 - l3) * s.a[0] + l1 * s.a[1] + l2 * s.a[2] + l3 * s.a[3];
            return 0.0;
        }
        const D3 vn = sx_point(s);
        if (it > 0 && dot(vn, vn) >= vv) break; // no progress: the previous simplex's closest points stand
        D3 a = mk(0, 0, 0), b = mk(0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < s.n) { a = a + s.l[k] * s.a[k]; b = b + s.l[k] * (s.a[k] - s.w[k]); }
        ca = a; cb = b;
        v = vn;
        if (dot(v, v) < 1e-24) break;
    }
    pa = ca; pb = cb;
    const D3 ab = ca - cb;
    const double d = sqrt(dot(ab, ab));
    return d < 1e-12 ? 0.0 : d;
}

struct ProxArgs {
    MrsBuffers b;
    float *dist, *p_self, *p_other; // (E,N,N+1), (E,N,N+1,3) x2 (optional)
    int E, N;
    double rc, hl, ground_z, max_dist;
    size_t T;
};

// Object.get_dist (Object.py:119-133) for every ordered pair of an env + every quadcopter against the ground: one workgroup per
// env.  Pairs whose bounding spheres are already further apart than max_dist are reported as +inf without running GJK (the
// reference's MAX_DIST returns nothing there).
// Round 4: (i) every UNORDERED pair goes through GJK once -- the iteration for (j, i) is the exact mirror image of the one for
// (i, j): v -> -v swaps the two support points, every simplex test is a dot product of two negated vectors -- and both
// directions are written from it, closest points swapped; (ii) the pairs that survive the cull are COMPACTED: chunks of 1024
// candidate pairs are tested first (one compare each) and the survivors appended to an LDS list, which the lanes then take
// densely.  Before, a surviving pair kept its whole wave waiting for the ~10^3 float64 instructions of its GJK while the
// other 63 lanes had nothing to do -- with a contact-range cull (collisions, get_contact_points: a handful of survivors per
// env) that was most of the 646 us the call took at N = 64 x 4096 (profiles/r03_sensor_bench.txt).
constexpr int PROX_CHUNK = 1024;
// WGS = workgroups per CU the register budget is cut for (round 5, N = 64 x 4096, us per call with WGS = 2 / 3): every pair through GJK,
// with points 759 / 708, without 725 / 613, max_dist 0.5: 325 / 298 -- but max_dist 0.02 with points, where nearly every pair is culled and
// the call is its 484 MB of stores: 217 / 283.  mrs_proximity takes 2 for contact-range queries and 3 otherwise.
template <int WGS>
__global__ __launch_bounds__(256, WGS) void k_proximity(const ProxArgs S)
{
    extern __shared__ double lc[];
    __shared__ int s_n;
    __shared__ unsigned s_list[PROX_CHUNK]; // i << 16 | j, i < j
    const int e = blockIdx.x, N = S.N;
    const size_t a0 = (size_t)e * N;
    stage_cylinders(S.b, S.T, a0, N, lc);
    __syncthreads();
    const double bound = sqrt(S.rc * S.rc + S.hl * S.hl);
    auto cyl = [&](int i) { return Cyl{mk(lc[6 * i], lc[6 * i + 1], lc[6 * i + 2]), mk(lc[6 * i + 3], lc[6 * i + 4], lc[6 * i + 5])}; };
    auto put = [&](int i, int j, double d, const D3 &pa, const D3 &pb) {
        const size_t o1 = (a0 + i) * (size_t)(N + 1) + j;
        S.dist[o1] = (float)d;
        if (S.p_self) { S.p_self[3 * o1] = (float)pa.x; S.p_self[3 * o1 + 1] = (float)pa.y; S.p_self[3 * o1 + 2] = (float)pa.z; }
        if (S.p_other) { S.p_other[3 * o1] = (float)pb.x; S.p_other[3 * o1 + 1] = (float)pb.y; S.p_other[3 * o1 + 2] = (float)pb.z; }
    };
    // the diagonal and the ground column: closed form, one lane each
    for (int idx = threadIdx.x; idx < 2 * N; idx += blockDim.x) {
        const int i = idx >> 1;
        const Cyl me = cyl(i);
        if (idx & 1) {
            // lowest point of the cylinder = support point along -z (the middle of the lowest line / cap when degenerate)
            const double az = me.a.z;
            D3 low = me.c + (az > 1e-12 ? -S.hl : (az < -1e-12 ? S.hl : 0.0)) * me.a;
            const D3 rad = mk(0, 0, -1) - (-az) * me.a;
            const double n = sqrt(dot(rad, rad));
            if (n > 1e-9) low = low + (S.rc / n) * rad;
            put(i, N, low.z - S.ground_z, low, mk(low.x, low.y, S.ground_z)); // signed: < 0 = sunk into the ground
        } else put(i, i, 0.0, me.c, me.c);
    }
    // the pairs i < j, in chunks of the ordered index i * N + j
    const int total = N * N;
    for (int base = 0; base < total; base += PROX_CHUNK) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        for (int idx = base + threadIdx.x; idx < min(base + PROX_CHUNK, total); idx += blockDim.x) {
            const int i = idx / N, j = idx - i * N;
            if (i >= j) continue;
            const D3 ci = mk(lc[6 * i], lc[6 * i + 1], lc[6 * i + 2]), cj = mk(lc[6 * j], lc[6 * j + 1], lc[6 * j + 2]);
            const D3 cc = ci - cj;
            if (sqrt(dot(cc, cc)) - 2 * bound > S.max_dist) { put(i, j, INFINITY, ci, ci); put(j, i, INFINITY, cj, cj); }
            else s_list[atomicAdd(&s_n, 1)] = ((unsigned)i << 16) | (unsigned)j;
        }
        __syncthreads();
        const int n = s_n;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const int i = (int)(s_list[k] >> 16), j = (int)(s_list[k] & 0xFFFFu);
            const Cyl me = cyl(i), o = cyl(j);
            D3 pa = me.c, pb = me.c;
            const double d = gjk_cyl_cyl(me, o, S.rc, S.hl, pa, pb);
            put(i, j, d, pa, pb);
            put(j, i, d, pb, pa);
        }
        __syncthreads();
    }
}

} // namespace mrs_sense


// ---- flocking metrics of the reference's analytics (examples/simulating_data/helper/MRSAnalytics.py:36-101) as device
// reductions: one workgroup per frame (episode, time step), positions and velocities of its N agents staged in LDS.
namespace mrs_metrics {
struct MetricArgs {
    const float *X; // (M,N,D) frames, D >= 6 = cat(pos, vel, ...)
    float *sep, *coh, *coh_nl, *lead, *vstd;
    int M, N, D;
};
__global__ __launch_bounds__(256) void k_flock_metrics(const MetricArgs S)
{
    extern __shared__ float4 lds[]; // [N] positions, [N] velocities
    __shared__ unsigned s_max[2];
    const int f = blockIdx.x, N = S.N;
    float4 *P = lds, *V = lds + N;
    const float *x = S.X + (size_t)f * N * S.D;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        P[j] = make_float4(x[j * S.D], x[j * S.D + 1], x[j * S.D + 2], 0.f);
        V[j] = make_float4(x[j * S.D + 3], x[j * S.D + 4], x[j * S.D + 5], 0.f);
    }
    if (threadIdx.x < 2) s_max[threadIdx.x] = 0u;
    __syncthreads();
    // separation (:61-72): distance to the closest OTHER position, zero distances masked with inf (:70);
    // cohesion (:82-93): the largest pairwise distance (all agents / without the leader, agent 0).
    // codist = (posi - posj).norm(dim=3): float32, d2 = fma(dz,dz,fma(dy,dy,dx*dx)) as torch's norm kernel evaluates it
    float mx = 0.f, mx_nl = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const float4 pi = P[i];
        float mn = INFINITY;
        for (int j = 0; j < N; ++j) {
            const float4 pj = P[j];
            const float dx = mrs::f32sub(pj.x, pi.x), dy = mrs::f32sub(pj.y, pi.y), dz = mrs::f32sub(pj.z, pi.z);
            const float d = mrs::f32sqrt(mrs::f32fma(dz, dz, mrs::f32fma(dy, dy, mrs::f32mul(dx, dx))));
            mn = fminf(mn, d == 0.f ? INFINITY : d);          // NaN frames (padding of short episodes) stay NaN-free here: fminf drops NaN
            mx = (d > mx || d != d) ? d : mx;                  // torch.max propagates NaN
            if (i > 0 && j > 0) mx_nl = (d > mx_nl || d != d) ? d : mx_nl;
        }
        if (S.sep) {
            // a frame of NaNs (Trainer.get_episodes pads short episodes with NaN): every distance is NaN, NaN == 0 is false,
            // min over NaNs is NaN in torch
            S.sep[(size_t)f * N + i] = (pi.x != pi.x) ? NAN : mn;
        }
    }
    // positive floats order like their bit patterns; NaN (0x7fc00000) is larger than every finite pattern => propagates
    atomicMax(&s_max[0], __float_as_uint(mx));
    atomicMax(&s_max[1], __float_as_uint(mx_nl));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S.coh) S.coh[f] = __uint_as_float(s_max[0]);
        if (S.coh_nl) S.coh_nl[f] = __uint_as_float(s_max[1]);
        if (S.lead) { // :95-101  | mean(pos[1:]) - pos[0] |
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int j = 1; j < N; ++j) { sx += P[j].x; sy += P[j].y; sz += P[j].z; }
            const float n1 = (float)(N - 1);
            const float dx = sx / n1 - P[0].x, dy = sy / n1 - P[0].y, dz = sz / n1 - P[0].z;
            S.lead[f] = sqrtf(dx * dx + dy * dy + dz * dz);
        }
        if (S.vstd) { // :44-53  sqrt(det(sum_i (v_i - vbar)(v_i - vbar)^T))
            double m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < N; ++j) { m[0] += V[j].x; m[1] += V[j].y; m[2] += V[j].z; }
            // vel_avg and the differences are float32 tensors upstream
            const float ax = (float)(m[0] / N), ay = (float)(m[1] / N), az = (float)(m[2] / N);
            for (int j = 0; j < N; ++j) {
                const double dx = (double)(V[j].x - ax), dy = (double)(V[j].y - ay), dz = (double)(V[j].z - az);
                c[0] += dx * dx; c[1] += dx * dy; c[2] += dx * dz; c[3] += dy * dy; c[4] += dy * dz; c[5] += dz * dz;
            }
            const double det = c[0] * (c[3] * c[5] - c[4] * c[4]) - c[1] * (c[1] * c[5] - c[4] * c[2]) + c[2] * (c[1] * c[4] - c[3] * c[2]);
            S.vstd[f] = (float)sqrt(det);
        }
    }
}
} // namespace mrs_metrics

extern "C" int mrs_flock_metrics(const float *X, int n_frames, int n_agents, int D, float *separation, float *cohesion,
                                 float *cohesion_noleader, float *dist_to_leader, float *vel_stddev, void *stream)
{
    if (!X || n_frames < 0 || n_agents < 1 || D < 6) return fail(MRS_E_ARG, "mrs_flock_metrics: bad argument (X, n_frames >= 0, n_agents >= 1, D >= 6)");
    if (n_frames == 0) return 0;
    if ((size_t)n_agents * 2 * sizeof(float4) > 64 * 1024) return fail(MRS_E_ARG, "mrs_flock_metrics: n_agents too large for one workgroup's LDS");
    mrs_metrics::MetricArgs S;
    S.X = X; S.sep = separation; S.coh = cohesion; S.coh_nl = cohesion_noleader; S.lead = dist_to_leader; S.vstd = vel_stddev;
    S.M = n_frames; S.N = n_agents; S.D = D;
    hipLaunchKernelGGL(mrs_metrics::k_flock_metrics, dim3(n_frames), dim3(256), (size_t)n_agents * 2 * sizeof(float4), (hipStream_t)stream, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_flock_metrics launch");
}

extern "C" int mrs_raycast(MrsHandle *h, const MrsBuffers *b, const float *offset, const float *directions, int n_rays, int body,
                           float range, int32_t *hit_obj, float *pos_world, float *pos_body, float *dist, void *stream)
{
    if (!h || !b || !offset || !directions || !hit_obj || !pos_world || !pos_body || !dist) return fail(MRS_E_ARG, "mrs_raycast: NULL argument");
    if (n_rays < 1) return fail(MRS_E_ARG, "mrs_raycast: n_rays must be >= 1");
    DeviceGuard dg(h->device);
    mrs_sense::RayArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.offset = offset; S.dirs = directions; S.hit = hit_obj; S.pos_world = pos_world; S.pos_body = pos_body; S.dist = dist;
    S.E = h->E; S.N = h->N; S.R = n_rays; S.body = body; S.range = range;
    S.rc = h->P.coll_radius; S.hl = h->P.coll_half_len; S.ground_z = h->P.ground_z; S.T = (size_t)h->E * h->N;
    // 108 bytes of LDS per agent: above 64 KB (N >= 607; mrs_create takes N up to 1024 = 108 KB of the CU's 160) the launch needs the
    // function attribute raised, once per handle = per device (the guard that stood here refused what used to run)
    const size_t lds = (size_t)h->N * (6 * sizeof(double) + 12 * sizeof(float)) + (size_t)((h->N + 63) & ~63) * 3 * sizeof(float);
    if (lds > 64 * 1024 && !h->raycast_big_lds) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&mrs_sense::k_raycast), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) return hipfail(ea, "mrs_raycast: raising the LDS limit");
        h->raycast_big_lds = true;
    }
    hipLaunchKernelGGL(mrs_sense::k_raycast, dim3(h->E), dim3(256), lds, (hipStream_t)stream, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_raycast launch");
}

extern "C" int mrs_proximity(MrsHandle *h, const MrsBuffers *b, double max_dist, float *dist, float *p_self, float *p_other, void *stream)
{
    if (!h || !b || !dist) return fail(MRS_E_ARG, "mrs_proximity: NULL argument");
    DeviceGuard dg(h->device);
    mrs_sense::ProxArgs S;
    memset(&S, 0, sizeof(S));
    S.b = *b; S.dist = dist; S.p_self = p_self; S.p_other = p_other; S.E = h->E; S.N = h->N;
    S.rc = h->P.coll_radius; S.hl = h->P.coll_half_len; S.ground_z = h->P.ground_z; S.max_dist = max_dist; S.T = (size_t)h->E * h->N;
    const double bound = std::sqrt(S.rc * S.rc + S.hl * S.hl);
    if (max_dist <= 4 * bound) hipLaunchKernelGGL(mrs_sense::k_proximity<2>, dim3(h->E), dim3(256), (size_t)h->N * 6 * sizeof(double), (hipStream_t)stream, S);
    else hipLaunchKernelGGL(mrs_sense::k_proximity<3>, dim3(h->E), dim3(256), (size_t)h->N * 6 * sizeof(double), (hipStream_t)stream, S);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hipfail(e, "mrs_proximity launch");
}
