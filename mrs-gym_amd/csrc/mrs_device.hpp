// mrs_device.hpp -- per-quadcopter device math of the fused step kernel (gfx950).
//
// One wavefront lane owns one quadcopter; everything here is straight-line register code with no
// cross-lane traffic.  The arithmetic keeps the reference's precision split: values "read back"
// from the simulator are truncated to float32 (Object.py:78-97) before the float64 controller math
// (QuadControl.py), the downwash pair term is float32 (Quadcopter.py:99-115), the rigid-body state
// itself is float64 like Bullet's.  Citations are file:line into the reference's mrsgym/ tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/mrs_hip.h"

#define MRS_DEV __device__ __forceinline__

namespace mrs {

constexpr double kPi = 3.14159265358979323846;

// float32 operations that must round exactly like the reference's numpy/torch float32 arithmetic.
// hipcc contracts a*b+c into an FMA by default and ROCm's __fmul_rn/__fadd_rn are plain operators,
// so each helper switches contraction off for its own statement (the flag travels with the
// instruction through inlining).  f32fma is the one place an FMA is wanted (torch's norm kernel).
MRS_DEV float f32mul(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
MRS_DEV float f32add(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
MRS_DEV float f32sub(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}
MRS_DEV float f32div(float a, float b)
{
#pragma clang fp contract(off)
    return a / b; // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
}
MRS_DEV float f32sqrt(float a) { return __builtin_sqrtf(a); } // correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt (__fsqrt_rn is the 1-ulp v_sqrt_f32: measured)
MRS_DEV float f32fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct V3 {
    double x, y, z;
};
struct M3 { // row-major rotation body->world
    double m00, m01, m02, m10, m11, m12, m20, m21, m22;
};

MRS_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
MRS_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
MRS_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
MRS_DEV V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
MRS_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MRS_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MRS_DEV double rsqrt64(double x);
MRS_DEV double sqrt64(double s) { return s > 0.0 ? s * rsqrt64(s) : 0.0; } // ~1 ulp, 11 instructions
MRS_DEV double norm(V3 a) { return sqrt64(dot(a, a)); }
MRS_DEV V3 mul(const M3 &R, V3 v)
{
    return V3{R.m00 * v.x + R.m01 * v.y + R.m02 * v.z, R.m10 * v.x + R.m11 * v.y + R.m12 * v.z,
              R.m20 * v.x + R.m21 * v.y + R.m22 * v.z};
}
MRS_DEV V3 mulT(const M3 &R, V3 v)
{
    return V3{R.m00 * v.x + R.m10 * v.y + R.m20 * v.z, R.m01 * v.x + R.m11 * v.y + R.m21 * v.z,
              R.m02 * v.x + R.m12 * v.y + R.m22 * v.z};
}
// v_max_f64 / v_min_f64: two instructions instead of two compares and four 32-bit selects
MRS_DEV double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

// Reciprocal and reciprocal square root to ~1 ulp: hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-26) plus
// two Newton steps (5 / 9 instructions) instead of the ~14-instruction correctly rounded division and
// the sqrt + division pair.  Arguments here are norms and cosines: positive, normal, finite.
MRS_DEV double rcp64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return r;
}
// 1 / x to ~1e-14 relative from the float32 reciprocal unit (v_rcp_f32, 1 ulp = 6e-8) and one Newton step: for factors that
// multiply a small term (the ground effect is ~1e-4 of a rotor's thrust), where the ~1e-16 of rcp64 buys nothing and
// v_rcp_f64 costs twice the issue slots of the float32 unit.  x positive, normal, inside the float32 range.
MRS_DEV double rcp64_coarse(double x)
{
    const double r = (double)__builtin_amdgcn_rcpf((float)x);
    return r * __builtin_fma(-x, r, 2.0);
}
MRS_DEV double rsqrt64(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * __builtin_fma(-0.5 * x, y * y, 1.5);
    y = y * __builtin_fma(-0.5 * x, y * y, 1.5);
    return y;
}

// ---- float64 trigonometry sized for this kernel -------------------------------------------------
// Euler angles live in [-pi, pi] and the integrator's half-angle in [0, pi/8], so the general
// libm routines (Payne-Hanek reduction, special-value branches) are dead weight per lane.  These use
// a two-term Cody-Waite reduction by at most +-2 quadrants and the classic fdlibm minimax kernels
// (< 1 ulp on |r| <= pi/4); arguments outside +-4 fall back to the library.
// Horner steps p <- a*b + C with the float64 coefficient C as an SGPR-pair operand of a VOP3 v_fma_f64.
// Left to itself the compiler selects the two-address v_fmac_f64, whose addend must sit in the destination
// VGPR pair: two v_mov_b32 per coefficient per use, ~600 of the step kernel's ~3000 vector instructions with
// the polynomials inlined at ~20 sites.  The "s" constraint makes it build C with two s_mov_b32 instead
// (scalar unit, off the VALU port the kernel is bound by).
MRS_DEV double fma_c(double a, double b, double c)
{
#ifdef MRS_HOST_CHECK // tools/host_f32: these functions compiled for the CPU (diagnostics, tests/test_device_math_host.py) -- never the product
    return __builtin_fma(a, b, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#endif
}
MRS_DEV double ksin(double r)
{
    const double z = r * r;
    double p = z * 1.58969099521155010221e-10 + -2.50507602534068634195e-08;
    p = fma_c(p, z, 2.75573137070700676789e-06);
    p = fma_c(p, z, -1.98412698298579493134e-04);
    p = fma_c(p, z, 8.33333333332248946124e-03);
    return r + (r * z) * fma_c(p, z, -1.66666666666666324348e-01);
}
MRS_DEV double kcos(double r)
{
    const double z = r * r;
    double p = z * -1.13596475577881948265e-11 + 2.08757232129817482790e-09;
    p = fma_c(p, z, -2.75573143513906633035e-07);
    p = fma_c(p, z, 2.48015872894767294178e-05);
    p = fma_c(p, z, -1.38888888888741095749e-03);
    p = fma_c(p, z, 4.16666666666666019037e-02);
    return (1.0 - 0.5 * z) + (z * z) * p;
}
MRS_DEV void fast_sincos(double x, double *s, double *c)
{
    if (!(fabs(x) <= 4.0)) { sincos(x, s, c); return; }
    const double k = __builtin_rint(x * 0.63661977236758134308);          // 2/pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);           // pi/2 hi
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);                  // pi/2 lo
    const double sr = ksin(r), cr = kcos(r);
    const int n = (int)k & 3;
    const double ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}
// atan on |t| <= tan(pi/8) (fdlibm's aT[] polynomial, valid to 7/16)
MRS_DEV double katan(double t)
{
    const double z = t * t, w = z * z;
    double e = w * 1.62858201153657823623e-02 + 4.97687799461593236017e-02;
    e = fma_c(e, w, 6.66107313738753120669e-02);
    e = fma_c(e, w, 9.09088713343650656196e-02);
    e = fma_c(e, w, 1.42857142725034663711e-01);
    e = fma_c(e, w, 3.33333333333329318027e-01);
    double o = w * -3.65315727442169155270e-02 + -5.83357013379057348645e-02;
    o = fma_c(o, w, -7.69187620504482999495e-02);
    o = fma_c(o, w, -1.11111104054623557880e-01);
    o = fma_c(o, w, -1.99999999998764832476e-01);
    return t - t * (z * e + w * o);
}
MRS_DEV double fast_atan2(double y, double x)
{
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    if (!(mx > 0.0) || !(mx < 1e300)) return atan2(y, x);                 // zeros / inf / NaN: library semantics
    const bool red = mn > 0.41421356237309503 * mx;
    const double t = (red ? mn - mx : mn) * rcp64(red ? mn + mx : mx);
    double a = katan(t);
    if (red) a = (a + 3.06161699786838301793e-17) + 7.85398163397448278999e-01;      // + pi/4 (lo, hi)
    if (ay > ax) a = (6.12323399573676603587e-17 - a) + 1.57079632679489655800e+00;  // pi/2 - a
    if (x < 0.0) a = (1.22464679914735320717e-16 - a) + 3.14159265358979311600e+00;  // pi - a
    return y < 0.0 ? -a : a;
}

// scipy Rotation.as_matrix of the normalised quaternion (Object.py:93-95)
MRS_DEV M3 quat_to_matrix_scipy(double qx, double qy, double qz, double qw)
{
    // |q|^2 = 1 + e with |e| <= 4e-7 here (a unit quaternion truncated to float32, Object.py:92-93):
    // 1/sqrt(1 + e) = 1 - e/2 + 3e^2/8 - 5e^3/16 (+35e^4/128 ~ 1e-26); a caller's non-unit quaternion takes the rsqrt
    const double e = (qx * qx + qy * qy + qz * qz + qw * qw) - 1.0;
    double rn;
    if (__builtin_amdgcn_ballot_w64(fabs(e) > 1e-5) == 0) rn = __builtin_fma(e, __builtin_fma(e, __builtin_fma(e, -0.3125, 0.375), -0.5), 1.0);
    else rn = rsqrt64(e + 1.0);
    const double x = qx * rn, y = qy * rn, z = qz * rn, w = qw * rn;
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M3 R;
    R.m00 = x2 - y2 - z2 + w2; R.m01 = 2 * (xy - zw);      R.m02 = 2 * (xz + yw);
    R.m10 = 2 * (xy + zw);      R.m11 = -x2 + y2 - z2 + w2; R.m12 = 2 * (yz - xw);
    R.m20 = 2 * (xz - yw);      R.m21 = 2 * (yz + xw);      R.m22 = -x2 - y2 + z2 + w2;
    return R;
}

// btMatrix3x3::setRotation (the matrix Bullet's integrator uses)
MRS_DEV M3 quat_to_matrix_bullet(double qx, double qy, double qz, double qw)
{
    const double d = qx * qx + qy * qy + qz * qz + qw * qw;
    // the state quaternion is re-normalised by every step: d = 1 + e with |e| ~ 1e-16, where 1/d = 1 - e + e^2 is exact
    // to double rounding; anything else (a caller's own quaternion through set_state) takes the division
    const double e = d - 1.0;
    double s;
    if (__builtin_amdgcn_ballot_w64(fabs(e) > 1e-8) == 0) s = 2.0 * __builtin_fma(e, e - 1.0, 1.0);
    else s = 2.0 * rcp64(d);
    const double xs = qx * s, ys = qy * s, zs = qz * s;
    const double wx = qw * xs, wy = qw * ys, wz = qw * zs;
    const double xx = qx * xs, xy = qx * ys, xz = qx * zs;
    const double yy = qy * ys, yz = qy * zs, zz = qz * zs;
    M3 R;
    R.m00 = 1.0 - (yy + zz); R.m01 = xy - wz;         R.m02 = xz + wy;
    R.m10 = xy + wz;         R.m11 = 1.0 - (xx + zz); R.m12 = yz - wx;
    R.m20 = xz - wy;         R.m21 = yz + wx;         R.m22 = 1.0 - (xx + yy);
    return R;
}

// as_euler('xyz') extrinsic: R = Rz(yaw) Ry(pitch) Rx(roll)  (Object.py:97)
MRS_DEV void matrix_to_euler(const M3 &R, double &roll, double &pitch, double &yaw)
{
    // pitch = asin(-R20) written as atan2(-R20, cos(pitch)) with cos(pitch) = |(R21, R22)| taken from the
    // matrix itself (no 1 - s^2 cancellation near +-pi/2)
    roll = fast_atan2(R.m21, R.m22);
    pitch = fast_atan2(-R.m20, sqrt64(R.m21 * R.m21 + R.m22 * R.m22));
    yaw = fast_atan2(R.m10, R.m00);
}

// from_euler('xyz', e).as_matrix() (QuadControl.py:77, :99)
MRS_DEV M3 euler_to_matrix(double roll, double pitch, double yaw)
{
    double sr, cr, sp, cp, sy, cy;
    fast_sincos(roll, &sr, &cr);
    fast_sincos(pitch, &sp, &cp);
    fast_sincos(yaw, &sy, &cy);
    M3 R;
    R.m00 = cy * cp; R.m01 = cy * sp * sr - sy * cr; R.m02 = cy * sp * cr + sy * sr;
    R.m10 = sy * cp; R.m11 = sy * sp * sr + cy * cr; R.m12 = sy * sp * cr - cy * sr;
    R.m20 = -sp;     R.m21 = cp * sr;                R.m22 = cp * cr;
    return R;
}

// from_euler('xyz', e).as_quat()  (Object.py:54-56)
MRS_DEV void euler_to_quat(double roll, double pitch, double yaw, double q[4])
{
    double sr, cr, sp, cp, sy, cy;
    fast_sincos(0.5 * roll, &sr, &cr);
    fast_sincos(0.5 * pitch, &sp, &cp);
    fast_sincos(0.5 * yaw, &sy, &cy);
    q[0] = sr * cp * cy - cr * sp * sy;
    q[1] = cr * sp * cy + sr * cp * sy;
    q[2] = cr * cp * sy - sr * sp * cy;
    q[3] = cr * cp * cy + sr * sp * sy;
}

// What the reference's getters hand to Python: float32 truncations (Object.py:78-97).
struct Observed {
    float px, py, pz, vx, vy, vz, wx, wy, wz;
    float roll, pitch, yaw;
    float r00, r01, r02, r10, r11, r12, r20, r21, r22; // get_ori(mat=True)
};

template <bool WANT_EULER, bool WANT_MAT>
MRS_DEV void observe(const double p[3], const double q[4], const double v[3], const double w[3], Observed &o)
{
    o.px = (float)p[0]; o.py = (float)p[1]; o.pz = (float)p[2];
    o.vx = (float)v[0]; o.vy = (float)v[1]; o.vz = (float)v[2];
    o.wx = (float)w[0]; o.wy = (float)w[1]; o.wz = (float)w[2];
    if (WANT_EULER || WANT_MAT) {
        // quat = torch.tensor(state[1]) is float32 BEFORE scipy normalises it (Object.py:92-93)
        const M3 R = quat_to_matrix_scipy((double)(float)q[0], (double)(float)q[1], (double)(float)q[2], (double)(float)q[3]);
        if (WANT_EULER) {
            double r, pt, y;
            matrix_to_euler(R, r, pt, y);
            o.roll = (float)r; o.pitch = (float)pt; o.yaw = (float)y;
        }
        if (WANT_MAT) {
            o.r00 = (float)R.m00; o.r01 = (float)R.m01; o.r02 = (float)R.m02;
            o.r10 = (float)R.m10; o.r11 = (float)R.m11; o.r12 = (float)R.m12;
            o.r20 = (float)R.m20; o.r21 = (float)R.m21; o.r22 = (float)R.m22;
        }
    }
}

// observe<true,true> plus Re = from_euler('xyz', float32 euler read-back).as_matrix() -- the matrix the attitude
// controller rebuilds from the angles it was handed (QuadControl.py:99) -- WITHOUT evaluating sin/cos again:
// theta = atan2(y, x) has sin = y/h, cos = x/h exactly, and its float32 rounding moves it by d = fl32(theta) - theta,
// |d| <= 2^-24 |theta|, so (sin, cos)(theta + d) = (s + c d - s d^2/2, c - s d - c d^2/2) to 1e-21.  All three angles
// share h = |(R21, R22)| = cos(pitch): ~35 instructions instead of three range-reduced sincos (~105).
MRS_DEV void rot_small(double s, double c, double d, double &so, double &co)
{
    const double t = 0.5 * d * d;
    so = __builtin_fma(c, d, s) - s * t;
    co = __builtin_fma(-s, d, c) - c * t;
}
// `rounded` (MrsParams.round_euler_readback, uniform over the launch): false takes Re = R, see below.
MRS_DEV void observe_ctrl(const double p[3], const double q[4], const double v[3], const double w[3], Observed &o, M3 &Re, bool rounded)
{
    o.px = (float)p[0]; o.py = (float)p[1]; o.pz = (float)p[2];
    o.vx = (float)v[0]; o.vy = (float)v[1]; o.vz = (float)v[2];
    o.wx = (float)w[0]; o.wy = (float)w[1]; o.wz = (float)w[2];
    const M3 R = quat_to_matrix_scipy((double)(float)q[0], (double)(float)q[1], (double)(float)q[2], (double)(float)q[3]);
    if (!rounded) {
    // Re := R.  from_euler(fl32(as_euler(R))) differs from R by the float32 rounding of the three angles
    // (<= 2^-24 |angle|, i.e. <= 2e-7 on the matrix elements): the reference's own read-back noise, not signal.
    // Only the ground-effect switch looks at the angles themselves (Quadcopter.py:80): roll < pi/2 and pitch < pi/2
    // on the float32 read-backs, restated on the matrix (fl32(roll) < fl32(pi/2) <=> roll < the float32 midpoint
    // below pi/2 <=> m22 > tan(2.19e-8) m21 or m21 < 0; pitch = asin(-m20) reaches fl32(pi/2) only at gimbal lock).
    Re = R;
    o.r00 = (float)R.m00; o.r01 = (float)R.m01; o.r02 = (float)R.m02;
    o.r10 = (float)R.m10; o.r11 = (float)R.m11; o.r12 = (float)R.m12;
    o.r20 = (float)R.m20; o.r21 = (float)R.m21; o.r22 = (float)R.m22;
    const bool roll_ok = (R.m22 > 2.19e-8 * R.m21) || (R.m21 < 0.0);
    const bool pitch_ok = !((R.m21 * R.m21 + R.m22 * R.m22) < 4.8e-16 && R.m20 < 0.0);
    o.roll = roll_ok ? 0.f : 2.f; o.pitch = pitch_ok ? 0.f : 2.f; o.yaw = 0.f; // only compared with pi/2 downstream
    return;
    }
    const double h = sqrt64(R.m21 * R.m21 + R.m22 * R.m22);
    double r, pt, y;
    r = fast_atan2(R.m21, R.m22); pt = fast_atan2(-R.m20, h); y = fast_atan2(R.m10, R.m00);
    o.roll = (float)r; o.pitch = (float)pt; o.yaw = (float)y;
    o.r00 = (float)R.m00; o.r01 = (float)R.m01; o.r02 = (float)R.m02;
    o.r10 = (float)R.m10; o.r11 = (float)R.m11; o.r12 = (float)R.m12;
    o.r20 = (float)R.m20; o.r21 = (float)R.m21; o.r22 = (float)R.m22;
    if (h > 1e-150) {
        const double rh = rcp64(h);
        double sr, cr, sp, cp, sy, cy;
        rot_small(R.m21 * rh, R.m22 * rh, (double)o.roll - r, sr, cr);
        rot_small(-R.m20, h, (double)o.pitch - pt, sp, cp);
        rot_small(R.m10 * rh, R.m00 * rh, (double)o.yaw - y, sy, cy);
        Re.m00 = cy * cp; Re.m01 = cy * sp * sr - sy * cr; Re.m02 = cy * sp * cr + sy * sr;
        Re.m10 = sy * cp; Re.m11 = sy * sp * sr + cy * cr; Re.m12 = sy * sp * cr - cy * sr;
        Re.m20 = -sp;     Re.m21 = cp * sr;                Re.m22 = cp * cr;
    } else { // gimbal lock to the last bit: the angles are whatever atan2 makes of rounding noise; take them literally
        Re = euler_to_matrix((double)o.roll, (double)o.pitch, (double)o.yaw);
    }
}

// Reciprocals of per-launch constants, computed once on the host (a wave-uniform float64 division would
// otherwise still cost every lane its ~12-instruction division sequence).
struct Recips {
    double inv_mass, inv_i0, inv_i1, inv_i2, inv_4kf, inv_dt; // host-computed: a float64 division is ~14 VALU per lane
    // float32 division by the controller's DT (QuadControl.py:62) as multiply + two fused multiply-adds, bit for bit the
    // correctly rounded quotient: r = RN(1 / dt32), q0 = RN(x r), q = RN(q0 + r (x - dt32 q0)) (Markstein; holds for every
    // divisor whose significand is not all ones -- ctrl_div_fast says so -- and checked over all 3.8e9 finite float32 x for
    // the default 0.01f).  The compiler's division is ten instructions, three times per agent-step.
    float inv_ctrl_dt32;
    int ctrl_div_fast;
};
MRS_DEV float div_ctrl_dt(float x, float dt32, const Recips &K)
{
    if (!K.ctrl_div_fast) return f32div(x, dt32);
    const float q0 = f32mul(x, K.inv_ctrl_dt32);
    // the refinement is exact arithmetic on FINITE NORMAL quotients only: x = +-inf or an overflowing x r would give
    // fma(-dt, inf, x) = NaN where the division gives +-inf (a diverged env carries inf in the reference and in the oracle),
    // and a zero or subnormal first quotient loses the sign of zero / is rounded twice -- those go through the division
    // proper.  One v_cmp_class_f32 (+-normal); measured against round 3's unguarded form: 22.95 / 23.01 against 22.51 / 22.62 us
    // per step with two compares and an or, inside the noise with the class test (tools/abl_run.sh).
    // Round 5: branch-free.  The refined quotient is formed unconditionally and a select keeps q0 where q0 is not +-normal: q0 IS
    // the division's result for x = +-inf, NaN and +-0 and for an overflowing product; a subnormal q0 (|x| < 1.2e-40) may differ from
    // the correctly rounded quotient by one subnormal ulp (1.4e-45).  The divergent branch around the division sequence cost
    // 0.3 - 0.9 us per step in a same-box A/B (gpurun_out/r4y/c3b.txt), three of them per agent-step.
#ifndef MRS_DIV_GUARD
#define MRS_DIV_GUARD 2 // A/B switch: 0 = round 3's unguarded form, 1 = round 4's branch to the division proper, 2 = select
#endif
#ifdef MRS_HOST_CHECK
    const bool normal = __builtin_isnormal(q0);
#else
    const bool normal = __builtin_amdgcn_class(q0, 0x008 | 0x100);
#endif
    if (MRS_DIV_GUARD == 1 && !normal) return f32div(x, dt32);
    const float q = __builtin_fmaf(__builtin_fmaf(-dt32, q0, x), K.inv_ctrl_dt32, q0);
    return (MRS_DIV_GUARD == 2 && !normal) ? q0 : q;
}

// Controller memory of one quadcopter, in registers for the duration of a step.
struct Pid {
    double ipx, ipy, ipz; // integral_pos_e   QuadControl.py:41-44
    double dvx, dvy, dvz; // d_vel_e          :57,:62
    double ivx, ivy, ivz; // integral_vel_e   :60-61,:66
    double iox, ioy, ioz; // integral_ori_e   :105,:108-110
    float lvx, lvy, lvz;  // last_vel_e       :56,:64   NaN = not created
    float ltx, lty, ltz;  // last_target_vel  :58-59,:65
};

// QuadControl.attitude_control (QuadControl.py:93-127).  Rt is the target rotation matrix.
// nta = |ta| and rn = 1/|ta| come from the caller, which has them already (accel_control normalises ta).
MRS_DEV void attitude_control(const MrsParams &P, const Recips &K, Pid &s, const M3 &Rt, const M3 &R, const Observed &o,
                              V3 ta, double nta, double rn, double rpm[4])
{
    // E = Rt^T R - R^T Rt ; rot_e = (E21, E02, E10)   (:101-102)
    const double a21 = Rt.m02 * R.m01 + Rt.m12 * R.m11 + Rt.m22 * R.m21;
    const double a12 = Rt.m01 * R.m02 + Rt.m11 * R.m12 + Rt.m21 * R.m22;
    const double a02 = Rt.m00 * R.m02 + Rt.m10 * R.m12 + Rt.m20 * R.m22;
    const double a20 = Rt.m02 * R.m00 + Rt.m12 * R.m10 + Rt.m22 * R.m20;
    const double a10 = Rt.m01 * R.m00 + Rt.m11 * R.m10 + Rt.m21 * R.m20;
    const double a01 = Rt.m00 * R.m01 + Rt.m10 * R.m11 + Rt.m20 * R.m21;
    const double ex = a21 - a12, ey = a02 - a20, ez = a10 - a01;
    const double dt = P.ctrl_dt;
    s.iox = clampd(s.iox - ex * dt, -1., 1.); // :108-110: clip to +-1500, then x and y to +-1 -- the second contains the first
    s.ioy = clampd(s.ioy - ey * dt, -1., 1.);
    s.ioz = clampd(s.ioz - ez * dt, -1500., 1500.);
    // :112-115  P=(7e4,7e4,6e4) I=(0,0,500) D=(2e4,2e4,1.2e4), angvel_e = 0 - angvel
    // (the x and y integral gains are zero: "+ 0 * integral" is dropped -- the integrals are clamped, hence finite)
    const double tx = clampd(-(70000. * ex) - 20000. * (double)o.wx, -3200., 3200.);
    const double ty = clampd(-(70000. * ey) - 20000. * (double)o.wy, -3200., 3200.);
    const double tz = clampd(-(60000. * ez) + 500. * s.ioz - 12000. * (double)o.wz, -3200., 3200.);
    double thrust = 0.;
    if (nta != 0) { // :117-122
        const double cosang = (ta.x * rn) * R.m02 + (ta.y * rn) * R.m12 + (ta.z * rn) * R.m22;
        thrust = rcp64(cosang > 0.2 ? cosang : 0.2) * nta * P.mass;
    }
    const double tp = (sqrt64(thrust * K.inv_4kf) - 4070.3) * (1.0 / 0.2685); // :123 (host-side reciprocals)
    // MixerMatrix (:27) rows (.5,-.5,-1) (.5,.5,1) (-.5,.5,-1) (-.5,-.5,1); clip [20000,65535]; rpm = .2685 pwm + 4070.3
    rpm[0] = 0.2685 * clampd(tp + (.5 * tx - .5 * ty - tz), 20000., 65535.) + 4070.3;
    rpm[1] = 0.2685 * clampd(tp + (.5 * tx + .5 * ty + tz), 20000., 65535.) + 4070.3;
    rpm[2] = 0.2685 * clampd(tp + (-.5 * tx + .5 * ty - tz), 20000., 65535.) + 4070.3;
    rpm[3] = 0.2685 * clampd(tp + (-.5 * tx - .5 * ty + tz), 20000., 65535.) + 4070.3;
}

// QuadControl.accel_control (QuadControl.py:73-90).  `R` = from_euler(ori float32) in float64.
MRS_DEV void accel_control(const MrsParams &P, const Recips &K, Pid &s, V3 ta_in, const M3 &R, const Observed &o,
                           double rpm[4])
{
    const V3 ta = v3(ta_in.x + 0., ta_in.y + 0., ta_in.z + P.ctrl_gravity); // :76
    const double d2 = dot(ta, ta);
    const double rn = rsqrt64(d2);
    V3 tz = v3(ta.x * rn, ta.y * rn, ta.z * rn); // :78 (|ta| = 0 -> rsq = inf, 0 * inf = NaN -> next line)
    if (isnan(tz.x) || isnan(tz.y) || isnan(tz.z)) tz = v3(0., 0., 1.); // :79-80
    // :77 rotation is cast to float32; :82 x_t = R[:,1] x z_t (not normalised); :83 y_t = z_t x x_t
    const V3 ycol = v3((double)(float)R.m01, (double)(float)R.m11, (double)(float)R.m21);
    const V3 tx = cross(ycol, tz);
    const V3 ty = cross(tz, tx);
    // :88-89 from_matrix() of the non-orthonormal [x_t y_t z_t] = nearest rotation = column
    // normalisation (columns are mutually orthogonal); :100 rebuilds the same matrix from its euler angles.
    // z_t is a unit vector and y_t = z_t x x_t is orthogonal to it, so |y_t| = |x_t| and |z_t| = 1 up to
    // one rounding: one reciprocal norm serves all three columns.
    const double nx = rsqrt64(dot(tx, tx));
    M3 Rt;
    Rt.m00 = tx.x * nx; Rt.m10 = tx.y * nx; Rt.m20 = tx.z * nx;
    Rt.m01 = ty.x * nx; Rt.m11 = ty.y * nx; Rt.m21 = ty.z * nx;
    Rt.m02 = tz.x; Rt.m12 = tz.y; Rt.m22 = tz.z;
    attitude_control(P, K, s, Rt, R, o, ta, d2 > 0.0 ? d2 * rn : 0.0, rn, rpm);
}

// QuadControl.vel_control (QuadControl.py:51-70): vel_e and the derivative numerator are float32 arithmetic
MRS_DEV V3 vel_control_accel(const MrsParams &P, const Recips &K, Pid &s, const Observed &o, float tvx, float tvy, float tvz)
{
    const float dt32 = (float)P.ctrl_dt;
    const float ex = f32sub(tvx, o.vx), ey = f32sub(tvy, o.vy), ez = f32sub(tvz, o.vz); // :54
    if (isnan(s.lvx)) { s.lvx = ex; s.lvy = ey; s.lvz = ez; s.dvx = s.dvy = s.dvz = 0.; }       // :55-57
    if (isnan(s.ltx)) { s.ltx = tvx; s.lty = tvy; s.ltz = tvz; }                                // :58-59
    // :62 d = (((e - e_last) - (tv - tv_last)) / DT) * 0.5 + d * 0.5
    const float hx = f32mul(div_ctrl_dt(f32sub(f32sub(ex, s.lvx), f32sub(tvx, s.ltx)), dt32, K), 0.5f);
    const float hy = f32mul(div_ctrl_dt(f32sub(f32sub(ey, s.lvy), f32sub(tvy, s.lty)), dt32, K), 0.5f);
    const float hz = f32mul(div_ctrl_dt(f32sub(f32sub(ez, s.lvz), f32sub(tvz, s.ltz)), dt32, K), 0.5f);
    s.dvx = (double)hx + s.dvx * 0.5; s.dvy = (double)hy + s.dvy * 0.5; s.dvz = (double)hz + s.dvz * 0.5;
    s.lvx = ex; s.lvy = ey; s.lvz = ez;       // :64
    s.ltx = tvx; s.lty = tvy; s.ltz = tvz;    // :65
    s.ivx = s.ivx + (double)f32mul(ex, dt32); // :66
    s.ivy = s.ivy + (double)f32mul(ey, dt32);
    s.ivz = s.ivz + (double)f32mul(ez, dt32);
    // :67-69 P=3 I=.1 D=1
    return v3(3. * (double)ex + .1 * s.ivx + 1. * s.dvx, 3. * (double)ey + .1 * s.ivy + 1. * s.dvy,
              3. * (double)ez + .1 * s.ivz + 1. * s.dvz);
}

// QuadControl.pos_control (QuadControl.py:35-48)
MRS_DEV V3 pos_control_accel(const MrsParams &P, Pid &s, const Observed &o, float tpx, float tpy, float tpz)
{
    const float dt32 = (float)P.ctrl_dt;
    const float ex = f32sub(tpx, o.px), ey = f32sub(tpy, o.py), ez = f32sub(tpz, o.pz); // :40
    s.ipx = s.ipx + (double)f32mul(ex, dt32); // :44
    s.ipy = s.ipy + (double)f32mul(ey, dt32);
    s.ipz = s.ipz + (double)f32mul(ez, dt32);
    // :45-47 P=1.5 I=.001 D=1 with d_pos_e = 0 - vel
    return v3(1.5 * (double)ex + .001 * s.ipx + 1. * (0.0 - (double)o.vx),
              1.5 * (double)ey + .001 * s.ipy + 1. * (0.0 - (double)o.vy),
              1.5 * (double)ez + .001 * s.ipz + 1. * (0.0 - (double)o.vz));
}

// 2-variable NNLS of the normal-equation block [[3,1],[1,3]] [x y]^T = [p q]^T, exact KKT enumeration
MRS_DEV void nnls2(double p, double q, double &x, double &y)
{
    const double xs = (3. * p - q) * 0.125, ys = (3. * q - p) * 0.125;
    if (xs >= 0. && ys >= 0.) { x = xs; y = ys; return; }
    const double y0 = q / 3., x0 = p / 3.;
    if (y0 >= 0. && p - y0 <= 0.) { x = 0.; y = y0; return; }
    if (x0 >= 0. && q - x0 <= 0.) { x = x0; y = 0.; return; }
    x = 0.; y = 0.;
}

// Quadcopter.set_control + nnlsRPM (Quadcopter.py:26-34, :172-208).  The mixer matrix A (:164)
// has A^T A = [[3,0,1,0],[0,3,0,1],[1,0,3,0],[0,1,0,3]]: the 4-variable NNLS the reference hands to
// scipy splits into two independent 2-variable problems {0,2} and {1,3}, solved in closed form.
MRS_DEV void set_control(const MrsParams &P, float c0, float c1, float c2, float c3, double rpm[4])
{
    const double thrust = (double)f32mul(c0, (float)P.mass);   // :27-30 float32 tensor * python float
    const double roll = (double)f32mul(c1, (float)P.ixx_file);
    const double pitch = (double)f32mul(c2, (float)P.iyy_file);
    const double yaw = (double)f32mul(c3, (float)P.izz_file);
    const double c = 0.70710678118654752440;
    const double B0 = thrust * (1 / P.kf), B1 = roll * (1 / (P.kf * P.arm)), B2 = pitch * (1 / (P.kf * P.arm)),
                 B3 = yaw * (1 / P.km); // :166, :202
    // sq = Ainv B with Ainv = A^T diag(1/4,1/2,1/2,1/4)
    double s0 = 0.25 * B0 + (0.5 * c) * B1 - (0.5 * c) * B2 - 0.25 * B3;
    double s1 = 0.25 * B0 + (0.5 * c) * B1 + (0.5 * c) * B2 + 0.25 * B3;
    double s2 = 0.25 * B0 - (0.5 * c) * B1 + (0.5 * c) * B2 - 0.25 * B3;
    double s3 = 0.25 * B0 - (0.5 * c) * B1 - (0.5 * c) * B2 + 0.25 * B3;
    if (fmin(fmin(s0, s1), fmin(s2, s3)) < 0) { // :204-207
        const double b0 = B0 + c * B1 - c * B2 - B3; // A^T B
        const double b1 = B0 + c * B1 + c * B2 + B3;
        const double b2 = B0 - c * B1 + c * B2 - B3;
        const double b3 = B0 - c * B1 - c * B2 + B3;
        nnls2(b0, b2, s0, s2);
        nnls2(b1, b3, s1, s3);
    }
    rpm[0] = sqrt(s0); rpm[1] = sqrt(s1); rpm[2] = sqrt(s2); rpm[3] = sqrt(s3); // :208
}

// One downwash pair term in the reference's float32 arithmetic (Quadcopter.py:103-110);
// (rx,ry,dz) = other - self.
//
// Default build: the same expression on the hardware's 1-ulp rcp / exp2 units, with the sqrt folded
// away (t^2 = dxy^2 / beta^2): ~20 VALU per pair instead of ~80.  The reference itself evaluates this
// term through numpy's SIMD float32 exp, which already differs from libm's by an ulp, so agreement
// beyond a few float32 ulps does not exist for this term (tests bound the force at 2e-5 relative).
// -DMRS_EXACT_F32=1 builds the operation-for-operation float32 form instead.
#ifndef MRS_EXACT_F32
#define MRS_EXACT_F32 0
#endif
struct DownwashConst {
    float c_alpha;  // dw1 * (prop_radius/4)^2
    float lg_alpha; // log2(c_alpha): folded into the exponent, alpha * e^x = rdz^2 * 2^(x log2 e + lg_alpha)
    float dw2, dw3;
    float pr32, dw1;
    // A pair whose dxy^2 exceeds zero_c * beta^2 (beta = dw2 |dz| + dw3) contributes an EXACT float32 zero: the term is
    // rdz^2 * exp2(lg_alpha - 0.7213 dxy^2 / beta^2) and v_exp_f32 returns 0 for arguments below -149 (-126 with denormal
    // results flushed); zero_c puts the argument below -150 with the rounding of the two reciprocals and products (< 1e-6
    // relative) to spare.  The pair loops of multi-wave envs skip a pass in which every lane's pairs are such (a wave vote).
    float zero_c;
};
MRS_DEV float downwash_zero_c(float lg_alpha) { return (150.0f + (lg_alpha > 0.f ? lg_alpha : 0.f)) * (1.0f / 0.72134752f) * 1.00001f; }
MRS_DEV DownwashConst downwash_const(const MrsParams &P)
{
    DownwashConst c;
    c.pr32 = (float)P.prop_radius; c.dw1 = (float)P.dw1; c.dw2 = (float)P.dw2; c.dw3 = (float)P.dw3;
    c.c_alpha = c.dw1 * (0.25f * c.pr32) * (0.25f * c.pr32);
    c.lg_alpha = __builtin_log2f(c.c_alpha);
    c.zero_c = downwash_zero_c(c.lg_alpha);
    return c;
}
MRS_DEV float downwash_pair_fast(float rx, float ry, float dz, const DownwashConst &c)
{
    const float d2 = rx * rx + ry * ry;
    const float rdz = __builtin_amdgcn_rcpf(dz);
    const float rb = __builtin_amdgcn_rcpf(c.dw2 * dz + c.dw3);
    const float ex = __builtin_amdgcn_exp2f(__builtin_fmaf((d2 * rb) * rb, -0.5f * 1.44269504088896341f, c.lg_alpha));
    const float f = -((rdz * rdz) * ex);
    return (dz > 0.f && d2 < 100.f) ? f : 0.f;   // Quadcopter.py:105: delta_z > 0 and delta_xy < 10
}
// The pair term as a function of (dxy^2, |dz|) only -- it is applied to whichever quadcopter of the
// pair is lower (Quadcopter.py:105: delta_z > 0), so one evaluation serves both orderings.
MRS_DEV float downwash_mag(float rx, float ry, float adz, const DownwashConst &c)
{
    const float d2 = rx * rx + ry * ry;
    const float rdz = __builtin_amdgcn_rcpf(adz);
    const float rb = __builtin_amdgcn_rcpf(c.dw2 * adz + c.dw3);
    const float ex = __builtin_amdgcn_exp2f(__builtin_fmaf((d2 * rb) * rb, -0.5f * 1.44269504088896341f, c.lg_alpha));
    return (adz > 0.f && d2 < 100.f) ? -((rdz * rdz) * ex) : 0.f;
}
// The same term with every rounding pinned (no compiler-chosen contraction) and dxy^2 handed in: the N = 64 kernels
// evaluate a pair either at the start of a step or -- carried over -- at the end of the previous one, inside the
// adjacency loop that forms dxy^2 anyway; both places must produce the same bits.
MRS_DEV float downwash_mag2(float d2, float adz, float dw2, float dw3, float lg_alpha)
{
    const float rdz = __builtin_amdgcn_rcpf(adz);
    const float rb = __builtin_amdgcn_rcpf(__builtin_fmaf(dw2, adz, dw3));
    const float ex = __builtin_amdgcn_exp2f(__builtin_fmaf(f32mul(f32mul(d2, rb), rb), -0.5f * 1.44269504088896341f, lg_alpha));
    return (adz > 0.f && d2 < 100.f) ? -f32mul(f32mul(rdz, rdz), ex) : 0.f;
}
// The three coefficients of the pair term, pinned in vector registers for the duration of a pair loop: as kernel
// arguments they would otherwise be re-read from the argument segment (s_load + wait) inside every pair once the
// scalar registers run short, and moved to a VGPR per use (a VOP3 takes one SGPR).
struct DownwashRegs {
    float dw2, dw3, lg, nzc; // nzc = -zero_c
};
MRS_DEV DownwashRegs downwash_regs(const DownwashConst &c)
{
    DownwashRegs r = {c.dw2, c.dw3, c.lg_alpha, -c.zero_c};
    asm volatile("" : "+v"(r.dw2), "+v"(r.dw3), "+v"(r.lg), "+v"(r.nzc));
    return r;
}
MRS_DEV float downwash_pair(float rx, float ry, float dz, float pr32, float dw1, float dw2, float dw3)
{
    const float dxy = f32sqrt(f32add(f32mul(rx, rx), f32mul(ry, ry))); // np.linalg.norm(rel[:2])
    float f = 0.f;
    if (dz > 0.f && dxy < 10.f) {
        const float rc = f32div(1.0f, f32mul(4.0f, dz));   // PropRadius/(4 dz) = reciprocal()*scalar
        const float q = f32mul(rc, pr32);
        const float alpha = f32mul(dw1, f32mul(q, q));
        const float beta = f32add(f32mul(dw2, dz), dw3);
        const float t = f32mul(f32div(1.0f, beta), dxy);   // np.float32 / tensor -> reciprocal()*other
        const float ex = expf(f32mul(-.5f, f32mul(t, t)));
        f = -f32mul(alpha, ex);
    }
    return f;
}

// BulletSim.step_sim -> stepSimulation (BulletSim.py:46-47).  [BULLET-KNOWLEDGE] btMultiBody ABA for a
// floating base with massless fixed links, applyDeltaVeeMultiDof (+-max_coord_vel clamp), contact,
// stepPositionsMultiDof (exponential map with Taylor branch and angular-motion threshold).
//
// Three stages so that the (rare, register-hungry) contact solve can run in its own compacted launch:
//   integrate_velocity : ABA + applyDeltaVee           -> unconstrained v, w
//   contact_solve      : only for bodies near the ground (needs_contact)
//   integrate_pose     : stepPositionsMultiDof with the final v, w
// park_z = ground_z + sqrt(coll_radius^2 + coll_half_len^2) + contact_threshold, formed once per call on the host (as a
// per-lane expression it was a correctly rounded float64 square root, ~18 vector instructions, in every step of every
// body).  A body within one rounding of the limit may fall on the other side than it did with the subtraction form: it
// has no rim point within the threshold either way, so its solve does nothing.
MRS_DEV bool needs_contact(int enable_contact, double park_z, double pz)
{
    return enable_contact && !(pz > park_z);
}

MRS_DEV void integrate_velocity(const MrsParams &P, const Recips &K, const M3 &R, double v[3], double w[3],
                                V3 fb_ext, V3 tb_ext)
{
    // btMultiBody's articulated-body pass for a floating base (oracle/mrs_oracle.c:orc_integrate follows it term by term):
    //   vb = R^T v, wb = R^T w, cor = wb x vb, ab = (fb + R^T m g)/m - k_l (1 + |vb|) vb - cor, vdot = R (ab + cor).
    // The Coriolis term cancels and |vb| = |v|, so the linear part is formed in the world frame directly:
    //   vdot = R fb / m + g - k_l (1 + |v|) v
    // (identical up to float64 rounding, 1e-16 relative; ~25 float64 instructions fewer).  The angular part needs wb.
    // R = quat_to_matrix_bullet(q): the caller has it already for the ground effect
    const V3 vw = v3(v[0], v[1], v[2]);
    const V3 wb = mulT(R, v3(w[0], w[1], w[2]));
    // |v|, |w| only scale the quadratic damping term k |v| v (k = 0.04): float32 square roots (1e-7 relative on a
    // term that is itself ~4 % of the velocity per second) instead of two float64 ones
    const double nv = (double)__builtin_sqrtf((float)dot(vw, vw)), nw = (double)__builtin_sqrtf((float)dot(wb, wb));
    const V3 Iw = v3(P.inertia[0] * wb.x, P.inertia[1] * wb.y, P.inertia[2] * wb.z);
    const V3 gyro = P.use_gyro ? cross(wb, Iw) : v3(0., 0., 0.);
    const double kl = P.lin_damp, ka = P.ang_damp;
    const V3 fw = mul(R, fb_ext);
    const double cl = kl + kl * nv;
    const V3 vdot = v3(fw.x * K.inv_mass - cl * vw.x, fw.y * K.inv_mass - cl * vw.y, (fw.z * K.inv_mass - P.gravity) - cl * vw.z);
    const V3 alb = v3(-(-tb_ext.x + Iw.x * (ka + ka * nw) + gyro.x) * K.inv_i0,
                      -(-tb_ext.y + Iw.y * (ka + ka * nw) + gyro.y) * K.inv_i1,
                      -(-tb_ext.z + Iw.z * (ka + ka * nw) + gyro.z) * K.inv_i2);
    const V3 wdot = mul(R, alb);
    const double dt = P.dt, mv = P.max_coord_vel;
    w[0] = clampd(w[0] + wdot.x * dt, -mv, mv); w[1] = clampd(w[1] + wdot.y * dt, -mv, mv); w[2] = clampd(w[2] + wdot.z * dt, -mv, mv);
    v[0] = clampd(v[0] + vdot.x * dt, -mv, mv); v[1] = clampd(v[1] + vdot.y * dt, -mv, mv); v[2] = clampd(v[2] + vdot.z * dt, -mv, mv);
}

// Ground contact, the build's own model (DESIGN.md "Row G"; the algorithm of oracle/mrs_oracle.c:contact_solve):
// the 4 body-fixed rim points (azimuths 45 + 90 k degrees) of the cap of the collision cylinder that faces the ground
// against z = ground_z, Bullet-style velocity-level rhs, sequential impulses with a friction pyramid along world x/y.
// Geometry, gaps and the rhs are formed in float64; the sweeps run in float32 on velocity CHANGES (dv, dw), which the
// caller adds to the float64 state.  Impulses are ~m g dt = 2.6e-3 N s, so float32 carries them to ~1e-10; the stated
// tolerance against the float64 oracle is 1e-4 per step, 1e-3 over a touchdown.
#ifndef MRS_SKIP_IDLE_POINTS
#define MRS_SKIP_IDLE_POINTS 1
#endif
#ifndef MRS_CONTACT_TOL
#define MRS_CONTACT_TOL 1e-7f
#endif
#ifndef MRS_CONTACT_STAG
#define MRS_CONTACT_STAG 0.5f
#endif
#ifndef MRS_CONTACT_RELTOL
#define MRS_CONTACT_RELTOL 1
#endif
struct F3 {
    float x, y, z;
};
// two float32 in an aligned register pair: the operand of the packed instructions (v_pk_fma_f32: two fused
// multiply-adds for the issue cost of one, tools/micro/valu_rates2.hip)
typedef float F2 __attribute__((ext_vector_type(2)));
// The rows' scalar type is a template parameter: float in every kernel; double only in the CPU suite, which runs the SAME
// statements in float64 against the oracle (tests/test_device_math_host.py: 1e-12 -- what is left between the kernel and the
// oracle is float32, not the algorithm).  (Rounds 3-4 derived the float64 text from this function by regular expressions.)
template <class S> struct ContactT;
template <> struct ContactT<float> {
    typedef float S2 __attribute__((ext_vector_type(2)));
    static MRS_DEV float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static MRS_DEV float rcp(float a) { return __builtin_amdgcn_rcpf(a); }
    static MRS_DEV float med3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
    static MRS_DEV float max(float a, float b) { return fmaxf(a, b); }
    static MRS_DEV float abs(float a) { return fabsf(a); }
};
template <> struct ContactT<double> {
    typedef double S2 __attribute__((ext_vector_type(2)));
    static MRS_DEV double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static MRS_DEV double rcp(double a) { return 1.0 / a; }
    static MRS_DEV double med3(double a, double lo, double hi) { return a < lo ? lo : (a > hi ? hi : a); }
    static MRS_DEV double max(double a, double b) { return a > b ? a : b; }
    static MRS_DEV double abs(double a) { return a < 0 ? -a : a; }
};
// ---- Round 5: the equal-share start is applied in its SYMMETRIC form.  A body pressed into the ground at the velocity clamp (100 m/s
// in every component under the reference's downwash singularity) needs impulses of ~0.7 N s per rim point whose angular responses --
// 4000 rad/s per N s each -- cancel between the points; added up as four float32 products they leave 1e-7 of themselves, and the
// sweeps then spend their float32 resolution on cancelling 100 m/s.  So the start's effect on the body is formed once, as
// dv_z = n l0 / m in float64 and dw = l0 Iw ((sum of the active levers) x z) with the sum taken as n cz + (a0 - a3) ca + (a2 - a1) cb
// (the half-diagonals of a body lying on all four points cancel identically), it goes into the body's float64 velocities at once,
// and the sweeps accumulate only their corrections to the velocities it leaves: a body pressed flat hands them a residual of zero.
// Worst teacher-forced error of the fastest bodies 3.2e-4 -> 1.8e-4, of everything below 50 m/s <= 6e-5 (profiles/r05_teacher_forced.txt).
// (Built, measured and NOT kept in round 5: a float64 re-linearisation after the first pair of sweeps -- impulses applied exactly
// through float64 geometry, right-hand sides formed anew, the remaining sweeps on small corrections -- brings every class below 1e-4
// (6e-5 ... 9e-5), but its register pressure costs the step kernel scratch memory, and a k_step that touches scratch at all runs
// 2 - 3 us slower (the scratch ring throttles the resident waves); and four lanes per body in impulse space, 25 % fewer instructions on
// the solving wave (-0.6 us per step) but without the velocity feedback that makes the one-lane form forgiving: 3e-4 ... 9e-4.)
// One lane per body, the rows in velocity space.  `body` is the body's float64 velocities: load(v, w) reads them, add(dv, dw) applies
// a change -- the start's at once, the sweeps' float32 corrections at the end (the fused kernel keeps them in its LDS stash: no
// float64 word is live across the sweeps).
template <class S, class Body>
MRS_DEV void contact_solve_rows(const MrsParams &P, const Recips &K, double pz, const double q[4], Body &&body, float *diag = nullptr)
{
    // Every product-sum below is written out with explicit fused multiply-adds under "contract(off)": the function is inlined
    // into several kernels (one-launch step, k_contact) and must round the same way in all of them.
#pragma clang fp contract(off)
    typedef ContactT<S> M;
    typedef typename M::S2 S2;
    struct S3 { S x, y, z; };
    // Everything from here on is S (float32 in the kernels), the geometry included (round 3; round 2 formed the rotation matrix, the rim points'
    // heights, the gaps and the right-hand sides in float64: ~150 float64 instructions and conversions at the head of the
    // workgroup's one serial chain).  What float32 costs: the levers (|r| <= 0.061 m) carry the quaternion's rounding, < 1e-8 m;
    // the gap is the float32 of (pz - ground_z), rounded once from the float64 difference, plus the lever's z, < 2e-9 m off;
    // times 1 / dt that is < 2e-7 m/s on a right-hand side -- three orders inside the stated 1e-4 per step.
    const S im = (S)K.inv_mass;
    const auto fm = [](S a, S b, S c) { return M::fma(a, b, c); };
    bool any = false, cap_down = false;
    S ln[4], lx[4], ly[4], Kn[4], Kx[4], Ky[4], rhs[4], gapv[4];
    // per point, constant over the sweeps: lever r and the angular responses Iw (r x d) of the three rows.
    // Branch-free: every lane prepares all four rim points; a point that is not within the contact threshold gets
    // zero effective masses, which turns its three rows into exact no-ops (impulses stay 0) -- no per-point
    // exec-mask round trips in the sweeps and nothing to zero-initialise.
    S3 r[4];
    S2 anxy[4], axxy[4], ayxy[4]; // x, y of the angular responses as pairs: dw.xy += a.xy * dl is one packed fma
    S anz[4], axz[4], ayz[4];
    S rsx = (S)0, rsy = (S)0, nact = (S)0; // the start's lever sum (x, y) and the number of active points
    S Ixx, Ixy, Ixz, Iyy, Iyz;
    // the geometry, the responses and the effective masses from the quaternion
    const auto prepare = [&](S qx, S qy, S qz, S qw) {
        const S c = (S)(P.coll_radius * 0.70710678118654752440), hl = (S)P.coll_half_len;
        const S i0 = (S)K.inv_i0, i1 = (S)K.inv_i1, i2 = (S)K.inv_i2;
        // btMatrix3x3::setRotation of the quaternion, s = 2 / |q|^2
        const S s2 = (S)2 * M::rcp(fm(qx, qx, fm(qy, qy, fm(qz, qz, qw * qw))));
        const S xs = qx * s2, ys = qy * s2, zs = qz * s2;
        const S r00 = (S)1 - fm(qy, ys, qz * zs), r01 = fm(qx, ys, -(qw * zs)), r02 = fm(qx, zs, qw * ys);
        const S r10 = fm(qx, ys, qw * zs), r11 = (S)1 - fm(qx, xs, qz * zs), r12 = fm(qy, zs, -(qw * xs));
        const S r20 = fm(qx, zs, -(qw * ys)), r21 = fm(qy, zs, qw * xs), r22 = (S)1 - fm(qx, xs, qy * ys);
        // world inverse inertia R diag(i0, i1, i2) R^T
        const S a00 = r00 * i0, a01 = r01 * i1, a02 = r02 * i2, a10 = r10 * i0, a11 = r11 * i1, a12 = r12 * i2;
        Ixx = fm(a00, r00, fm(a01, r01, a02 * r02)); Ixy = fm(a00, r10, fm(a01, r11, a02 * r12)); Ixz = fm(a00, r20, fm(a01, r21, a02 * r22));
        Iyy = fm(a10, r10, fm(a11, r11, a12 * r12)); Iyz = fm(a10, r20, fm(a11, r21, a12 * r22));
        const S Izz = fm(r20 * i0, r20, fm(r21 * i1, r21, (r22 * i2) * r22));
        // only the rim of the cap facing the ground carries contacts (oracle: lower_cap): fold its sign into the cap's offset
        cap_down = r22 >= (S)0;
        const S shl = cap_down ? -hl : hl;
        const S3 cz = {shl * r02, shl * r12, shl * r22};
        // the four rim points (+-c, +-c) in the cap's plane: centre offset +- (cx + cy) and +- (cx - cy)
        const S3 ca = {c * (r00 + r01), c * (r10 + r11), c * (r20 + r21)}, cb = {c * (r00 - r01), c * (r10 - r11), c * (r20 - r21)};
        const S d0 = (S)(pz - P.ground_z);
        const S thr = (S)P.contact_threshold, rdt = (S)K.inv_dt, erdt = (S)(P.erp * K.inv_dt);
        any = false; nact = (S)0;
        S aact[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // k & 1: -cx, k & 2: -cy  =>  k = 0: +ca, 1: -cb, 2: +cb, 3: -ca
            const S3 &o = (k == 0 || k == 3) ? ca : cb;
            const bool plus = (k == 0 || k == 2);
            const S rx = plus ? cz.x + o.x : cz.x - o.x, ry = plus ? cz.y + o.y : cz.y - o.y, rz = plus ? cz.z + o.z : cz.z - o.z;
            r[k] = S3{rx, ry, rz};
            const S dist = d0 + rz;
            const bool act = dist <= thr;
            any |= act;
            // Iw u for u = r x z = (ry,-rx,0), r x x = (0,rz,-ry), r x y = (-rz,0,rx)
            const S3 an = {fm(Ixx, ry, -(Ixy * rx)), fm(Ixy, ry, -(Iyy * rx)), fm(Ixz, ry, -(Iyz * rx))};
            const S3 ax = {fm(Ixy, rz, -(Ixz * ry)), fm(Iyy, rz, -(Iyz * ry)), fm(Iyz, rz, -(Izz * ry))};
            const S3 ay = {fm(Ixz, rx, -(Ixx * rz)), fm(Iyz, rx, -(Ixy * rz)), fm(Izz, rx, -(Ixz * rz))};
            anxy[k] = S2{an.x, an.y}; anz[k] = an.z; axxy[k] = S2{ax.x, ax.y}; axz[k] = ax.z; ayxy[k] = S2{ay.x, ay.y}; ayz[k] = ay.z;
            // effective masses 1 / (1/m + (r x d) . Iw (r x d))
            const S kn = M::rcp(fm(ry, an.x, fm(-rx, an.y, im)));
            const S kx = M::rcp(fm(rz, ax.y, fm(-ry, ax.z, im)));
            const S ky = M::rcp(fm(rx, ay.z, fm(-rz, ay.x, im)));
            Kn[k] = act ? kn : (S)0; Kx[k] = act ? kx : (S)0; Ky[k] = act ? ky : (S)0;
            // Bullet-style target normal velocity: open gap -> let the point close it this step; penetration -> erp push-out
            gapv[k] = -dist * (dist > (S)0 ? rdt : erdt);
            aact[k] = act ? (S)1 : (S)0;
            nact += aact[k];
            __builtin_amdgcn_sched_barrier(0); // one point after the other: interleaved, the four points' temporaries cost three spilled registers
        }
        // the active levers' sum in its symmetric form, n cz + (a0 - a3) ca + (a2 - a1) cb (x, y: rs x z = (rs.y, -rs.x, 0))
        const S sa = aact[0] - aact[3], sb = aact[2] - aact[1];
        rsx = fm(nact, cz.x, fm(sa, ca.x, sb * cb.x)); rsy = fm(nact, cz.y, fm(sa, ca.y, sb * cb.y));
    };
    prepare((S)q[0], (S)q[1], (S)q[2], (S)q[3]);
    if (!any) return;
    const S mu = (S)P.friction;
    // Start of the sweeps (the oracle's contact_solve does the same): every active point carries the equal share of the
    // impulse that stops the mean closing velocity, l0 = m max(sum rhs, 0) / n^2 -- exact for a body lying flat, which
    // then needs no iteration; the sweeps correct it for everything else.  From a cold start 90 % of the grounded
    // bodies needed 8-10 sweeps, with this start 86 % are done after the first pair (tools/probes/sweeps_probe.py).
    V3 v, w;
    body.load(v, w);
    S v0x = (S)v.x, v0y = (S)v.y, v0z = (S)v.z, w0x = (S)w.x, w0y = (S)w.y, w0z = (S)w.z; // the unconstrained velocities as they come
    S rsum = (S)0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lx[k] = ly[k] = (S)0;
        rhs[k] = gapv[k] - fm(w0x, r[k].y, fm(-w0y, r[k].x, v0z));
        rsum += Kn[k] != (S)0 ? rhs[k] : (S)0;
    }
    const S l0 = M::max(rsum, (S)0) * M::rcp(nact * nact * im); // m * mean(rhs) / n
    const S rest = (S)(P.mass * P.gravity * P.dt);
    // the start's effect goes into the body's float64 velocities now, in its symmetric form (above); the changes dv, dw the rows
    // accumulate start from zero.  (For every body alike: choosing per body by the size of the start -- built -- saves the common
    // path ~50 instructions but makes a body's roundings depend on a threshold, and the few bodies whose sweeps do not converge
    // amplify any such difference; a wave vote on it would make results depend on which bodies share a wave.)
    S dvx = (S)0, dvy = (S)0, dvz = (S)0, dwz = (S)0;
    S2 dwxy = {(S)0, (S)0};
    {
        const V3 bv = v3(0., 0., (double)(nact * l0) * K.inv_mass);
        const V3 bw = v3((double)(l0 * fm(Ixx, rsy, -(Ixy * rsx))), (double)(l0 * fm(Ixy, rsy, -(Iyy * rsx))), (double)(l0 * fm(Ixz, rsy, -(Iyz * rsx))));
        body.add(bv, bw);
        v0z = (S)(v.z + bv.z); w0x = (S)(w.x + bw.x); w0y = (S)(w.y + bw.y); w0z = (S)(w.z + bw.z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ln[k] = Kn[k] != (S)0 ? l0 : (S)0;
            rhs[k] = gapv[k] - fm(w0x, r[k].y, fm(-w0y, r[k].x, v0z));
        }
    }
    // the tangential point velocities of that state, per point: the friction rows then need only the CHANGES dv, dw (3 fused
    // operations per row instead of re-forming v0 + dv, w0 + dw every time)
    S c0x[4], c0y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c0x[k] = fm(w0y, r[k].z, fm(-w0z, r[k].y, v0x));
        c0y[k] = fm(w0z, r[k].x, fm(-w0x, r[k].z, v0y));
    }
    // the sweeps gain ~1.5 digits each (measured on the oracle); a lane stops once a whole sweep moved no
    // impulse by more than 1e-7 of the resting impulse m g dt -- float32 cannot resolve less anyway --
    // and the wave leaves the loop when its last lane has (at most solver_iters sweeps, like the oracle).
    // Convergence is looked at on every second sweep only (the bookkeeping is ~8 % of a sweep).
    const S tol = (S)MRS_CONTACT_TOL * rest + (S)1e-30f;
    // a rim point that is within the threshold for NO body of the wave is skipped: its rows are exact no-ops (zero effective masses,
    // impulses stay 0), so this changes no result and no body's result depends on its wave-mates -- only the time does
    bool pt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if defined(MRS_HOST_CHECK) || !MRS_SKIP_IDLE_POINTS
        pt[k] = true;
#else
        pt[k] = __builtin_amdgcn_ballot_w64(Kn[k] != (S)0) != 0;
#endif
    }
    auto sweep = [&](auto track, S &moved) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!pt[k]) continue;
            const S rx = r[k].x, ry = r[k].y, rz = r[k].z;
            { // normal: u = (ry, -rx, 0)
                const S dvn = fm(-dwxy.y, rx, fm(dwxy.x, ry, dvz));
                const S nl = M::max(fm(Kn[k], rhs[k] - dvn, ln[k]), (S)0);
                const S dl = nl - ln[k];
                ln[k] = nl;
                if (track) moved = M::max(moved, M::abs(dl));
                dvz = fm(dl, im, dvz);
                dwxy = __builtin_elementwise_fma(anxy[k], S2{dl, dl}, dwxy); dwz = fm(anz[k], dl, dwz);
            }
            const S lim = mu * ln[k];
            { // friction x: u = (0, rz, -ry)
                const S vt = fm(-dwz, ry, fm(dwxy.y, rz, c0x[k] + dvx));
                const S nl = M::med3(fm(-Kx[k], vt, lx[k]), -lim, lim); // friction pyramid: one v_med3_f32
                const S dl = nl - lx[k];
                lx[k] = nl;
                if (track) moved = M::max(moved, M::abs(dl));
                dvx = fm(dl, im, dvx);
                dwxy = __builtin_elementwise_fma(axxy[k], S2{dl, dl}, dwxy); dwz = fm(axz[k], dl, dwz);
            }
            { // friction y: u = (-rz, 0, rx)
                const S vt = fm(-dwxy.x, rz, fm(dwz, rx, c0y[k] + dvy));
                const S nl = M::med3(fm(-Ky[k], vt, ly[k]), -lim, lim);
                const S dl = nl - ly[k];
                ly[k] = nl;
                if (track) moved = M::max(moved, M::abs(dl));
                dvy = fm(dl, im, dvy);
                dwxy = __builtin_elementwise_fma(ayxy[k], S2{dl, dl}, dwxy); dwz = fm(ayz[k], dl, dwz);
            }
        }
    };
    int it = 0;
    S prev_moved = (S)3.0e38f;
    // one pair of sweeps and the stopping rule; returns true when the body is done
    const auto sweep_pair = [&]() -> bool {
        S moved = (S)0;
        sweep(std::false_type{}, moved);
        if (it + 1 >= P.solver_iters) return true;
        sweep(std::true_type{}, moved);
#ifdef MRS_TIMELINE // diagnostic: per lane, the first even sweep count at which it had converged
        if (diag && diag[0] == 0.f && moved <= tol) diag[0] = (float)(it + 2);
#endif
#if MRS_CONTACT_RELTOL
        // ... of the resting impulse or of the largest impulse of this body, whichever is larger: a touchdown's impulses are
        // ~100 resting ones, and float32 sweeps cannot move them by less than 6e-8 of themselves -- measured against the
        // resting impulse alone such a body never "converges" and keeps its whole wave in the loop for all the sweeps
        if (moved <= M::max(tol, (S)MRS_CONTACT_TOL * M::max(M::max(ln[0], ln[1]), M::max(ln[2], ln[3])))) return true;
#else
        if (moved <= tol) return true;
#endif
        // ... or once a pair of sweeps has moved the impulses by at least half of what the pair before it did: 4 % of the
        // grounded bodies never get below the tolerance (the clamps of the friction pyramid chatter), ten sweeps leave them
        // no better off than four, and each of them kept its whole wave of 64 in the loop (tools/probes/sweeps_probe.py)
        if (moved >= (S)MRS_CONTACT_STAG * prev_moved) return true;
        prev_moved = moved;
        it += 2;
        return !(it < P.solver_iters);
    };
    bool done = !(it < P.solver_iters);
    while (!done) done = sweep_pair();
#ifdef MRS_TIMELINE
    if (diag) diag[1] = (float)(it + 2 < P.solver_iters ? it + 2 : P.solver_iters);
#endif
    body.add(v3((double)dvx, (double)dvy, (double)dvz), v3((double)dwxy.x, (double)dwxy.y, (double)dwz));
}


// A body LYING FLAT AT REST on the ground (nine out of ten grounded bodies of a rollout: crashed quadcopters whose rotor
// torques cancel) is the one case in which the sweeps of contact_solve_f32 have nothing to do: all four rim points are
// active with the same gap and the same closing velocity, the equal-share start IS the solution (every point carries a
// quarter of the impulse that stops the body, dv_z = max(rhs, 0)), and the friction rows hold the four points where they
// are: v_xy = 0, w = 0.  Evaluated here in the body's own lane: such a body is never listed for the solve.
// "At rest": |R20|, |R21|, |w|, |v_xy| < MRS_REST_EPS = 1e-6 -- the float32 sweeps leave residuals of 1e-8 ... 1e-7 on a body they
// have just put down, so a tighter bound is never met on the GPU (measured: with 1e-9 no body ever took this path) -- and
// the common gap not within that margin of the contact threshold.  Against the sweeps' own result: the rim points' gaps
// differ by < 1.2e-7 m, i.e. their right-hand sides by < 1.2e-5 m/s, and the velocities set to zero here are ones the
// sweeps would have brought to within 1e-7 of zero; 1e-5 m/s per step is inside the stated 1e-4 (oracle/mrs_oracle.c:
// contact_solve has no such shortcut; the parity tests compare the two through touchdown and rest).
// A body that is being lifted (rhs <= 0: no normal impulse, hence no friction) keeps its velocities.
// Returns true when the body is dealt with (v, w updated, or no point within the threshold).
#ifndef MRS_REST_EPS
#define MRS_REST_EPS 1e-6
#endif
// Round 4: the same argument does not need the body to be at REST, only FLAT.  For a body lying flat (|R20|, |R21| < MRS_FLAT_EPS, all
// four rim points within the threshold with one common gap) with ANY unconstrained velocity (v, w) the rows have two closed-form
// fixed points, and the sweeps -- whose equal-share start already is most of the way there -- converge to them:
//   (A) lifting: every rim point's right-hand side rhs_k = (u - v_z) - (w x r_k)_z is <= 0 (u = the common target normal
//       velocity -gap/dt or -erp gap/dt; bounded by (u - v_z) + (|w_x| + |w_y|) r, the levers' x, y components being <= r): no
//       normal impulse anywhere, hence no friction -- the sweeps are an exact no-op, the body keeps its velocities;
//   (B) sticking: the contact can hold the body -- post-solve v = (0, 0, u), w = 0 -- iff impulses p_k at the four points exist with
//       sum p_k = P = m (v_post - v), sum r_k x p_k = L = -I w (I = diag(I0, I0, I2) in the world frame too: the body is flat and
//       I0 = I1), normal parts >= 0 and tangential parts inside the pyramid mu n_k.  Tested conservatively, yaw-free and without a
//       square root: the friction forces sum to (P_x, P_y) and act hl below the centre, so the normals must supply the torque
//       M = (L_x - hl P_y, L_y + hl P_x) about the horizontal axes; the distribution n_k = P_z / 4 +- ... over the square of
//       half-side r / sqrt 2 has its smallest member >= n_min = P_z / 4 - |M| / (2 r), the tangential parts are at most
//       f_max = max(|P_x|, |P_y|) / 4 + |L_z| / (4 r) per axis; n_min > 0 and f_max <= mu n_min  <=>  s = 2 r (P_z / 4 - f_max / mu) > 0
//       and |M|^2 < s^2.
// Measured on the oracle (tools/contact_lab.py captures of the four BASELINE workloads, 70 000 contact problems): the bodies
// that pass (B) end within 2.2e-6 m/s (worst; 99 %: 3e-7) of what 400 float64 sweeps give them and within the same of the
// shipped 10, (A) is exact; 88 % of C2's contact problems (a swarm lying on the ground under hover thrust +- 5 %), 61 % of C4's,
// 5 % of C5's, 1 % of C3's -- the bodies whose float32 sweeps used to end 1e-4 ... 2.5e-4 rad/s from the float64 ones
// (tests/test_gpu_teacher.py) now end at exactly zero.  The margin factor keeps the test strictly inside the pyramid.
// The rest case above is the special case w = 0, v_xy = 0.  MrsParams.rest_shortcut = 0 sends every body through the sweeps.
// How flat is flat (MRS_FLAT_EPS): the closed forms ignore the tilt, i.e. the levelling rotation ~ tilt / dt that the rows would
// give the body, and a body that is finished here every step is never levelled.  Teacher-forced against the oracle (500 steps of
// C3 / C2 / C4, tools/probes/ab_eps_r4.sh, profiles/r04_ab.txt), error of the bodies lying still before the step, median / 99 %:
// bound 1e-6 (round 3's, then for bodies at rest only): 4.9e-5 / 8.4e-5; 3e-7: 1.5e-5 / 2.5e-5; 1e-7: 4.9e-6 / 8.6e-6;
// 3e-8: 1.4e-6 / 2.7e-6 (rest-only shortcut at 1e-6, round 3: 2.2e-6 / 4.3e-6 -- its bodies had been levelled by the sweeps first).
// The step's duration: C4 45.0 us at every bound against 47.6 rest-only; C3 22.3 at 1e-6 against 22.8 at the tighter bounds and 22.9
// rest-only -- the half microsecond would have been bought with the levelling of bodies tilted by up to a microradian, and was left.
// Round 4, last: the oracle has the same closed forms (oracle/mrs_oracle.c:contact_flat_closed_form) -- they are the fixed point of
// the rows, the sweeps' ten-sweep result is what is approximate (an outlier of the teacher-forced C4 run: a flat body sliding at
// 2.2 m/s; closed form and 400 sweeps agree to 1e-6, ten sweeps are 2.3e-3 away) -- so the bound is a statement about the MODEL, not
// about parity: a body within a microradian of flat is treated as flat, in the kernel and in the oracle alike, as round 3's
// at-rest shortcut already did.
#ifndef MRS_FLAT_EPS
#define MRS_FLAT_EPS 1e-6
#endif
MRS_DEV bool contact_at_rest(const MrsParams &P, const Recips &K, double pz, const double q[4], double v[3], double w[3])
{
    const double eps = MRS_REST_EPS;
    // third row of btMatrix3x3::setRotation for a unit quaternion (|q|^2 - 1 ~ 1e-16 after every step's normalisation; a
    // caller's own non-unit quaternion fails the tests below or changes the bound by its relative error)
    const double r20 = 2.0 * (q[0] * q[2] - q[3] * q[1]), r21 = 2.0 * (q[1] * q[2] + q[3] * q[0]);
    const double r22 = 1.0 - 2.0 * (q[0] * q[0] + q[1] * q[1]);
    const double dist = (pz - P.ground_z) - P.coll_half_len * fabs(r22);
    if (!(fmax(fabs(r20), fabs(r21)) < MRS_FLAT_EPS) || !(fabs(dist - P.contact_threshold) > eps)) return false;
#ifndef MRS_FLAT_SHORTCUT
#define MRS_FLAT_SHORTCUT 1 // A/B switch (tools/abl_build.sh): 0 = round 3's form, bodies at rest only
#endif
    if (!MRS_FLAT_SHORTCUT && !(fmax(fmax(fabs(w[0]), fabs(w[1])), fmax(fmax(fabs(w[2]), fabs(v[0])), fabs(v[1]))) < eps)) return false;
    if (dist > P.contact_threshold) return true; // flat and clear of the ground: no point within the threshold
    const double u = -dist * (dist > 0 ? K.inv_dt : P.erp * K.inv_dt); // the rim points' common target normal velocity
    const double up = u - v[2];
    if (up + (fabs(w[0]) + fabs(w[1])) * P.coll_radius <= 0.0) return true; // (A) lifting: no impulse at any point
    if (!(P.inertia[0] == P.inertia[1])) return false;
    // (B) sticking
    const double Pz = P.mass * up, Px = -P.mass * v[0], Py = -P.mass * v[1];
    const double Mx = -P.inertia[0] * w[0] - P.coll_half_len * Py, My = -P.inertia[0] * w[1] + P.coll_half_len * Px;
    const double fm = 0.25 * fmax(fabs(Px), fabs(Py)) + fabs(P.inertia[2] * w[2]) * (0.25 / P.coll_radius);
    const double s = 2.0 * P.coll_radius * (0.25 * Pz - fm / P.friction);
    // strictly inside (1 % of margin on the torque, a resting impulse's 1e-6 on s): a body ON the boundary goes to the sweeps
    if (!(s > 8.0 * eps * P.mass) || !((Mx * Mx + My * My) * 1.02 < s * s)) return false;
    v[0] = v[1] = 0.0; v[2] = u;
    w[0] = w[1] = w[2] = 0.0;
    return true;
}

// the body's float64 velocities as the solvers see them: load = read them as they are now, add = apply a change
struct ContactBodyRegs {
    double *v, *w;
    MRS_DEV void load(V3 &vv, V3 &ww) const { vv = v3(v[0], v[1], v[2]); ww = v3(w[0], w[1], w[2]); }
    MRS_DEV void add(const V3 &dv, const V3 &dw) const { v[0] += dv.x; v[1] += dv.y; v[2] += dv.z; w[0] += dw.x; w[1] += dw.y; w[2] += dw.z; }
};
MRS_DEV void contact_stage(const MrsParams &P, const Recips &K, const double p[3], const double q[4], double v[3], double w[3])
{
    contact_solve_rows<float>(P, K, p[2], q, ContactBodyRegs{v, w});
}

MRS_DEV void integrate_pose(double dt, double p[3], double q[4], const double v[3], const double w[3]);
MRS_DEV void integrate_pose(const MrsParams &P, double p[3], double q[4], const double v[3], const double w[3])
{
    integrate_pose(P.dt, p, q, v, w);
}
MRS_DEV void integrate_pose(double dt, double p[3], double q[4], const double v[3], const double w[3])
{
    // btMultiBody::stepPositionsMultiDof for the floating base: p += dt v; q <- dorn * q with the exponential map
    //   fAngle = |w|;  if (fAngle dt > pi/4) fAngle = pi / (4 dt);
    //   axis = w * (fAngle < 0.001 ? dt/2 - dt^3 0.020833333333 fAngle^2 : sin(fAngle dt / 2) / fAngle);  dorn = (axis, cos(fAngle dt / 2))
    // in terms of x = (half angle)^2 = dt^2 |w|^2 / 4, without the square root and the division:
    //   sin(h) / fAngle = (dt/2) sinc(h) = (dt/2) S(x),  cos(h) = C(x),  and the angular-motion threshold is x <= (pi/8)^2
    // (at the threshold both forms agree, so clamping x reproduces that branch; Bullet's small-angle Taylor branch is
    // the first two terms of S, its remainder < 1e-25).  S, C: Taylor to x^7 / x^8, remainder < 1e-17 on x <= (pi/8)^2.
    // (written as the fused multiply-add it is: the one-launch step forms the same expression ahead of time for its
    // adjacency pass and the two must agree to the bit)
    p[0] = __builtin_fma(dt, v[0], p[0]); p[1] = __builtin_fma(dt, v[1], p[1]); p[2] = __builtin_fma(dt, v[2], p[2]);
    const double x = fmin(0.25 * (dt * dt) * (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]), 0.15421256876702122);
    double S = fma_c(x, 1.0 / 1307674368000.0, -1.0 / 6227020800.0);
    S = fma_c(S, x, 1.0 / 39916800.0);
    S = fma_c(S, x, -1.0 / 362880.0);
    S = fma_c(S, x, 1.0 / 5040.0);
    S = fma_c(S, x, -1.0 / 120.0);
    S = fma_c(S, x, 1.0 / 6.0);
    S = __builtin_fma(-S, x, 1.0);
    double C = fma_c(x, -1.0 / 1307674368000.0 / 16.0 * 1.0, 1.0 / 87178291200.0);
    C = fma_c(C, x, -1.0 / 479001600.0);
    C = fma_c(C, x, 1.0 / 3628800.0);
    C = fma_c(C, x, -1.0 / 40320.0);
    C = fma_c(C, x, 1.0 / 720.0);
    C = fma_c(C, x, -1.0 / 24.0);
    C = fma_c(C, x, 0.5);
    const double dw = __builtin_fma(-C, x, 1.0);
    const double sc = (0.5 * dt) * S;
    const double ax = w[0] * sc, ay = w[1] * sc, az = w[2] * sc;
    const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const double nx = dw * qx + ax * qw + ay * qz - az * qy;
    const double ny = dw * qy + ay * qw + az * qx - ax * qz;
    const double nz = dw * qz + az * qw + ax * qy - ay * qx;
    const double nw2 = dw * qw - ax * qx - ay * qy - az * qz;
    // |dorn| = 1 to rounding unless the threshold clamped x (|w| dt > pi/4): 1/sqrt(1 + e) = 1 - e/2 + 3e^2/8 is exact then
    const double e = (nx * nx + ny * ny + nz * nz + nw2 * nw2) - 1.0;
    double rnn;
    if (__builtin_amdgcn_ballot_w64(fabs(e) > 1e-8) == 0) rnn = __builtin_fma(e, __builtin_fma(e, 0.375, -0.5), 1.0);
    else rnn = rsqrt64(e + 1.0);
    q[0] = nx * rnn; q[1] = ny * rnn; q[2] = nz * rnn; q[3] = nw2 * rnn;
}

} // namespace mrs
