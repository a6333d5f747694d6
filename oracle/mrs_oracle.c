/*
 * mrs_oracle.c -- CPU restatement of the mrsgym step() hot path (see mrs_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, imported by or called from the product.
 *
 * Arithmetic mirrors the reference's mixed precision: every value read back from
 * Bullet is truncated to float32 (Object.py:78-97), then numpy runs float32 or
 * float64 depending on operand dtypes (NEP-50 rules, numpy 2.2).  Where the
 * reference computes in float32 the code below uses `float` explicitly.
 *
 * Build with -ffp-contract=off so no FMA contraction changes the rounding.
 */
#include "mrs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.14159265358979323846

/* ------------------------------------------------------------------ params */

void orc_params_default(OrcParams *p)
{
    memset(p, 0, sizeof(*p));
    /* cf2x.urdf:5 <properties .../> */
    p->arm = 0.0397; p->kf = 3.16e-10; p->km = 7.94e-12; p->thrust2weight = 2.25;
    p->gnd_eff_coeff = 11.36859; p->prop_radius = 2.31348e-2;
    p->drag_xy = 9.1785e-7; p->drag_z = 10.311e-7;
    p->dw1 = 2267.18; p->dw2 = .16; p->dw3 = -.11;
    /* cf2x.urdf:11-12 */
    p->mass = 0.027; p->ixx_file = 1.4e-5; p->iyy_file = 1.4e-5; p->izz_file = 2.17e-5;
    /* cf2x.urdf:42,54,66,78 */
    p->prop_x[0] = 0.028;  p->prop_y[0] = 0.028;
    p->prop_x[1] = -0.028; p->prop_y[1] = 0.028;
    p->prop_x[2] = -0.028; p->prop_y[2] = -0.028;
    p->prop_x[3] = 0.028;  p->prop_y[3] = -0.028;
    /* cf2x.urdf:34 <cylinder radius=".06" length=".025"/> */
    p->coll_radius = 0.06; p->coll_half_len = 0.0125;
    /* BulletSim.py:13-14, :52-57 */
    p->gravity = 9.81; p->dt = 0.01; p->ctrl_gravity = 9.81; p->ctrl_dt = 0.01;
    /* [BULLET-KNOWLEDGE] no URDF_USE_INERTIA_FROM_FILE (EnvCreator.py:60): Bullet takes the
     * inertia of the compound collision shape's AABB box; the cylinder is imported as a convex
     * hull with margin 0.001 that enters the AABB twice (recalcLocalAabb + getAabb). */
    {
        double hx = p->coll_radius + 0.002, hz = p->coll_half_len + 0.002;
        double lx = 2 * hx, lz = 2 * hz;
        p->inertia[0] = p->mass / 12.0 * (lx * lx + lz * lz);
        p->inertia[1] = p->mass / 12.0 * (lx * lx + lz * lz);
        p->inertia[2] = p->mass / 12.0 * (lx * lx + lx * lx);
    }
    /* btMultiBody ctor: m_linearDamping(0.04f), m_angularDamping(0.04f) (float literals) */
    p->lin_damp = (double)0.04f; p->ang_damp = (double)0.04f;
    p->max_coord_vel = 100.0; p->use_gyro = 1;
    /* plane.urdf:24 box 30 30 1 centred at the origin, placed at pos 0 (EnvCreator.py:11) */
    p->ground_z = 0.5;
    p->friction = 1.5 * 0.5; /* plane.urdf:5 lateral 1.5 x default link friction 0.5 */
    p->erp = 0.2; p->contact_threshold = 0.02; p->solver_iters = 10; p->enable_contact = 1; p->pair_contact = 1; p->rest_shortcut = 1;
}

void orc_derived(const OrcParams *p, double out[7])
{
    /* Quadcopter.calculate_parameters, Quadcopter.py:156-162 */
    double gf = p->gravity * p->mass;
    double hover = sqrt(gf / (4 * p->kf));
    double maxrpm = sqrt((p->thrust2weight * gf) / (4 * p->kf));
    double maxthrust = 4. * p->kf * maxrpm * maxrpm;
    out[0] = gf; out[1] = hover; out[2] = maxrpm; out[3] = maxthrust;
    out[4] = sqrt(2.0) * p->arm * p->kf * maxrpm * maxrpm;
    out[5] = 2. * p->km * maxrpm * maxrpm;
    out[6] = 0.25 * p->prop_radius * sqrt((15 * maxrpm * maxrpm * p->kf * p->gnd_eff_coeff) / maxthrust);
}

void orc_pid_init(OrcPid *pid, int n)
{
    for (int i = 0; i < n; ++i) {
        memset(&pid[i], 0, sizeof(OrcPid));
        for (int k = 0; k < 3; ++k) { pid[i].last_vel_e[k] = NAN; pid[i].last_target_vel[k] = NAN; }
    }
}

/* --------------------------------------------------------------- rotations */

void orc_euler_to_quat(const double e[3], double q[4])
{
    /* R.from_euler('xyz', e) extrinsic: q = qz * qy * qx  (Object.py:55) */
    double cr = cos(0.5 * e[0]), sr = sin(0.5 * e[0]);
    double cp = cos(0.5 * e[1]), sp = sin(0.5 * e[1]);
    double cy = cos(0.5 * e[2]), sy = sin(0.5 * e[2]);
    q[0] = sr * cp * cy - cr * sp * sy;
    q[1] = cr * sp * cy + sr * cp * sy;
    q[2] = cr * cp * sy - sr * sp * cy;
    q[3] = cr * cp * cy + sr * sp * sy;
}

void orc_quat_to_matrix(const double qin[4], double R[9])
{
    /* scipy Rotation.as_matrix on the normalised quaternion (Object.py:93-95) */
    double n = sqrt(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
    double x = qin[0] / n, y = qin[1] / n, z = qin[2] / n, w = qin[3] / n;
    double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    R[0] = x2 - y2 - z2 + w2; R[1] = 2 * (xy - zw);      R[2] = 2 * (xz + yw);
    R[3] = 2 * (xy + zw);      R[4] = -x2 + y2 - z2 + w2; R[5] = 2 * (yz - xw);
    R[6] = 2 * (xz - yw);      R[7] = 2 * (yz + xw);      R[8] = -x2 - y2 + z2 + w2;
}

static void matrix_to_euler(const double R[9], double e[3])
{
    /* extrinsic xyz: R = Rz(yaw) Ry(pitch) Rx(roll) */
    double s = -R[6];
    if (s > 1.0) s = 1.0;
    if (s < -1.0) s = -1.0;
    e[0] = atan2(R[7], R[8]);
    e[1] = asin(s);
    e[2] = atan2(R[3], R[0]);
}

void orc_quat_to_euler(const double q[4], double e[3])
{
    double R[9];
    orc_quat_to_matrix(q, R);
    matrix_to_euler(R, e);
}

static void euler_to_matrix(const double e[3], double R[9])
{
    double q[4];
    orc_euler_to_quat(e, q);
    orc_quat_to_matrix(q, R);
}

void orc_matrix_to_euler_nearest(const double M[9], double e[3])
{
    /* R.from_matrix(M).as_euler('xyz') (QuadControl.py:89).  scipy 1.15.3 replaces a
     * non-orthogonal M by the nearest rotation (SVD); the controller's M always has mutually
     * orthogonal columns of norms (s, s, 1), for which that equals column normalisation
     * (checked against scipy in tools/gen_golden.py). */
    double Q[9];
    for (int c = 0; c < 3; ++c) {
        double n = sqrt(M[c] * M[c] + M[3 + c] * M[3 + c] + M[6 + c] * M[6 + c]);
        Q[c] = M[c] / n; Q[3 + c] = M[3 + c] / n; Q[6 + c] = M[6 + c] / n;
    }
    matrix_to_euler(Q, e);
}

void orc_observe(const double pos[3], const double quat[4], const double vel[3], const double angvel[3],
                 float opos[3], float oeuler[3], float ovel[3], float oangvel[3], float omat[9])
{
    /* Object.py:78-97: torch.tensor(tuple) -> float32; quaternion truncated before scipy */
    double q32[4], e[3], R[9];
    for (int k = 0; k < 3; ++k) {
        opos[k] = (float)pos[k]; ovel[k] = (float)vel[k]; oangvel[k] = (float)angvel[k];
    }
    for (int k = 0; k < 4; ++k) q32[k] = (double)(float)quat[k];
    orc_quat_to_matrix(q32, R);
    matrix_to_euler(R, e);
    for (int k = 0; k < 3; ++k) oeuler[k] = (float)e[k];
    if (omat) for (int k = 0; k < 9; ++k) omat[k] = (float)R[k];
}

void orc_observe_batch(int n, const double *pos, const double *quat, const double *vel, const double *angvel,
                       float *opos, float *oeuler, float *ovel, float *oangvel)
{
    for (int i = 0; i < n; ++i)
        orc_observe(pos + 3 * i, quat + 4 * i, vel + 3 * i, angvel + 3 * i, opos + 3 * i, oeuler + 3 * i,
                    ovel + 3 * i, oangvel + 3 * i, 0);
}

/* -------------------------------------------------------------- controller */

static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* QuadControl.py:15-31 */
static const double POS_P[3] = {1.5, 1.5, 1.5}, POS_I[3] = {.001, .001, .001}, POS_D[3] = {1., 1., 1.};
static const double VEL_P[3] = {3., 3., 3.}, VEL_I[3] = {.1, .1, .1}, VEL_D[3] = {1., 1., 1.};
static const double ORI_P[3] = {70000., 70000., 60000.}, ORI_I[3] = {.0, .0, 500.}, ORI_D[3] = {20000., 20000., 12000.};
static const double MIXER[4][3] = {{.5, -.5, -1}, {.5, .5, 1}, {-.5, .5, -1}, {-.5, -.5, 1}};
#define MIN_PWM 20000.0
#define MAX_PWM 65535.0
#define PWM2RPM_A 0.2685
#define PWM2RPM_B 4070.3

void orc_attitude_control(const OrcParams *p, OrcPid *s, const double target_ori[3], const float ori[3],
                          const float angvel[3], const double ta[3], double rpm[4])
{
    /* QuadControl.py:93-127 */
    double e[3] = {ori[0], ori[1], ori[2]};
    double R[9], Rt[9];
    euler_to_matrix(e, R);           /* :99 float64 rotation from the float32 euler */
    euler_to_matrix(target_ori, Rt); /* :100 */
    /* :101 E = Rt^T R - R^T Rt ; :102 rot_e = (E[2,1], E[0,2], E[1,0]) */
    double A[9]; /* A = Rt^T R */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = Rt[0 + i] * R[0 + j] + Rt[3 + i] * R[3 + j] + Rt[6 + i] * R[6 + j];
    double rot_e[3] = {A[7] - A[5], A[2] - A[6], A[3] - A[1]}; /* (R^T Rt) = A^T */
    double angvel_e[3], tt[3];
    for (int k = 0; k < 3; ++k) angvel_e[k] = 0.0 - (double)angvel[k]; /* :106 */
    for (int k = 0; k < 3; ++k) {
        s->integral_ori_e[k] = s->integral_ori_e[k] - rot_e[k] * p->ctrl_dt; /* :108 */
        s->integral_ori_e[k] = clipd(s->integral_ori_e[k], -1500., 1500.);   /* :109 */
    }
    s->integral_ori_e[0] = clipd(s->integral_ori_e[0], -1., 1.); /* :110 */
    s->integral_ori_e[1] = clipd(s->integral_ori_e[1], -1., 1.);
    for (int k = 0; k < 3; ++k) {
        tt[k] = -(ORI_P[k] * rot_e[k]) + ORI_I[k] * s->integral_ori_e[k] + ORI_D[k] * angvel_e[k]; /* :112-114 */
        tt[k] = clipd(tt[k], -3200., 3200.);                                                       /* :115 */
    }
    double nta = norm3(ta), thrust;
    if (nta != 0) { /* :117-122 */
        double cosang = (ta[0] / nta) * R[2] + (ta[1] / nta) * R[5] + (ta[2] / nta) * R[8];
        double ratio = 1 / (cosang > 0.2 ? cosang : 0.2);
        thrust = ratio * nta * p->mass;
    } else {
        thrust = 0.;
    }
    double tp = (sqrt(thrust / (4 * p->kf)) - PWM2RPM_B) / PWM2RPM_A; /* :123 */
    for (int i = 0; i < 4; ++i) {
        double pwm = tp + (MIXER[i][0] * tt[0] + MIXER[i][1] * tt[1] + MIXER[i][2] * tt[2]); /* :124 */
        pwm = clipd(pwm, MIN_PWM, MAX_PWM);                                                   /* :125 */
        rpm[i] = PWM2RPM_A * pwm + PWM2RPM_B;                                                 /* :126 */
    }
}

void orc_accel_control(const OrcParams *p, OrcPid *s, const double ta_in[3], const float ori[3],
                       const float angvel[3], double rpm[4])
{
    /* QuadControl.py:73-90 */
    double ta[3] = {ta_in[0] + 0., ta_in[1] + 0., ta_in[2] + p->ctrl_gravity}; /* :76 */
    double e[3] = {ori[0], ori[1], ori[2]}, Rd[9];
    euler_to_matrix(e, Rd);
    float R32[9];
    for (int k = 0; k < 9; ++k) R32[k] = (float)Rd[k]; /* :77 .float() */
    double n = norm3(ta);
    double tz[3] = {ta[0] / n, ta[1] / n, ta[2] / n}; /* :78 */
    if (isnan(tz[0]) || isnan(tz[1]) || isnan(tz[2])) { tz[0] = 0.; tz[1] = 0.; tz[2] = 1.; } /* :79-80 */
    double ycol[3] = {R32[1], R32[4], R32[7]}, tx[3], ty[3];
    cross3(ycol, tz, tx); /* :82 not normalised */
    cross3(tz, tx, ty);   /* :83 */
    double M[9] = {tx[0], ty[0], tz[0], tx[1], ty[1], tz[1], tx[2], ty[2], tz[2]}; /* :88 columns */
    double target_ori[3];
    orc_matrix_to_euler_nearest(M, target_ori); /* :89 */
    orc_attitude_control(p, s, target_ori, ori, angvel, ta, rpm); /* :90 */
}

void orc_vel_control(const OrcParams *p, OrcPid *s, const float vel[3], const float ori[3],
                     const float angvel[3], const float tv[3], double rpm[4])
{
    /* QuadControl.py:51-70; vel_e and the derivative numerator are float32 arithmetic */
    const float dt32 = (float)p->ctrl_dt;
    float vel_e[3];
    double ta[3];
    for (int k = 0; k < 3; ++k) vel_e[k] = tv[k] - vel[k]; /* :54 */
    if (isnan(s->last_vel_e[0])) {                          /* :55-57 */
        for (int k = 0; k < 3; ++k) { s->last_vel_e[k] = vel_e[k]; s->d_vel_e[k] = 0.; }
    }
    if (isnan(s->last_target_vel[0])) /* :58-59 */
        for (int k = 0; k < 3; ++k) s->last_target_vel[k] = tv[k];
    for (int k = 0; k < 3; ++k) {
        float a = vel_e[k] - s->last_vel_e[k];
        float b = tv[k] - s->last_target_vel[k];
        float c = a - b;
        float d = c / dt32;
        float h = d * 0.5f;
        s->d_vel_e[k] = (double)h + s->d_vel_e[k] * 0.5; /* :62 */
        s->last_vel_e[k] = vel_e[k];                     /* :64 */
        s->last_target_vel[k] = tv[k];                   /* :65 */
        float inc = vel_e[k] * dt32;
        s->integral_vel_e[k] = s->integral_vel_e[k] + (double)inc; /* :66 */
        ta[k] = VEL_P[k] * (double)vel_e[k] + VEL_I[k] * s->integral_vel_e[k] + VEL_D[k] * s->d_vel_e[k]; /* :67-69 */
    }
    orc_accel_control(p, s, ta, ori, angvel, rpm); /* :70 */
}

void orc_pos_control(const OrcParams *p, OrcPid *s, const float pos[3], const float vel[3], const float ori[3],
                     const float angvel[3], const float tp[3], double rpm[4])
{
    /* QuadControl.py:35-48 */
    const float dt32 = (float)p->ctrl_dt;
    double ta[3];
    for (int k = 0; k < 3; ++k) {
        float pos_e = tp[k] - pos[k];           /* :40 float32 */
        double d_pos_e = 0.0 - (double)vel[k]; /* :43 float64 zeros - float32 */
        float inc = pos_e * dt32;
        s->integral_pos_e[k] = s->integral_pos_e[k] + (double)inc; /* :44 */
        ta[k] = POS_P[k] * (double)pos_e + POS_I[k] * s->integral_pos_e[k] + POS_D[k] * d_pos_e; /* :45-47 */
    }
    orc_accel_control(p, s, ta, ori, angvel, rpm); /* :48 */
}

/* ------------------------------------------------------------- NNLS mixer */

static int solve_sym(int n, double *M /* n*n row-major, destroyed */, double *b /* in/out */)
{
    /* Gaussian elimination with partial pivoting (stands in for LAPACK sysv) */
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(M[c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(M[r * n + c]) > best) { best = fabs(M[r * n + c]); piv = r; }
        if (best == 0.0) return -1;
        if (piv != c) {
            for (int k = 0; k < n; ++k) { double t = M[c * n + k]; M[c * n + k] = M[piv * n + k]; M[piv * n + k] = t; }
            double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int r = c + 1; r < n; ++r) {
            double f = M[r * n + c] / M[c * n + c];
            for (int k = c; k < n; ++k) M[r * n + k] -= f * M[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double acc = b[r];
        for (int k = r + 1; k < n; ++k) acc -= M[r * n + k] * b[k];
        b[r] = acc / M[r * n + r];
    }
    return 0;
}

static void mixer_A(double A[16])
{
    /* Quadcopter.py:164 */
    const double c = 1 / sqrt(2.0);
    const double a[16] = {1, 1, 1, 1, c, c, -c, -c, -c, c, c, -c, -1, 1, -1, 1};
    memcpy(A, a, sizeof(a));
}

static int lawson_hanson4(const double A[16], const double b[4], int maxiter, double x[4])
{
    /* scipy.optimize.nnls (1.15, Lawson-Hanson on the normal equations) as called at
     * Quadcopter.py:205-207 with maxiter = 3*4 */
    double AtA[16], Atb[4], w[4], s[4];
    int P[4] = {0, 0, 0, 0};
    const double tol = 10 * 4 * 2.220446049250313e-16;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) acc += A[4 * k + i] * A[4 * k + j];
            AtA[4 * i + j] = acc;
        }
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += b[k] * A[4 * k + i];
        Atb[i] = acc; w[i] = acc; x[i] = 0; s[i] = 0;
    }
    int iter = 0;
    for (;;) {
        int allP = P[0] && P[1] && P[2] && P[3], any = 0;
        if (allP) break;
        for (int i = 0; i < 4; ++i) if (!P[i] && w[i] > tol) any = 1;
        if (!any) break;
        int k = 0; double best = -INFINITY;
        for (int i = 0; i < 4; ++i) { double v = P[i] ? 0.0 * w[i] : w[i]; if (v > best) { best = v; k = i; } }
        P[k] = 1;
        for (;;) {
            int idx[4], n = 0; double M[16], rhs[4];
            for (int i = 0; i < 4; ++i) if (P[i]) idx[n++] = i;
            for (int i = 0; i < n; ++i) { rhs[i] = Atb[idx[i]]; for (int j = 0; j < n; ++j) M[i * n + j] = AtA[4 * idx[i] + idx[j]]; }
            solve_sym(n, M, rhs);
            for (int i = 0; i < 4; ++i) s[i] = 0;
            for (int i = 0; i < n; ++i) s[idx[i]] = rhs[i];
            double smin = INFINITY;
            for (int i = 0; i < n; ++i) if (s[idx[i]] < smin) smin = s[idx[i]];
            if (!(iter < maxiter && smin < 0)) break;
            iter++;
            double alpha = INFINITY;
            for (int i = 0; i < 4; ++i)
                if (P[i] && s[i] < 0) { double a = x[i] / (x[i] - s[i]); if (a < alpha) alpha = a; }
            for (int i = 0; i < 4; ++i) { x[i] *= (1 - alpha); x[i] += alpha * s[i]; }
            for (int i = 0; i < 4; ++i) if (x[i] <= tol) P[i] = 0;
        }
        for (int i = 0; i < 4; ++i) x[i] = s[i];
        for (int i = 0; i < 4; ++i) {
            double acc = 0;
            for (int j = 0; j < 4; ++j) acc += AtA[4 * i + j] * x[j];
            w[i] = Atb[i] - acc;
        }
        if (iter == maxiter) return -1;
    }
    return iter;
}

int orc_nnls_rpm(const OrcParams *p, double thrust, double tx, double ty, double tz, double rpm[4])
{
    /* Quadcopter.nnlsRPM, Quadcopter.py:202-208 */
    double A[16], B[4], sq[4];
    mixer_A(A);
    const double bc[4] = {1 / p->kf, 1 / (p->kf * p->arm), 1 / (p->kf * p->arm), 1 / p->km}; /* :166 */
    B[0] = thrust * bc[0]; B[1] = tx * bc[1]; B[2] = ty * bc[2]; B[3] = tz * bc[3];
    /* Ainv = inv(A): rows of A are orthogonal with squared norms (4,2,2,4) => Ainv = A^T diag(1/4,1/2,1/2,1/4) */
    const double rn[4] = {0.25, 0.5, 0.5, 0.25};
    double mn = INFINITY;
    for (int i = 0; i < 4; ++i) {
        double acc = 0;
        for (int k = 0; k < 4; ++k) acc += (A[4 * k + i] * rn[k]) * B[k];
        sq[i] = acc;
        if (acc < mn) mn = acc;
    }
    int it = 0;
    if (mn < 0) { /* :204-207 */
        it = lawson_hanson4(A, B, 12, sq);
    }
    for (int i = 0; i < 4; ++i) rpm[i] = sqrt(sq[i]);
    return it;
}

int orc_set_control(const OrcParams *p, const float c[4], double rpm[4])
{
    /* Quadcopter.py:27-30: 0-d float32 tensors times python floats stay float32 */
    float thrust = c[0] * (float)p->mass;
    float roll = c[1] * (float)p->ixx_file;
    float pitch = c[2] * (float)p->iyy_file;
    float yaw = c[3] * (float)p->izz_file;
    return orc_nnls_rpm(p, (double)thrust, (double)roll, (double)pitch, (double)yaw, rpm);
}

/* -------------------------------------------------------------- adjacency */

void orc_adjacency(int n, const float *pos, double comm_range, float *A)
{
    /* MRS.calc_A, MRS.py:117-124.  float32 throughout: copos = pos_i - pos_j, codist = norm(dim=2),
     * diagonal = inf, A = (codist <= COMM_RANGE).float() */
    if (isinf(comm_range) && comm_range > 0) { /* :118-119 */
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) A[i * n + j] = (i == j) ? 0.f : 1.f;
        return;
    }
    const float cr = (float)comm_range;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) { A[i * n + j] = 0.f; continue; }
            float dx = pos[3 * i + 0] - pos[3 * j + 0];
            float dy = pos[3 * i + 1] - pos[3 * j + 1];
            float dz = pos[3 * i + 2] - pos[3 * j + 2];
            /* torch's float32 norm kernel contracts the sum of squares into FMAs on this host:
             * d2 = fma(dz,dz, fma(dy,dy, dx*dx)); pinned bit-exactly by tests/golden/F3 (4096
             * planted near-threshold pairs per range). */
            float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            float d = sqrtf(d2);
            A[i * n + j] = (d <= cr) ? 1.f : 0.f;
        }
}

void orc_adjacency_batch(int E, int n, const float *pos, double comm_range, float *A)
{
    for (int e = 0; e < E; ++e) orc_adjacency(n, pos + (size_t)e * n * 3, comm_range, A + (size_t)e * n * n);
}

/* ------------------------------------------------------------- Reynolds expert (caller of the path)
 * examples/simulating_data/helper/Reynolds.py:80-110 (forward_batch) with the controller of
 * Reynolds_Node.py:26-38, for the configuration its own caller uses (gen_data.py:33: D >= 6, K = 1).
 * forward_batch overwrites the adjacency it is given with ones - eye (Reynolds.py:83), takes the states of
 * the PREVIOUS step Xs[..., 1] (:87) and reduces to
 *     n_ij = x_j - x_i  (6 relative dims),
 *     out_i = 0.5 (sum_j n_p |n_p| - 3 sum_j A_ij n_p / (A_ij - 1 + |n_p|^3) + 3 sum_j n_v |n_v|),
 *     out_i *= min(|out_i|, 1) / |out_i|,   NaN -> 0  (:105).
 * float32 throughout, norms as torch evaluates them (fma chain, see orc_adjacency); the sums over j run in
 * index order here whereas torch's reduction order is its own: agreement to a few float32 ulp of the sum. */
static float norm3f(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }

void orc_reynolds(int E, int n, int D, const float *x_prev, float *actions)
{
    for (int e = 0; e < E; ++e) {
        const float *X = x_prev + (size_t)e * n * D;
        for (int i = 0; i < n; ++i) {
            float rp[3] = {0, 0, 0}, inv[3] = {0, 0, 0}, rv[3] = {0, 0, 0};
            for (int j = 0; j < n; ++j) {
                const float A = (i == j) ? 0.0f : 1.0f;
                float np_[3], nv[3];
                for (int k = 0; k < 3; ++k) {
                    /* Z[...,1] = A_ji Y_j - A_ij X_i, Z[...,0] = 0 in the relative dims (Reynolds.py:99-103) */
                    np_[k] = A * X[j * D + k] - A * X[i * D + k];
                    nv[k] = A * X[j * D + 3 + k] - A * X[i * D + 3 + k];
                }
                const float dp = norm3f(np_[0], np_[1], np_[2]), dv = norm3f(nv[0], nv[1], nv[2]);
                const float den = (A - 1.0f) + dp * dp * dp;
                for (int k = 0; k < 3; ++k) {
                    rp[k] += np_[k] * dp;
                    inv[k] += A * (np_[k] / den);
                    rv[k] += nv[k] * dv;
                }
            }
            float o[3];
            for (int k = 0; k < 3; ++k) o[k] = 0.5f * ((1.0f * rp[k] + 3.0f * (-inv[k])) + 3.0f * rv[k]);
            const float no = norm3f(o[0], o[1], o[2]);
            const float mag = (no > 1.0f ? 1.0f : no) / no;     /* torch.clamp(norm, max=1) / norm */
            for (int k = 0; k < 3; ++k) {
                float v = o[k] * mag;
                actions[((size_t)e * n + i) * 3 + k] = (v != v) ? 0.0f : v;
            }
        }
    }
}

/* ------------------------------------------------------------- integration */

static void quat_to_matrix_bullet(const double q[4], double R[9])
{
    /* btMatrix3x3::setRotation */
    double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    double s = 2.0 / d;
    double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
    double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
    double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
    R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}

static void matvec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
static void matTvec(const double R[9], const double v[3], double o[3])
{
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}

/* Ground contact: the build's own model (Bullet's GJK/EPA persistent manifold + PGS is not
 * reproducible).  4 body-fixed points (azimuths 45+90k deg) on the rim of whichever
 * cap of the collision cylinder faces the ground,
 * Bullet-style velocity-level rhs (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint),
 * sequential impulses with a friction pyramid along world x/y. */
/* Diagnostic tap (tools/contact_lab.py, single-threaded runs only): the inputs of every contact problem, 17 doubles each
 * (z, R[9], v[3], w[3], the body's address as an identity across steps), for offline experiments with the solver. */
static double *g_cdump = 0;
static long g_cdump_cap = 0, g_cdump_n = 0;
void orc_contact_dump(double *buf, long cap) { g_cdump = buf; g_cdump_cap = cap; g_cdump_n = 0; }
long orc_contact_dump_count(void) { return g_cdump_n; }

/* A body lying FLAT on the ground (|R20|, |R21| < flat_eps: all four rim points of the cap that faces the ground within the
 * threshold, with one common gap) whose twelve rows have a closed-form fixed point is given that fixed point instead of sweeps
 * that stop short of it (OrcParams.rest_shortcut; the kernel's contact_at_rest, mrs-gym_amd/csrc/mrs_device.hpp, is this function):
 *   lifting  -- no rim point's right-hand side (u - v_z) - (w x r_k)_z is positive (bounded by (u - v_z) + (|w_x| + |w_y|) r): no normal
 *               impulse, hence no friction: v, w unchanged (what the sweeps return, exactly);
 *   sticking -- the contact can hold the body, v = (0, 0, u), w = 0, if impulses at the four points exist with the required sum
 *               P = m (v_post - v) and moment L = -I w (I = diag(I0, I0, I2): flat, I0 = I1), normals >= 0, tangential parts inside the
 *               pyramid; tested conservatively: s = 2 r (P_z / 4 - f_max / mu) > 0, |M|^2 < s^2 with M = (L_x - hl P_y, L_y + hl P_x),
 *               f_max = max(|P_x|, |P_y|) / 4 + |L_z| / (4 r).
 * u = the rim points' common target normal velocity, -gap / dt (open) or -erp gap / dt (penetrating).  tests/test_oracle_physics.py
 * checks the closed forms against 400 sweeps of the rows themselves (<= 3e-6 m/s).  Returns 1 when the body is dealt with. */
static int contact_flat_closed_form(const OrcParams *p, const double pos[3], const double R[9], double v[3], double w[3])
{
    const double eps = 1e-6, flat_eps = 1e-6;
    const double dist = (pos[2] - p->ground_z) - p->coll_half_len * fabs(R[8]);
    if (!(fmax(fabs(R[6]), fabs(R[7])) < flat_eps) || !(fabs(dist - p->contact_threshold) > eps)) return 0;
    if (dist > p->contact_threshold) return 1;
    const double u = -dist * (dist > 0 ? 1.0 / p->dt : p->erp * (1.0 / p->dt));
    const double up = u - v[2];
    if (up + (fabs(w[0]) + fabs(w[1])) * p->coll_radius <= 0.0) return 1;
    if (!(p->inertia[0] == p->inertia[1])) return 0;
    const double Pz = p->mass * up, Px = -p->mass * v[0], Py = -p->mass * v[1];
    const double Mx = -p->inertia[0] * w[0] - p->coll_half_len * Py, My = -p->inertia[0] * w[1] + p->coll_half_len * Px;
    const double fm = 0.25 * fmax(fabs(Px), fabs(Py)) + fabs(p->inertia[2] * w[2]) * (0.25 / p->coll_radius);
    const double s = 2.0 * p->coll_radius * (0.25 * Pz - fm / p->friction);
    if (!(s > 8.0 * eps * p->mass) || !((Mx * Mx + My * My) * 1.02 < s * s)) return 0;
    v[0] = v[1] = 0.0; v[2] = u;
    w[0] = w[1] = w[2] = 0.0;
    return 1;
}

static void contact_solve_ex(const OrcParams *p, const double pos[3], const double R[9], double v[3], double w[3], int n_sweeps, int early_exit,
                             int closed_forms);
/* diagnostic hook (tools/teacher_replay.py): n > 0 = every solve runs exactly n sweeps, no early exit; 0 = the model's rule */
static int g_force_sweeps = 0;
void orc_debug_force_sweeps(int n) { g_force_sweeps = n; }
static void contact_solve(const OrcParams *p, const double pos[3], const double R[9], double v[3], double w[3])
{
    if (g_force_sweeps > 0) contact_solve_ex(p, pos, R, v, w, g_force_sweeps, 0, p->rest_shortcut);
    else contact_solve_ex(p, pos, R, v, w, p->solver_iters, 1, p->rest_shortcut);
}
/* test hook: the twelve rows of one body swept n_sweeps times with neither the closed forms nor the early exits -- what the
 * closed forms and the capped sweeps are measured against (tests/test_oracle_physics.py) */
void orc_contact_rows(const OrcParams *p, const double pos[3], const double quat[4], double v[3], double w[3], int n_sweeps)
{
    double R[9];
    quat_to_matrix_bullet(quat, R);
    contact_solve_ex(p, pos, R, v, w, n_sweeps, 0, 0);
}
/* test hook: the model's contact solve of one body on its own (closed forms by p->rest_shortcut, the model's stopping rules):
 * what tests/test_device_math_host.py holds the kernel's contact function against */
void orc_contact_solve(const OrcParams *p, const double pos[3], const double quat[4], double v[3], double w[3])
{
    double R[9];
    quat_to_matrix_bullet(quat, R);
    contact_solve_ex(p, pos, R, v, w, p->solver_iters, 1, p->rest_shortcut);
}
static void contact_solve_ex(const OrcParams *p, const double pos[3], const double R[9], double v[3], double w[3], int n_sweeps, int early_exit,
                             int closed_forms)
{
    const double bound = sqrt(p->coll_radius * p->coll_radius + p->coll_half_len * p->coll_half_len);
    if (pos[2] - bound - p->contact_threshold > p->ground_z) return;
    if (closed_forms && contact_flat_closed_form(p, pos, R, v, w)) return;
    if (g_cdump && g_cdump_n < g_cdump_cap) {
        double *d = g_cdump + 17 * g_cdump_n++;
        d[0] = pos[2]; d[16] = (double)(size_t)pos;
        for (int k = 0; k < 9; ++k) d[1 + k] = R[k];
        for (int k = 0; k < 3; ++k) { d[10 + k] = v[k]; d[13 + k] = w[k]; }
    }
    const double c = p->coll_radius * 0.70710678118654752440;
    double r[8][3], dist[8], lam_n[8], lam_t[8][2], kn[8], kt[8][2], rhs[8];
    int active[8], nact = 0;
    double Iinv_b[3] = {1.0 / p->inertia[0], 1.0 / p->inertia[1], 1.0 / p->inertia[2]};
    for (int k = 0; k < 8; ++k) {
        double pb[3] = {(k & 1) ? -c : c, (k & 2) ? -c : c, (k & 4) ? -p->coll_half_len : p->coll_half_len};
        matvec(R, pb, r[k]);
        dist[k] = pos[2] + r[k][2] - p->ground_z;
        /* only the rim of the cap that faces the ground carries contacts: at every azimuth the other cap's
         * point is 2 h |R22| higher, it can only come within the threshold once the body is tilted by more
         * than ~37 degrees, where the lower rim already supports it */
        const int lower_cap = (R[8] >= 0) ? ((k & 4) != 0) : ((k & 4) == 0);
        active[k] = lower_cap && dist[k] <= p->contact_threshold;
        lam_n[k] = 0; lam_t[k][0] = 0; lam_t[k][1] = 0;
        if (!active[k]) continue;
        nact++;
        /* effective masses for directions z, x, y:  1/m + d.((Iw^-1 (r x d)) x r) */
        for (int a = 0; a < 3; ++a) {
            double d[3] = {a == 1, a == 2, a == 0}; /* a=0: z, a=1: x, a=2: y */
            double rxd[3], tb[3], tw[3], cr[3];
            cross3(r[k], d, rxd);
            matTvec(R, rxd, tb);
            tb[0] *= Iinv_b[0]; tb[1] *= Iinv_b[1]; tb[2] *= Iinv_b[2];
            matvec(R, tb, tw);
            cross3(tw, r[k], cr);
            double K = 1.0 / p->mass + (d[0] * cr[0] + d[1] * cr[1] + d[2] * cr[2]);
            if (a == 0) kn[k] = 1.0 / K; else kt[k][a - 1] = 1.0 / K;
        }
        double vrel_n = v[2] + (w[0] * r[k][1] - w[1] * r[k][0]);
        double pen = dist[k], poserr = 0, velerr = -vrel_n;
        if (pen > 0) velerr -= pen / p->dt; else poserr = -pen * p->erp / p->dt;
        rhs[k] = poserr + velerr; /* target normal velocity change */
    }
    if (!nact) return;
    double v0[3] = {v[0], v[1], v[2]}, w0[3] = {w[0], w[1], w[2]};
    /* Start of the sweeps: every active point carries the equal share of the impulse that stops the mean closing
     * velocity, l0 = m * max(sum rhs, 0) / n^2 -- exact for a body lying flat, which then needs no iteration at all;
     * the sweeps correct it for everything else.  (Round 2: from a cold start a resting body took 8-10 sweeps.) */
    {
        double rsum = 0;
        for (int k = 0; k < 8; ++k) if (active[k]) rsum += rhs[k];
        const double l0 = p->mass * (rsum > 0 ? rsum : 0) / ((double)nact * (double)nact);
        for (int k = 0; k < 8; ++k) {
            if (!active[k]) continue;
            lam_n[k] = l0;
            double imp[3] = {0, 0, l0}, rxi[3], tb[3], tw[3];
            v[2] += l0 / p->mass;
            cross3(r[k], imp, rxi); matTvec(R, rxi, tb);
            tb[0] *= Iinv_b[0]; tb[1] *= Iinv_b[1]; tb[2] *= Iinv_b[2];
            matvec(R, tb, tw);
            w[0] += tw[0]; w[1] += tw[1]; w[2] += tw[2];
        }
    }
    /* The sweeps stop once a pair of sweeps has moved no impulse by more than 1e-7 of the resting impulse m g dt, or of
     * the body's largest normal impulse, or has stopped making progress (looked at after every second sweep, at most
     * solver_iters sweeps). */
    const double tol = 1e-7 * (p->mass * p->gravity * p->dt) + 1e-30;
    double moved = 0, prev_moved = 3.0e38;
    for (int it = 0; it < n_sweeps; ++it) {
        if ((it & 1) == 0) moved = 0;
        const int track = (it & 1);
        for (int k = 0; k < 8; ++k) {
            if (!active[k]) continue;
            /* normal */
            {
                double dvn = (v[2] - v0[2]) + ((w[0] - w0[0]) * r[k][1] - (w[1] - w0[1]) * r[k][0]);
                double dl = kn[k] * (rhs[k] - dvn);
                double nl = lam_n[k] + dl;
                if (nl < 0) nl = 0;
                dl = nl - lam_n[k]; lam_n[k] = nl;
                if (track && fabs(dl) > moved) moved = fabs(dl);
                double imp[3] = {0, 0, dl}, rxi[3], tb[3], tw[3];
                v[2] += dl / p->mass;
                cross3(r[k], imp, rxi); matTvec(R, rxi, tb);
                tb[0] *= Iinv_b[0]; tb[1] *= Iinv_b[1]; tb[2] *= Iinv_b[2];
                matvec(R, tb, tw);
                w[0] += tw[0]; w[1] += tw[1]; w[2] += tw[2];
            }
            /* friction along world x, y: drive tangential point velocity to zero */
            for (int a = 0; a < 2; ++a) {
                double d[3] = {a == 0, a == 1, 0};
                double wxr[3];
                cross3(w, r[k], wxr);
                double vt = d[0] * (v[0] + wxr[0]) + d[1] * (v[1] + wxr[1]);
                double dl = -kt[k][a] * vt;
                double lim = p->friction * lam_n[k];
                double nl = clipd(lam_t[k][a] + dl, -lim, lim);
                dl = nl - lam_t[k][a]; lam_t[k][a] = nl;
                if (track && fabs(dl) > moved) moved = fabs(dl);
                double imp[3] = {d[0] * dl, d[1] * dl, 0}, rxi[3], tb[3], tw[3];
                v[0] += imp[0] / p->mass; v[1] += imp[1] / p->mass;
                cross3(r[k], imp, rxi); matTvec(R, rxi, tb);
                tb[0] *= Iinv_b[0]; tb[1] *= Iinv_b[1]; tb[2] *= Iinv_b[2];
                matvec(R, tb, tw);
                w[0] += tw[0]; w[1] += tw[1]; w[2] += tw[2];
            }
        }
        if (track && early_exit) {
            double lmax = 0;
            for (int k = 0; k < 8; ++k) if (active[k] && lam_n[k] > lmax) lmax = lam_n[k];
            const double t2 = 1e-7 * lmax;
            if (moved <= (tol > t2 ? tol : t2)) break;
            /* ... or once a pair of sweeps has moved the impulses by at least half of what the pair before it did (no
             * progress: the clamps of the friction pyramid chatter for a few per cent of the grounded bodies) */
            if (moved >= 0.5 * prev_moved) break;
            prev_moved = moved;
        }
    }
}

/* first half of stepSimulation for one body: forces -> unconstrained velocities */
static void integrate_velocities(const OrcParams *p, const double quat[4], double vel[3], double angvel[3],
                                 const double fb_ext[3], const double tb_ext[3])
{
    /* [BULLET-KNOWLEDGE] btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof for a
     * floating base whose links are all fixed and massless, then applyDeltaVeeMultiDof, contact
     * solve, stepPositionsMultiDof.  Called from BulletSim.step_sim (BulletSim.py:46-47). */
    double R[9], vb[3], wb[3], gb[3];
    const double dt = p->dt, k_l = p->lin_damp, k_a = p->ang_damp;
    quat_to_matrix_bullet(quat, R);
    matTvec(R, vel, vb);
    matTvec(R, angvel, wb);
    const double gw[3] = {0, 0, -p->gravity * p->mass};
    matTvec(R, gw, gb);
    double fb[3] = {fb_ext[0] + gb[0], fb_ext[1] + gb[1], fb_ext[2] + gb[2]};
    double nv = norm3(vb), nw = norm3(wb);
    double Iw[3] = {p->inertia[0] * wb[0], p->inertia[1] * wb[1], p->inertia[2] * wb[2]};
    double gyro[3] = {0, 0, 0}, cor[3], ab[3], alb[3], vdot[3], wdot[3], tmp[3];
    if (p->use_gyro) cross3(wb, Iw, gyro);
    cross3(wb, vb, cor);
    for (int k = 0; k < 3; ++k) {
        /* zeroAccSpatFrc = -(F) + damping + coriolis ; acc = -zeroAcc / inertia */
        double zl = -fb[k] + p->mass * vb[k] * (k_l + k_l * nv) + p->mass * cor[k];
        double za = -tb_ext[k] + Iw[k] * (k_a + k_a * nw) + gyro[k];
        ab[k] = -zl / p->mass;
        alb[k] = -za / p->inertia[k];
    }
    for (int k = 0; k < 3; ++k) tmp[k] = ab[k] + cor[k];
    matvec(R, tmp, vdot);
    matvec(R, alb, wdot);
    for (int k = 0; k < 3; ++k) {
        angvel[k] = clipd(angvel[k] + wdot[k] * dt, -p->max_coord_vel, p->max_coord_vel);
        vel[k] = clipd(vel[k] + vdot[k] * dt, -p->max_coord_vel, p->max_coord_vel);
    }
}

/* second half: ground contact on the unconstrained velocities, then stepPositionsMultiDof */
static void contact_and_pose(const OrcParams *p, double pos[3], double quat[4], double vel[3], double angvel[3])
{
    const double dt = p->dt;
    double R[9];
    quat_to_matrix_bullet(quat, R);
    if (p->enable_contact) contact_solve(p, pos, R, vel, angvel);
    /* stepPositionsMultiDof */
    for (int k = 0; k < 3; ++k) pos[k] += dt * vel[k];
    double fAngle = norm3(angvel), axis[3];
    if (fAngle * dt > 0.25 * PI_D) fAngle = 0.5 * (0.5 * PI_D) / dt; /* ANGULAR_MOTION_THRESHOLD */
    double sc;
    if (fAngle < 0.001)
        sc = 0.5 * dt - (dt * dt * dt) * 0.020833333333 * fAngle * fAngle;
    else
        sc = sin(0.5 * fAngle * dt) / fAngle;
    for (int k = 0; k < 3; ++k) axis[k] = angvel[k] * sc;
    double dw = cos(fAngle * dt * 0.5);
    /* world orientation q <- dq * q */
    double qx = quat[0], qy = quat[1], qz = quat[2], qw = quat[3];
    double nx = dw * qx + axis[0] * qw + axis[1] * qz - axis[2] * qy;
    double ny = dw * qy + axis[1] * qw + axis[2] * qx - axis[0] * qz;
    double nz = dw * qz + axis[2] * qw + axis[0] * qy - axis[1] * qx;
    double nw2 = dw * qw - axis[0] * qx - axis[1] * qy - axis[2] * qz;
    double nn = sqrt(nx * nx + ny * ny + nz * nz + nw2 * nw2);
    quat[0] = nx / nn; quat[1] = ny / nn; quat[2] = nz / nn; quat[3] = nw2 / nn;
}

void orc_integrate(const OrcParams *p, double pos[3], double quat[4], double vel[3], double angvel[3],
                   const double fb_ext[3], const double tb_ext[3])
{
    /* one body on its own (no other body to touch): what the fixture generator's stand-in for p.stepSimulation calls */
    integrate_velocities(p, quat, vel, angvel, fb_ext, tb_ext);
    contact_and_pose(p, pos, quat, vel, angvel);
}

/* Quad-quad contact, the build's own model (SURVEY.md row G says Bullet has hull-hull contacts; its GJK manifold and
 * solver are not reproducible here): every quadcopter is a sphere of its collision radius; a pair whose spheres are
 * within the contact threshold gets, per body, half of the normal velocity change that closes the gap this step /
 * pushes the overlap out with erp -- the same velocity-level right-hand side as the ground rows, one pass, no
 * friction, no torque.  Geometry and relative velocity in float32 on the float32 read-back positions (what the
 * kernel's LDS tile holds); applied to the unconstrained velocities, before the ground rows.  opos: pre-step positions. */
static void pair_contact(const OrcParams *p, int N, float (*opos)[3], double *vel)
{
    const float r2 = 2.0f * (float)p->coll_radius, thr = (float)p->contact_threshold;
    const float rc2 = (r2 + thr) * (r2 + thr), inv_dt = (float)(1.0 / p->dt), erp_dt = (float)(p->erp / p->dt);
    float dv[N][3];
    for (int i = 0; i < N; ++i) {
        dv[i][0] = dv[i][1] = dv[i][2] = 0.f;
        const float vix = (float)vel[3 * i], viy = (float)vel[3 * i + 1], viz = (float)vel[3 * i + 2];
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            const float rx = opos[i][0] - opos[j][0], ry = opos[i][1] - opos[j][1], rz = opos[i][2] - opos[j][2];
            const float d2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
            if (!(d2 <= rc2) || !(d2 > 0.f)) continue;
            const float d = sqrtf(d2), rd = 1.0f / d;
            const float nx = rx * rd, ny = ry * rd, nz = rz * rd;
            const float ux = vix - (float)vel[3 * j], uy = viy - (float)vel[3 * j + 1], uz = viz - (float)vel[3 * j + 2];
            const float vn = fmaf(uz, nz, fmaf(uy, ny, ux * nx));
            const float gap = d - r2;
            const float rhs = -vn - gap * (gap > 0.f ? inv_dt : erp_dt);
            if (rhs > 0.f) {
                const float h = 0.5f * rhs;
                dv[i][0] = fmaf(h, nx, dv[i][0]); dv[i][1] = fmaf(h, ny, dv[i][1]); dv[i][2] = fmaf(h, nz, dv[i][2]);
            }
        }
    }
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < 3; ++k) vel[3 * i + k] += (double)dv[i][k];
}


/* ------------------------------------------------------------------- spawn */

/* MRS.generate_start_pos (MRS.py:127-154) as a function of the samples its START_POS hands out: cand[r][i] = what agent i
 * receives if it is re-sampled in round r (round 0 = the first layout; per-agent draws :146-148 and joint draws :149-151
 * look the same in this layout).  Pinned by tests/golden/F5b (the reference driven by a replay distribution).
 * Returns the number of candidate rounds consumed (>= 1), or -1 if the layout still collides after n_rounds. */
int orc_spawn_from(int N, int n_rounds, const float *cand, double agent_radius, float *pos)
{
    const float min_dist = (float)(2 * agent_radius); /* codist (float32) < 2*AGENT_RADIUS: the scalar joins in float32 */
    unsigned char *col = (unsigned char *)malloc((size_t)N * N);
    int *flag = (int *)malloc(sizeof(int) * (size_t)N);
    memcpy(pos, cand, sizeof(float) * 3 * (size_t)N);
    int used = 1;
    for (;;) {
        long total = 0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                /* get_relative_position(...).norm(dim=2), :135: torch's float32 norm = sqrt(fma(dz,dz,fma(dy,dy,dx*dx))) */
                const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
                const float d = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
                col[(size_t)i * N + j] = (i != j) && (d < min_dist); /* diagonal = inf, :136 */
                total += col[(size_t)i * N + j];
            }
        if (!total) break;                                   /* :137 */
        if (used >= n_rounds) { used = -1; break; }
        for (int i = 0; i < N; ++i) flag[i] = 0;
        while (total) {                                      /* :140-144 */
            /* torch.mode(torch.where(collisions)[0]): the row index that occurs most often, the smallest among equals */
            int best = -1, bestc = 0;
            for (int i = 0; i < N; ++i) {
                int c = 0;
                for (int j = 0; j < N; ++j) c += col[(size_t)i * N + j];
                if (c > bestc) { bestc = c; best = i; }
            }
            flag[best] = 1;
            for (int j = 0; j < N; ++j) {
                total -= col[(size_t)best * N + j] + col[(size_t)j * N + best];
                col[(size_t)best * N + j] = 0; col[(size_t)j * N + best] = 0;
            }
        }
        for (int i = 0; i < N; ++i)                          /* :145-151 */
            if (flag[i]) memcpy(pos + 3 * i, cand + ((size_t)used * N + i) * 3, sizeof(float) * 3);
        used++;
    }
    free(col); free(flag);
    return used;
}

/* -------------------------------------------------------------------- step */

static void step_env(const OrcParams *p, int N, double *pos, double *quat, double *vel, double *angvel,
                     OrcPid *pid, const float *actions, int action_type, int adim, double *speeds_out,
                     double *wrench_out, double hclip)
{
    /* scratch on the stack in chunks is awkward for large N: use VLAs bounded by N */
    float opos[N][3], oeul[N][3], ovel[N][3], oang[N][3], omat[N][9];
    double fb[N][3], tb[N][3];
    for (int i = 0; i < N; ++i) {
        orc_observe(&pos[3 * i], &quat[4 * i], &vel[3 * i], &angvel[3 * i], opos[i], oeul[i], ovel[i], oang[i], omat[i]);
        fb[i][0] = fb[i][1] = fb[i][2] = 0; tb[i][0] = tb[i][1] = tb[i][2] = 0;
    }
    if (action_type != ORC_ACT_NONE && actions) {
        for (int i = 0; i < N; ++i) { /* Environment.set_actions loop 1, Environment.py:91-92 */
            const float *a = &actions[(size_t)i * adim];
            double rpm[4], F[4], zt;
            int f32_speeds = 0;
            float s32[4];
            switch (action_type) {
            case ORC_ACT_SET_SPEEDS: f32_speeds = 1; for (int k = 0; k < 4; ++k) { s32[k] = a[k]; rpm[k] = a[k]; } break;
            case ORC_ACT_SET_CONTROL: orc_set_control(p, a, rpm); break;
            case ORC_ACT_TARGET_ACCEL: { double ta[3] = {a[0], a[1], a[2]}; orc_accel_control(p, &pid[i], ta, oeul[i], oang[i], rpm); } break;
            case ORC_ACT_TARGET_VEL: orc_vel_control(p, &pid[i], ovel[i], oeul[i], oang[i], a, rpm); break;
            case ORC_ACT_TARGET_POS: orc_pos_control(p, &pid[i], opos[i], ovel[i], oeul[i], oang[i], a, rpm); break;
            case ORC_ACT_TARGET_ORI: { double to[3] = {a[0], a[1], a[2]}; double ta[3] = {0., 0., 9.81}; /* Quadcopter.py:64 */
                                       orc_attitude_control(p, &pid[i], to, oeul[i], oang[i], ta, rpm); } break;
            default: rpm[0] = rpm[1] = rpm[2] = rpm[3] = 0; break;
            }
            /* set_speeds, Quadcopter.py:38-45 */
            if (f32_speeds) {
                float t[4];
                for (int k = 0; k < 4; ++k) { float sq = s32[k] * s32[k]; F[k] = (double)(sq * (float)p->kf); t[k] = sq * (float)p->km; }
                float z = -t[0] + t[1]; z = z - t[2]; z = z + t[3];
                zt = (double)z;
            } else {
                double t[4];
                for (int k = 0; k < 4; ++k) { double sq = rpm[k] * rpm[k]; F[k] = sq * p->kf; t[k] = sq * p->km; }
                zt = ((-t[0] + t[1]) - t[2]) + t[3];
            }
            if (speeds_out) for (int k = 0; k < 4; ++k) speeds_out[4 * i + k] = rpm[k];
            /* dynamics(): loop 2, Environment.py:93-94 -> Quadcopter.py:69-115.  All reads are of the
             * pre-step state, so the two loops can be fused per agent. */
            double Rb[9];
            quat_to_matrix_bullet(&quat[4 * i], Rb);
            /* ground effect :70-87 */
            double G[4] = {0, 0, 0, 0};
            {
                int cond = (oeul[i][0] < (float)(PI_D / 2)) && (oeul[i][1] < (float)(PI_D / 2)); /* :80 (np.abs of a bool) */
                for (int k = 0; k < 4; ++k) {
                    double h = pos[3 * i + 2] + (Rb[6] * p->prop_x[k] + Rb[7] * p->prop_y[k] + Rb[8] * p->prop_z[k]);
                    if (h < hclip) h = hclip; /* :77 */
                    double ratio = p->prop_radius / (4 * h);
                    double g;
                    if (f32_speeds) {
                        float sq = s32[k] * s32[k];
                        float t1 = sq * (float)p->kf;
                        float t2 = t1 * (float)p->gnd_eff_coeff;
                        g = (double)t2 * (ratio * ratio);
                    } else {
                        g = ((rpm[k] * rpm[k]) * p->kf) * p->gnd_eff_coeff * (ratio * ratio);
                    }
                    G[k] = cond ? g : 0.0;
                }
            }
            /* props: force along body z at the prop link COM */
            for (int k = 0; k < 4; ++k) {
                double f = F[k] + G[k];
                fb[i][2] += f;
                tb[i][0] += p->prop_y[k] * f;
                tb[i][1] += -p->prop_x[k] * f;
            }
            tb[i][2] += zt;
            /* drag :88-98 (applied in LINK_FRAME: the already-rotated vector is a body-frame force) */
            {
                double sum;
                if (f32_speeds) {
                    float acc = 0.f;
                    for (int k = 0; k < 4; ++k) { float t = (float)(2 * PI_D) * s32[k]; t = t / 60.f; acc = acc + t; }
                    sum = (double)acc;
                } else {
                    sum = 0;
                    for (int k = 0; k < 4; ++k) sum += 2 * PI_D * rpm[k] / 60;
                }
                double dfac[3] = {-1 * p->drag_xy * sum, -1 * p->drag_xy * sum, -1 * p->drag_z * sum};
                double t[3] = {dfac[0] * (double)ovel[i][0], dfac[1] * (double)ovel[i][1], dfac[2] * (double)ovel[i][2]};
                for (int r = 0; r < 3; ++r)
                    fb[i][r] += (double)omat[i][3 * r] * t[0] + (double)omat[i][3 * r + 1] * t[1] + (double)omat[i][3 * r + 2] * t[2];
            }
            /* downwash :99-115, float32 arithmetic on float32 positions */
            {
                double acc = 0;
                for (int j = 0; j < N; ++j) {
                    float rx = opos[j][0] - opos[i][0], ry = opos[j][1] - opos[i][1], dz = opos[j][2] - opos[i][2];
                    float dxy = sqrtf(rx * rx + ry * ry);
                    if (dz > 0 && dxy < 10) {
                        float rc = 1.0f / (4.0f * dz);          /* torch __rtruediv__: reciprocal() * scalar */
                        float q = rc * (float)p->prop_radius;
                        float alpha = (float)p->dw1 * (q * q);
                        float beta = (float)p->dw2 * dz + (float)p->dw3;
                        float t = (1.0f / beta) * dxy;          /* np.float32 / tensor defers to __rtruediv__ */
                        float ex = expf(-.5f * (t * t));
                        acc += (double)(-alpha * ex);
                    }
                }
                fb[i][2] += acc;
            }
        }
    }
    if (wrench_out)
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) { wrench_out[6 * i + k] = fb[i][k]; wrench_out[6 * i + 3 + k] = tb[i][k]; }
    /* BulletSim.step_sim: unconstrained velocities of every body, contacts between bodies, ground contact + pose */
    for (int i = 0; i < N; ++i) integrate_velocities(p, &quat[4 * i], &vel[3 * i], &angvel[3 * i], fb[i], tb[i]);
    if (p->enable_contact && p->pair_contact && N > 1) pair_contact(p, N, opos, vel);
    for (int i = 0; i < N; ++i) contact_and_pose(p, &pos[3 * i], &quat[4 * i], &vel[3 * i], &angvel[3 * i]);
}

void orc_step(const OrcParams *p, int E, int N, double *pos, double *quat, double *vel, double *angvel,
              OrcPid *pid, const float *actions, int action_type, int adim, double *speeds_out, double *wrench_out,
              int nthreads)
{
    double d[7];
    orc_derived(p, d);
    const double hclip = d[6];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        size_t o = (size_t)e * N;
        step_env(p, N, pos + 3 * o, quat + 4 * o, vel + 3 * o, angvel + 3 * o, pid + o,
                 actions ? actions + o * adim : 0, action_type, adim, speeds_out ? speeds_out + 4 * o : 0,
                 wrench_out ? wrench_out + 6 * o : 0, hclip);
    }
    (void)nthreads;
}

/* The whole per-step hot path of MRS.step for the CPU baseline: step + newest observation slice
 * (state_fn = cat(pos, vel), README.md:28-29) + newest adjacency (MRS.calc_A), env-parallel. */
void orc_step_full(const OrcParams *p, int E, int N, double *pos, double *quat, double *vel, double *angvel,
                   OrcPid *pid, const float *actions, int action_type, int adim, double comm_range,
                   float *obs /* [E][N][6] */, float *A /* [E][N][N] or NULL */, int nthreads)
{
    double d[7];
    orc_derived(p, d);
    const double hclip = d[6];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int e = 0; e < E; ++e) {
        size_t o = (size_t)e * N;
        step_env(p, N, pos + 3 * o, quat + 4 * o, vel + 3 * o, angvel + 3 * o, pid + o,
                 actions ? actions + o * adim : 0, action_type, adim, 0, 0, hclip);
        float p32[N][3];
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < 3; ++k) {
                p32[i][k] = (float)pos[3 * (o + i) + k];
                obs[6 * (o + i) + k] = p32[i][k];
                obs[6 * (o + i) + 3 + k] = (float)vel[3 * (o + i) + k];
            }
        }
        if (A) orc_adjacency(N, &p32[0][0], comm_range, A + o * N);
    }
}
