// Diagnostic: the kernel's own float32 contact solve (mrs_device.hpp contact_solve_f32), compiled for the CPU.
//   extern "C" host_contact(params, pz, quat, v, w, dv_out, dw_out)
#include "hip/hip_runtime.h"
#include <cstring>
#include "../../mrs-gym_amd/csrc/mrs_device.hpp"
using namespace mrs;
// the same statements with the scalar type as a parameter: mrs_device.hpp contact_solve_rows<S> (a template since round 5; rounds 3-4
// derived the float64 text by regular expressions).  dv[2] carries the start's float64 share as well.
template <class S>
static void rows(const MrsParams *P, double pz, const double *q, const double *v, const double *w, double *dv, double *dw)
{
    Recips K{};
    K.inv_mass = 1.0 / P->mass; K.inv_i0 = 1.0 / P->inertia[0]; K.inv_i1 = 1.0 / P->inertia[1]; K.inv_i2 = 1.0 / P->inertia[2]; K.inv_dt = 1.0 / P->dt;
    double vv[3] = {v[0], v[1], v[2]}, ww[3] = {w[0], w[1], w[2]};
    contact_solve_rows<S>(*P, K, pz, q, ContactBodyRegs{vv, ww});
    for (int i = 0; i < 3; ++i) { dv[i] = vv[i] - v[i]; dw[i] = ww[i] - w[i]; }
}
extern "C" void host_contact_f64(const MrsParams *P, double pz, const double *q, const double *v, const double *w, double *dv, double *dw) { rows<double>(P, pz, q, v, w, dv, dw); }
extern "C" void host_contact_f32t(const MrsParams *P, double pz, const double *q, const double *v, const double *w, double *dv, double *dw) { rows<float>(P, pz, q, v, w, dv, dw); }
extern "C" void host_contact(const MrsParams *P, double pz, const double *q, const double *v, const double *w, float *dv, float *dw)
{
    double a[3], b[3];
    rows<float>(P, pz, q, v, w, a, b);
    for (int i = 0; i < 3; ++i) { dv[i] = (float)a[i]; dw[i] = (float)b[i]; }
}

// One body through the step kernel's rigid-body pipeline (mrs_kernels.hip k_step: integrate_velocity -> contact_at_rest |
// contact_stage -> integrate_pose), the device functions themselves: tests/test_device_math_host.py holds it against the oracle.
extern "C" void host_body_step(const MrsParams *P, double *p, double *q, double *v, double *w, const double *fb, const double *tb)
{
    Recips K{};
    K.inv_mass = 1.0 / P->mass; K.inv_i0 = 1.0 / P->inertia[0]; K.inv_i1 = 1.0 / P->inertia[1]; K.inv_i2 = 1.0 / P->inertia[2]; K.inv_dt = 1.0 / P->dt;
    const double park_z = P->ground_z + std::sqrt(P->coll_radius * P->coll_radius + P->coll_half_len * P->coll_half_len) + P->contact_threshold;
    const M3 Rb = quat_to_matrix_bullet(q[0], q[1], q[2], q[3]);
    integrate_velocity(*P, K, Rb, v, w, v3(fb[0], fb[1], fb[2]), v3(tb[0], tb[1], tb[2]));
    if (needs_contact(P->enable_contact, park_z, p[2]) && !(P->rest_shortcut && contact_at_rest(*P, K, p[2], q, v, w))) contact_stage(*P, K, p, q, v, w);
    integrate_pose(*P, p, q, v, w);
}

static Recips host_recips(const MrsParams &P)
{   // as fill_common_uncached (mrs_kernels.hip) forms it
    Recips K{};
    K.inv_mass = 1.0 / P.mass; K.inv_i0 = 1.0 / P.inertia[0]; K.inv_i1 = 1.0 / P.inertia[1]; K.inv_i2 = 1.0 / P.inertia[2];
    K.inv_4kf = 1.0 / (4 * P.kf); K.inv_dt = 1.0 / P.dt;
    const float dt32 = (float)P.ctrl_dt;
    uint32_t bits;
    memcpy(&bits, &dt32, sizeof(bits));
    K.inv_ctrl_dt32 = (float)(1.0 / (double)dt32);
    K.ctrl_div_fast = std::isnormal(dt32) && (bits & 0x7FFFFFu) != 0x7FFFFFu && std::isnormal(K.inv_ctrl_dt32);
    return K;
}
extern "C" int host_pid_bytes(void) { return (int)sizeof(Pid); }
extern "C" void host_pid_init(void *mem)
{
    Pid s = {};
    s.lvx = s.lvy = s.lvz = s.ltx = s.lty = s.ltz = __builtin_nanf("");   // "attribute not created yet" (mrs_hip.h, MrsBuffers.pid)
    memcpy(mem, &s, sizeof(s));
}
// One controller call of one quadcopter in the order k_step makes it (mrs_kernels.hip, "Outer loops of the cascade first ..."):
// mode 3 set_target_accel, 4 set_target_vel, 5 set_target_pos, 6 set_target_ori (MRS_ACT_*); the controller memory goes through
// the float32 records of MrsBuffers.pid between calls like in the kernel.  tests/test_device_math_host.py: reference fixture F1.
extern "C" void host_controller(const MrsParams *Pp, void *pid_mem, int mode, const float *pos, const float *euler, const float *vel, const float *angvel,
                                const float *act, double *rpm)
{
    const MrsParams &P = *Pp;
    const Recips K = host_recips(P);
    Pid s;
    memcpy(&s, pid_mem, sizeof(s));
    double p[3] = {pos[0], pos[1], pos[2]}, v[3] = {vel[0], vel[1], vel[2]}, w[3] = {angvel[0], angvel[1], angvel[2]}, q[4];
    euler_to_quat((double)euler[0], (double)euler[1], (double)euler[2], q);      // k_set_state
    V3 ta = v3(0., 0., 0.);
    if (mode == MRS_ACT_TARGET_VEL || mode == MRS_ACT_TARGET_POS) {
        Observed o0;
        observe<false, false>(p, q, v, w, o0);
        if (mode == MRS_ACT_TARGET_POS) ta = pos_control_accel(P, s, o0, act[0], act[1], act[2]);
        else ta = vel_control_accel(P, K, s, o0, act[0], act[1], act[2]);
    }
    Observed ob;
    M3 R;
    observe_ctrl(p, q, v, w, ob, R, P.round_euler_readback != 0);
    if (mode == MRS_ACT_TARGET_ORI) {
        const M3 Rt = euler_to_matrix((double)act[0], (double)act[1], (double)act[2]);
        attitude_control(P, K, s, Rt, R, ob, v3(0., 0., 9.81), 9.81, 1.0 / 9.81, rpm);
    } else {
        if (mode == MRS_ACT_TARGET_ACCEL) ta = v3((double)act[0], (double)act[1], (double)act[2]);
        accel_control(P, K, s, ta, R, ob, rpm);
    }
    // the records are float32
    s.ipx = (float)s.ipx; s.ipy = (float)s.ipy; s.ipz = (float)s.ipz; s.dvx = (float)s.dvx; s.dvy = (float)s.dvy; s.dvz = (float)s.dvz;
    s.ivx = (float)s.ivx; s.ivy = (float)s.ivy; s.ivz = (float)s.ivz; s.iox = (float)s.iox; s.ioy = (float)s.ioy; s.ioz = (float)s.ioz;
    memcpy(pid_mem, &s, sizeof(s));
}
extern "C" int host_set_control(const MrsParams *P, const float *c, double *rpm)
{
    set_control(*P, c[0], c[1], c[2], c[3], rpm);
    return 0;
}
