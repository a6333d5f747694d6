"""Writes tools/host_f32/contact_variant.inc: the text of contact_solve_f32 (mrs_device.hpp) with its scalar type as a template
parameter CT, so that the host harness can run the SAME statements in float64 (what does float32 cost, and where)."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "mrs-gym_amd", "csrc", "mrs_device.hpp")).read()
a = s.index("MRS_DEV void contact_solve_f32(")
b = s.index("\n}\n", a) + 3
f = s[a:b]
f = f.replace("MRS_DEV void contact_solve_f32(const MrsParams &P, const Recips &K, double pz, const double q[4], const V3 &v, const V3 &w, F3 &dv_out, F3 &dw_out,\n                               float *diag = nullptr)",
              "template <typename CT, typename ST> void contact_solve_t(const MrsParams &P, const Recips &K, double pz, const double q[4], const V3 &v, const V3 &w, double *dv_out, double *dw_out, int mode)")
f = f.replace("dv_out = F3{0.f, 0.f, 0.f}; dw_out = F3{0.f, 0.f, 0.f};", "for (int i = 0; i < 3; ++i) dv_out[i] = dw_out[i] = 0;")
f = f.replace("dv_out = F3{dvx, dvy, dvz}; dw_out = F3{dwxy.x, dwxy.y, dwz};", "dv_out[0] = dvx; dv_out[1] = dvy; dv_out[2] = dvz; dw_out[0] = dwxy.x; dw_out[1] = dwxy.y; dw_out[2] = dwz;")
f = re.sub(r"\bF3\b", "T3<CT>", f)
f = re.sub(r"\bF2\b", "T2<CT>", f)
f = re.sub(r"\bfloat\b", "CT", f)
f = f.replace("__builtin_fmaf", "tfma").replace("__builtin_amdgcn_rcpf", "trcp").replace("__builtin_amdgcn_fmed3f", "tmed3")
f = f.replace("__builtin_elementwise_fma", "tpfma").replace("fmaxf", "tmax").replace("fabsf", "tabs")
f = f.replace("__builtin_amdgcn_sched_barrier(0);", "")
open(os.path.join(ROOT, "tools", "host_f32", "contact_variant.inc"), "w").write(f)
print(len(f.splitlines()), "lines")

# --- second form: the set-up (geometry, effective masses, right-hand sides) in CT, the sweeps in ST
cut = f.index("    if (!any) return;")
head_end = f.index("{", f.index("contact_solve_t")) + 1
A, B = f[head_end:cut], f[cut:]
A = A.replace("#pragma clang fp contract(off)", "")
names = ["ln", "lx", "ly", "Kn", "Kx", "Ky", "rhs", "r", "anxy", "axxy", "ayxy", "anz", "axz", "ayz", "im", "w0x", "w0y", "w0z", "any"]
for n in names:
    A = re.sub(r"\b%s\b" % n, n + "_a", A)
A = A.replace("for (int i = 0; i < 3; ++i) dv_out[i] = dw_out[i] = 0;", "")
B = re.sub(r"\bCT\b", "ST", B)
g = f[:head_end].replace("contact_solve_t", "contact_solve_split") + '''
#pragma clang fp contract(off)
    for (int i = 0; i < 3; ++i) dv_out[i] = dw_out[i] = 0;
    ST ln[4], lx[4], ly[4], Kn[4], Kx[4], Ky[4], rhs[4], anz[4], axz[4], ayz[4], im, w0x, w0y, w0z;
    T3<ST> r[4];
    T2<ST> anxy[4], axxy[4], ayxy[4];
    bool any;
    const auto fm = [](ST a, ST b, ST c) { return tfma(a, b, c); };
    {
''' + A + '''
        for (int k = 0; k < 4; ++k) {
            ln[k] = (ST)ln_a[k]; lx[k] = (ST)lx_a[k]; ly[k] = (ST)ly_a[k]; Kn[k] = (ST)Kn_a[k]; Kx[k] = (ST)Kx_a[k]; Ky[k] = (ST)Ky_a[k]; rhs[k] = (ST)rhs_a[k];
            anz[k] = (ST)anz_a[k]; axz[k] = (ST)axz_a[k]; ayz[k] = (ST)ayz_a[k];
            r[k] = T3<ST>{(ST)r_a[k].x, (ST)r_a[k].y, (ST)r_a[k].z};
            anxy[k] = T2<ST>{(ST)anxy_a[k].x, (ST)anxy_a[k].y}; axxy[k] = T2<ST>{(ST)axxy_a[k].x, (ST)axxy_a[k].y}; ayxy[k] = T2<ST>{(ST)ayxy_a[k].x, (ST)ayxy_a[k].y};
        }
        im = (ST)im_a; w0x = (ST)w0x_a; w0y = (ST)w0y_a; w0z = (ST)w0z_a; any = any_a;
    }
''' + B
open(os.path.join(ROOT, "tools", "host_f32", "contact_split.inc"), "w").write(g)
