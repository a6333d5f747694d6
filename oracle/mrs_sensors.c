/*
 * mrs_sensors.c -- CPU restatement of the reference's geometry sensors (mrsgym/Object.py:100-174) against the
 * analytic scene of env_generator('simple') (EnvCreator.py:7-13): the ground box of plane.urdf:24 (30 x 30 x 1 m,
 * centred at the origin => top face z = ground_z) and one collision cylinder per quadcopter (cf2x.urdf:34:
 * radius .06, length .025, axis = body z).
 *
 * TEST INFRASTRUCTURE ONLY (see mrs_oracle.h).
 *
 * Parity status: PARITY UNPINNED.  The reference answers these queries through pybullet (rayTestBatch,
 * getClosestPoints, getContactPoints, getOverlappingObjects), which is absent here; the reference holds no fixtures.
 * What is restated is the geometry those calls are asked about, in exact arithmetic on the ideal shapes.  Known
 * Bullet-side differences, all below a millimetre: the URDF cylinder is imported as a 32-gon prism (sagitta 0.29 mm)
 * and convex shapes carry a collision margin that rounds their edges.  Pinned here instead: brute-force sampling
 * checks in tests/test_oracle_sensors.py.
 */
#include "mrs_oracle.h"

#include <math.h>
#include <string.h>

#define GROUND_HALF_XY 15.0 /* plane.urdf:24 <box size="30 30 1"/> */
#define GROUND_THICK 1.0

typedef struct { double x, y, z; } v3;
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul(double s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static double len(v3 a) { return sqrt(dot(a, a)); }

/* Bullet's matrix of the (un-normalised) state quaternion: the geometry queries see the body where Bullet has it */
static void qmat(const double q[4], double R[9])
{
    const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], s = 2.0 / d;
    const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
    const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
    const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
    R[0] = 1 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1 - (xx + yy);
}
static v3 mtv(const double R[9], v3 a) { return V(R[0] * a.x + R[3] * a.y + R[6] * a.z, R[1] * a.x + R[4] * a.y + R[7] * a.z, R[2] * a.x + R[5] * a.y + R[8] * a.z); }

/* ------------------------------------------------------------------------------------------------ rays */
/* segment o + t d, t in [0,1], against the capped cylinder |xy| <= rc, |z| <= hl in ITS frame; returns t or -1.
 * A segment that starts inside does not hit (Bullet's convex cast reports nothing from inside a convex shape). */
static double ray_cylinder(v3 o, v3 d, double rc, double hl)
{
    if (o.x * o.x + o.y * o.y <= rc * rc && fabs(o.z) <= hl) return -1.0;
    double best = -1.0;
    const double a = d.x * d.x + d.y * d.y;
    if (a > 0) { /* side wall */
        const double b = o.x * d.x + o.y * d.y, c = o.x * o.x + o.y * o.y - rc * rc, disc = b * b - a * c;
        if (disc >= 0) {
            const double t = (-b - sqrt(disc)) / a; /* entering root */
            if (t >= 0 && t <= 1 && fabs(o.z + t * d.z) <= hl) best = t;
        }
    }
    if (d.z != 0) { /* the cap facing the ray */
        const double zc = d.z > 0 ? -hl : hl, t = (zc - o.z) / d.z;
        if (t >= 0 && t <= 1) {
            const double x = o.x + t * d.x, y = o.y + t * d.y;
            if (x * x + y * y <= rc * rc && (best < 0 || t < best)) best = t;
        }
    }
    return best;
}

/* segment against the axis-aligned box [lo,hi]; entering parameter or -1 (starts inside: no hit) */
static double ray_box(v3 o, v3 d, v3 lo, v3 hi)
{
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    double t0 = 0, t1 = 1;
    int inside = 1;
    for (int k = 0; k < 3; ++k) inside &= (oo[k] >= l[k] && oo[k] <= h[k]);
    if (inside) return -1.0;
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0) {
            if (oo[k] < l[k] || oo[k] > h[k]) return -1.0;
        } else {
            double a = (l[k] - oo[k]) / dd[k], b = (h[k] - oo[k]) / dd[k];
            if (a > b) { const double s = a; a = b; b = s; }
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
            if (t0 > t1) return -1.0;
        }
    }
    return t0;
}

/* Object.raycast (Object.py:150-174) for agent `self` of one env.
 * pos[N][3], quat[N][4]: Bullet's float64 state.  offset[R][3], dirs[R][3]: float32 as the caller's tensors.
 * Outputs per ray: hit object (-1 none, 0..N-1 quadcopter, N ground), "pos world" (hit - rotated offset, :166),
 * "pos" (body frame, :167), "dist" (:171); misses are zeros (:168-170). */
void orc_raycast(const OrcParams *p, int N, const double *pos, const double *quat, int self, const float *offset,
                 const float *dirs, int n_rays, int body, float range, int *hit_obj, float *pos_world, float *pos_body, float *dist)
{
    /* what Object.get_ori(mat=True) / get_pos() hand to raycast: float32 (Object.py:86-97) */
    float opos[3], oe[3], ov[3], ow[3], Rm[9];
    const double zero[3] = {0, 0, 0};
    orc_observe(pos + 3 * self, quat + 4 * self, zero, zero, opos, oe, ov, ow, Rm);
    for (int r = 0; r < n_rays; ++r) {
        /* :151 directions *= RANGE ; :159-160 rotate into the world if body=True -- all float32 tensor arithmetic */
        float dl[3], of[3], dw[3], ofw[3];
        for (int k = 0; k < 3; ++k) { dl[k] = dirs[3 * r + k] * range; of[k] = offset[3 * r + k]; }
        for (int k = 0; k < 3; ++k) {
            if (body) {
                ofw[k] = Rm[3 * k] * of[0] + Rm[3 * k + 1] * of[1] + Rm[3 * k + 2] * of[2];
                dw[k] = Rm[3 * k] * dl[0] + Rm[3 * k + 1] * dl[1] + Rm[3 * k + 2] * dl[2];
            } else { ofw[k] = of[k]; dw[k] = dl[k]; }
        }
        float st[3], en[3];
        for (int k = 0; k < 3; ++k) { st[k] = ofw[k] + opos[k]; en[k] = dw[k] + st[k]; } /* :161-163 */
        const v3 o = V(st[0], st[1], st[2]), d = V((double)en[0] - st[0], (double)en[1] - st[1], (double)en[2] - st[2]);
        double best = -1.0;
        int obj = -1;
        const double tg = ray_box(o, d, V(-GROUND_HALF_XY, -GROUND_HALF_XY, p->ground_z - GROUND_THICK), V(GROUND_HALF_XY, GROUND_HALF_XY, p->ground_z));
        if (tg >= 0) { best = tg; obj = N; }
        for (int j = 0; j < N; ++j) {
            double Rj[9];
            qmat(quat + 4 * j, Rj);
            const v3 c = V(pos[3 * j], pos[3 * j + 1], pos[3 * j + 2]);
            const double t = ray_cylinder(mtv(Rj, sub(o, c)), mtv(Rj, d), p->coll_radius, p->coll_half_len);
            if (t >= 0 && (best < 0 || t < best)) { best = t; obj = j; }
        }
        hit_obj[r] = obj;
        if (obj < 0) {
            for (int k = 0; k < 3; ++k) { pos_world[3 * r + k] = 0; pos_body[3 * r + k] = 0; }
            dist[r] = 0;
            continue;
        }
        /* ray[3] = hit position (float64 in Bullet) -> torch.tensor(...) float32, minus the rotated offset (:166) */
        float pw[3];
        const double hit[3] = {o.x + best * d.x, o.y + best * d.y, o.z + best * d.z};
        for (int k = 0; k < 3; ++k) pw[k] = (float)hit[k] - ofw[k];
        /* :167 pos = R^T pos_world - R^T pos */
        float pb[3];
        for (int k = 0; k < 3; ++k) {
            const float a = Rm[k] * pw[0] + Rm[3 + k] * pw[1] + Rm[6 + k] * pw[2];
            const float b = Rm[k] * opos[0] + Rm[3 + k] * opos[1] + Rm[6 + k] * opos[2];
            pb[k] = a - b;
        }
        for (int k = 0; k < 3; ++k) { pos_world[3 * r + k] = pw[k]; pos_body[3 * r + k] = pb[k]; }
        dist[r] = sqrtf(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
    }
}

/* --------------------------------------------------------------------------------- closest points (GJK) */
typedef struct { v3 c, a; double rc, hl; } Cyl; /* centre, unit axis */
static v3 cyl_support(const Cyl *s, v3 d)
{
    const double da = dot(d, s->a);
    v3 rad = sub(d, mul(da, s->a));
    const double n = len(rad);
    v3 r = add(s->c, mul(da >= 0 ? s->hl : -s->hl, s->a));
    if (n > 1e-300) r = add(r, mul(s->rc / n, rad));
    return r;
}

typedef struct { v3 w[4], a[4], b[4]; double l[4]; int n; } Simplex;

/* closest point to the origin on the simplex; reduces it to the supporting face; returns 1 if the origin is inside
 * a tetrahedron.  Barycentric weights in s->l. */
static void closest_segment(Simplex *s)
{
    const v3 A = s->w[0], B = s->w[1], ab = sub(B, A);
    const double t = -dot(A, ab), dn = dot(ab, ab);
    if (t <= 0 || dn <= 0) { s->n = 1; s->l[0] = 1; return; }
    if (t >= dn) { s->w[0] = s->w[1]; s->a[0] = s->a[1]; s->b[0] = s->b[1]; s->n = 1; s->l[0] = 1; return; }
    s->l[1] = t / dn; s->l[0] = 1 - s->l[1];
}
static void keep(Simplex *s, int i0, int i1, int i2, int n, double l0, double l1, double l2)
{
    const int idx[3] = {i0, i1, i2};
    const double l[3] = {l0, l1, l2};
    Simplex t = *s;
    for (int k = 0; k < n; ++k) { s->w[k] = t.w[idx[k]]; s->a[k] = t.a[idx[k]]; s->b[k] = t.b[idx[k]]; s->l[k] = l[k]; }
    s->n = n;
}
static void closest_triangle(Simplex *s)
{ /* Ericson, Real-Time Collision Detection 5.1.5, for the point p = origin */
    const v3 a = s->w[0], b = s->w[1], c = s->w[2];
    const v3 ab = sub(b, a), ac = sub(c, a), ap = mul(-1, a);
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) { keep(s, 0, 0, 0, 1, 1, 0, 0); return; }
    const v3 bp = mul(-1, b);
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) { keep(s, 1, 0, 0, 1, 1, 0, 0); return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); keep(s, 0, 1, 0, 2, 1 - v, v, 0); return; }
    const v3 cp = mul(-1, c);
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) { keep(s, 2, 0, 0, 1, 1, 0, 0); return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); keep(s, 0, 2, 0, 2, 1 - w, w, 0); return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); keep(s, 1, 2, 0, 2, 1 - w, w, 0); return; }
    const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    s->l[0] = 1 - v - w; s->l[1] = v; s->l[2] = w;
}
static v3 simplex_point(const Simplex *s)
{
    v3 p = V(0, 0, 0);
    for (int k = 0; k < s->n; ++k) p = add(p, mul(s->l[k], s->w[k]));
    return p;
}
static int closest_tetra(Simplex *s)
{
    static const int F[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    static const int O[4] = {3, 1, 2, 0};
    Simplex best = *s;
    double bd = -1;
    int outside_any = 0;
    for (int f = 0; f < 4; ++f) {
        const v3 a = s->w[F[f][0]], b = s->w[F[f][1]], c = s->w[F[f][2]], d = s->w[O[f]];
        const v3 n = cross(sub(b, a), sub(c, a));
        const double so = dot(mul(-1, a), n), sd = dot(sub(d, a), n);
        if (so * sd < 0 || sd == 0) { /* origin on the other side of this face than the fourth vertex */
            outside_any = 1;
            Simplex t = *s;
            keep(&t, F[f][0], F[f][1], F[f][2], 3, 0, 0, 0);
            closest_triangle(&t);
            const v3 q = simplex_point(&t);
            const double dd = dot(q, q);
            if (bd < 0 || dd < bd) { bd = dd; best = t; }
        }
    }
    if (!outside_any) return 1;
    *s = best;
    return 0;
}

/* closest points of two convex cylinders; returns the distance (0 if they touch or overlap, pa = pb then) */
static double gjk_cyl_cyl(const Cyl *A, const Cyl *B, v3 *pa, v3 *pb)
{
    Simplex s;
    s.n = 0;
    v3 v = sub(A->c, B->c);
    if (dot(v, v) < 1e-24) v = V(1, 0, 0);
    for (int it = 0; it < 64; ++it) {
        const v3 sa = cyl_support(A, mul(-1, v)), sb = cyl_support(B, v), w = sub(sa, sb);
        const double vv = dot(v, v);
        if (s.n > 0 && vv - dot(v, w) <= 1e-14 * vv + 1e-30) break; /* no closer support point: v is the answer */
        int dup = 0; /* a support point already in the simplex: converged to rounding */
        for (int k = 0; k < s.n; ++k) dup |= (s.w[k].x == w.x && s.w[k].y == w.y && s.w[k].z == w.z);
        if (dup) break;
        const Simplex prev = s;
        s.w[s.n] = w; s.a[s.n] = sa; s.b[s.n] = sb; s.n++;
        int inside = 0;
        if (s.n == 1) s.l[0] = 1;
        else if (s.n == 2) closest_segment(&s);
        else if (s.n == 3) closest_triangle(&s);
        else inside = closest_tetra(&s);
        if (inside) { /* origin = sum l_i w_i inside the tetrahedron => sum l_i a_i = sum l_i b_i lies in both bodies */
            const v3 e1 = sub(s.w[1], s.w[0]), e2 = sub(s.w[2], s.w[0]), e3 = sub(s.w[3], s.w[0]), o = mul(-1, s.w[0]);
            const double vol = dot(e1, cross(e2, e3));
            double l1 = dot(o, cross(e2, e3)) / vol, l2 = dot(e1, cross(o, e3)) / vol, l3 = dot(e1, cross(e2, o)) / vol;
            if (!(fabs(vol) > 0)) { l1 = l2 = l3 = 0.25; }
            const double l0 = 1 - l1 - l2 - l3;
            const v3 x = add(add(mul(l0, s.a[0]), mul(l1, s.a[1])), add(mul(l2, s.a[2]), mul(l3, s.a[3])));
            *pa = *pb = x;
            return 0.0;
        }
        const v3 vn = simplex_point(&s);
        if (prev.n > 0 && dot(vn, vn) >= vv) { s = prev; break; } /* sliver simplex on a curved rim: no progress, keep the best */
        v = vn;
        if (dot(v, v) < 1e-24) break;
    }
    v3 a = V(0, 0, 0), b = V(0, 0, 0);
    for (int k = 0; k < s.n; ++k) { a = add(a, mul(s.l[k], s.a[k])); b = add(b, mul(s.l[k], s.b[k])); }
    *pa = a; *pb = b;
    const double d = len(sub(a, b));
    return d < 1e-12 ? 0.0 : d;
}

static Cyl make_cyl(const OrcParams *p, const double *pos, const double *quat)
{
    double R[9];
    qmat(quat, R);
    Cyl c;
    c.c = V(pos[0], pos[1], pos[2]);
    c.a = V(R[2], R[5], R[8]);
    const double n = len(c.a);
    c.a = mul(1.0 / n, c.a);
    c.rc = p->coll_radius; c.hl = p->coll_half_len;
    return c;
}

/* cylinder against the top face of the ground box: the lowest point of the cylinder (support point along -z; the
 * centre of the lower cap when the axis is vertical) and its foot on the plane; signed distance (< 0: penetration). */
static double cyl_ground(const OrcParams *p, const Cyl *c, v3 *pc, v3 *pg)
{
    const double az = c->a.z;
    v3 low = add(c->c, mul(az > 1e-12 ? -c->hl : (az < -1e-12 ? c->hl : 0.0), c->a)); /* on edge: the middle of the lowest line */
    v3 rad = sub(V(0, 0, -1), mul(-az, c->a)); /* -z minus its axial part */
    const double n = len(rad);
    if (n > 1e-9) low = add(low, mul(c->rc / n, rad));
    *pc = low;
    *pg = V(low.x, low.y, p->ground_z);
    return low.z - p->ground_z;
}

/* Object.get_dist (Object.py:119-133) of agent `self` against every other object of its env.
 * dist[N+1]: distance to quadcopter j (j < N; entry `self` = 0) and to the ground (index N);
 * pself[N+1][3], pother[N+1][3]: the closest points, world frame, float64. */
void orc_closest(const OrcParams *p, int N, const double *pos, const double *quat, int self, double *dist, double *pself, double *pother)
{
    const Cyl me = make_cyl(p, pos + 3 * self, quat + 4 * self);
    for (int j = 0; j <= N; ++j) {
        v3 a = me.c, b = me.c;
        double d = 0;
        if (j == N) d = cyl_ground(p, &me, &a, &b);
        else if (j != self) {
            const Cyl o = make_cyl(p, pos + 3 * j, quat + 4 * j);
            d = gjk_cyl_cyl(&me, &o, &a, &b);
        }
        dist[j] = d;
        pself[3 * j] = a.x; pself[3 * j + 1] = a.y; pself[3 * j + 2] = a.z;
        pother[3 * j] = b.x; pother[3 * j + 1] = b.y; pother[3 * j + 2] = b.z;
    }
}

/* all-pairs form for one env: D[N][N+1] (row i = orc_closest(self = i) distances) */
void orc_proximity(const OrcParams *p, int N, const double *pos, const double *quat, double *D)
{
    for (int i = 0; i < N; ++i) {
        const Cyl me = make_cyl(p, pos + 3 * i, quat + 4 * i);
        for (int j = 0; j <= N; ++j) {
            v3 a, b;
            double d = 0;
            if (j == N) d = cyl_ground(p, &me, &a, &b);
            else if (j != i) {
                const Cyl o = make_cyl(p, pos + 3 * j, quat + 4 * j);
                d = gjk_cyl_cyl(&me, &o, &a, &b);
            }
            D[(size_t)i * (N + 1) + j] = d;
        }
    }
}
