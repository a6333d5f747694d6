"""Batched stand-ins for the reference's per-object Python layer.

Environment  <- mrsgym/Environment.py:84-124 (the per-agent loops become tensor views)
QuadView     <- mrsgym/Object.py:78-97 + mrsgym/Quadcopter.py:21-22 (what a state_fn may call)
StateFnCompiler turns a user state_fn(quad) into either a fused observation spec (written by the
step kernel itself) or a vmapped tensor program -- never a Python loop over E*N agents.
GUI / debug entry points (Environment.py:127-306) are accepted and ignored: there is no GUI on a headless GPU
env.  The geometry sensors of Object.py:100-174 (raycast, get_dist,
get_contact_points, collision, get_closest_objects) are batched kernels against the analytic scene (mrs_sensors.hpp).
"""
import warnings

import torch

from . import native


class QuadView:
    """One quadcopter (index i) of every env; getters return float32 tensors like Object.get_*:
    shape (3,) when N_ENVS == 1 (the reference's shape), (E,3) otherwise."""

    def __init__(self, env, idx):
        self.env = env
        self._i = int(idx)
        self.uid = int(idx)

    def get_idx(self):  # Quadcopter.py:21-22
        return self._i

    def _take(self, planes):
        m = self.env._mrs
        sh = m.shard
        v = planes.view(planes.shape[0], sh.E, sh.N)[:, :, self._i].t().to(torch.float32)  # (E,k)
        return v[0] if sh.E == 1 else v

    def get_pos(self):  # Object.py:86-87
        return self._take(self.env._mrs.shard.pos)

    def get_vel(self):  # Object.py:78-79
        return self._take(self.env._mrs.shard.vel)

    def get_angvel(self, mat=False):  # Object.py:82-83
        return self._take(self.env._mrs.shard.angvel)

    def get_ori(self, mat=False):  # Object.py:90-97
        e = self.env.get_ori(mat=mat)
        sel = e[self._i] if self.env._mrs.shard.E == 1 else e[:, self._i]
        return sel

    def get_data(self, name):  # Object.py:37-38
        return self.env.get_data(name)

    def set_data(self, name, val):
        self.env.set_data(name, val)

    def set_state(self, pos=None, ori=None, vel=None, angvel=None):  # Object.py:42-65, one agent of every env
        m = self.env._mrs
        sh = m.shard

        def widen(x, cur):
            if x is None:
                return None
            full = cur.clone()
            full[:, self._i] = torch.as_tensor(x, device=sh.device, dtype=torch.float32).reshape(sh.E, -1)
            return full
        ori_full = None
        if ori is not None:
            ori_t = torch.as_tensor(ori, device=sh.device, dtype=torch.float32)
            k = ori_t.numel() // sh.E
            if k == 3:
                cur = self.env._batched(self.env.get_ori())
            elif k == 4:
                cur = sh.view(sh.quat).to(torch.float32)
            else:
                cur = self.env._batched(self.env.get_ori(mat=True)).reshape(sh.E, sh.N, 9)
            ori_full = cur.clone()
            ori_full[:, self._i] = ori_t.reshape(sh.E, -1)
        sh.set_state(pos=widen(pos, sh.view(sh.pos).to(torch.float32)), ori=ori_full,
                     vel=widen(vel, sh.view(sh.vel).to(torch.float32)),
                     angvel=widen(angvel, sh.view(sh.angvel).to(torch.float32)))

    # ------------------------------------------------------------------ sensors (Object.py:100-174)
    # One batched kernel answers the query for EVERY quadcopter of every env (Environment.raycast / proximity, cached
    # until the state changes); a QuadView picks its own row, so a per-agent callback loop costs one launch, not N.
    def raycast(self, offset=None, directions=None, body=True, RANGE=100.0):  # Object.py:150-174
        """dict: "object" (list of QuadView / GroundView / None for one env, else (E,R) int tensor: -1 miss, N ground),
        "pos world", "pos" (R,3), "dist" (R,).  The reference scales `directions` by RANGE IN PLACE (:151), which also
        corrupts its own default argument from the second call on; the caller's tensor is left alone here."""
        offset = torch.zeros(3) if offset is None else offset
        directions = torch.tensor([1., 0., 0.]) if directions is None else directions
        out = self.env.raycast(offset, directions, body=body, RANGE=RANGE)
        one = self.env._mrs.shard.E == 1
        pick = (lambda t: t[self._i]) if one else (lambda t: t[:, self._i])
        res = {k: pick(v) for k, v in out.items()}
        if one:
            res["object"] = [self.env.object_by_index(int(j)) for j in res["object"].tolist()]
        return res

    def get_dist(self, other, MAX_DIST=float('inf'), body=False):  # Object.py:119-133
        """One env: the reference's dict (one closest pair, or empty tensors beyond MAX_DIST).  Batched envs: tensors
        with a leading E axis plus a boolean 'valid' (E,) instead of empty results.  body=True returns the points in
        this quadcopter's frame, R^T (p - pos) (what Object.py:129-131 is after; as written it mis-broadcasts)."""
        j = other.uid if not isinstance(other, int) else other
        # pairs further apart than MAX_DIST are culled on their bounding spheres ahead of GJK (they come back as +inf)
        dist, ps, po = self.env.proximity(max_dist=float(MAX_DIST), points=True)
        sh = self.env._mrs.shard
        b = lambda t: t.unsqueeze(0) if sh.E == 1 else t
        d, a, c = b(dist)[:, self._i, j], b(ps)[:, self._i, j], b(po)[:, self._i, j]
        if body:
            Rm, p = b(self.env.get_ori(mat=True))[:, self._i], b(self.env.get_pos())[:, self._i]
            a = torch.einsum('eji,ej->ei', Rm, a - p)
            c = torch.einsum('eji,ej->ei', Rm, c - p)
        valid = d <= MAX_DIST
        if sh.E == 1:
            if not bool(valid[0]):
                return {'closest pos self': torch.zeros(0, 3), 'closest pos other': torch.zeros(0, 3), 'distance': torch.zeros(0)}
            return {'closest pos self': a, 'closest pos other': c, 'distance': d}
        return {'closest pos self': a, 'closest pos other': c, 'distance': d, 'valid': valid}

    def get_contact_points(self, other=None, body=False):  # Object.py:100-116
        """Objects within the contact threshold of this quadcopter (MrsParams.contact_threshold, Bullet's contact
        breaking threshold): one point per touching object -- the closest pair.  'normal force' is not retained by the
        fused step (its impulses live in registers): zeros.  Batched envs (N_ENVS > 1): the same content as PADDED tensors,
        'object' (E,M) int64 object indices (quadcopter index, N = the ground, -1 = padding), 'pos' (E,M,3), 'normal force'
        (E,M,3), 'distance' (E,M) (+inf padding) and 'count' (E,) valid entries per env, M = N; entries in ascending object index,
        as the one-env form lists them."""
        sh = self.env._mrs.shard
        if sh.E != 1:
            return self._contact_points_batched(other, body)
        thr = float(sh.params.contact_threshold)
        dist, ps, po = self.env.proximity(max_dist=thr, points=True)     # only pairs within the threshold go through GJK
        d = dist[self._i].clone()
        d[self._i] = float('inf')
        if other is not None:
            keep = torch.zeros_like(d, dtype=torch.bool)
            keep[other.uid if not isinstance(other, int) else other] = True
            d = torch.where(keep, d, torch.full_like(d, float('inf')))
        idx = torch.nonzero(d <= thr).flatten()
        if idx.numel() == 0:
            return {'object': [], 'pos': torch.zeros(0, 3), 'normal force': torch.zeros(0, 3), 'distance': torch.zeros(0)}
        pts = ps[self._i, idx]
        if body:
            Rm, p = self.env.get_ori(mat=True)[self._i], self.env.get_pos()[self._i]
            pts = (pts - p) @ Rm
        return {'object': [self.env.object_by_index(int(j)) for j in idx.tolist()], 'pos': pts,
                'normal force': torch.zeros(idx.numel(), 3, device=pts.device), 'distance': d[idx]}

    def _padded(self, keep, d):
        """keep (E,K) bool -> (order (E,K): the kept columns first, ascending, then the rest; count (E,))"""
        order = torch.sort((~keep).to(torch.int8), dim=1, stable=True).indices
        return order, keep.sum(1)

    def _contact_points_batched(self, other, body):
        sh = self.env._mrs.shard
        thr = float(sh.params.contact_threshold)
        dist, ps, po = self.env.proximity(max_dist=thr, points=True)   # (E,N,N+1), (E,N,N+1,3) x2; culled beyond the threshold
        d = dist[:, self._i].clone()                                # (E,N+1)
        d[:, self._i] = float('inf')
        if other is not None:
            j = other.uid if not isinstance(other, int) else other
            only = torch.zeros(d.shape[1], dtype=torch.bool, device=d.device); only[j] = True
            d = torch.where(only, d, torch.full_like(d, float('inf')))
        keep = d <= thr
        order, count = self._padded(keep, d)
        M = sh.N                                                    # at most N - 1 quadcopters + the ground
        order = order[:, :M]
        valid = torch.arange(M, device=d.device)[None, :] < count[:, None]
        pts = torch.gather(ps[:, self._i], 1, order[:, :, None].expand(-1, -1, 3))
        if body:
            Rm, p = self.env.get_ori(mat=True)[:, self._i], self.env.get_pos()[:, self._i]
            pts = torch.einsum('eji,emj->emi', Rm, pts - p[:, None])
        zero = torch.zeros_like(pts)
        return {'object': torch.where(valid, order, torch.full_like(order, -1)), 'pos': torch.where(valid[:, :, None], pts, zero),
                'normal force': zero, 'distance': torch.where(valid, torch.gather(d, 1, order), torch.full_like(d[:, :M], float('inf'))),
                'count': count}

    def collision(self):  # Object.py:136-137
        c = self.env.collisions()
        return bool(c[self._i]) if self.env._mrs.shard.E == 1 else c[:, self._i]

    def get_closest_objects(self, radius):  # Object.py:140-147
        """Objects whose bounding box overlaps pos +- radius and that are within `radius` of this quadcopter's hull
        (the reference's own two tests; it lists the quadcopter itself too, its AABB overlaps and its distance is 0)."""
        sh = self.env._mrs.shard
        if sh.E != 1:
            # batched envs: {'object': (E,M) int64 object indices in the one-env form's order (quadcopters ascending, then the
            # ground = N), -1 padded, M = N + 1; 'count': (E,)}
            dist = self.env.proximity(max_dist=float(radius))           # (E,N,N+1); +inf beyond the radius
            pos = self.env.get_pos()                                    # (E,N,3)
            bound = float((sh.params.coll_radius ** 2 + sh.params.coll_half_len ** 2) ** 0.5)
            me = pos[:, self._i]
            box = ((pos - me[:, None]).abs() <= radius + bound).all(-1)
            keep_q = box & (dist[:, self._i, :sh.N] <= radius)
            keep_g = (me[:, 2] - radius <= float(sh.params.ground_z)) & (dist[:, self._i, sh.N] <= radius)
            keep = torch.cat([keep_q, keep_g[:, None]], 1)
            order, count = self._padded(keep, None)
            valid = torch.arange(sh.N + 1, device=keep.device)[None, :] < count[:, None]
            return {'object': torch.where(valid, order, torch.full_like(order, -1)), 'count': count}
        dist = self.env.proximity(max_dist=float(radius))
        pos = self.env.get_pos()
        bound = float((sh.params.coll_radius ** 2 + sh.params.coll_half_len ** 2) ** 0.5)
        out = []
        for j in range(sh.N):
            box = bool(((pos[j] - pos[self._i]).abs() <= radius + bound).all())
            if box and float(dist[self._i, j]) <= radius:
                out.append(self.env.agents[j])
        gz = float(sh.params.ground_z)
        if float(pos[self._i, 2]) - radius <= gz and float(dist[self._i, sh.N]) <= radius:
            out.append(self.env.ground)
        return out

    def get_image(self, *a, **k):  # Object.py:177-195: a rasterised camera image -- GUI / rendering, out of scope
        raise NotImplementedError("camera images are rendered by pybullet's GUI/TinyRenderer; not on the accelerated path")


class GroundView:
    """The plane.urdf object of env_generator('simple') (EnvCreator.py:11): static 30 x 30 x 1 m box, top face z = ground_z."""

    def __init__(self, env, uid):
        self.env, self.uid = env, int(uid)

    def get_pos(self):
        return torch.zeros(3)

    def get_vel(self):
        return torch.zeros(3)

    get_angvel = get_vel

    def get_ori(self, mat=False):
        return torch.eye(3) if mat else torch.zeros(3)

    def collision(self):
        return bool(self.env.collisions(ground=True).any())


class Environment:
    """World container facade (Environment.py:8-124)."""

    def __init__(self, mrs):
        self._mrs = mrs
        self.sim = mrs.sim
        self.agents = [QuadView(self, i) for i in range(mrs.N_AGENTS)]
        self.ground = GroundView(self, mrs.N_AGENTS)   # uid N: the index the sensor kernels report for it
        self.objects = [self.ground]
        self.controlled = []
        self.object_dict = {a.uid: a for a in self.agents}
        self.object_dict[self.ground.uid] = self.ground
        self._sensor_cache = {}
        self.agent_idxs = {a: i for i, a in enumerate(self.agents)}
        self.data = {}
        self.debug_names = {}

    def _batched(self, t):
        return t.unsqueeze(0) if self._mrs.shard.E == 1 else t

    def _out(self, t):
        return t[0] if self._mrs.shard.E == 1 else t

    def set_data(self, name, val):  # Environment.py:26-31
        self.data[name] = val

    def get_data(self, name):
        return self.data.get(name, None)

    # state read-back: float32, (N,k) for one env, (E,N,k) otherwise  (Environment.py:111-124)
    def get_pos(self):
        sh = self._mrs.shard
        return self._out(sh.view(sh.pos).to(torch.float32))

    def get_vel(self):
        sh = self._mrs.shard
        return self._out(sh.view(sh.vel).to(torch.float32))

    def get_angvel(self):
        sh = self._mrs.shard
        return self._out(sh.view(sh.angvel).to(torch.float32))

    def get_ori(self, mat=False):
        """Object.get_ori: euler 'xyz' (roll,pitch,yaw) or 3x3 matrix of the float32-truncated quaternion."""
        sh = self._mrs.shard
        if not mat:
            out = torch.empty(sh.E, sh.N, 3, dtype=torch.float32, device=sh.device)
            sh.observe(out, fields=("ori",))
            return self._out(out)
        q = sh.view(sh.quat).to(torch.float32).to(torch.float64)
        q = q / q.norm(dim=-1, keepdim=True)
        x, y, z, w = q.unbind(-1)
        R = torch.stack([x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w], -1)
        return self._out(R.reshape(sh.E, sh.N, 3, 3).to(torch.float32))

    # ------------------------------------------------------------------ batched sensors (Object.py:100-174 for all agents)
    def object_by_index(self, j):
        """index reported by the sensor kernels -> object: -1 None, 0..N-1 quadcopters, N the ground"""
        return None if j < 0 else (self.ground if j == self._mrs.N_AGENTS else self.agents[j])

    def _cached(self, key, fn):
        ver = self._mrs.shard.version
        hit = self._sensor_cache.get(key)
        if hit is None or hit[0] != ver:
            if len(self._sensor_cache) > 16:
                self._sensor_cache.clear()
            hit = self._sensor_cache[key] = (ver, fn())
        return hit[1]

    def raycast(self, offset, directions, body=True, RANGE=100.0):
        """Object.raycast (Object.py:150-174) from every quadcopter: dict of (N,R,..) / (E,N,R,..) tensors; "object" holds
        indices (-1 miss, j < N quadcopter j, N the ground)."""
        o = torch.as_tensor(offset, dtype=torch.float32).reshape(-1, 3)
        d = torch.as_tensor(directions, dtype=torch.float32).reshape(-1, 3)
        key = ("ray", tuple(o.flatten().tolist()), tuple(d.flatten().tolist()), bool(body), float(RANGE))
        out = self._cached(key, lambda: self._mrs.shard.raycast(o, d, body=body, RANGE=RANGE))
        return {k: self._out(v) for k, v in out.items()}

    def proximity(self, max_dist=float('inf'), points=False):
        """Object.get_dist (Object.py:119-133) for every ordered pair: distances (N,N+1) / (E,N,N+1), column N = ground."""
        out = self._cached(("prox", float(max_dist), bool(points)), lambda: self._mrs.shard.proximity(max_dist, points))
        return tuple(self._out(t) for t in out) if points else self._out(out)

    def collisions(self, ground=False):
        """Object.collision (Object.py:136-137) for every quadcopter: (N,) / (E,N) bool -- something within the contact
        threshold (a contact point exists; its distance is then < 0.04).  ground=True: only contacts with the ground."""
        sh = self._mrs.shard
        thr = float(sh.params.contact_threshold)
        d = self._cached(("prox", thr, False), lambda: sh.proximity(thr, False))
        if ground:
            return self._out(d[..., sh.N] <= thr)
        eye = torch.zeros(sh.N, sh.N + 1, dtype=torch.bool, device=d.device)
        eye[torch.arange(sh.N), torch.arange(sh.N)] = True
        return self._out((d.masked_fill(eye, float('inf')) <= thr).any(-1))

    def get_X(self, state_fn):  # Environment.py:84-87
        return self._mrs._obs.evaluate(state_fn)

    def set_state(self, pos, ori, vel, angvel):  # Environment.py:97-103 (None = keep)
        self._mrs.shard.set_state(pos=pos, ori=ori, vel=vel, angvel=angvel)

    def set_actions(self, actions, behaviour='set_controls'):  # Environment.py:90-94 (forces only exist inside step)
        raise NotImplementedError("set_actions is fused into MRS.step on this backend")

    def update_controlled(self):  # Environment.py:106-108
        for c in self.controlled:
            c.update(self)

    # GUI / debug: no-ops
    def draw_links(self, A):
        return None

    def get_keyboard_events(self):
        return {}

    def get_mouse_events(self):
        return None

    def _noop(self, *a, **k):
        return None

    add_line = add_text = add_param = remove_debug = set_colour = set_camera = record = set_collisions = _noop

    def read_param(self, name):
        return None


class _ProbeQuad:
    """Records which getters a state_fn calls, handing out random float32 vectors."""

    def __init__(self, gen):
        self.calls = []
        self._gen = gen

    def _mk(self, name, k=3):
        v = torch.rand(k, generator=self._gen) * 2 - 1
        self.calls.append((name, v))
        return v.clone()

    def get_pos(self):
        return self._mk("pos")

    def get_vel(self):
        return self._mk("vel")

    def get_ori(self, mat=False):
        if mat:
            raise _NotFusable()
        return self._mk("ori")

    def get_angvel(self, mat=False):
        return self._mk("angvel")

    def __getattr__(self, name):
        raise _NotFusable()


class _NotFusable(Exception):
    pass


class _VmapQuad:
    """Per-agent view handed to a generic state_fn under torch.func.vmap."""

    def __init__(self, env, pos, vel, ori, angvel, idx):
        self.env, self._p, self._v, self._o, self._w, self._idx = env, pos, vel, ori, angvel, idx

    def get_pos(self):
        return self._p

    def get_vel(self):
        return self._v

    def get_ori(self, mat=False):
        if mat:
            raise NotImplementedError("get_ori(mat=True) inside a non-fusable state_fn")
        return self._o

    def get_angvel(self, mat=False):
        return self._w

    def get_idx(self):
        return self._idx

    def get_data(self, name):
        # magent.py:35-37 style: quad.get_data("target_vel")[quad.get_idx(), :] -- under vmap the index is a batched
        # 0-d tensor, which plain tensor indexing would .item(); hand out a view that indexes with index_select
        val = self.env.get_data(name)
        if isinstance(val, torch.Tensor) and val.dim() >= 1:
            return _AgentIndexable(val, self._idx)
        return val


class _AgentIndexable:
    """A tensor of env data that a vmapped state_fn may index with its own (batched) agent index."""

    def __init__(self, t, idx):
        self._t, self._idx = t, idx

    def __getitem__(self, key):
        first, rest = (key[0], key[1:]) if isinstance(key, tuple) else (key, ())
        if first is self._idx:
            row = torch.index_select(self._t, 0, first.reshape(1).to(self._t.device)).squeeze(0)
            return row[rest] if rest else row
        return self._t[key]

    def __getattr__(self, name):
        return getattr(self._t, name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        unwrap = lambda x: x._t if isinstance(x, _AgentIndexable) else x
        args = tuple(unwrap(a) for a in args)
        kwargs = {k: unwrap(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)


class StateFnCompiler:
    """state_fn(quad) -> 1-D tensor (Environment.py:85).

    1. OBS=("pos","vel",...) given: fused, the step kernel writes the slice itself.
    2. state_fn recognised as a concatenation of getters (README.md:28-29 cat(pos, vel) and friends):
       fused as well.  Recognition = run it twice on probe objects that return random vectors and
       check the result is exactly the concatenation of what it asked for.
    3. anything else: evaluated for all E*N agents at once with torch.func.vmap over device tensors.
    """

    def __init__(self, mrs):
        self.m = mrs
        self.fused = False
        self.fields = None
        self._compiled = False
        self._warned_loop = False

    def _recognise(self, fn):
        specs = []
        for seed in (1, 2):
            g = torch.Generator().manual_seed(seed)
            q = _ProbeQuad(g)
            try:
                out = fn(q)
            except Exception:
                return None
            if not isinstance(out, torch.Tensor) or out.dim() != 1 or not q.calls:
                return None
            want = torch.cat([v for _, v in q.calls])
            if out.shape != want.shape or not torch.equal(out.to(torch.float32), want):
                return None
            specs.append(tuple(n for n, _ in q.calls))
        if specs[0] != specs[1] or len(specs[0]) > 8:
            return None
        return specs[0]

    def compile(self):
        m = self.m
        fields = None
        if m.OBS is not None:
            fields = tuple(m.OBS)
        elif m.state_fn is not None:
            fields = self._recognise(m.state_fn)
        if fields is not None:
            self.fused, self.fields = True, fields
            m.shard.set_obs_fields(fields)
            m._ensure_xbuf(m.shard.D)
        else:
            self.fused = False
        self._compiled = True

    def evaluate(self, fn):
        """Environment.get_X(fn): (N,D) / (E,N,D) float32."""
        m = self.m
        sh = m.shard
        if fn is None:
            raise TypeError("'NoneType' object is not callable")   # what Environment.py:85 does without a state_fn
        env = m.env
        pos = sh.view(sh.pos).to(torch.float32).reshape(sh.T, 3)
        vel = sh.view(sh.vel).to(torch.float32).reshape(sh.T, 3)
        ang = sh.view(sh.angvel).to(torch.float32).reshape(sh.T, 3)
        ori = env._batched(env.get_ori()).reshape(sh.T, 3)
        idx = torch.arange(sh.N, device=sh.device).repeat(sh.E)

        def one(p, v, o, w, i):
            out = fn(_VmapQuad(env, p, v, o, w, i))
            return out if isinstance(out, torch.Tensor) else torch.as_tensor(out)
        try:
            X = torch.func.vmap(one)(pos, vel, ori, ang, idx)
        except Exception as exc:
            if sh.E != 1:
                raise
            # single env: the reference's own per-agent loop (Environment.py:85-86) -- correct but N Python calls
            # per step; say so once instead of being quietly slow
            if not self._warned_loop:
                self._warned_loop = True
                warnings.warn("state_fn cannot be batched with torch.func.vmap (%s: %s); falling back to the per-agent "
                              "Python loop of the reference (Environment.py:85-86)" % (type(exc).__name__, exc), RuntimeWarning)
            X = torch.stack([torch.as_tensor(fn(a)) for a in env.agents], dim=0)
        X = X.to(torch.float32).reshape(sh.E, sh.N, -1)
        return X[0] if sh.E == 1 else X

    def write_into(self, out):
        """Fill `out` (E,N,D) with the current X."""
        m = self.m
        if self.fused:
            m.shard.observe(out)
        else:
            X = self.evaluate(m.state_fn)
            out.copy_(X.unsqueeze(0) if m.shard.E == 1 else X)
        return out
