"""Batched stand-ins for the reference's per-object Python layer.

Environment  <- mrsgym/Environment.py:84-124 (the per-agent loops become tensor views)
QuadView     <- mrsgym/Object.py:78-97 + mrsgym/Quadcopter.py:21-22 (what a state_fn may call)
StateFnCompiler turns a user state_fn(quad) into either a fused observation spec (written by the
step kernel itself) or a vmapped tensor program -- never a Python loop over E*N agents.
GUI / debug / sensor entry points (Environment.py:127-306, Object.py:100-195) are accepted and
ignored: there is no GUI on a headless GPU env.
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

    # sensors: out of scope on the accelerated path
    def collision(self):
        raise NotImplementedError("contact/raycast/camera sensors are not on the accelerated path")

    get_contact_points = get_dist = get_closest_objects = raycast = get_image = lambda self, *a, **k: QuadView.collision(self)


class Environment:
    """World container facade (Environment.py:8-124)."""

    def __init__(self, mrs):
        self._mrs = mrs
        self.sim = mrs.sim
        self.agents = [QuadView(self, i) for i in range(mrs.N_AGENTS)]
        self.objects = []            # the ground box is implicit in MrsParams.ground_z
        self.controlled = []
        self.object_dict = {a.uid: a for a in self.agents}
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
