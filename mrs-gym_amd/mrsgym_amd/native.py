"""ctypes binding of the C-ABI in include/mrs_hip.h (libmrs_hip.so).

There is NO CPU fallback: if the HIP library is missing or no GPU is present the product path
raises.  PyTorch is used only to own device memory and streams.
"""
import ctypes as C
import math
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRS_HIP_LIB") or os.path.join(os.path.dirname(_PKG), "lib", "libmrs_hip.so")  # env: A/B kernel builds

ACT = {
    None: 0,
    "set_speeds": 1,
    "set_control": 2,
    "set_target_accel": 3,
    "set_target_vel": 4,
    "set_target_pos": 5,
    "set_target_ori": 6,
}
ACT_DIM = {0: 0, 1: 4, 2: 4, 3: 3, 4: 3, 5: 3, 6: 3}
OBS = {"pos": 0, "vel": 1, "ori": 2, "euler": 2, "angvel": 3, "quat": 4}
OBS_WIDTH = {0: 3, 1: 3, 2: 3, 3: 3, 4: 4}
ORI_EULER, ORI_QUAT, ORI_MATRIX = 0, 1, 2
STATUS_NAN_ACTION, STATUS_SPAWN_FAIL, STATUS_SPAWN_MORE = 1, 2, 4
EXPORTS = [
    "mrs_abi_version", "mrs_last_error", "mrs_params_default", "mrs_params_derived", "mrs_create", "mrs_destroy",
    "mrs_set_params", "mrs_adj_words", "mrs_obs_dim", "mrs_pid_reset", "mrs_set_state", "mrs_set_state_f64",
    "mrs_step", "mrs_observe", "mrs_adjacency", "mrs_adjacency_expand", "mrs_spawn", "mrs_reynolds",
    "mrs_raycast", "mrs_proximity", "mrs_flock_metrics", "mrs_spawn_from", "mrs_step_n",
]


class MrsParams(C.Structure):
    _fields_ = [
        ("mass", C.c_double), ("arm", C.c_double), ("kf", C.c_double), ("km", C.c_double), ("thrust2weight", C.c_double),
        ("ixx_file", C.c_double), ("iyy_file", C.c_double), ("izz_file", C.c_double),
        ("gnd_eff_coeff", C.c_double), ("prop_radius", C.c_double), ("drag_xy", C.c_double), ("drag_z", C.c_double),
        ("dw1", C.c_double), ("dw2", C.c_double), ("dw3", C.c_double),
        ("prop_x", C.c_double * 4), ("prop_y", C.c_double * 4), ("prop_z", C.c_double * 4),
        ("coll_radius", C.c_double), ("coll_half_len", C.c_double),
        ("gravity", C.c_double), ("dt", C.c_double), ("ctrl_gravity", C.c_double), ("ctrl_dt", C.c_double),
        ("inertia", C.c_double * 3),
        ("lin_damp", C.c_double), ("ang_damp", C.c_double), ("max_coord_vel", C.c_double),
        ("use_gyro", C.c_int32), ("enable_contact", C.c_int32),
        ("ground_z", C.c_double), ("friction", C.c_double), ("erp", C.c_double), ("contact_threshold", C.c_double),
        ("solver_iters", C.c_int32), ("round_euler_readback", C.c_int32),
        ("pair_contact", C.c_int32), ("rest_shortcut", C.c_int32),
    ]


class MrsBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("pos", "quat", "vel", "angvel", "pid", "obs", "adj", "rpm", "status", "adj_dense")]


_lib = None
_NAN = float("nan")


class MrsNativeError(RuntimeError):
    pass


def lib():
    """Load libmrs_hip.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MrsNativeError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
        L.mrs_abi_version.restype = C.c_int
        L.mrs_last_error.restype = C.c_char_p
        L.mrs_params_default.argtypes = [C.POINTER(MrsParams)]
        L.mrs_params_derived.argtypes = [C.POINTER(MrsParams), C.POINTER(C.c_double)]
        L.mrs_create.argtypes = [C.POINTER(MrsParams), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.mrs_destroy.argtypes = [vp]
        L.mrs_destroy.restype = None
        L.mrs_set_params.argtypes = [vp, C.POINTER(MrsParams)]
        L.mrs_adj_words.argtypes = [C.c_int]
        L.mrs_obs_dim.argtypes = [i32p, C.c_int]
        L.mrs_pid_reset.argtypes = [vp, C.POINTER(MrsBuffers), vp, vp]
        L.mrs_set_state.argtypes = [vp, C.POINTER(MrsBuffers), vp, vp, C.c_int, vp, vp, vp, vp]
        L.mrs_set_state_f64.argtypes = [vp, C.POINTER(MrsBuffers), vp, vp, vp, vp, vp, vp]
        L.mrs_step.argtypes = [vp, C.POINTER(MrsBuffers), vp, C.c_int, i32p, C.c_int, C.c_double, vp]
        L.mrs_step_n.argtypes = [vp, C.POINTER(MrsBuffers), vp, C.c_int, C.c_int, C.c_int64, i32p, C.c_int, C.c_double, C.c_int64, C.c_int64, vp]
        L.mrs_observe.argtypes = [vp, C.POINTER(MrsBuffers), i32p, C.c_int, vp]
        L.mrs_adjacency.argtypes = [vp, C.POINTER(MrsBuffers), C.c_double, vp]
        L.mrs_adjacency_expand.argtypes = [vp, vp, vp, C.c_int, vp]
        L.mrs_spawn.argtypes = [vp, C.POINTER(MrsBuffers), C.c_uint64, C.c_int64, C.c_double,
                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, vp, vp]
        L.mrs_reynolds.argtypes = [vp, vp, C.c_int, vp, vp]
        L.mrs_raycast.argtypes = [vp, C.POINTER(MrsBuffers), vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp]
        L.mrs_proximity.argtypes = [vp, C.POINTER(MrsBuffers), C.c_double, vp, vp, vp, vp]
        L.mrs_spawn_from.argtypes = [vp, C.POINTER(MrsBuffers), vp, C.c_int, C.c_int, C.c_double, vp, vp]
        L.mrs_flock_metrics.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
        for n in EXPORTS:
            if n not in ("mrs_last_error", "mrs_destroy"):
                getattr(L, n).restype = C.c_int
        if L.mrs_abi_version() != 5:
            raise MrsNativeError("libmrs_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise MrsNativeError("%s failed (%d): %s" % (what, rc, lib().mrs_last_error().decode()))


def default_params():
    p = MrsParams()
    _check(lib().mrs_params_default(C.byref(p)), "mrs_params_default")
    return p


def derived(p):
    out = (C.c_double * 7)()
    _check(lib().mrs_params_derived(C.byref(p), out), "mrs_params_derived")
    names = ["GravityForce", "HoverRPM", "MaxRPM", "MaxThrust", "MaxXYTorque", "MaxZTorque", "GroundEffectHClip"]
    return dict(zip(names, list(out)))


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the stream handle without building a Stream object


def _stream_handle(device_index, device):
    """Handle of torch's current stream on `device`: 0.3 us through the raw accessor against 4.3 us through
    torch.cuda.current_stream() (a quarter of the host cost of one step call, tools/host_cost.py)."""
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device).cuda_stream


def _stream(device):
    return C.c_void_p(_stream_handle(device.index if device.index is not None else torch.cuda.current_device(), device))


def flock_metrics(X, want=("separation", "cohesion", "cohesion_noleader", "dist_to_leader", "vel_stddev")):
    """mrs_flock_metrics on frames X (..., N, D>=6) float32 (device): dict of tensors shaped like the leading axes."""
    if not X.is_cuda:
        raise MrsNativeError("flock_metrics runs on the GPU only (X is on %s)" % X.device)
    X = X.to(torch.float32).contiguous()
    lead_shape, N, D = X.shape[:-2], X.shape[-2], X.shape[-1]
    M = int(X.numel() // (N * D)) if N * D else 0
    out = {}
    for k in want:
        out[k] = torch.empty(lead_shape + ((N,) if k == "separation" else ()), dtype=torch.float32, device=X.device)
    g = lambda k: _ptr(out[k]) if k in out else None
    with torch.cuda.device(X.device):
        _check(lib().mrs_flock_metrics(_ptr(X), M, N, D, g("separation"), g("cohesion"), g("cohesion_noleader"), g("dist_to_leader"),
                                       g("vel_stddev"), _stream(X.device)), "mrs_flock_metrics")
    return out


class SwarmShard:
    """Device-resident state of E envs x N quadcopters on one GPU + the kernels that advance it.

    Thin object over the C-ABI: owns the torch tensors (SoA, agent-major) and a MrsHandle.
    """

    def __init__(self, n_envs, n_agents, device, params=None, obs_fields=("pos", "vel"), want_rpm=False):
        device = torch.device(device)
        if device.type != "cuda":
            raise MrsNativeError("mrsgym_amd runs on an AMD GPU only (device=%s); there is no CPU path" % device)
        if not torch.cuda.is_available():
            raise MrsNativeError("no GPU visible to PyTorch-ROCm")
        self.L = lib()
        self.E, self.N = int(n_envs), int(n_agents)
        self.T = self.E * self.N
        self.device = device
        self.params = params or default_params()
        self.W = self.L.mrs_adj_words(self.N)
        h = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        _check(self.L.mrs_create(C.byref(self.params), self.E, self.N, idx, C.byref(h)), "mrs_create")
        self.h = h
        self._dev_index = int(idx)
        f64 = dict(dtype=torch.float64, device=device)
        self.pos = torch.zeros(3, self.T, **f64)
        self.quat = torch.zeros(4, self.T, **f64)
        self.quat[3] = 1.0
        self.vel = torch.zeros(3, self.T, **f64)
        self.angvel = torch.zeros(3, self.T, **f64)
        self.pid = torch.zeros(5, self.T, 4, dtype=torch.float32, device=device)   # five planes of 16-byte records (include/mrs_hip.h)
        self.pid[2, :, 2:] = float("nan")     # last_vel_e / last_target_vel "attribute not created yet"
        self.pid[3] = float("nan")
        self.status = torch.zeros(self.E, dtype=torch.int32, device=device)
        self.rpm = torch.zeros(4, self.T, dtype=torch.float32, device=device) if want_rpm else None
        self.version = 0                      # bumped by everything that changes the state (sensor caches key on it)
        self.set_obs_fields(obs_fields)
        self.adj = torch.zeros(self.E, self.N, self.W, dtype=torch.int64, device=device)
        self.obs = None
        self._pb = self._buffers()
        self._pb_ref = C.byref(self._pb)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.mrs_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def set_obs_fields(self, fields):
        codes = [OBS[f] if isinstance(f, str) else int(f) for f in fields]
        self.obs_codes = (C.c_int32 * max(1, len(codes)))(*codes)
        self.n_obs = len(codes)
        self.D = sum(OBS_WIDTH[c] for c in codes)

    def set_params(self, params):
        self.params = params
        _check(self.L.mrs_set_params(self.h, C.byref(params)), "mrs_set_params")

    def _buffers(self, obs=None, adj=None, adj_dense=None):
        b = MrsBuffers()
        b.adj_dense = adj_dense.data_ptr() if adj_dense is not None else None
        b.pos, b.quat, b.vel, b.angvel = self.pos.data_ptr(), self.quat.data_ptr(), self.vel.data_ptr(), self.angvel.data_ptr()
        b.pid = self.pid.data_ptr()
        b.obs = obs.data_ptr() if obs is not None else None
        b.adj = adj.data_ptr() if adj is not None else None
        b.rpm = self.rpm.data_ptr() if self.rpm is not None else None
        b.status = self.status.data_ptr()
        return b

    @staticmethod
    def _f32(x, device, shape):
        if x is None:
            return None
        t = torch.as_tensor(x, device=device).to(torch.float32)
        if t.numel() * shape[0] == shape[0] * shape[1] * shape[2] and shape[0] > 1:
            t = t.reshape(shape[1:]).unsqueeze(0).expand(shape)     # one (N,k) value shared by every env
        return t.reshape(shape).contiguous()

    def _out(self, t, name, shape, dtype):
        """An output tensor of a tensor-taking method: the kernels write `shape` elements of `dtype` through its
        data_ptr(), so anything smaller, of another type, on another device or strided is refused HERE -- a mis-sized
        buffer is otherwise a GPU memory fault, not an exception (round 3: a (E,N,1) adjacency buffer handed to an
        N = 256 swarm, four words per row, took the process down with "Write access to a read-only page").  The
        `*_ptr` fast paths of MRS.step skip this: their slots are allocated by HistoryRing with these shapes."""
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            raise ValueError("%s must be a torch tensor, got %s" % (name, type(t).__name__))
        n = 1
        for k in shape:
            n *= int(k)
        if t.device != self.device:
            raise ValueError("%s is on %s, the shard is on %s" % (name, t.device, self.device))
        if t.dtype != dtype:
            raise ValueError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
        if t.numel() < n:
            raise ValueError("%s has %d elements, the kernel writes %s = %d" % (name, t.numel(), "x".join(str(int(k)) for k in shape), n))
        if t.data_ptr() % t.element_size():
            raise ValueError("%s must be aligned to its element size" % name)
        return t

    # ------------------------------------------------------------------ C-ABI calls
    def set_state(self, pos=None, ori=None, vel=None, angvel=None, env_mask=None):
        """Environment.set_state semantics: None keeps the current value (MRS.set)."""
        E, N, dev = self.E, self.N, self.device
        pos = self._f32(pos, dev, (E, N, 3))
        vel = self._f32(vel, dev, (E, N, 3))
        angvel = self._f32(angvel, dev, (E, N, 3))
        kind = ORI_EULER
        if ori is not None:
            ori = torch.as_tensor(ori, device=dev).to(torch.float32)
            if E > 1 and ori.numel() in (3 * N, 4 * N, 9 * N) and ori.shape[0] == N:
                ori = ori.unsqueeze(0).expand(E, *ori.shape)         # one (N,k) value shared by every env
            per_agent = ori.numel() // (E * N)      # 3 euler | 4 quat | 9 matrix (Object.py:51-56)
            if per_agent * E * N != ori.numel() or per_agent not in (3, 4, 9):
                raise ValueError("ori has %d elements, expected (E,N,3|4|3x3)" % ori.numel())
            kind = {3: ORI_EULER, 4: ORI_QUAT, 9: ORI_MATRIX}[per_agent]
            ori = ori.reshape(E, N, per_agent).contiguous()
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=dev).to(torch.uint8).contiguous()
        b = self._buffers()
        self.version += 1
        _check(self.L.mrs_set_state(self.h, C.byref(b), _ptr(pos), _ptr(ori), kind, _ptr(vel), _ptr(angvel), _ptr(mask),
                                    _stream(dev)), "mrs_set_state")

    def set_state_f64(self, pos=None, quat=None, vel=None, angvel=None, env_mask=None):
        dev = self.device

        def g(x, k):
            return None if x is None else torch.as_tensor(x, device=dev).to(torch.float64).reshape(self.E, self.N, k).contiguous()
        pos, quat, vel, angvel = g(pos, 3), g(quat, 4), g(vel, 3), g(angvel, 3)
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=dev).to(torch.uint8).contiguous()
        b = self._buffers()
        self.version += 1
        _check(self.L.mrs_set_state_f64(self.h, C.byref(b), _ptr(pos), _ptr(quat), _ptr(vel), _ptr(angvel), _ptr(mask),
                                        _stream(dev)), "mrs_set_state_f64")

    def pid_reset(self, env_mask=None):
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
        b = self._buffers()
        _check(self.L.mrs_pid_reset(self.h, C.byref(b), _ptr(mask), _stream(self.device)), "mrs_pid_reset")

    def step(self, actions, action_type, obs_out=None, adj_out=None, comm_range=float("nan"), dense_out=None):
        """One fused step.  actions: (E,N,adim) float32 device tensor (or None with action_type None).
        dense_out (E,N,N) float32: the adjacency also as the matrices the reference returns (needs adj_out)."""
        at = action_type if isinstance(action_type, int) else ACT.get(action_type, -1)
        if actions is None:
            at = 0
        elif at > 0:
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
            if actions.numel() != self.T * ACT_DIM[at]:
                raise ValueError("actions has %d elements, expected (E,N,%d)" % (actions.numel(), ACT_DIM[at]))
        self._out(obs_out, "obs_out", (self.E, self.N, self.D), torch.float32)
        self._out(adj_out, "adj_out", (self.E, self.N, self.W), torch.int64)
        self._out(dense_out, "dense_out", (self.E, self.N, self.N), torch.float32)
        if dense_out is not None and adj_out is None:
            raise ValueError("dense_out needs adj_out (the packed rows) as well")
        b = self._buffers(obs_out, adj_out, dense_out)
        self.version += 1
        cr = float(comm_range) if adj_out is not None else float("nan")
        rc = self.L.mrs_step(self.h, C.byref(b), _ptr(actions), at, self.obs_codes, self.n_obs if obs_out is not None else 0,
                             cr, _stream(self.device))
        if rc == -2:
            raise AttributeError("'Quadcopter' object has no attribute %r" % (action_type,))   # Environment.py:92
        _check(rc, "mrs_step")

    def step_ptr(self, actions, at, obs_ptr, adj_ptr, comm_range, dense_ptr=0):
        """Per-step fast path for MRS.step: `actions` is a contiguous float32 device tensor (or None),
        `at` the integer ACTION_TYPE, obs_ptr / adj_ptr / dense_ptr raw device addresses (0 = skip).  The MrsBuffers
        struct is persistent (the state tensors are never reallocated); only three fields change per step."""
        b = self._pb
        b.obs = obs_ptr or None
        b.adj = adj_ptr or None
        b.adj_dense = dense_ptr or None
        self.version += 1
        rc = self.L.mrs_step(self.h, self._pb_ref, actions.data_ptr() if actions is not None else None, at, self.obs_codes,
                             self.n_obs if obs_ptr else 0, comm_range if adj_ptr else _NAN,
                             _stream_handle(self._dev_index, self.device))
        if rc:
            _check(rc, "mrs_step")

    def step_n(self, actions, action_type, n_substeps, obs_out=None, adj_out=None, comm_range=float("nan")):
        """n_substeps steps in one launch (mrs_step_n).  actions: (S,E,N,adim) for per-substep actions or (E,N,adim) held for
        all substeps; obs_out (S,E,N,D) / adj_out (S,E,N,W): one slice per substep, substep s at index s."""
        at = action_type if isinstance(action_type, int) else ACT.get(action_type, -1)
        S = int(n_substeps)
        stride = 0
        if actions is None:
            at = 0
        elif at > 0:
            if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
                actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
            per = self.T * ACT_DIM[at]
            if actions.numel() == S * per and S > 1:
                stride = per
            elif actions.numel() != per:
                raise ValueError("actions has %d elements, expected (E,N,%d) or (%d,E,N,%d)" % (actions.numel(), ACT_DIM[at], S, ACT_DIM[at]))
        if S < 1:
            raise ValueError("n_substeps must be >= 1")
        self._out(obs_out, "obs_out", (S, self.E, self.N, self.D), torch.float32)
        self._out(adj_out, "adj_out", (S, self.E, self.N, self.W), torch.int64)
        b = self._buffers(obs_out, adj_out)
        self.version += 1
        cr = float(comm_range) if adj_out is not None else float("nan")
        rc = self.L.mrs_step_n(self.h, C.byref(b), _ptr(actions), at, S, stride, self.obs_codes, self.n_obs if obs_out is not None else 0, cr,
                               self.T * self.D if obs_out is not None else 0, self.T * self.W if adj_out is not None else 0, _stream(self.device))
        if rc == -2:
            raise AttributeError("'Quadcopter' object has no attribute %r" % (action_type,))
        _check(rc, "mrs_step_n")

    def step_n_ptr(self, actions, at, n, act_stride, obs_ptr, obs_stride, adj_ptr, adj_stride, comm_range):
        """mrs_step_n with raw slot pointers (MRS.step_n): strides in elements, may be negative."""
        b = self._pb
        b.obs = obs_ptr or None
        b.adj = adj_ptr or None
        b.adj_dense = None
        self.version += 1
        rc = self.L.mrs_step_n(self.h, self._pb_ref, actions.data_ptr() if actions is not None else None, at, n, act_stride, self.obs_codes,
                               self.n_obs if obs_ptr else 0, comm_range if adj_ptr else _NAN, obs_stride, adj_stride,
                               torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _check(rc, "mrs_step_n")

    def observe(self, obs_out, fields=None):
        """Newest observation slice of the current state; `fields` overrides the shard's fused spec for this call only."""
        codes, n, D = self.obs_codes, self.n_obs, self.D
        if fields is not None:
            lst = [OBS[f] if isinstance(f, str) else int(f) for f in fields]
            codes, n, D = (C.c_int32 * max(1, len(lst)))(*lst), len(lst), sum(OBS_WIDTH[c] for c in lst)
        if obs_out is None:
            raise ValueError("obs_out is required")
        self._out(obs_out, "obs_out", (self.E, self.N, D), torch.float32)
        b = self._buffers(obs_out, None)
        _check(self.L.mrs_observe(self.h, C.byref(b), codes, n, _stream(self.device)), "mrs_observe")

    def adjacency(self, adj_out, comm_range, dense_out=None):
        if adj_out is None:
            raise ValueError("adj_out is required")
        self._out(adj_out, "adj_out", (self.E, self.N, self.W), torch.int64)
        self._out(dense_out, "dense_out", (self.E, self.N, self.N), torch.float32)
        b = self._buffers(None, adj_out, dense_out)
        _check(self.L.mrs_adjacency(self.h, C.byref(b), float(comm_range), _stream(self.device)), "mrs_adjacency")

    def adjacency_expand(self, packed, dense_out):
        n = packed.numel() // (self.N * self.W)
        if packed.device != self.device or packed.dtype != torch.int64 or not packed.is_contiguous() or n * self.N * self.W != packed.numel():
            raise ValueError("packed must be a contiguous int64 tensor (M, N=%d, W=%d) on %s" % (self.N, self.W, self.device))
        self._out(dense_out, "dense_out", (n, self.N, self.N), torch.float32)
        if dense_out is None:
            raise ValueError("dense_out is required")
        _check(self.L.mrs_adjacency_expand(self.h, _ptr(packed), _ptr(dense_out), n, _stream(self.device)),
               "mrs_adjacency_expand")

    def spawn(self, seed, env_index_base=0, agent_radius=0.3, ori_lo=(0., 0., -math.pi / 2), ori_hi=(0., 0., math.pi / 2),
              max_rounds=10000, env_mask=None):
        lo = (C.c_float * 3)(*[float(x) for x in ori_lo])
        hi = (C.c_float * 3)(*[float(x) for x in ori_hi])
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
        b = self._buffers()
        self.version += 1
        _check(self.L.mrs_spawn(self.h, C.byref(b), int(seed) & (2 ** 64 - 1), int(env_index_base), float(agent_radius),
                                lo, hi, int(max_rounds), _ptr(mask), _stream(self.device)), "mrs_spawn")

    def spawn_from(self, candidates, agent_radius=0.3, env_mask=None, resume=False):
        """mrs_spawn_from: candidates (E, R, N, 3) float32 device tensor; positions only (see include/mrs_hip.h)."""
        c = candidates.to(device=self.device, dtype=torch.float32).contiguous()
        if c.dim() != 4 or c.shape[0] != self.E or c.shape[2] != self.N or c.shape[3] != 3:
            raise ValueError("candidates must be (E=%d, rounds, N=%d, 3), got %s" % (self.E, self.N, tuple(c.shape)))
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.uint8).contiguous()
        b = self._buffers()
        self.version += 1
        _check(self.L.mrs_spawn_from(self.h, C.byref(b), _ptr(c), int(c.shape[1]), int(bool(resume)), float(agent_radius), _ptr(mask),
                                     _stream(self.device)), "mrs_spawn_from")

    def reynolds(self, x_prev, actions_out=None):
        """Reynolds expert (mrs_reynolds): x_prev (E,N,D>=6) float32 device tensor -> (E,N,3) target velocities."""
        if x_prev.dtype != torch.float32 or not x_prev.is_contiguous() or x_prev.device != self.device:
            x_prev = x_prev.to(device=self.device, dtype=torch.float32).contiguous()
        D = x_prev.numel() // (self.E * self.N)
        if D * self.E * self.N != x_prev.numel():
            raise ValueError("x_prev has %d elements, expected (E=%d, N=%d, D)" % (x_prev.numel(), self.E, self.N))
        if actions_out is None:
            actions_out = torch.empty(self.E, self.N, 3, dtype=torch.float32, device=self.device)
        _check(self.L.mrs_reynolds(self.h, _ptr(x_prev), D, _ptr(actions_out), _stream(self.device)), "mrs_reynolds")
        return actions_out

    # ------------------------------------------------------------------ geometry sensors (Object.py:100-174)
    def raycast(self, offset, directions, body=True, RANGE=100.0):
        """Object.raycast for every quadcopter of every env: offset (3,) or (R,3), directions (3,) or (R,3).
        Returns dict of device tensors: object (E,N,R) int32 [-1 miss, j < N quadcopter, N ground], "pos world" /
        "pos" (E,N,R,3), dist (E,N,R)."""
        dev = self.device
        d = torch.as_tensor(directions, dtype=torch.float32, device=dev).reshape(-1, 3).contiguous()
        o = torch.as_tensor(offset, dtype=torch.float32, device=dev).reshape(-1, 3)
        o = o.expand(d.shape[0], 3).contiguous()
        R = d.shape[0]
        hit = torch.empty(self.E, self.N, R, dtype=torch.int32, device=dev)
        pw = torch.empty(self.E, self.N, R, 3, dtype=torch.float32, device=dev)
        pb = torch.empty_like(pw)
        dist = torch.empty(self.E, self.N, R, dtype=torch.float32, device=dev)
        b = self._buffers()
        _check(self.L.mrs_raycast(self.h, C.byref(b), _ptr(o), _ptr(d), R, int(bool(body)), float(RANGE), _ptr(hit), _ptr(pw),
                                  _ptr(pb), _ptr(dist), _stream(dev)), "mrs_raycast")
        return {"object": hit, "pos world": pw, "pos": pb, "dist": dist}

    def proximity(self, max_dist=float("inf"), points=False):
        """Object.get_dist for every ordered pair: (E,N,N+1) float32 distances (column N = ground; +inf beyond max_dist),
        plus the closest points (E,N,N+1,3) x2 if asked."""
        dev = self.device
        dist = torch.empty(self.E, self.N, self.N + 1, dtype=torch.float32, device=dev)
        ps = torch.empty(self.E, self.N, self.N + 1, 3, dtype=torch.float32, device=dev) if points else None
        po = torch.empty_like(ps) if points else None
        b = self._buffers()
        _check(self.L.mrs_proximity(self.h, C.byref(b), float(max_dist), _ptr(dist), _ptr(ps), _ptr(po), _stream(dev)), "mrs_proximity")
        return (dist, ps, po) if points else dist

    # ------------------------------------------------------------------ views (E,N,k), zero-copy
    def view(self, t):
        k = t.shape[0]
        return t.view(k, self.E, self.N).permute(1, 2, 0)

    def state_dict(self):
        return {k: getattr(self, k).clone() for k in ("pos", "quat", "vel", "angvel", "pid")}

    def load_state_dict(self, sd):
        self.version += 1
        for k in ("pos", "quat", "vel", "angvel", "pid"):
            getattr(self, k).copy_(sd[k])
