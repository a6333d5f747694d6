"""Multi-GPU: env instances shard embarrassingly (no cross-env term anywhere: downwash, adjacency
and contact are intra-env -- Quadcopter.py:101, MRS.py:120), one process per GPU, contiguous blocks of
E/G envs by GLOBAL env index.  Nothing is exchanged inside step(); the only collective is an
all-gather of the newest observation slice (E_local, N, D) float32 so that every rank can see the
joint observation.  The reference has no distributed code at all.

The gather runs on a side stream and is double-buffered: step t+1's kernel overlaps gather t.
Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

Two forms of the exchange (MRS_DIST_ALLGATHER = "collective" | "direct", or ObsAllGather(mode=...)):
  collective  one all_gather_into_tensor; ring or direct is RCCL's choice (the default: the form every round has run);
  direct      SURVEY.md section 5's one-shot form spelled out -- every rank sends its slice to each of its peers and receives
              theirs straight into place, one grouped batch of point-to-point operations (RCCL: one ncclGroup, all seven xGMI
              links of a GPU driven at once, no hop through a neighbour; no zero-padding for unequal shards either).
A consumer that needs the joint tensor only every k-th step says so (every=k): the steps in between exchange nothing.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_envs_total, rank, world):
    """Contiguous env block of `rank`: [lo, hi) in global env indices (remainder to the low ranks)."""
    base, rem = divmod(int(n_envs_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _forced():
    """MRS_DIST_FORCE=1: take the collective path with ONE rank too -- the rehearsal of the RCCL leg on a one-GPU box
    (communicator set-up, all_gather_into_tensor on the side stream, the event hand-shake): tests/test_gpu_dist.py."""
    return os.environ.get("MRS_DIST_FORCE") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MRS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class ObsAllGather:
    """all-gather of the ranks' (E_local, N, D) slices into (sum of E_local, N, D) in rank order, overlapped with compute.
    The shards may differ in size (E not divisible by the world size)."""

    def __init__(self, e_local, n_agents, d, device, group=None, buffers=2, mode=None, every=1):
        self.group = group
        self.mode = mode or os.environ.get("MRS_DIST_ALLGATHER", "collective")
        if self.mode not in ("collective", "direct"):
            raise ValueError("ObsAllGather mode must be 'collective' or 'direct', got %r" % (self.mode,))
        self.every = max(1, int(every))
        self.calls = 0          # gather() calls so far: the exchange runs on calls every-1, 2*every-1, ...
        self.latest = None      # the buffer of the last exchange (what a call in between returns)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.e_local = int(e_local)
        self.collective = dist.is_initialized() and (self.world > 1 or _forced())
        # shard sizes of every rank (shard_range hands the remainder of E % world to the low ranks): exchanged once here
        if self.collective:
            mine = torch.tensor([self.e_local], dtype=torch.int64, device=self.device)
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine, group=group)
            self.sizes = [int(t.item()) for t in every]
        else:
            self.sizes = [self.e_local]
        self.offsets = [sum(self.sizes[:r]) for r in range(self.world)]
        self.e_max = max(self.sizes)
        self.uneven = min(self.sizes) != self.e_max and self.mode == "collective"   # the direct form sends exact slices
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.mode == "direct" and self.collective and self.world > 1 and torch.device(device).type == "cuda" \
                and dist.get_backend(group) != "nccl":
            raise RuntimeError("ObsAllGather(mode='direct') on GPU tensors needs the nccl (RCCL) backend; '%s' carries "
                               "point-to-point operations for host tensors only" % dist.get_backend(group))
        self.out = [torch.empty(sum(self.sizes), n_agents, d, dtype=torch.float32, device=self.device) for _ in range(buffers)]
        # unequal shards: the collective needs equal contributions, so every rank sends e_max envs (its slice, zero-padded)
        # and the valid parts are compacted into `out` behind the collective, on the same side stream
        self.stage = [torch.zeros(self.e_max, n_agents, d, dtype=torch.float32, device=self.device) for _ in range(buffers)] if self.uneven else None
        self.padded = [torch.empty(self.world * self.e_max, n_agents, d, dtype=torch.float32, device=self.device) for _ in range(buffers)] if self.uneven else None
        self.k = 0
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.done = [None] * buffers

    def _compact(self, k):
        for r in range(self.world):
            self.out[k][self.offsets[r]:self.offsets[r] + self.sizes[r]].copy_(self.padded[k][r * self.e_max:r * self.e_max + self.sizes[r]])

    def _p2p(self, out, src):
        """The one-shot form: this rank's slice to every peer, every peer's slice into its place in `out` -- one grouped batch."""
        me = self.rank
        out[self.offsets[me]:self.offsets[me] + self.sizes[me]].copy_(src)
        # P2POp's `peer` is a GLOBAL rank; r indexes sizes / offsets by the rank inside self.group (with a sub-group
        # the sends and receives would otherwise address the wrong processes)
        peer = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
        ops = []
        for r in range(self.world):
            if r == me or self.sizes[r] == 0:
                continue
            ops.append(dist.P2POp(dist.irecv, out[self.offsets[r]:self.offsets[r] + self.sizes[r]], peer(r), group=self.group))
        if self.sizes[me] > 0:
            for r in range(self.world):
                if r != me:
                    ops.append(dist.P2POp(dist.isend, src, peer(r), group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def gather(self, newest):
        """Start gathering `newest` (contiguous (E_local,N,D)); returns the output buffer, complete after wait().
        With every = k > 1 only every k-th call exchanges anything; the calls in between return the buffer of the last
        exchange (None before the first one) and cost nothing.

        Ordering (GPU): the collective runs on a side stream behind an event recorded on the caller's stream NOW, so
        it starts after the step kernel that wrote `newest` AND after everything the caller has enqueued so far --
        including its reads of the output buffer this call is about to reuse.  Back-pressure: the caller's stream
        first waits for the previous gather that used this buffer, so at most `buffers` gathers are ever in flight;
        `newest` is a history-ring slot that the step kernel overwrites HISTORY_SLOTS - K steps later, so `buffers`
        must stay below that (2 against >= K + 1 by construction of HistoryRing).  A consumer must call wait() (or
        use its own wait_event on done[k]) before reading the returned buffer."""
        self.calls += 1
        if self.calls % self.every:
            return self.latest
        k = self.k
        self.k = (k + 1) % len(self.out)
        out = self.latest = self.out[k]
        if not self.collective:
            out.copy_(newest)
            return out
        direct = self.mode == "direct"
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            if self.done[k] is not None:
                cur.wait_event(self.done[k])                             # back-pressure: gather t - buffers has landed
            src = newest
            if self.uneven:
                self.stage[k][:self.e_local].copy_(newest)               # on the caller's stream, behind the step kernel
                src = self.stage[k]
            ready = torch.cuda.Event()
            ready.record(cur)                                            # the step kernel that wrote `newest`
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                if direct:
                    for w in self._p2p(out, src):
                        w.wait()                                         # RCCL: stream-ordered on the side stream, not a host wait
                else:
                    dist.all_gather_into_tensor(self.padded[k] if self.uneven else out, src, group=self.group)
                if self.uneven:
                    self._compact(k)
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.done[k] = ev
        elif direct:
            for w in self._p2p(out, newest.contiguous()):
                w.wait()
        elif self.uneven:
            self.stage[k][:self.e_local].copy_(newest)
            dist.all_gather(list(self.padded[k].chunk(self.world, dim=0)), self.stage[k], group=self.group)
            self._compact(k)
        else:
            parts = list(out.chunk(self.world, dim=0))
            dist.all_gather(parts, newest.contiguous(), group=self.group)
        return out

    def wait(self):
        """Make the compute stream wait for every gather issued so far."""
        if self.cuda:
            for ev in self.done:
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)


def gather_global_state(local, group=None):
    """Concatenate per-rank (E_local, ...) tensors in rank order (tests: 1-GPU vs sharded bitwise equality)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _forced()):
        return local
    world = dist.get_world_size(group)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    sizes = [int(t.item()) for t in every]
    pad = torch.zeros((max(sizes),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)   # shards may differ by one env
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
