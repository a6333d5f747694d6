"""K_HOPS history stack of the reference (the deques of MRS.py:87-114) as a device ring.

The reference keeps `deque`s of the last K+1 observation / adjacency tensors, newest first, and
re-stacks them every step.  Here the K+1 newest slices always occupy CONSECUTIVE slots
[head, head+K] of one buffer of L >> K+1 slots, so the stacked result is a zero-copy view and the
step kernel writes the newest slice straight into slot `head`.  The window slides towards slot 0;
when it gets there the K surviving slices are moved to the far end (once every L-K steps).

Padding rules pinned by tests/golden/F4: X is padded with copies of the first slice (MRS.py:92-93),
A with zeros (MRS.py:107-108).

Everything per-step is plain Python integer work: slot pointers are base + slot*stride and the
window views are built once per head position and cached (a torch view op costs microseconds, the
step kernel tens of them).
"""
import torch


class HistoryRing:
    def __init__(self, k_hops, slice_shape, dtype, device, slots=0, pad="copy", view_fn=None):
        self.K = int(k_hops)
        if not slots:
            # default length 16 windows: the K surviving slices move once per L-K steps, a device copy queued between two
            # step kernels (8 windows cost the bench loop 0.8 us per step with the copy staged through a clone, 16 without
            # the clone 0.2 us; much longer rings -- 1 GiB was tried -- stalled the first process on a fresh box for ~50 ms
            # at a wrap inside the timed region, twice out of twice)
            slots = 16 * (self.K + 1)
        self.L = max(int(slots), 2 * (self.K + 1))
        self.pad = pad
        self.buf = torch.zeros((self.L,) + tuple(slice_shape), dtype=dtype, device=device)
        self._base = self.buf.data_ptr()
        self._stride = self.buf[0].numel() * self.buf.element_size()
        self._view_fn = view_fn or (lambda w: w)
        self._views = {}
        self.clear()

    def clear(self):
        """deque([]) (MRS.py:185-186)."""
        self.head = self.L - (self.K + 1)
        self.count = 0
        self.wraps = getattr(self, "wraps", 0) + 2      # every window handed out so far is gone (see intact())
        if self.pad == "zero":
            self.buf[self.head:].zero_()

    def next_slot(self):
        """Slot the newest slice must be written to (appendleft + pop, MRS.py:89-91 / :104-106)."""
        if self.count > 0:
            if self.head == 0:
                new_head = self.L - (self.K + 1)
                if self.K > 0:
                    self.buf[new_head + 1:new_head + 1 + self.K].copy_(self.buf[0:self.K])   # L >= 2(K+1): no overlap
                self.head = new_head
                self.wraps += 1
            else:
                self.head -= 1
        return self.head

    def intact(self, head, wraps):
        """Do the slots [head, head + K] still hold what they held when the window stood there (at wrap count `wraps`)?  New slices
        are written below the head, so nothing up there changes until the head wraps; after ONE wrap the copied slices sit in the
        top K slots and the new head descends from L - K - 1: intact while it has not reached the old window."""
        return self.wraps == wraps or (self.wraps == wraps + 1 and head + self.K < self.head)

    def ptr(self, slot):
        return self._base + slot * self._stride

    def committed(self):
        """Call after the newest slice has been written (possibly asynchronously, same stream)."""
        if self.count == 0 and self.K > 0:
            if self.pad == "copy":
                self.buf[self.head + 1:self.head + 1 + self.K] = self.buf[self.head].unsqueeze(0)
            else:
                self.buf[self.head + 1:self.head + 1 + self.K].zero_()
        if self.count <= self.K:
            self.count += 1

    def newest(self):
        return self.buf[self.head]

    def window(self):
        """(K+1, ...) newest first, a view."""
        return self.buf[self.head:self.head + self.K + 1]

    def view(self):
        """view_fn(window()), cached per head position."""
        v = self._views.get(self.head)
        if v is None:
            v = self._views[self.head] = self._view_fn(self.window())
        return v
