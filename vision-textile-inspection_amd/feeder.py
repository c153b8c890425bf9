"""Host -> HBM frame feed for the predict path (device half).

In the reference a frame arrives in pageable host memory (`ret, frame = cap.read()`, main.py:188) and goes through
`cv2.cvtColor` (measurement.py:205) into `model.predict` (measurement.py:208-210), which copies it to the device synchronously.
At the rates this engine runs (1.2 MB per 640x640 frame, tens of thousands of frames/s) that copy is the boundary's real cost,
so the feed is a ring of PINNED host staging buffers and device buffers:

    slot k:  host fills pinned[k]  ->  async H2D on a copy stream  ->  event  ->  compute stream runs vti_predict on dev[k]

While the compute stream works on slot k, the copy stream moves slot k+1 and the host fills slot k+2.  Ordering is by events only
(no host synchronisation except when the host wants to REUSE a pinned slot whose copy has not finished):
    h2d_done[k]   recorded on the copy stream after the copy   -- the compute stream waits for it before reading dev[k];
                                                                  the host waits for it before rewriting pinned[k]
    consumed[k]   recorded on the compute stream after predict -- the copy stream waits for it before overwriting dev[k]

PyTorch is plumbing here (pinned / device allocations, streams, events); the arithmetic is libvti.so's.
"""
import numpy as np
import torch


class FrameFeeder:
    """Ring of `depth` (pinned host, device) uint8 frame buffers [B, H0, W0, 3] in front of Engine.predict_into / YOLO."""

    def __init__(self, B, H0, W0, depth=3, device=0):
        if depth < 2:
            raise ValueError("FrameFeeder needs at least two slots (one in flight, one being filled)")
        if not torch.cuda.is_available():
            raise RuntimeError("FrameFeeder needs a ROCm GPU (no CPU fallback)")
        self.B, self.H0, self.W0, self.depth = B, H0, W0, depth
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        shape = (B, H0, W0, 3)
        self.pinned = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.h2d_done = [None] * depth
        self.consumed = [None] * depth
        self.count = [0] * depth            # frames valid in the slot
        self._next = 0

    # ---- host side -------------------------------------------------------------------------
    def host_view(self, slot):
        """numpy view [B, H0, W0, 3] of the slot's PINNED staging buffer: grab camera frames straight into it
        (`cap.read(image=view[i])`), then submit(slot).  Blocks only if the slot's previous copy is still in flight."""
        ev = self.h2d_done[slot]
        if ev is not None:
            ev.synchronize()
        return self.pinned[slot].numpy()

    def next_slot(self):
        s = self._next
        self._next = (s + 1) % self.depth
        return s

    def submit(self, slot, n=None):
        """Start the async H2D copy of the slot's first n frames (default: all B) on the copy stream."""
        n = self.B if n is None else int(n)
        if not 0 < n <= self.B:
            raise ValueError("submit: n must be in 1..B")
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])     # the previous predict on dev[slot] has finished reading it
            self.dev[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.h2d_done[slot] = ev
        self.count[slot] = n
        return slot

    def put(self, frames):
        """Convenience: copy a host batch (ndarray / CPU tensor, uint8 [n<=B, H0, W0, 3]) into the next slot's pinned buffer and
        submit it.  (A producer that can write into host_view() directly saves this host-side memcpy.)"""
        slot = self.next_slot()
        view = self.host_view(slot)
        a = frames.numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
        if a.dtype != np.uint8 or a.ndim != 4 or a.shape[1:] != (self.H0, self.W0, 3) or a.shape[0] > self.B:
            raise ValueError(f"put: expected uint8 [<= {self.B}, {self.H0}, {self.W0}, 3], got {a.dtype} {a.shape}")
        view[:a.shape[0]] = a
        return self.submit(slot, a.shape[0])

    # ---- device side -----------------------------------------------------------------------
    def frames(self, slot):
        """Device frames of a submitted slot, ordered after its copy on the CURRENT stream."""
        ev = self.h2d_done[slot]
        if ev is None:
            raise RuntimeError("slot was never submitted")
        torch.cuda.current_stream().wait_event(ev)
        return self.dev[slot][:self.count[slot]]

    def release(self, slot):
        """Call after the last kernel that reads frames(slot) was enqueued on the current stream: the slot's device buffer may be
        overwritten by a later submit once those kernels have run."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.consumed[slot] = ev

    def predict_into(self, engine, slot, out, **kw):
        """engine.predict_into on a submitted slot (event-chained, no host synchronisation)."""
        x = self.frames(slot)
        engine.predict_into(x, out, **kw)
        self.release(slot)
        return out
