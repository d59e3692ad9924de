"""Multi-GPU frame sharding: tile-interleaved ownership + one gather of tile buffers to rank 0.

No reference counterpart (the reference drives one wgpu device).  Layout: 8x8 screen tile t belongs to rank
t % N and is the (t // N)-th tile in that rank's compact tile-major buffer of 16-byte texels {r,g,b,id};
every rank's message is texel[ceil(T/N)*64], so the gather is one equal-sized message per rank sent
straight to rank 0 over its own xGMI link.  The scene is read-only and replicated: every rank builds and
uploads it; there is no data-path collective besides the gather.

torch / torch.distributed are plumbing only (device memory for the message, the RCCL call).
"""
from __future__ import annotations

import numpy as np

TEXEL_BYTES = 16


def tiles_of_rank(width: int, height: int, rank: int, count: int):
    """(tile indices owned by `rank`, padded per-rank tile count, total tiles)."""
    total = (width // 8) * (height // 8)
    return np.arange(rank, total, count), -(-total // count), total


def texels_to_frame(texels: np.ndarray):
    """uint32[..., 4] texels -> (rgb f32[..., 3], ids u32[...])."""
    t = np.ascontiguousarray(texels, dtype=np.uint32)
    return t[..., :3].copy().view(np.float32), t[..., 3].copy()


def assemble_numpy(msgs, width: int, height: int, count: int) -> np.ndarray:
    """Host twin of vrt_assemble for CPU (gloo) tests: msgs[r] = rank r's message as uint32[slots*4]."""
    tiles_x = width // 8
    frame = np.zeros((height, width, 4), dtype=np.uint32)
    for r in range(count):
        tl, padded, _ = tiles_of_rank(width, height, r, count)
        m = np.asarray(msgs[r], dtype=np.uint32).reshape(padded, 8, 8, 4)
        for k, t in enumerate(tl):
            ty, tx = divmod(int(t), tiles_x)
            frame[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = m[k]
    return frame


def pack_tiles_numpy(frame: np.ndarray, rank: int, count: int) -> np.ndarray:
    """Inverse of assemble_numpy for one rank: cut the rank's tiles out of a texel frame uint32[h][w][4]."""
    h, w, _ = frame.shape
    tl, padded, _ = tiles_of_rank(w, h, rank, count)
    m = np.zeros((padded, 8, 8, 4), dtype=np.uint32)
    tiles_x = w // 8
    for k, t in enumerate(tl):
        ty, tx = divmod(int(t), tiles_x)
        m[k] = frame[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
    return m.reshape(-1)


class FrameGather:
    """One process per GPU: owns the message tensors, binds the backend's output to them and gathers.

    `dist` is torch.distributed with an initialised process group (nccl = RCCL on the GPU box, gloo on CPU).
    Two message / receive buffers ping-pong so that the gather of frame k (on the collective's own stream)
    overlaps the render of frame k+1 and the de-interleave of frame k-1 (`submit` / `drain`)."""

    def __init__(self, torch, dist, rank: int, count: int, width: int, height: int, device):
        self.torch, self.dist, self.rank, self.count = torch, dist, rank, count
        self.width, self.height = width, height
        _, self.tiles_padded, _ = tiles_of_rank(width, height, rank, count)
        self.slots = self.tiles_padded * 64
        self.msgs = [torch.zeros(self.slots * 4, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = [torch.zeros((count, self.slots * 4), dtype=torch.int32, device=device) if rank == 0 else None
                     for _ in range(2)]
        self.frame = torch.zeros((height, width, 4), dtype=torch.int32, device=device) if rank == 0 else None
        self.k = 0            # frames submitted
        self.pending = None   # (work, buffer index) of the gather still in flight
        # single-buffer aliases (tests, simple callers)
        self.msg, self.gathered = self.msgs[0], self.recv[0]

    def bind(self, gpu, which: int = 0):
        """Make the backend render straight into message buffer `which`."""
        gpu.bind_output(self.msgs[which].data_ptr())

    def gather(self, which: int = 0, async_op: bool = False):
        """One gather of equal-sized messages to rank 0 (RCCL: N-1 direct sends to the root)."""
        return self.dist.gather(self.msgs[which], list(self.recv[which].unbind(0)) if self.rank == 0 else None, dst=0,
                                async_op=async_op)

    def assemble(self, gpu, which: int = 0):
        """Rank 0: scatter the gathered tile buffers into the row-major texel frame on the device."""
        gpu.assemble(self.recv[which].data_ptr(), self.frame.data_ptr(), self.slots * TEXEL_BYTES)

    # ---- pipelined frames ----
    def submit(self, gpu, render):
        """Render frame k into message k&1, start its gather, then finish frame k-1 (wait + assemble on rank 0).
        Waiting for gather k-1 here also guarantees message (k+1)&1 is free before the next render overwrites it."""
        w = self.k & 1
        self.bind(gpu, w)
        render()
        work = self.gather(w, async_op=True)
        self._finish_pending(gpu)
        self.pending = (work, w)
        self.k += 1

    def _finish_pending(self, gpu):
        if self.pending is None:
            return
        work, w = self.pending
        work.wait()   # stream-ordered for RCCL (the current stream waits), host-blocking for gloo
        if self.rank == 0:
            self.assemble(gpu, w)
        self.pending = None

    def drain(self, gpu):
        """Finish the last submitted frame."""
        self._finish_pending(gpu)
