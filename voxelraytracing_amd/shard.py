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
    """One process per GPU: owns the message tensor, binds the backend's output to it and gathers.

    `dist` is torch.distributed with an initialised process group (nccl = RCCL on the GPU box, gloo on CPU)."""

    def __init__(self, torch, dist, rank: int, count: int, width: int, height: int, device):
        self.torch, self.dist, self.rank, self.count = torch, dist, rank, count
        self.width, self.height = width, height
        _, self.tiles_padded, _ = tiles_of_rank(width, height, rank, count)
        self.slots = self.tiles_padded * 64
        self.msg = torch.zeros(self.slots * 4, dtype=torch.int32, device=device)
        self.gathered = torch.zeros((count, self.slots * 4), dtype=torch.int32, device=device) if rank == 0 else None
        self.frame = torch.zeros((height, width, 4), dtype=torch.int32, device=device) if rank == 0 else None

    def bind(self, gpu):
        """Make the backend render straight into the message tensor."""
        gpu.bind_output(self.msg.data_ptr())

    def gather(self):
        """One gather of equal-sized messages to rank 0 (RCCL: N-1 direct sends to the root)."""
        self.dist.gather(self.msg, list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0)

    def assemble(self, gpu):
        """Rank 0: scatter the gathered tile buffers into the row-major texel frame on the device."""
        gpu.assemble(self.gathered.data_ptr(), self.frame.data_ptr(), self.slots * TEXEL_BYTES)
