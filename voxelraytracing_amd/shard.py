"""Multi-GPU frame sharding: tile-interleaved ownership + one gather of packed tile buffers to rank 0.

No reference counterpart (the reference drives one wgpu device).  Layout: 8x8 screen tile t belongs to rank
t % N and is the (t // N)-th tile in that rank's compact tile-major buffer; every rank's message is one
packed {rgb f32[slots][3], ids u32[slots]} block of slots = ceil(T/N)*64 pixel slots (16 B each), so the
gather is one equal-sized message per rank sent straight to rank 0 over its own xGMI link.  The scene is
read-only and replicated: every rank builds/uploads it, there is no data-path collective besides the gather.

torch / torch.distributed are plumbing only (device memory for the message, the RCCL call).
"""
from __future__ import annotations

import numpy as np


def tiles_of_rank(width: int, height: int, rank: int, count: int):
    """(tile indices owned by `rank`, padded per-rank tile count, total tiles)."""
    total = (width // 8) * (height // 8)
    return np.arange(rank, total, count), -(-total // count), total


def pack_layout(tiles_padded: int):
    """Byte layout of one rank's message: (slots, rgb_bytes, ids_bytes, total_bytes)."""
    slots = tiles_padded * 64
    return slots, slots * 12, slots * 4, slots * 16


def assemble_numpy(msgs, width: int, height: int, count: int):
    """Host twin of vrt_assemble for CPU (gloo) tests: msgs[r] is rank r's packed message as uint8."""
    tiles_x = width // 8
    rgb = np.zeros((height, width, 3), dtype=np.float32)
    ids = np.zeros((height, width), dtype=np.uint32)
    for r in range(count):
        tl, padded, _ = tiles_of_rank(width, height, r, count)
        slots, rgb_b, ids_b, _ = pack_layout(padded)
        m = np.asarray(msgs[r], dtype=np.uint8)
        m_rgb = m[:rgb_b].view(np.float32).reshape(padded, 8, 8, 3)
        m_ids = m[rgb_b:rgb_b + ids_b].view(np.uint32).reshape(padded, 8, 8)
        for k, t in enumerate(tl):
            ty, tx = divmod(int(t), tiles_x)
            rgb[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = m_rgb[k]
            ids[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = m_ids[k]
    return rgb, ids


def pack_tiles_numpy(rgb: np.ndarray, ids: np.ndarray, rank: int, count: int) -> np.ndarray:
    """Inverse of assemble_numpy for one rank: cut the rank's tiles out of full frames into a packed message."""
    h, w = ids.shape
    tl, padded, _ = tiles_of_rank(w, h, rank, count)
    slots, rgb_b, ids_b, total_b = pack_layout(padded)
    m = np.zeros(total_b, dtype=np.uint8)
    m_rgb = m[:rgb_b].view(np.float32).reshape(padded, 8, 8, 3)
    m_ids = m[rgb_b:rgb_b + ids_b].view(np.uint32).reshape(padded, 8, 8)
    tiles_x = w // 8
    for k, t in enumerate(tl):
        ty, tx = divmod(int(t), tiles_x)
        m_rgb[k] = rgb[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
        m_ids[k] = ids[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
    return m


class FrameGather:
    """One process per GPU: owns the packed message tensor, binds the backend's output into it and gathers.

    `dist` is torch.distributed with an initialised process group (nccl = RCCL on the GPU box, gloo on CPU)."""

    def __init__(self, torch, dist, rank: int, count: int, width: int, height: int, device):
        self.torch, self.dist, self.rank, self.count = torch, dist, rank, count
        self.width, self.height = width, height
        _, self.tiles_padded, _ = tiles_of_rank(width, height, rank, count)
        self.slots, self.rgb_bytes, self.ids_bytes, self.msg_bytes = pack_layout(self.tiles_padded)
        self.msg = torch.zeros(self.msg_bytes, dtype=torch.uint8, device=device)
        self.gathered = torch.zeros((count, self.msg_bytes), dtype=torch.uint8, device=device) if rank == 0 else None
        if rank == 0:
            self.frame_rgb = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
            self.frame_ids = torch.zeros((height, width), dtype=torch.int32, device=device)

    def bind(self, gpu):
        """Make the backend render straight into the message tensor (rgb block, then ids block)."""
        base = self.msg.data_ptr()
        gpu.bind_output(base, base + self.rgb_bytes)

    def gather(self):
        """One gather of equal-sized messages to rank 0 (RCCL: N-1 direct sends to the root)."""
        if self.count == 1:
            if self.rank == 0:
                self.gathered[0].copy_(self.msg)
            return
        self.dist.gather(self.msg, list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0)

    def assemble(self, gpu):
        """Rank 0: scatter the gathered tile buffers into the row-major frame on the device."""
        assert self.rank == 0
        base = self.gathered.data_ptr()
        gpu.assemble(base, base + self.rgb_bytes, self.frame_rgb.data_ptr(), self.frame_ids.data_ptr(), self.msg_bytes)
