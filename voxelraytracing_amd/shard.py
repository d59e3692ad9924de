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
RECORD_BYTES = 8   # VRT_FLAG_COMPACT: {id word | norm.y sign, water_dist} — what the gather root needs to shade the pixel itself


def tiles_of_rank(width: int, height: int, rank: int, count: int, root_weight: int = 1):
    """(tile indices owned by `rank` in increasing order, padded per-rank tile count of ranks >= 1, total tiles).

    Tiles are dealt out in periods of P = root_weight + count - 1: the first root_weight of every period to rank 0,
    then one to each of ranks 1..count-1 (include/vrt.h, vrt_config.shard_root_weight); weight 1 = t % count."""
    total = (width // 8) * (height // 8)
    period = root_weight + count - 1
    t = np.arange(total)
    r = t % period
    mine = t[r < root_weight] if rank == 0 else t[r == root_weight + rank - 1]
    return mine, -(-total // period), total


def texels_to_frame(texels: np.ndarray):
    """uint32[..., 4] texels -> (rgb f32[..., 3], ids u32[...])."""
    t = np.ascontiguousarray(texels, dtype=np.uint32)
    return t[..., :3].copy().view(np.float32), t[..., 3].copy()


def assemble_numpy(msgs, width: int, height: int, count: int, root_weight: int = 1, frame: np.ndarray | None = None) -> np.ndarray:
    """Host twin of vrt_assemble for CPU (gloo) tests: msgs[r] = rank r's message as uint32[slots*4].
    With `frame` given, rank 0's tiles are already in it (the in-place root) and msgs[0] is ignored."""
    tiles_x = width // 8
    in_place = frame is not None
    if frame is None:
        frame = np.zeros((height, width, 4), dtype=np.uint32)
    for r in range(1 if in_place else 0, count):
        tl, _, _ = tiles_of_rank(width, height, r, count, root_weight)
        m = np.asarray(msgs[r], dtype=np.uint32).reshape(-1, 8, 8, 4)
        for k, t in enumerate(tl):
            ty, tx = divmod(int(t), tiles_x)
            frame[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] = m[k]
    return frame


def pack_tiles_numpy(frame: np.ndarray, rank: int, count: int, root_weight: int = 1) -> np.ndarray:
    """Inverse of assemble_numpy for one rank: cut the rank's tiles out of a texel frame uint32[h][w][4]."""
    h, w, _ = frame.shape
    tl, padded, _ = tiles_of_rank(w, h, rank, count, root_weight)
    m = np.zeros((max(padded, len(tl)), 8, 8, 4), dtype=np.uint32)
    tiles_x = w // 8
    for k, t in enumerate(tl):
        ty, tx = divmod(int(t), tiles_x)
        m[k] = frame[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
    return m.reshape(-1)


def root_weight_model(count: int, frame_ms: float, message_bytes_total: float, link_gbs: float = 60.0, assemble_ms: float = 0.02) -> int:
    """First guess for vrt_config.shard_root_weight (bench.py then measures the neighbours): the root's own tiles
    never cross a link, every other rank's tiles all arrive over that rank's one xGMI link, so the root should
    trace w0 of every w0 + N - 1 tiles such that its render + de-interleave time matches one link's transfer time:
        w0/P * frame_ms + assemble_ms  =  (message_bytes_total / P) / link_gbs        (P = w0 + N - 1)."""
    if count <= 1:
        return 1
    link_ms = message_bytes_total / (link_gbs * 1e9) * 1e3   # the whole frame over one link
    best, best_t = 1, None
    for w0 in range(1, 65):
        period = w0 + count - 1
        t = max(w0 / period * frame_ms + assemble_ms, link_ms / period, frame_ms / period)
        if best_t is None or t < best_t - 1e-9:
            best, best_t = w0, t
    return best


def expected_scaling(count: int, frame_ms_1gpu: float, pixels: int, bytes_per_pixel: int, root_weight: int, link_gbs: float = 60.0,
                     assemble_ms_whole_frame: float = 0.02, host_us_per_frame: float = 0.0, fixed_ms_per_launch: float = 0.016) -> dict:
    """What a frame sharded over `count` GPUs should take, term by term — the arithmetic of DESIGN.md section 7, stated before
    the run so that a measured scaling curve can be read against it (bench.py prints it as config.expected_scaling on every
    N > 1 line).  Tiles are dealt in periods of P = w0 + N - 1: w0 to the root (never cross a link), one to each other rank.

    frame_ms_1gpu: the unsharded frame on one GPU (measured in the same run); fixed_ms_per_launch: what a launch costs however
    few tiles it has (an all-sky frame: profiles/r02_fixed_cost.txt) — a share's render time does not fall below it;
    link_gbs: one xGMI link, one direction (60-75 GB/s measured point to point; the guide's 153 GB/s is the link's peak);
    the messages of the N - 1 ranks arrive over N - 1 different links at once.  Returns the terms, the bound that wins and
    the predicted speed-up over one GPU."""
    if count <= 1:
        return {"n": 1, "predicted_ms": frame_ms_1gpu, "speedup": 1.0, "bound": "one GPU"}
    period = root_weight + count - 1
    march = max(frame_ms_1gpu - fixed_ms_per_launch, 0.0)
    root_render = fixed_ms_per_launch + march * root_weight / period
    rank_render = fixed_ms_per_launch + march / period
    message_bytes = pixels * bytes_per_pixel / period                  # one rank's message: its own link
    link_ms = message_bytes / (link_gbs * 1e9) * 1e3
    assemble = assemble_ms_whole_frame * (count - 1) / period          # the root scatters / shades the other ranks' pixels
    root_total = root_render + assemble
    host_ms = host_us_per_frame * 1e-3
    terms = {"root_render_plus_assemble_ms": root_total, "rank_render_ms": rank_render, "link_ms_per_message": link_ms, "host_ms_per_frame": host_ms}
    # pipelined (frames in flight, ping-pong messages): the steady state is the slowest stage, not the sum
    bound = max(terms, key=lambda k: terms[k])
    predicted = terms[bound]
    return {"n": count, "root_weight": root_weight, "period": period, "message_bytes_per_rank": message_bytes, "link_gbs_assumed": link_gbs,
            **terms, "bound": bound, "predicted_ms": predicted, "speedup": frame_ms_1gpu / predicted if predicted > 0 else None,
            "efficiency": (frame_ms_1gpu / predicted / count) if predicted > 0 else None,
            "note": "steady-state frame period = the slowest of: the root's render + assembly, a shard's render, one message over one link, "
                    "the host's per-frame issue cost; DESIGN.md section 7"}


class FrameGather:
    """One process per GPU: owns the message tensors, binds the backend's output to them and gathers.

    `dist` is torch.distributed with an initialised process group (nccl = RCCL on the GPU box, gloo on CPU).
    Two message / receive / frame buffer sets ping-pong so that the gather of batch k (on the collective's own stream)
    overlaps the render of batch k+1 and the de-interleave of batch k-1 (`submit` / `drain`).

    batch: frames per gather.  A rank's share of a 1080p frame is a few tens of microseconds of GPU work, less than
    what one collective call costs the host; `batch` frames are rendered into consecutive slices of one message and
    travel in one gather (fewer, larger collectives), each still assembled into its own frame buffer.

    in_place (needs a root context created with row_major=True): rank 0 renders its own tiles straight into the
    row-major frame and contributes nothing to the gather but an unused message-sized slot; with root_weight > 1
    it also takes a larger share of the tiles than the ranks whose tiles have to cross a link.

    compact (needs in_place; the other ranks' contexts created with compact=True): messages carry 8 bytes per pixel
    instead of the 16-byte texel and the root shades them while de-interleaving (vrt_assemble_compact): half the bytes
    over every link."""

    def __init__(self, torch, dist, rank: int, count: int, width: int, height: int, device, root_weight: int = 1,
                 in_place: bool = False, compact: bool = False, batch: int = 1):
        self.torch, self.dist, self.rank, self.count = torch, dist, rank, count
        self.width, self.height = width, height
        self.root_weight, self.in_place, self.batch = root_weight, in_place, batch
        assert in_place or root_weight == 1, "a weighted root renders in place"
        assert in_place or not compact, "compact messages are shaded by an in-place root"
        assert batch >= 1
        self.compact = compact
        self.slot_bytes = RECORD_BYTES if compact else TEXEL_BYTES
        words = self.slot_bytes // 4
        _, self.tiles_padded, _ = tiles_of_rank(width, height, rank, count, root_weight)
        self.slots = self.tiles_padded * 64
        self.frame_words = self.slots * words          # int32 words of one frame's message
        self.msgs = [torch.zeros(batch * self.frame_words, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = [torch.zeros((count, batch * self.frame_words), dtype=torch.int32, device=device) if rank == 0 else None
                     for _ in range(2)]
        n_sets = 2 if (in_place or batch > 1) else 1
        self.frames = [[torch.zeros((height, width, 4), dtype=torch.int32, device=device) if rank == 0 else None
                        for _ in range(batch)] for _ in range(n_sets)]
        self.frame = self.frames[0][0]   # the last completed frame (rank 0)
        self.k = 0            # batches submitted
        self.pending = None   # (work, buffer set, frames in it) of the gather still in flight
        self.reset_host_profile()
        # single-buffer aliases (tests, simple callers)
        self.msg, self.gathered = self.msgs[0], self.recv[0]

    # ---- what the collective costs THIS rank's host thread (bench.py: config.expected_scaling's host term, measured) ----
    def reset_host_profile(self):
        self._host = {"collectives": 0, "frames": 0, "gather_call_s": 0.0, "wait_s": 0.0, "assemble_s": 0.0}

    def host_profile(self) -> dict:
        """Microseconds of host time per collective: the gather call, the wait for it (stream-ordered for RCCL: enqueuing the wait;
        host-blocking for gloo) and, on rank 0, the assembly launches of its frames — and the frames a collective carried."""
        h, n = self._host, max(self._host["collectives"], 1)
        return {"collectives": h["collectives"], "frames_per_collective": h["frames"] / n, "gather_call_us": h["gather_call_s"] / n * 1e6,
                "wait_us": h["wait_s"] / n * 1e6, "assemble_us": h["assemble_s"] / n * 1e6}

    def _frame_of(self, which: int, j: int = 0):
        return self.frames[which % len(self.frames)][j]

    def bind(self, gpu, which: int = 0, j: int = 0):
        """Make the backend render straight into slice j of message buffer `which` (the in-place root: into frame j of set `which`)."""
        if self.in_place and self.rank == 0:
            gpu.bind_output(self._frame_of(which, j).data_ptr())
        else:
            gpu.bind_output(self.msgs[which].data_ptr() + j * self.frame_words * 4)

    def gather(self, which: int = 0, async_op: bool = False, nframes: int | None = None):
        """One gather of equal-sized messages to rank 0 (RCCL: N-1 direct sends to the root)."""
        import time
        n = (self.batch if nframes is None else nframes) * self.frame_words
        t0 = time.perf_counter()
        work = self.dist.gather(self.msgs[which][:n], [row[:n] for row in self.recv[which].unbind(0)] if self.rank == 0 else None,
                                dst=0, async_op=async_op)
        self._host["gather_call_s"] += time.perf_counter() - t0
        self._host["collectives"] += 1
        self._host["frames"] += self.batch if nframes is None else nframes
        return work

    def assemble(self, gpu, which: int = 0, j: int = 0):
        """Rank 0: scatter frame j of the gathered tile buffers into its row-major texel frame on the device."""
        f = self._frame_of(which, j)
        gpu.assemble(self.recv[which].data_ptr() + j * self.frame_words * 4, f.data_ptr(), self.batch * self.frame_words * 4,
                     **({"compact": True} if self.compact else {}))
        self.frame = f

    # ---- pipelined frames ----
    def submit(self, gpu, render, nframes: int = 1):
        """Render `nframes` (<= batch) frames into message set k&1, start their gather, then finish the batch before
        (wait + assemble on rank 0).  Waiting for that gather here also guarantees message set (k+1)&1 is free before
        the next batch overwrites it."""
        assert 1 <= nframes <= self.batch
        w = self.k & 1
        for j in range(nframes):
            self.bind(gpu, w, j)
            render()
        work = self.gather(w, async_op=True, nframes=nframes)
        self._finish_pending(gpu)
        self.pending = (work, w, nframes)
        self.k += 1

    def _finish_pending(self, gpu):
        if self.pending is None:
            return
        import time
        work, w, nframes = self.pending
        t0 = time.perf_counter()
        work.wait()   # stream-ordered for RCCL (the current stream waits), host-blocking for gloo
        t1 = time.perf_counter()
        if self.rank == 0:
            for j in range(nframes):
                self.assemble(gpu, w, j)
        self._host["wait_s"] += t1 - t0
        self._host["assemble_s"] += time.perf_counter() - t1
        self.pending = None

    def drain(self, gpu):
        """Finish the last submitted batch."""
        self._finish_pending(gpu)


class ReplicatedWorld:
    """Keeps the N replicas of the scene coherent while the world changes.

    The scene is read-only during a frame and replicated on every GPU; between frames the reference's client
    mutates it in three ways — chunks arrive from the server (`GiveChunkData`, client/src/lib.rs:110-118), the grid is
    recentred on the player (`center_chunks`, lib.rs:55-65 / world.rs:297-308) and voxels are edited (`set_voxel`,
    main.rs:352-362).  Rank 0 is the process that talks to the server and reads the input; it broadcasts the *command*
    (the message bytes, the anchor, the edit), every rank applies it to its own host world — the allocators are
    deterministic, so the pools stay byte-identical — and uploads only the range the command touched
    (`vrt_write_nodes`, as the reference does).  `verify()` compares a digest of every replica's pool and chunk table.
    No node data crosses a link except the server's own messages (a few KB per chunk)."""

    CMD_CHUNKS, CMD_EDIT, CMD_RECENTER = 1, 2, 3

    def __init__(self, torch, dist, rank: int, count: int, world, gpu, device="cpu"):
        self.torch, self.dist, self.rank, self.count = torch, dist, rank, count
        self.world, self.gpu, self.device = world, gpu, device
        self.uploaded_nodes = 0   # node words re-uploaded so far (what a whole-pool upload would have cost: max_nodes each time)

    # ---- transport: rank 0's command to everybody ----
    def _bcast(self, kind: int, ints=(), payload: bytes = b""):
        t = self.torch
        if self.count > 1:
            hdr = t.zeros(8, dtype=t.int64, device=self.device)
            if self.rank == 0:
                hdr[0], hdr[1] = kind, len(payload)
                for i, v in enumerate(ints):
                    hdr[2 + i] = int(v)
            self.dist.broadcast(hdr, src=0)
            kind, n = int(hdr[0]), int(hdr[1])
            ints = [int(v) for v in hdr[2:]]
            if n:
                buf = t.frombuffer(bytearray(payload), dtype=t.uint8).to(self.device) if self.rank == 0 else t.empty(n, dtype=t.uint8, device=self.device)
                self.dist.broadcast(buf, src=0)
                payload = bytes(buf.cpu().numpy().tobytes())
        return kind, list(ints), payload

    def _upload(self, start: int, n: int):
        self.gpu.write_nodes(self.world.nodes_ptr(), start, start + n)
        self.uploaded_nodes += n

    def _roots(self):
        self.gpu.write_chunk_roots(self.world.chunk_roots())
        self.gpu.write_world_data(self.world.world_data())

    # ---- the three mutations; arguments are read on rank 0 only ----
    def ingest_chunk_msgs(self, data: bytes = b""):
        """Rank 0 passes the bytes received from the server; every rank creates the chunks and uploads their ranges.
        Returns [(chunk pos, root, node count)] (CmdResult.updated_chunks) and leaves an incomplete tail unconsumed."""
        _, _, data = self._bcast(self.CMD_CHUNKS, (), data)
        updated, off = [], 0
        while off < len(data):
            try:
                got = self.world.ingest_chunk_msg(data[off:])
            except Exception as e:      # PosOutOfBounds = received_oob_chunks: the message is consumed, nothing to upload
                if getattr(e, "kind", "") != "PosOutOfBounds":
                    raise
                off += e.consumed
                continue
            if got is None:
                break
            used, pos, root, n = got
            self._upload(root, n)
            updated.append((pos, root, n))
            off += used
        if updated:
            self._roots()
        return updated, off

    def set_voxel(self, pos=(0, 0, 0), voxel: int = 0):
        """The edit of main.rs:352-362 on every replica; returns the re-uploaded (start, count) or None (NoChange...)."""
        _, v, _ = self._bcast(self.CMD_EDIT, (pos[0], pos[1], pos[2], voxel))
        try:
            start, n = self.world.set_voxel((v[0], v[1], v[2]), v[3])
        except Exception as e:
            if getattr(e, "kind", "") == "OutOfMemory" and getattr(e, "range", None):
                self._upload(*e.range)   # the edit stopped half-way: the pool changed (same voxels, deeper tree), keep the GPU copy in step
                return None
            if getattr(e, "kind", "") in ("NoChange", "NoChunk", "PosOutOfBounds", "OutOfMemory"):
                return None
            raise
        self._upload(start, n)
        return start, n

    def recenter(self, anchor=(0, 0, 0)):
        """GameState::center_chunks: the grid follows the player; chunks that fell out are freed, the table moves."""
        _, v, _ = self._bcast(self.CMD_RECENTER, anchor)
        removed = self.world.center_chunks((v[0], v[1], v[2]))
        self._roots()
        return removed

    # ---- coherence check ----
    def digest(self) -> int:
        import zlib
        return zlib.crc32(self.world.chunk_roots().tobytes(), zlib.crc32(memoryview(self.world.nodes())))

    def verify(self) -> bool:
        """True when every replica's pool and chunk table have the same digest."""
        if self.count == 1:
            return True
        t = self.torch
        d = t.tensor([self.digest()], dtype=t.int64, device=self.device)
        lo, hi = d.clone(), d.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        return int(lo[0]) == int(hi[0])
