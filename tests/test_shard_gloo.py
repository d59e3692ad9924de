"""The N > 1 path on CPU: two (and three) processes over gloo run the same FrameGather the GPU bench uses —
tile ownership, message layout, the gather call — with the oracle standing in for the HIP render, and
rank 0 checks the assembled frame against the unsharded one."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from voxelraytracing_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["VRT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from oracle import orc
from voxelraytracing_amd import scenes, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
sc = scenes.c2((136, 72))                       # 17 x 9 = 153 tiles: not divisible by 2 or 3 -> padded messages
w, h = sc.size
o = orc.from_package_scene(sc)
tiles, padded, total = shard.tiles_of_rank(w, h, rank, world)
# "render" only this rank's tiles (the oracle renders rectangles; one 8x8 rectangle per owned tile)
frame = np.zeros((h, w, 4), dtype=np.uint32)
for t in tiles:
    ty, tx = divmod(int(t), w // 8)
    rgb, ids, _, _ = o.render(orc.MODE_PRIMARY_SHADOW, w, h, rect=(tx * 8, ty * 8, tx * 8 + 8, ty * 8 + 8), threads=1)
    frame[ty*8:ty*8+8, tx*8:tx*8+8, :3] = rgb[ty*8:ty*8+8, tx*8:tx*8+8].view(np.uint32)
    frame[ty*8:ty*8+8, tx*8:tx*8+8, 3] = ids[ty*8:ty*8+8, tx*8:tx*8+8]
fg = shard.FrameGather(torch, dist, rank, world, w, h, torch.device("cpu"))
assert fg.tiles_padded == padded and fg.msg.numel() == padded * 64 * 4
packed = torch.from_numpy(shard.pack_tiles_numpy(frame, rank, world).view(np.int32))
fg.msg.copy_(packed)
dist.barrier()
fg.gather()
dist.barrier()
# the pipelined path bench.py uses: three frames through submit/drain with a stand-in backend
class FakeGpu:
    def __init__(self): self.assembled = []
    def bind_output(self, ptr): self.bound = ptr
    def assemble(self, gathered_ptr, dst_ptr, stride): self.assembled.append(gathered_ptr)
fake = FakeGpu()
for k in range(3):
    fg.submit(fake, lambda: fg.msgs[fg.k & 1].copy_(packed + k))
fg.drain(fake)
assert fg.pending is None and fg.k == 3
if rank == 0:
    assert fake.assembled == [fg.recv[0].data_ptr(), fg.recv[1].data_ptr(), fg.recv[0].data_ptr()]
    assert torch.equal(fg.recv[1][0], packed + 1) and torch.equal(fg.recv[0][0], packed + 2)
    fg.recv[0].sub_(2)
if rank == 0:
    got = shard.assemble_numpy(fg.gathered.numpy().view(np.uint32), w, h, world)
    rgb, ids = shard.texels_to_frame(got)
    f_rgb, f_ids, _, _ = o.render(orc.MODE_PRIMARY_SHADOW, w, h, threads=2)
    assert np.array_equal(ids, f_ids) and np.array_equal(rgb, f_rgb), "assembled frame differs from the unsharded one"
    print("GLOO_SHARD_OK", world, int((ids != 0).sum()))
# ---- the weighted, in-place root bench.py uses for N > 1: rank 0 holds w0 of every w0 + N - 1 tiles in the frame itself ----
W0 = 3
tiles, padded, total = shard.tiles_of_rank(w, h, rank, world, W0)
frame = np.zeros((h, w, 4), dtype=np.uint32)
for t in tiles:
    ty, tx = divmod(int(t), w // 8)
    rgb, ids, _, _ = o.render(orc.MODE_PRIMARY_SHADOW, w, h, rect=(tx * 8, ty * 8, tx * 8 + 8, ty * 8 + 8), threads=1)
    frame[ty*8:ty*8+8, tx*8:tx*8+8, :3] = rgb[ty*8:ty*8+8, tx*8:tx*8+8].view(np.uint32)
    frame[ty*8:ty*8+8, tx*8:tx*8+8, 3] = ids[ty*8:ty*8+8, tx*8:tx*8+8]
fgw = shard.FrameGather(torch, dist, rank, world, w, h, torch.device("cpu"), root_weight=W0, in_place=True)
assert fgw.tiles_padded == padded and (rank == 0 or len(tiles) <= padded)
if rank == 0:
    fgw.frames[0][0].copy_(torch.from_numpy(frame.view(np.int32)))    # what the in-place root's render leaves in the frame
else:
    fgw.msgs[0].copy_(torch.from_numpy(shard.pack_tiles_numpy(frame, rank, world, W0).view(np.int32)[:fgw.msgs[0].numel()]))
fgw.gather(0)
dist.barrier()
if rank == 0:
    got = shard.assemble_numpy(fgw.recv[0].numpy().view(np.uint32), w, h, world, W0, frame=fgw.frames[0][0].numpy().view(np.uint32).copy())
    rgb, ids = shard.texels_to_frame(got)
    assert np.array_equal(ids, f_ids) and np.array_equal(rgb, f_rgb), "weighted in-place frame differs from the unsharded one"
    print("GLOO_WEIGHTED_OK", world)
# ---- compact messages (8 bytes per pixel slot, shaded at the root on the GPU): half the size, same transport ----
fgc = shard.FrameGather(torch, dist, rank, world, w, h, torch.device("cpu"), root_weight=W0, in_place=True, compact=True)
assert fgc.msgs[0].numel() * 2 == fgw.msgs[0].numel() and fgc.slot_bytes == 8
fgc.msgs[1].copy_(torch.arange(fgc.msgs[1].numel(), dtype=torch.int32) * (rank + 1))
fgc.gather(1)
dist.barrier()
if rank == 0:
    for r in range(1, world):
        assert torch.equal(fgc.recv[1][r], torch.arange(fgc.msgs[1].numel(), dtype=torch.int32) * (r + 1))
    print("GLOO_COMPACT_OK", world)
# ---- several frames per gather: 3 + 2 frames through submit/drain, each assembled from its own slice ----
fgb = shard.FrameGather(torch, dist, rank, world, w, h, torch.device("cpu"), root_weight=W0, in_place=True, compact=True, batch=3)
class BatchGpu:
    def __init__(self): self.bound, self.assembled = [], []
    def bind_output(self, ptr): self.bound.append(ptr)
    def assemble(self, gathered_ptr, dst_ptr, stride, compact=False): self.assembled.append((gathered_ptr, dst_ptr, stride, compact))
bg = BatchGpu()
frame_no = [0]
def fake_render():
    k = fgb.k & 1   # the message set being filled; bg.bound[-1] is the slice just bound
    if rank != 0:
        off = (bg.bound[-1] - fgb.msgs[k].data_ptr()) // 4
        fgb.msgs[k][off:off + fgb.frame_words] = 1000 * frame_no[0] + rank
    frame_no[0] += 1
fgb.reset_host_profile()
fgb.submit(bg, fake_render, 3)
fgb.submit(bg, fake_render, 2)
fgb.drain(bg)
assert frame_no[0] == 5 and len(bg.bound) == 5
hp = fgb.host_profile()   # what the two collectives cost this rank's host thread (bench.py: expected_scaling's host term)
assert hp["collectives"] == 2 and hp["frames_per_collective"] == 2.5 and hp["gather_call_us"] > 0 and hp["wait_us"] >= 0 and hp["assemble_us"] >= 0
if rank == 0:
    fw = fgb.frame_words
    assert [a[0] for a in bg.assembled] == [fgb.recv[0].data_ptr() + j * fw * 4 for j in range(3)] + [fgb.recv[1].data_ptr() + j * fw * 4 for j in range(2)]
    assert all(a[2] == 3 * fw * 4 and a[3] for a in bg.assembled)
    assert [a[1] for a in bg.assembled] == [fgb.frames[0][j].data_ptr() for j in range(3)] + [fgb.frames[1][j].data_ptr() for j in range(2)]
    for r in range(1, world):
        for j in range(3):
            assert (fgb.recv[0][r][j * fw:(j + 1) * fw] == 1000 * j + r).all()
        for j in range(2):
            assert (fgb.recv[1][r][j * fw:(j + 1) * fw] == 1000 * (3 + j) + r).all()
    print("GLOO_BATCH_OK", world)
dist.destroy_process_group()
'''


REPLICA_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["VRT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from voxelraytracing_amd import shard
from voxelraytracing_amd.world import ClientWorld
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
class FakeGpu:                       # records what would go to the device
    def __init__(self): self.ranges, self.roots = [], None
    def write_nodes(self, ptr, a, b): self.ranges.append((a, b))
    def write_chunk_roots(self, r): self.roots = r.copy()
    def write_world_data(self, w): self.wd = w
# the "server": a generated world whose chunks are sent as GiveChunkData messages (rank 0 holds the connection)
server = ClientWorld((2, 1, 2), 1 << 22, 4)
server.generate(0, 11)
cw, gpu = ClientWorld((1, 1, 1), 1 << 21, 2), FakeGpu()       # the client sees chunks (0..1)^3 at first
rw = shard.ReplicatedWorld(torch, dist, rank, world, cw, gpu)
stream = b"".join(server.encode_chunk_msg((x, y, z)) for z in range(3) for y in range(2) for x in range(3)) if rank == 0 else b""
cut = len(stream) - 7 if rank == 0 else 0                     # the last message arrives in two TCP reads
updated, used = rw.ingest_chunk_msgs(stream[:cut])
assert len(updated) == 8 and cw.populated_count() == 8, (len(updated), cw.populated_count())   # 10 of the 18 were outside the grid
assert [b - a for a, b in gpu.ranges] == [n for _, _, n in updated] and gpu.roots is not None
assert rw.verify()
# edits near the terrain surface, decided on rank 0
rng = np.random.default_rng(3)
done = 0
for _ in range(40):
    p = rng.integers(0, 64, 3) if rank == 0 else (0, 0, 0)
    r = rw.set_voxel(tuple(int(v) for v in p), int(rng.choice([0, 4, 3])) if rank == 0 else 0)
    done += r is not None
assert done > 10 and rw.verify()
# the player walks +x: the grid follows, one slab of chunks falls out, new ones arrive
removed = rw.recenter((2, 1, 1))
assert removed == 4 and cw.populated_count() == 4
more = b"".join(server.encode_chunk_msg((2, y, z)) for z in range(2) for y in range(2)) if rank == 0 else b""
updated, _ = rw.ingest_chunk_msgs(more)
assert len(updated) == 4 and cw.populated_count() == 8 and rw.verify()
assert cw.get_voxel((70, 5, 20)) == server.get_voxel((70, 5, 20))
# a replica that drifts is noticed
if rank == world - 1:
    cw.set_voxel((40, 60, 40), 4 if cw.get_voxel((40, 60, 40)) != 4 else 47)
assert not rw.verify()
total = rw.uploaded_nodes
if rank == 0:
    print("REPLICA_OK", world, total, cw.max_nodes())
dist.destroy_process_group()
'''


def test_replicated_world_stays_coherent_over_gloo(tmp_path):
    """Chunk streaming, voxel edits and recentring on 2 replicas: commands are broadcast, pools stay byte-identical,
    and only the touched ranges are uploaded."""
    script = tmp_path / "replica_worker.py"
    script.write_text(REPLICA_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VRT_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
    line = [ln for ln in outs[0].splitlines() if ln.startswith("REPLICA_OK")][0].split()
    assert int(line[2]) < int(line[3])   # everything uploaded in the whole session is less than one whole-pool upload


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_tile_shard_gather_over_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VRT_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-2000:]}"
    assert f"GLOO_SHARD_OK {world}" in outs[0] and f"GLOO_WEIGHTED_OK {world}" in outs[0] and f"GLOO_COMPACT_OK {world}" in outs[0] and f"GLOO_BATCH_OK {world}" in outs[0]


def test_layout_helpers_round_trip():
    rng = np.random.default_rng(3)
    w, h = 72, 40
    frame = rng.integers(0, 2 ** 32, size=(h, w, 4), dtype=np.uint64).astype(np.uint32)
    for n in (1, 2, 5, 45, 64):
        msgs = [shard.pack_tiles_numpy(frame, r, n) for r in range(n)]
        sizes = {m.size for m in msgs}
        assert len(sizes) == 1  # equal-sized messages
        assert np.array_equal(shard.assemble_numpy(msgs, w, h, n), frame)
        owned = np.concatenate([shard.tiles_of_rank(w, h, r, n)[0] for r in range(n)])
        assert sorted(owned.tolist()) == list(range((w // 8) * (h // 8)))
    for n, w0 in ((2, 4), (3, 2), (8, 2), (5, 9)):   # weighted root: still a partition, ranks >= 1 fit the padded message
        msgs = [shard.pack_tiles_numpy(frame, r, n, w0) for r in range(n)]
        assert len({m.size for m in msgs[1:]}) == 1
        assert np.array_equal(shard.assemble_numpy(msgs, w, h, n, w0), frame)
        root_only = shard.assemble_numpy([None] + msgs[1:], w, h, n, w0, frame=np.zeros_like(frame))
        mine = shard.tiles_of_rank(w, h, 0, n, w0)[0]
        keep = np.ones((h // 8, w // 8), dtype=bool)
        keep[np.divmod(mine, w // 8)] = False
        px = np.kron(keep, np.ones((8, 8), dtype=bool))
        assert np.array_equal(root_only[px], frame[px]) and not root_only[~px].any()


def test_root_weight_model():
    """The root takes more of the frame the slower the links are relative to the render; never less than an equal share."""
    assert shard.root_weight_model(1, 0.14, 33e6) == 1
    fast_links = [shard.root_weight_model(n, 0.14, 33e6, link_gbs=1e4) for n in (2, 4, 8)]
    slow_links = [shard.root_weight_model(n, 0.14, 33e6, link_gbs=20.0) for n in (2, 4, 8)]
    assert fast_links == [1, 1, 1] and all(s > 1 for s in slow_links)
    assert slow_links[0] >= slow_links[1] >= slow_links[2]


def test_expected_scaling_states_the_bound_before_the_run():
    """shard.expected_scaling (what bench.py prints as config.expected_scaling on every N > 1 line): C3's primary + shadow frame
    at 1080p is bound by one message over one link and predicted under 2 x at N = 8; C5's 34 ms path-trace frame by a shard's
    render, near N x."""
    from voxelraytracing_amd import shard
    c3 = shard.expected_scaling(8, 0.133, 1920 * 1080, 8, 2)
    assert c3["bound"] in ("link_ms_per_message", "root_render_plus_assemble_ms") and c3["speedup"] < 4.0
    assert abs(c3["message_bytes_per_rank"] - 1920 * 1080 * 8 / 9) < 1
    c5 = shard.expected_scaling(8, 34.0, 3840 * 2160, 16, 2)
    assert c5["bound"] == "root_render_plus_assemble_ms" and 3.5 < c5["speedup"] < 8.0
    even = shard.expected_scaling(8, 34.0, 3840 * 2160, 16, 1)
    assert even["speedup"] > c5["speedup"] and even["speedup"] <= 8.0
    assert shard.expected_scaling(1, 1.0, 100, 16, 1)["speedup"] == 1.0
