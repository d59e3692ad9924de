"""`ClientCmd::GiveChunkData(ChunkPos, Cow<[Node]>, NodeAlloc)` — the chunk payload of the reference's wire protocol
(common/src/net.rs:46-55, bincode 2.0.1 serde, standard config).  The reference holds no captured traffic, so the
bytes are pinned by messages assembled here by hand from bincode's published encoding, and by round trips through
the host mirror; what the client does with a message (client/src/lib.rs:110-118) is checked on the world."""
import struct

import numpy as np
import pytest

from voxelraytracing_amd import world as W
from voxelraytracing_amd.world import ClientWorld, SetVoxelErr


def _varint(v):
    if v < 251:
        return bytes([v])
    if v <= 0xFFFF:
        return bytes([251]) + struct.pack("<H", v)
    if v <= 0xFFFFFFFF:
        return bytes([252]) + struct.pack("<I", v)
    return bytes([253]) + struct.pack("<Q", v)


def _zigzag(v):
    return _varint((v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1)


def _give_chunk_data(pos, nodes, alloc=((0, 2), [(1, 2)], 0)):
    """variant 5, IVec3 as three zig-zag varints, the node slice (len + one u16 varint each), then NodeAlloc
    {range: Range, free_mem: Vec<Range>, last_used_addr}; the server always sends NodeAlloc::new(0..1, 1..2)."""
    (rs, re), free, last = alloc
    out = _varint(5) + b"".join(_zigzag(c) for c in pos)
    out += _varint(len(nodes)) + b"".join(_varint(int(n)) for n in nodes)
    out += _varint(rs) + _varint(re) + _varint(len(free)) + b"".join(_varint(a) + _varint(b) for a, b in free) + _varint(last)
    return out


def test_hand_assembled_message_bytes_and_ingest():
    w = ClientWorld((1, 1, 1), 1 << 18, 2)       # chunks (0..1)^3
    leaf = _give_chunk_data((1, 0, 1), [40])
    # enum index 5 | zigzag(1)=2, 0, 2 | len 1 | node 40 | range 0,2 | 1 free span 1,2 | last_used 0
    assert leaf == bytes([5, 2, 0, 2, 1, 40, 0, 2, 1, 1, 2, 0])
    used, pos, root, n = w.ingest_chunk_msg(leaf)
    assert (used, pos, n) == (len(leaf), (1, 0, 1), 1) and root >= 1
    assert w.nodes()[root] == 40 and w.get_voxel((32 + 5, 7, 32 + 9)) == 40
    # split nodes have bit 15 set: every such word travels as 251 + u16 LE; negative chunk coordinates zig-zag to odd numbers
    tree = W.svo_build_bottom_up(W.gen_dense_superflat((0, 0, 0)))
    msg = _give_chunk_data((-3, 0, 200), tree)
    assert msg[:5] == bytes([5, 5, 0, 251, 0x90]) and msg[5] == 0x01          # zigzag(-3)=5, zigzag(200)=400 -> 251 + 0x0190
    assert msg[6:9] == bytes([251]) + struct.pack("<H", len(tree))            # 5289 nodes
    assert msg[9] == 251 and struct.unpack("<H", msg[10:12])[0] == tree[0] and tree[0] & 0x8000
    with pytest.raises(SetVoxelErr) as e:                                     # outside this client's grid: received_oob_chunks
        w.ingest_chunk_msg(msg)
    assert e.value.kind == "PosOutOfBounds" and e.value.consumed == len(msg)
    used, pos, root, n = w.ingest_chunk_msg(_give_chunk_data((0, 0, 0), tree) + b"\x05trailing bytes of the next command")
    assert pos == (0, 0, 0) and n == len(tree) and np.array_equal(w.nodes()[root:root + n], tree)
    assert used == len(_give_chunk_data((0, 0, 0), tree))
    assert w.get_voxel((3, 12, 3)) == 40 and w.get_voxel((3, 13, 3)) == 0     # Superflat: grass at y = 12


def test_incomplete_and_foreign_messages():
    w = ClientWorld((1, 1, 1), 1 << 18, 2)
    tree = W.svo_build_bottom_up(W.gen_dense_superflat((0, 0, 0)))
    msg = _give_chunk_data((0, 1, 0), tree)
    for cut in (0, 1, 3, 7, 8, 100, len(msg) - 1):       # bincode's UnexpectedEnd: keep the bytes, wait for more
        assert w.ingest_chunk_msg(msg[:cut]) is None
    assert w.populated_count() == 0
    with pytest.raises(ValueError, match="not a GiveChunkData"):
        w.ingest_chunk_msg(bytes([2, 3]) + b"bye")         # ClientCmd::Kick("bye")
    with pytest.raises(ValueError, match="malformed"):
        w.ingest_chunk_msg(bytes([5, 0, 0, 0, 1, 252, 0, 0, 1, 0]))   # a node word that does not fit u16
    with pytest.raises(ValueError, match="malformed"):
        w.ingest_chunk_msg(bytes([5, 0, 0, 0, 253]) + struct.pack("<Q", 1 << 40))   # absurd node count
    assert w.ingest_chunk_msg(msg)[0] == len(msg)


def test_server_side_encoding_round_trips_a_generated_world():
    """Every chunk of a procedural world, encoded as the server would send it and ingested by a fresh client, gives
    the same voxels; a re-sent chunk reuses its slot (world.rs:315-326)."""
    src = ClientWorld((1, 1, 1), 1 << 21, 2)
    src.generate(0, 7)
    dst = ClientWorld((1, 1, 1), 1 << 21, 2)
    stream = b"".join(src.encode_chunk_msg((x, y, z)) for z in range(2) for y in range(2) for x in range(2))
    assert stream[0] == 5
    roots = {}
    while stream:
        used, pos, root, n = dst.ingest_chunk_msg(stream)
        roots[pos] = (root, n)
        stream = stream[used:]
    assert len(roots) == 8 and dst.populated_count() == 8
    rng = np.random.default_rng(5)
    for p in rng.integers(0, 64, size=(3000, 3)):
        assert dst.get_voxel(tuple(int(v) for v in p)) == src.get_voxel(tuple(int(v) for v in p))
    again = dst.ingest_chunk_msg(src.encode_chunk_msg((1, 1, 0)))
    assert (again[2], again[3]) == roots[(1, 1, 0)]
    with pytest.raises(KeyError):
        ClientWorld((1, 1, 1), 1 << 12, 2).encode_chunk_msg((0, 0, 0))
