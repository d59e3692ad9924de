#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with the CPU oracle.

The reference (Rust + WGSL on wgpu) has no tests or fixtures and cannot run here, so the vectors are the
oracle's own output ("parity unpinned" by the reference); they pin the oracle against
regressions and give the GPU tests inputs that do not depend on the oracle being runnable.  Scenes come
from the host library's deterministic builders.  Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import orc  # noqa: E402
from voxelraytracing_amd import scenes  # noqa: E402

CASES = {
    "c1_64": lambda: scenes.c1_flat((64, 64)),
    "c1_256": lambda: scenes.c1_flat((256, 256)),
    "c2_128x72": lambda: scenes.c2((128, 72)),
    "c4_128x72": lambda: scenes.c4((128, 72), bounces=4),
}
PATH_SPP, PATH_SEED = 2, 11


def main():
    orc.build()
    for name, make in CASES.items():
        sc = make()
        o = orc.from_package_scene(sc)
        out = {}
        modes = ((orc.MODE_PATH, "path"),) if name.startswith("c4") else ((orc.MODE_PRIMARY, "primary"), (orc.MODE_PRIMARY_SHADOW, "shadow"))
        for mode, tag in modes:
            rgb, ids, steps, st = o.render(mode, *sc.size, want_steps=True, spp=PATH_SPP, seed=PATH_SEED)
            out[f"{tag}_rgb"], out[f"{tag}_ids"], out[f"{tag}_steps"] = rgb, ids, steps
            out[f"{tag}_stats"] = np.array([st.primary_rays, st.secondary_rays, st.hits, st.steps, st.node_visits,
                                            st.primary_steps, st.primary_node_visits], dtype=np.uint64)
        out["eye"], out["rot"] = np.array(sc.eye, dtype=np.float32), np.array(sc.rot, dtype=np.float32)
        out["cam_bytes"] = np.frombuffer(bytes(sc.cam), dtype=np.uint8)
        out["nodes_crc"] = np.array([np.bitwise_xor.reduce(sc.world.nodes().astype(np.uint64) * np.arange(1, sc.world.max_nodes() + 1, dtype=np.uint64))])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
