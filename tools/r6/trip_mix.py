"""tools/r6/trip_mix.py — what the wave-level march trips of a frame are (needs the instrumented build tools/ab/libvrt_tripstats.so:
VRT_LIB=tools/ab/libvrt_tripstats.so): trips in the inner loop, trips through the general step, and of those the ones with a lane that
stops, with a lane in water, with neither (bricks / liquids only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from voxelraytracing_amd import scenes, graphics as g
from util import gpu_for_scene

for name, sc in (("C2", scenes.c2()), ("C3", scenes.c3())):
    gpu = gpu_for_scene(sc)
    gpu.render(g.MODE_PRIMARY_SHADOW)
    s = gpu.stats()
    fast, slow, water, other, stop, stoponly = s.steps, s.node_visits, s.primary_steps, s.primary_node_visits, s.hits, s.secondary_rays
    tot = fast + slow
    print(f"{name}: trips {tot}  inner loop {fast} ({fast / tot:.3f})  general step {slow} ({slow / tot:.3f}): with a lane in water {water} ({water / tot:.3f}), "
          f"with a stop {stop} ({stop / tot:.3f}), of which nothing but stops and plain air {stoponly} ({stoponly / tot:.3f}), neither stop nor water {other} ({other / tot:.3f})")
    gpu.close()
