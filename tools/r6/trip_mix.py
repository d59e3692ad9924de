"""tools/r6/trip_mix.py — what the wave-level march trips of a frame are: trips inside the hand-written loop, trips through the general step, and
of those the ones with a lane that stops, with a lane in water, with neither.  Needs the instrumented build:
    git apply tools/r6/tripstats.patch && bash tools/ab/build_variant.sh tripstats "" && git apply -R tools/r6/tripstats.patch
    VRT_LIB=tools/ab/libvrt_tripstats.so python tools/r6/trip_mix.py          (on the GPU box)
(profiles/r06_step_asm.txt section 2 was counted with the same counters when the loop still left for every stop and every split cell.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from voxelraytracing_amd import scenes, graphics as g
from util import gpu_for_scene

for name, sc in (("C2", scenes.c2()), ("C3", scenes.c3())):
    gpu = gpu_for_scene(sc)
    gpu.render(g.MODE_PRIMARY_SHADOW)
    s = gpu.stats()
    fast, slow, water, other, stop = s.steps, s.node_visits, s.primary_steps, s.primary_node_visits, s.hits
    tot = fast + slow
    print(f"{name}: trips {tot}  inner loop {fast} ({fast / tot:.3f})  general step {slow} ({slow / tot:.3f}): with a lane in water {water} ({water / tot:.3f}), "
          f"with a stop {stop} ({stop / tot:.3f}), neither stop nor water {other} ({other / tot:.3f})")
    gpu.close()
