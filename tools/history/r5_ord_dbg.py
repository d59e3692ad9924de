import ctypes as C, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
os.environ["VRT_TILE_ORDER_MOVING"] = "1"   # (history: the order kernel carried per-phase clock stamps under -DVRT_EXP_ORDDBG while it lived in the experiments build; they went when it became the product's)
from voxelraytracing_amd import Gpu, MODE_PRIMARY_SHADOW, _ffi, scenes, graphics as g
sc = scenes.c2()
gpu = Gpu(sc.world.max_nodes(), sc.world.size_in_chunks(), sc.size)
gpu.upload_world(sc.world, sc.materials); gpu.write_settings(sc.settings)
gpu.set_frames_in_flight(1)
lib = _ffi.vrt(); lib.vrt_dbg_order.argtypes = [C.c_void_p]
buf = np.zeros(8, dtype=np.uint64)
for k in range(30):
    gpu.write_cam_data(g.cam_data_create((sc.rot[0] + 0.3 * k, sc.rot[1] + 0.9 * k, 0.0), (sc.eye[0] + 0.5 * k, sc.eye[1], sc.eye[2] - 0.4 * k), 70.0, (1920.0, 1080.0)))
    gpu.render(MODE_PRIMARY_SHADOW)
    gpu.synchronize()
    lib.vrt_dbg_order(buf.ctypes.data)
    if k > 25: print("phases (us): block max %.1f, dilate+count %.1f, scan %.1f, place+store %.1f; total %.1f" % (tuple((int(buf[i+1])-int(buf[i]))/100 for i in range(4)) + ((int(buf[4])-int(buf[0]))/100,)))
