"""voxelraytracing_amd — MI355X-native backend for the per-pixel SVO ray-march of
MasonFeurer/VoxelRayTracing (clientdesktop/src/graphics/ray_tracer.wgsl and its host feed).

The product is two native libraries built in-tree (see __graft_entry__.build):
``libvrt.so`` (HIP kernels + the C ABI of include/vrt.h) and ``libvrt_host.so`` (C++ mirror of the
reference's world/camera API, include/vrt_host.h).  This package only binds them.
"""
from . import _ffi  # noqa: F401
from .graphics import (Gpu, VrtError, cam_data_create, axis_rot_to_ray, make_settings, std_materials,  # noqa: F401
                       CamData, Material, Settings, WorldData, MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_PATH)
from .world import ClientWorld, Node, SetVoxelErr  # noqa: F401
