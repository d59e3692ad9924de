// graphics.hpp — C++ host mirror of the reference's GPU boundary types, on top of the vrt_* C ABI.
//
//   Material, CamData::create, WorldData::from, Settings   <- clientdesktop/src/graphics/mod.rs:20-143
//   NodeBuffer, SimpleBuffer, ArrayBuffer, Buffers          <- clientdesktop/src/graphics/shader.rs:7-143
//   PixelShader::encode_pass, GpuResources                  <- shader.rs:295-380, mod.rs:145-212
// A frame loop written against the reference (main.rs:398-455) maps call for call; wgpu's Gpu/encoder
// arguments disappear because the backend owns its device and stream.
#pragma once

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/vrt.h"
#include "world.hpp"

namespace vrt {

using Material = vrt_material;
using Settings = vrt_settings;

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Vec2 { float x = 0, y = 0; };
struct UVec2 { uint32_t x = 0, y = 0; };

// Column-major 4x4 with glam 0.31's operation order (a*b: column j = sum_k a.col(k) * b[j][k], left to right).
struct Mat4 {
    float m[16];
    static Mat4 identity() { Mat4 r; std::memset(r.m, 0, sizeof r.m); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
    static Mat4 from_translation(Vec3 t) { Mat4 r = identity(); r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; return r; }
    static Mat4 from_rotation_x(float a) { Mat4 r = identity(); const float s = std::sin(a), c = std::cos(a); r.m[5] = c; r.m[6] = s; r.m[9] = -s; r.m[10] = c; return r; }
    static Mat4 from_rotation_y(float a) { Mat4 r = identity(); const float s = std::sin(a), c = std::cos(a); r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r; }
    static Mat4 from_rotation_z(float a) { Mat4 r = identity(); const float s = std::sin(a), c = std::cos(a); r.m[0] = c; r.m[1] = s; r.m[4] = -s; r.m[5] = c; return r; }
    Mat4 operator*(const Mat4 &b) const {
        Mat4 r;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++)
                r.m[j * 4 + i] = m[i] * b.m[j * 4] + m[4 + i] * b.m[j * 4 + 1] + m[8 + i] * b.m[j * 4 + 2] + m[12 + i] * b.m[j * 4 + 3];
        return r;
    }
    // inverse of perspective_rh(fovy, aspect, near, far) in closed form: the projection is
    // cols (w,0,0,0),(0,h,0,0),(0,0,r,-1),(0,0,r*n,0), so its inverse is
    // cols (1/w,0,0,0),(0,1/h,0,0),(0,0,0,1/(r*n)),(0,0,-1,1/n).  glam inverts with a general cofactor
    // expansion; entries agree to a few ulp and the shader reads only columns 0 and 1 (ray_tracer.wgsl:163).
    static Mat4 perspective_rh_inverse(float fovy, float aspect, float z_near, float z_far) {
        const float s = std::sin(0.5f * fovy), c = std::cos(0.5f * fovy);
        const float h = c / s, w = h / aspect, r = z_far / (z_near - z_far);
        Mat4 o;
        std::memset(o.m, 0, sizeof o.m);
        o.m[0] = 1.0f / w;
        o.m[5] = 1.0f / h;
        o.m[11] = 1.0f / (r * z_near);
        o.m[14] = -1.0f;
        o.m[15] = 1.0f / z_near;
        return o;
    }
};

inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }

struct CamData : vrt_cam_data {
    // CamData::create(cam /*rot, degrees*/, eye, fov /*degrees*/, proj_size) — mod.rs:92-111
    static CamData create(Vec3 cam, Vec3 eye, float fov, Vec2 proj_size) {
        const Mat4 inv_view = Mat4::from_translation(eye) * Mat4::from_rotation_x(to_radians(cam.x)) *
                              Mat4::from_rotation_y(-to_radians(cam.y)) * Mat4::from_rotation_z(to_radians(cam.z));
        const Mat4 inv_proj = Mat4::perspective_rh_inverse(to_radians(fov), proj_size.x / proj_size.y, 0.001f, 1000.0f);
        CamData c;
        std::memset(static_cast<vrt_cam_data *>(&c), 0, sizeof(vrt_cam_data));
        c.pos[0] = eye.x; c.pos[1] = eye.y; c.pos[2] = eye.z;
        std::memcpy(c.inv_view_mat, inv_view.m, sizeof inv_view.m);
        std::memcpy(c.inv_proj_mat, inv_proj.m, sizeof inv_proj.m);
        c.proj_size[0] = proj_size.x; c.proj_size[1] = proj_size.y;
        return c;
    }
};

struct WorldData : vrt_world_data {
    // WorldData::from(&ClientWorld) — mod.rs:121-130
    static WorldData from(const ClientWorld &w) {
        WorldData d;
        std::memset(static_cast<vrt_world_data *>(&d), 0, sizeof(vrt_world_data));
        const VoxelPos mn = w.min_voxel();
        d.min[0] = mn.x; d.min[1] = mn.y; d.min[2] = mn.z;
        d.size = w.size_in_voxels();
        d.size_in_chunks = w.size_in_chunks();
        return d;
    }
};

// Player camera fields, client/src/player.rs:31-70 (physics is out of scope)
struct Player {
    float fov = 70.0f;
    Vec3 pos, cam_pos_, rot;  // rot in degrees
    float height = 4.0f;
    explicit Player(Vec3 p) : pos(p), cam_pos_{p.x, p.y + 4.0f, p.z} {}
    Vec3 cam_pos() const { return cam_pos_; }
    // facing() = axis_rot_to_ray(rot in radians), common/src/math.rs:131-146
    Vec3 facing() const {
        const float rx = to_radians(rot.x), ry = to_radians(rot.y);
        const float r = std::cos(rx);
        return {r * -std::sin(ry), -std::sin(rx), r * -std::cos(ry)};
    }
};

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };

// Owns the vrt context; stands where the reference passes `&Gpu`.
class Gpu {
public:
    Gpu(uint32_t max_nodes, uint32_t world_size, UVec2 result_size, int device = -1, uint32_t shard_rank = 0, uint32_t shard_count = 1) {
        vrt_config cfg{max_nodes, world_size, result_size.x, result_size.y, device, shard_rank, shard_count, 0, 0, 0, {0}};
        if (vrt_create(&cfg, &ctx_) != VRT_OK) throw GpuError(vrt_last_error(nullptr));
    }
    // One context over several devices (vrt_config.device_ids): the frame is traced by all of them and lands on devices[0].
    Gpu(uint32_t max_nodes, uint32_t world_size, UVec2 result_size, const std::vector<int> &devices) {
        vrt_config cfg{max_nodes, world_size, result_size.x, result_size.y, -1, 0, 0, 0, 0, (uint32_t)devices.size(), {0}};
        for (size_t i = 0; i < devices.size() && i < VRT_MAX_DEVICES; i++) cfg.device_ids[i] = devices[i];
        if (vrt_create(&cfg, &ctx_) != VRT_OK) throw GpuError(vrt_last_error(nullptr));
    }
    ~Gpu() { vrt_destroy(ctx_); }
    Gpu(const Gpu &) = delete;
    Gpu &operator=(const Gpu &) = delete;
    vrt_ctx *ctx() const { return ctx_; }
    void check(int rc) const { if (rc != VRT_OK) throw GpuError(vrt_last_error(ctx_)); }

private:
    vrt_ctx *ctx_ = nullptr;
};

// NodeBuffer, shader.rs:7-41
struct NodeBuffer {
    uint32_t size_;
    explicit NodeBuffer(uint32_t size) : size_(size & ~1u) {}
    uint32_t size() const { return size_; }
    void write(const Gpu &gpu, const Node *src_nodes, NodeRange range) const {
        gpu.check(vrt_write_nodes(gpu.ctx(), reinterpret_cast<const uint16_t *>(src_nodes), range.start, range.end));
    }
};

// ArrayBuffer<NodeAddr>, shader.rs:117-143 (the chunk_roots table)
struct ChunkRootsBuffer {
    uint32_t size_;
    uint32_t size() const { return size_; }
    // tag: ChunkGrid::roots_generation() — an unchanged table is then not even compared (vrt_write_chunk_roots_tagged)
    void write(const Gpu &gpu, uint64_t offset, const std::vector<NodeAddr> &items, uint64_t tag = 0) const {
        gpu.check(vrt_write_chunk_roots_tagged(gpu.ctx(), (uint32_t)offset, items.data(), (uint32_t)items.size(), tag));
    }
};

// Crosshair, mod.rs:63-80 (the C ABI's struct is its layout; the defaults are Default::default()'s)
struct Crosshair : vrt_crosshair {
    Crosshair() : vrt_crosshair{{1.0f, 1.0f, 1.0f, 0.33f}, 2u, 5.0f, {0u, 0u}} {}
};

// Buffers, shader.rs:43-81. SimpleBuffer<T>::write becomes a typed setter.
struct Buffers {
    NodeBuffer nodes;
    ChunkRootsBuffer chunk_roots;
    // the blit's two uniforms (shader.rs:51-52): the reference writes them every frame (main.rs:429-432) in front of its two passes.
    // Together they are vrt_set_presentation's declaration — an unchanged one costs a compare — under which a frame whose window has
    // the texture's size stores the window's image in its own launch.  present_flags: VRT_PRESENT_SKIP_TEXELS for a client that
    // never reads the f32 frame back.
    mutable float screen_size_[2] = {0.0f, 0.0f};
    uint32_t present_flags = 0u;
    Buffers(uint32_t max_nodes, uint32_t world_size) : nodes(max_nodes), chunk_roots{world_size * world_size * world_size} {}
    void write_screen_size(const Gpu &, const float (&size)[2]) const { screen_size_[0] = size[0]; screen_size_[1] = size[1]; }
    void write_crosshair(const Gpu &g, const Crosshair &c) const {
        g.check(vrt_set_presentation(g.ctx(), &c, (uint32_t)screen_size_[0], (uint32_t)screen_size_[1], present_flags));
    }
    void write_cam_data(const Gpu &g, const CamData &c) const { g.check(vrt_set_camera(g.ctx(), &c)); }
    void write_settings(const Gpu &g, const Settings &s) const { g.check(vrt_set_settings(g.ctx(), &s)); }
    void write_world_data(const Gpu &g, const WorldData &w) const { g.check(vrt_set_world(g.ctx(), &w)); }
    void write_voxel_materials(const Gpu &g, uint64_t idx, const std::vector<Material> &m) const {
        g.check(vrt_write_materials(g.ctx(), (uint32_t)idx, m.data(), (uint32_t)m.size()));
    }
    void resize_chunk_buffer(const Gpu &g, uint32_t world_size) {  // shader.rs:74-80
        g.check(vrt_resize_world(g.ctx(), world_size));
        chunk_roots.size_ = world_size * world_size * world_size;
    }
};

// PixelShader, shader.rs:295-380
struct PixelShader {
    vrt_mode mode;
    // encode_pass(encoder, workgroups) + submit: workgroups must be result_size/8 as at main.rs:452
    void encode_pass(const Gpu &g, UVec2 /*workgroups*/) const {
        vrt_render_opts o{};
        o.mode = mode;
        g.check(vrt_render(g.ctx(), &o));
    }
};

// ScreenShader, shader.rs:216-294: encode_pass(encoder, view) blits the result texture into the window's image under the crosshair.
// Here the image stays on the device (vrt_present_device: the pointer a window system takes through interop), enqueued behind the
// frame — or, declared through Buffers::write_screen_size / write_crosshair, already stored by the frame's own launch.
struct ScreenShader {
    struct View { void *rgba8_device = nullptr; uint64_t bytes = 0; };
    View encode_pass(const Gpu &g, const Crosshair &c, UVec2 window) const {
        View v;
        g.check(vrt_present_device(g.ctx(), &c, window.x, window.y, &v.rgba8_device, &v.bytes));
        return v;
    }
};

// GpuResources, mod.rs:145-212
struct GpuResources {
    UVec2 result_size;
    Buffers buffers;
    PixelShader ray_tracer{VRT_MODE_PRIMARY};
    PixelShader shadow_tracer{VRT_MODE_PRIMARY_SHADOW};
    ScreenShader screen_shader;
    GpuResources(UVec2 result_size_, uint32_t max_nodes, uint32_t world_size) : result_size(result_size_), buffers(max_nodes, world_size) {}
    void use_new_world_size(const Gpu &g, uint32_t world_size) { buffers.resize_chunk_buffer(g, world_size); }
    void resize_result_texture(const Gpu &g, UVec2 new_size) { g.check(vrt_resize_output(g.ctx(), new_size.x, new_size.y)); result_size = new_size; }
};

}  // namespace vrt
