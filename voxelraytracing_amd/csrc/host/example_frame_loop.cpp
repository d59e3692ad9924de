// example_frame_loop.cpp — the reference's join_game + frame loop (clientdesktop/src/main.rs:189-229,
// 278-297, 398-455) written against the C++ host mirror, with the wgpu backend replaced by libvrt.so.
//
//   vrt_frame_loop <out.bin> [width height [devices]]      devices = a comma-separated list of HIP ordinals: one context
//                                                          over several devices (vrt_config.device_ids); "0,0,0" rehearses
//                                                          the multi-device path on one GPU
// builds the C1 world (2^3 chunks, Superflat rule via Svo::set_node), uploads it, renders one primary frame,
// edits two voxels the way update_input does (re-uploading the chunk's range), renders a primary+shadow frame
// and writes {w, h, ids[w*h], rgb[w*h*3]} of the second frame to out.bin.  tests/test_cpp_host.py compares
// the file with the same sequence driven through the Python bindings.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "graphics.hpp"
#include "materials.hpp"
#include "worldgen.hpp"

using namespace vrt;

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.bin [w h]\n", argv[0]); return 2; }
    const uint32_t W = argc > 3 ? (uint32_t)std::atoi(argv[2]) : 256, H = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 256;
    try {
        // join_game: ClientWorld::new(player_chunk, max_nodes, size) — main.rs:199
        const uint32_t max_nodes = 1u << 18, world_size = 2;
        ClientWorld world({1, 1, 1}, max_nodes, world_size);
        std::vector<uint16_t> dense(32768);
        std::vector<Node> scratch(NODES_PER_CHUNK + 64);
        for (int32_t z = 0; z < 2; z++)
            for (int32_t y = 0; y < 2; y++)
                for (int32_t x = 0; x < 2; x++) {
                    fill_dense_superflat({x, y, z}, dense.data());
                    const uint32_t used = build_svo_by_set_node(dense.data(), scratch.data(), (uint32_t)scratch.size());
                    if (used == 1 && scratch[0].w == 0) continue;  // all air: leave the cell empty
                    SetVoxelErr err;
                    world.create_chunk({x, y, z}, scratch.data(), used, err);  // GameState::process_cmd, lib.rs:112-119
                    if (err != SetVoxelErr::Ok) throw GpuError("create_chunk failed");
                }

        // GpuResources::new(gpu, fmt, result_size, max_nodes, world_size) — main.rs:211-217
        std::vector<int> devices;
        if (argc > 4)
            for (const char *p = argv[4]; *p;) {
                devices.push_back(std::atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        std::unique_ptr<Gpu> gpu_holder(devices.size() > 1 ? new Gpu(max_nodes, world_size, {W, H}, devices) : new Gpu(max_nodes, world_size, {W, H}));
        Gpu &gpu = *gpu_holder;
        GpuResources res({W, H}, max_nodes, world_size);
        res.buffers.nodes.write(gpu, world.nodes(), {0, res.buffers.nodes.size()});  // main.rs:218
        std::vector<Material> mats(256);
        for (int i = 0; i < 256; i++) {
            mats[i] = Material{};
            if (i < kStdVoxelCount) {
                mats[i].color[0] = kStdVoxels[i].r; mats[i].color[1] = kStdVoxels[i].g; mats[i].color[2] = kStdVoxels[i].b;
                mats[i].is_empty = kStdVoxels[i].is_empty; mats[i].is_liquid = kStdVoxels[i].is_liquid;
            }
        }
        res.buffers.write_voxel_materials(gpu, 0, mats);  // main.rs:219-223

        Player player({32.5f, 16.5f, 60.5f});  // cam_pos = pos + (0,4,0)
        player.rot = {15.0f, 0.0f, 0.0f};
        Settings settings{};
        settings.max_ray_bounces = 3; settings.sun_intensity = 4.0f;
        settings.sky_color[0] = 0.81f; settings.sky_color[1] = 0.93f; settings.sky_color[2] = 1.0f;
        settings.sun_pos[0] = 10000.0f; settings.sun_pos[1] = 20000.0f; settings.sun_pos[2] = 5000.0f;

        const Crosshair crosshair;   // Default::default(), main.rs:183
        ScreenShader::View view;
        auto draw_frame = [&](const PixelShader &shader) {  // main.rs:426-454
            res.buffers.write_settings(gpu, settings);
            res.buffers.write_screen_size(gpu, {(float)W, (float)H});   // main.rs:429-431 (the result texture's size)
            res.buffers.write_crosshair(gpu, crosshair);               // :432
            res.buffers.write_cam_data(gpu, CamData::create(player.rot, player.cam_pos(), player.fov, {(float)W, (float)H}));
            res.buffers.chunk_roots.write(gpu, 0, world.chunk_roots(), world.roots_generation());
            res.buffers.write_world_data(gpu, WorldData::from(world));
            shader.encode_pass(gpu, {W / 8, H / 8});
            view = res.screen_shader.encode_pass(gpu, crosshair, {W, H});   // :454
        };
        draw_frame(res.ray_tracer);

        // update_input: set_voxel + re-upload of the chunk's whole range — main.rs:340-362
        const VoxelPos edits[2] = {{32, 12, 40}, {30, 13, 44}};
        const uint16_t vox[2] = {0, 4};
        for (int i = 0; i < 2; i++) {
            const Chunk *c = nullptr;
            if (world.set_voxel(edits[i], Voxel(vox[i]), &c) != SetVoxelErr::Ok) throw GpuError("set_voxel failed");
            res.buffers.nodes.write(gpu, world.nodes(), c->range);
        }
        draw_frame(res.shadow_tracer);

        std::vector<float> rgb((size_t)W * H * 3);
        std::vector<uint32_t> ids((size_t)W * H);
        gpu.check(vrt_read_output(gpu.ctx(), rgb.data(), ids.data(), nullptr));
        FILE *f = std::fopen(argv[1], "wb");
        if (!f) throw GpuError("cannot open output file");
        std::fwrite(&W, 4, 1, f); std::fwrite(&H, 4, 1, f);
        std::fwrite(ids.data(), 4, ids.size(), f);
        std::fwrite(rgb.data(), 4, rgb.size(), f);
        // the window's image of the last frame (left on the device by the blit — here: stored by the frame's own launch)
        std::vector<uint8_t> image((size_t)W * H * 4);
        gpu.check(vrt_present(gpu.ctx(), &crosshair, W, H, image.data()));
        if (view.bytes != image.size() || !view.rgba8_device) throw GpuError("ScreenShader::encode_pass returned no image");
        std::fwrite(image.data(), 1, image.size(), f);
        std::fclose(f);
        const Vec3 fc = player.facing();
        std::printf("frame_loop ok %ux%u facing %.6f %.6f %.6f\n", W, H, fc.x, fc.y, fc.z);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "frame_loop failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
