/*
 * vrt_oracle.c — CPU ORACLE (test infrastructure, see vrt_oracle.h).  Plain C99, scalar f32.
 *
 * Build:  gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * Float semantics are strict IEEE-754 binary32: no FMA contraction, no reassociation, correctly
 * rounded / and sqrt.  Every expression below is written out in the evaluation order the WGSL text
 * gives (left-to-right for a+b+c, component-wise for vectors); where WGSL leaves the result
 * implementation-defined the choice made here is stated in a comment and repeated in DESIGN.md.
 */
#include "vrt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }

/* WGSL `min(a,b)` with a NaN operand is implementation-defined.  Choice: IEEE minNum — a NaN
 * operand is ignored (what v_min_f32 / Vulkan NMin-class hardware does).  Written as compares so
 * that ±0 ties are decided the same way everywhere: returns b unless a < b or b is NaN. */
static inline float orc_min(float a, float b) { return (a < b || b != b) ? a : b; }

/* WGSL i32(f32): saturating, NaN -> 0. Only the NaN case can occur here (pos is range-checked). */
static inline int32_t orc_f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

/* WGSL sign(): 1, -1, or the operand itself when it is 0 (keeps -0) or NaN. */
static inline float orc_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x); }

/* WGSL clamp = min(max(e, lo), hi). Operands here are never NaN except through e. */
static inline float orc_clamp(float e, float lo, float hi) {
    float m = (e > lo) ? e : lo; /* max(e, lo); a NaN e yields lo */
    return orc_min(m, hi);
}

/* WGSL smoothstep(e0,e1,x): t = clamp((x-e0)/(e1-e0),0,1); t*t*(3-2t). */
static inline float orc_smoothstep(float e0, float e1, float x) {
    float t = orc_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* WGSL mix(a,b,t) = a*(1-t) + b*t. */
static inline float orc_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static inline float orc_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

/* WGSL normalize(v) = v / length(v), length = sqrt(dot(v,v)). */
static inline v3 orc_normalize(v3 v) {
    float len = sqrtf(orc_dot(v, v));
    return V3(v.x / len, v.y / len, v.z / len);
}

/* ------------------------------------------------------------------------------------------ */
/* Node decode: ray_tracer.wgsl:38-51                                                          */
/* ------------------------------------------------------------------------------------------ */

uint32_t orc_get_node(const uint32_t *pairs, uint32_t idx) {
    uint32_t pair = pairs[idx >> 1];
    uint32_t shift = (idx & 1u) * 16u;
    return (pair >> shift) & 0x0000FFFFu;
}

/* The pool is little-endian u16; reading nodes[idx] equals get_node on the u32-pair view
 * (tests/test_oracle_kat.py checks the equivalence).  What a storage read past the end of the buffer
 * yields is implementation-defined in WGSL (wgpu never faults); choice: it reads 0, an air leaf — what the
 * HIP backend's range-checked buffer loads return.  Scenes under test never trigger it. */
static inline uint32_t node_at(const orc_scene *s, uint32_t idx) {
    if (idx >= s->n_nodes) return 0;
    return s->nodes[idx];
}
static inline int node_is_split(uint32_t n) { return (n >> 15) != 0; }
static inline uint32_t node_voxel(uint32_t n) { return n & 0x7FFFu; }
static inline uint32_t node_child_idx(uint32_t n) { return n & 0x7FFFu; }

typedef struct {
    uint32_t idx;
    v3 min, max, center;
    float size;
    uint32_t root;
    uint32_t depth; /* not in the WGSL struct; kept for the node-visit statistic */
} found_node;

/* find_chunk_node, ray_tracer.wgsl:76-114 */
static found_node find_chunk_node(const orc_scene *s, v3 pos, uint32_t max_depth, v3 min, uint32_t root) {
    v3 center = V3(min.x + 32.0f / 2.0f, min.y + 32.0f / 2.0f, min.z + 32.0f / 2.0f);
    float size = 32.0f;
    uint32_t idx = 0, depth = 0;
    for (;;) {
        uint32_t node = node_at(s, root + idx);
        if (!node_is_split(node) || depth == max_depth) {
            found_node out;
            float hs = size * 0.5f;
            out.idx = idx;
            out.min = V3(center.x - hs, center.y - hs, center.z - hs);
            out.max = V3(center.x + hs, center.y + hs, center.z + hs);
            out.center = center;
            out.size = size;
            out.root = root;
            out.depth = depth;
            return out;
        }
        size *= 0.5f;
        int gx = pos.x >= center.x, gy = pos.y >= center.y, gz = pos.z >= center.z;
        uint32_t child = (uint32_t)gx | ((uint32_t)gy << 1) | ((uint32_t)gz << 2);
        idx = node_child_idx(node) + child;
        float q = size * 0.5f;
        center.x += q * (float)(gx * 2 - 1);
        center.y += q * (float)(gy * 2 - 1);
        center.z += q * (float)(gz * 2 - 1);
        depth += 1;
    }
}

/* find_node, ray_tracer.wgsl:116-125 */
static found_node find_node(const orc_scene *s, v3 pos, uint32_t max_depth) {
    uint32_t w = s->world.size_in_chunks;
    int32_t cx = orc_f2i(floorf(pos.x / 32.0f));
    int32_t cy = orc_f2i(floorf(pos.y / 32.0f));
    int32_t cz = orc_f2i(floorf(pos.z / 32.0f));
    v3 min = V3((float)(cx * 32), (float)(cy * 32), (float)(cz * 32));
    uint32_t chunk_idx = (uint32_t)cx + (uint32_t)cy * w + (uint32_t)cz * w * w;
    /* A read past the table: WGSL leaves it to the implementation (any in-bounds element, or zero); the index is CLAMPED
     * here, as naga's Restrict policy does.  Positions inside (0, size) cannot get here when size = 32 * size_in_chunks
     * (i32(NaN) is 0); a WorldData that breaks that can — tests/golden/wgsl_oob.npz holds the shader's text under both
     * policies for such a scene (and the C ABI refuses it). */
    if (chunk_idx >= s->n_chunk_roots) chunk_idx = s->n_chunk_roots - 1;
    uint32_t root = s->chunk_roots[chunk_idx];
    return find_chunk_node(s, pos, max_depth, min, root);
}

void orc_find_node(const orc_scene *scene, const float pos[3], uint32_t max_depth, uint32_t *out10) {
    found_node f = find_node(scene, V3(pos[0], pos[1], pos[2]), max_depth);
    float fl[7] = {f.min.x, f.min.y, f.min.z, f.max.x, f.max.y, f.max.z, f.size};
    out10[0] = f.idx; out10[1] = f.root; out10[2] = f.depth;
    memcpy(out10 + 3, fl, sizeof fl);
}

/* voxel_mats[voxel]: the buffer holds 256 materials (shader.rs:48) and a voxel id has 15 bits: ids >= 256 read past it, which
 * WGSL leaves to the implementation.  CLAMPED to 255 here and in the kernels (naga's Restrict policy);
 * tests/golden/wgsl_oob.npz holds the reference's shader text under the clamp and under the zero-value policy, and
 * tests/test_oracle_vs_reference_wgsl.py::test_reads_past_an_arrays_end_follow_the_clamp_policy pins the choice. */
static inline const orc_material *mat_at(const orc_scene *s, uint32_t voxel) {
    return &s->materials[voxel > 255u ? 255u : voxel];
}

typedef struct {
    int hit;
    v3 color; /* result.material.color after face shading */
    v3 norm, pos;
    float water_dist;
    uint32_t voxel;
    uint32_t iter_count;
    uint64_t node_visits;
} hit_result;

/* ray_world, ray_tracer.wgsl:182-316 */
static hit_result ray_world(const orc_scene *s, v3 origin, v3 dir) {
    v3 mask = V3(dir.x >= 0.0f ? 1.0f : 0.0f, dir.y >= 0.0f ? 1.0f : 0.0f, dir.z >= 0.0f ? 1.0f : 0.0f);
    v3 imask = V3(1.0f - mask.x, 1.0f - mask.y, 1.0f - mask.z);

    v3 ray_pos = origin;
    /* :188-190 */
    if (ray_pos.x - floorf(ray_pos.x) < 0.001f || ray_pos.y - floorf(ray_pos.y) < 0.001f ||
        ray_pos.z - floorf(ray_pos.z) < 0.001f) {
        ray_pos.x += 0.001f * dir.x;
        ray_pos.y += 0.001f * dir.y;
        ray_pos.z += 0.001f * dir.z;
    }

    const float world_min = 0.0f;
    const float world_max = world_min + (float)s->world.size;

    hit_result result;
    memset(&result, 0, sizeof result); /* WGSL `var result: HitResult;` is zero-initialised */

    /* :197-200 */
    if ((ray_pos.x <= world_min || ray_pos.y <= world_min || ray_pos.z <= world_min) ||
        (ray_pos.x >= world_max || ray_pos.y >= world_max || ray_pos.z >= world_max)) {
        return result;
    }

    /* :206-210 */
    v3 unit_step_size = V3(
        sqrtf(1.0f + (dir.y / dir.x) * (dir.y / dir.x) + (dir.z / dir.x) * (dir.z / dir.x)),
        sqrtf(1.0f + (dir.x / dir.y) * (dir.x / dir.y) + (dir.z / dir.y) * (dir.z / dir.y)),
        sqrtf(1.0f + (dir.x / dir.z) * (dir.x / dir.z) + (dir.y / dir.z) * (dir.y / dir.z)));

    uint32_t voxel = 0;
    v3 norm = V3(0.0f, 0.0f, 0.0f);
    float dist_entered_water = -1.0f;
    float total_len = 0.0f;
    uint32_t iter_count = 0;

    while (iter_count < 500u) {
        iter_count += 1u;

        found_node fn = find_node(s, ray_pos, 5u);
        voxel = node_voxel(node_at(s, fn.root + fn.idx));
        result.node_visits += fn.depth + 1;

        int is_liquid = mat_at(s, voxel)->is_liquid == 1u;

        if (voxel != 0u && !is_liquid) break;
        if (!is_liquid) {
            if (dist_entered_water != -1.0f) {
                result.water_dist += total_len - dist_entered_water;
                dist_entered_water = -1.0f;
            }
        }
        if (is_liquid) {
            if (dist_entered_water == -1.0f) dist_entered_water = total_len;
        }
        /* :243-245 */
        v3 axis_dist = V3(
            ((ray_pos.x - fn.min.x) * imask.x + (fn.max.x - ray_pos.x) * mask.x) * unit_step_size.x,
            ((ray_pos.y - fn.min.y) * imask.y + (fn.max.y - ray_pos.y) * mask.y) * unit_step_size.y,
            ((ray_pos.z - fn.min.z) * imask.z + (fn.max.z - ray_pos.z) * mask.z) * unit_step_size.z);

        /* :247-270 */
        float step;
        if (axis_dist.x == 0.0f) {
            if (axis_dist.y == 0.0f) step = axis_dist.z;
            else if (axis_dist.z == 0.0f) step = axis_dist.y;
            else step = orc_min(axis_dist.y, axis_dist.z);
        } else {
            if (axis_dist.y == 0.0f) {
                if (axis_dist.z == 0.0f) step = axis_dist.x;
                else step = orc_min(axis_dist.x, axis_dist.z);
            } else {
                if (axis_dist.z == 0.0f) step = orc_min(axis_dist.y, axis_dist.x);
                else step = orc_min(axis_dist.x, orc_min(axis_dist.y, axis_dist.z));
            }
        }
        total_len += step;
        float ex = step == axis_dist.x ? 1.0f : 0.0f;
        float ey = step == axis_dist.y ? 1.0f : 0.0f;
        float ez = step == axis_dist.z ? 1.0f : 0.0f;
        /* :272 */
        norm = V3(ex * -orc_sign(dir.x), ey * -orc_sign(dir.y), ez * -orc_sign(dir.z));
        /* :274-283   dir*(step+0.001)*e + dir*step*(1-e), with f32(step != axis) spelled as WGSL does */
        float nx = step != axis_dist.x ? 1.0f : 0.0f;
        float ny = step != axis_dist.y ? 1.0f : 0.0f;
        float nz = step != axis_dist.z ? 1.0f : 0.0f;
        ray_pos.x += dir.x * (step + 0.001f) * ex + dir.x * step * nx;
        ray_pos.y += dir.y * (step + 0.001f) * ey + dir.y * step * ny;
        ray_pos.z += dir.z * (step + 0.001f) * ez + dir.z * step * nz;

        /* :285-290 */
        if ((ray_pos.x < world_min || ray_pos.y < world_min || ray_pos.z < world_min) ||
            (ray_pos.x >= world_max || ray_pos.y >= world_max || ray_pos.z >= world_max)) {
            if (dist_entered_water != -1.0f) result.water_dist += total_len - dist_entered_water;
            result.iter_count = iter_count;
            result.voxel = 0;
            return result;
        }
    }

    /* :293-309 */
    result.hit = 1;
    result.pos = ray_pos;
    result.norm = norm;
    const orc_material *m = mat_at(s, voxel);
    result.color = V3(m->color[0], m->color[1], m->color[2]);
    if (result.norm.x != 0.0f) { result.color.x *= 0.5f; result.color.y *= 0.5f; result.color.z *= 0.5f; }
    if (result.norm.z != 0.0f) { result.color.x *= 0.7f; result.color.y *= 0.7f; result.color.z *= 0.7f; }
    if (result.norm.y == -1.0f) { result.color.x *= 0.2f; result.color.y *= 0.2f; result.color.z *= 0.2f; }
    if (dist_entered_water != -1.0f) result.water_dist += total_len - dist_entered_water;

    /* :311-314 */
    if (s->settings.show_step_count == 1u) {
        float f = orc_clamp((float)iter_count / 500.0f, 0.0f, 1.0f);
        result.color = V3(f, f, f);
    }
    result.voxel = voxel;
    result.iter_count = iter_count;
    return result;
}

/* ray_sky, ray_tracer.wgsl:144-157 */
static v3 ray_sky(const orc_scene *s, v3 origin, v3 dir) {
    const v3 horizon_color = {1.0f, 0.3f, 0.0f};
    const float void_color = 0.03f;
    const float sun_size = 0.01f;

    float ground_to_sky_t = orc_smoothstep(-0.01f, 0.0f, dir.y);
    float sky_gradient_t = powf(orc_smoothstep(0.0f, 0.4f, dir.y), 0.35f);
    v3 sky_gradient = V3(orc_mix(horizon_color.x, s->settings.sky_color[0], sky_gradient_t),
                         orc_mix(horizon_color.y, s->settings.sky_color[1], sky_gradient_t),
                         orc_mix(horizon_color.z, s->settings.sky_color[2], sky_gradient_t));
    v3 sun_dir = orc_normalize(V3(s->settings.sun_pos[0] - (float)s->world.min[0] - origin.x,
                                  s->settings.sun_pos[1] - (float)s->world.min[1] - origin.y,
                                  s->settings.sun_pos[2] - (float)s->world.min[2] - origin.z));
    float sun = (orc_dot(dir, sun_dir) > (1.0f - sun_size) && ground_to_sky_t >= 1.0f) ? 1.0f : 0.0f;
    float add = sun * s->settings.sun_intensity;
    return V3(orc_mix(void_color, sky_gradient.x, ground_to_sky_t) + add,
              orc_mix(void_color, sky_gradient.y, ground_to_sky_t) + add,
              orc_mix(void_color, sky_gradient.z, ground_to_sky_t) + add);
}

static uint32_t orc_unorm8(float x) {
    /* float -> unorm8 conversion of a colour attachment / storage texture: clamp, scale, round to nearest even */
    float c = x > 0.0f ? x : 0.0f;   /* NaN -> 0 */
    c = c < 1.0f ? c : 1.0f;
    return (uint32_t)rintf(c * 255.0f) & 0xFFu;
}

/* screen_shader.wgsl:43-65 over the rgba8unorm result texture (ray_tracer.wgsl:179). */
void orc_present(const float *rgb, uint32_t w, uint32_t h, uint32_t screen_w, uint32_t screen_h,
                 const orc_crosshair *ch, uint8_t *rgba8) {
    const float ssx = (float)screen_w, ssy = (float)screen_h;
    const float cx = ssx * 0.5f, cy = ssy * 0.5f;            /* screen_center (:45) */
    for (uint32_t sy = 0; sy < screen_h; sy++)
        for (uint32_t sx = 0; sx < screen_w; sx++) {
            /* tex_coord at the pixel centre (the interpolated vs_main output, :33-40) */
            const float u = ((float)sx + 0.5f) / ssx, v = ((float)sy + 0.5f) / ssy;
            const float px = u * ssx, py = v * ssy;          /* screen_pos (:44) */
            float mask = 0.0f;
            if (ch->style == 1u) {                           /* dot (:48-50) */
                const float dx = cx - px, dy = cy - py;
                mask = (sqrtf(dx * dx + dy * dy) < ch->size ? 1.0f : 0.0f) * ch->color[3];
            }
            if (ch->style == 2u) {                           /* cross (:51-59) */
                const float dx = fabsf(cx - px), dy = fabsf(cy - py);
                const float wd = ch->size * 0.25f;
                mask = (((dx < ch->size && dy < wd) || (dy < ch->size && dx < wd)) ? 1.0f : 0.0f) * ch->color[3];
            }
            /* textureSample(tex, tex_s, tex_coord) (:61).  The sampler (texture.rs:31-44) is mag Nearest / min Linear with
             * lod_min_clamp = lod_max_clamp = 1.0 over a single mip level: the level of detail is clamped to 1 for every
             * sample, and a level of detail > 0 selects the *minification* filter (Vulkan 1.3 §16.5.7 "if lambda <= 0 the
             * texture is magnified, otherwise minified"; the one level there is serves as level 1 clamped back to 0).  So
             * the blit is bilinear at every window size — at 1:1 the sample points are the texel centres, the weights
             * (1, 0), the result the texel itself.  Bilinear as the spec words it on unnormalised coordinates, f32:
             * i0 = floor(u W - 1/2), a = frac(u W - 1/2), ClampToEdge on the indices,
             * ((t00 (1-a) + t10 a) (1-b) + (t01 (1-a) + t11 a) b), texels decoded from rgba8unorm (k / 255, alpha 1). */
            const float ut = u * (float)w - 0.5f, vt = v * (float)h - 0.5f;
            const float fu = floorf(ut), fv = floorf(vt);
            const float a = ut - fu, b = vt - fv;
            int32_t x0 = (int32_t)fu, y0 = (int32_t)fv, x1 = x0 + 1, y1 = y0 + 1;
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 < 0) x1 = 0;
            if (y1 < 0) y1 = 0;
            if (x0 > (int32_t)w - 1) x0 = (int32_t)w - 1;
            if (y0 > (int32_t)h - 1) y0 = (int32_t)h - 1;
            if (x1 > (int32_t)w - 1) x1 = (int32_t)w - 1;
            if (y1 > (int32_t)h - 1) y1 = (int32_t)h - 1;
            const float *t00 = rgb + ((size_t)y0 * w + (size_t)x0) * 3, *t10 = rgb + ((size_t)y0 * w + (size_t)x1) * 3;
            const float *t01 = rgb + ((size_t)y1 * w + (size_t)x0) * 3, *t11 = rgb + ((size_t)y1 * w + (size_t)x1) * 3;
            float texel[4];
            for (int k = 0; k < 3; k++) {
                const float c00 = (float)orc_unorm8(t00[k]) / 255.0f, c10 = (float)orc_unorm8(t10[k]) / 255.0f;
                const float c01 = (float)orc_unorm8(t01[k]) / 255.0f, c11 = (float)orc_unorm8(t11[k]) / 255.0f;
                const float top = c00 * (1.0f - a) + c10 * a, bot = c01 * (1.0f - a) + c11 * a;
                texel[k] = top * (1.0f - b) + bot * b;
            }
            {   /* alpha: 1 where the compute pass stored a texel, 0 (the fresh texture) beyond its workgroups (main.rs:452);
                 * through the same expression — exactly 1 when all four are 1 */
                const uint32_t cw = w & ~7u, chh = h & ~7u;
                const float a00 = ((uint32_t)x0 < cw && (uint32_t)y0 < chh) ? 1.0f : 0.0f, a10 = ((uint32_t)x1 < cw && (uint32_t)y0 < chh) ? 1.0f : 0.0f;
                const float a01 = ((uint32_t)x0 < cw && (uint32_t)y1 < chh) ? 1.0f : 0.0f, a11 = ((uint32_t)x1 < cw && (uint32_t)y1 < chh) ? 1.0f : 0.0f;
                const float top = a00 * (1.0f - a) + a10 * a, bot = a01 * (1.0f - a) + a11 * a;
                texel[3] = top * (1.0f - b) + bot * b;
            }
            const float cc[4] = {ch->color[0], ch->color[1], ch->color[2], 1.0f};
            uint8_t *o = rgba8 + ((size_t)sy * screen_w + sx) * 4;
            for (int k = 0; k < 4; k++) o[k] = (uint8_t)orc_unorm8(texel[k] * (1.0f - mask) + cc[k] * mask);  /* :60-63 */
        }
}

void orc_ray_sky(const orc_scene *scene, const float origin[3], const float dir[3], float *rgb) {
    v3 c = ray_sky(scene, V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* create_ray_from_screen, ray_tracer.wgsl:159-171.
 * WGSL `v * M` is the row-vector product: (v*M)[i] = dot(v, M[i]) with M[i] the i-th COLUMN. */
static void create_ray_from_screen(const orc_scene *s, int32_t sx, int32_t sy, v3 *origin, v3 *dir) {
    const orc_cam_data *c = &s->cam;
    float x = ((float)sx * 2.0f) / c->proj_size[0] - 1.0f;
    float y = ((float)sy * 2.0f) / c->proj_size[1] - 1.0f;
    float clip[4] = {x, -y, -1.0f, 1.0f};
    float e0[2];
    for (int i = 0; i < 2; i++) {
        const float *col = &c->inv_proj_mat[i * 4];
        e0[i] = clip[0] * col[0] + clip[1] * col[1] + clip[2] * col[2] + clip[3] * col[3];
    }
    float eye[4] = {e0[0], e0[1], -1.0f, 0.0f};
    float w[3];
    for (int i = 0; i < 3; i++) {
        const float *col = &c->inv_view_mat[i * 4];
        w[i] = eye[0] * col[0] + eye[1] * col[1] + eye[2] * col[2] + eye[3] * col[3];
    }
    *dir = orc_normalize(V3(w[0], w[1], w[2]));
    *origin = V3(c->pos[0] - (float)s->world.min[0], c->pos[1] - (float)s->world.min[1],
                 c->pos[2] - (float)s->world.min[2]);
}

static uint32_t id_word(const hit_result *r) {
    uint32_t id = r->voxel & ORC_ID_VOXEL_MASK;
    if (r->hit) id |= ORC_ID_HIT;
    if (r->norm.x != 0.0f) id |= ORC_ID_NX;
    if (r->norm.y != 0.0f) id |= ORC_ID_NY;
    if (r->norm.z != 0.0f) id |= ORC_ID_NZ;
    if (r->water_dist != 0.0f) id |= ORC_ID_WATER;
    return id;
}

/* overlay_color + ray_color, ray_tracer.wgsl:127-142 */
static v3 ray_color_from(const orc_scene *s, v3 origin, v3 dir, const hit_result *rs) {
    v3 sky = ray_sky(s, origin, dir);
    float fh = rs->hit ? 1.0f : 0.0f, fm = rs->hit ? 0.0f : 1.0f;
    v3 color = V3(rs->color.x * fh + sky.x * fm, rs->color.y * fh + sky.y * fm, rs->color.z * fh + sky.z * fm);
    if (rs->water_dist != 0.0f) {
        float factor = orc_clamp(rs->water_dist / 14.0f, 0.8f, 1.0f);
        color.x = color.x * (1.0f - factor) + 0.2f * factor;
        color.y = color.y * (1.0f - factor) + 0.5f * factor;
        color.z = color.z * (1.0f - factor) + 1.0f * factor;
    }
    return color;
}

uint32_t orc_ray_world(const orc_scene *scene, const float origin[3], const float dir[3],
                       float *color, float *out) {
    hit_result r = ray_world(scene, V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]));
    if (color) { color[0] = r.color.x; color[1] = r.color.y; color[2] = r.color.z; }
    if (out) {
        out[0] = r.pos.x; out[1] = r.pos.y; out[2] = r.pos.z;
        out[3] = r.norm.x; out[4] = r.norm.y; out[5] = r.norm.z;
        out[6] = r.water_dist; out[7] = (float)r.iter_count;
    }
    return id_word(&r);
}

/* ------------------------------------------------------------------------------------------ */
/* Build-defined secondary rays                                                                */
/* ------------------------------------------------------------------------------------------ */

/* A shadow ray is launched from every primary hit that stopped on a solid voxel (not from misses,
 * not from 500-step exhaustion in air/water).  Origin: hit.pos + norm * ORC_SHADOW_BIAS — hit.pos
 * lies 0.001*|dir_axis| INSIDE the solid voxel on the entry axes (ray_tracer.wgsl:274-283; the
 * stale path tracer starts its bounce there and self-hits), so stepping back along the
 * entry normal by 0.002 puts the origin >= 0.001 outside the face.  Direction: towards sun_pos from
 * that origin.  The ray is marched by the same ray_world; occluded = its hit flag (so liquids are
 * transparent and exhaustion counts as occluded).  Shading: the pixel's final colour (after the water
 * overlay) is multiplied by ORC_SHADOW_FACTOR when occluded. */
static int shadow_ray(const orc_scene *s, const hit_result *prim, hit_result *sh_out) {
    v3 so = V3(prim->pos.x + prim->norm.x * ORC_SHADOW_BIAS, prim->pos.y + prim->norm.y * ORC_SHADOW_BIAS,
               prim->pos.z + prim->norm.z * ORC_SHADOW_BIAS);
    v3 sd = orc_normalize(V3(s->settings.sun_pos[0] - (float)s->world.min[0] - so.x,
                             s->settings.sun_pos[1] - (float)s->world.min[1] - so.y,
                             s->settings.sun_pos[2] - (float)s->world.min[2] - so.z));
    *sh_out = ray_world(s, so, sd);
    return sh_out->hit;
}

static int is_solid_hit(const orc_scene *s, const hit_result *r) {
    return r->hit && r->voxel != 0u && mat_at(s, r->voxel)->is_liquid != 1u;
}

typedef struct {
    uint32_t id;
    v3 color;
    uint32_t steps_primary, steps_shadow;
    uint64_t visits_primary, visits_shadow;
    int shadow_launched;
    hit_result prim;
    v3 dir;
} pixel_result;

static pixel_result trace_pixel(const orc_scene *s, int mode, uint32_t px, uint32_t py) {
    pixel_result pr;
    memset(&pr, 0, sizeof pr);
    v3 origin, dir;
    create_ray_from_screen(s, (int32_t)px, (int32_t)py, &origin, &dir);
    hit_result rs = ray_world(s, origin, dir);
    v3 color = ray_color_from(s, origin, dir, &rs);
    uint32_t id = id_word(&rs);
    pr.steps_primary = rs.iter_count;
    pr.visits_primary = rs.node_visits;
    if (mode == ORC_MODE_PRIMARY_SHADOW && is_solid_hit(s, &rs)) {
        hit_result sh;
        int occluded = shadow_ray(s, &rs, &sh);
        id |= ORC_ID_SHADOW_RAY;
        pr.shadow_launched = 1;
        pr.steps_shadow = sh.iter_count;
        pr.visits_shadow = sh.node_visits;
        if (occluded) {
            id |= ORC_ID_SHADOWED;
            color.x *= ORC_SHADOW_FACTOR; color.y *= ORC_SHADOW_FACTOR; color.z *= ORC_SHADOW_FACTOR;
        }
    }
    pr.id = id;
    pr.color = color;
    pr.prim = rs;
    pr.dir = dir;
    return pr;
}

/* ------------------------------------------------------------------------------------------ */
/* Build-defined path trace (ORC_MODE_PATH), after the stale path_tracer.wgsl:56-76,149-194        */
/* ------------------------------------------------------------------------------------------ */
/* The reference's path tracer is compiled but never dispatched and no longer matches the host's node /
 * Material layouts; this mode keeps its structure — PCG state seeded per pixel, Box-Muller
 * direction, dir = normalize(mix(reflect, normalize(norm + rand_dir), scatter)), throughput *= colour,
 * light += sky * throughput on a miss, at most settings.max_ray_bounces segments — on top of the LIVE
 * ray_world (ray_tracer.wgsl:182-316), with two fixes: the bounce starts at hit.pos + norm * ORC_SHADOW_BIAS
 * (the stale shader starts inside the voxel it just hit and self-intersects), and log(0) is avoided by
 * clamping the uniform sample to 1e-10.  log and cos are spelled out in +,-,*,/ so that gcc and hipcc
 * produce the same bits (libm and ocml do not agree to the last ulp, and a one-ulp different bounce
 * direction eventually hits a different voxel).  The live Material has no emission, so light enters only
 * through the sky; water is transparent to path segments and does not tint them. */

static inline float orc_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ln(x) for normal x > 0: x = m * 2^e with m in (sqrt(1/2), sqrt(2)], ln m = 2 atanh((m-1)/(m+1)) */
static float orc_log(float x) {
    uint32_t b = orc_f2bits(x);
    int32_t e = (int32_t)(b >> 23) - 127;
    float m = orc_bits2f((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = z * (0.333333343f + z * (0.2f + z * (0.142857149f + z * (0.111111112f + z * 0.0909090936f))));
    return (float)e * 0.693147182f + (s + s * p) * 2.0f;
}

/* cos(2*pi*u) for u in [0,1]: quadrant + Taylor polynomials on [0, pi/2) */
static float orc_cos2pi(float u) {
    float t = u * 4.0f;
    float q = floorf(t);
    float a = (t - q) * 1.57079637f;
    float a2 = a * a;
    float sn = a * (1.0f + a2 * (-0.166666672f + a2 * (0.00833333377f + a2 * (-0.000198412701f + a2 * (2.75573188e-06f + a2 * -2.50521079e-08f)))));
    float cs = 1.0f + a2 * (-0.5f + a2 * (0.0416666679f + a2 * (-0.00138888892f + a2 * (2.48015876e-05f + a2 * (-2.75573199e-07f + a2 * 2.08767559e-09f)))));
    switch ((int32_t)q & 3) {
        case 0: return cs;
        case 1: return -sn;
        case 2: return -cs;
        default: return sn;
    }
}

/* rng_next_norm, path_tracer.wgsl:62-66 */
static float rng_next_norm(uint32_t *state) {
    float u1 = orc_rng_next(state);
    float u2 = orc_rng_next(state);
    if (u2 < 1.0e-10f) u2 = 1.0e-10f;
    float rho = sqrtf(-2.0f * orc_log(u2));
    return rho * orc_cos2pi(u1);
}

/* rng_next_dir, path_tracer.wgsl:67-72 */
static v3 rng_next_dir(uint32_t *state) {
    float x = rng_next_norm(state);
    float y = rng_next_norm(state);
    float z = rng_next_norm(state);
    return orc_normalize(V3(x, y, z));
}

float orc_test_log(float x) { return orc_log(x); }
float orc_test_cos2pi(float u) { return orc_cos2pi(u); }
void orc_test_rng_dir(uint32_t *state, float *out3) { v3 d = rng_next_dir(state); out3[0] = d.x; out3[1] = d.y; out3[2] = d.z; }

typedef struct {
    uint32_t id;
    v3 light;
    uint32_t segments, steps;
    uint64_t visits;
    uint32_t steps_primary;
    uint64_t visits_primary;
    int primary_hit;
} path_result;

/* ray_color, path_tracer.wgsl:149-194 (see the block comment above for the deliberate differences) */
static path_result trace_path(const orc_scene *s, uint32_t px, uint32_t py, uint32_t rng) {
    path_result pr;
    memset(&pr, 0, sizeof pr);
    v3 origin, dir;
    create_ray_from_screen(s, (int32_t)px, (int32_t)py, &origin, &dir);
    v3 thr = V3(1.0f, 1.0f, 1.0f);
    for (uint32_t bounce = 0; bounce < s->settings.max_ray_bounces; bounce++) {
        hit_result rs = ray_world(s, origin, dir);
        pr.segments += 1;
        pr.steps += rs.iter_count;
        pr.visits += rs.node_visits;
        if (bounce == 0) {
            pr.id = id_word(&rs);
            pr.steps_primary = rs.iter_count;
            pr.visits_primary = rs.node_visits;
            pr.primary_hit = rs.hit;
        }
        if (!rs.hit) {
            v3 sky = ray_sky(s, origin, dir);
            pr.light.x += sky.x * thr.x;
            pr.light.y += sky.y * thr.y;
            pr.light.z += sky.z * thr.z;
            break;
        }
        float d = orc_dot(rs.norm, dir);
        v3 spec = V3(dir.x - 2.0f * rs.norm.x * d, dir.y - 2.0f * rs.norm.y * d, dir.z - 2.0f * rs.norm.z * d);
        v3 rd = rng_next_dir(&rng);
        v3 sc = orc_normalize(V3(rs.norm.x + rd.x, rs.norm.y + rd.y, rs.norm.z + rd.z));
        float scatter = mat_at(s, rs.voxel)->scatter;
        v3 nd = orc_normalize(V3(orc_mix(spec.x, sc.x, scatter), orc_mix(spec.y, sc.y, scatter), orc_mix(spec.z, sc.z, scatter)));
        thr.x *= rs.color.x; thr.y *= rs.color.y; thr.z *= rs.color.z;
        origin = V3(rs.pos.x + rs.norm.x * ORC_SHADOW_BIAS, rs.pos.y + rs.norm.y * ORC_SHADOW_BIAS, rs.pos.z + rs.norm.z * ORC_SHADOW_BIAS);
        dir = nd;
    }
    return pr;
}

/* seed of sample s of pixel (px,py): path_tracer.wgsl:328 plus the per-sample stride of SURVEY §8d */
static uint32_t path_seed(uint32_t px, uint32_t py, uint32_t w, uint32_t h, uint32_t sample, uint32_t seed) {
    return py * w + px + sample * (w * h) + seed * 0x9E3779B9u;
}

uint32_t orc_trace_pixel(const orc_scene *scene, int mode, uint32_t px, uint32_t py,
                         float *rgb, float *dir, float *out) {
    pixel_result pr = trace_pixel(scene, mode, px, py);
    if (rgb) { rgb[0] = pr.color.x; rgb[1] = pr.color.y; rgb[2] = pr.color.z; }
    if (dir) { dir[0] = pr.dir.x; dir[1] = pr.dir.y; dir[2] = pr.dir.z; }
    if (out) {
        out[0] = pr.prim.pos.x; out[1] = pr.prim.pos.y; out[2] = pr.prim.pos.z;
        out[3] = pr.prim.norm.x; out[4] = pr.prim.norm.y; out[5] = pr.prim.norm.z;
        out[6] = pr.prim.water_dist; out[7] = (float)pr.prim.iter_count;
    }
    return pr.id;
}

void orc_render(const orc_scene *scene, int mode, uint32_t w, uint32_t h,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                float *rgb, uint32_t *ids, uint32_t *steps, orc_stats *stats,
                int threads, uint32_t spp, uint32_t seed) {
    uint64_t t_prim = 0, t_sec = 0, t_hits = 0, t_steps = 0, t_visits = 0, t_psteps = 0, t_pvisits = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    else omp_set_num_threads(omp_get_num_procs());
#else
    (void)threads;
#endif
    /* main.rs:452 dispatches result_tex_size / 8 workgroups of 8x8 invocations per axis (integer division) with no bounds
     * test in the shader: the pixels beyond the last whole 8x8 tile are never invoked and keep the fresh texture's zeros
     * (the result texture is 1080 rows at the window's aspect, main.rs:257-262 — any width).  The caller passes zeroed
     * arrays; NDC still comes from the full w x h (cam.proj_size). */
    if (x1 > (w & ~7u)) x1 = w & ~7u;
    if (y1 > (h & ~7u)) y1 = h & ~7u;
    if (x0 > x1) x0 = x1;
    if (y0 > y1) y0 = y1;
    /* 8-row bands, dynamic schedule. */
    int32_t nbands = (int32_t)((y1 - y0 + 7u) / 8u);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : t_prim, t_sec, t_hits, t_steps, t_visits, t_psteps, t_pvisits)
    for (int32_t band = 0; band < nbands; band++) {
        uint32_t ya = y0 + (uint32_t)band * 8u, yb = ya + 8u > y1 ? y1 : ya + 8u;
        for (uint32_t py = ya; py < yb; py++) {
            for (uint32_t px = x0; px < x1; px++) {
                size_t o = (size_t)py * w + px;
                if (mode == ORC_MODE_PATH) {
                    v3 sum = V3(0.0f, 0.0f, 0.0f);
                    uint32_t id = 0, sp = 0, ss = 0;
                    const uint32_t nspp = spp ? spp : 1u;
                    for (uint32_t sm = 0; sm < nspp; sm++) {
                        path_result p = trace_path(scene, px, py, path_seed(px, py, w, h, sm, seed));
                        sum.x += p.light.x; sum.y += p.light.y; sum.z += p.light.z;
                        if (sm == 0) { id = p.id; sp = p.steps_primary; t_hits += (uint64_t)(p.primary_hit != 0); }
                        if (sm == 0) ss = p.steps - p.steps_primary;
                        t_prim += 1;
                        t_sec += p.segments - 1;
                        t_steps += p.steps;
                        t_visits += p.visits;
                        t_psteps += p.steps_primary;
                        t_pvisits += p.visits_primary;
                    }
                    if (rgb) { rgb[o * 3 + 0] = sum.x / (float)nspp; rgb[o * 3 + 1] = sum.y / (float)nspp; rgb[o * 3 + 2] = sum.z / (float)nspp; }
                    if (ids) ids[o] = id;
                    if (steps) steps[o] = sp | (ss << 16);
                    continue;
                }
                pixel_result pr = trace_pixel(scene, mode, px, py);
                if (rgb) { rgb[o * 3 + 0] = pr.color.x; rgb[o * 3 + 1] = pr.color.y; rgb[o * 3 + 2] = pr.color.z; }
                if (ids) ids[o] = pr.id;
                if (steps) steps[o] = pr.steps_primary | (pr.steps_shadow << 16);
                t_prim += 1;
                t_sec += (uint64_t)pr.shadow_launched;
                t_hits += (uint64_t)(pr.prim.hit != 0);
                t_steps += pr.steps_primary + pr.steps_shadow;
                t_visits += pr.visits_primary + pr.visits_shadow;
                t_psteps += pr.steps_primary;
                t_pvisits += pr.visits_primary;
            }
        }
    }
    if (stats) {
        stats->primary_rays = t_prim; stats->secondary_rays = t_sec; stats->hits = t_hits;
        stats->steps = t_steps; stats->node_visits = t_visits;
        stats->primary_steps = t_psteps; stats->primary_node_visits = t_pvisits;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Camera: CamData::create (clientdesktop/src/graphics/mod.rs:92-111) over glam 0.31.0          */
/* ------------------------------------------------------------------------------------------ */
/* glam 0.31.0 (Cargo.lock:905-906) is a crates.io dependency, not vendored under /root/reference.
 * Its published algorithms, restated: Mat4 is column-major; from_rotation_x(a) = cols
 * X,(0,c,s,0),(0,-s,c,0),W; from_rotation_y(a) = (c,0,-s,0),Y,(s,0,c,0),W; from_rotation_z(a) =
 * (c,s,0,0),(-s,c,0,0),Z,W; from_translation(t) = X,Y,Z,(t,1); perspective_rh(fovy,aspect,n,f):
 * h = cos(fovy/2)/sin(fovy/2), w = h/aspect, r = f/(n-f), cols (w,0,0,0),(0,h,0,0),(0,0,r,-1),
 * (0,0,r*n,0); a*b column j = a.x_axis*b[j].x + a.y_axis*b[j].y + a.z_axis*b[j].z + a.w_axis*b[j].w
 * (left-to-right); inverse() = cofactor expansion / determinant.  f32::to_radians = x * (PI/180).
 * Differences to the real crate are bounded by sinf/cosf libm-vs-Rust-std and the inverse's
 * association order: a few ulp in the matrix entries, documented in DESIGN.md. */

static void m4_mul(const float *a, const float *b, float *o) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r[j * 4 + i] = a[0 * 4 + i] * b[j * 4 + 0] + a[1 * 4 + i] * b[j * 4 + 1] +
                           a[2 * 4 + i] * b[j * 4 + 2] + a[3 * 4 + i] * b[j * 4 + 3];
    memcpy(o, r, sizeof r);
}

static void m4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }

static void m4_inverse(const float *m, float *o) {
    /* cofactor expansion in double, rounded once — any correct inverse is within ulps of glam's */
    double a[16], inv[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    inv[0] = a[5]*a[10]*a[15] - a[5]*a[11]*a[14] - a[9]*a[6]*a[15] + a[9]*a[7]*a[14] + a[13]*a[6]*a[11] - a[13]*a[7]*a[10];
    inv[4] = -a[4]*a[10]*a[15] + a[4]*a[11]*a[14] + a[8]*a[6]*a[15] - a[8]*a[7]*a[14] - a[12]*a[6]*a[11] + a[12]*a[7]*a[10];
    inv[8] = a[4]*a[9]*a[15] - a[4]*a[11]*a[13] - a[8]*a[5]*a[15] + a[8]*a[7]*a[13] + a[12]*a[5]*a[11] - a[12]*a[7]*a[9];
    inv[12] = -a[4]*a[9]*a[14] + a[4]*a[10]*a[13] + a[8]*a[5]*a[14] - a[8]*a[6]*a[13] - a[12]*a[5]*a[10] + a[12]*a[6]*a[9];
    inv[1] = -a[1]*a[10]*a[15] + a[1]*a[11]*a[14] + a[9]*a[2]*a[15] - a[9]*a[3]*a[14] - a[13]*a[2]*a[11] + a[13]*a[3]*a[10];
    inv[5] = a[0]*a[10]*a[15] - a[0]*a[11]*a[14] - a[8]*a[2]*a[15] + a[8]*a[3]*a[14] + a[12]*a[2]*a[11] - a[12]*a[3]*a[10];
    inv[9] = -a[0]*a[9]*a[15] + a[0]*a[11]*a[13] + a[8]*a[1]*a[15] - a[8]*a[3]*a[13] - a[12]*a[1]*a[11] + a[12]*a[3]*a[9];
    inv[13] = a[0]*a[9]*a[14] - a[0]*a[10]*a[13] - a[8]*a[1]*a[14] + a[8]*a[2]*a[13] + a[12]*a[1]*a[10] - a[12]*a[2]*a[9];
    inv[2] = a[1]*a[6]*a[15] - a[1]*a[7]*a[14] - a[5]*a[2]*a[15] + a[5]*a[3]*a[14] + a[13]*a[2]*a[7] - a[13]*a[3]*a[6];
    inv[6] = -a[0]*a[6]*a[15] + a[0]*a[7]*a[14] + a[4]*a[2]*a[15] - a[4]*a[3]*a[14] - a[12]*a[2]*a[7] + a[12]*a[3]*a[6];
    inv[10] = a[0]*a[5]*a[15] - a[0]*a[7]*a[13] - a[4]*a[1]*a[15] + a[4]*a[3]*a[13] + a[12]*a[1]*a[7] - a[12]*a[3]*a[5];
    inv[14] = -a[0]*a[5]*a[14] + a[0]*a[6]*a[13] + a[4]*a[1]*a[14] - a[4]*a[2]*a[13] - a[12]*a[1]*a[6] + a[12]*a[2]*a[5];
    inv[3] = -a[1]*a[6]*a[11] + a[1]*a[7]*a[10] + a[5]*a[2]*a[11] - a[5]*a[3]*a[10] - a[9]*a[2]*a[7] + a[9]*a[3]*a[6];
    inv[7] = a[0]*a[6]*a[11] - a[0]*a[7]*a[10] - a[4]*a[2]*a[11] + a[4]*a[3]*a[10] + a[8]*a[2]*a[7] - a[8]*a[3]*a[6];
    inv[11] = -a[0]*a[5]*a[11] + a[0]*a[7]*a[9] + a[4]*a[1]*a[11] - a[4]*a[3]*a[9] - a[8]*a[1]*a[7] + a[8]*a[3]*a[5];
    inv[15] = a[0]*a[5]*a[10] - a[0]*a[6]*a[9] - a[4]*a[1]*a[10] + a[4]*a[2]*a[9] + a[8]*a[1]*a[6] - a[8]*a[2]*a[5];
    double det = a[0]*inv[0] + a[1]*inv[4] + a[2]*inv[8] + a[3]*inv[12];
    double rdet = 1.0 / det;
    for (int i = 0; i < 16; i++) o[i] = (float)(inv[i] * rdet);
}

static inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); }

void orc_cam_data_create(const float rot_deg[3], const float eye[3], float fov_deg,
                         const float proj_size[2], orc_cam_data *out) {
    float t[16], rx[16], ry[16], rz[16], m[16];
    m4_identity(t); t[12] = eye[0]; t[13] = eye[1]; t[14] = eye[2];
    float a = to_radians(rot_deg[0]);
    float s = sinf(a), c = cosf(a);
    m4_identity(rx); rx[5] = c; rx[6] = s; rx[9] = -s; rx[10] = c;
    a = -to_radians(rot_deg[1]); s = sinf(a); c = cosf(a);
    m4_identity(ry); ry[0] = c; ry[2] = -s; ry[8] = s; ry[10] = c;
    a = to_radians(rot_deg[2]); s = sinf(a); c = cosf(a);
    m4_identity(rz); rz[0] = c; rz[1] = s; rz[4] = -s; rz[5] = c;
    m4_mul(t, rx, m); m4_mul(m, ry, m); m4_mul(m, rz, m);

    float fovy = to_radians(fov_deg);
    float aspect = proj_size[0] / proj_size[1];
    float z_near = 0.001f, z_far = 1000.0f;
    float sf = sinf(0.5f * fovy), cf = cosf(0.5f * fovy);
    float hh = cf / sf, ww = hh / aspect, r = z_far / (z_near - z_far);
    float p[16];
    memset(p, 0, sizeof p);
    p[0] = ww; p[5] = hh; p[10] = r; p[11] = -1.0f; p[14] = r * z_near;

    memset(out, 0, sizeof *out);
    out->pos[0] = eye[0]; out->pos[1] = eye[1]; out->pos[2] = eye[2];
    memcpy(out->inv_view_mat, m, sizeof m);
    m4_inverse(p, out->inv_proj_mat);
    out->proj_size[0] = proj_size[0]; out->proj_size[1] = proj_size[1];
}

/* axis_rot_to_ray, common/src/math.rs:131-146 */
void orc_axis_rot_to_ray(const float rot[3], float *out3) {
    float r = cosf(rot[0]);
    out3[0] = r * -sinf(rot[1]);
    out3[2] = r * -cosf(rot[1]);
    out3[1] = -sinf(rot[0]);
}

/* rng_next, path_tracer.wgsl:56-61 */
float orc_rng_next(uint32_t *state) {
    *state = *state * 747796405u + 2891336453u;
    uint32_t result = ((*state >> ((*state >> 28u) + 4u)) ^ *state) * 277803737u;
    result = (result >> 22u) ^ result;
    return (float)result / 4294967295.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* NodeAlloc: common/src/world/mod.rs:213-313                                                  */
/* ------------------------------------------------------------------------------------------ */

static void na_push(orc_node_alloc *a, uint32_t s, uint32_t e) {
    if (a->n_free == a->cap_free) {
        a->cap_free = a->cap_free ? a->cap_free * 2 : 8;
        a->free_start = (uint32_t *)realloc(a->free_start, a->cap_free * sizeof(uint32_t));
        a->free_end = (uint32_t *)realloc(a->free_end, a->cap_free * sizeof(uint32_t));
    }
    a->free_start[a->n_free] = s;
    a->free_end[a->n_free] = e;
    a->n_free++;
}

/* NodeAlloc::new, :224-231 */
void orc_node_alloc_init(orc_node_alloc *a, uint32_t used_start, uint32_t used_end,
                         uint32_t free_start, uint32_t free_end) {
    memset(a, 0, sizeof *a);
    (void)free_start; /* assert_eq!(used.end, free.start) */
    a->range_start = used_start;
    a->range_end = free_end;
    na_push(a, used_end, free_end);
    a->last_used_addr = used_end - 1;
}

void orc_node_alloc_destroy(orc_node_alloc *a) {
    free(a->free_start); free(a->free_end);
    memset(a, 0, sizeof *a);
}

/* find_next, :255-273: the lowest-address span holding >= 8 nodes */
static int na_find_next(const orc_node_alloc *a, uint32_t *which) {
    uint32_t earliest = 0, earliest_addr = UINT32_MAX;
    for (uint32_t i = 0; i < a->n_free; i++) {
        uint32_t s = a->free_start[i], e = a->free_end[i];
        uint32_t len = e > s ? e - s : 0; /* saturating_sub */
        if (len < 8) continue;
        if (s < earliest_addr) { earliest_addr = s; earliest = i; }
    }
    if (earliest_addr == UINT32_MAX) return 0;
    *which = earliest;
    return 1;
}

/* next, :275-286 */
int orc_node_alloc_next(orc_node_alloc *a, uint32_t *addr) {
    uint32_t i;
    if (!na_find_next(a, &i)) return 0;
    uint32_t result = a->free_start[i];
    a->free_start[i] += 8;
    if (a->free_start[i] + 1 == a->free_end[i]) {
        /* Vec::remove keeps order */
        memmove(a->free_start + i, a->free_start + i + 1, (a->n_free - i - 1) * sizeof(uint32_t));
        memmove(a->free_end + i, a->free_end + i + 1, (a->n_free - i - 1) * sizeof(uint32_t));
        a->n_free--;
    }
    if (result + 7 > a->last_used_addr) a->last_used_addr = result + 7;
    *addr = result;
    return 1;
}

/* peek, :288-291 */
int orc_node_alloc_peek(const orc_node_alloc *a, uint32_t *addr) {
    uint32_t i;
    if (!na_find_next(a, &i)) return 0;
    *addr = a->free_start[i];
    return 1;
}

/* free, :293-307 */
void orc_node_alloc_free(orc_node_alloc *a, uint32_t addr) {
    uint32_t rs = addr, re = addr + 8;
    for (uint32_t i = 0; i < a->n_free; i++) {
        if (a->free_start[i] == re) { a->free_start[i] -= 8; return; }
        if (a->free_end[i] == rs) { a->free_end[i] += 8; return; }
    }
    na_push(a, rs, re);
}

/* move_end, :234-242 */
void orc_node_alloc_move_end(orc_node_alloc *a, uint32_t new_end) {
    for (uint32_t i = 0; i < a->n_free; i++) {
        if (a->free_end[i] == a->range_end) { a->free_end[i] = new_end; break; }
    }
    a->range_end = new_end;
}

/* ------------------------------------------------------------------------------------------ */
/* Svo: common/src/world/mod.rs:323-471                                                        */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint32_t idx, depth, size; float cx, cy, cz; } svo_found;

/* find_node, :366-395 */
static svo_found svo_find(const uint16_t *nodes, uint32_t root, uint32_t svo_size,
                          const uint32_t pos[3], uint32_t max_depth) {
    uint32_t size = svo_size, idx = root, depth = 0;
    float c = (float)size * 0.5f;
    float cx = c, cy = c, cz = c;
    for (;;) {
        uint16_t node = nodes[idx];
        if (!(node & 0x8000u) || depth == max_depth) {
            svo_found f = {idx, depth, size, cx, cy, cz};
            return f;
        }
        size /= 2;
        uint32_t gx = (float)pos[0] >= cx, gy = (float)pos[1] >= cy, gz = (float)pos[2] >= cz;
        uint32_t child = gx | (gy << 1) | (gz << 2);
        idx = (uint32_t)(node & 0x7FFFu) + child;
        cx += (float)size * 0.5f * (float)((int)gx * 2 - 1);
        cy += (float)size * 0.5f * (float)((int)gy * 2 - 1);
        cz += (float)size * 0.5f * (float)((int)gz * 2 - 1);
        depth += 1;
    }
}

void orc_svo_find_node(const uint16_t *nodes, uint32_t root, uint32_t svo_size,
                       const uint32_t pos[3], uint32_t max_depth, uint32_t *out3, float *c) {
    svo_found f = svo_find(nodes, root, svo_size, pos, max_depth);
    out3[0] = f.idx; out3[1] = f.depth; out3[2] = f.size;
    if (c) { c[0] = f.cx; c[1] = f.cy; c[2] = f.cz; }
}

/* node_parent, :332-364 — walks from the root towards node_in.center, stops one level above it */
static int svo_parent(const uint16_t *nodes, uint32_t root, uint32_t svo_size,
                      const svo_found *in, svo_found *out) {
    if (in->depth == 0) return 0;
    uint32_t size = svo_size, idx = root, depth = 0;
    float c = (float)size * 0.5f;
    float cx = c, cy = c, cz = c;
    for (;;) {
        uint16_t node = nodes[idx];
        if (!(node & 0x8000u) || depth == in->depth - 1) {
            out->idx = idx; out->depth = depth; out->size = size; out->cx = cx; out->cy = cy; out->cz = cz;
            return 1;
        }
        size /= 2;
        uint32_t gx = in->cx >= cx, gy = in->cy >= cy, gz = in->cz >= cz;
        uint32_t child = gx | (gy << 1) | (gz << 2);
        idx = (uint32_t)(node & 0x7FFFu) + child;
        cx += (float)size * 0.5f * (float)((int)gx * 2 - 1);
        cy += (float)size * 0.5f * (float)((int)gy * 2 - 1);
        cz += (float)size * 0.5f * (float)((int)gz * 2 - 1);
        depth += 1;
    }
}

/* set_node, :397-459 */
int orc_svo_set_node(uint16_t *nodes, uint32_t root, uint32_t svo_size, const uint32_t pos[3],
                     uint16_t voxel, uint32_t target_depth, orc_node_alloc *alloc) {
    svo_found node = svo_find(nodes, root, svo_size, pos, target_depth);
    uint16_t parent_voxel = nodes[node.idx] & 0x7FFFu;
    if (parent_voxel == voxel) return 0;

    while (node.depth < target_depth) {
        uint32_t first_child;
        if (!orc_node_alloc_next(alloc, &first_child)) return 1;
        /* assert!(first_child < Voxel::MAX_VALUE) :416 — a chunk can address <= 32766 */
        if (first_child >= 32767u) return 2; /* the reference panics here */
        for (int i = 0; i < 8; i++) nodes[first_child + i] = parent_voxel;
        nodes[node.idx] = (uint16_t)(first_child | 0x8000u);
        node.size /= 2;
        uint32_t gx = (float)pos[0] >= node.cx, gy = (float)pos[1] >= node.cy, gz = (float)pos[2] >= node.cz;
        uint32_t child = gx | (gy << 1) | (gz << 2);
        node.idx = first_child + child;
        node.cx += (float)node.size * 0.5f * (float)((int)gx * 2 - 1);
        node.cy += (float)node.size * 0.5f * (float)((int)gy * 2 - 1);
        node.cz += (float)node.size * 0.5f * (float)((int)gz * 2 - 1);
        node.depth += 1;
    }
    nodes[node.idx] = voxel & 0x7FFFu;

    for (;;) {
        svo_found parent;
        if (!svo_parent(nodes, root, svo_size, &node, &parent)) break;
        node = parent;
        uint32_t parent_idx = node.idx;
        uint32_t idx = nodes[parent_idx] & 0x7FFFu;
        const uint16_t *ch = nodes + idx;
        int eq = ch[0] == ch[1] && ch[0] == ch[2] && ch[0] == ch[3] && ch[0] == ch[4] &&
                 ch[0] == ch[5] && ch[0] == ch[6] && ch[0] == ch[7];
        if (eq) {
            orc_node_alloc_free(alloc, idx);
            nodes[parent_idx] = voxel & 0x7FFFu;
        } else {
            break;
        }
    }
    return 0;
}

/* server/src/world/gen.rs:171-286, reduced to its SVO-building loop (columns x, then z, y ascending) */
uint32_t orc_build_chunk_by_set_node(const uint16_t *dense, uint16_t *nodes, uint32_t cap) {
    orc_node_alloc alloc;
    orc_node_alloc_init(&alloc, 0, 1, 1, cap);
    memset(nodes, 0, cap * sizeof(uint16_t));
    for (uint32_t x = 0; x < 32; x++)
        for (uint32_t z = 0; z < 32; z++)
            for (uint32_t y = 0; y < 32; y++) {
                uint16_t v = dense[x + 32 * (y + 32 * z)];
                if (v == 0) continue;
                uint32_t pos[3] = {x, y, z};
                if (orc_svo_set_node(nodes, 0, 32, pos, v, 5, &alloc)) {
                    orc_node_alloc_destroy(&alloc);
                    return 0;
                }
            }
    uint32_t used = alloc.last_used_addr + 1;
    orc_node_alloc_destroy(&alloc);
    return used;
}
