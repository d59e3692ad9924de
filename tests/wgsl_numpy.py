"""A second, independent restatement of the reference's live shader — clientdesktop/src/graphics/ray_tracer.wgsl,
entry point `update` (:173-180) — written from the WGSL text alone, vectorised over pixels in numpy binary32.

Test infrastructure: it exists to check the C oracle (oracle/vrt_oracle.c), which was written separately and in a different
shape (scalar C, one pixel at a time, with extra bookkeeping).  Two restatements that agree bit for bit on voxel ids, normals,
water flags and per-pixel iteration counts do not pin parity to the reference (nothing here can run WGSL), but a slip of the
pen in either would show.  Builtins follow the WGSL specification's formulas: normalize(v) = v / sqrt(dot(v, v)),
dot = x x + y y + z z left to right, mix(a, b, t) = a (1 - t) + b t, smoothstep = t t (3 - 2 t) with t = clamp(...),
clamp = min(max(e, lo), hi), sign(0) = 0, v * M = (dot(v, column_i))_i.

Every `# :NNN` is the shader line the statement restates.  Only finite, in-world cameras are handled (the oracle's
treatment of NaN / infinite operands is covered by its own known-answer tests)."""
import numpy as np

F = np.float32
ID_VOXEL_MASK = 0x7FFF
ID_HIT, ID_NX, ID_NY, ID_NZ, ID_WATER = 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20


def _floats(struct, name, n):
    return np.array(list(getattr(struct, name))[:n], dtype=F)


def _dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _normalize(v):
    length = np.sqrt(_dot3(v, v))
    return [v[0] / length, v[1] / length, v[2] / length]


def _clamp(e, lo, hi):
    return np.minimum(np.maximum(e, F(lo)), F(hi))


def _smoothstep(e0, e1, x):
    t = _clamp((x - F(e0)) / (F(e1) - F(e0)), 0.0, 1.0)
    return t * t * (F(3.0) - F(2.0) * t)


def _mix(a, b, t):
    return a * (F(1.0) - t) + b * t


def render_primary(nodes_u16, chunk_roots, materials, cam, settings, world, w, h):
    """-> rgb [h, w, 3] f32, id words [h, w] u32 (the oracle's layout: voxel | hit | normal axes | water), iterations [h, w]."""
    nodes_u16 = np.asarray(nodes_u16, dtype=np.uint16)
    # binding 6 is array<u32>: two 16-bit nodes per word, even index in the low half (:38-42; shader.rs:22-40 writes the
    # pool as little-endian u32 pairs)
    padded = np.concatenate([nodes_u16, np.zeros(nodes_u16.size & 1, dtype=np.uint16)])
    pairs = padded[0::2].astype(np.uint32) | (padded[1::2].astype(np.uint32) << np.uint32(16))
    roots = np.asarray(chunk_roots, dtype=np.uint32)
    mat_liquid = np.array([materials[i].is_liquid for i in range(256)], dtype=np.uint32)
    mat_color = np.array([list(materials[i].color) for i in range(256)], dtype=F)
    cam_pos = _floats(cam, "pos", 3)
    inv_view = _floats(cam, "inv_view_mat", 16).reshape(4, 4)    # [column][row]
    inv_proj = _floats(cam, "inv_proj_mat", 16).reshape(4, 4)
    proj = _floats(cam, "proj_size", 2)
    wmin = np.array(list(world.min)[:3], dtype=np.int32)
    wsize, wchunks = int(world.size), int(world.size_in_chunks)
    sky_color = _floats(settings, "sky_color", 3)
    sun_pos = _floats(settings, "sun_pos", 3)
    sun_intensity = F(settings.sun_intensity)
    show_steps = int(settings.show_step_count) == 1

    def get_node(idx):                                            # :38-42
        pair = pairs[idx >> 1]
        return (pair >> ((idx & 1) * 16).astype(np.uint32)) & np.uint32(0xFFFF)

    # ---- create_ray_from_screen (:159-171), for every invocation the dispatch makes: main.rs:452, tex_size / 8 workgroups ----
    cw, chh = (w // 8) * 8, (h // 8) * 8
    py, px = np.meshgrid(np.arange(chh, dtype=np.int32), np.arange(cw, dtype=np.int32), indexing="ij")
    px, py = px.ravel(), py.ravel()
    n = px.size
    x = (px.astype(F) * F(2.0)) / proj[0] - F(1.0)               # :160
    y = (py.astype(F) * F(2.0)) / proj[1] - F(1.0)               # :161
    clip = [x, -y, np.full(n, -1.0, dtype=F), np.full(n, 1.0, dtype=F)]   # :162
    eye0 = [clip[0] * inv_proj[i][0] + clip[1] * inv_proj[i][1] + clip[2] * inv_proj[i][2] + clip[3] * inv_proj[i][3] for i in range(4)]  # :163
    eye = [eye0[0], eye0[1], np.full(n, -1.0, dtype=F), np.zeros(n, dtype=F)]                                                           # :164
    wdir = [eye[0] * inv_view[i][0] + eye[1] * inv_view[i][1] + eye[2] * inv_view[i][2] + eye[3] * inv_view[i][3] for i in range(3)]
    d = _normalize(wdir)                                          # :165
    origin = [np.full(n, cam_pos[k] - F(wmin[k]), dtype=F) for k in range(3)]   # :168

    # ---- ray_world (:182-316) ----
    mask = [(d[k] >= 0).astype(F) for k in range(3)]              # :184
    imask = [F(1.0) - mask[k] for k in range(3)]                  # :185
    pos = [origin[k].copy() for k in range(3)]
    nudge = (pos[0] - np.floor(pos[0]) < F(0.001)) | (pos[1] - np.floor(pos[1]) < F(0.001)) | (pos[2] - np.floor(pos[2]) < F(0.001))   # :188
    for k in range(3):
        pos[k] = np.where(nudge, pos[k] + F(0.001) * d[k], pos[k])   # :189
    wmax = F(0.0) + F(wsize)                                      # :193
    outside = (pos[0] <= 0) | (pos[1] <= 0) | (pos[2] <= 0) | (pos[0] >= wmax) | (pos[1] >= wmax) | (pos[2] >= wmax)   # :197
    with np.errstate(divide="ignore", invalid="ignore"):
        dx, dy, dz = d
        unit = [np.sqrt(F(1.0) + (dy / dx) * (dy / dx) + (dz / dx) * (dz / dx)),     # :208
                np.sqrt(F(1.0) + (dx / dy) * (dx / dy) + (dz / dy) * (dz / dy)),     # :209
                np.sqrt(F(1.0) + (dx / dz) * (dx / dz) + (dy / dz) * (dy / dz))]     # :210

    hit = np.zeros(n, dtype=bool)
    water_dist = np.zeros(n, dtype=F)
    voxel = np.zeros(n, dtype=np.uint32)
    norm = [np.zeros(n, dtype=F) for _ in range(3)]
    dew = np.full(n, -1.0, dtype=F)                               # :217
    total_len = np.zeros(n, dtype=F)
    iters = np.zeros(n, dtype=np.uint32)
    live = ~outside                                               # lanes still inside the while loop
    for _ in range(500):                                          # :221
        a = np.flatnonzero(live)
        if a.size == 0:
            break
        iters[a] += 1                                             # :222
        p = [pos[k][a] for k in range(3)]
        # find_node (:116-125)
        cc = [np.floor(p[k] / F(32.0)).astype(np.int32) for k in range(3)]
        cmin = [(cc[k] * 32).astype(F) for k in range(3)]
        cidx = cc[0].astype(np.uint32) + cc[1].astype(np.uint32) * np.uint32(wchunks) + cc[2].astype(np.uint32) * np.uint32(wchunks * wchunks)
        root = roots[cidx]
        # find_chunk_node (:76-114)
        center = [cmin[k] + F(16.0) for k in range(3)]
        size = np.full(a.size, 32.0, dtype=F)
        idx = np.zeros(a.size, dtype=np.uint32)
        descending = np.ones(a.size, dtype=bool)
        for depth in range(6):
            node = get_node(root + idx)
            stop = ((node >> 15) == 0) | (depth == 5)             # :88
            descending &= ~stop
            if not descending.any():
                break
            g = descending
            size = np.where(g, size * F(0.5), size)               # :98
            gt = [(p[k] >= center[k]).astype(np.int32) for k in range(3)]   # :100-104
            child = gt[0].astype(np.uint32) | (gt[1].astype(np.uint32) << 1) | (gt[2].astype(np.uint32) << 2)
            idx = np.where(g, (node & np.uint32(0x7FFF)) + child, idx)      # :106
            for k in range(3):
                center[k] = np.where(g, center[k] + (size * F(0.5)) * (gt[k] * 2 - 1).astype(F), center[k])   # :107-108
        nmin = [center[k] - size * F(0.5) for k in range(3)]     # :91
        nmax = [center[k] + size * F(0.5) for k in range(3)]     # :92
        v = get_node(root + idx) & np.uint32(0x7FFF)             # :225
        voxel[a] = v
        liquid = mat_liquid[np.minimum(v, 255)] == 1             # :227
        solid = (v != 0) & ~liquid                                # :229
        hit[a[solid]] = True
        live[a[solid]] = False
        # lanes that go on
        go = ~solid
        a, p = a[go], [p[k][go] for k in range(3)]
        liquid, nmin, nmax = liquid[go], [q[go] for q in nmin], [q[go] for q in nmax]
        leaving = ~liquid & (dew[a] != F(-1.0))                   # :232-237
        water_dist[a[leaving]] += total_len[a[leaving]] - dew[a[leaving]]
        dew[a[leaving]] = F(-1.0)
        entering = liquid & (dew[a] == F(-1.0))                   # :238-243
        dew[a[entering]] = total_len[a[entering]]
        ad = [((p[k] - nmin[k]) * imask[k][a] + (nmax[k] - p[k]) * mask[k][a]) * unit[k][a] for k in range(3)]   # :244-246
        zx, zy, zz = ad[0] == 0, ad[1] == 0, ad[2] == 0
        mn = np.minimum
        step = np.where(zx,
                        np.where(zy, ad[2], np.where(zz, ad[1], mn(ad[1], ad[2]))),
                        np.where(zy, np.where(zz, ad[0], mn(ad[0], ad[2])),
                                 np.where(zz, mn(ad[1], ad[0]), mn(ad[0], mn(ad[1], ad[2])))))   # :248-270
        total_len[a] = total_len[a] + step                        # :271
        sel = [(step == ad[k]).astype(F) for k in range(3)]
        for k in range(3):
            sgn = np.where(d[k][a] > 0, F(1.0), np.where(d[k][a] < 0, F(-1.0), d[k][a]))
            norm[k][a] = sel[k] * -sgn                            # :272
            pos[k][a] = p[k] + (d[k][a] * (step + F(0.001)) * sel[k] + d[k][a] * step * (F(1.0) - sel[k]))   # :274-283
        q = [pos[k][a] for k in range(3)]
        out = (q[0] < 0) | (q[1] < 0) | (q[2] < 0) | (q[0] >= wmax) | (q[1] >= wmax) | (q[2] >= wmax)          # :285
        still_wet = out & (dew[a] != F(-1.0))                     # :286-288
        water_dist[a[still_wet]] += total_len[a[still_wet]] - dew[a[still_wet]]
        live[a[out]] = False
    # ran out of iterations: "return not air OR max steps already" (:291) — a hit with whatever was looked up last
    hit[live] = True
    color = mat_color[np.minimum(voxel, 255)].copy()              # :296
    color[norm[0] != 0] *= F(0.5)                                 # :298-300
    color[norm[2] != 0] *= F(0.7)                                 # :301-303
    color[norm[1] == F(-1.0)] *= F(0.2)                           # :304-306
    wet_hit = hit & (dew != F(-1.0))                              # :307-309
    water_dist[wet_hit] += total_len[wet_hit] - dew[wet_hit]
    if show_steps:                                                # :311-314
        f = _clamp(iters.astype(F) / F(500.0), 0.0, 1.0)
        color = np.stack([f, f, f], axis=1)
    # a miss returns `result` as it stood: no normal, no material (:199, :289)
    for k in range(3):
        norm[k][~hit] = 0

    # ---- ray_sky (:144-157) ----
    g2s = _smoothstep(-0.01, 0.0, d[1])                           # :149
    with np.errstate(invalid="ignore"):
        grad_t = np.power(_smoothstep(0.0, 0.4, d[1]), F(0.35)).astype(F)   # :150
    horizon = np.array([1.0, 0.3, 0.0], dtype=F)
    grad = [_mix(horizon[k], sky_color[k], grad_t) for k in range(3)]       # :151
    sun_dir = _normalize([sun_pos[k] - F(wmin[k]) - origin[k] for k in range(3)])   # :152
    sun = ((_dot3(d, sun_dir) > F(1.0) - F(0.01)) & (g2s >= F(1.0))).astype(F)      # :154
    sky = [_mix(F(0.03), grad[k], g2s) + sun * sun_intensity for k in range(3)]     # :156

    # ---- ray_color (:131-142) ----
    fh, fm = hit.astype(F), (~hit).astype(F)
    rgb = np.stack([color[:, k] * fh + sky[k] * fm for k in range(3)], axis=1)      # :135
    wet = water_dist != 0                                         # :137
    factor = _clamp(water_dist / F(14.0), 0.8, 1.0)               # :138
    over = np.array([0.2, 0.5, 1.0], dtype=F)
    for k in range(3):
        rgb[:, k] = np.where(wet, rgb[:, k] * (F(1.0) - factor) + over[k] * factor, rgb[:, k])   # :128, :139

    ids = (voxel & np.uint32(ID_VOXEL_MASK)) * hit.astype(np.uint32)   # the oracle reports the voxel of a hit only
    ids |= np.where(hit, np.uint32(ID_HIT), np.uint32(0))
    ids |= np.where(norm[0] != 0, np.uint32(ID_NX), np.uint32(0)) | np.where(norm[1] != 0, np.uint32(ID_NY), np.uint32(0)) \
        | np.where(norm[2] != 0, np.uint32(ID_NZ), np.uint32(0))
    ids |= np.where(wet, np.uint32(ID_WATER), np.uint32(0))
    out_rgb = np.zeros((h, w, 3), dtype=F)
    out_ids = np.zeros((h, w), dtype=np.uint32)
    out_it = np.zeros((h, w), dtype=np.uint32)
    out_rgb[:chh, :cw] = rgb.reshape(chh, cw, 3)
    out_ids[:chh, :cw] = ids.reshape(chh, cw)
    out_it[:chh, :cw] = iters.reshape(chh, cw)
    return out_rgb, out_ids, out_it


def present(rgb, screen_size, color=(1.0, 1.0, 1.0, 0.33), style=2, size=5.0):
    """fs_main of clientdesktop/src/graphics/screen_shader.wgsl:43-65 over the rgba8unorm result texture, vectorised over the
    window's pixels: -> rgba8 [screen_h, screen_w, 4].  The texture is what `update` stored (ray_tracer.wgsl:179: clamp,
    x 255, round to nearest even; alpha 1) where the compute pass ran (main.rs:452: tex_size / 8 workgroups per axis) and
    the fresh texture's zeros elsewhere.  The sampler (texture.rs:31-44: ClampToEdge, mag Nearest, min Linear, lod clamped
    to [1, 1] on a one-level texture) samples with its minification filter at every size — the reading DESIGN.md section 2
    argues — i.e. bilinear on unnormalised coordinates u W - 1/2."""
    rgb = np.asarray(rgb, dtype=F)
    h, w, _ = rgb.shape
    sw, sh = screen_size
    cw, chh = (w // 8) * 8, (h // 8) * 8
    tex = np.zeros((h, w, 4), dtype=F)
    q = np.rint(np.clip(np.nan_to_num(rgb, nan=0.0), 0.0, 1.0) * F(255.0)).astype(F) / F(255.0)
    tex[:chh, :cw, :3] = q[:chh, :cw]
    tex[:chh, :cw, 3] = F(1.0)
    ssx, ssy = F(sw), F(sh)
    sy, sx = np.meshgrid(np.arange(sh), np.arange(sw), indexing="ij")
    u = (sx.astype(F) + F(0.5)) / ssx                               # tex_coord at the pixel centre (:33-40)
    v = (sy.astype(F) + F(0.5)) / ssy
    px, py = u * ssx, v * ssy                                        # :44
    cx, cy = ssx * F(0.5), ssy * F(0.5)                              # :45
    mask = np.zeros((sh, sw), dtype=F)
    if style == 1:                                                   # :48-50
        dx, dy = cx - px, cy - py
        mask = (np.sqrt(dx * dx + dy * dy) < F(size)).astype(F) * F(color[3])
    if style == 2:                                                   # :51-59
        dx, dy = np.abs(cx - px), np.abs(cy - py)
        wd = F(size) * F(0.25)
        mask = (((dx < F(size)) & (dy < wd)) | ((dy < F(size)) & (dx < wd))).astype(F) * F(color[3])
    ut, vt = u * F(w) - F(0.5), v * F(h) - F(0.5)
    fu, fv = np.floor(ut), np.floor(vt)
    a, b = (ut - fu)[..., None], (vt - fv)[..., None]
    x0 = np.clip(fu.astype(np.int64), 0, w - 1); x1 = np.clip(fu.astype(np.int64) + 1, 0, w - 1)
    y0 = np.clip(fv.astype(np.int64), 0, h - 1); y1 = np.clip(fv.astype(np.int64) + 1, 0, h - 1)
    top = tex[y0, x0] * (F(1.0) - a) + tex[y0, x1] * a
    bot = tex[y1, x0] * (F(1.0) - a) + tex[y1, x1] * a
    texel = top * (F(1.0) - b) + bot * b                             # :61
    cc = np.array([color[0], color[1], color[2], 1.0], dtype=F)
    out = texel * (F(1.0) - mask[..., None]) + cc * mask[..., None]  # :60-63
    return np.rint(np.clip(out, 0.0, 1.0) * F(255.0)).astype(np.uint8)
