//! The flat dumps tests/golden/export_scenes.py writes (its docstring is the format's specification): a scene — what the harness
//! uploads — and the arrays a fixture expects.  Little endian, no crate needed.

use std::{fs, io, path::Path};

pub const SCENE_MAGIC: &[u8; 8] = b"VRTSCN01";
pub const EXPECT_MAGIC: &[u8; 8] = b"VRTEXP01";
/// Sizes of the reference's `#[repr(C)]` uniform structs (clientdesktop/src/graphics/mod.rs:20-28, 82-91, 113-120, 132-143) =
/// include/vrt.h's; tests/test_wgpu_check.py holds these constants to the header.
pub const SCENE_HEADER_BYTES: usize = 32;
pub const CAM_DATA_BYTES: usize = 160;
pub const SETTINGS_BYTES: usize = 48;
pub const WORLD_DATA_BYTES: usize = 32;
pub const MATERIAL_BYTES: usize = 32;
pub const EXPECT_HEADER_BYTES: usize = 24;

pub struct Scene {
    pub width: u32,
    pub height: u32,
    /// NodeBuffer capacity in nodes (even): the harness allocates max_nodes / 2 words, as NodeBuffer::new does (shader.rs:9-16)
    pub max_nodes: u32,
    pub cam_data: Vec<u8>,
    pub settings: Vec<u8>,
    pub world_data: Vec<u8>,
    pub materials: Vec<u8>,
    pub chunk_roots: Vec<u8>,
    /// the pool's words up to its last non-zero one, as NodeBuffer::write packs them (two nodes per u32, shader.rs:22-40)
    pub node_words: Vec<u8>,
}

fn bad(msg: String) -> io::Error {
    io::Error::new(io::ErrorKind::InvalidData, msg)
}

fn u32_at(b: &[u8], at: usize) -> u32 {
    u32::from_le_bytes([b[at], b[at + 1], b[at + 2], b[at + 3]])
}

impl Scene {
    pub fn read(path: &Path) -> io::Result<Scene> {
        let b = fs::read(path)?;
        if b.len() < SCENE_HEADER_BYTES || &b[..8] != SCENE_MAGIC {
            return Err(bad(format!("{}: not a scene dump", path.display())));
        }
        let (width, height, max_nodes) = (u32_at(&b, 8), u32_at(&b, 12), u32_at(&b, 16));
        let (n_words, n_roots, n_mats) = (u32_at(&b, 20) as usize, u32_at(&b, 24) as usize, u32_at(&b, 28) as usize);
        let mut at = SCENE_HEADER_BYTES;
        let mut take = |n: usize| -> io::Result<Vec<u8>> {
            if at + n > b.len() {
                return Err(bad(format!("{}: truncated", path.display())));
            }
            let v = b[at..at + n].to_vec();
            at += n;
            Ok(v)
        };
        let cam_data = take(CAM_DATA_BYTES)?;
        let settings = take(SETTINGS_BYTES)?;
        let world_data = take(WORLD_DATA_BYTES)?;
        let materials = take(MATERIAL_BYTES * n_mats)?;
        let chunk_roots = take(4 * n_roots)?;
        let node_words = take(4 * n_words)?;
        if at != b.len() || width % 8 != 0 || height % 8 != 0 || max_nodes % 2 != 0 || 2 * n_words > max_nodes as usize {
            return Err(bad(format!("{}: inconsistent header", path.display())));
        }
        Ok(Scene { width, height, max_nodes, cam_data, settings, world_data, materials, chunk_roots, node_words })
    }
}

/// What a fixture recorded per pixel (row-major, `width * height` entries each; empty = the fixture does not hold that field).
#[derive(Default)]
pub struct Expect {
    pub width: u32,
    pub height: u32,
    pub shader_crc: u32,
    pub rgb: Vec<[f32; 3]>,
    pub hit: Vec<u8>,
    pub voxel: Vec<u32>,
    pub iters: Vec<u32>,
    pub norm: Vec<[f32; 3]>,
    pub water_dist: Vec<f32>,
    pub pos: Vec<[f32; 3]>,
}

impl Expect {
    pub fn read(path: &Path) -> io::Result<Expect> {
        let b = fs::read(path)?;
        if b.len() < EXPECT_HEADER_BYTES || &b[..8] != EXPECT_MAGIC {
            return Err(bad(format!("{}: not an expectation file", path.display())));
        }
        let mut e = Expect { width: u32_at(&b, 8), height: u32_at(&b, 12), shader_crc: u32_at(&b, 20), ..Default::default() };
        let fields = u32_at(&b, 16);
        let n = (e.width * e.height) as usize;
        let mut at = EXPECT_HEADER_BYTES;
        let need = |at: usize, bytes: usize| -> io::Result<()> {
            if at + bytes > b.len() { Err(bad(format!("{}: truncated", path.display()))) } else { Ok(()) }
        };
        let f32_at = |at: usize| f32::from_bits(u32_at(&b, at));
        let mut vec3s = |at: &mut usize| -> io::Result<Vec<[f32; 3]>> {
            need(*at, 12 * n)?;
            let v = (0..n).map(|i| [f32_at(*at + 12 * i), f32_at(*at + 12 * i + 4), f32_at(*at + 12 * i + 8)]).collect();
            *at += 12 * n;
            Ok(v)
        };
        if fields & 1 != 0 { e.rgb = vec3s(&mut at)?; }
        if fields & 2 != 0 {
            need(at, n)?;
            e.hit = b[at..at + n].to_vec();
            at += n + (4 - n % 4) % 4;
        }
        let mut words = |at: &mut usize| -> io::Result<Vec<u32>> {
            need(*at, 4 * n)?;
            let v = (0..n).map(|i| u32_at(&b, *at + 4 * i)).collect();
            *at += 4 * n;
            Ok(v)
        };
        if fields & 4 != 0 { e.voxel = words(&mut at)?; }
        if fields & 8 != 0 { e.iters = words(&mut at)?; }
        if fields & 16 != 0 { e.norm = vec3s(&mut at)?; }
        if fields & 32 != 0 { e.water_dist = words(&mut at)?.into_iter().map(f32::from_bits).collect(); }
        if fields & 64 != 0 { e.pos = vec3s(&mut at)?; }
        if at != b.len() {
            return Err(bad(format!("{}: trailing bytes", path.display())));
        }
        Ok(e)
    }
}

/// zlib's crc32 (reflected 0xEDB88320), as the fixtures stamp the shader text they were made from.
pub fn crc32(data: &[u8]) -> u32 {
    let mut c = !0u32;
    for &byte in data {
        c ^= byte as u32;
        for _ in 0..8 {
            c = if c & 1 != 0 { (c >> 1) ^ 0xEDB8_8320 } else { c >> 1 };
        }
    }
    !c
}
