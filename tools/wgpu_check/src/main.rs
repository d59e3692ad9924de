//! wgpu_check — the reference's shader through the reference's toolchain, against this repository's fixtures.
//!
//!     python tests/golden/export_scenes.py tools/wgpu_check/scenes          # (the repository's root; writes the dumps)
//!     cd tools/wgpu_check && cargo run --release -- --shader <checkout>/clientdesktop/src/graphics/ray_tracer.wgsl
//!
//! For every scene dump in `scenes/` (tests/golden/export_scenes.py documents the format) two passes run on the default adapter:
//! the shader AS IT STANDS, entry point `update`, whose rgba8unorm texture is compared with the fixture's colour quantised the same
//! way; and the shader with a debug entry point added (src/patch.rs), whose per-pixel HitResult, last voxel and iteration count are
//! compared with the fixture's arrays — integers exactly, floats by their bit patterns, with the largest difference in units in
//! the last place reported where bits differ.  The fixtures were made by an interpreter of WGSL written for the repository
//! (tests/wgsl_interp.py); a run of this binary without differences pins them — and with them the CPU oracle and the HIP kernels
//! that reproduce them bit for bit — to naga + the backend's own reading of the same text.  The two `oob_*` scenes are compared
//! with both readings of an index past an array's end (clamp / zero): which one matches is reported, not judged.
//!
//! Exit status: 0 = every scene matched (float fields to the bit), 1 = differences (listed), 2 = could not run.

mod gpu;
mod patch;
mod scene;

use scene::{Expect, Scene};
use std::path::{Path, PathBuf};

struct Args {
    shader: PathBuf,
    scenes: PathBuf,
    cases: Vec<String>,
    any_shader: bool,
}

fn parse_args() -> Result<Args, String> {
    let mut a = Args { shader: PathBuf::new(), scenes: PathBuf::from("scenes"), cases: Vec::new(), any_shader: false };
    let mut it = std::env::args().skip(1);
    while let Some(arg) = it.next() {
        match arg.as_str() {
            "--shader" => a.shader = it.next().ok_or("--shader needs a path")?.into(),
            "--scenes" => a.scenes = it.next().ok_or("--scenes needs a directory")?.into(),
            "--any-shader" => a.any_shader = true,
            "-h" | "--help" => return Err("usage: wgpu_check --shader <ray_tracer.wgsl> [--scenes DIR] [--any-shader] [case ...]".into()),
            c => a.cases.push(c.to_string()),
        }
    }
    if a.shader.as_os_str().is_empty() {
        return Err("--shader <path to the reference's clientdesktop/src/graphics/ray_tracer.wgsl> is required".into());
    }
    Ok(a)
}

/// textureStore to rgba8unorm of an f32 channel: clamp, scale, round half to even (what the fixtures' comparison uses too).
fn unorm8(x: f32) -> i32 {
    let c = if x.is_nan() { 0.0 } else { x.clamp(0.0, 1.0) };
    (c * 255.0).round_ties_even() as i32
}

fn ulps(a: f32, b: f32) -> u64 {
    if a.to_bits() == b.to_bits() || (a.is_nan() && b.is_nan()) {
        return 0;
    }
    let key = |x: f32| -> i64 { let b = x.to_bits() as i64; if b & 0x8000_0000 != 0 { 0x8000_0000 - b } else { b } };
    (key(a) - key(b)).unsigned_abs()
}

#[derive(Default)]
struct Diff {
    pixels: usize,
    colour_off_by_more_than_one: usize,
    hit: usize,
    voxel: usize,
    iters: usize,
    float_bits: usize,
    max_ulps: u64,
    first: Vec<String>,
}

impl Diff {
    fn clean(&self) -> bool {
        self.colour_off_by_more_than_one + self.hit + self.voxel + self.iters + self.float_bits == 0
    }
    fn integers_clean(&self) -> bool {
        self.hit + self.voxel + self.iters == 0
    }
    fn note(&mut self, what: String) {
        if self.first.len() < 8 {
            self.first.push(what);
        }
    }
}

fn rec_u32(r: &[u8], at: usize) -> u32 {
    u32::from_le_bytes([r[at], r[at + 1], r[at + 2], r[at + 3]])
}

fn compare(e: &Expect, rgba8: &[u8], records: &[u8]) -> Diff {
    let mut d = Diff { pixels: (e.width * e.height) as usize, ..Default::default() };
    let w = e.width as usize;
    for i in 0..d.pixels {
        let (x, y) = (i % w, i / w);
        if !e.rgb.is_empty() {
            for k in 0..3 {
                let (got, want) = (rgba8[4 * i + k] as i32, unorm8(e.rgb[i][k]));
                if (got - want).abs() > 1 {
                    d.colour_off_by_more_than_one += 1;
                    d.note(format!("({x},{y}) colour[{k}] {got} != {want}"));
                    break;
                }
            }
        }
        let r = &records[i * patch::DEBUG_RECORD_BYTES..(i + 1) * patch::DEBUG_RECORD_BYTES];
        let (hit, voxel, iters) = (rec_u32(r, 0), rec_u32(r, 4), rec_u32(r, 8));
        if !e.hit.is_empty() && hit != e.hit[i] as u32 {
            d.hit += 1;
            d.note(format!("({x},{y}) hit {hit} != {}", e.hit[i]));
        }
        if !e.voxel.is_empty() && voxel != e.voxel[i] {
            d.voxel += 1;
            d.note(format!("({x},{y}) voxel {voxel} != {}", e.voxel[i]));
        }
        if !e.iters.is_empty() && iters != e.iters[i] {
            d.iters += 1;
            d.note(format!("({x},{y}) iterations {iters} != {}", e.iters[i]));
        }
        let f = |at: usize| f32::from_bits(rec_u32(r, at));
        let mut floats: Vec<(&str, f32, f32)> = Vec::new();
        if !e.water_dist.is_empty() { floats.push(("water_dist", f(12), e.water_dist[i])); }
        if !e.norm.is_empty() { for k in 0..3 { floats.push(("norm", f(16 + 4 * k), e.norm[i][k])); } }
        if !e.pos.is_empty() { for k in 0..3 { floats.push(("pos", f(32 + 4 * k), e.pos[i][k])); } }
        let mut off = false;
        for (name, got, want) in floats {
            let u = ulps(got, want);
            if u != 0 {
                off = true;
                d.max_ulps = d.max_ulps.max(u);
                d.note(format!("({x},{y}) {name} {got:e} != {want:e} ({u} ulp)"));
            }
        }
        d.float_bits += off as usize;
    }
    d
}

fn report(case: &str, tag: &str, d: &Diff) {
    if d.clean() {
        println!("{case} [{tag}]: {} pixels, no difference", d.pixels);
        return;
    }
    println!("{case} [{tag}]: {} pixels — hit {} voxel {} iterations {} differ; float fields differ in {} pixels (largest {} ulp); colour off by more than one step in {}",
             d.pixels, d.hit, d.voxel, d.iters, d.float_bits, d.max_ulps, d.colour_off_by_more_than_one);
    for line in &d.first {
        println!("    {line}");
    }
}

fn expectations(dir: &Path, case: &str) -> Vec<(String, PathBuf)> {
    ["wgsl", "clamp", "zero"].iter().map(|t| (t.to_string(), dir.join(format!("{case}.{t}.vrtexpect")))).filter(|(_, p)| p.exists()).collect()
}

fn main() {
    let args = match parse_args() {
        Ok(a) => a,
        Err(e) => { eprintln!("{e}"); std::process::exit(2); }
    };
    let text = match std::fs::read(&args.shader) {
        Ok(t) => t,
        Err(e) => { eprintln!("{}: {e}", args.shader.display()); std::process::exit(2); }
    };
    let crc = scene::crc32(&text);
    let text = String::from_utf8(text).expect("the shader is not UTF-8");
    // the march function, the ray set-up it is fed from, and the locals the fixtures record (names of the reference's text)
    let instrumented = match patch::instrument(&text, "ray_world", "create_ray_from_screen", "voxel", "iter_count", "result") {
        Ok(s) => s,
        Err(e) => { eprintln!("cannot instrument the shader: {}", e.0); std::process::exit(2); }
    };
    let mut cases = args.cases.clone();
    if cases.is_empty() {
        let mut names: Vec<String> = std::fs::read_dir(&args.scenes).map(|rd| rd.filter_map(|e| e.ok()).filter_map(|e| {
            let p = e.path();
            (p.extension().map_or(false, |x| x == "vrtscene")).then(|| p.file_stem().unwrap().to_string_lossy().into_owned())
        }).collect()).unwrap_or_default();
        names.sort();
        cases = names;
    }
    if cases.is_empty() {
        eprintln!("no scene dumps in {} (python tests/golden/export_scenes.py {})", args.scenes.display(), args.scenes.display());
        std::process::exit(2);
    }
    let gpu = gpu::Gpu::new();
    println!("adapter: {} ({:?}, driver {})", gpu.adapter_info.name, gpu.adapter_info.backend, gpu.adapter_info.driver_info);
    println!("shader: {} (crc32 {crc:08x})", args.shader.display());
    let (mut differing, mut float_only) = (0, 0);
    for case in &cases {
        let sc = match Scene::read(&args.scenes.join(format!("{case}.vrtscene"))) {
            Ok(s) => s,
            Err(e) => { eprintln!("{e}"); std::process::exit(2); }
        };
        let exps = expectations(&args.scenes, case);
        if exps.is_empty() {
            eprintln!("{case}: no .vrtexpect file");
            std::process::exit(2);
        }
        let plain = gpu::run_pass(&gpu, &sc, &text, false);
        let debug = gpu::run_pass(&gpu, &sc, &instrumented, true);
        let mut results = Vec::new();
        for (tag, path) in &exps {
            let e = match Expect::read(path) {
                Ok(e) => e,
                Err(err) => { eprintln!("{err}"); std::process::exit(2); }
            };
            if e.shader_crc != crc && !args.any_shader {
                eprintln!("{case}: the fixture was made from a shader text with crc32 {:08x}, this file has {crc:08x} (--any-shader compares anyway)", e.shader_crc);
                std::process::exit(2);
            }
            if (e.width, e.height) != (sc.width, sc.height) {
                eprintln!("{case}: expectation and scene disagree on the frame's size");
                std::process::exit(2);
            }
            let d = compare(&e, &plain.rgba8, &debug.records);
            report(case, tag, &d);
            results.push((tag.clone(), d));
        }
        if results.len() > 1 {
            // an out-of-range scene: which reading of a read past an array's end this backend has
            let matching: Vec<&str> = results.iter().filter(|(_, d)| d.integers_clean()).map(|(t, _)| t.as_str()).collect();
            println!("{case}: this backend reads past an array's end as: {}", if matching.is_empty() { "neither recorded policy".to_string() } else { matching.join(" / ") });
            continue;   // (reported, not judged)
        }
        let d = &results[0].1;
        if !d.integers_clean() || d.colour_off_by_more_than_one != 0 {
            differing += 1;
        } else if d.float_bits != 0 {
            float_only += 1;
        }
    }
    println!("{} scenes: {} with differing integers or colours, {} with float bits differing only", cases.len(), differing, float_only);
    std::process::exit(if differing + float_only == 0 { 0 } else { 1 });
}
