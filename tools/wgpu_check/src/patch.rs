//! Instrumentation of the reference's shader text, done at run time on the text read from the maintainer's checkout (nothing of
//! it is stored in this repository): the march function keeps the two locals the fixtures record beside its result — the voxel of
//! the last lookup and the iteration count — in private variables, and a second entry point writes them and the HitResult's
//! fields into a storage buffer at binding 8.  The reference's own entry point and every function it calls are left as they
//! are; the instrumented text is only used for the second pass.

/// One record per pixel of the debug pass (48 bytes; WGSL layout: four scalars, then two vec4).
pub const DEBUG_RECORD_BYTES: usize = 48;
pub const DEBUG_BINDING: u32 = 8;
pub const DEBUG_ENTRY: &str = "vrt_dbg_update";

#[derive(Debug)]
pub struct PatchError(pub String);

/// Index of the brace that closes the block opened at `open` (which must be a '{'); `//` comments are skipped.
fn matching_brace(src: &[u8], open: usize) -> Option<usize> {
    let (mut depth, mut i) = (0usize, open);
    while i < src.len() {
        match src[i] {
            b'/' if i + 1 < src.len() && src[i + 1] == b'/' => {
                while i < src.len() && src[i] != b'\n' {
                    i += 1;
                }
                continue;
            }
            b'{' => depth += 1,
            b'}' => {
                depth -= 1;
                if depth == 0 {
                    return Some(i);
                }
            }
            _ => {}
        }
        i += 1;
    }
    None
}

/// The shader with the debug entry point.  `march_fn` is the name of the function whose locals are wanted (the reference's march
/// loop), `voxel_local` / `iters_local` the locals, `result_local` what it returns: every `return <result_local>;` that follows the
/// declaration of both locals inside that function first copies them out.
pub fn instrument(src: &str, march_fn: &str, ray_fn: &str, voxel_local: &str, iters_local: &str, result_local: &str) -> Result<String, PatchError> {
    let head = format!("fn {}(", march_fn);
    let at = src.find(&head).ok_or_else(|| PatchError(format!("no function `{}` in the shader", march_fn)))?;
    let open = at + src[at..].find('{').ok_or_else(|| PatchError("no body".into()))?;
    let close = matching_brace(src.as_bytes(), open).ok_or_else(|| PatchError("unbalanced braces".into()))?;
    let body = &src[open..=close];
    let decl = |name: &str| -> Result<usize, PatchError> {
        let pat = format!("var {}", name);
        body.find(&pat).ok_or_else(|| PatchError(format!("`{}` is not declared in `{}`", pat, march_fn)))
    };
    let after = decl(voxel_local)?.max(decl(iters_local)?);
    let ret = format!("return {};", result_local);
    let copy_out = format!("vrt_dbg_voxel_ = {}; vrt_dbg_iters_ = {}; {}", voxel_local, iters_local, ret);
    let (before, tail) = body.split_at(after);
    let n_returns = tail.matches(&ret).count();
    if n_returns == 0 {
        return Err(PatchError(format!("no `{}` behind the locals' declarations", ret)));
    }
    let mut out = String::with_capacity(src.len() + 2048);
    out.push_str(&src[..open]);
    out.push_str(before);
    out.push_str(&tail.replace(&ret, &copy_out));
    out.push_str(&src[close + 1..]);
    // the debug pass's own declarations (this harness's text, not the reference's)
    let compute = ["@", "compute"].concat();
    out.push_str(&format!(
        "\n\nstruct VrtDbgRec_ {{ hit: u32, voxel: u32, iters: u32, water_dist: f32, norm: vec4<f32>, pos: vec4<f32> }}\n\
         @group(0) @binding({binding}) var<storage, read_write> vrt_dbg_: array<VrtDbgRec_>;\n\
         var<private> vrt_dbg_voxel_: u32 = 0u;\n\
         var<private> vrt_dbg_iters_: u32 = 0u;\n\
         {compute} @workgroup_size(8, 8, 1)\n\
         fn {entry}(@builtin(global_invocation_id) id: vec3<u32>) {{\n\
         \x20   let ray = {ray_fn}(vec2<i32>(id.xy));\n\
         \x20   let r = {march_fn}(ray);\n\
         \x20   let w = textureDimensions(output_texture_).x;\n\
         \x20   vrt_dbg_[id.y * w + id.x] = VrtDbgRec_(u32(r.hit), vrt_dbg_voxel_, vrt_dbg_iters_, r.water_dist, vec4<f32>(r.norm, 0.0), vec4<f32>(r.pos, 0.0));\n\
         }}\n",
        binding = DEBUG_BINDING, compute = compute, entry = DEBUG_ENTRY, ray_fn = ray_fn, march_fn = march_fn));
    Ok(out)
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn copies_the_locals_out_before_the_returns_that_follow_them() {
        let src = "fn walk(r: Ray) -> Res {\n var result: Res;\n if early { return result; }\n var voxel: u32;\n var iter_count: u32 = 0u;\n \
                   loop { if done { return result; } } // } not a brace\n return result;\n}\nfn other() {}\n";
        let out = instrument(src, "walk", "make_ray", "voxel", "iter_count", "result").unwrap();
        assert_eq!(out.matches("vrt_dbg_voxel_ = voxel;").count(), 2);
        assert!(out.contains("if early { return result; }"));
        assert!(out.contains("fn other() {}"));
        assert!(out.contains(DEBUG_ENTRY));
    }
}
