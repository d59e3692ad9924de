#!/usr/bin/env python3
"""tools/isa_mix.py [--out profiles/rNN_isa_mix] — instruction mix of the default march kernel, from its ISA.

Compiles csrc/vrt_kernels.hip to gfx950 assembly with the Makefile's flags (hipcc -S, device only), takes the kernel the
bench times (primary_shadow_wave_kernel<0, false, false, 1>: grid march, primary + shadow in one launch), finds its two
march loops (primary ray, shadow ray) and classifies every instruction of every basic block of each loop:

    valu_simple   full-rate VALU (add / sub / mul / fma / and / or / xor / shifts / mov, incl. the VOP3 forms with modifiers)
    valu_half     half-rate VALU: compares, v_cndmask, conversions, min / max, the three-operand integer ops
                  (v_bfi, v_mad_*24, v_min3 / v_max3, v_lshl_or, v_add3, v_or3, v_and_or, v_add_lshl), v_div_fixup, v_*_co_*
    valu_pk       packed f32 (v_pk_*: two floats per instruction)
    valu_trans    transcendental (rcp / sqrt / rsq / exp / log / sin / cos)
    salu          scalar ALU (incl. the exec-mask bookkeeping of divergent control flow)
    branch        s_branch / s_cbranch_*
    wait_nop      s_waitcnt / s_nop
    vmem / lds / smem

(the classes and their issue costs are measured by tools/valu_rates.hip -> profiles/rNN_valu_issue_rates.txt).  The *fast
path* is what a wave executes on a march step in which every lane is in a plain air leaf of the cell grid: since the march
is two loops (DESIGN.md §3 m) that is exactly the inner loop — since round 6 an asm statement of vrt_march.h, (r), found by its labels.  Writes <out>.json (counts, code-object hash of the library
in the tree) and <out>.txt (the listing of both inner loops, block by block).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNEL = "_ZN3vrt26primary_shadow_wave_kernelILi0ELb0ELb0ELi1ELb0EEEvNS_11FrameParamsE"
HIPFLAGS = ("-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
            "-fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wno-unused-value").split()

TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
HALF = ("v_cmp", "v_cmpx", "v_cndmask", "v_cvt", "v_floor", "v_ceil", "v_trunc", "v_rndne", "v_fract", "v_min", "v_max", "v_med3", "v_bfi",
        "v_bfe", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24", "v_mul_i32_i24", "v_lshl_or", "v_lshl_add", "v_add3", "v_or3",
        "v_and_or", "v_add_lshl", "v_xad", "v_div_fixup", "v_div_scale", "v_div_fmas", "v_perm", "v_alignbit", "v_sad", "v_mul_lo",
        "v_mul_hi", "v_mad_u64", "v_ldexp", "v_frexp", "v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_bcnt", "v_ffb")


def classify(op: str) -> str:
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_") and ("_co_" in op or op.startswith(HALF)):
        return "valu_half"
    if op.startswith("v_"):
        return "valu_simple"
    if op in ("s_waitcnt", "s_nop") or op.startswith("s_waitcnt"):
        return "wait_nop"
    if op.startswith(("s_branch", "s_cbranch")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def kernel_text(asm: str) -> list:
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start + 1:end + 1]


def blocks_of(lines):
    """[(label, [instruction lines], loop header this block belongs to (innermost) or None, is an inner-loop header)] in layout order."""
    out, cur, label, inloop, hdr = [], [], "entry", None, False
    pending = None   # a label line whose loop comment continues on the next comment line
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m:
            out.append((label, cur, inloop, hdr))
            label, cur = (m.group(1) if l.startswith(".") else "bb." + m.group(1)), []
            c = m.group(2) or ""
            hdr = "=>This Inner Loop Header" in c
            mm = re.search(r"in Loop: Header=(BB\d+_\d+)", c)
            inloop = (".L" + mm.group(1)) if mm else (label if hdr else None)
            pending = label
            continue
        if pending and l.strip().startswith(";") and "This Inner Loop Header" in l:   # "Parent Loop ..." / "=>  This Inner Loop Header: Depth=2"
            hdr, inloop = True, pending
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith((".", ";")) or t.startswith(";;#"):
            continue
        pending = None
        cur.append(t)
    out.append((label, cur, inloop, hdr))
    return out


def count(insts):
    c = {}
    for t in insts:
        k = classify(t.split()[0])
        c[k] = c.get(k, 0) + 1
    return c


def add(a, b):
    return {k: a.get(k, 0) + b.get(k, 0) for k in set(a) | set(b)}


def write_out(res, listing, args):
    try:
        from voxelraytracing_amd import _ffi
        res["code_object_sha256"] = _ffi.code_object_sha256()
    except Exception as e:   # the library is not built: the mix still stands for the source
        res["code_object_sha256"] = None
        res["note"] = f"libvrt.so not hashed: {e}"
    json.dump(res, open(args.out + ".json", "w"), indent=1, sort_keys=True)
    with open(args.out + ".txt", "w") as f:
        f.write(f"# {res['kernel']}: ISA of the two march loops, block by block (tools/isa_mix.py; hipcc -S, gfx950)\n")
        for name in ("primary", "shadow"):
            f.write(f"# {name} fast path per step: {json.dumps(res['loops'][name]['fast_path'], sort_keys=True)}\n")
        f.write("\n".join(listing) + "\n")
    for name in ("primary", "shadow"):
        print(name, "fast path per step:", json.dumps(res["loops"][name]["fast_path"], sort_keys=True))


def hand_written(text, args):
    """Round 6: the inner loop is an asm statement of vrt_march.h (r) — labels .Lvrt_step_N (one trip per step: the lookup, the test),
    .Lvrt_planes_N (the exit planes), .Lvrt_move_N (which distances are the step), .Lvrt_advance_N (the move, the loop's branch),
    .Lvrt_zero_N (some lane's smallest distance is zero or NaN: rare, out of line), .Lvrt_split_N / .Lvrt_leave_N ((s): lanes in split cells, the voxel of the brick), .Lvrt_out_N.
    What a fast step executes: `step`, `planes`, `move` and `advance` up to and including the loop's own branch."""
    total = {}
    for b in blocks_of(text):
        total = add(total, count(b[1]))
    res = {"kernel": "primary_shadow_wave_kernel<0, false, false, 1>", "loops": {}, "whole_kernel_static": total, "inner_loop": "hand-written (vrt_march.h (r))"}
    starts = [i for i, l in enumerate(text) if re.match(r"^\s*\.Lvrt_step_\d+:", l)]
    assert len(starts) == 2, f"expected the inner march loop of the primary and of the shadow ray, found {len(starts)}"
    listing = []
    for name, i0 in zip(("primary", "shadow"), starts):
        n = re.match(r"^\s*\.Lvrt_step_(\d+):", text[i0]).group(1)
        i1 = next(i for i in range(i0, len(text)) if text[i].strip().startswith(f".Lvrt_out_{n}:"))
        blocks, lab = {}, None
        for l in text[i0:i1]:
            m = re.match(r"^\s*\.Lvrt_(\w+?)_\d+:", l)
            if m:
                lab = m.group(1)
                blocks[lab] = []
                continue
            t = l.split(";")[0].strip()
            if t and not t.startswith((".", ";")):
                blocks[lab].append(t)
        last = "advance" if "advance" in blocks else "move"
        k = next(i for i, t in enumerate(blocks[last]) if t.startswith("s_cbranch_scc")) + 1
        blocks["exit"], blocks[last] = blocks[last][k:], blocks[last][:k]   # behind the loop's branch: after kMaxSteps lookups only
        per_block = {b: count(v) for b, v in blocks.items()}
        fast = [b for b in ("step", "planes", "move", "advance") if b in blocks]
        fp = {}
        for b in fast:
            fp = add(fp, per_block[b])
        others = [b for b in blocks if b not in fast]
        res["loops"][name] = {"header": f".Lvrt_step_{n}", "fast_path_blocks": fast, "fast_path": fp, "per_block": per_block,
                              "rarely_executed_blocks": [b for b in others if b in ("exit", "zero")],
                              "split_cell_step_blocks": [b for b in others if b == "split"],
                              "stop_and_leave_blocks": [b for b in others if b in ("decide", "leave")]}
        listing.append(f"==== {name} ray: inner march loop (.Lvrt_step_{n}) = the fast path, one trip per step ====")
        for b in fast + others:
            listing.append(f"{b}:   {per_block[b]}" + ("" if b in fast else "   (not in the fast path: a step through air voxels of split cells adds `split`)" if b == "split" else "   (not in the fast path)"))
            listing += ["    " + t for t in blocks[b]]
    write_out(res, listing, args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_isa_mix"))
    ap.add_argument("--asm", default=None, help="an existing .s instead of compiling")
    args = ap.parse_args()
    if args.asm:
        asm = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            s = os.path.join(d, "vrt_kernels.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", *HIPFLAGS, "-S", "--cuda-device-only", "-o", s,
                                   os.path.join(ROOT, "voxelraytracing_amd", "csrc", "vrt_kernels.hip")], stderr=subprocess.DEVNULL)
            asm = open(s).read()
    text = kernel_text(asm)
    if any(re.match(r"^\s*\.Lvrt_step_\d+:", l) for l in text):
        return hand_written(text, args)
    bl = blocks_of(text)
    headers = [b[0] for b in bl if b[3]]
    assert len(headers) == 2, f"expected the inner march loop of the primary and of the shadow ray, found {headers}"
    res = {"kernel": "primary_shadow_wave_kernel<0, false, false, 1>", "loops": {}}
    listing = []
    total = {}
    for b in bl:
        total = add(total, count(b[1]))
    res["whole_kernel_static"] = total
    for name, h in zip(("primary", "shadow"), headers):
        # the inner loop (DESIGN.md §3 m) IS the fast path: the steps in which no lane of the wave has anything to decide
        loop = [b for b in bl if b[2] == h]
        per_block = {b[0]: count(b[1]) for b in loop}
        # the bit-pattern minimum a wave falls back to when some lane's smallest distance is zero or NaN (vrt_march.h (p)):
        # the block behind `v_cmp_nlt_f32 vcc, 0, <min>; s_cbranch_vccz <over it>` is skipped on (practically) every
        # step — listed, not counted into the fast path
        rare = [loop[i + 1][0] for i in range(len(loop) - 1)
                if len(loop[i][1]) >= 2 and loop[i][1][-1].startswith("s_cbranch_vccz") and loop[i][1][-2].startswith("v_cmp_nlt_f32")
                and " 0, " in loop[i][1][-2]]
        fp = {}
        for lab in per_block:
            if lab not in rare:
                fp = add(fp, per_block[lab])
        res["loops"][name] = {"header": h, "fast_path_blocks": [b[0] for b in loop if b[0] not in rare], "fast_path": fp, "per_block": per_block,
                              "rarely_executed_blocks": rare}
        listing.append(f"==== {name} ray: inner march loop (header {h}) = the fast path, one trip per step ====")
        for b in loop:
            listing.append(f"{b[0]}:   {per_block[b[0]]}" + ("   (not in the fast path: only when some lane's smallest distance is zero or NaN)" if b[0] in rare else ""))
            listing += ["    " + t for t in b[1]]
    write_out(res, listing, args)


if __name__ == "__main__":
    main()
