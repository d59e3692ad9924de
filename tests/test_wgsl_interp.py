"""Known answers for tests/wgsl_interp.py itself.

The reference-shader fixtures (tests/golden/wgsl_*.npz) are only as good as the interpreter that executed the shader, so the
interpreter is held to answers that follow from the WGSL specification alone — small programs, each result worked out by
hand from the rule it exercises (the section of the spec is named), none of them taken from the oracle or the kernels.
The places where WGSL leaves the answer to the implementation are listed in the interpreter's header; the tests of those
say which choice they pin."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wgsl_interp as W   # noqa: E402

F32, I32, U32 = np.float32, np.int32, np.uint32


def run(src, fn="f", *args):
    return W.Module(src).call(fn, *args)


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def vec(*xs):
    return W.Vec("f32", [F32(x) for x in xs])


# ---- integers: two's complement, wrap-around, shifts, division (WGSL 8.7 arithmetic, 8.9 bit expressions) ----

def test_u32_multiplication_and_addition_wrap_modulo_two_to_the_32():
    # the PCG step of path_tracer.wgsl:57 on state 1: 747796405 + 2891336453 = 3639132858 (< 2^32);
    # on that state again: (3639132858 * 747796405 + 2891336453) mod 2^32, worked with Python integers
    src = "fn f(s: u32) -> u32 { return s * 747796405u + 2891336453u; }"
    assert int(run(src, "f", U32(1))) == 3639132858
    assert int(run(src, "f", U32(3639132858))) == (3639132858 * 747796405 + 2891336453) % 2**32


def test_i32_wraps_and_negation_of_the_minimum_is_itself():
    src = "fn f(a: i32) -> i32 { return a + 1i; }  fn g(a: i32) -> i32 { return -a; }"
    assert int(run(src, "f", I32(2147483647))) == -2147483648
    assert int(run(src, "g", I32(-2147483648))) == -2147483648


def test_shifts_are_logical_for_u32_arithmetic_for_i32_and_use_the_low_five_bits_of_the_count():
    src = """fn a(x: u32, n: u32) -> u32 { return x >> n; }
             fn b(x: i32, n: u32) -> i32 { return x >> n; }
             fn c(x: u32, n: u32) -> u32 { return x << n; }"""
    assert int(run(src, "a", U32(0x80000000), U32(31))) == 1
    assert int(run(src, "b", I32(-8), U32(1))) == -4               # sign-extending
    assert int(run(src, "c", U32(0xFFFFFFFF), U32(4))) == 0xFFFFFFF0
    assert int(run(src, "a", U32(0x80000000), U32(33))) == 0x40000000   # count mod 32 (the dynamic-shift rule)


def test_integer_division_truncates_towards_zero_and_remainder_takes_the_dividends_sign():
    src = "fn d(a: i32, b: i32) -> i32 { return a / b; }  fn r(a: i32, b: i32) -> i32 { return a % b; }"
    assert int(run(src, "d", I32(-7), I32(2))) == -3
    assert int(run(src, "r", I32(-7), I32(2))) == -1
    assert int(run(src, "r", I32(7), I32(-2))) == 1


def test_bitwise_operators_and_precedence():
    # & binds tighter than | is NOT a WGSL rule (mixing them needs parentheses); shifts need them too.  With parentheses:
    src = "fn f(a: u32, b: u32) -> u32 { return ((a >> 4u) & 0xFu) | ((b & 1u) << 8u) | (~a & 0x1000u); }"
    assert int(run(src, "f", U32(0xAB), U32(3))) == (0xA | 0x100 | 0x1000)


# ---- floating point: binary32, one rounding per operation, no contraction (WGSL 14.6 floating point evaluation) ----

def test_every_f32_operation_rounds_to_binary32_on_its_own():
    # 16777216 + 1 is not a binary32 number: the sum rounds to even (16777216), so (a + 1) - a is 0, not 1
    src = "fn f(a: f32) -> f32 { return (a + 1.0) - a; }"
    assert float(run(src, "f", F32(16777216.0))) == 0.0
    # a * b + c with the product rounded first: 1.0000001 * 1.0000001 = 1.0000002384... rounds to 1 + 2 ulp; a fused
    # multiply-add of (a * a - (1 + 2 ulp)) would return the product's rounding error, the unfused form returns 0
    a = np.nextafter(F32(1.0), F32(2.0))
    src = "fn f(a: f32, c: f32) -> f32 { return a * a - c; }"
    assert float(run(src, "f", a, F32(a * a))) == 0.0


def test_division_by_zero_and_zero_over_zero_are_ieee():
    src = "fn f(a: f32, b: f32) -> f32 { return a / b; }"
    assert float(run(src, "f", F32(1.0), F32(0.0))) == float("inf")
    assert float(run(src, "f", F32(-1.0), F32(0.0))) == float("-inf")
    assert np.isnan(run(src, "f", F32(0.0), F32(0.0)))
    assert bits(run(src, "f", F32(0.0), F32(-3.0))) == 0x80000000   # -0


def test_comparisons_with_nan_are_false_and_not_equal_is_true():
    src = """fn lt(a: f32, b: f32) -> bool { return a < b; }  fn ge(a: f32, b: f32) -> bool { return a >= b; }
             fn eq(a: f32, b: f32) -> bool { return a == b; }  fn ne(a: f32, b: f32) -> bool { return a != b; }"""
    nan = F32("nan")
    assert not run(src, "lt", nan, F32(1)) and not run(src, "ge", nan, F32(1)) and not run(src, "eq", nan, nan)
    assert run(src, "ne", nan, nan)


def test_abstract_float_literals_become_f32_when_they_meet_one():
    # 0.001 as binary32 is 0x3A83126F (the constant the march's nudge uses, ray_tracer.wgsl:188-190)
    src = "fn f(x: f32) -> f32 { return x + 0.001; }"
    assert bits(run(src, "f", F32(0.0))) == 0x3A83126F
    # 4294967295.0 is not a binary32 number: it becomes 2^32, so f32(0xFFFFFFFF) / 4294967295.0 is exactly 1 (path_tracer.wgsl:60)
    src = "fn f(r: u32) -> f32 { return f32(r) / 4294967295.0; }"
    assert float(run(src, "f", U32(0xFFFFFFFF))) == 1.0


# ---- conversions (WGSL 8.5 / 14.6.2: f32 -> integer truncates and clamps; u32 -> f32 rounds to nearest even) ----

def test_float_to_integer_truncates_and_saturates():
    src = "fn i(x: f32) -> i32 { return i32(x); }  fn u(x: f32) -> u32 { return u32(x); }"
    assert int(run(src, "i", F32(-3.9))) == -3 and int(run(src, "i", F32(3.9))) == 3
    assert int(run(src, "i", F32(3e9))) == 2147483647 and int(run(src, "i", F32(-3e9))) == -2147483648
    assert int(run(src, "u", F32(-1.5))) == 0 and int(run(src, "u", F32(5e9))) == 0xFFFFFFFF
    assert int(run(src, "i", F32("nan"))) == 0          # (implementation's choice, named in the interpreter's header: what the GPU's v_cvt does)


def test_integer_to_float_rounds_to_nearest_even():
    src = "fn f(x: u32) -> f32 { return f32(x); }  fn g(x: i32) -> f32 { return f32(x); }"
    assert float(run(src, "f", U32(16777217))) == 16777216.0      # halfway: to even
    assert float(run(src, "f", U32(16777219))) == 16777220.0
    assert float(run(src, "g", I32(-16777217))) == -16777216.0


def test_bitcast_like_reinterpretation_is_not_a_conversion():
    src = "fn f(x: u32) -> i32 { return i32(x); }  fn g(x: i32) -> u32 { return u32(x); }"
    assert int(run(src, "f", U32(0xFFFFFFFF))) == -1     # u32 -> i32: reinterpretation of the bits (WGSL: value-preserving modulo 2^32)
    assert int(run(src, "g", I32(-2))) == 0xFFFFFFFE


# ---- vectors and matrices (WGSL 8.7: componentwise; matrix * vector is the linear combination of the columns) ----

def test_vector_arithmetic_is_componentwise_and_scalars_broadcast():
    src = """fn f(a: vec3<f32>, b: vec3<f32>, s: f32) -> vec3<f32> { return a * b + s; }
             fn g(a: vec3<f32>) -> vec3<f32> { return -a.zyx; }
             fn h(a: vec3<f32>) -> vec2<f32> { return a.xz * 2.0; }"""
    assert [float(x) for x in run(src, "f", vec(1, 2, 3), vec(4, 5, 6), F32(0.5)).v] == [4.5, 10.5, 18.5]
    assert [float(x) for x in run(src, "g", vec(1, 2, 3)).v] == [-3.0, -2.0, -1.0]
    assert [float(x) for x in run(src, "h", vec(1, 2, 3)).v] == [2.0, 6.0]


def test_matrix_times_vector_combines_the_columns_and_vector_times_matrix_the_rows():
    # mat4x4(c0, c1, c2, c3): the arguments are COLUMNS.  M * v = c0*v.x + c1*v.y + c2*v.z + c3*v.w
    src = """fn m() -> mat4x4<f32> { return mat4x4<f32>(vec4<f32>(1.0, 2.0, 3.0, 4.0), vec4<f32>(5.0, 6.0, 7.0, 8.0),
                                                     vec4<f32>(9.0, 10.0, 11.0, 12.0), vec4<f32>(13.0, 14.0, 15.0, 16.0)); }
             fn f(v: vec4<f32>) -> vec4<f32> { return m() * v; }
             fn g(v: vec4<f32>) -> vec4<f32> { return v * m(); }"""
    assert [float(x) for x in run(src, "f", vec(1, 0, 0, 1)).v] == [14.0, 16.0, 18.0, 20.0]        # c0 + c3
    assert [float(x) for x in run(src, "g", vec(1, 0, 0, 1)).v] == [5.0, 13.0, 21.0, 29.0]         # dot(v, column i)


def test_vector_constructors_take_mixed_pieces_and_splat():
    src = """fn f(a: vec2<f32>) -> vec4<f32> { return vec4<f32>(a, 0.0, 1.0); }
             fn g() -> vec3<f32> { return vec3<f32>(2.0); }
             fn h(a: vec3<f32>) -> vec4<f32> { return vec4(a.xy, a.z, 7.0); }"""
    assert [float(x) for x in run(src, "f", vec(3, 4)).v] == [3.0, 4.0, 0.0, 1.0]
    assert [float(x) for x in run(src, "g").v] == [2.0, 2.0, 2.0]
    assert [float(x) for x in run(src, "h", vec(1, 2, 3)).v] == [1.0, 2.0, 3.0, 7.0]


def test_comparison_of_vectors_is_componentwise_and_select_picks_per_component():
    # select(f, t, cond): t where cond (WGSL 17.3: note the order)
    src = """fn f(a: vec3<f32>, b: vec3<f32>) -> vec3<f32> { return select(a, b, a < b); }
             fn g(c: bool) -> f32 { return select(1.0, 2.0, c); }"""
    assert [float(x) for x in run(src, "f", vec(1, 5, 3), vec(2, 4, 3)).v] == [2.0, 5.0, 3.0]
    assert float(run(src, "g", True)) == 2.0 and float(run(src, "g", False)) == 1.0


# ---- built-in functions with answers the spec fixes (WGSL 17.5) ----

def test_floor_fract_sign_abs_and_the_sign_of_zero():
    src = """fn a(x: f32) -> f32 { return floor(x); }  fn b(x: f32) -> f32 { return fract(x); }
             fn c(x: f32) -> f32 { return sign(x); }   fn d(x: f32) -> f32 { return abs(x); }"""
    assert float(run(src, "a", F32(-0.25))) == -1.0 and float(run(src, "a", F32(2.0))) == 2.0
    assert float(run(src, "b", F32(-0.25))) == 0.75          # x - floor(x)
    assert [float(run(src, "c", F32(v))) for v in (-3.0, 0.0, 5.0)] == [-1.0, 0.0, 1.0]
    assert bits(run(src, "d", F32(-0.0))) == 0


def test_min_and_max_ignore_a_nan_operand():
    # (named in the interpreter's header: IEEE minNum / maxNum, what the spec words min and max as and what gfx950 does)
    src = "fn lo(a: f32, b: f32) -> f32 { return min(a, b); }  fn hi(a: f32, b: f32) -> f32 { return max(a, b); }"
    nan = F32("nan")
    assert float(run(src, "lo", nan, F32(2))) == 2.0 and float(run(src, "lo", F32(2), nan)) == 2.0
    assert float(run(src, "hi", nan, F32(-2))) == -2.0
    assert float(run(src, "lo", F32(1), F32(2))) == 1.0 and float(run(src, "hi", F32(1), F32(2))) == 2.0


def test_dot_length_normalize_and_distance_in_single_operations():
    src = """fn d(a: vec3<f32>, b: vec3<f32>) -> f32 { return dot(a, b); }
             fn l(a: vec3<f32>) -> f32 { return length(a); }
             fn n(a: vec3<f32>) -> vec3<f32> { return normalize(a); }
             fn s(a: vec3<f32>, b: vec3<f32>) -> f32 { return distance(a, b); }"""
    assert float(run(src, "d", vec(1, 2, 3), vec(4, -5, 6))) == 12.0
    assert float(run(src, "l", vec(3, 4, 12))) == 13.0
    assert [float(x) for x in run(src, "n", vec(0, 3, 4)).v] == [0.0, float(F32(3) / F32(5)), float(F32(4) / F32(5))]
    assert float(run(src, "s", vec(1, 1, 1), vec(4, 5, 1))) == 5.0
    assert all(np.isnan(x) for x in run(src, "n", vec(0, 0, 0)).v)    # 0 / 0: what the march's axis-parallel rays rely on (:209-213)


def test_mix_clamp_and_smoothstep_by_their_defining_formulas():
    src = """fn m(a: f32, b: f32, t: f32) -> f32 { return mix(a, b, t); }
             fn c(x: f32) -> f32 { return clamp(x, 0.0, 1.0); }
             fn s(x: f32) -> f32 { return smoothstep(0.0, 2.0, x); }"""
    assert float(run(src, "m", F32(2), F32(6), F32(0.25))) == 3.0
    assert [float(run(src, "c", F32(v))) for v in (-1.0, 0.5, 7.0)] == [0.0, 0.5, 1.0]
    assert float(run(src, "s", F32(1.0))) == 0.5 and float(run(src, "s", F32(-1.0))) == 0.0 and float(run(src, "s", F32(9.0))) == 1.0


def test_reflect_is_e1_minus_two_dot_e2_e1_e2():
    src = "fn f(i: vec3<f32>, n: vec3<f32>) -> vec3<f32> { return reflect(i, n); }"
    try:
        r = W.Module(src).call("f", vec(1, -1, 0), vec(0, 1, 0))
    except (NameError, KeyError, TypeError):
        pytest.skip("reflect is not among the built-ins the reference's live shaders call")
    assert [float(x) for x in r.v] == [1.0, 1.0, 0.0]


# ---- statements: scoping, loops, pointers, structs, arrays (WGSL 9, 7.4) ----

def test_loops_break_continue_and_while():
    src = """fn f(n: u32) -> u32 {
                 var sum = 0u;
                 var i = 0u;
                 loop {
                     if (i >= n) { break; }
                     i += 1u;
                     if ((i & 1u) == 0u) { continue; }
                     sum += i;
                 }
                 return sum;
             }
             fn g(n: i32) -> i32 { var k = n; var c = 0i; while (k > 0i) { k = k / 2i; c += 1i; } return c; }"""
    assert int(run(src, "f", U32(10))) == 1 + 3 + 5 + 7 + 9
    assert int(run(src, "g", I32(1000))) == 10


def test_a_for_loop_if_the_interpreter_has_one():
    src = "fn f(n: u32) -> u32 { var s = 0u; for (var i = 0u; i < n; i++) { s += i * i; } return s; }"
    try:
        r = run(src, "f", U32(5))
    except SyntaxError:
        pytest.skip("`for` is not in the subset the reference's shaders use")
    assert int(r) == 0 + 1 + 4 + 9 + 16


def test_inner_scopes_shadow_and_end():
    src = """fn f() -> i32 {
                 var a = 1i;
                 { var a = 5i; a += 1i; }
                 if (true) { let b = a + 10i; a = b; }
                 return a;
             }"""
    assert int(run(src, "f")) == 11


def test_pointers_to_function_variables_are_read_and_written_through():
    src = """fn bump(p: ptr<function, u32>) -> u32 { *p = *p + 3u; return *p * 2u; }
             fn f() -> u32 { var s = 4u; let r = bump(&s); return r + s; }"""
    assert int(run(src, "f")) == 14 + 7


def test_function_arguments_are_passed_by_value():
    src = """fn change(v: vec3<f32>) -> f32 { var w = v; w.x = 100.0; return w.x; }
             fn f() -> f32 { var v = vec3<f32>(1.0, 2.0, 3.0); let r = change(v); return r + v.x; }"""
    assert float(run(src, "f")) == 101.0


def test_structs_members_and_assignment_through_member_and_swizzle_paths():
    src = """struct Hit { pos: vec3<f32>, n: u32, }
             fn f() -> f32 {
                 var h: Hit;
                 h.pos = vec3<f32>(1.0, 2.0, 3.0);
                 h.pos.y = 7.0;
                 h.n = 2u;
                 var g = h;          // a copy
                 g.pos.x = 50.0;
                 return h.pos.x + h.pos.y + f32(h.n) + g.pos.x;
             }
             fn z() -> u32 { var h: Hit; return h.n + u32(h.pos.z); }"""
    assert float(run(src, "f")) == 1.0 + 7.0 + 2.0 + 50.0
    assert int(run(src, "z")) == 0      # a variable without an initialiser holds the zero value (WGSL 7.3)


def test_arrays_index_and_construct():
    src = """fn f(i: u32) -> f32 { var a = array<f32, 4>(1.5, 2.5, 3.5, 4.5); a[1] = 10.0; return a[i] + a[1]; }"""
    assert float(run(src, "f", U32(3))) == 14.5


def test_short_circuit_operators_do_not_evaluate_the_right_side():
    src = """fn side(p: ptr<function, u32>) -> bool { *p = *p + 1u; return true; }
             fn f(a: bool) -> u32 { var n = 0u; let r = a && side(&n); let s = a || side(&n); return n; }"""
    assert int(run(src, "f", False)) == 1     # && skipped it, || ran it
    assert int(run(src, "f", True)) == 1      # && ran it, || skipped it


def test_module_scope_bindings_and_storage_arrays():
    src = """struct Params { scale: f32, count: u32, }
             @group(0) @binding(0) var<uniform> params: Params;
             @group(0) @binding(1) var<storage, read> data: array<u32>;
             fn f() -> f32 { var s = 0u; var i = 0u; loop { if (i >= params.count) { break; } s += data[i]; i += 1u; } return f32(s) * params.scale; }"""
    m = W.Module(src)
    m.bind("params", W.Struct("Params", {"scale": F32(0.5), "count": U32(3)}))
    m.bind("data", np.array([10, 20, 30, 40], dtype=np.uint32))
    assert float(m.call("f")) == 30.0


# ---- the reference's own helper functions, against values worked out from their text ----

@pytest.mark.skipif(not os.path.exists("/root/reference/clientdesktop/src/graphics/path_tracer.wgsl"), reason="needs the reference's shader text")
def test_rng_next_of_the_reference_by_hand():
    """path_tracer.wgsl:56-61 on state 0, worked with Python integers: the interpreter executes the text, the test the arithmetic."""
    m = W.Module(open("/root/reference/clientdesktop/src/graphics/path_tracer.wgsl").read())
    s = (0 * 747796405 + 2891336453) % 2**32
    r = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) % 2**32
    r = (r >> 22) ^ r
    scope = {"rng": U32(0)}
    got = m.call("rng_next", W.Ref(scope, "rng"))
    assert int(scope["rng"]) == s
    assert float(got) == float(F32(F32(r) / F32(4294967296.0)))
