"""A small WGSL interpreter — TEST INFRASTRUCTURE ONLY.

It executes WGSL *source text* on the CPU: a tokenizer, a recursive-descent parser and a tree-walking evaluator for the
subset of the language the reference's shaders use (structs, module-scope `var<uniform|storage>` bindings, functions,
let / var, assignment and compound assignment through member and swizzle paths, if / else, loop / while / break / return,
scalar, vector and 4x4-matrix arithmetic in binary32, conversions, the built-in functions they call, textureStore).  It
knows nothing about ray marching or octrees: tests/golden/make_wgsl_fixtures.py points it at the reference's own
`clientdesktop/src/graphics/ray_tracer.wgsl` (read from /root/reference where that exists) and records what the shader
computes per pixel; those records are the fixtures that pin oracle/vrt_oracle.c to the reference's source rather than to a
reading of it.

Arithmetic: every f32 operation is one numpy.float32 operation (round-to-nearest-even binary32, no contraction), i32 / u32
wrap, shifts take the low five bits of the count.  Where WGSL leaves the result to the implementation this interpreter
takes, and says here, the same choice the oracle documents:
  min / max of a number and a NaN  -> the number (IEEE minNum; the WGSL spec's own wording for min and max)
  i32(f32), u32(f32)               -> truncation, saturating; NaN -> 0
  normalize(v)                     -> v / sqrt(v.x*v.x + v.y*v.y + v.z*v.z), sums left to right
  dot(a, b)                        -> a.x*b.x + a.y*b.y (+ ...), left to right, no fused multiply-add
  mix(a, b, t)                     -> a*(1-t) + b*t;  smoothstep: t = clamp((x-lo)/(hi-lo), 0, 1), t*t*(3-2t);  clamp = min(max(x, lo), hi)
  pow(x, y)                        -> the binary32 rounding of the double-precision power
  v * M (vector times matrix)      -> component i = dot(v, column i of M)            (WGSL: transpose(M) * v)
  M * v                            -> component i = dot(row i of M, v), left to right
  an array index past the end      -> an error by default (the thirteen scene fixtures never do it); Module(src, oob="clamp" | "zero")
                                      runs the two policies an implementation may take — the index clamped to the last element, or a
                                      zero value — for tests/golden/wgsl_oob.npz, which records the shader under both
"""
from __future__ import annotations

import math
import re

import numpy as np

F32, I32, U32 = np.float32, np.int32, np.uint32
np.seterr(all="ignore")


# ------------------------------------------------------------------------------------------------------------------
# tokens
# ------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/) |
    (?P<num>0[xX][0-9a-fA-F]+[iu]? | (?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fiu]?) |
    (?P<id>[A-Za-z_][A-Za-z0-9_]*) |
    (?P<op><<=|>>=|->|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|[-+*/%&|^!~<>=(){}\[\],;:.@])
""", re.X | re.S)


def tokenize(src: str):
    out, pos = [], 0
    while pos < len(src):
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError(f"cannot tokenize at {src[pos:pos + 30]!r}")
        pos = m.end()
        if m.lastgroup != "ws":
            out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


# ------------------------------------------------------------------------------------------------------------------
# types and values
# ------------------------------------------------------------------------------------------------------------------
class Vec:
    """A vector of 2..4 scalars of one type ('f32', 'i32', 'u32', 'bool', or 'abs_i' / 'abs_f' for untyped literals)."""
    __slots__ = ("t", "v")

    def __init__(self, t, v):
        self.t, self.v = t, list(v)

    def __repr__(self):
        return f"vec{len(self.v)}<{self.t}>{tuple(self.v)}"


class Mat4:
    __slots__ = ("cols",)

    def __init__(self, cols):
        self.cols = cols   # four Vec('f32', 4): column-major, as WGSL stores it


class Struct:
    __slots__ = ("name", "f")

    def __init__(self, name, fields):
        self.name, self.f = name, fields


class Ref:
    """ptr<function, T>: the scope that holds a variable, and its name."""
    __slots__ = ("scope", "name")

    def __init__(self, scope, name):
        self.scope, self.name = scope, name


def copy_value(x):
    if isinstance(x, Vec):
        return Vec(x.t, x.v)
    if isinstance(x, Struct):
        return Struct(x.name, {k: copy_value(v) for k, v in x.f.items()})
    if isinstance(x, Mat4):
        return Mat4([copy_value(c) for c in x.cols])
    return x


def scalar_type(x):
    if isinstance(x, (bool, np.bool_)):
        return "bool"
    if isinstance(x, np.float32):
        return "f32"
    if isinstance(x, np.int32):
        return "i32"
    if isinstance(x, np.uint32):
        return "u32"
    if isinstance(x, int):
        return "abs_i"
    if isinstance(x, float):
        return "abs_f"
    raise TypeError(f"not a scalar: {x!r}")


def f32_to_int(x, lo, hi):
    x = float(x)
    if math.isnan(x):
        return 0
    return int(min(max(math.trunc(x) if math.isfinite(x) else (hi if x > 0 else lo), lo), hi))


def convert_scalar(x, t):
    """Value conversion T(x) of one scalar."""
    s = scalar_type(x)
    if t == "f32":
        return F32(1.0 if x else 0.0) if s == "bool" else F32(x)
    if t == "i32":
        if s in ("f32", "abs_f"):
            return I32(f32_to_int(x, -2 ** 31, 2 ** 31 - 1))
        if s == "bool":
            return I32(1 if x else 0)
        return I32(((int(x) + 2 ** 31) % 2 ** 32) - 2 ** 31)
    if t == "u32":
        if s in ("f32", "abs_f"):
            return U32(f32_to_int(x, 0, 2 ** 32 - 1))
        if s == "bool":
            return U32(1 if x else 0)
        return U32(int(x) % 2 ** 32)
    if t == "bool":
        return bool(x != 0)
    raise TypeError(t)


def concretize(x, t):
    """An untyped literal takes the type its context asks for (AbstractInt -> i32 / u32 / f32, AbstractFloat -> f32)."""
    s = scalar_type(x)
    if s == "abs_i":
        return convert_scalar(x, t) if t in ("f32", "i32", "u32") else x
    if s == "abs_f":
        if t != "f32" and t != "abs_f":
            raise TypeError(f"an abstract float cannot become {t}")
        return F32(x) if t == "f32" else x
    return x


def unify(a, b):
    """Two scalars of one concrete type (abstract literals follow the other operand)."""
    sa, sb = scalar_type(a), scalar_type(b)
    if sa == sb:
        return a, b, sa
    if sa.startswith("abs") and not sb.startswith("abs"):
        return concretize(a, sb), b, sb
    if sb.startswith("abs") and not sa.startswith("abs"):
        return a, concretize(b, sa), sa
    if {sa, sb} == {"abs_i", "abs_f"}:
        return float(a), float(b), "abs_f"
    raise TypeError(f"operands of different types: {sa} and {sb}")


def default_concrete(x):
    """What a `let` / `var` without a type makes of an untyped literal: i32, f32."""
    if isinstance(x, Vec):
        if x.t == "abs_i":
            return Vec("i32", [I32(e) for e in x.v])
        if x.t == "abs_f":
            return Vec("f32", [F32(e) for e in x.v])
        return x
    if isinstance(x, list):   # an array value
        return [default_concrete(e) for e in x]
    s = scalar_type(x) if not isinstance(x, (Struct, Mat4)) else None
    if s == "abs_i":
        return I32(x)
    if s == "abs_f":
        return F32(x)
    return x


_INT_BITS = {"i32": (I32, -2 ** 31), "u32": (U32, 0)}


def wrap_int(v, t):
    v = int(v)
    if t == "u32":
        return U32(v % 2 ** 32)
    if t == "i32":
        return I32(((v + 2 ** 31) % 2 ** 32) - 2 ** 31)
    return v   # abstract


def binary_scalar(op, a, b):
    if op in ("&&", "||"):
        return (bool(a) and bool(b)) if op == "&&" else (bool(a) or bool(b))
    if op in ("<<", ">>"):
        ta = scalar_type(a)
        n = int(b) & 31
        if ta == "abs_i":
            return (int(a) << n) if op == "<<" else (int(a) >> n)
        return wrap_int((int(a) << n) if op == "<<" else (int(a) >> n), ta)   # >> of i32 is arithmetic (Python's is), of u32 logical
    a, b, t = unify(a, b)
    if op in ("==", "!=", "<", "<=", ">", ">="):
        return bool({"==": a == b, "!=": a != b, "<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b}[op])
    if t == "bool":
        if op in ("&", "|"):
            return bool(a & b) if op == "&" else bool(a | b)
        raise TypeError(f"{op} on bool")
    if t in ("f32", "abs_f"):
        if op == "+":
            r = a + b
        elif op == "-":
            r = a - b
        elif op == "*":
            r = a * b
        elif op == "/":
            if t == "abs_f":
                r = a / b
            else:
                r = np.divide(a, b)   # IEEE: x / 0 = +-inf, 0 / 0 = NaN
        else:
            raise TypeError(f"{op} on {t}")
        return F32(r) if t == "f32" else float(r)
    # integers
    ia, ib = int(a), int(b)
    if op == "+":
        r = ia + ib
    elif op == "-":
        r = ia - ib
    elif op == "*":
        r = ia * ib
    elif op == "/":
        r = 0 if ib == 0 else int(ia / ib) if t != "u32" else ia // ib
    elif op == "%":
        r = 0 if ib == 0 else int(math.fmod(ia, ib))
    elif op == "&":
        r = ia & ib
    elif op == "|":
        r = ia | ib
    elif op == "^":
        r = ia ^ ib
    else:
        raise TypeError(op)
    return wrap_int(r, t)


def vec_type_of(a, b):
    ta = a.t if isinstance(a, Vec) else scalar_type(a)
    tb = b.t if isinstance(b, Vec) else scalar_type(b)
    if ta == tb:
        return ta
    if ta.startswith("abs") and not tb.startswith("abs"):
        return tb
    if tb.startswith("abs") and not ta.startswith("abs"):
        return ta
    if {ta, tb} == {"abs_i", "abs_f"}:
        return "abs_f"
    raise TypeError(f"{ta} with {tb}")


def binary(op, a, b):
    if isinstance(a, Vec) and isinstance(b, Mat4):
        if op != "*":
            raise TypeError("vector (op) matrix")
        return Vec("f32", [dot(a, c) for c in b.cols])
    if isinstance(a, Mat4) and isinstance(b, Vec):
        if op != "*":
            raise TypeError("matrix (op) vector")
        # the linear combination of the columns, component i = dot(row i, v), summed left to right (no fused multiply-add)
        return Vec("f32", [dot(Vec("f32", [c.v[i] for c in a.cols]), b) for i in range(4)])
    if isinstance(a, Mat4) or isinstance(b, Mat4):
        raise TypeError("only vector * matrix and matrix * vector are implemented")
    if isinstance(a, Vec) or isinstance(b, Vec):
        n = len(a.v) if isinstance(a, Vec) else len(b.v)
        av = a.v if isinstance(a, Vec) else [a] * n
        bv = b.v if isinstance(b, Vec) else [b] * n
        if len(av) != len(bv):
            raise TypeError("vectors of different sizes")
        r = [binary_scalar(op, x, y) for x, y in zip(av, bv)]
        return Vec(scalar_type(r[0]), r)
    return binary_scalar(op, a, b)


def unary(op, a):
    if isinstance(a, Vec):
        r = [unary(op, x) for x in a.v]
        return Vec(scalar_type(r[0]), r)
    if op == "!":
        return not bool(a)
    if op == "-":
        t = scalar_type(a)
        if t == "f32":
            return F32(-a)
        if t in ("i32", "u32"):
            return wrap_int(-int(a), t)
        return -a
    if op == "~":   # bitwise complement of an integer
        t = scalar_type(a)
        if t in ("i32", "u32"):
            return wrap_int(~int(a), t)
        return ~a
    raise TypeError(op)


# ---- built-in functions (each on scalars; vectors go component by component) ----
def _f(x):
    return concretize(x, "f32") if scalar_type(x).startswith("abs") else x


def dot(a, b):
    t = F32(a.v[0]) * F32(b.v[0])
    for x, y in zip(a.v[1:], b.v[1:]):
        t = F32(t + F32(F32(x) * F32(y)))
    return F32(t)


def _min(a, b):
    a, b, t = unify(a, b)
    if t in ("f32", "abs_f"):
        if a != a:
            return b
        if b != b:
            return a
    return b if b < a else a


def _max(a, b):
    a, b, t = unify(a, b)
    if t in ("f32", "abs_f"):
        if a != a:
            return b
        if b != b:
            return a
    return b if b > a else a


def _clamp(x, lo, hi):
    return _min(_max(x, lo), hi)


def _mix(a, b, t):
    a, b, t = F32(_f(a)), F32(_f(b)), F32(_f(t))
    return F32(F32(a * F32(F32(1.0) - t)) + F32(b * t))


def _smoothstep(lo, hi, x):
    lo, hi, x = F32(_f(lo)), F32(_f(hi)), F32(_f(x))
    t = _clamp(F32(np.divide(F32(x - lo), F32(hi - lo))), F32(0.0), F32(1.0))
    return F32(F32(t * t) * F32(F32(3.0) - F32(F32(2.0) * t)))


def _sign(x):
    x = _f(x)
    if scalar_type(x) == "f32":
        return F32(1.0) if x > 0 else F32(-1.0) if x < 0 else F32(0.0) if x == 0 else x
    return wrap_int((int(x) > 0) - (int(x) < 0), scalar_type(x))


def _pow(x, y):
    x, y = float(_f(x)), float(_f(y))
    try:
        return F32(math.pow(x, y))
    except (ValueError, OverflowError):
        return F32(np.power(np.float64(x), np.float64(y)))


def componentwise(fn, *args):
    n = next((len(a.v) for a in args if isinstance(a, Vec)), None)
    if n is None:
        return fn(*args)
    cols = [a.v if isinstance(a, Vec) else [a] * n for a in args]
    r = [fn(*xs) for xs in zip(*cols)]
    return Vec(scalar_type(r[0]), r)


BUILTINS = {
    "floor": lambda x: componentwise(lambda e: F32(np.floor(F32(_f(e)))), x),
    "fract": lambda x: componentwise(lambda e: F32(F32(_f(e)) - F32(np.floor(F32(_f(e))))), x),
    "sqrt": lambda x: componentwise(lambda e: F32(np.sqrt(F32(_f(e)))), x),
    "abs": lambda x: componentwise(lambda e: F32(abs(_f(e))) if scalar_type(_f(e)) == "f32" else wrap_int(abs(int(e)), scalar_type(e)), x),
    "min": lambda a, b: componentwise(_min, a, b),
    "max": lambda a, b: componentwise(_max, a, b),
    "clamp": lambda x, lo, hi: componentwise(_clamp, x, lo, hi),
    "mix": lambda a, b, t: componentwise(_mix, a, b, t),
    "smoothstep": lambda lo, hi, x: componentwise(_smoothstep, lo, hi, x),
    "sign": lambda x: componentwise(_sign, x),
    "pow": lambda x, y: componentwise(_pow, x, y),
    "log": lambda x: componentwise(lambda e: F32(math.log(float(_f(e)))) if float(_f(e)) > 0 else F32(-np.inf) if float(_f(e)) == 0 else F32(np.nan), x),
    "cos": lambda x: componentwise(lambda e: F32(math.cos(float(_f(e)))), x),
    "sin": lambda x: componentwise(lambda e: F32(math.sin(float(_f(e)))), x),
    "exp": lambda x: componentwise(lambda e: F32(math.exp(float(_f(e)))), x),
    "dot": lambda a, b: dot(a, b),
    "length": lambda v: F32(np.sqrt(dot(v, v))),
    "distance": lambda a, b: F32(np.sqrt(dot(binary("-", a, b), binary("-", a, b)))),
    "normalize": lambda v: binary("/", v, F32(np.sqrt(dot(v, v)))),
    "any": lambda v: any(bool(e) for e in v.v) if isinstance(v, Vec) else bool(v),
    "all": lambda v: all(bool(e) for e in v.v) if isinstance(v, Vec) else bool(v),
    "select": lambda f, t, c: componentwise(lambda a, b, cc: b if cc else a, f, t, c),
}


# ------------------------------------------------------------------------------------------------------------------
# parser: source -> a small tree of tuples
# ------------------------------------------------------------------------------------------------------------------
_VEC = {"vec2": 2, "vec3": 3, "vec4": 4}
_PREC = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", "<=", ">", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]


class Parser:
    def __init__(self, src):
        self.t = tokenize(src)
        self.i = 0
        self.no_gt = 0   # inside template brackets '>' closes the list

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, val):
        if self.t[self.i][1] == val and self.t[self.i][0] in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError(f"expected {val!r}, found {self.t[self.i][1]!r} (token {self.i})")

    def ident(self):
        kind, v = self.next()
        if kind != "id":
            raise SyntaxError(f"expected an identifier, found {v!r}")
        return v

    # ---- types ----
    def parse_type(self):
        name = self.ident()
        args = []
        if self.peek()[1] == "<":
            self.next()
            while True:
                if self.peek()[0] == "num":     # array<T, N>
                    args.append((self.next()[1], ()))
                else:
                    args.append(self.parse_type())
                if self.accept(","):
                    continue
                break
            self.expect(">")
        return (name, tuple(args))

    def skip_attributes(self):
        attrs = {}
        while self.accept("@"):
            name = self.ident()
            vals = []
            if self.accept("("):
                while not self.accept(")"):
                    vals.append(self.next()[1])
                    self.accept(",")
            attrs[name] = vals
        return attrs

    # ---- module ----
    def parse_module(self):
        structs, globals_, funcs = {}, {}, {}
        while self.peek()[0] != "eof":
            attrs = self.skip_attributes()
            if self.accept("struct"):
                name = self.ident()
                self.expect("{")
                fields = []
                while not self.accept("}"):
                    self.skip_attributes()
                    f = self.ident()
                    self.expect(":")
                    fields.append((f, self.parse_type()))
                    self.accept(",")
                self.accept(";")
                structs[name] = fields
            elif self.accept("var"):
                space = []
                if self.accept("<"):
                    while not self.accept(">"):
                        space.append(self.next()[1])
                        self.accept(",")
                name = self.ident()
                self.expect(":")
                ty = self.parse_type()
                self.expect(";")
                globals_[name] = (ty, attrs, space)
            elif self.accept("fn"):
                name = self.ident()
                self.expect("(")
                params = []
                while not self.accept(")"):
                    pattrs = self.skip_attributes()
                    p = self.ident()
                    self.expect(":")
                    params.append((p, self.parse_type(), pattrs))
                    self.accept(",")
                ret = None
                if self.accept("->"):
                    self.skip_attributes()
                    ret = self.parse_type()
                funcs[name] = (params, ret, self.parse_block(), attrs)
            elif self.accept("const") or self.accept("alias"):
                raise SyntaxError("module-scope const / alias: not in the subset")
            else:
                raise SyntaxError(f"unexpected {self.peek()[1]!r} at module scope")
        return structs, globals_, funcs

    # ---- statements ----
    def parse_block(self):
        self.expect("{")
        body = []
        while not self.accept("}"):
            body.append(self.parse_statement())
        return body

    def parse_statement(self):
        tok = self.peek()[1]
        if tok == "{":
            return ("block", self.parse_block())
        if tok in ("let", "var"):
            self.next()
            name = self.ident()
            ty = None
            if self.accept(":"):
                ty = self.parse_type()
            init = None
            if self.accept("="):
                init = self.parse_expr()
            self.expect(";")
            return ("decl", name, ty, init)
        if tok == "if":
            self.next()
            cond = self.parse_expr()
            then = self.parse_block()
            other = None
            if self.accept("else"):
                other = [self.parse_statement()] if self.peek()[1] == "if" else self.parse_block()
            return ("if", cond, then, other)
        if tok == "loop":
            self.next()
            return ("loop", self.parse_block())
        if tok == "while":
            self.next()
            cond = self.parse_expr()
            return ("while", cond, self.parse_block())
        if tok == "break":
            self.next()
            self.expect(";")
            return ("break",)
        if tok == "continue":
            self.next()
            self.expect(";")
            return ("continue",)
        if tok == "return":
            self.next()
            e = None if self.peek()[1] == ";" else self.parse_expr()
            self.expect(";")
            return ("return", e)
        lhs = self.parse_unary()
        op = self.peek()[1]
        if op in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
            self.next()
            rhs = self.parse_expr()
            self.expect(";")
            return ("assign", lhs, op, rhs)
        self.expect(";")
        return ("expr", lhs)

    # ---- expressions ----
    def parse_expr(self, level=0):
        if level == len(_PREC):
            return self.parse_unary()
        lhs = self.parse_expr(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in _PREC[level] and not (self.no_gt and self.peek()[1] in (">", ">>", ">=")):
            op = self.next()[1]
            rhs = self.parse_expr(level + 1)
            lhs = ("bin", op, lhs, rhs)
        return lhs

    def parse_unary(self):
        if self.peek()[0] == "op" and self.peek()[1] in ("-", "!", "~"):
            op = self.next()[1]
            return ("un", op, self.parse_unary())
        if self.peek()[0] == "op" and self.peek()[1] == "&":     # address-of: a pointer to a function-scope variable
            self.next()
            return ("addr", self.parse_unary())
        if self.peek()[0] == "op" and self.peek()[1] == "*":     # indirection
            self.next()
            return ("deref", self.parse_unary())
        return self.parse_postfix(self.parse_primary())

    def parse_postfix(self, e):
        while True:
            if self.accept("."):
                e = ("member", e, self.ident())
            elif self.accept("["):
                idx = self.parse_expr()
                self.expect("]")
                e = ("index", e, idx)
            else:
                return e

    def parse_args(self):
        self.expect("(")
        args = []
        while not self.accept(")"):
            args.append(self.parse_expr())
            self.accept(",")
        return args

    def parse_primary(self):
        kind, v = self.next()
        if kind == "num":
            return ("lit", parse_number(v))
        if v == "(":
            e = self.parse_expr()
            self.expect(")")
            return e
        if kind != "id":
            raise SyntaxError(f"unexpected {v!r} in an expression")
        if v in ("true", "false"):
            return ("lit", v == "true")
        if v in _VEC or v in ("array", "mat4x4"):
            targ = None
            if self.peek()[1] == "<":
                self.i -= 1
                full = self.parse_type()        # vec3<f32>, array<vec2<f32>, 6>
                targ = full[1][0] if full[1] else None
            return ("construct", v, targ, self.parse_args())
        if self.peek()[1] == "(":
            return ("call", v, self.parse_args())
        return ("var", v)


def parse_number(s):
    if s[:2].lower() == "0x":
        suffix = s[-1] if s[-1] in "iu" else ""
        v = int(s[:-1] if suffix else s, 16)
        return U32(v) if suffix == "u" else I32(v) if suffix == "i" else v
    suffix = s[-1] if s[-1] in "fiu" else ""
    body = s[:-1] if suffix else s
    if suffix == "u":
        return U32(int(body))
    if suffix == "i":
        return I32(int(body))
    if suffix == "f":
        return F32(float(body))
    return float(body) if any(c in body for c in ".eE") else int(body)


# ------------------------------------------------------------------------------------------------------------------
# evaluator
# ------------------------------------------------------------------------------------------------------------------
class _Break(Exception):
    pass


class _Continue(Exception):
    pass


class _Return(Exception):
    def __init__(self, value):
        self.value = value


_SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def zero_like(x):
    """A zero value of x's type (a scalar, Vec, Mat4 or Struct)."""
    if isinstance(x, Struct):
        return Struct(x.name, {k: zero_like(v) for k, v in x.f.items()})
    if isinstance(x, Vec):
        return Vec(x.t, [zero_like(e) for e in x.v])
    if isinstance(x, Mat4):
        return Mat4([zero_like(c) for c in x.cols])
    if isinstance(x, bool):
        return False
    return type(x)(0)


class Module:
    """A parsed WGSL module with its resource bindings.  `bind(name, value)` sets a module-scope variable (a Struct for a
    uniform, a Python sequence for a storage array — arrays are only ever indexed); `call(name, *args)` runs a function;
    `hooks[name] = fn(locals, result)` is called whenever function `name` returns; `texture_stores` collects textureStore calls."""

    def __init__(self, src: str, oob: str = "refuse"):
        self.structs, self.globals, self.funcs = Parser(src).parse_module()
        # what an array read past its end yields is left to the implementation (WGSL 'out-of-bounds access': any in-bounds
        # element or zero).  "refuse": such a read is an error here (the default: a fixture then cannot depend on it);
        # "clamp": the index is clamped to the last element (naga's BoundsCheckPolicy::Restrict); "zero": the read yields a
        # zero value of the element's type (ReadZeroSkipWrite).  tests/golden/wgsl_oob.npz holds both for the two reads of
        # ray_tracer.wgsl that can go past their arrays (:121-124 chunk_roots_, :226 voxel_mats).
        assert oob in ("refuse", "clamp", "zero")
        self.oob = oob
        self.bound = {}
        self.hooks = {}
        self.externals = {}
        self.texture_stores = []

    def bind(self, name, value):
        if name not in self.globals:
            raise KeyError(name)
        self.bound[name] = value

    # ---- construction of values from types ----
    def zero(self, ty):
        name, args = ty
        if name in ("f32",):
            return F32(0)
        if name == "i32":
            return I32(0)
        if name == "u32":
            return U32(0)
        if name == "bool":
            return False
        if name in _VEC:
            t = args[0][0]
            return Vec(t, [self.zero((t, ()))] * _VEC[name])
        if name == "mat4x4":
            return Mat4([Vec("f32", [F32(0)] * 4) for _ in range(4)])
        if name in self.structs:
            return Struct(name, {f: self.zero(t) for f, t in self.structs[name]})
        raise TypeError(f"no zero value for {name}")

    def coerce(self, value, ty):
        """A value initialising / assigned to a place of declared type `ty` (only untyped literals change)."""
        if ty is None:
            return default_concrete(value)
        name, args = ty
        if name in ("f32", "i32", "u32"):
            return concretize(value, name)
        if name in _VEC and isinstance(value, Vec) and value.t.startswith("abs"):
            t = args[0][0]
            return Vec(t, [concretize(e, t) for e in value.v])
        return value

    # ---- expressions ----
    def eval(self, e, env):
        k = e[0]
        if k == "lit":
            return e[1]
        if k == "var":
            return self.lookup(e[1], env)
        if k == "bin":
            op = e[1]
            if op == "&&":
                a = self.eval(e[2], env)
                return bool(a) and bool(self.eval(e[3], env))
            if op == "||":
                a = self.eval(e[2], env)
                return bool(a) or bool(self.eval(e[3], env))
            return binary(op, self.eval(e[2], env), self.eval(e[3], env))
        if k == "un":
            return unary(e[1], self.eval(e[2], env))
        if k == "addr":
            if e[1][0] != "var":
                raise TypeError("& of something that is not a variable")
            for scope in reversed(env):
                if e[1][1] in scope:
                    return Ref(scope, e[1][1])
            raise NameError(e[1][1])
        if k == "deref":
            r = self.eval(e[1], env)
            return r.scope[r.name]
        if k == "member":
            base = self.eval(e[1], env)
            return self.member(base, e[2])
        if k == "index":
            base = self.eval(e[1], env)
            idx = int(self.eval(e[2], env))
            if isinstance(base, Vec):
                return base.v[idx]
            if isinstance(base, Mat4):
                return base.cols[idx]
            if idx < 0 or idx >= len(base):
                if self.oob == "clamp":
                    return base[min(max(idx, 0), len(base) - 1)]
                if self.oob == "zero":
                    return zero_like(base[0])
                raise IndexError(f"array index {idx} outside [0, {len(base)}): what such a read yields is implementation-defined")
            return base[idx]
        if k == "construct":
            return self.construct(e[1], e[2], [self.eval(a, env) for a in e[3]])
        if k == "call":
            return self.call_expr(e[1], [self.eval(a, env) for a in e[2]])
        raise TypeError(k)

    def lookup(self, name, env):
        for scope in reversed(env):
            if name in scope:
                return scope[name]
        if name in self.bound:
            return self.bound[name]
        raise NameError(name)

    @staticmethod
    def member(base, name):
        if isinstance(base, Struct):
            return base.f[name]
        if isinstance(base, Vec):
            idx = [_SWZ[c] for c in name]
            return base.v[idx[0]] if len(idx) == 1 else Vec(base.t, [base.v[i] for i in idx])
        raise TypeError(f".{name} of {base!r}")

    def construct(self, name, targ, args):
        if name in _VEC:
            n = _VEC[name]
            flat = []
            for a in args:
                flat.extend(a.v if isinstance(a, Vec) else [a])
            if len(flat) == 1:
                flat = flat * n
            if len(flat) != n:
                raise TypeError(f"{name} from {len(flat)} components")
            if targ is not None:
                t = targ[0]
                return Vec(t, [convert_scalar(x, t) for x in flat])
            t = None
            for x in flat:   # all components take one type; untyped literals follow the typed ones
                s = scalar_type(x)
                if not s.startswith("abs"):
                    t = s
                    break
            if t is None:
                t = "abs_f" if any(scalar_type(x) == "abs_f" for x in flat) else "abs_i"
                return Vec(t, [float(x) if t == "abs_f" else x for x in flat])
            return Vec(t, [concretize(x, t) for x in flat])
        if name == "array":
            return [default_concrete(copy_value(a)) for a in args]
        if name == "mat4x4":   # four columns, or sixteen scalars column by column
            flat = []
            for a in args:
                flat.extend(a.v if isinstance(a, Vec) else [a])
            if len(flat) != 16:
                raise TypeError(f"mat4x4 from {len(flat)} components")
            return Mat4([Vec("f32", [convert_scalar(x, "f32") for x in flat[4 * c:4 * c + 4]]) for c in range(4)])
        raise TypeError(f"constructor {name}")

    def call_expr(self, name, args):
        if name in ("f32", "i32", "u32", "bool"):
            a = args[0]
            if isinstance(a, Vec):
                return Vec(name, [convert_scalar(x, name) for x in a.v])
            return convert_scalar(a, name)
        if name == "textureStore":
            self.texture_stores.append((args[1], args[2]))
            return None
        if name in self.externals:     # what the module's environment provides (textureSample: the sampler is not WGSL text)
            return self.externals[name](*args)
        if name in self.funcs:
            return self.call(name, *args)
        if name in BUILTINS:
            return BUILTINS[name](*args)
        if name in self.structs:
            return Struct(name, {f: self.coerce(copy_value(a), t) for (f, t), a in zip(self.structs[name], args)})
        raise NameError(f"function {name}")

    # ---- statements ----
    def call(self, name, *args):
        params, ret, body, _ = self.funcs[name]
        frame = {p: self.coerce(copy_value(a), t) for (p, t, _), a in zip(params, args)}
        env = [frame]
        result = None
        try:
            self.exec_block(body, env, new_scope=False)
        except _Return as r:
            result = r.value
        if name in self.hooks:
            self.hooks[name](frame, result)
        return result

    def exec_block(self, body, env, new_scope=True):
        if new_scope:
            env.append({})
        try:
            for s in body:
                self.exec(s, env)
        finally:
            if new_scope:
                env.pop()

    def exec(self, s, env):
        k = s[0]
        if k == "decl":
            _, name, ty, init = s
            v = self.zero(ty) if init is None else self.coerce(copy_value(self.eval(init, env)), ty)
            # function-scope names live in the function's own frame as well, so that a hook sees them after the return
            env[-1][name] = v
            if len(env) > 1:
                env[0].setdefault("__locals__", {})[name] = None
                env[0]["__locals__"][name] = v
        elif k == "assign":
            _, lhs, op, rhs = s
            val = self.eval(rhs, env)
            if op != "=":
                val = binary(op[:-1], self.eval(lhs, env), val)
            self.store(lhs, val, env)
        elif k == "if":
            if bool(self.eval(s[1], env)):
                self.exec_block(s[2], env)
            elif s[3] is not None:
                self.exec_block(s[3], env)
        elif k == "loop":
            while True:
                try:
                    self.exec_block(s[1], env)
                except _Break:
                    break
                except _Continue:
                    continue
        elif k == "while":
            while bool(self.eval(s[1], env)):
                try:
                    self.exec_block(s[2], env)
                except _Break:
                    break
                except _Continue:
                    continue
        elif k == "break":
            raise _Break()
        elif k == "continue":
            raise _Continue()
        elif k == "return":
            raise _Return(None if s[1] is None else copy_value(self.eval(s[1], env)))
        elif k == "expr":
            self.eval(s[1], env)
        elif k == "block":
            self.exec_block(s[1], env)
        else:
            raise TypeError(k)

    def store(self, lhs, val, env):
        """lhs = val through a path of members / swizzle components / indices that ends in a local variable."""
        if lhs[0] == "var":
            name = lhs[1]
            for scope in reversed(env):
                if name in scope:
                    old = scope[name]
                    new = copy_value(val)
                    if isinstance(old, Vec) and isinstance(new, Vec) and new.t.startswith("abs"):
                        new = Vec(old.t, [concretize(e, old.t) for e in new.v])
                    elif not isinstance(old, (Vec, Struct, Mat4)) and not isinstance(new, (Vec, Struct, Mat4)):
                        new = concretize(new, scalar_type(old))
                    scope[name] = new
                    if "__locals__" in env[0] and name in env[0]["__locals__"]:
                        env[0]["__locals__"][name] = new
                    return
            raise NameError(f"assignment to {name}")
        if lhs[0] == "member":
            base = self.eval(lhs[1], env)   # containers are mutable objects: the place itself
            name = lhs[2]
            if isinstance(base, Struct):
                old = base.f[name]
                new = copy_value(val)
                if isinstance(old, Vec) and isinstance(new, Vec) and new.t.startswith("abs"):
                    new = Vec(old.t, [concretize(e, old.t) for e in new.v])
                elif not isinstance(old, (Vec, Struct, Mat4)):
                    new = concretize(new, scalar_type(old))
                base.f[name] = new
                return
            if isinstance(base, Vec):
                idx = [_SWZ[c] for c in name]
                vals = val.v if isinstance(val, Vec) else [val]
                for i, x in zip(idx, vals):
                    base.v[i] = concretize(x, base.t)
                return
        if lhs[0] == "index":
            base = self.eval(lhs[1], env)
            i = int(self.eval(lhs[2], env))
            if isinstance(base, Vec):
                base.v[i] = concretize(val, base.t)
                return
            if isinstance(base, list):   # an element of a function-scope array
                old = base[i]
                base[i] = concretize(val, scalar_type(old)) if not isinstance(old, (Vec, Struct, Mat4, list)) else copy_value(val)
                return
        if lhs[0] == "deref":
            r = self.eval(lhs[1], env)
            old = r.scope[r.name]
            r.scope[r.name] = concretize(val, scalar_type(old)) if not isinstance(old, (Vec, Struct, Mat4)) else copy_value(val)
            return
        raise TypeError(f"cannot assign through {lhs[0]}")
