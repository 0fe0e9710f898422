#!/usr/bin/env python3
"""DEV-TIME ONLY: a small interpreter for the subset of WGSL that the reference's skinning path is written in.

Nothing in this image can execute a shader (no WebGPU, naga, tint, wgpu). What CAN be done is to take the reference's
shader TEXT — the body of `@vertex fn vs(...)` (engine/src/engine.ts:245-276) and of the skin-matrix compute shader's
`fn main(...)` (:919-928) — out of /root/reference at fixture-generation time, parse it, and evaluate it statement by
statement on real inputs. The formula is then not re-typed by this build: a changed or mis-read line of the shader changes
the numbers. `tools/ref_wgsl_run.py` uses this to write tests/golden/ref_wgsl.npz; the oracle is held to those vectors.

Evaluation model (WGSL leaves precision and the association of the sums inside matrix products to the implementation; this is
the one the oracle documents as canonical): every scalar operation is ONE IEEE binary32 operation, no fused multiply-add;
    mat * vec   = ((m[0]*v.x + m[1]*v.y) + m[2]*v.z) + m[3]*v.w     columns left to right
    mat * mat   = columns: (a * b)[c] = a * b[c]
    vec op vec, vec * scalar, scalar * vec: component-wise
    normalize(v) = v / sqrt((v.x*v.x + v.y*v.y) + v.z*v.z)
    select(f, t, cond) = cond ? t : f
Supported syntax: let / var declarations (with or without a type), assignment and `+=` to names, members and indexed
elements, `for (var i = 0u; i < Nu; i++) { ... }`, `if (cond) { return; }`, `return`, calls to vec2f / vec3f / vec4f /
mat3x3f / select / normalize, member access (.x .y .z .w .xyz, struct fields), indexing, + - * / and comparisons.
Anything else raises — silently skipping shader code would defeat the purpose. Statements that touch bindings the caller did
not provide (the camera uniforms behind `output.position`) are reported back as skipped, by name.
"""
import re

import numpy as np

F = np.float32


class Unbound(Exception):
    pass


class _Return(Exception):
    pass


TOKEN = re.compile(r"\s*(?:(//[^\n]*)|(\d+\.\d*(?:[eE][-+]?\d+)?f?|\d+[uif]?)|([A-Za-z_]\w*)|(\+\+|\+=|-=|\*=|<=|>=|==|!=|&&|\|\||[-+*/<>=.,;:()\[\]{}@!]))")


def tokenize(src):
    out, pos = [], 0
    src = src.rstrip()
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            if src[pos:].strip() == "":
                break
            raise SyntaxError("cannot tokenize WGSL at: %r" % src[pos:pos + 40])
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            out.append(("num", m.group(2)))
        elif m.group(3):
            out.append(("id", m.group(3)))
        else:
            out.append(("op", m.group(4)))
    return out


def function_body(source, header_regex, containing=None):
    """Text between the braces of the first function whose header matches `header_regex` (and whose body contains
    `containing`, when given: the reference has three `@vertex fn vs`), and the header itself."""
    if containing is not None:
        for m in re.finditer(header_regex, source):
            body, head = function_body(source[m.start():], header_regex)
            if containing in body:
                return body, head
        raise ValueError("no function matching %s contains %r" % (header_regex, containing))
    m = re.search(header_regex, source)
    if not m:
        raise ValueError("function not found: " + header_regex)
    i = source.index("{", m.end() - 1) if source[m.end() - 1] != "{" else m.end() - 1
    depth, j = 0, i
    while True:
        if source[j] == "{":
            depth += 1
        elif source[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return source[i + 1:j], source[m.start():i]


# ---------------------------------------------------------------- values
class Mat:
    """column-major matrix: cols[c] is a float32 vector of R rows"""
    def __init__(self, cols):
        self.cols = [np.asarray(c, dtype=F) for c in cols]


def _mat_vec(m, v):
    acc = None
    for c, col in enumerate(m.cols):
        term = (col * F(v[c])).astype(F)
        acc = term if acc is None else (acc + term).astype(F)
    return acc


def _mul(a, b):
    if isinstance(a, Mat) and isinstance(b, Mat):
        return Mat([_mat_vec(a, col) for col in b.cols])
    if isinstance(a, Mat):
        return _mat_vec(a, b)
    if isinstance(b, Mat):
        raise TypeError("vec * mat is not used by the skinning path")
    return (np.asarray(a, dtype=F) * np.asarray(b, dtype=F)).astype(F) if isinstance(a, np.ndarray) or isinstance(b, np.ndarray) else F(F(a) * F(b))


def _arith(op, a, b):
    if op == "*":
        return _mul(a, b)
    if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)) and not isinstance(a, (bool,)):
        return {"+": a + b, "-": a - b, "/": a // b}[op]
    a32, b32 = (np.asarray(a, dtype=F) if isinstance(a, np.ndarray) else F(a)), (np.asarray(b, dtype=F) if isinstance(b, np.ndarray) else F(b))
    r = {"+": lambda: a32 + b32, "-": lambda: a32 - b32, "/": lambda: a32 / b32}[op]()
    return r.astype(F) if isinstance(r, np.ndarray) else F(r)


SWZ = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


class Interp:
    def __init__(self, tokens, env, builtins=None):
        self.t, self.i, self.env, self.skipped = tokens, 0, dict(env), []

    # ---- token helpers
    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def take(self, val=None):
        tok = self.peek()
        if val is not None and tok[1] != val:
            raise SyntaxError("expected %r, got %r (token %d)" % (val, tok[1], self.i))
        self.i += 1
        return tok

    # ---- statements
    def run(self):
        try:
            while self.peek()[0] != "eof":
                self.statement(True)
        except _Return:
            pass
        return self.env

    def block(self, execute):
        self.take("{")
        while self.peek()[1] != "}":
            self.statement(execute)
        self.take("}")

    def skip_type(self):
        depth = 0
        while True:
            v = self.peek()[1]
            if v in ("=", ";") and depth == 0:
                return
            if v == "<":
                depth += 1
            elif v == ">":
                depth -= 1
            self.i += 1

    def statement(self, execute):
        kind, v = self.peek()
        if v in ("let", "var"):
            self.take()
            name = self.take()[1]
            if self.peek()[1] == ":":
                self.take(":")
                self.skip_type()
            val = {}
            if self.peek()[1] == "=":
                self.take("=")
                val = self.guarded(execute, name)
            self.take(";")
            if execute and val is not Unbound:
                self.env[name] = val
            return
        if v == "for":
            self.take(); self.take("(")
            assert self.take()[1] == "var"
            var = self.take()[1]
            self.take("=")
            start = self.expr(True)
            self.take(";")
            assert self.take()[1] == var
            cmp_op = self.take()[1]
            limit = self.expr(True)
            self.take(";")
            assert self.take()[1] == var and self.take()[1] == "++"
            self.take(")")
            body_at = self.i
            n = int(limit) - int(start) if cmp_op == "<" else int(limit) - int(start) + 1
            if not execute or n <= 0:
                self.block(False)
                return
            for k in range(n):
                self.i = body_at
                self.env[var] = int(start) + k
                self.block(True)
            return
        if v == "if":
            self.take(); self.take("(")
            cond = self.expr(execute)
            self.take(")")
            self.block(execute and bool(cond))
            if self.peek()[1] == "else":
                self.take()
                self.block(execute and not bool(cond))
            return
        if v == "return":
            self.take()
            val = None
            if self.peek()[1] != ";":
                val = self.guarded(execute, "return")
            self.take(";")
            if execute:
                self.env["__return__"] = val
                raise _Return()
            return
        # assignment:  lvalue (= | +=) expr ;
        target = [self.take()[1]]
        while self.peek()[1] in (".", "["):
            if self.take()[1] == ".":
                target.append(("m", self.take()[1]))
            else:
                idx = self.expr(execute)
                self.take("]")
                target.append(("i", idx))
        op = self.take()[1]
        if op not in ("=", "+="):
            raise SyntaxError("unsupported statement starting at %r %r" % (target, op))
        val = self.guarded(execute, ".".join(str(x if isinstance(x, str) else x[1]) for x in target))
        self.take(";")
        if not execute or val is Unbound:
            return
        self.assign(target, op, val)

    def guarded(self, execute, what):
        """evaluate the expression; a reference to a binding the caller did not supply skips the statement (and is reported)"""
        start = self.i
        try:
            return self.expr(execute)
        except Unbound as e:
            self.skipped.append("%s  (needs %s)" % (what, e))
            self.i = start
            depth = 0
            while not (self.peek()[1] == ";" and depth == 0):
                depth += {"(": 1, "[": 1, ")": -1, "]": -1}.get(self.peek()[1], 0)
                self.i += 1
            return Unbound

    def assign(self, target, op, val):
        name = target[0]
        if len(target) == 1:
            self.env[name] = _arith("+", self.env[name], val) if op == "+=" else val
            return
        obj = self.env.setdefault(name, {})
        for kind, key in target[1:-1]:
            obj = obj[key]
        kind, key = target[-1]
        if kind == "m" and isinstance(obj, dict):
            obj[key] = _arith("+", obj[key], val) if op == "+=" else val
        elif kind == "m":
            obj[SWZ[key]] = _arith("+", obj[SWZ[key]], val) if op == "+=" else F(val)
        else:
            obj[int(key)] = _arith("+", obj[int(key)], val) if op == "+=" else val

    # ---- expressions
    def expr(self, ex):
        left = self.additive(ex)
        while self.peek()[1] in ("<", ">", "<=", ">=", "==", "!="):
            op = self.take()[1]
            right = self.additive(ex)
            if ex:
                left = {"<": left < right, ">": left > right, "<=": left <= right, ">=": left >= right, "==": left == right, "!=": left != right}[op]
        return left

    def additive(self, ex):
        left = self.multiplicative(ex)
        while self.peek()[1] in ("+", "-"):
            op = self.take()[1]
            right = self.multiplicative(ex)
            if ex:
                left = _arith(op, left, right)
        return left

    def multiplicative(self, ex):
        left = self.unary(ex)
        while self.peek()[1] in ("*", "/"):
            op = self.take()[1]
            right = self.unary(ex)
            if ex:
                left = _arith(op, left, right)
        return left

    def unary(self, ex):
        if self.peek()[1] == "-":
            self.take()
            v = self.unary(ex)
            return (-v if ex else v)
        return self.postfix(ex)

    def postfix(self, ex):
        v = self.primary(ex)
        while self.peek()[1] in (".", "["):
            if self.take()[1] == ".":
                name = self.take()[1]
                if not ex:
                    continue
                if isinstance(v, dict):
                    if name not in v:
                        raise Unbound(name)
                    v = v[name]
                elif all(ch in SWZ for ch in name):
                    integer = np.issubdtype(np.asarray(v).dtype, np.integer)
                    if len(name) == 1:
                        v = int(v[SWZ[name]]) if integer else F(v[SWZ[name]])
                    else:
                        v = np.array([v[SWZ[ch]] for ch in name], dtype=np.asarray(v).dtype if integer else F)
                else:
                    raise SyntaxError("member ." + name)
            else:
                idx = self.expr(ex)
                self.take("]")
                if not ex:
                    continue
                if isinstance(v, Mat):
                    v = v.cols[int(idx)]
                elif isinstance(v, (list, tuple)):
                    v = v[int(idx)]
                else:
                    e = v[int(idx)]
                    v = e if isinstance(e, (Mat, dict)) else (int(e) if np.issubdtype(np.asarray(e).dtype, np.integer) else F(e))
        return v

    def call_args(self, ex):
        self.take("(")
        args = []
        while self.peek()[1] != ")":
            args.append(self.expr(ex))
            if self.peek()[1] == ",":
                self.take()
        self.take(")")
        return args

    def primary(self, ex):
        kind, v = self.take()
        if kind == "num":
            if v.endswith("u") or v.endswith("i"):
                return int(v[:-1])
            if "." in v or "e" in v.lower() or v.endswith("f"):
                return F(v.rstrip("f"))
            return int(v)
        if v == "(":
            e = self.expr(ex)
            self.take(")")
            return e
        if kind != "id":
            raise SyntaxError("unexpected token %r" % v)
        if self.peek()[1] == "<":                          # vec4<f32>( ... ) style constructor
            depth = 0
            while True:
                t = self.take()[1]
                depth += {"<": 1, ">": -1}.get(t, 0)
                if depth == 0:
                    break
        if self.peek()[1] == "(":
            args = self.call_args(ex)
            if not ex:
                return None
            if v in ("vec2f", "vec3f", "vec4f", "vec2", "vec3", "vec4"):
                n = int(v[3])
                flat = []
                for a in args:
                    flat.extend(list(a) if isinstance(a, np.ndarray) else [a])
                if len(flat) == 1:
                    flat = flat * n
                if len(flat) != n:
                    raise SyntaxError("%s() with %d components" % (v, len(flat)))
                return np.array(flat, dtype=F)
            if v in ("mat3x3f", "mat4x4f"):
                return Mat(args)
            if v == "select":
                f, t, c = args
                return t if bool(c) else f
            if v == "normalize":
                x = np.asarray(args[0], dtype=F)
                d = F(0)
                first = True
                for c in x:
                    sq = F(c * c)
                    d = sq if first else F(d + sq)
                    first = False
                return (x / F(np.sqrt(d))).astype(F)
            raise SyntaxError("function %s() is not part of the supported subset" % v)
        if v in ("true", "false"):
            return v == "true"
        if not ex:
            return None
        if v not in self.env:
            raise Unbound(v)
        return self.env[v]


def run_function(source, header_regex, env):
    """Interpret one function body. Returns (environment after the run, list of skipped statements, header text)."""
    body, header = function_body(source, header_regex)
    it = Interp(tokenize(body), env)
    out = it.run()
    return out, it.skipped, header
