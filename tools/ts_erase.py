#!/usr/bin/env python3
"""Type eraser for the host package: reze-engine_amd/host/src/*.ts (authored, annotated TypeScript) -> reze-engine_amd/host/*.js (what Node
runs). The image has no `tsc`; TypeScript's annotations are erasable by construction, and the host sources keep to a subset in which
erasure is purely textual and line-local, so that the shipped .js is the .ts minus its types — same lines otherwise, same comments:

  * `interface X { ... }`, `type X = ...`, `declare ...` and `import type ...` statements at the top level -> removed (whole statement)
  * `import { A, B } from './x'` / `import * as fs from 'fs'` / `export { A, B }` (top level, one line)  -> `const { A, B } = require('./x')` /
    `const fs = require('fs')` / `module.exports = { A, B }`: the shipped package is CommonJS (Node 12 loads it as it is)
  * class field declarations without an initialiser (`  name: T`, `  readonly name?: T`)            -> removed (whole line)
  * `function f(a: T, b?: U, ...r: V[]): R {`, `method(a: T): R {`, `constructor(private x: T) {`,
    `get p(): T {`, `static async m(a: T): Promise<R> {` — one line or a parameter per line        -> `f(a, b, ...r) {`
  * `const x: T = e` / `let x: T = e` / `let x: T`                                                   -> `const x = e` / `let x`
  * `(e as T)` casts where T is a plain type name, optionally `[]` / `<...>`                        -> `(e)`
Nothing else is touched: arrow functions carry no annotations in these sources (their parameter types are inferred or implicit),
there are no enums, namespaces, decorators, parameter properties or non-null assertions; a method may carry a type-parameter list (`guard<T>(...)`).

    python tools/ts_erase.py            regenerate every host/*.js from host/src/*.ts
    python tools/ts_erase.py --check    exit 1 when a shipped .js differs from what its .ts erases to (tests/test_host_js.py)
    python tools/ts_erase.py FILE.ts    print the erasure of one file
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "reze-engine_amd", "host", "src")
DST = os.path.join(ROOT, "reze-engine_amd", "host")

OPEN, CLOSE = "([{<", ")]}>"


def split_top(s, sep=","):
    """split at top-level separators; '<' '>' only count as brackets when they look like generics (not ' < ' / ' > ' / '=>')"""
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        ch = s[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and i > 0 and (s[i - 1].isalnum() or s[i - 1] == "_"):
            depth += 1
        elif ch == ">" and i > 0 and s[i - 1] != "=" and depth > 0 and not (s[i - 1] == " "):
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip() or out:
        out.append(cur)
    return out


def top_level_index(s, needle):
    """index of the first top-level occurrence of `needle` (not inside brackets / generics), or -1"""
    depth, i = 0, 0
    while i < len(s):
        ch = s[i]
        if s.startswith(needle, i) and depth == 0:
            return i
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == "<" and i > 0 and (s[i - 1].isalnum() or s[i - 1] == "_"):
            depth += 1
        elif ch == ">" and i > 0 and s[i - 1] not in "= " and depth > 0:
            depth -= 1
        i += 1
    return -1


def erase_param(p):
    """`private readonly a?: T = d` -> `a = d`"""
    lead = re.match(r"^\s*", p).group(0)
    body = p.strip()
    if not body:
        return p
    body = re.sub(r"^((private|public|protected|readonly)\s+)+", "", body)
    m = re.match(r"^(\.\.\.)?([A-Za-z_$][\w$]*)\??\s*:", body)
    if not m:
        return lead + body
    rest = body[m.end():]
    eq = top_level_index(rest, " = ")
    default = rest[eq:] if eq >= 0 else ""
    return lead + (m.group(1) or "") + m.group(2) + default


def erase_params(s):
    return ",".join(erase_param(p) for p in split_top(s)) if s.strip() else s


SIG_HEAD = re.compile(r"^(\s*)((?:export\s+)?(?:(?:static|async|get|set)\s+)*(?:function\s*\*?\s*)?(?:[A-Za-z_$][\w$]*|constructor)?)\($")
SIG_LINE = re.compile(r"^(\s*)((?:(?:static|async|get|set)\s+)*(?:function\s*\*?\s*)?(?:[A-Za-z_$][\w$]*)?)(<[^<>()]*>)?\(")
KEYWORDS = {"if", "for", "while", "switch", "catch", "return", "typeof", "await", "new", "throw", "else", "do", "super", "function"}


def match_paren(s, start):
    """index of the ')' matching the '(' at s[start]; -1 if it is not on this line (strings are not expected inside signatures)"""
    depth = 0
    for i in range(start, len(s)):
        if s[i] in "([{":
            depth += 1
        elif s[i] in ")]}":
            depth -= 1
            if depth == 0:
                return i
    return -1


def erase_signature_line(code):
    """a one-line function / method head `name(params): R {` (possibly followed by a body on the same line) -> types removed; None if
    the line is not one"""
    m = SIG_LINE.match(code)
    if not m:
        return None
    head = m.group(2).strip()
    words = head.split()
    name = words[-1] if words else ""
    is_fn = head.startswith("function") or "function" in words
    if not is_fn and (name in KEYWORDS or not name):
        return None
    close = match_paren(code, m.end() - 1)
    if close < 0:
        return None
    after = code[close + 1:]
    # a head is followed by `{` — directly or behind a return type; a call statement is not
    am = re.match(r"^(\s*:\s*(?P<ret>.+?))?\s*\{(?P<rest>.*)$", after)
    if not am:
        return None
    if am.group("ret") is not None:
        # the return type ends at the LAST top-level ' {' that opens the body: find the split where the type part has balanced brackets
        full = after
        colon = full.index(":")
        k = top_level_index(full[colon + 1:], " {")
        if k < 0:
            return None
        after = " {" + full[colon + 1 + k + 2:]
    elif not is_fn and not re.match(r"^\s*\{", after):
        return None
    params = code[m.end():close]
    return m.group(1) + m.group(2) + "(" + erase_params(params) + ")" + after       # (a type-parameter list `<T>` behind the name goes too)


def erase(src):
    lines = src.split("\n")
    out = []
    i = 0
    depth_class = []       # brace depth at which each open class body sits
    depth = 0
    in_params = None       # indentation of the multi-line signature being erased
    in_block_comment = False
    while i < len(lines):
        ln = lines[i]
        st = ln.strip()
        # comments pass through untouched
        if in_block_comment:
            out.append(ln)
            if "*/" in ln:
                in_block_comment = False
            i += 1
            continue
        if st.startswith("/*") and "*/" not in st:
            in_block_comment = True
            out.append(ln)
            i += 1
            continue
        if st.startswith("//") or st.startswith("/*") or st.startswith("*"):
            out.append(ln)
            i += 1
            continue
        code, cmt = ln, ""
        ci = ln.find(" // ")
        if ci >= 0 and ln[:ci].count("'") % 2 == 0 and ln[:ci].count('"') % 2 == 0 and ln[:ci].count("`") % 2 == 0:
            code, cmt = ln[:ci], ln[ci:]
        # ---- whole statements that vanish ----
        if depth == 0 and re.match(r"^import\s+type\s", st):
            i += 1
            continue
        # ---- module syntax: the shipped files are CommonJS ----
        m = re.match(r"^import (\{[^}]*\}) from ('[^']+')$", st) if depth == 0 else None
        if m:
            out.append("const %s = require(%s)%s" % (m.group(1), m.group(2), cmt))
            i += 1
            continue
        m = re.match(r"^import \* as (\w+) from ('[^']+')$", st) if depth == 0 else None
        if m:
            out.append("const %s = require(%s)%s" % (m.group(1), m.group(2), cmt))
            i += 1
            continue
        m = re.match(r"^export (\{[^}]*\})$", st) if depth == 0 else None
        if m:
            out.append("module.exports = %s%s" % (m.group(1), cmt))
            i += 1
            continue
        if depth == 0 and re.match(r"^(export\s+)?(interface\s+\w+|type\s+\w+\s*(<[^>]*>)?\s*=|declare\s)", st):
            d = 0
            while True:
                d += lines[i].count("{") + lines[i].count("(") - lines[i].count("}") - lines[i].count(")")
                nxt = lines[i + 1].strip() if i + 1 < len(lines) else ""
                i += 1
                if d <= 0 and not nxt.startswith("|") and not lines[i - 1].rstrip().endswith(("|", "&", "=")):
                    break
            continue
        # ---- multi-line parameter lists ----
        if in_params is not None:
            if re.match(r"^\)\s*(:\s*.+?)?\s*\{$", st):
                out.append(" " * in_params + ") {" + cmt)
                depth += 1
                in_params = None
            else:
                had = code.rstrip().endswith(",")
                out.append(erase_param(code.rstrip().rstrip(",")) + ("," if had else "") + cmt)
            i += 1
            continue
        m = SIG_HEAD.match(code.rstrip())
        if m and (m.group(2).strip().split() or [""])[-1] not in KEYWORDS and m.group(2).strip():
            out.append(code.rstrip() + cmt)
            in_params = len(m.group(1))
            i += 1
            continue
        in_class = bool(depth_class) and depth == depth_class[-1] + 1
        # ---- class fields without an initialiser ----
        if in_class and re.match(r"^((private|public|protected|readonly|static|declare)\s+)*[A-Za-z_$][\w$]*[?!]?\s*:\s", st) and not st.endswith("{") and top_level_index(st, " = ") < 0:
            i += 1
            continue
        new = erase_signature_line(code)
        if new is not None:
            code = new
        else:
            # ---- local declarations: const x: T = e   /   let x: T ----
            m = re.match(r"^(\s*(?:export\s+)?(?:const|let|var)\s+[A-Za-z_$][\w$]*)\s*:\s*(.*)$", code)
            if m:
                eq = top_level_index(m.group(2), " = ")
                code = m.group(1) + (m.group(2)[eq:] if eq >= 0 else "")
        # ---- casts ----
        code = re.sub(r"\s+as\s+(?:unknown\s+as\s+)?[A-Za-z_$][\w$.]*(?:<[^<>()]*>)?(?:\[\])*(?=[\s)\],;.]|$)", "", code)
        out.append(code + cmt)
        # brace bookkeeping (strings with braces do not occur at statement level in these sources; template literals are single-line)
        stripped = re.sub(r"'(?:[^'\\]|\\.)*'|\"(?:[^\"\\]|\\.)*\"|`(?:[^`\\]|\\.)*`", "", code)
        if re.match(r"^\s*(export\s+)?(default\s+)?(abstract\s+)?class\s+\w+", stripped) and "{" in stripped:
            depth_class.append(depth)
        depth += stripped.count("{") - stripped.count("}")
        while depth_class and depth <= depth_class[-1]:
            depth_class.pop()
        i += 1
    return "\n".join(out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:
        sys.stdout.write(erase(open(args[0]).read()))
        return 0
    rc = 0
    names = sorted(f for f in os.listdir(SRC) if f.endswith(".ts") and not f.endswith(".d.ts"))
    for f in names:
        js = erase(open(os.path.join(SRC, f)).read())
        dst = os.path.join(DST, f[:-3] + ".js")
        if "--check" in sys.argv:
            have = open(dst).read() if os.path.exists(dst) else None
            if have != js:
                import difflib
                sys.stderr.write("%s is not what %s erases to — run python tools/ts_erase.py\n" % (os.path.relpath(dst, ROOT), os.path.relpath(os.path.join(SRC, f), ROOT)))
                sys.stderr.write("".join(list(difflib.unified_diff((have or "").splitlines(True), js.splitlines(True), "shipped", "erased"))[:40]))
                rc = 1
        else:
            open(dst, "w").write(js)
    if "--check" not in sys.argv:
        print("erased %d files into %s" % (len(names), os.path.relpath(DST, ROOT)))
    return rc


if __name__ == "__main__":
    sys.exit(main())
