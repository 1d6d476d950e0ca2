#!/usr/bin/env python3
"""Generate the Rust side of the C ABI from include/pbrtgpu.h.

    tools/gen_rust_ffi.py            print the Rust module (src/gpu/ffi.rs of the reference tree)
    tools/gen_rust_ffi.py --update   rewrite the generated block of INTEGRATION.md in place
    tools/gen_rust_ffi.py --layout   print "struct size align" and "struct.field offset size" lines (what the C compiler must agree with)

The header is the single source: every `#define`, enum, struct and prototype in it comes out as the `#[repr(C)]` /
`extern "C"` declaration a pbrt-r3 maintainer binds (src/core/api/parse_context.rs:5-66 and
src/core/integrator/integrator.rs:6-9 are the seams the binding sits behind).  tests/test_host.py regenerates the block and
compares it with INTEGRATION.md, and compiles a C program that prints sizeof / offsetof of every struct to compare with the
layout computed here -- so a new field or ABI bump that forgets the document fails the CPU suite.
This image has no rustc: the output is checked for layout against the C compiler, not compiled.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pbrtgpu.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED FFI (tools/gen_rust_ffi.py) -->", "<!-- END GENERATED FFI -->"

SCALARS = {  # C type -> (Rust type, size, align)
    "int32_t": ("i32", 4, 4), "uint32_t": ("u32", 4, 4), "uint64_t": ("u64", 8, 8), "int64_t": ("i64", 8, 8),
    "uint16_t": ("u16", 2, 2), "uint8_t": ("u8", 1, 1), "float": ("f32", 4, 4), "double": ("f64", 8, 8),
    "int": ("c_int", 4, 4), "size_t": ("usize", 8, 8), "char": ("c_char", 1, 1), "void": ("c_void", 0, 1),
}
RUST_KEYWORDS = {"type", "where", "in", "ref", "move", "match", "loop", "fn", "impl", "use", "mod", "self", "box"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_name(name):
    n = name.lower()
    return n + "_" if n in RUST_KEYWORDS else n


class Header:
    def __init__(self, path=HEADER):
        raw = open(path).read()
        text = strip_comments(raw)
        self.defines = []      # (name, value text)
        for m in re.finditer(r"^#define\s+(PT_[A-Z0-9_]+)\s+(.+?)\s*$", text, flags=re.M):
            self.defines.append((m.group(1), m.group(2).strip()))
        self.enums = []        # (name, [(constant, value)])
        self.enum_names = set()
        for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            items, nxt = [], 0
            for part in m.group(1).split(","):
                part = part.strip()
                if not part:
                    continue
                if "=" in part:
                    k, v = [x.strip() for x in part.split("=")]
                    nxt = int(v, 0)
                else:
                    k = part
                items.append((k, nxt))
                nxt += 1
            self.enums.append((m.group(2), items))
            self.enum_names.add(m.group(2))
        self.opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text)
        self.structs = []      # (name, [(field, ctype, is_ptr, is_const, dims)])
        for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for decl in m.group(1).split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                mm = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
                const, ctype, star, rest = bool(mm.group(1)), mm.group(2), mm.group(3), mm.group(4)
                for d in rest.split(","):
                    d = d.strip()
                    ptr = bool(star)
                    if d.startswith("*"):
                        ptr, d = True, d[1:].strip()
                    dm = re.match(r"(\w+)((?:\[\d+\])*)$", d)
                    dims = [int(x) for x in re.findall(r"\[(\d+)\]", dm.group(2))]
                    fields.append((dm.group(1), ctype, ptr, const, dims))
            self.structs.append((m.group(2), fields))
        self.struct_names = {n for n, _ in self.structs}
        self.functions = []    # (name, return ctype text, [(ctype text, name)])
        body = re.sub(r"typedef\s+(enum|struct)\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
        for m in re.finditer(r"^\s*((?:const\s+)?\w+\s*\**)\s*(pt_\w+)\s*\(([^;{]*?)\)\s*;", body, flags=re.M | re.S):
            ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
            params = []
            if args and args != "void":
                for a in args.split(","):
                    a = a.strip()
                    am = re.match(r"(.*?)(\w+)$", a)
                    params.append((am.group(1).strip(), am.group(2)))
            self.functions.append((name, ret, params))

    # ---- layout, as the System V x86-64 / AArch64 C ABI lays plain structs out
    def type_layout(self, ctype, ptr, dims):
        if ptr:
            size, align = 8, 8
        elif ctype in SCALARS:
            _, size, align = SCALARS[ctype]
        elif ctype in self.enum_names:
            size, align = 4, 4
        elif ctype in self.struct_names:
            size, align, _ = self.struct_layout(ctype)
        else:
            raise ValueError("unknown type " + ctype)
        for d in dims:
            size *= d
        return size, align

    def struct_layout(self, name):
        fields = dict(self.structs)[name]
        off, amax, out = 0, 1, []
        for fname, ctype, ptr, const, dims in fields:
            size, align = self.type_layout(ctype, ptr, dims)
            off = (off + align - 1) // align * align
            out.append((fname, off, size))
            off += size
            amax = max(amax, align)
        return (off + amax - 1) // amax * amax, amax, out

    # ---- Rust text
    def rust_type(self, ctype, ptr, const, dims):
        if ctype in SCALARS:
            t = SCALARS[ctype][0]
        elif ctype in self.enum_names:
            t = "c_int"           # C enums cross the boundary as int
        else:
            t = ctype
        if ptr:
            t = ("*const " if const else "*mut ") + t
        for d in reversed(dims):
            t = "[%s; %d]" % (t, d)
        return t

    def rust_param_type(self, text):
        const = "const" in text.split()
        stars = text.count("*")
        base = text.replace("const", "").replace("*", "").strip()
        t = self.rust_type(base, False, False, [])
        for i in range(stars):
            # the innermost level carries the const; an out-pointer to a pointer is mutable
            t = ("*const " if (const and i == 0) else "*mut ") + t
        return t

    def rust_module(self):
        L = []
        L.append("// GENERATED by tools/gen_rust_ffi.py from include/pbrtgpu.h (ABI %s) -- do not edit; uncompiled (no rustc in the build image),"
                 % dict(self.defines)["PT_ABI_VERSION"])
        L.append("// struct sizes and field offsets are checked against the C compiler by tests/test_host.py::test_rust_ffi_block_matches_header")
        L.append("#![allow(non_camel_case_types, non_upper_case_globals)]")
        L.append("use std::os::raw::{c_char, c_int, c_void};")
        L.append("")
        for name, val in self.defines:
            v = val
            if re.fullmatch(r"\(?-?[\d.]+f\)?", v):
                L.append("pub const %s: f32 = %s;" % (name, v.strip("()").rstrip("f")))
            elif re.fullmatch(r"\d+u", v):
                L.append("pub const %s: u32 = %s;" % (name, v[:-1]))
            elif re.fullmatch(r"\d+", v):
                L.append("pub const %s: c_int = %s;" % (name, v))
        L.append("")
        for name, items in self.enums:
            L.append("// %s" % name)
            line = ""
            for kv in items:
                item = "pub const %s: c_int = %d;" % kv
                if line and len(line) + 1 + len(item) > 128:
                    L.append(line)
                    line = ""
                line += (" " if line else "") + item
            L.append(line)
        L.append("")
        for o in self.opaque:
            L.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % o)
        L.append("")
        for name, fields in self.structs:
            size, align, _ = self.struct_layout(name)
            has_ptr = any(f[2] for f in fields)
            L.append("#[repr(C)]%s" % ("" if has_ptr or name == "pt_scene_desc" else " #[derive(Clone, Copy)]"))
            L.append("pub struct %s {      // %d bytes, align %d" % (name, size, align))
            line = "   "
            for fname, ctype, ptr, const, dims in fields:
                item = " pub %s: %s," % (rust_name(fname), self.rust_type(ctype, ptr, const, dims))
                if len(line) + len(item) > 128:
                    L.append(line)
                    line = "   "
                line += item
            L.append(line)
            L.append("}")
        L.append("")
        L.append('#[link(name = "pbrtgpu")]')
        L.append('extern "C" {')
        for name, ret, params in self.functions:
            ps = ", ".join("%s: %s" % (rust_name(pn), self.rust_param_type(pt)) for pt, pn in params)
            if ret == "void":
                r = ""
            elif ret == "pt_status":
                r = " -> c_int"      # a pt_status value
            else:
                r = " -> " + self.rust_param_type(ret)
            L.append("    pub fn %s(%s)%s;" % (name, ps, r))
        L.append("}")
        return "\n".join(L) + "\n"

    def layout_lines(self):
        out = []
        for name, _ in self.structs:
            size, align, fields = self.struct_layout(name)
            out.append("%s %d %d" % (name, size, align))
            for fname, off, fsize in fields:
                out.append("%s.%s %d %d" % (name, fname, off, fsize))
        return out

    def layout_c_program(self):
        """A C program that prints the same lines from the real header."""
        L = ['#include <stdio.h>', '#include <stddef.h>', '#include "pbrtgpu.h"', "int main(void) {"]
        for name, fields in self.structs:
            L.append('    printf("%s %%zu %%zu\\n", sizeof(%s), _Alignof(%s));' % (name, name, name))
            for fname, ctype, ptr, const, dims in fields:
                L.append('    printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (name, fname, name, fname, name, fname))
        L += ["    return 0;", "}"]
        return "\n".join(L) + "\n"


def generated_block(h=None):
    h = h or Header()
    return BEGIN + "\n```rust\n" + h.rust_module() + "```\n" + END


def main():
    h = Header()
    if "--layout" in sys.argv:
        print("\n".join(h.layout_lines()))
    elif "--c-program" in sys.argv:
        sys.stdout.write(h.layout_c_program())
    elif "--update" in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN), doc.index(END) + len(END)
        open(DOC, "w").write(doc[:a] + generated_block(h) + doc[b:])
    else:
        sys.stdout.write(h.rust_module())


if __name__ == "__main__":
    main()
