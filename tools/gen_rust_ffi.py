#!/usr/bin/env python3
"""Generate bindings/rust/fyrox_hip_sys.rs -- the Rust `extern "C"` declarations of include/fyrox_hip.h (what a
Fyrox maintainer drops into `fyrox-impl/src/scene/mesh/hip_sys.rs`; INTEGRATION.md shows the safe wrappers on top).
The Rust toolchain is not part of this repository's build image, so the file is generated mechanically from the C
header (one declaration per prototype / struct / enum constant, types mapped 1:1) and kept in step by
tests/test_abi.py::test_rust_ffi_matches_header; it is not compiled here.

  python tools/gen_rust_ffi.py            # rewrite bindings/rust/fyrox_hip_sys.rs
  python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fyrox_hip.h")
OUT = os.path.join(ROOT, "bindings", "rust", "fyrox_hip_sys.rs")

SCALARS = {"int": "c_int", "float": "f32", "double": "f64", "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64",
           "int32_t": "i32", "int64_t": "i64", "size_t": "usize", "char": "c_char", "void": "c_void"}


def strip_comments(src: str) -> str:
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_name(c_struct: str) -> str:
    return "".join(p.capitalize() for p in c_struct.split("_"))          # fyx_skin_desc -> FyxSkinDesc


def map_type(ctype: str, structs) -> str:
    """C declarator -> Rust: the qualifiers are read level by level (`const T*` / `T const*` qualify the pointee,
    `T* const*` the inner pointer): `const float*` -> *const f32, `fyx_ctx**` -> *mut *mut FyxCtx,
    `fyx_ctx* const*` -> *const *mut FyxCtx, `float* const*` -> *const *mut f32."""
    levels = [lv.split() for lv in ctype.strip().split("*")]       # [base tokens][qualifiers after the 1st *][after the 2nd] ...
    base_tokens = [x for x in levels[0] if x not in ("const", "struct")]
    consts = ["const" in lv for lv in levels]                      # consts[i]: what the (i+1)-th pointer points at is const
    base = " ".join(base_tokens)
    if base in SCALARS:
        r = SCALARS[base]
    elif base in structs or base == "fyx_ctx":
        r = rust_name(base)
    else:
        raise ValueError(f"unmapped C type {ctype!r}")
    for i in range(len(levels) - 1):
        r = ("*const " if consts[i] else "*mut ") + r
    return r


def parse(src: str):
    src = strip_comments(src)
    defines = re.findall(r"^#define\s+(FYX_[A-Z0-9_]+)\s+([^\n]+)$", src, flags=re.M)
    enums = []
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}\s*\w*\s*;", src, flags=re.S):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, val = [x.strip() for x in item.split("=")]
                nxt = int(val, 0)
            else:
                name = item
            enums.append((name, nxt))
            nxt += 1
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(.+?)\s*((?:\**\w+(?:\[\d+\])?\s*,\s*)*\**\w+(?:\[\d+\])?)$", decl)
            ctype, names = m.group(1), m.group(2)
            for n in names.split(","):
                n = n.strip()
                stars = n.count("*")
                n = n.replace("*", "")
                arr = re.match(r"(\w+)\[(\d+)\]", n)
                fields.append((arr.group(1) if arr else n, ctype + "*" * stars, int(arr.group(2)) if arr else 0))
        structs[name] = fields
    protos = []
    # drop struct / enum bodies and preprocessor lines, then every `;`-terminated chunk that ends in `fyx_x(...)` is a prototype
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", ";", src, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\}\s*;", ";", body, flags=re.S)
    body = re.sub(r"^\s*#.*$", "", body, flags=re.M)
    body = re.sub(r'extern\s+"C"\s*\{', ";", body)
    for chunk in re.sub(r"\s+", " ", body).split(";"):
        m = re.match(r"^ ?(.*?) ?\b(fyx_[a-z0-9_]+) ?\(([^()]*)\) ?$", chunk)
        if not m:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                arr = re.match(r"(.+?)\s*(\w+)\[(\w*)\]$", a)          # float out_aabb[6] / id[FYX_N] -> pointer
                if arr:
                    params.append((arr.group(2), arr.group(1) + "*"))
                    continue
                mm = re.match(r"(.+?[\s\*])(\w+)$", a)
                params.append((mm.group(2), mm.group(1)))
        protos.append((name, ret, params))
    return defines, enums, structs, protos


def generate() -> str:
    defines, enums, structs, protos = parse(open(HEADER).read())
    o = ["// @generated by tools/gen_rust_ffi.py from include/fyrox_hip.h -- do not edit.",
         "// Raw FFI of libfyrox_hip.so.  All functions return a fyx_status (0 = FYX_OK, < 0 = error; the message is",
         "// fyx_last_error(ctx)); nothing unwinds across the boundary.  One context per engine thread (!Send, !Sync).",
         "#![allow(non_camel_case_types, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_void};", "",
         "#[repr(C)] pub struct FyxCtx { _private: [u8; 0] }", ""]
    for name, val in defines:
        val = val.strip()
        if re.fullmatch(r"0x[0-9a-fA-F]+u|\d+u", val):
            o.append(f"pub const {name}: u32 = {val[:-1]};")
        elif re.fullmatch(r"-?\d+", val):
            o.append(f"pub const {name}: i32 = {val};")
    o.append("")
    for name, val in enums:
        o.append(f"pub const {name}: i32 = {val};")
    o.append("")
    for sname, fields in structs.items():
        o.append("#[repr(C)] #[derive(Clone, Copy)]")
        o.append(f"pub struct {rust_name(sname)} {{")
        for fname, ctype, arr in fields:
            rt = map_type(ctype, structs)
            o.append(f"    pub {fname}: {('[%s; %d]' % (rt, arr)) if arr else rt},")
        o.append("}")
        o.append("")
    o.append('#[link(name = "fyrox_hip")]')
    o.append('extern "C" {')
    for name, ret, params in protos:
        ps = ", ".join(f"{('r#' + n) if n in ('type', 'ref', 'in', 'loop', 'match', 'move') else n}: {map_type(t, structs)}" for n, t in params)
        r = ret.strip()
        rr = "" if r == "void" else f" -> {map_type(r, structs)}"
        o.append(f"    pub fn {name}({ps}){rr};")
    o.append("}")
    o.append("")
    return "\n".join(o)


def main():
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions, {text.count('pub struct ')} structs")


if __name__ == "__main__":
    main()
