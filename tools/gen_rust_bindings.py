#!/usr/bin/env python3
"""Generates the Rust FFI declarations of the two C-ABI libraries from their headers.

    include/gswt_hip.h   -> rust/src/gswt_hip_sys.rs    (libgswt_hip.so: the GPU half of the hot path)
    include/gswt_host.h  -> rust/src/gswt_host_sys.rs   (libgswt_host.so: scene loader + WangTile worker)

and rewrites the two fenced blocks of INTEGRATION.md between the markers
`<!-- BEGIN GENERATED gswt_hip_sys -->` / `<!-- END GENERATED gswt_hip_sys -->` (and `gswt_host_sys`).
The headers are the single source of truth: tests/test_abi_symbols.py re-runs this generator and fails when the
committed Rust files or the INTEGRATION.md blocks are stale, and checks every `#[repr(C)]` struct's size and field offsets
against what gcc computes for the C declaration.

There is no Rust toolchain in the build image, so the output is source only (reviewed by eye, layout-checked by the test).
    python tools/gen_rust_bindings.py            # rewrite the files
    python tools/gen_rust_bindings.py --check    # exit 1 when something would change
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCALARS = {
    "float": "f32", "double": "f64", "int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "char": "c_char",
    "uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16", "int16_t": "i16", "uint32_t": "u32", "int32_t": "i32",
    "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "unsigned long long": "u64", "long long": "i64", "void": "c_void",
}
SIZES = {"f32": 4, "f64": 8, "c_int": 4, "c_uint": 4, "c_char": 1, "u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4,
         "u64": 8, "i64": 8, "usize": 8}


def camel(name: str) -> str:
    """gswt_render_config -> GswtRenderConfig"""
    return "".join(p.capitalize() for p in name.split("_"))


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def rust_type(ctype: str, known: set[str]) -> str:
    """C declarator type (no name) -> Rust.  Handles const / pointer chains such as `const int32_t *const **`."""
    toks = re.findall(r"\*|const|[A-Za-z_][A-Za-z_0-9 ]*?(?=\s*(?:\*|const\b|$))", ctype.strip())
    toks = [t.strip() for t in toks if t.strip()]
    # base type = tokens before the first '*', minus const
    base_const = False
    i = 0
    base_words = []
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            base_words.append(toks[i])
        i += 1
    base = " ".join(base_words)
    if base in SCALARS:
        rt = SCALARS[base]
    elif base in known:
        rt = camel(base)
    else:
        raise ValueError(f"unknown C type '{base}' in '{ctype}'")
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*"
        rt = ("*const " if pointee_const else "*mut ") + rt
        i += 1
        pointee_const = False
        while i < len(toks) and toks[i] == "const":
            pointee_const = True
            i += 1
    return rt


def split_decl(decl: str):
    """`const uint32_t *a, *b` / `float x[3]` / `uint32_t lod, tile` -> [(ctype, name, array_len | None)]"""
    decl = " ".join(decl.split())
    first, *rest = [d.strip() for d in decl.split(",")]
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)\s*(\[\s*(\d+)\s*\])?$", first)
    lead = m.group(1).strip()
    stars = ""
    while lead.endswith("*"):
        stars = "*" + stars
        lead = lead[:-1].strip()
    # `const T *const` forms keep their consts in `lead`
    out = [((lead + " " + stars).strip(), m.group(2), int(m.group(4)) if m.group(4) else None)]
    for r in rest:
        m2 = re.match(r"^(\**)\s*([A-Za-z_][A-Za-z_0-9]*)\s*(\[\s*(\d+)\s*\])?$", r)
        out.append(((lead + " " + m2.group(1)).strip(), m2.group(2), int(m2.group(4)) if m2.group(4) else None))
    return out


class Header:
    def __init__(self, path: str, known_structs: dict | None = None):
        self.path = path
        self.raw = open(path).read()
        src = strip_comments(self.raw)
        self.opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)
        self.structs = {}      # name -> [(ctype, field, array_len)]
        for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
            fields = []
            for decl in m.group(1).split(";"):
                if decl.strip():
                    fields.extend(split_decl(decl))
            self.structs[m.group(2)] = fields
        self.consts = []       # (name, value)
        for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", src, flags=re.S):
            for item in m.group(1).split(","):
                if "=" in item:
                    k, v = item.split("=")
                    self.consts.append((k.strip(), int(v.strip())))
        for m in re.finditer(r"#define\s+(GSWT_\w+)\s+\(?\s*(-?\d+)\s*\)?\s*$", src, flags=re.M):
            self.consts.append((m.group(1), int(m.group(2))))
        self.funcs = []        # (ret ctype, name, [(ctype, argname, array_len)])
        for m in re.finditer(r"GSWT_API\s+([\w\s\*]+?)\b(gswt_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
            ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
            alist = []
            if args and args != "void":
                for a in args.split(","):
                    alist.extend(split_decl(a))
            self.funcs.append((ret, name, alist))
        self.known = set(self.opaque) | set(self.structs) | set(known_structs or ())


def struct_layout(fields, known_layouts, known_names):
    """repr(C) layout of a parsed struct: (size, align, [(name, offset)])."""
    off, align, offs = 0, 1, []
    for ctype, name, n in fields:
        rt = rust_type(ctype, known_names)
        if rt.startswith("*"):
            sz = al = 8
        elif rt in SIZES:
            sz = al = SIZES[rt]
        else:
            sz, al, _ = known_layouts[rt]
        off = (off + al - 1) // al * al
        offs.append((name, off))
        off += sz * (n or 1)
        align = max(align, al)
    return (off + align - 1) // align * align, align, offs


def emit(h: Header, lib: str, uses: str, cites: dict[str, str]) -> str:
    out = [f"// GENERATED by tools/gen_rust_bindings.py from include/{os.path.basename(h.path)} -- do not edit.",
           f"// FFI declarations of lib{lib}.so.  Struct layouts are #[repr(C)] mirrors of the header, field for field.",
           "#![allow(non_camel_case_types, dead_code)]", uses, ""]
    for name, val in h.consts:
        out.append(f"pub const {name}: c_int = {val};")
    out.append("")
    for name in h.opaque:
        out.append(f"#[repr(C)] pub struct {camel(name)} {{ _private: [u8; 0] }}")
    out.append("")
    for name, fields in h.structs.items():
        if name in cites:
            out.append(f"/// {cites[name]}")
        out.append("#[repr(C)] #[derive(Clone, Copy)]")
        out.append(f"pub struct {camel(name)} {{")
        for ctype, fname, n in fields:
            rt = rust_type(ctype, h.known)
            out.append(f"    pub {fname}: {'[%s; %d]' % (rt, n) if n else rt},")
        out.append("}")
        out.append("")
    out.append(f'#[link(name = "{lib}")]')
    out.append('extern "C" {')
    for ret, name, args in h.funcs:
        ra = []
        for ctype, aname, n in args:
            rt = rust_type(ctype, h.known)
            if n:                       # `const float pos[3]` decays to a pointer
                rt = ("*const " if "const" in ctype.split() else "*mut ") + rt
            ra.append(f"{aname}: {rt}")
        rret = "" if ret == "void" else f" -> {rust_type(ret, h.known)}"
        line = f"    pub fn {name}({', '.join(ra)}){rret};"
        if len(line) > 128:
            line = f"    pub fn {name}(\n        " + ",\n        ".join(ra) + f",\n    ){rret};"
        out.append(line)
    out.append("}")
    return "\n".join(out) + "\n"


HIP_CITES = {
    "gswt_camera_uniforms": "camera::CameraUniforms (camera.rs:158-167), 176 B",
    "gswt_scene_uniforms": "renderer::SceneUniforms (renderer.rs:602-622), 160 B",
    "gswt_tile_uniforms": "renderer::TileUniforms (renderer.rs:675-689), 80 B",
    "gswt_base_list": "one PreloadData.tile_base_data[lod][tile][view] (structure.rs:546-554)",
    "gswt_draw": "one iteration of the draw loop of GSWTRenderer::render (renderer.rs:466-591)",
    "gswt_render_config": "the RenderConfig fields read on the hot path (structure.rs:346-388) + shard selection",
    "gswt_proxy_uniforms": "proxy::Uniforms (proxy.rs:470-511), 224 B",
    "gswt_merge_group": "one MergedFrom tile: view + members (wangtile.rs:595-670)",
    "gswt_merge_member": "one member of a merged group",
    "gswt_timings": "per-stage device times and workload sizes of the last frame",
}
HOST_CITES = {
    "gswt_user_data": "structure::UserData (structure.rs:15-65), the fields the worker reads",
    "gswt_configured": "what WangTile::configure adds (wangtile.rs:349-432)",
    "gswt_scene_data": "structure::SceneData (structure.rs:466-474)",
    "gswt_sorted_tile": "one element of SortData.tile_instance_vec + render_data_vec (structure.rs:488-509,670-694)",
    "gswt_sort_data": "structure::SortData (structure.rs:489-494)",
    "gswt_preload": "structure::PreloadData (structure.rs:731-736)",
}


def generate():
    hip = Header(os.path.join(ROOT, "include", "gswt_hip.h"))
    host = Header(os.path.join(ROOT, "include", "gswt_host.h"), known_structs=hip.known)
    hip_rs = emit(hip, "gswt_hip", "use std::os::raw::{c_char, c_int, c_uint, c_void};", HIP_CITES)
    host_rs = emit(host, "gswt_host", "use std::os::raw::{c_char, c_int, c_uint, c_void};\nuse crate::gswt_hip_sys::*;", HOST_CITES)
    return hip, host, hip_rs, host_rs


def splice(doc: str, tag: str, body: str) -> str:
    a, b = f"<!-- BEGIN GENERATED {tag} -->", f"<!-- END GENERATED {tag} -->"
    i, j = doc.index(a) + len(a), doc.index(b)
    return doc[:i] + "\n```rust\n" + body + "```\n" + doc[j:]


def main():
    check = "--check" in sys.argv
    _, _, hip_rs, host_rs = generate()
    targets = {os.path.join(ROOT, "rust", "src", "gswt_hip_sys.rs"): hip_rs, os.path.join(ROOT, "rust", "src", "gswt_host_sys.rs"): host_rs}
    ipath = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(ipath).read()
    targets[ipath] = splice(splice(doc, "gswt_hip_sys", hip_rs), "gswt_host_sys", host_rs)
    stale = []
    for path, text in targets.items():
        old = open(path).read() if os.path.exists(path) else None
        if old != text:
            stale.append(path)
            if not check:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                open(path, "w").write(text)
    if stale:
        print(("stale: " if check else "wrote: ") + ", ".join(os.path.relpath(p, ROOT) for p in stale))
        return 1 if check else 0
    print("up to date")
    return 0


if __name__ == "__main__":
    sys.exit(main())
