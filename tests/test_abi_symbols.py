"""The C-ABI libraries load on a CPU-only machine and export every symbol their headers declare
(no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"GSWT_API\s+[\w\s\*]+?\b(gswt_\w+)\s*\(", src)))


@pytest.mark.parametrize("header,libname", [("gswt_hip.h", "libgswt_hip.so"), ("gswt_host.h", "libgswt_host.so")])
def test_library_exports_every_declared_symbol(header, libname):
    names = _declared(header)
    assert len(names) >= 15
    lib = C.CDLL(os.path.join(ROOT, "gswt_renderer_amd", "lib", libname))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_tables_match_headers():
    from gswt_renderer_amd import _lib, host
    assert sorted(_lib.SYMBOLS) == _declared("gswt_hip.h")
    assert sorted(host.HOST_SYMBOLS) == _declared("gswt_host.h")
    _lib.load()
    host.load()


def test_struct_layouts():
    from gswt_renderer_amd import _lib as L
    assert C.sizeof(L.CameraUniforms) == 176      # camera.rs:158-167
    assert C.sizeof(L.SceneUniforms) == 160       # renderer.rs:602-622
    assert C.sizeof(L.TileUniforms) == 80         # renderer.rs:675-689
    assert L.SceneUniforms.transition_dist_vec.offset == 64 and L.SceneUniforms.scene_scale.offset == 144
    assert L.TileUniforms.tile_id.offset == 32 and L.TileUniforms.offset.offset == 48
    assert L.CameraUniforms.focal.offset == 128 and L.CameraUniforms.cam_pos.offset == 160


def test_product_never_imports_oracle():
    """The product path must not route through oracle/ (only tests, smoke and bench's cpu_baseline may):
    no import, include, link or dlopen of anything under oracle/ from the product package."""
    pkg = os.path.join(ROOT, "gswt_renderer_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|from\s+\.+\s*oracle|#include\s*[\"<][^\">]*oracle|libgswt_oracle|gswt_oracle|orc_\w+\s*\()")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), (dirpath, f, pat.search(text).group(0))


# ---- Rust binding source (rust/) is generated from the headers and layout-checked ----------------------
def _rust_structs(text):
    """#[repr(C)] structs of a generated .rs file: name -> [(field, rust type, array len)]"""
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*\npub struct (\w+) \{\n(.*?)\n\}", text, flags=re.S):
        fields = []
        for line in m.group(2).splitlines():
            fm = re.match(r"\s*pub (\w+): (.*),$", line)
            t = fm.group(2)
            am = re.match(r"\[(.*); (\d+)\]$", t)
            fields.append((fm.group(1), am.group(1) if am else t, int(am.group(2)) if am else None))
        out[m.group(1)] = fields
    return out


def _rust_layout(fields, layouts):
    prim = {"f32": 4, "f64": 8, "c_int": 4, "c_uint": 4, "u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "u64": 8, "i64": 8, "usize": 8}
    off, align, offs = 0, 1, {}
    for name, t, n in fields:
        if t.startswith("*"):
            sz = al = 8
        elif t in prim:
            sz = al = prim[t]
        else:
            sz, al, _ = layouts[t]
        off = (off + al - 1) // al * al
        offs[name] = off
        off += sz * (n or 1)
        align = max(align, al)
    return (off + align - 1) // align * align, align, offs


def test_rust_bindings_are_generated_from_the_headers_and_match_their_layout(tmp_path):
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_bindings as g
    hip, host, hip_rs, host_rs = g.generate()
    # 1. committed files and the INTEGRATION.md blocks are what the generator produces from the headers today
    assert open(os.path.join(ROOT, "rust", "src", "gswt_hip_sys.rs")).read() == hip_rs
    assert open(os.path.join(ROOT, "rust", "src", "gswt_host_sys.rs")).read() == host_rs
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for tag, body in (("gswt_hip_sys", hip_rs), ("gswt_host_sys", host_rs)):
        blk = doc.split(f"<!-- BEGIN GENERATED {tag} -->")[1].split(f"<!-- END GENERATED {tag} -->")[0]
        assert blk.strip() == ("```rust\n" + body + "```").strip()
    # 2. every exported function is declared, with the C parameter names in order
    for hdr, rs, header_name in ((hip, hip_rs, "gswt_hip.h"), (host, host_rs, "gswt_host.h")):
        declared = _declared(header_name)
        in_rs = re.findall(r"pub fn (gswt_\w+)\(", rs)
        assert sorted(in_rs) == declared
        for ret, name, args in hdr.funcs:
            sig = re.search(r"pub fn %s\((.*?)\)( -> [^;]+)?;" % name, rs, flags=re.S).group(1)
            assert re.findall(r"(\w+): ", sig) == [a[1] for a in args], name
    # 3. struct layout: size and every field offset of the Rust #[repr(C)] text == gcc's layout of the C typedef
    structs = list(hip.structs) + list(host.structs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gswt_host.h"', 'int main(void) {']
    for hdr in (hip, host):
        for name, fields in hdr.structs.items():
            lines.append(f'  printf("S {name} %zu\\n", sizeof({name}));')
            for _, fname, _ in fields:
                lines.append(f'  printf("F {name} {fname} %zu\\n", offsetof({name}, {fname}));')
    lines.append("  return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    c_size, c_off = {}, {}
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        p = line.split()
        if p[0] == "S":
            c_size[p[1]] = int(p[2])
        else:
            c_off[(p[1], p[2])] = int(p[3])
    rs_structs = {**_rust_structs(hip_rs), **_rust_structs(host_rs)}
    layouts = {}
    for name in structs:
        rname = g.camel(name)
        assert rname in rs_structs, rname
        c_fields = (hip.structs.get(name) or host.structs.get(name))
        assert [f[0] for f in rs_structs[rname]] == [f[1] for f in c_fields], name        # names and order
        size, align, offs = _rust_layout(rs_structs[rname], layouts)
        layouts[rname] = (size, align, offs)
        assert size == c_size[name], (name, size, c_size[name])
        for fname, off in offs.items():
            assert off == c_off[(name, fname)], (name, fname)
    assert c_size["gswt_camera_uniforms"] == 176 and c_size["gswt_scene_uniforms"] == 160 and c_size["gswt_tile_uniforms"] == 80
    assert c_size["gswt_proxy_uniforms"] == 224
    # 4. the hand-written wrapper only calls functions the headers declare
    lib_rs = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    known = set(_declared("gswt_hip.h")) | set(_declared("gswt_host.h"))
    used = set(re.findall(r"\b(gswt_\w+)\s*\(", lib_rs))
    assert used and used <= known, used - known
