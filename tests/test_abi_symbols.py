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
