#!/usr/bin/env python3
"""Mutated PLY headers and truncated bodies through gswt_tileset_set_ply (scene.rs:72-212 parses untrusted bytes); run against the sanitizer
build like tools/fuzz_zip_loader.py (GSWT_HOST_LIB=... LD_PRELOAD=libasan libubsan)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gswt_renderer_amd import host, synth
if os.environ.get("GSWT_HOST_LIB"):
    host.HOST_LIB_PATH = os.environ["GSWT_HOST_LIB"]
rng = np.random.default_rng(3)
lib = host.load()
ts = host.TileSet.from_vertices([[np.zeros((1, 62), np.float32)]])
good = bytearray(synth.write_ply(rng.normal(size=(5, 62)).astype(np.float32)))
hdr = good.index(b"end_header\n") + 11
ok = rej = 0
for it in range(6000):
    b = bytearray(good)
    k = it % 4
    if k == 0:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, hdr))] = int(rng.integers(0, 256))
    elif k == 1:
        b = b[:int(rng.integers(0, len(b)))]
    elif k == 2:
        p = int(rng.integers(0, hdr)); b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
    else:
        p = int(rng.integers(0, hdr)); del b[p:p + int(rng.integers(1, 20))]
    buf = np.frombuffer(bytes(b) or b"\0", dtype=np.uint8)
    rc = lib.gswt_tileset_set_ply(ts._h, 0, 0, buf.ctypes.data, len(b))
    if rc == 0: ok += 1; ts.rows(0, 0)
    else: rej += 1
print(ok, "loaded", rej, "rejected, no crash")
