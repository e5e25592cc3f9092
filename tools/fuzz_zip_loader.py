#!/usr/bin/env python3
"""Mutated tile zips through gswt_load_scene_zip_mem (the loader parses untrusted bytes: scene.rs:1030-1141 panics where this one must return a
status).  Meant to run against a sanitizer build of libgswt_host.so:
  GSWT_HOST_LIB=/tmp/sanlib/libgswt_host.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 python tools/fuzz_zip_loader.py [iterations] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gswt_renderer_amd import host, synth

if os.environ.get("GSWT_HOST_LIB"):
    host.HOST_LIB_PATH = os.environ["GSWT_HOST_LIB"]
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
verts = [[rng.normal(size=(int(rng.integers(1, 6)), 62)).astype(np.float32) for _ in range(4)] for _ in range(2)]
good = bytearray(synth.tile_zip_bytes(verts))
ok = rejected = 0
for it in range(n_iter):
    b = bytearray(good)
    kind = it % 4
    if kind == 0:                                   # a few random bytes anywhere
        for _ in range(int(rng.integers(1, 8))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif kind == 1:                                 # truncation
        b = b[:int(rng.integers(0, len(b)))]
    elif kind == 2:                                 # bytes in the last 200 (central directory / end record: counts, offsets, sizes)
        for _ in range(int(rng.integers(1, 6))):
            b[len(b) - 1 - int(rng.integers(0, min(200, len(b))))] = int(rng.integers(0, 256))
    else:                                           # a 32-bit field somewhere set to an extreme
        p = int(rng.integers(0, len(b) - 4))
        b[p:p + 4] = int(rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, 0xFFFFFFFE, len(b), len(b) + 1])).to_bytes(4, "little")
    try:
        ts = host.TileSet.from_zip(bytes(b))
        l, t = ts.dims()
        for i in range(l):
            for j in range(t):
                ts.rows(i, j)
        ts.close()
        ok += 1
    except host.GSWTHostError:
        rejected += 1
print(f"{n_iter} mutated zips: {ok} loaded, {rejected} rejected with a status, no crash")
