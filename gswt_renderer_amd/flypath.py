"""Fly-path playback for the benchmark harness: the reference's only benchmark mechanism (gui.rs:964-997 plays a path and
prints mean +- std of the frame / sort / update times).

Keyframes use the reference's JSON schema (control.rs:383-392: timestamp, position_x/y/z, target_x/y/z; up is always +z,
control.rs:313); interpolation is FlyPathControl::handle_events' Catmull-Rom with the end points extrapolated
(control.rs:487-527).  The reference plays a path against the wall clock; the benchmark samples it at `n` evenly spaced
times so that every run renders the same cameras.  Harness only: camera control is outside the product's scope.
"""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(name_or_path: str):
    """-> list of (timestamp, position[3], target[3]); `c3` etc. name the paths shipped in flypaths/."""
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(_HERE, "flypaths", name_or_path + ".json")
    frames = json.load(open(path))
    return [(float(f["timestamp"]), np.array([f["position_x"], f["position_y"], f["position_z"]], dtype=np.float32),
             np.array([f["target_x"], f["target_y"], f["target_z"]], dtype=np.float32)) for f in frames]


def _catmull_rom(p0, p1, p2, p3, t):
    t2, t3 = t * t, t * t * t
    return 0.5 * ((2.0 * p1) + (-p0 + p2) * t + (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3) * t2 + (-p0 + 3.0 * p1 - 3.0 * p2 + p3) * t3)


def evaluate(keyframes, ela_time: float):
    """Camera (position, target) at `ela_time` seconds, control.rs:473-527; None once the path has finished."""
    if len(keyframes) < 2 or ela_time >= keyframes[-1][0]:
        return None
    fi = 0
    while ela_time >= keyframes[fi + 1][0]:
        fi += 1
    t = np.float32((ela_time - keyframes[fi][0]) / (keyframes[fi + 1][0] - keyframes[fi][0]))
    out = []
    for k in (1, 2):                      # position, then target
        p1, p2 = keyframes[fi][k], keyframes[fi + 1][k]
        p0 = keyframes[0][k] * 2.0 - keyframes[1][k] if fi == 0 else keyframes[fi - 1][k]
        p3 = keyframes[fi + 1][k] * 2.0 - keyframes[fi][k] if fi + 2 >= len(keyframes) else keyframes[fi + 2][k]
        out.append(_catmull_rom(p0, p1, p2, p3, t).astype(np.float32))
    return out[0], out[1]


def sample(keyframes, n: int):
    """`n` cameras at evenly spaced times over the path's duration (the last one just before the end)."""
    t0, t1 = keyframes[0][0], keyframes[-1][0]
    return [evaluate(keyframes, t0 + (t1 - t0) * k / n) for k in range(n)]
