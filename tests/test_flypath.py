"""Fly-path playback of the benchmark harness (gswt_renderer_amd/flypath.py) against the closed form of
FlyPathControl::handle_events (control.rs:473-527): Catmull-Rom through the keyframes, end points extrapolated."""
import json

import numpy as np

from gswt_renderer_amd import flypath


def test_path_hits_its_keyframes_and_ends():
    keys = flypath.load("c3")
    assert len(keys) >= 2 and keys[0][0] == 0.0
    for t, pos, tgt in keys[:-1]:
        p, g = flypath.evaluate(keys, t)
        assert np.allclose(p, pos, atol=1e-6) and np.allclose(g, tgt, atol=1e-6)      # t = 0 on a segment returns p1
    assert flypath.evaluate(keys, keys[-1][0]) is None                                 # finished (control.rs:480-484)
    assert flypath.evaluate(keys, keys[-1][0] + 1.0) is None


def test_catmull_rom_closed_form_and_extrapolated_ends(tmp_path):
    frames = [dict(timestamp=float(k), position_x=float(k * k), position_y=2.0 * k, position_z=1.0, target_x=0.0, target_y=float(k), target_z=3.0)
              for k in range(4)]
    path = tmp_path / "p.json"
    path.write_text(json.dumps(frames))
    keys = flypath.load(str(path))
    P = [np.array([k * k, 2.0 * k, 1.0], dtype=np.float64) for k in range(4)]

    def cr(p0, p1, p2, p3, t):
        return 0.5 * (2 * p1 + (-p0 + p2) * t + (2 * p0 - 5 * p1 + 4 * p2 - p3) * t * t + (-p0 + 3 * p1 - 3 * p2 + p3) * t ** 3)
    # first segment: p0 = 2 p1 - p2 (control.rs:489-494); middle: real neighbours; last: p3 = 2 p2 - p1 (:498-503)
    for seg, (p0, p1, p2, p3) in enumerate([(2 * P[0] - P[1], P[0], P[1], P[2]), (P[0], P[1], P[2], P[3]), (P[1], P[2], P[3], 2 * P[3] - P[2])]):
        for t in (0.25, 0.5, 0.9):
            pos, tgt = flypath.evaluate(keys, seg + t)
            assert np.allclose(pos, cr(p0, p1, p2, p3, t), atol=1e-5)
            assert abs(tgt[1] - (seg + t)) < 1e-5                                       # a linear track stays linear under Catmull-Rom
    cams = flypath.sample(keys, 30)
    assert len(cams) == 30 and all(c is not None for c in cams)
    assert np.allclose(cams[0][0], P[0])
