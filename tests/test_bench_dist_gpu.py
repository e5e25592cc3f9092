"""bench.py's N > 1 code on the one-GPU box: the RCCL path with one rank (GSWT_BENCH_FORCE_DIST=1:
gswt_comm_init + gswt_render_gather behind the C ABI) and rank 0 of a two-rank run without its peer (GSWT_BENCH_FAKE_WORLD=2:
lock-step worker, deferred swap-ins switched at agreed frames, column bands, unshard).  Each runs bench.py as its own
process (it redirects stdout and owns a process group) and checks the one JSON line."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "24", "--warmup", "8", "--no-cpu-baseline",
           "--static-steps", "12", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def _check_line(d, n_gpus):
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == 24 and d["warmup"] == 8
    assert math.isfinite(d["value"]) and d["value"] > 0
    assert d["config"]["workload"].startswith("tiny")
    assert "roofline" in d and d["roofline"]["bound"] == "hbm"
    assert "collective" in d
    assert d["dist_check_max_abs_diff"] == 0.0          # gathered frame == the same camera rendered unsharded, bit for bit


def test_bench_rccl_path_with_one_rank():
    d, err = _run_bench({"GSWT_BENCH_FORCE_DIST": "1"})
    _check_line(d, 1)
    assert "gswt_render_gather" in d["collective"], (d["collective"], err[-1500:])


def test_bench_rank_zero_of_a_fake_world_of_two():
    d, err = _run_bench({"GSWT_BENCH_FAKE_WORLD": "2"})          # (--gpus 2 itself insists on torch.distributed.run: the hook stands in for the launcher)
    _check_line(d, 2)
    assert d["scaling"] == "strong"
    assert d["sort_events"]["swapped_in"] >= 1, d["sort_events"]     # the lock-step worker's results were swapped in
