"""bench.py's command line without a GPU: the options the driver passes exist, and the documented ones parse."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_lists_the_contract_options():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    for opt in ("--gpus", "--steps", "--warmup", "--workload", "--mode", "--graph", "--no-graph", "--segment", "--in-flight", "--device-worker",
                "--timing-every", "--static-steps", "--no-cpu-baseline"):
        assert opt in r.stdout, opt


def test_run_scripts_reference_existing_tools():
    for script in ("tools/profile_c3.sh", "tools/run_bench_lines.sh", "tools/run_round_check.sh"):
        text = open(os.path.join(ROOT, script)).read()
        for tok in text.replace("$", " ").split():
            if tok.startswith("tools/") and tok.endswith(".py"):
                assert os.path.exists(os.path.join(ROOT, tok)), (script, tok)


def test_auto_segment_fills_the_chip_and_keeps_dense_frames_long():
    """bench.py's rule for GSWT_OPT_SEGMENT (pairs per compositor work item): the dense-frame length (4 x pairs per screen tile, at least the
    library's default) unless that leaves fewer than ~2 048 work items -- one rank's band of a sharded frame."""
    sys.path.insert(0, ROOT)
    import bench
    from gswt_renderer_amd import _lib as L
    assert L.GSWT_DEFAULT_SEGMENT == 1536
    cases = {  # name: (pairs, screen tiles) -> segment
        "c3": (2.66e6, 8160, 1536), "c3d": (8.2e6, 8160, 4096), "c5": (21.4e6, 32400, 2816), "c3s": (5.3e6, 8160, 2816),
        "one of 8 column bands of c3": (345e3, 1020, 256), "one of 4": (690e3, 2040, 512), "one of 2": (1.35e6, 4080, 768), "c1": (3e4, 1200, 256),
    }
    for name, (pairs, tiles, want) in cases.items():
        seg = bench.auto_segment(pairs / tiles, pairs)
        assert seg == want and seg % 256 == 0 and 256 <= seg <= 4096, (name, seg, want)
    for p in (1.0, 2047.0, 2049.0 * 256, 1e9):
        for ppt in (0.0, 1.0, 1e4):
            seg = bench.auto_segment(ppt, p)
            assert seg % 256 == 0 and 256 <= seg <= 4096
    # the options the round's A/B lines use exist
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    for opt in ("--order", "--depth-sort", "--item-order", "--no-chunk-cull", "--vertex-stage", "--composite"):
        assert opt in r.stdout, opt
