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
