import os
import sys

import pytest

try:
    # PyTorch-ROCm bundles its own libamdhip64; it has to be the copy of the HIP runtime the process loads FIRST, or its
    # later initialisation finds "No HIP GPUs" (libgswt_hip.so alone would pull in /opt/rocm's copy).  Tests that hand
    # torch device buffers to the C ABI need both.
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU-only tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def renderer():
    """One GSWTRenderer (one HIP context) for the whole GPU session."""
    from gswt_renderer_amd.renderer import GSWTRenderer
    r = GSWTRenderer(0)
    yield r
    r.close()
