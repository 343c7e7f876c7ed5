import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jurassic-gpu_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _torch_first():
    """PyTorch's ROCm wheel brings its own HIP runtime.  In a process that uses both torch and libjurassic_hip.so the
    runtime that is loaded first serves both, and torch only finds its devices when that is its own: initialise
    torch's side before any test loads the library (harmless without a GPU)."""
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


def pytest_sessionstart(session):
    _torch_first()
    """The built library and tools are not in the history (only in the working tree): build what is missing
    (hipcc cross-compiles without a GPU; about two minutes from scratch, nothing when up to date)."""
    pkg = os.path.join(ROOT, "jurassic-gpu_amd")
    need = ["libjurassic_hip.so", "libjurassic_hip_nd2378.so", "formod", "formod_nd2378", "climatology", "limb", "nadir"]
    if not all(os.path.exists(os.path.join(pkg, f)) for f in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build()
    import bench
    orc.set_threads(bench.usable_cores())      # the affinity mask may show more threads than the cgroup quota grants
    return orc
