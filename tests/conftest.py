import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so importlib)."""
    return importlib.import_module("pbrt-r3_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same soname as /opt/rocm's).  Whichever copy a process maps
    # first serves every later library, and a second copy does not find the GPU -- so the tests that hand the film to
    # torch.distributed (test_gpu_dist.py) need torch's copy in place before libpbrtgpu.so is opened, as bench.py has it.
    import torch  # noqa: F401
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()
