import os
import sys

import pytest

# PyTorch's ROCm wheel bundles its own HIP / HSA runtime.  In one process the first HIP runtime loaded wins;
# loading libldpc_toolbox.so (linked against /opt/rocm) BEFORE torch leaves torch with a runtime it was not
# built for ("No HIP GPUs are available").  The tests use torch for device buffers, so it goes first.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding as ob
    ob.build()
    return ob


@pytest.fixture(scope="session")
def toy_h():
    """The 4x6 matrix of the reference's decoder tests (src/decoder/flooding.rs:143-151)."""
    from ldpc_toolbox_amd import SparseMatrix
    h = SparseMatrix(4, 6)
    h.insert_row(0, [0, 1, 3])
    h.insert_row(1, [1, 2, 4])
    h.insert_row(2, [0, 4, 5])
    h.insert_row(3, [2, 3, 5])
    return h
