import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs what ships: every model follows its machine's tuning record (dnastore_amd/tune/) exactly as bench.py and
# smoke() do; nothing is timed at model creation (autotune is opt-in: test_gpu_viterbi.py::test_row_program_autotune).

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DATA = os.path.join(GOLDEN, "ref_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_data():
    return REF_DATA


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """The PyTorch ROCm wheel bundles its own HIP runtime (torch/lib/libamdhip64.so) while libdnastore_amd.so links
    the system one (/opt/rocm/lib/libamdhip64.so.7): a process that uses both ends up with two runtimes.  That works
    when PyTorch's comes up first (the order bench.py uses); bringing it up after the library has already created and
    destroyed device state can fail with "No HIP GPUs are available".  So on a GPU box the test session initialises
    torch.cuda before anything else; without a GPU this is a no-op."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
