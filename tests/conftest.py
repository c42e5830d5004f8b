import os, sys
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """The PyTorch-ROCm wheel bundles its own HIP runtime; libqpgpu.so links the system one (/opt/rocm). Both coexist in a
    process only if torch's is initialised first (otherwise torch reports "No HIP GPUs are available"), so GPU sessions
    bring torch.cuda up before any qpgpu call. On a machine without a GPU this is a no-op."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
