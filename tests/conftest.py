import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def jload(arr):
    return json.loads(str(arr))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _restore_global_switches(request):
    """GPU tests flip process-wide switches (compute dtype, kernel A/B selectors); a test that fails half-way must not leave
    them set for the tests after it."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    try:
        import torch
        from eventpretrain_amd import ops
        ops.set_compute_dtype(torch.float32)
        ops.set_window_mfma(True)
        ops.set_fused_attention(True)
        ops.set_grad_side(True)
        ops.set_wgrad_g4(True)
        ops.set_wgrad_xcd_order(True)
        ops.set_deferred_grads(True)
        from eventpretrain_amd._lib import call
        call("evp_gemm_set_variant", 1)
        call("evp_gemm_set_variant", 10)
        call("evp_gemm_set_variant", 100)
        call("evp_gemm_set_variant", 19)
        call("evp_voxel_set_debug", 0)
    except Exception:
        pass
