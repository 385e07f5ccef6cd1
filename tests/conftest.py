import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Both fp32 arithmetics of the engine are product paths: `bf16x3` (the default: fp32 operands split exactly into three bf16 values,
# six partial products on the bf16 matrix core, fp32 accumulate) and `f32` (v_mfma_f32_32x32x2_f32).  Every -m gpu test runs once
# per mode, so the driver's GPU record proves the whole suite -- kernels, networks, steps, goldens -- in both.
# TNR_TEST_MMA=f32 | bf16x3 restricts the run to one mode (kernel experiments).
MMA_MODES = [m for m in os.environ.get("TNR_TEST_MMA", "bf16x3,f32").split(",") if m]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None and "mma_mode" in metafunc.fixturenames:
        metafunc.parametrize("mma_mode", MMA_MODES, indirect=True, ids=["mma_" + m for m in MMA_MODES])


@pytest.fixture(autouse=True)
def mma_mode(request, monkeypatch):
    """GPU tests: the matrix-core arithmetic of the fp32 path for this test (ops.FP32_MMA / ops.MMA); None elsewhere."""
    mode = getattr(request, "param", None)
    if mode is None:
        yield None
        return
    from trainner_amd import hip, ops
    code = {"f32": hip.MMA_F32, "bf16x3": hip.MMA_BF16X3}[mode]
    monkeypatch.setattr(ops, "FP32_MMA", code)
    monkeypatch.setattr(ops, "MMA", code)
    yield mode


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
