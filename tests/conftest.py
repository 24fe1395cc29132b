import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--arith", default="", help="arithmetic forms for the whole run, e.g. --arith desc_conv=direct,pose_conv=direct "
                     "(bufferx_amd.config.ARITH_FORMS; product and oracle both follow cfg.arith -- nothing reads the environment)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    spec = config.getoption("--arith")
    if spec:
        import bufferx_amd.config as C
        for kv in spec.split(","):
            k, v = kv.split("=")
            assert k in C.ARITH_FORMS and v in C.ARITH_FORMS[k], (k, v, C.ARITH_FORMS)
            C.ARITH_DEFAULT[k] = v


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def bx():
    import bufferx_amd
    return bufferx_amd


@pytest.fixture(scope="session")
def packed(bx):
    return bx.weights.fold_and_pack(bx.weights.synthetic_state_dict(0))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
