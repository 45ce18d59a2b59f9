import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref

    if not Ref.available():
        pytest.skip("oracle/_ref/libhsrans_ref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the product has no CPU fallback)")
    import hypersonic_rans_amd as H

    return H.Context(0)
