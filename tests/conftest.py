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
def ref(request):
    """The real reference (oracle/_ref/libhsrans_ref.so, built here from /root/reference and shipped to the GPU box with the
    snapshot).  A GPU run without it must not go green by skipping every reference-stream test: it FAILS instead; only a
    CPU-only checkout without /root/reference may skip."""
    from oracle_lib import Ref

    if not Ref.available():
        import torch

        markexpr = request.config.getoption("-m") or ""
        if torch.cuda.is_available() or ("gpu" in markexpr and "not gpu" not in markexpr):
            pytest.fail("oracle/_ref/libhsrans_ref.so is missing on a GPU run: build it in the container (python -c 'import __graft_entry__ as g; g.build()') "
                        "so that it travels with the snapshot; the reference-stream tests must not be skipped here")
        pytest.skip("oracle/_ref/libhsrans_ref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the product has no CPU fallback)")
    import hypersonic_rans_amd as H

    return H.Context(0)
