import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates rvseg contexts; torch (if present) is imported first so both share a HIP runtime."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    import rovinasemanticsegmentation_amd as rv

    made = []

    def make(**params):
        ctx = rv.Context(**params)
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()
