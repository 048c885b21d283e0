import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bench_model():
    """The reference benchmark model (make_model defaults): (fly, world, compiled)."""
    from flygym_amd.models import make_model

    fly, world, _ = make_model()
    return fly, world, world.compile_model()


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle as orc

    orc.build()
    return orc
