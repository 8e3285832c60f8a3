import sys
from pathlib import Path

import pytest

# torch ships its own ROCm runtime libraries: when the product library (linked against /opt/rocm) initialises HIP first, a
# later `import torch` finds no GPU ("No HIP GPUs are available"). Importing torch before anything touches HIP makes both
# share one runtime, whatever subset of the test modules is collected.
try:
    import torch  # noqa: F401
except ImportError:   # the CPU tier of a host without torch still runs the non-distributed tests
    torch = None

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Product library + checkers built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    import helpers
    helpers.build_all()
    return helpers


_index_cache = {}


@pytest.fixture(scope="session")
def small_index(built):
    """gencode_small.fa HostIndex per k (built on the CPU)."""
    def get(k):
        if k not in _index_cache:
            _index_cache[k] = built.pa.build_index(str(built.FASTA), k, 8)
        return _index_cache[k]
    return get
