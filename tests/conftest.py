import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libraries():
    """The .so files are git-ignored build artefacts; (re)build whatever is missing or stale
    (hipcc cross-compiles for gfx950 without a GPU)."""
    from numpower_amd import build
    build.build_all()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (oracle/np_oracle.c), built on demand."""
    from oracle import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def hip():
    """The device library through ctypes; skips nothing: a missing .so is a hard error."""
    from numpower_amd import device
    device.init(int(os.environ.get("NP_TEST_DEVICE", "0")))
    return device


@pytest.fixture(scope="module", autouse=True)
def _pool_back_to_the_driver_between_modules():
    """The caching pool keeps every freed block (by exact 2 MiB size class) for reuse, and the full-size modules free buffers of 7 - 17 GB in
    a dozen different sizes: behind each test module the cache goes back to the driver, so that one pytest process never holds more than
    one module's worth of the device (only if the library was used at all: the CPU tier never loads it on account of this)."""
    yield
    from numpower_amd import _lib
    if _lib._lib is not None and os.environ.get("NP_TEST_KEEP_POOL") != "1":
        import ctypes
        freed = ctypes.c_size_t()
        _lib._lib.np_pool_trim(ctypes.byref(freed))
