import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# a run that dies on SIGABRT prints the aborting thread's native stack first (kaldi_amd/csrc/common.cc; read when the library is loaded)
os.environ.setdefault("KAMD_ABORT_BACKTRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    # a fresh checkout has no built library (the .so files are git-ignored): build it once, as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  Building is not a fallback:
    # without hipcc the tests that need the library fail loudly.
    lib = os.path.join(ROOT, "kaldi_amd", "lib", "libkaldi_amd.so")
    if not os.path.exists(lib) and not os.environ.get("KAMD_LIB"):
        try:
            import __graft_entry__
            __graft_entry__.build()
        except Exception as e:                          # noqa: BLE001 - reported, then the tests speak for themselves
            sys.stderr.write("conftest: building the HIP library failed: %s\n" % e)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
