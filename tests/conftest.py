import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suite needs the in-tree library.  It normally exists (python -m sttm_amd.build / __graft_entry__.build()); on a fresh
    checkout build it once here (hipcc cross-compiles without a GPU, ~100 s) instead of failing every test that loads it."""
    lib = os.path.join(REPO, "sttm_amd", "lib", "libsttm_hip.so")
    if not os.path.exists(lib):
        try:
            from sttm_amd import build as b
            b.build()
        except Exception as e:           # noqa: BLE001  -- the tests that need the library will report the real problem
            sys.stderr.write(f"[conftest] could not build libsttm_hip.so: {e}\n")
