import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _cap_cpu_threads():
    """The oracle and the CPU clip generator are made of many small torch ops; with one thread per core of a 128 / 256-CPU GPU host
    they crawl (a 16-frame clip: 10 s; bench.py's cpu_baseline leg measured 32 threads as the fastest setting).  Results do not depend
    on the thread count."""
    try:
        import torch
        if torch.get_num_threads() > 32:
            torch.set_num_threads(32)
    except Exception:                    # noqa: BLE001
        pass


def pytest_sessionstart(session):
    _cap_cpu_threads()
    """The suite needs the in-tree library.  It normally exists (python -m sttm_amd.build / __graft_entry__.build()); on a fresh
    checkout build it once here (hipcc cross-compiles without a GPU, ~100 s) instead of failing every test that loads it."""
    lib = os.path.join(REPO, "sttm_amd", "lib", "libsttm_hip.so")
    if not os.path.exists(lib):
        try:
            from sttm_amd import build as b
            b.build()
        except Exception as e:           # noqa: BLE001  -- the tests that need the library will report the real problem
            sys.stderr.write(f"[conftest] could not build libsttm_hip.so: {e}\n")
