import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pose-graph-initialization_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, PKG)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts():
    """Built .so files are git-ignored: (re)build them when missing or older than their sources."""
    csrc = [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc"))]
    host = [os.path.join(PKG, "host", f) for f in os.listdir(os.path.join(PKG, "host"))]
    hdr = [os.path.join(ROOT, "include", "pgi.h"), os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")]
    if (_stale(os.path.join(PKG, "libpgi.so"), csrc + hdr) or _stale(os.path.join(PKG, "libpgi_host.so"), host + hdr)
            or _stale(os.path.join(PKG, "test_host_api"), host + hdr)
            or any(_stale(os.path.join(PKG, exe), host + [os.path.join(ROOT, "tests", "cpp", exe + ".cpp")])
                   for exe in ("test_astar", "test_scheduler", "test_pipeline", "test_distributed"))):
        subprocess.check_call(["make", "-C", PKG, "-s"])
    if _stale(os.path.join(ROOT, "oracle", "libpgi_oracle.so"),
              [os.path.join(ROOT, "oracle", f) for f in ("pgi_oracle.c", "pgi_oracle.h")]):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    yield
