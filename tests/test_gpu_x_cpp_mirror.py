"""Runs the C++ mirror's re-encoding of the reference tests (tests/cpp/test_mirror.cpp) on the GPU."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_mirror")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "kryst_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "CPP_MIRROR_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
