"""Rehearses the multi-rank code path on the one GPU a test box has (tools/dist_rehearsal.py): RCCL is loaded after
torch.distributed, collectives run with a one-rank communicator, results must be bit-identical to the plain path."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_forced_collective_path_matches_plain_path():
    env = dict(os.environ, KRYST_FORCE_COMM="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_rehearsal.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist rehearsal ok" in r.stdout
