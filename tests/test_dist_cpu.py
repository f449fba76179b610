"""world_size-2 (and 3) gloo rehearsal of the row-partitioned path on CPU: partitioning, halo plan, local
numbering, rank-ordered inner products (tests/dist_cpu_worker.py)."""
import os
import socket
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,N,kind,method", [(2, 8, "poisson", "cg"), (2, 8, "poisson", "pcg"),
                                                  (3, 9, "aniso", "pcg"), (2, 6, "poisson", "pcg")])
def test_partitioned_cg_matches_oracle(world, N, kind, method):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_cpu_worker.py"), str(N), kind, method],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=300)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    assert "DIST_OK" in outs[0]
