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
                                                  (3, 9, "aniso", "pcg"), (2, 6, "poisson", "pcg"),
                                                  (8, 16, "poisson", "pcg")])        # BASELINE config 4's shape: Jacobi-PCG over 8 k-slabs
def test_partitioned_cg_matches_oracle(world, N, kind, method):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2" if world <= 4 else "1")
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


_RDZV = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
from kryst_amd.launch import Rendezvous
r = Rendezvous.from_env(timeout=60)
uid = bytes(range(128))
assert r.broadcast_bytes(uid if r.rank == 0 else None) == uid            # the RCCL unique id travels from rank 0
r.barrier()
assert r.allreduce_max(1.5 * r.rank) == 1.5 * (r.world - 1)              # max over ranks of the timed region
assert r.gather({"rank": r.rank, "ms": 0.25 * r.rank}) == [{"rank": i, "ms": 0.25 * i} for i in range(r.world)]
r.barrier()
r.close()
print("RDZV_OK")
'''


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_socket_rendezvous_without_torch(world):
    """kryst_amd/launch.py: the no-torch launcher plumbing bench.py --launcher socket uses (unique id broadcast, barrier, max,
    gather), also when the first port of its range is already taken by somebody else."""
    port = _free_port()
    squatter = socket.socket()                                              # occupy MASTER_PORT + 1: the rendezvous must move on
    try:
        squatter.bind(("127.0.0.1", port + 1)); squatter.listen(1)
    except OSError:
        squatter = None
    procs = [subprocess.Popen([sys.executable, "-c", _RDZV, ROOT],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=120)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        if squatter:
            squatter.close()
    assert all(p.returncode == 0 and "RDZV_OK" in o for p, o in zip(procs, outs)), "\n".join(o[-2000:] for o in outs)
