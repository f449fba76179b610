"""bench.py's launcher plumbing without a GPU: the group that lets several RANK THREADS of one process take part in the collectives
around the timed region (ThreadedGroup: rehearsals of 8 ranks as 4 processes x 2 threads), and the environment override that only the
process's leader applies between two barriers."""
import importlib.util
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_threaded_group_collectives_over_a_solo_inner_group():
    B = _bench()
    per = 3
    shared = B.ThreadedGroup.Shared(B.SoloGroup(), per)
    out = [None] * per

    def run(t):
        g = B.ThreadedGroup(shared, t)
        assert (g.rank, g.world, g.env_leader) == (t, per, t == 0)
        g.barrier()
        mx = g.allreduce_max(10.0 + t)
        got = g.gather({"rank": t})
        bc = g.broadcast_bytes(b"id" if t == 0 else None)
        with B.env_override(g, KRYST_TEST_KNOB="7"):
            inside = os.environ.get("KRYST_TEST_KNOB")
        g.barrier()
        out[t] = (mx, got, bc, inside, os.environ.get("KRYST_TEST_KNOB"))
        g.close()

    ts = [threading.Thread(target=run, args=(t,)) for t in range(per)]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert all(o is not None for o in out)
    for mx, got, bc, inside, after in out:
        assert mx == 12.0 and got == [{"rank": 0}, {"rank": 1}, {"rank": 2}] and bc == b"id" and inside == "7" and after is None


def test_stage_markers_and_batches_argument(capsys):
    B = _bench()
    B.stage(5, "timed iterations")
    assert "[bench rank 5 +" in capsys.readouterr().err and B._STAGE[5] == "timed iterations"
    sys.argv, old = ["bench.py", "--help"], sys.argv
    try:
        try:
            B.main()
        except SystemExit:
            pass
    finally:
        sys.argv = old
    assert "--batches" in capsys.readouterr().out
