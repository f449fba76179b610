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


def test_plain_gpus_n_launches_its_own_ranks_without_touching_the_gpu(tmp_path):
    """`python3 bench.py --gpus 2` started the way the driver starts `--gpus 1` (no WORLD_SIZE): the process becomes the launcher -- it starts
    the two rank processes itself, maps neither libkryst_hip.so nor the HIP runtime (nor torch), relays the ranks' stage markers, and returns
    the first non-zero exit code of a rank (here: there is no GPU, every rank leaves at the device check with code 2)."""
    import subprocess
    maps = tmp_path / "launcher.maps"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KRYST_BENCH_LAUNCHER_MAPS=str(maps), KRYST_BENCH_WATCHDOG_S="120")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert "[bench launcher" in r.stderr and "started 2 rank process(es)" in r.stderr, r.stderr[-2000:]
    assert "[bench rank 0 +" in r.stderr and "[bench rank 1 +" in r.stderr, r.stderr[-2000:]      # every rank's markers come through
    assert "must be launched by torch.distributed.run" not in r.stderr
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])                                     # the ranks' own "no device for this rank" exit
    assert "one GPU per rank is required" in r.stderr
    assert r.stdout.strip() == ""                                                                  # nothing but rank 0's JSON line ever goes to stdout
    m = maps.read_text()
    for name in ("libkryst_hip", "libamdhip64", "libhsa-runtime", "libtorch", "librccl"):
        assert name not in m, name


def test_launcher_rejects_ranks_that_do_not_divide():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--ranks-per-process", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not a multiple" in r.stderr
