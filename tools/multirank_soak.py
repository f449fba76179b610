#!/usr/bin/env python3
"""Post-fix validation of the multi-rank path: the shared-GPU cases of tests/test_gpu_z_multirank_shim.py, each started REPS times
in fresh processes, one after the other (never in parallel copies), every rank's stage log kept.  Writes one line per run and a
summary; exit code 1 when a run failed (the failing ranks' log tails are printed -- their last stage marker names the stage).

    python tools/multirank_soak.py [REPS=10] [out=gpurun_out/multirank_soak.log] [case ...]       case = P,N,kind   e.g. 4,1031,random

This is a regression check for the round-2 fault (hipMemset on the null stream zeroing a preconditioner's argument block after the
first apply had written it: DESIGN.md section 6, tools/micro/nullstream_memset.hip), not a fault hunt: the cause was found by
reading and is demonstrated deterministically by the micro test."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim", "librccl_shim.so")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "multirank_soak.log")
    cases = [tuple(c.split(",")) for c in sys.argv[3:]] or [("4", "1031", "random"), ("3", "2500", "random")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    bad = 0
    with open(out, "w") as log:
        def say(msg):
            print(msg, flush=True); log.write(msg + "\n"); log.flush()
        say(f"# multirank soak: {reps} sequential runs per case, cases {cases}, shim {os.path.basename(SHIM)}")
        for P, N, kind in cases:
            for rep in range(reps):
                with tempfile.TemporaryDirectory() as tmp:
                    env = dict(os.environ, KRYST_RCCL_LIB=SHIM, KRYST_STENCIL_HOST="0")
                    t0 = time.time()
                    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multirank_worker.py"), str(r), P, tmp, N, kind],
                                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(int(P))]
                    outs = []
                    for p in procs:
                        try:
                            o, _ = p.communicate(timeout=300)
                        except subprocess.TimeoutExpired:
                            p.kill(); o, _ = p.communicate()
                            o += "\n[soak] killed after 300 s"
                        outs.append(o)
                    ok = all(p.returncode == 0 for p in procs)
                    say(f"case P={P} N={N} {kind} run {rep + 1}/{reps}: {'ok' if ok else 'FAILED'} in {time.time() - t0:.1f} s, "
                        f"exit codes {[p.returncode for p in procs]}")
                    if not ok:
                        bad += 1
                        for r, o in enumerate(outs):
                            say(f"---- rank {r} log tail ----\n{o[-2500:]}")
                        say("# stopping at the first failure (no further GPU work after a fault)")
                        say(f"# summary: {bad} failed")
                        return 1
        say(f"# summary: all {reps * len(cases)} runs ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
