cd /root/repo
for v in "" _nt3; do
  echo "lib$v: "; KRYST_HIP_LIB=/root/repo/kryst_amd/lib/libkryst_hip$v.so timeout -k 10 300 python3 tools/fgmres_only.py 256 30 120 | cut -c1-140 || exit 1
  KRYST_HIP_LIB=/root/repo/kryst_amd/lib/libkryst_hip$v.so timeout -k 10 600 python3 tools/bench_configs.py 256 64 | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'][:60], round(d['iterations_per_sec'], 1))" || exit 1
done
