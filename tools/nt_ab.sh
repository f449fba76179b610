cd /root/repo
for v in "" _nt1 _nt2 _nt3 ""; do
  echo -n "lib$v: "; KRYST_HIP_LIB=/root/repo/kryst_amd/lib/libkryst_hip$v.so timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config1_256']['value'], d['roofline']['ms_per_launch'])" || exit 1
done
