cd /root/repo
for i in 1 2; do
  for na in 0 1; do
    echo "== KRYST_NO_ARENA=$na"; KRYST_NO_ARENA=$na timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config1_256']['value'])" || exit 1
  done
done
