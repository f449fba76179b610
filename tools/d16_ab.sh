cd /root/repo
for g in 256 512; do for c in 0 1 2; do echo "== grid $g COMPRESS=$c"; KRYST_SPMV_KERNEL=3 KRYST_SPMV_COMPRESS=$c timeout -k 10 200 python3 tools/spmv_only.py $g 30 1 | head -1 || exit 1; done; done
timeout -k 10 400 python3 bench.py --steps 200 --warmup 20
