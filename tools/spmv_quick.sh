cd /root/repo
for g in 256 512; do echo -n "grid $g: "; timeout -k 10 200 python3 tools/spmv_only.py $g 30 1 2>/dev/null | sed -n 1p; done
