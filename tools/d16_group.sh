cd /root/repo
for g in 256 512; do for grp in 1 2 4 8 16 32 64; do echo -n "grid $g GROUP=$grp: "; KRYST_SPMV_GROUP=$grp timeout -k 10 200 python3 tools/spmv_only.py $g 30 1 2>/dev/null | sed -n 1p; done; done
