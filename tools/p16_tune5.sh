cd /root/repo
for G in 512 256; do for tpw in 1 2 4 8 16; do for grp in 8 16; do echo -n "grid $G TPW=$tpw GROUP=$grp: "; KRYST_SPMV_GROUP=$grp KRYST_SPMV_PATTERN_TPW=$tpw timeout -k 10 200 python3 tools/spmv_only.py $G 30 1 2>/dev/null | sed -n 1p; done; done; done
