cd /root/repo
for G in 512 256; do for grp in 4 8; do for bpc in 5 6 7 12 14; do echo -n "grid $G GROUP=$grp PBPC=$bpc: "; KRYST_SPMV_GROUP=$grp KRYST_SPMV_PATTERN_BLOCKS_PER_CU=$bpc timeout -k 10 200 python3 tools/spmv_only.py $G 30 1 2>/dev/null | sed -n 1p; done; done; done
