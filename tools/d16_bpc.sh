cd /root/repo
for g in 256 512; do for bpc in 0 8 16 4; do echo -n "grid $g BPC=$bpc: "; KRYST_SPMV_GROUP=8 KRYST_SPMV_BLOCKS_PER_CU=$bpc timeout -k 10 200 python3 tools/spmv_only.py $g 30 1 2>/dev/null | sed -n 1p; done; done
