cd /root/repo
G=${1:-512}
for grp in 1 2 4 8 16 64; do for bpc in 8 6 4 16; do echo -n "grid $G GROUP=$grp PBPC=$bpc: "; KRYST_SPMV_GROUP=$grp KRYST_SPMV_PATTERN_BLOCKS_PER_CU=$bpc timeout -k 10 200 python3 tools/spmv_only.py $G 30 1 2>/dev/null | sed -n 1p; done; done
