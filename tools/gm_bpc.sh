cd /root/repo
for b in 2 3 4 6 8; do echo -n "GMRES_BPC=$b: "; KRYST_GMRES_BLOCKS_PER_CU=$b timeout -k 10 200 python3 tools/fgmres_only.py 256 30 120 | grep '"gmres"' | cut -c1-120; done
