cd /root/repo
for bpc in 2 4 8; do
  echo "== DOT_BPC=$bpc"; KRYST_DOT_BLOCKS_PER_CU=$bpc timeout -k 10 200 python3 tools/fgmres_only.py 256 30 120 2>&1 | grep '"fgmres"' || exit 1
done
