#!/bin/bash
# The whole GPU parity suite under every fallback switch (each line must end in "passed").
cd /root/repo
for cfg in "KRYST_SPMV_COMPRESS=0" "KRYST_SPMV_COMPRESS=1" "KRYST_SPMV_COMPRESS=2" "KRYST_SPMV_REUSE_DIAG=0" "KRYST_ILU_WAVE=0" "KRYST_ILU_GRID=0" "KRYST_ILU_GRID=0 KRYST_ILU_SYNCFREE=0" "KRYST_NO_ARENA=1" "KRYST_ILU_GRAPH=0" "KRYST_STENCIL_HOST=1"; do
  echo -n "$cfg: "; env $cfg timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -1 || exit 1
done
