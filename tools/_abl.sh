for v in "" _abl1 _abl2 _abl3; do
  for g in 512 256; do
    echo -n "lib$v grid $g: "; KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip$v.so python tools/spmv_only.py $g 30 1 poisson 2>&1 | grep "^grid" | sed 's/.*nq 1: //'
  done
done
