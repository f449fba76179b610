cd /root/repo
for v in "" _abl1 _abl2 _abl3; do
  echo "lib$v"
  for g in 256 512; do KRYST_ILU_WAVE=2 KRYST_HIP_LIB=kryst_amd/lib/libkryst_hip$v.so timeout -k 10 200 python3 tools/ilu_only.py $g 20 true 2>&1 | tail -1; done
done
