cd /root/repo
python3 tools/tune_plain.py 512 3 > gpurun_out/tune_plain_512.txt 2>&1 || exit 1
python3 tools/tune_plain.py 256 3 > gpurun_out/tune_plain_256.txt 2>&1 || exit 1
bash tools/plain_pmc.sh 512 base512 > gpurun_out/pmc_base512.txt 2>&1 || exit 1
bash tools/plain_pmc.sh 512 nt512 KRYST_SPMV_NT=1 > gpurun_out/pmc_nt512.txt 2>&1 || exit 1
bash tools/plain_pmc.sh 512 ntg8_512 KRYST_SPMV_NT=1 KRYST_SPMV_GROUP=8 > gpurun_out/pmc_ntg8_512.txt 2>&1 || exit 1
bash tools/plain_pmc.sh 256 base256 > gpurun_out/pmc_base256.txt 2>&1 || exit 1
echo ALL_OK
