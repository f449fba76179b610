cd /root/repo
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "plain_kernel_variants or kernel_forms" > gpurun_out/tile_parity.txt 2>&1 || { tail -30 gpurun_out/tile_parity.txt; exit 1; }
tail -3 gpurun_out/tile_parity.txt
python3 tools/tune_plain.py 512 3 tile > gpurun_out/tune_tile_512.txt 2>&1 || exit 1
python3 tools/tune_plain.py 256 3 tile > gpurun_out/tune_tile_256.txt 2>&1 || exit 1
echo ALL_OK
