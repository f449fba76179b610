cd /root/repo
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "pipelined or kernel_forms or general_matrices" > gpurun_out/pipe_parity.txt 2>&1 || { tail -30 gpurun_out/pipe_parity.txt; exit 1; }
tail -3 gpurun_out/pipe_parity.txt
python3 tools/tune_plain.py 512 3 pipe > gpurun_out/tune_pipe_512.txt 2>&1 || exit 1
python3 tools/tune_plain.py 256 3 pipe > gpurun_out/tune_pipe_256.txt 2>&1 || exit 1
echo ALL_OK
