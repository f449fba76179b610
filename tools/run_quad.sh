cd /root/repo
timeout -k 10 600 python3 -m pytest tests/test_gpu_0_parity.py -x -q -k "structured_grid or give_up or plane_kernels or ilu" > gpurun_out/quad_parity.txt 2>&1; tail -12 gpurun_out/quad_parity.txt
for w in 2 1; do for g in 128 256 384 512; do KRYST_ILU_WAVE=$w timeout -k 10 200 python3 tools/ilu_only.py $g 20 true 2>&1 | tail -1; done; done
