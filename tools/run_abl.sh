cd /root/repo
timeout -k 10 120 python3 -m pytest tests/test_gpu_parity.py -x -q -k "structured_grid" 2>&1 | tail -4 || exit 1
bash tools/run_quad.sh
