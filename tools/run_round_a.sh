cd /root/repo
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/gputests_b.txt 2>&1; tail -5 gpurun_out/gputests_b.txt
bash tools/bench_rehearsal.sh 2 torch > gpurun_out/bench_mr2_torch.json 2> gpurun_out/bench_mr2_torch.err; tail -c 1500 gpurun_out/bench_mr2_torch.json
bash tools/bench_rehearsal.sh 3 socket > gpurun_out/bench_mr3_socket.json 2> gpurun_out/bench_mr3_socket.err; tail -c 1500 gpurun_out/bench_mr3_socket.json
timeout -k 10 900 python3 bench.py > gpurun_out/bench_n1_a.json 2> gpurun_out/bench_n1_a.err; tail -c 3000 gpurun_out/bench_n1_a.json; tail -5 gpurun_out/bench_n1_a.err
