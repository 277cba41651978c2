# full GPU suite on the ABI-v2 / stripped-source tree
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputest_a.txt
cat gpurun_out/r06_gputest_a.txt
