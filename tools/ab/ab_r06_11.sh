# the reference's (p + 1 - high) scatter weight restated in make_tap: the 1 x 65536 image at the standard tolerance; full suite
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gputest_b.txt; cat gpurun_out/r06_gputest_b.txt
python tools/exp.py cfg2 cfg3 cfg4 cfg5 --label r06-mid --steps 20 2>&1 | grep -v amdgpu.ids
