# state of the tree: full GPU suite, smoke, the bench line with traffic fields
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_gputest_final.txt; cat gpurun_out/r06_gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_line.err; tail -c 400 gpurun_out/r06_bench_line.json
