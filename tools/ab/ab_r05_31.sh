# r05 calls 31 + 32 (re-entry after the container was replaced: the summaries of call 29 were lost with it).
# Same collection as call 29 on the final sources (kernel_sources_sha16 e190a69e3f8e83ab), in two gpurun calls:
#   31: tools/collect_profiles.sh r05c + smoke + full GPU suite (751 passed, 16 skipped; 6.7 GPU-minutes)
#   32: bench.py (traffic stamp valid) + tools/collect_extra.sh r05
mkdir -p gpurun_out
bash tools/collect_profiles.sh r05c > gpurun_out/collect_r05c.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r05c.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_r05c.log 2>&1
timeout 300 python bench.py > gpurun_out/bench_final_r05c.json 2> gpurun_out/bench_final_r05c.err
timeout 1100 bash tools/collect_extra.sh r05 > gpurun_out/collect_extra_r05.log 2>&1
# locally: python tools/summarize_profile.py gpurun_out/prof_r05c r05; python tools/summarize_extra.py r05
