# r05 call 29 (final collection on the sources with the rewritten 16-bit grad_bias kernel and bias epilogues):
# full GPU suite + smoke + tools/collect_profiles.sh r05 + tools/collect_extra.sh r05 + bias cost of every BASELINE shard
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python tools/bias_cost.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab_r05_29.txt 2>&1
bash tools/collect_profiles.sh r05 >> gpurun_out/ab_r05_29.txt 2>&1
bash tools/collect_extra.sh r05 >> gpurun_out/ab_r05_29.txt 2>&1
tail -60 gpurun_out/ab_r05_29.txt
