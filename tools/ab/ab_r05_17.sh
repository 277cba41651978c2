# r05 call 17: final collections on the final sources (the ablation blocks added after call 12 changed the source hash)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
} > gpurun_out/ab_r05_17.txt 2>&1
bash tools/collect_profiles.sh r05 >> gpurun_out/ab_r05_17.txt 2>&1
bash tools/collect_extra.sh r05 >> gpurun_out/ab_r05_17.txt 2>&1
tail -12 gpurun_out/ab_r05_17.txt
