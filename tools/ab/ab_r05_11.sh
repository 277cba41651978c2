# r05 call 11: full GPU suite + smoke + the profile collections of the final tree
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
} > gpurun_out/ab_r05_11.txt 2>&1
bash tools/collect_profiles.sh r05 >> gpurun_out/ab_r05_11.txt 2>&1
bash tools/collect_extra.sh r05 >> gpurun_out/ab_r05_11.txt 2>&1
tail -30 gpurun_out/ab_r05_11.txt
