# r05 call 7: full GPU suite + smoke + profile collection of the current tree (tools/collect_profiles.sh r05)
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
} > gpurun_out/ab_r05_7.txt 2>&1
bash tools/collect_profiles.sh r05 >> gpurun_out/ab_r05_7.txt 2>&1
cat gpurun_out/ab_r05_7.txt | tail -40
