# r05 call 12: full GPU suite after the bf16 fix (long 2-D entries, MFMA sums for fp16 only), small-forward plans, final collections
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "## small forwards, mixed tail plan"
python tools/small_fwd.py 2>&1 | grep -v amdgpu.ids
echo "## small forwards, round-4 plan (MDCONV_FWD_TAIL=1: no split below one dispatch round)"
MDCONV_FWD_TAIL=1 python tools/small_fwd.py 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg2 cfg3 cfg4 cfg5 --label final --steps 20 2>&1 | grep -v amdgpu.ids
} > gpurun_out/ab_r05_12.txt 2>&1
bash tools/collect_profiles.sh r05 >> gpurun_out/ab_r05_12.txt 2>&1
bash tools/collect_extra.sh r05 >> gpurun_out/ab_r05_12.txt 2>&1
tail -45 gpurun_out/ab_r05_12.txt
