python tools/exp.py cfg3 --label q4 --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_NHWC_VEC=0 python tools/exp.py cfg3 --label scalar --steps 20 2>&1 | grep -v amdgpu.ids
python tools/exp.py cfg3 --label q4 --steps 20 2>&1 | grep -v amdgpu.ids
MDCONV_HP_NHWC_VEC=0 python tools/exp.py cfg3 --label scalar --steps 20 2>&1 | grep -v amdgpu.ids
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o p -- python $GRAFT_REPO_ROOT/tools/exp.py cfg3 --steps 10 > /dev/null 2>&1; grep -h "nchw_to_nhwc" /tmp/p1/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_hp.py -m gpu -q -x 2>&1 | tail -3
