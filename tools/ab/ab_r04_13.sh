MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg3 cfg5 --label plan --steps 5 2>&1 | grep -v amdgpu.ids | sort -u
for s in 2 3; do
MDCONV_HP_G2_SLOTS=$s python tools/exp.py cfg5 --label g2slots$s --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_HP_G2_SLOTS=$s MDCONV_BWD_FORK=0 python tools/exp.py cfg5 --label g2slots$s-nofork --steps 10 2>&1 | grep -v amdgpu.ids
done
python tools/exp.py cfg5 --label default --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=0 python tools/exp.py cfg5 --label default-nofork --steps 10 2>&1 | grep -v amdgpu.ids
