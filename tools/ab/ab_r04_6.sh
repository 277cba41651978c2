MDCONV_BW_SPLITS=14 python tools/exp.py cfg2 --label split-s14 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=14 MDCONV_BWD_FORK=2 python tools/exp.py cfg2 --label split-s14-g2first 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=14 MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label split-s14-nofork 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=14 MDCONV_FORK_PRIO=0 python tools/exp.py cfg2 --label split-s14-prio0 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=14 MDCONV_FORK_PRIO=1 python tools/exp.py cfg2 --label split-s14-priolow 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=10 python tools/exp.py cfg2 --label split-s10 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=12 python tools/exp.py cfg2 --label split-s12 2>&1 | grep -v amdgpu.ids
MDCONV_BW_SPLITS=16 python tools/exp.py cfg2 --label split-s16 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=28 python tools/exp.py cfg2 --label split-wide-s28 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=28 MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label split-wide-s28-nofork 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=14 python tools/exp.py cfg2 --label split-wide-s14 2>&1 | grep -v amdgpu.ids
MDCONV_BD_SPLIT=0 MDCONV_BW_SPLITS=14 python tools/exp.py cfg2 --label fused-s14 2>&1 | grep -v amdgpu.ids
