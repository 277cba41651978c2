D=modulated_deform_conv_amd
for v in nogather noweight nobarrier nogw nomfma nogwb; do
  MDCONV_FWD_TAIL=0 MDCONV_LIB=$PWD/$D/libmdconv_hip_fa_$v.so python tools/exp.py cfg2 --label abl-$v --steps 10 2>&1 | grep -v amdgpu.ids
done
MDCONV_FWD_TAIL=0 python tools/exp.py cfg2 --label tail0 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=2 python tools/exp.py cfg2 --label tail2 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=3 python tools/exp.py cfg2 --label tail3 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TAIL=4 python tools/exp.py cfg2 --label tail4 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TILE=256x64 MDCONV_FWD_TAIL=0 python tools/exp.py cfg2 --label 256x64-tail0 2>&1 | grep -v amdgpu.ids
MDCONV_FWD_TILE=256x64 MDCONV_DEBUG_PLAN=1 python tools/exp.py cfg2 --label 256x64-tail 2>&1 | grep -v amdgpu.ids | sort -u
MDCONV_FWD_TILE=256x64 MDCONV_FWD_TAIL=4 python tools/exp.py cfg2 --label 256x64-tail4 2>&1 | grep -v amdgpu.ids
for s in 12 14 16 18; do
MDCONV_BW_SPLITS=$s python tools/exp.py cfg2 --label s$s 2>&1 | grep -v amdgpu.ids
MDCONV_BWD_FORK=2 MDCONV_BW_SPLITS=$s python tools/exp.py cfg2 --label s$s-gemm2first 2>&1 | grep -v amdgpu.ids
done
MDCONV_BWD_FORK=2 python tools/exp.py cfg2 --label s21-gemm2first 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=21 python tools/exp.py cfg2 --label wide-s21 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=14 python tools/exp.py cfg2 --label wide-s14 2>&1 | grep -v amdgpu.ids
MDCONV_BW_WIDE=1 MDCONV_BWD_FORK=2 python tools/exp.py cfg2 --label wide-gemm2first 2>&1 | grep -v amdgpu.ids
