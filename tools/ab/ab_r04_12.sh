for s in 0 18 28 37 56; do
  MDCONV_BW_SPLITS=$s python tools/exp.py cfg4 --label cfg4-s$s --steps 10 2>&1 | grep -v amdgpu.ids
done
MDCONV_BWD_FORK=0 python tools/exp.py cfg4 --label cfg4-nofork --steps 10 2>&1 | grep -v amdgpu.ids
