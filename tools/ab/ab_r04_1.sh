set -x
export MDCONV_DEBUG_PLAN=1
python tools/exp.py cfg2 --label default 2>&1 | sort -u | tail -5
unset MDCONV_DEBUG_PLAN
MDCONV_FWD_TAIL=0 MDCONV_BW_SPLITS=29 python tools/exp.py cfg2 --label r03-equivalent
MDCONV_FWD_TAIL=0 python tools/exp.py cfg2 --label fwdtail0
MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label nofork
MDCONV_BWD_FORK=0 MDCONV_BW_SPLITS=29 python tools/exp.py cfg2 --label nofork-s29
MDCONV_BW_SPLITS=20 python tools/exp.py cfg2 --label s20
MDCONV_BW_SPLITS=14 python tools/exp.py cfg2 --label s14
MDCONV_BW_SPLITS=42 python tools/exp.py cfg2 --label s42
MDCONV_BW_WIDE=1 python tools/exp.py cfg2 --label wide
MDCONV_BW_WIDE=1 MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label wide-nofork
MDCONV_BW_WIDE=1 MDCONV_BW_SPLITS=57 MDCONV_BWD_FORK=0 python tools/exp.py cfg2 --label wide-nofork-s57
MDCONV_FWD_TAIL=4 python tools/exp.py cfg2 --label fwdtail4
python tools/exp.py cfg2:16 cfg2:8 cfg2:4 cfg4 --label default
MDCONV_FWD_TAIL=0 MDCONV_BW_SPLITS=29 python tools/exp.py cfg2:16 --label r03eq
MDCONV_FWD_TAIL=0 MDCONV_BW_SPLITS=58 python tools/exp.py cfg2:8 --label r03eq
MDCONV_FWD_TAIL=0 MDCONV_BW_SPLITS=14 python tools/exp.py cfg2:4 --label r03eq
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape_oracle.py -m gpu -x -q 2>&1 | tail -5
