# group-padded layout: random shapes (2 / 3 / 4 groups of 8 ... 120 channels), default kernels and hp_bwd3 forced; the general campaigns again
timeout 260 python tools/fuzz_more.py --seconds 200 --first 90000 --pad 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
MDCONV_HP_BWD=4 timeout 200 python tools/fuzz_more.py --seconds 140 --first 91000 --pad 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
MDCONV_HP_BWD=2 timeout 200 python tools/fuzz_more.py --seconds 100 --first 92000 --pad 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
timeout 200 python tools/fuzz_more.py --seconds 100 --first 93000 --wide 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
