# fresh-seed campaigns on the final sources of the round (36a19caa0e290c1a)
timeout 200 python tools/fuzz_more.py --seconds 120 --first 300000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
timeout 200 python tools/fuzz_more.py --seconds 120 --first 310000 --wide 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
MDCONV_HP_BWD=4 timeout 200 python tools/fuzz_more.py --seconds 100 --first 320000 --dg 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
timeout 200 python tools/fuzz_more.py --seconds 100 --first 330000 --pad 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
MDCONV_PAD_CHANNELS=1 timeout 200 python tools/fuzz_more.py --seconds 100 --first 340000 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-420
