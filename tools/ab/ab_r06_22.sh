# random-shape campaigns on the final sources, fresh seeds: original domain, wide geometry, the round-6 domain of hp_bwd3 (forced)
timeout 400 python tools/fuzz_more.py --seconds 300 --first 40000 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/fuzz_r06_a.txt; cat gpurun_out/fuzz_r06_a.txt | cut -c1-600
timeout 400 python tools/fuzz_more.py --seconds 300 --first 50000 --wide 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/fuzz_r06_b.txt; cat gpurun_out/fuzz_r06_b.txt | cut -c1-600
MDCONV_HP_BWD=4 timeout 400 python tools/fuzz_more.py --seconds 300 --first 60000 --dg 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/fuzz_r06_c.txt; cat gpurun_out/fuzz_r06_c.txt | cut -c1-600
