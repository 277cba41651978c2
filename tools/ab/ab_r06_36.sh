# random-shape campaigns on the final sources of the round, fresh seeds (fp32: matrix path vs generic path, every 4th vs the oracle; 16-bit vs the oracle)
mkdir -p gpurun_out
timeout 400 python tools/fuzz_more.py --seconds 300 --first 110000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 > gpurun_out/fuzz_r06_final_a.txt
timeout 400 python tools/fuzz_more.py --seconds 300 --first 120000 --wide 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 > gpurun_out/fuzz_r06_final_b.txt
MDCONV_HP_BWD=4 timeout 300 python tools/fuzz_more.py --seconds 200 --first 130000 --dg 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 > gpurun_out/fuzz_r06_final_c.txt
timeout 300 python tools/fuzz_more.py --seconds 200 --first 140000 --pad 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 > gpurun_out/fuzz_r06_final_d.txt
MDCONV_DG_PLAN=split timeout 200 python tools/fuzz_more.py --seconds 120 --first 150000 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 > gpurun_out/fuzz_r06_final_e.txt
cat gpurun_out/fuzz_r06_final_?.txt
