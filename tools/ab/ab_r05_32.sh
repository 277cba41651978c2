# r05 call 33: random-shape campaigns with fresh seeds on the final sources (after the 16-bit grad_bias rewrite of experiment 28;
# half of the shapes carry a bias): mid-size extents and the wide geometry (16-125 taps, strides / dilations to 3)
mkdir -p gpurun_out
timeout 520 python tools/fuzz_more.py --seconds 400 --first 30000 > gpurun_out/fuzz_r05_final.txt 2>&1
timeout 520 python tools/fuzz_more.py --seconds 400 --first 30000 --wide > gpurun_out/fuzz_r05_final_wide.txt 2>&1
tail -3 gpurun_out/fuzz_r05_final.txt gpurun_out/fuzz_r05_final_wide.txt
