for i in 1 2 3; do
MDCONV_BENCH_EXCHANGE=after MDCONV_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$i timeout 300 python bench.py --gpus 1 --scaling strong --no-cpu-baseline --no-other-configs --steps 20 > gpurun_out/after.out 2> gpurun_out/after.err; echo rc=$?
grep -v "alt_rsmi\|^$" gpurun_out/after.err | tail -3; cut -c1-700 gpurun_out/after.out
MDCONV_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$i timeout 300 python bench.py --gpus 1 --scaling strong --no-cpu-baseline --no-other-configs --steps 20 > gpurun_out/cap.out 2> gpurun_out/cap.err; echo rc=$?
grep -v "alt_rsmi\|^$" gpurun_out/cap.err | tail -3; cut -c1-700 gpurun_out/cap.out
done
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
