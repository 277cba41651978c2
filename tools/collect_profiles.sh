#!/bin/bash
# Run ON THE GPU BOX (via gpurun): collects everything profiles/rNN_* is made of into
# gpurun_out/prof_<tag>/ -- kernel stats, separate FETCH/WRITE PMC passes, the un-profiled bench
# line (with cpu_baseline) and the SQ counters MfmaUtil is derived from.
#   usage: tools/collect_profiles.sh <tag>     then (locally) python tools/summarize_profile.py gpurun_out/prof_<tag> <tag>
set -u
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$ROOT/gpurun_out/prof_$1
rm -rf $D; mkdir -p $D
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_sources_sha16())" > $D/kernel_sources_sha16.txt
cd /tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --sustain-s 0 > $D/bench_stdout.txt 2>&1
timeout 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -o pmc_fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustain-s 0 > /dev/null 2>&1
timeout 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D -o pmc_write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --sustain-s 0 > /dev/null 2>&1
cd $ROOT
timeout 400 python bench.py > $D/bench_full.json 2> $D/bench_full.err
timeout 200 python bench.py --graph --no-cpu-baseline --no-other-configs > $D/bench_graph.json 2>> $D/bench_full.err
MDCONV_BWD_FORK=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs > $D/bench_nofork.json 2>> $D/bench_full.err
{
  bash tools/pmc_kernel.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" sq1
  bash tools/pmc_kernel.sh "GRBM_GUI_ACTIVE GRBM_TA_BUSY" sq2
} > $D/sq_counters.txt 2>&1
tail -c 600 $D/bench_full.json
