#!/bin/bash
# Run ON THE GPU BOX: everything profiles/<tag>_other_configs.md / <tag>_hp_counters.md are made of
# (beyond tools/collect_profiles.sh <tag>).   usage: tools/collect_extra.sh r03
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r03}
D=$ROOT/gpurun_out/${TAG}_extra
rm -rf $D; mkdir -p $D
cd $ROOT
{
  echo "## other configurations, 1 GPU (tools/bench_configs.py)"
  python tools/bench_configs.py cfg3 cfg4 cfg5 2>&1 | grep -v amdgpu.ids
  echo "## cfg2 strong-scaling shards on 1 GPU: B = 32 / 16 / 8 / 4 (predicted ceiling of --scaling strong)"
  python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 2>&1 | grep -v amdgpu.ids
  echo "## the same shards replayed from a HIP graph (no host launch latency in the loop)"
  python tools/bench_configs.py cfg2 cfg2:16 cfg2:8 cfg2:4 --graph 2>&1 | grep -v amdgpu.ids
  echo "## 16-bit configurations with the round-2 kernels: MDCONV_HP_BWD=2 (fused tap-stationary backward), MDCONV_HP_C2I=1 (one-pass gather)"
  MDCONV_HP_BWD=2 MDCONV_HP_C2I=1 python tools/bench_configs.py cfg3 cfg5 2>&1 | grep -v amdgpu.ids
  echo "## same with the 16-bit path disabled (round-1 widen/narrow through the fp32 kernels)"
  MDCONV_HP=0 python tools/bench_configs.py cfg3 cfg5 2>&1 | grep -v amdgpu.ids
} > $D/configs.txt 2>&1
for c in cfg3 cfg4 cfg5; do bash tools/prof_cfg.sh $c > $D/stats_$c.txt 2>&1; done
for c in cfg3 cfg5; do
  bash tools/pmc_cfg.sh "FETCH_SIZE" $c hp_ > $D/fetch_$c.txt 2>&1
  bash tools/pmc_cfg.sh "WRITE_SIZE" $c hp_ > $D/write_$c.txt 2>&1
  bash tools/pmc_cfg.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $c hp_ > $D/sq_$c.txt 2>&1
  bash tools/pmc_cfg.sh "GRBM_GUI_ACTIVE GRBM_TA_BUSY" $c hp_ > $D/grbm_$c.txt 2>&1
done
bash tools/pmc_cfg.sh "FETCH_SIZE" cfg4 "" > $D/fetch_cfg4.txt 2>&1
bash tools/pmc_cfg.sh "WRITE_SIZE" cfg4 "" > $D/write_cfg4.txt 2>&1
bash tools/pmc_cfg.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" cfg4 "" > $D/sq_cfg4.txt 2>&1
bash tools/pmc_cfg.sh "GRBM_GUI_ACTIVE GRBM_TA_BUSY" cfg4 "" > $D/grbm_cfg4.txt 2>&1
# cache path: L1 (TCP) accesses per vector-memory instruction, L2 (TCC) hit rate
for c in cfg3 cfg4 cfg5; do
  bash tools/pmc_cfg.sh "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $c "" > $D/tcp_$c.txt 2>&1
  bash tools/pmc_cfg.sh "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" $c "" > $D/tcc_$c.txt 2>&1
  bash tools/pmc_cfg.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" $c "" > $D/vmem_$c.txt 2>&1
done
(cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -o fetchcal -- $ROOT/tools/ubench_fetch > $D/fetchcal.log 2>&1)
./tools/ubench_gather16 > $D/ubench_gather16.txt 2>&1
tail -3 $D/configs.txt
