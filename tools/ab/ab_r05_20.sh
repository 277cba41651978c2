# r05 call 20: kernel timelines of one step (rocprofv3 --kernel-trace + tools/gap_report.py): cfg2 eager, cfg2 B=4 shard, cfg5, cfg4
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
{
for c in cfg2 cfg2:4 cfg5 cfg4 cfg3; do
  D=$ROOT/gpurun_out/trace_$c
  rm -rf $D; mkdir -p $D
  (cd /tmp && timeout 280 rocprofv3 --kernel-trace --output-format csv -d $D -o p -- python $ROOT/tools/exp.py $c --steps 6 > $D/log.txt 2>&1)
  echo "== $c"; grep step $D/log.txt
  f=$(ls $D/*kernel_trace.csv | head -1)
  case $c in cfg2*|cfg4) first=pack_weights;; *) first=hp_pack_fwd;; esac
  python tools/gap_report.py $f $first
done
} > gpurun_out/ab_r05_20.txt 2>&1
cat gpurun_out/ab_r05_20.txt
