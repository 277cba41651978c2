bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
tail -2 gpurun_out/collect_r04.log | cut -c1-300
python tools/gap_report.py gpurun_out/prof_r04/*kernel_trace.csv 2>&1 | tail -20
bash tools/collect_extra.sh r04 > gpurun_out/collect_extra_r04.log 2>&1
tail -5 gpurun_out/collect_extra_r04.log
