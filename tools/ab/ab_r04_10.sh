bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
tail -3 gpurun_out/collect_r04.log
python tools/gap_report.py gpurun_out/prof_r04/*kernel_trace.csv 2>&1 | tail -40
