# profile collection of the round (dry run of the final collection): bench stats + PMC + per-config step traffic + extras
bash tools/collect_profiles.sh r06 2>&1 | tail -3
bash tools/collect_steps.sh r06 2>&1 | tail -5
bash tools/collect_extra.sh r06 2>&1 | tail -5
