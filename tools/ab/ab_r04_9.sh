python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python tools/exp.py cfg2 cfg3 cfg4 cfg5 --label now 2>&1 | grep -v amdgpu.ids
python tools/graph_check.py 2>&1 | grep -v amdgpu.ids
