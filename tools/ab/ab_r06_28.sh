# after the few-tile forward rule and the spilling-instance crossover: hp tests, the crossover shapes on the default path, both sweeps vs round 5
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_hp_forced.py -m gpu -x -q 2>&1 | tail -3
S="m3:f16:B1:C256:O256:4x14x14 m3:f16:B2:C256:O256:4x14x14 m3:f16:B4:C256:O256:4x14x14 m3:f16:B8:C256:O256:4x14x14 m3:f16:B2:C128:O256:4x14x14 m3:f16:B8:C128:O256:4x14x14 m3:f16:B4:C64:O256:8x14x14 m2:f16:B4:C256:O256:14x14 m2:f16:B16:C256:O256:14x14 m2:f16:B4:C256:O256:28x28 m2:f16:B1:C256:O256:56x56 m2:f16:B4:C128:O256:28x28 m2:f16:B16:C128:O256:28x28"
python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids
bash tools/ab/ab_r06_26.sh > /dev/null 2>&1
