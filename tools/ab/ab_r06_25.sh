# few-tile rule counted in rows of 8 output blocks: the regressed lines again, round-5 tree beside the current one
S="m2:f16:B8:C2048:O512:7x7:dg4 m2:f16:B8:C2048:O512:7x7:dg1 m2:f16:B16:C512:O512:7x7:dg4 m2:f16:B16:C512:O512:7x7:dg1 m2:f16:B8:C1024:O1024:7x7:dg4 m2:f16:B8:C1024:O1024:7x7:dg1 m3:f16:B4:C256:O256:4x7x7:dg1 m2:f16:B16:C256:O256:14x14:dg1 m2:f16:B8:C512:O512:14x14:dg1"
for tree in /root/repo/_r5 /root/repo; do
  echo "=== $tree"
  (cd $tree && python tools/prof_shape.py $S --n 20 2>&1 | grep -v amdgpu.ids)
done
timeout 900 python -m pytest tests/test_gpu_hp.py -m gpu -x -q -k "few_tile or route" 2>&1 | tail -3
