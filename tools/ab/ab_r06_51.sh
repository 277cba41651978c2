export MDCONV_QUIET=1
for a in "2d groups f16" "3d groups f16" "2d g1 small f32" "2d g1 small f16" "3d g1 small f32" "3d g1 small f16" "2d g1 large f32" "2d g1 large f16"; do
  f=gpurun_out/anom_$(echo $a | tr ' ' '_').txt
  timeout 700 python tools/anomaly_sweep.py $a 2>&1 | grep -v amdgpu.ids > $f
  echo "$a: $(grep -c . $f) lines, $(grep -c '<<<' $f) flagged"
done
