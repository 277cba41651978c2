mkdir -p gpurun_out
rm -f gpurun_out/b21.txt
for v in default fng fni fnm fnall; do
L=""; [ $v != default ] && L=$PWD/modulated_deform_conv_amd/libmdconv_hip_$v.so
echo "== $v" >> gpurun_out/b21.txt
MDCONV_LIB=$L python - >> gpurun_out/b21.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
wl = bench.Workload("cfg5", "cuda")
for _ in range(3): wl.forward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): wl.forward()
e1.record(); torch.cuda.synchronize()
print("cfg5 fwd %.3f ms" % (e0.elapsed_time(e1) / 10))
PY
done
cat gpurun_out/b21.txt
