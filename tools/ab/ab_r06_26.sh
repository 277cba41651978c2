# the realistic-shape sweep and a second shape set, round-5 tree beside the current tree on ONE box
mkdir -p gpurun_out/ab26
for tree in _r5 .; do
  tag=$( [ $tree = . ] && echo r6 || echo r5 )
  (cd $tree && python tools/realistic_sweep.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/ab26/base_$tag.txt
  (cd $tree && python tools/realistic_sweep.py --more --dtypes f32,f16,bf16 2>&1 | grep -v amdgpu.ids) > gpurun_out/ab26/more_$tag.txt
done
python - <<'PY'
for what in ("base", "more"):
    a = [l.rstrip() for l in open(f"gpurun_out/ab26/{what}_r5.txt")]
    b = [l.rstrip() for l in open(f"gpurun_out/ab26/{what}_r6.txt")]
    print("==", what)
    for x, y in zip(a, b):
        kx, ky = x.split("DG=")[0], y.split("DG=")[0]
        mx = float(x.split(" ms")[0].split()[-1]); my = float(y.split(" ms")[0].split()[-1])
        flag = "  <<< slower" if my > 1.07 * mx else ("  faster" if my < 0.93 * mx else "")
        print("%s %8.3f -> %8.3f  %5.2f%s%s" % (x.split("  ")[0] if False else x[:62], mx, my, my / mx, flag, ("   " + y.split("<--")[1]) if "<--" in y else ""))
PY
