# 16-bit backward with hp_bwd3: scatter lists before the fork (MDCONV_HP_CSR_FIRST=1) vs beside hp_gemm2 (shipped)
S="m2:f16:B32:C256:O256:56x56:dg1 m2:f16:B8:C256:O256:56x56:dg1 m2:f16:B64:C64:O64:56x56:dg1 m2:f16:B64:C128:O128:28x28:dg1 m2:f16:B16:C64:O64:56x56:dg4 m3:f16:B32:C64:O64:8x28x28:dg1"
for i in 1 2; do
echo "== beside"; python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids; python tools/exp.py cfg5 --label beside --steps 20 2>&1 | grep -v amdgpu.ids
echo "== first"; MDCONV_HP_CSR_FIRST=1 python tools/prof_shape.py $S 2>&1 | grep -v amdgpu.ids; MDCONV_HP_CSR_FIRST=1 python tools/exp.py cfg5 --label csr-first --steps 20 2>&1 | grep -v amdgpu.ids
done
