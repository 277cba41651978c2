D=modulated_deform_conv_amd
MDCONV_BD_CL=0 python tools/exp.py cfg2 --label bdcl0-baseline --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BD_CL=0 MDCONV_LIB=$PWD/$D/libmdconv_hip_b1pure.so python tools/exp.py cfg2 --label b1pure --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BD_CL=0 MDCONV_BD_TPW=1 MDCONV_LIB=$PWD/$D/libmdconv_hip_b1pure.so python tools/exp.py cfg2 --label b1pure-tpw1 --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BD_CL=0 MDCONV_BD_PERCU=3 MDCONV_LIB=$PWD/$D/libmdconv_hip_b1pure3.so python tools/exp.py cfg2 --label b1pure3 --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BD_CL=0 MDCONV_BD_PERCU=3 MDCONV_BD_TPW=1 MDCONV_LIB=$PWD/$D/libmdconv_hip_b1pure3.so python tools/exp.py cfg2 --label b1pure3-tpw1 --steps 10 2>&1 | grep -v amdgpu.ids
MDCONV_BD_CL=0 MDCONV_BWD_FORK=0 MDCONV_LIB=$PWD/$D/libmdconv_hip_b1pure.so python tools/exp.py cfg2 --label b1pure-nofork --steps 10 2>&1 | grep -v amdgpu.ids
