#!/usr/bin/env python3
"""Compact per-kernel resource table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py hp_bwd [filter-substring]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modulated_deform_conv_amd", "csrc")
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-c",
                    os.path.join(csrc, src + ".hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                   capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = m.group(1)
for c in rows:
    n = re.sub(r"^void mdconv::\(anonymous namespace\)::", "", c["name"])
    n = re.sub(r"\(.*$", "", n)
    if flt in n:
        print("%-70s v%-4s a%-3s s%-3s scr%-5s occ%s lds%s" % (n[:70], c.get("vgpr"), c.get("agpr"), c.get("sgpr"), c.get("scratch"), c.get("occ"), c.get("lds")))
