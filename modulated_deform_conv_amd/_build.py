"""In-tree hipcc build of libmdconv_hip.so (gfx950 only).

``python -m modulated_deform_conv_amd._build`` or ``__graft_entry__.build()``.  The .so stays
next to this file (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmdconv_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=fast",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmdconv_hip.so")
    return exe


def _deps():
    return glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "mdconv.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


# mfma_bwd_data.hip: without SLP vectorisation.  Packed fp32 math is no faster beside MFMAs (v_pk_fma_f32 issues at
# 1.35-1.45x the v_fma_f32 time for twice the work, tools/ubench_pkfma.hip; GEMM-1 measures the same with the
# vectoriser on or off), and the vectoriser's register pairs keep hipcc from fusing the DPP
# operands of the drain's quad sums; the flag keeps that instance scratch-free at 253-256 VGPRs.
FILE_FLAGS = {"mfma_bwd_data.hip": ["-fno-slp-vectorize"]}


def _compile(src):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if _stale(obj, [src] + _deps()):
        cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, r.stderr
    return obj, ""


def build_variant(name, extra_flags):
    """Developer aid: build csrc with extra -D flags into libmdconv_hip_<name>.so (load it with
    MDCONV_LIB=<path>).  Used for ablation experiments only."""
    obj_dir = os.path.join(CSRC, "_obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        r = subprocess.run([_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + list(extra_flags) + ["-c", src, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        objs.append(obj)
    lib = os.path.join(HERE, "libmdconv_hip_%s.so" % name)
    subprocess.run([_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs, check=True)
    return lib


def build_files_variant(name, src_basenames, extra_flags):
    """Developer aid: libmdconv_hip_<name>.so = the default objects with the named sources recompiled (in parallel) under
    extra -D flags -- A/B builds that touch one kernel family; load it with MDCONV_LIB=<path>."""
    build()
    obj_dir = os.path.join(CSRC, "_obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)

    def one(b):
        obj = os.path.join(obj_dir, b + ".o")
        r = subprocess.run([_hipcc()] + FLAGS + FILE_FLAGS.get(b, []) + list(extra_flags) + ["-c", os.path.join(CSRC, b), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return b + ".o", obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        swapped = dict(ex.map(one, src_basenames))
    objs = [swapped.get(os.path.basename(o), o) for o in sorted(glob.glob(os.path.join(OBJ, "*.o")))]
    lib = os.path.join(HERE, "libmdconv_hip_%s.so" % name)
    subprocess.run([_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs, check=True)
    return lib


def build_one_file_variant(name, src_basename, extra_flags):
    return build_files_variant(name, [src_basename], extra_flags)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if force:
        for f in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(f)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            if log:
                print(log, file=sys.stderr)
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
