#!/usr/bin/env python3
"""Build libqdiff_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain
C-ABI shared object (include/qdiff_hip.h).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libqdiff_hip.so")
ARCH = "gfx950"
SOURCES = ["errors.cpp", "igemm_dma.hip", "quantize.hip", "norm_quant.hip", "attn_i8.hip", "bmm_i8.hip", "temb_mlp.hip", "fakequant.hip", "boxcal.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "qdiff_hip.h")]
# correctly-rounded fp32 division / sqrt are hipcc defaults; keep them explicit because the
# quantisers must reproduce torch's `round(x / delta)` bit for bit.
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-fno-fast-math", "-Wno-unused-value"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libqdiff_hip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + CFLAGS + ["-x", "hip", "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_variant(name, defines, only=None):
    """A second library with extra -D flags (A/B measurements: QDIFF_HIP_LIB=<path> selects it in qdiff/hip.py).
    only: the sources the flags concern — the others are linked from the main build's objects (build() first)."""
    objdir = os.path.join(HERE, "build", name)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        if only and src not in only:
            objs.append(_compile(src))
            continue
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if _stale(obj, [path] + HEADERS):
            r = subprocess.run([_hipcc()] + CFLAGS + [f"-D{d}" for d in defines] + ["-x", "hip", "-c", path, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        objs.append(obj)
    lib = os.path.join(LIBDIR, f"libqdiff_hip_{name}.so")
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


def build(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            path = os.path.join(OBJDIR, f)
            if os.path.isdir(path):                    # objects of a build_variant()
                shutil.rmtree(path)
            else:
                os.remove(path)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:                  # python build.py --variant occ3 [--only igemm_dma.hip] QD_MT1_OCC=3
        i = sys.argv.index("--variant")
        rest = sys.argv[i + 2:]
        only = None
        if rest and rest[0] == "--only":
            only, rest = rest[1].split(","), rest[2:]
        os.makedirs(OBJDIR, exist_ok=True)
        print(build_variant(sys.argv[i + 1], rest, only))
    else:
        print(build(force="--force" in sys.argv))
