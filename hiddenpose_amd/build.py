"""In-tree build of libhiddenpose_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m hiddenpose_amd.build [--force]
    python -m hiddenpose_amd.build --asan-host     # test-only: the HIP-free host sources under ASan + UBSan (see below)

Objects go to hiddenpose_amd/csrc/build/, the library next to this file so that it
travels with the source snapshot to the GPU box.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libhiddenpose_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
          "-Wall", "-Wno-unused-function", "-munsafe-fp-atomics"] + os.environ.get("HP_EXTRA_DEFS", "").split()   # A/B builds


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")))


def _stamp(src: str) -> str:
    h = hashlib.sha1()
    h.update(open(src, "rb").read())
    for hdr in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(open(hdr, "rb").read())
    h.update(" ".join(COMMON).encode())
    return h.hexdigest()


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    stamp_file = obj + ".sha1"
    stamp = _stamp(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [HIPCC] + COMMON
    if src.endswith(".cpp"):
        cmd += ["-ffp-contract=off", "-x", "hip"]  # host-only float32 constants must not be FMA-contracted
    elif "HP_BUILD: -ffp-contract=off" in open(src).read(2048):
        cmd += ["-ffp-contract=off"]  # kernels that reproduce NumPy float32 arithmetic bit for bit
    cmd += ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    open(stamp_file, "w").write(stamp)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread", "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)/1e6:.1f} MB) from {len(srcs)} sources")
    return LIB


# SURVEY section 5's safety net for the host code: the translation units that touch no device (the error string, the LCT
# constant builder and the Radiance container parser, which reads UNTRUSTED file bytes) compiled by the plain host compiler
# with AddressSanitizer + UndefinedBehaviorSanitizer into a test-only library.  tests/test_host_asan.py loads it in a child
# process (libasan preloaded) and runs the host ABI tests and a fuzz loop against it.  No device code, no HIP runtime.
HOST_ONLY = ["hp_error.cpp", "lct_host.cpp", "rgbe_host.cpp", "range_host.cpp"]
ASAN_LIB = os.path.join(OBJ, "libhiddenpose_host_asan.so")
HOSTCXX = os.environ.get("HOSTCXX", "g++")


def build_asan_host(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in HOST_ONLY]
    deps = srcs + sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    if not force and os.path.exists(ASAN_LIB) and os.path.getmtime(ASAN_LIB) >= max(os.path.getmtime(d) for d in deps):
        return ASAN_LIB
    cmd = [HOSTCXX, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"),
           "-I", CSRC, "-o", ASAN_LIB] + srcs + ["-lpthread", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host sanitizer build failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {ASAN_LIB} ({os.path.getsize(ASAN_LIB)/1e6:.1f} MB) from {len(srcs)} host-only sources")
    return ASAN_LIB


def sanitizer_runtime() -> str:
    """Paths of libasan / libubsan for LD_PRELOAD (a sanitized library in an unsanitized python needs its runtime first)."""
    out = []
    for name in ("libasan.so", "libubsan.so"):
        r = subprocess.run([HOSTCXX, f"-print-file-name={name}"], capture_output=True, text=True)
        path = r.stdout.strip()
        if r.returncode == 0 and os.path.isabs(path) and os.path.exists(path):
            out.append(os.path.realpath(path))
    return ":".join(out)


if __name__ == "__main__":
    if "--asan-host" in sys.argv:
        build_asan_host(force="--force" in sys.argv)
    else:
        build(force="--force" in sys.argv)
