"""In-tree build of libhiddenpose_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m hiddenpose_amd.build [--force]

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
          "-Wall", "-Wno-unused-function", "-munsafe-fp-atomics"]


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
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)/1e6:.1f} MB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
