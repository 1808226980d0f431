"""Memory / matrix-core instruction sequence of compiled kernels: where does a kernel wait for its loads?

    python tools/isa_sequence.py conv_kernels.hip 'k_igemmILi128ELb0ELb1ELi0' [max_tokens]

Compiles hiddenpose_amd/csrc/<file> for gfx950 to assembly (device only) and prints, per kernel whose mangled name matches
the regular expression, the run-length-encoded sequence of  LD (buffer/global load)  w(N) (s_waitcnt vmcnt(N))  M (MFMA)
dw / dr (LDS write / read)  B (s_barrier)  ST (store / atomic).  A software-pipelined loop shows up as `LD.. M.. B w(7) dw ..
w(0) dw`: loads before the MFMAs, staged waits after them.  `LD.. w(0) .. M..`, or `w(0)` followed by v_mov at the top of every
step, is a prefetch that is waited for on the spot (DESIGN 4.3 "what the ISA showed")."""
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], re.compile(sys.argv[2])
maxtok = int(sys.argv[3]) if len(sys.argv) > 3 else 120
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", "include", "-I", "hiddenpose_amd/csrc",
                    "-munsafe-fp-atomics", "-S", "--cuda-device-only", f"hiddenpose_amd/csrc/{src}", "-o", f.name],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(f.name).read().split("\n")
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if not m or not pat.search(m.group(1)):
        continue
    seq, prev, c, copies = [], None, 0, 0
    for j in range(i, len(lines)):
        t = lines[j].strip().split()
        if not t:
            continue
        if t[0] == "s_endpgm":
            break
        k = None
        if t[0].startswith(("buffer_load", "global_load")):
            k = "LD"
        elif t[0] == "s_waitcnt" and "vmcnt" in lines[j]:
            k = "w" + [x for x in t if "vmcnt" in x][0][5:]
            if "vmcnt(0)" in lines[j] and any("v_mov_b32" in x for x in lines[j + 1:j + 4]):
                copies += 1
        elif t[0].startswith("v_mfma"):
            k = "M"
        elif t[0].startswith("ds_write"):
            k = "dw"
        elif t[0].startswith("ds_read"):
            k = "dr"
        elif t[0] == "s_barrier":
            k = "B"
        elif t[0].startswith(("buffer_store", "global_store", "global_atomic", "buffer_atomic")):
            k = "ST"
        if k is None:
            continue
        if k == prev:
            c += 1
        else:
            if prev:
                seq.append(f"{prev}{c if c > 1 else ''}")
            prev, c = k, 1
    seq.append(f"{prev}{c if c > 1 else ''}")
    print(m.group(1)[:100], f"[{len(seq)} tokens, vmcnt(0)+v_mov: {copies}]")
    print("  " + " ".join(seq[:maxtok]))
