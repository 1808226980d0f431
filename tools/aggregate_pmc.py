#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md
prescribes) of `bench.py --steps S --warmup W` into HBM bytes per training step per kernel family.
gfx950 correction: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streams -> x2; WRITE_SIZE exact.
usage: aggregate_pmc.py <fetch.csv> <write.csv> <steps+warmup> <out.csv> [<stamped.json>]
The optional last argument also writes the per-family totals with "_csrc_sha" = the sha of the kernel sources the
counters were collected on (bench.csrc_sha); bench.py quotes `roofline.traffic` from it only when the sha matches."""
import collections
import csv
import json
import os
import sys

csv.field_size_limit(1 << 30)

FAMILY = [("k_wgrad", "conv_wgrad"), ("k_stem_wgrad", "conv_wgrad"), ("k_stem_fwd", "conv_igemm(fwd+dgrad)"), ("k_stem_dgrad", "conv_stem_dgrad"), ("k_igemm", "conv_igemm(fwd+dgrad)"),
          ("k_bn_", "batchnorm"), ("k_stem_bwd", "stem_bn_pool"), ("k_bn_relu_pool3", "stem_bn_pool"), ("k_dconv3_wgrad", "dconv3_wgrad"), ("k_dconv3", "dconv3_fwd+dgrad"),
          ("k_stencil_c1", "dconv3_fwd+dgrad"), ("k_fold_replicate", "dconv3_fwd+dgrad"),
          ("k_maxpool3", "maxpool3"), ("k_axis_", "lct"), ("k_gn_", "groupnorm"), ("k_plane_stats", "groupnorm"), ("k_affine_relu", "groupnorm"),
          ("k_upsample", "upsample"), ("k_ups_", "upsample"), ("at::native", "aten(autograd adds, Adam)")]


def family(name):
    for key, fam in FAMILY:
        if key in name:
            return fam
    return "other_hip" if "hp::" in name else "runtime(memset/copy)"


def agg(path):
    d = collections.defaultdict(float)
    n = collections.defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            d[fam] += float(r["Counter_Value"])
            n[fam] += 1
    return d, n


fe, nfe = agg(sys.argv[1])
wr, _ = agg(sys.argv[2])
steps = int(sys.argv[3])
rows = []
for fam in fe:
    fetch_gb = fe[fam] * 1024 * 2 / 1e9 / steps   # KB -> B, x2 gfx950 correction
    write_gb = wr.get(fam, 0.0) * 1024 / 1e9 / steps
    rows.append((fam, nfe[fam] // steps, fetch_gb, write_gb, fetch_gb + write_gb))
rows.sort(key=lambda r: -r[4])
with open(sys.argv[4], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel_family", "launches_per_step", "hbm_fetch_GB_per_step(x2 corrected)", "hbm_write_GB_per_step",
                "hbm_total_GB_per_step"])
    for r in rows:
        w.writerow([r[0], r[1]] + [f"{v:.2f}" for v in r[2:]])
totals = {r[0]: round(r[4], 2) for r in rows}
json.dump(totals, open(sys.argv[4].replace(".csv", ".json"), "w"), indent=1)
if len(sys.argv) > 5:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    totals["_csrc_sha"] = bench.csrc_sha()
    json.dump(totals, open(sys.argv[5], "w"), indent=1)
print(open(sys.argv[4]).read())
