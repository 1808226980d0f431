#!/usr/bin/env python3
"""Write a synthetic dataset tree in the layout NlosPoseDataset walks (utils/nlos_pose_dataloader.py:36-62):
    <root>/<pose>/<split>/{meas/<id>.hdr, vol/<id>.mat, joints/<id>.joints}
Real samples are 600 stacked 256 x 256 Radiance frames, a 256^3 occupancy volume and 24 joints in metres.
usage: make_synthetic_dataset.py <root> [samples per split = 2] [frame side = 256]"""
import os
import sys

import numpy as np
from scipy.io import savemat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ingest_oracle as io  # noqa: E402  (only its .hdr writer)
from hiddenpose_amd import testing as hpt  # noqa: E402


def main():
    root = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    side = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    g = np.random.Generator(np.random.PCG64(410))
    for split in ("train", "test"):
        base = os.path.join(root, "pose0", split)
        for sub in ("meas", "vol", "joints"):
            os.makedirs(os.path.join(base, sub), exist_ok=True)
        for k in range(n):
            sid = f"person_{split}_{k}"
            rgbe = hpt.synthetic_rgbe(600, side, side, seed=1000 * (split == "test") + k)
            with open(os.path.join(base, "meas", sid + ".hdr"), "wb") as f:
                f.write(io.rgbe_write(rgbe, rle=False))
            savemat(os.path.join(base, "vol", sid + ".mat"), {"vol": (g.random((256, side, side)) < 0.02).astype(np.float32)},
                    do_compression=True)
            np.savetxt(os.path.join(base, "joints", sid + ".joints"), g.random((24, 3)) * 0.6 - 0.3)
    print(f"wrote {2 * n} samples under {root}")


if __name__ == "__main__":
    main()
