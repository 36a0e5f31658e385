"""Generates tests/golden/oracle_gops/*.npz: the CPU oracle's closed-loop GOPs that tests/test_gpu_frames.py holds the GPU path to
(bits, PSNR and every quantised latent per frame). Run in the build container (CPU, minutes):
    python tests/golden/make_oracle_gops.py [--all] [--only substring]
Default: the configurations of the default GPU run; --all adds the ones behind --runslow. Existing files are kept.
The oracle itself is pinned to the reference by tests/test_oracle_golden.py; this script only runs it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from helpers import ORACLE_GOP_DIR, compute_oracle_gop, oracle_gop_name, save_oracle_gop  # noqa: E402
from lssvc_amd.preprocess import interlayer_padding  # noqa: E402


def dataset(ph, pw, scale, frames, seed):
    pad = interlayer_padding(ph, pw, scale)
    (H, W), bl = pad["HR_padded_size"], pad["LR_padded_size"]
    return (frames, H, W, seed, 0.55, scale, tuple(bl))


DEFAULT = [(32, 128, 128, 5, 0.55, 2.0, (64, 64)),             # test_gop_drift_vs_oracle
           (32, 128, 128, 7, 0.55, 2.0, (64, 64)),             # test_gop_drift_symbol_aware[7] (round 5: all 32 frames by default)
           (3, 384, 640, 3, 0.55, 2.0, (192, 320)),            # test_frames_384x640_vs_oracle
           dataset(240, 416, 2.0, 3, 7), dataset(240, 416, 2.0, 3, 8), dataset(240, 416, 2.0, 3, 9),
           dataset(240, 416, 1.5, 3, 7), dataset(240, 416, 1.5, 3, 8), dataset(240, 416, 1.5, 3, 9)]
SLOW = [(32, 128, 128, 8, 0.55, 2.0, (64, 64)), (32, 128, 128, 9, 0.55, 2.0, (64, 64)),
        dataset(480, 832, 2.0, 2, 7), dataset(480, 832, 2.0, 2, 8), dataset(480, 832, 2.0, 2, 9),
        dataset(720, 1280, 1.5, 2, 7), dataset(720, 1280, 1.5, 2, 8), dataset(720, 1280, 1.5, 2, 9)]


def main():
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", "6")))
    os.makedirs(ORACLE_GOP_DIR, exist_ok=True)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for cfg in DEFAULT + (SLOW if "--all" in sys.argv else []):
        name = oracle_gop_name(*cfg)
        path = os.path.join(ORACLE_GOP_DIR, name + ".npz")
        if (only and only not in name) or os.path.exists(path):
            continue
        t0 = time.time()
        _, _, rows = compute_oracle_gop(*cfg)
        save_oracle_gop(path, rows)
        print("%s: %d frames, %.0f s, %.1f KB" % (name, len(rows), time.time() - t0, os.path.getsize(path) / 1024), flush=True)


if __name__ == "__main__":
    main()
