"""Pin the harness row to the REFERENCE's own `run_test` (test.py:121-537): build container only.

    python tests/golden/make_harness_golden.py

A 4-frame 128x128 8-bit 4:2:0 clip (integer-exact synthetic planes, stored in the fixture) and seeded synthetic
checkpoints go through the reference's per-sequence harness -- its YUV reader, `ycbcr420_to_rgb` (scipy zoom),
padding, bicubic base layer, the I / P model calls, the in-place clamp, `rgb_to_ycbcr420`, the RGB and Y/U/V PSNRs and
the result aggregation -- exactly as test.py's `encode_one` would call it (estimate mode, GOP 4 = I P P P), and the three
result dicts are stored after the reference's own `filter_dict` (common.py:25-37), together with its colour-conversion
outputs for frame 0. MS-SSIM fields are not a parity metric here (pytorch_msssim is absent, the stand-in returns 0) and
times are machine-dependent: the consuming tests skip both.
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from ref_import import import_reference, REFERENCE_ROOT  # noqa: E402
from lssvc_amd.synth import synth_state_dict, synth_clip_exact  # noqa: E402

FRAMES, GOP, H, W, SEED, GAIN = 4, 4, 128, 128, 6, 0.6


def yuv_planes():
    """8-bit planes of the clip: channel 0 of the integer-exact synthetic clip as luma, channels 1 and 2 decimated by two
    as chroma (any 8-bit planes are a valid 4:2:0 picture)."""
    clip = synth_clip_exact(FRAMES, H, W, seed=SEED).numpy()
    return clip[:, 0], clip[:, 1, ::2, ::2].copy(), clip[:, 2, ::2, ::2].copy()


def main():
    IntraSS, LSSVC_extend = import_reference()
    spec = importlib.util.spec_from_file_location("lssvc_reference_test_py", os.path.join(REFERENCE_ROOT, "test.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)                      # the reference's harness module, unmodified
    from src.utils.functional import ycbcr420_to_rgb, rgb_to_ycbcr420
    from src.utils.common import filter_dict

    y, u, v = yuv_planes()
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "x1.yuv")
    with open(path, "wb") as f:
        for t in range(FRAMES):
            f.write(y[t].tobytes() + u[t].tobytes() + v[t].tobytes())
    inet = IntraSS.from_state_dict(dict(synth_state_dict("intra_ss", SEED, GAIN))).eval()
    pnet = LSSVC_extend()
    pnet.load_dict(synth_state_dict("lssvc_extend", SEED, GAIN))
    pnet.eval()
    args = {"frame_num": FRAMES, "gop_size": GOP, "write_stream": False, "ratio": "x2", "yuv_path_el": path,
            "x1": {"height": H, "width": W}, "video_path": "seq0", "bin_folder": None, "decoded_frame_folder": None}
    torch.manual_seed(0)
    torch.set_num_threads(8)
    bl, el, fl = ref.run_test(pnet, inet, args, torch.device("cpu"))
    out = {"meta": {"frames": FRAMES, "gop": GOP, "height": H, "width": W, "seed": SEED, "gain": GAIN, "ratio": "x2"},
           "BL": filter_dict(bl), "EL": filter_dict(el), "FL": filter_dict(fl),
           "frame_bpp": {"BL": [float(b) for b in bl["frame_bpp"]], "EL": [float(b) for b in el["frame_bpp"]]},
           "frame_type": [int(t) for t in el["frame_type"]]}
    with open(os.path.join(HERE, "harness_x2.json"), "w") as f:
        json.dump(out, f, indent=1, default=float)          # numpy scalars (the Y/U/V PSNRs are np.float64 / float32)
    # the reference's colour conversions on frame 0 (functional.py:16-58)
    y0 = y[0:1].astype(np.float32) / 255
    uv0 = np.stack([u[0], v[0]]).astype(np.float32) / 255
    rgb0 = ycbcr420_to_rgb(y0, uv0)
    y_back, uv_back = rgb_to_ycbcr420(rgb0)
    np.savez_compressed(os.path.join(HERE, "harness_x2_clip.npz"), y=y, u=u, v=v, rgb0=rgb0.astype(np.float32),
                        y_back=y_back.astype(np.float32), uv_back=uv_back.astype(np.float32))
    print(json.dumps(out["EL"], indent=1, default=float))
    print("frame bpp", out["frame_bpp"])


if __name__ == "__main__":
    main()
