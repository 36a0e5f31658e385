"""Standalone check of the harness's multi-process path (not a pytest: the parent must not initialise the GPU before it
spawns workers). Builds a synthetic 4:2:0 clip + seeded checkpoints on disk, runs `--worker 2` (GOP jobs on spawned
workers) and then `--worker 1` (in-process) and compares the result files."""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import colour_torch_ref as CT  # noqa: E402
from lssvc_amd import harness as H
from lssvc_amd.synth import synth_clip, synth_state_dict

def main():
    d = tempfile.mkdtemp(prefix="lssvc_harness_")
    os.makedirs(os.path.join(d, "data", "seq0"))
    frames, gop, hw = 6, 2, 128
    clip = synth_clip(frames, hw, hw, seed=9).float() / 255.0
    with open(os.path.join(d, "data", "seq0", "x1.yuv"), "wb") as f:
        for t in range(frames):
            for p in CT.rgb_to_yuv420(clip[t:t + 1]):
                f.write(p.mul(255).round().clamp(0, 255).byte().numpy().tobytes())
    torch.save(synth_state_dict("intra_ss", 9, 0.6), os.path.join(d, "i.pth"))
    torch.save(synth_state_dict("lssvc_extend", 9, 0.6), os.path.join(d, "p.pth"))
    cfg = {"SYN": {"test": 1, "base_path": os.path.join(d, "data"), "x1": {"width": hw, "height": hw}, "x2": {"width": 64, "height": 64},
                   "sequences": {"seq0": {"frames": frames, "gop": gop}}}}
    with open(os.path.join(d, "cfg.json"), "w") as f:
        json.dump(cfg, f)
    base = ["--i_frame_model_path", os.path.join(d, "i.pth"), "--model_path", os.path.join(d, "p.pth"), "--test_config",
            os.path.join(d, "cfg.json"), "--cuda", "1"]
    H.main(base + ["--worker", "2", "--output_path", os.path.join(d, "out2")])          # spawns first: parent has no GPU context yet
    H.main(base + ["--worker", "1", "--output_path", os.path.join(d, "out1")])
    ok = True
    for tag in ("BL", "EL", "FL"):
        a = json.load(open(os.path.join(d, "out1", "x2_%s.json" % tag)))["SYN"]["seq0"]["p.pth"]
        b = json.load(open(os.path.join(d, "out2", "x2_%s.json" % tag)))["SYN"]["seq0"]["p.pth"]
        for k in a:
            if "time" in k:
                continue
            if a[k] != b[k]:
                ok = False
                print("MISMATCH", tag, k, a[k], b[k])
    print("harness selftest:", "OK (2 spawned workers == in-process, bit for bit)" if ok else "FAILED")
    return 0 if ok else 1

if __name__ == "__main__":
    sys.exit(main())
