"""End-to-end throughput of the test.py-compatible harness at 1080p (what a user of the reference actually runs): a synthetic
1920x1080 4:2:0 clip and seeded checkpoints on disk, `--worker 1`, estimate mode, ratio x2, GOP 32; file reads, colour
conversion, padding, bicubic base layer, both codecs, PSNRs and the JSON result files all inside the clock.
    python tools/harness_bench.py [frames=64]         (LSSVC_GRAPH=1: hipGraph frame plans)"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import colour_torch_ref as CT  # noqa: E402
from lssvc_amd import harness as H
from lssvc_amd.synth import synth_clip, synth_state_dict


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    d = tempfile.mkdtemp(prefix="lssvc_hbench_")
    os.makedirs(os.path.join(d, "data", "seq0"))
    h, w, gop = 1080, 1920, 32
    clip = synth_clip(frames, h, w, seed=0).float() / 255.0
    with open(os.path.join(d, "data", "seq0", "x1.yuv"), "wb") as f:
        for t in range(frames):
            for p in CT.rgb_to_yuv420(clip[t:t + 1]):
                f.write(p.mul(255).round().clamp(0, 255).byte().numpy().tobytes())
    del clip
    torch.save(synth_state_dict("intra_ss", 0, 0.55), os.path.join(d, "i.pth"))
    torch.save(synth_state_dict("lssvc_extend", 0, 0.55), os.path.join(d, "p.pth"))

    def cfg(n):
        c = {"SYN": {"test": 1, "base_path": os.path.join(d, "data"), "x1": {"width": w, "height": h}, "x2": {"width": w // 2, "height": h // 2},
                     "sequences": {"seq0": {"frames": n, "gop": gop}}}}
        path = os.path.join(d, "cfg%d.json" % n)
        with open(path, "w") as f:
            json.dump(c, f)
        return path
    base = ["--i_frame_model_path", os.path.join(d, "i.pth"), "--model_path", os.path.join(d, "p.pth"), "--cuda", "1", "--worker", "1"]
    t0 = time.time()
    H.main(base + ["--test_config", cfg(min(frames, 34)), "--output_path", os.path.join(d, "warm")])     # loads the nets, warms the allocator / captures the plans
    t1 = time.time()
    H.main(base + ["--test_config", cfg(frames), "--output_path", os.path.join(d, "out")])
    torch.cuda.synchronize()
    t2 = time.time()
    r = json.load(open(os.path.join(d, "out", "x2_FL.json")))["SYN"]["seq0"]["p.pth"]
    print(json.dumps({"harness_frames_per_s": round(frames / (t2 - t1), 3), "frames": frames, "seconds": round(t2 - t1, 2), "first_call_seconds": round(t1 - t0, 2),
                      "graph": os.environ.get("LSSVC_GRAPH", "0"), "ave_all_frame_bpp": r.get("ave_all_frame_bpp"), "ave_all_frame_psnr": r.get("ave_all_frame_psnr")}))


if __name__ == "__main__":
    main()
