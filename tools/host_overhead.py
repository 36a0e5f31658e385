"""Host-side cost per frame: code tiny frames (GPU work negligible) so the wall time is Python + ctypes + launch cost."""
import sys, time, torch
sys.path.insert(0, ".")
from lssvc_amd import IntraSS, LSSVC_extend
from lssvc_amd.synth import synth_state_dict, synth_clip
from lssvc_amd.preprocess import make_layers
dev = torch.device("cuda:0")
inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, 0.55)).to(dev).eval()
pnet = LSSVC_extend(); pnet.load_dict(synth_state_dict("lssvc_extend", 0, 0.55)); pnet.to(dev).eval()
clip = synth_clip(2, 128, 128).to(dev).float() / 255
xb0, xe0, pad = make_layers(clip[0:1], 2.0); xb1, xe1, _ = make_layers(clip[1:2], 2.0)
inet.set_scale_information(2.0, pad["HR_padded_size"], (0, 0, 0, 0)); pnet.set_scale_information(2.0, pad["HR_padded_size"], (0, 0, 0, 0))
with torch.no_grad():
    r = inet.encode_decode(xb0, xe0, None, None)
    dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None, "ref_feature_el": r["feature_el"]}
    for _ in range(3):
        out = pnet.encode_decode(xb1, xe1, dpb)
    torch.cuda.synchronize(); t0 = time.time()
    n = 20
    for _ in range(n):
        out = pnet.encode_decode(xb1, xe1, dpb)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    t0 = time.time()
    for _ in range(n):
        r = inet.encode_decode(xb0, xe0, None, None)
    torch.cuda.synchronize(); di = (time.time() - t0) / n
print("P-frame at 128x128: %.2f ms per frame (host-bound);  I-frame: %.2f ms" % (1e3 * dt, 1e3 * di))
