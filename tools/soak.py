import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from lssvc_amd import IntraSS, LSSVC_extend
from lssvc_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
inet = IntraSS.from_state_dict(synth_state_dict("intra_ss", 0, bench.GAIN)).to(dev).eval()
pnet = LSSVC_extend(); pnet.load_dict(synth_state_dict("lssvc_extend", 0, bench.GAIN)); pnet.to(dev).eval()
x_bls, x_els, pad = bench.build_inputs(dev, 0, 32)
ref = None
with torch.no_grad():
    for k in range(6):
        torch.cuda.synchronize(); t0 = time.time()
        bits, dpb = bench.encode_gop(inet, pnet, x_bls, x_els, pad["HR_padded_size"])
        torch.cuda.synchronize(); dt = time.time() - t0
        tot = sum(b[0] + b[1] for b in bits)
        if ref is None: ref = (tot, dpb["ref_frame_el"].clone())
        same = (tot == ref[0]) and torch.equal(dpb["ref_frame_el"], ref[1])
        print("GOP %d: %.3f s  %.2f fps  bits %.3f  identical_to_first=%s  mem alloc %.2f GB reserved %.2f GB" % (
            k, dt, 32 / dt, tot, same, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
