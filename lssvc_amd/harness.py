"""test.py-compatible evaluation harness on top of the MI355X model classes.

    python -m lssvc_amd.harness --i_frame_model_path I.pth --model_path P.pth --test_config cfg.json \\
        --cuda 1 --worker 8 --output_path out [--write_stream 1 --stream_path out_bin] [--force_frame_num N]

Same command line, dataset config schema (`recommend_test_config.json`), sequencing and result files
(`{output_path}/{ratio}_{BL,EL,FL}.json`, keys of src/utils/common.py:25-37) as the reference's test.py, so its RD
scripts keep working. What is different, deliberately:
  * everything per frame runs on the GPU, as HIP kernels (csrc/prepost.hip via lssvc_amd.prepost): 4:2:0 -> RGB
    (test.py:185-186 does it with scipy on the host), zero padding, the MATLAB-bicubic base layer, the codec itself,
    clamping and the PSNR sums; only the raw 8-bit planes go up and a handful of scalars come back per frame;
  * work is sharded at GOP granularity, not per sequence (a GOP restarts from an I-frame with no carried state,
    test.py:219-227, so results are identical): with --worker 8 a single 96-frame sequence keeps 3 GPUs busy instead of
    1, and `--worker N` processes are pinned `process_idx % gpu_num` exactly like test.py:648-656;
  * MS-SSIM (pytorch_msssim, absent here and out of scope per SURVEY section 2) is not computed: every *_msssim field is
    written as JSON null, so a script that plots it fails loudly instead of silently consuming zeros;
  * encoder-side RDO and the MV / warp-frame / context PNG dumps are not built: their flags parse (command lines written
    for test.py keep working) but switching one on is an error.
There is no CPU mode: --cuda must be on.
"""
import argparse
import concurrent.futures
import json
import multiprocessing
import os
import time

import numpy as np
import torch

from . import preprocess
from .hip_ops import T
from .prepost import FramePrep, psnr_from_sum
from .shard import split_gops

_PREP = {}


def _prep(device):
    key = str(device)
    if key not in _PREP:
        _PREP[key] = FramePrep(device)
    return _PREP[key]

RATIO_FACTOR = {"x1_5": 1.5, "x2": 2.0, "x3": 3.0, "x4": 4.0}          # test.py:27-33
RATIO_LIST = ["x2", "x1_5"]                                            # test.py:681

RESULT_KEYS = ["i_frame_num", "p_frame_num",
               "ave_i_frame_bpp", "ave_i_frame_psnr", "ave_i_frame_rgb_psnr", "ave_i_frame_msssim", "ave_i_frame_rgb_msssim",
               "ave_i_frame_YUV_psnr",
               "ave_p_frame_bpp", "ave_p_frame_psnr", "ave_p_frame_rgb_psnr", "ave_p_frame_msssim", "ave_p_frame_rgb_msssim",
               "ave_p_frame_YUV_psnr",
               "ave_all_frame_bpp", "ave_all_frame_psnr", "ave_all_frame_rgb_psnr", "ave_all_frame_msssim",
               "ave_all_frame_rgb_msssim", "ave_all_frame_YUV_psnr", "encoding_time", "decoding_time"]


def str2bool(v):
    return str(v).lower() in ("yes", "y", "true", "t", "1")


def parse_args(argv=None):
    """test.py:36-85: the same flags. RDO / context-dump options are accepted but must stay off."""
    p = argparse.ArgumentParser(description="LSSVC evaluation on MI355X (test.py-compatible)")
    p.add_argument("--i_frame_model_name", type=str, default="IntraNoAR")
    p.add_argument("--i_frame_model_path", type=str, nargs="+", required=True)
    p.add_argument("--force_intra", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--force_frame_num", type=int, default=-1)
    p.add_argument("--force_intra_period", type=int, default=-1)
    p.add_argument("--intra_rdo", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--inter_mv_rdo", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--inter_feature_rdo", type=str2bool, nargs="?", const=True, default=False)
    # accepted for command-line compatibility (test.py:45-56); they only parameterise the encoder-side RDO, which is off
    p.add_argument("--intra_lmbda", type=float, nargs="+")
    p.add_argument("--intra_rdo_iter_to_exit", type=int, default=60)
    p.add_argument("--intra_rdo_iter_to_reduce", type=int, default=20)
    p.add_argument("--inter_lmbda", type=float, nargs="+")
    p.add_argument("--inter_mv_rdo_iter_to_exit", type=int, default=60)
    p.add_argument("--inter_mv_rdo_iter_to_reduce", type=int, default=20)
    p.add_argument("--inter_feature_rdo_iter_to_exit", type=int, default=60)
    p.add_argument("--inter_feature_rdo_iter_to_reduce", type=int, default=20)
    p.add_argument("--model_path", type=str, nargs="+")
    p.add_argument("--model_name", type=str, default="LSSVC_net")
    p.add_argument("--test_config", type=str, required=True)
    p.add_argument("--worker", "-w", type=int, default=1)
    p.add_argument("--cuda", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--cuda_device", default=None)
    p.add_argument("--write_stream", type=str2bool, nargs="?", const=True, default=False)
    p.add_argument("--stream_path", type=str, default="out_bin")
    p.add_argument("--save_decoded_frame", type=str2bool, default=False)
    p.add_argument("--decoded_frame_path", type=str, default="decoded_frames")
    # dump options of test.py:67-73 other than decoded frames: accepted, must stay off (MV / warp / context PNG dumps)
    p.add_argument("--save_decoded_mv", type=str2bool, default=False)
    p.add_argument("--save_warp_frame", type=str2bool, default=False)
    p.add_argument("--save_decoded_context", type=str2bool, default=False)
    p.add_argument("--decoded_mv_path", type=str, default="decoded_mv")
    p.add_argument("--warp_frame_path", type=str, default="warp_frame")
    p.add_argument("--decoded_context_path", type=str, default="decoded_context")
    p.add_argument("--decoding_profiling", type=str2bool, default=False)
    p.add_argument("--output_path", type=str, required=True)
    p.add_argument("--verbose", type=int, default=0)
    p.add_argument("--precision", type=str, default=None, choices=[None, "f16x3", "f32"], help="conv arithmetic (DESIGN.md 9)")
    args = p.parse_args(argv)
    if args.intra_rdo or args.inter_mv_rdo or args.inter_feature_rdo:
        p.error("encoder-side RDO is not part of the hot path this build covers")
    if args.save_decoded_mv or args.save_warp_frame or args.save_decoded_context:
        p.error("only --save_decoded_frame is supported (MV / warp-frame / context dumps are debugging aids outside the hot path)")
    if args.force_intra:
        args.model_path = args.i_frame_model_path
    if not args.model_path:
        p.error("--model_path is required unless --force_intra")
    if len(args.model_path) != len(args.i_frame_model_path):
        p.error("--i_frame_model_path and --model_path must list the same number of checkpoints")
    return args


# ------------------------------------------------------------------------------------------------- I/O + colour
class YUV420Reader:
    """8-bit planar 4:2:0 (`src/utils/video_reader.py:120-161`), with a start frame so a worker can open a GOP."""

    def __init__(self, path, width, height, start_frame=0):
        if not path.endswith(".yuv"):
            path += ".yuv"
        if width % 2 or height % 2:
            raise ValueError("4:2:0 needs even dimensions, got %dx%d" % (width, height))
        self.width, self.height = width, height
        self.frame_bytes = width * height * 3 // 2
        self.file = open(path, "rb")
        self.file.seek(start_frame * self.frame_bytes)

    def read(self):
        """-> (y (H,W), u (H/2,W/2), v (H/2,W/2)) uint8, or None at end of file."""
        raw = self.file.read(self.frame_bytes)
        if len(raw) < self.frame_bytes:
            return None
        a = np.frombuffer(raw, dtype=np.uint8).copy()
        n = self.width * self.height
        y = a[:n].reshape(self.height, self.width)
        u = a[n:n + n // 4].reshape(self.height // 2, self.width // 2)
        v = a[n + n // 4:].reshape(self.height // 2, self.width // 2)
        return y, u, v

    def close(self):
        self.file.close()


def _crop(x, pad):
    """F.pad(x, inverse_padding_size(p)) (test.py:251-252): drop the right/bottom padding."""
    return x[:, :, :x.shape[2] - pad[3], :x.shape[3] - pad[1]]


LOOKAHEAD = os.environ.get("LSSVC_LOOKAHEAD", "1") == "1"      # estimate mode: BL(t+1) beside EL(t) (code_frames)


# ------------------------------------------------------------------------------------------------- one GOP
def code_frames(i_net, p_net, reader, first_frame, n_frames, gop_size, ratio, device, bin_folder=None, png_folder=None):
    """test.py:182-311 for frames [first_frame, first_frame+n): returns one record per frame. `first_frame` must sit
    on a GOP boundary."""
    assert first_frame % gop_size == 0
    scale = RATIO_FACTOR[ratio]
    pad = preprocess.interlayer_padding(reader.height, reader.width, scale)
    (h_blp, w_blp), (h_elp, w_elp) = pad["LR_padded_size"], pad["HR_padded_size"]
    records, dpb = [], None
    if bin_folder is not None:
        for tag in ("BL", "EL"):
            os.makedirs(os.path.join(bin_folder, ratio, tag), exist_ok=True)
    prep = _prep(device)
    (h_bl, w_bl), (h_el, w_el) = pad["LR_size"], pad["HR_size"]
    def prepare(frame_idx):
        planes = reader.read()
        if planes is None:
            raise ValueError("sequence ends before frame %d" % frame_idx)
        # 8-bit planes up; colour conversion, padding, the bicubic base layer and the BL reference planes on the device
        # (csrc/prepost.hip): test.py:185-199
        y8, u8, v8 = (torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True) for a in planes)
        f_el, yuv_el = prep.frame_from_yuv420(y8, u8, v8, (h_elp, w_elp))
        f_bl = prep.bicubic(f_el, (h_blp, w_blp))
        return f_el, yuv_el, f_bl, prep.rgb_to_yuv420(f_bl, h_bl, w_bl), f_el.to_nchw(), f_bl.to_nchw()

    # estimate mode: a P-frame's call names the next P-frame's base-layer input, whose base layer is then coded beside this frame's
    # enhancement layer (LSSVC_extend.forward_one_frame; results unchanged) -- the loop prepares one frame ahead
    last = first_frame + n_frames - 1
    ahead = None
    for frame_idx in range(first_frame, first_frame + n_frames):
        f_el, (y_el, u_el, v_el), f_bl, (y_bl, u_bl, v_bl), x_el, x_bl = ahead if ahead is not None else prepare(frame_idx)
        ahead = None
        look = {}
        if LOOKAHEAD and bin_folder is None and p_net is not None and frame_idx % gop_size != 0:
            if frame_idx < last and (frame_idx + 1) % gop_size != 0:
                ahead = prepare(frame_idx + 1)
            look = dict(next_x_bl=(ahead[5] if ahead is not None else None), frame_id=frame_idx)
        i_net.set_scale_information(scale, (h_elp, w_elp), (0, 0, 0, 0))
        bins = (None, None)
        if bin_folder is not None:
            bins = (os.path.join(bin_folder, ratio, "BL", "%d.bin" % frame_idx), os.path.join(bin_folder, ratio, "EL", "%d.bin" % frame_idx))
        rec = {"frame": frame_idx, "enc_bl": 0.0, "dec_bl": 0.0, "enc_el": 0.0, "dec_el": 0.0}
        if frame_idx % gop_size == 0 or p_net is None:
            r = i_net.encode_decode(x_bl, x_el, bins[0], bins[1], pic_height_bl=h_blp, pic_width_bl=w_blp,
                                    pic_height_el=h_elp, pic_width_el=w_elp)
            dpb = {"ref_frame_bl": r["x_hat_bl"], "ref_frame_el": r["x_hat_el"], "ref_feature_bl": None,
                   "ref_feature_el": r["feature_el"]}
            rec["type"] = 0
        else:
            p_net.set_scale_information(scale, (h_elp, w_elp), (0, 0, 0, 0))
            r = p_net.encode_decode(x_bl, x_el, dpb, bins[0], bins[1], pic_width=w_elp, pic_height=h_elp,
                                    pic_width_bl=w_blp, pic_height_bl=h_blp, **look)
            dpb = r["dpb"]
            rec["type"] = 1
            rec.update(enc_bl=r.get("encoding_time_BL", 0.0), dec_bl=r.get("decoding_time_BL", 0.0),
                       enc_el=r.get("encoding_time_EL", 0.0), dec_el=r.get("decoding_time_EL", 0.0))
        rec["bits_bl"], rec["bits_el"] = float(r["bit_bl"]), float(r["bit_el"])
        # clamp the reconstructions in place (they are the next frame's references, test.py:249-250), then the eight
        # squared-error sums of the frame -- RGB and Y, U, V of both layers over the unpadded crop -- in one D2H read
        dpb["ref_frame_bl"].clamp_(0, 1)
        dpb["ref_frame_el"].clamp_(0, 1)
        hat = {"bl": T.from_nchw(dpb["ref_frame_bl"]), "el": T.from_nchw(dpb["ref_frame_el"])}
        prep.sqdiff_frames(hat["bl"], f_bl, h_bl, w_bl, 0)
        prep.sqdiff_frames(hat["el"], f_el, h_el, w_el, 1)
        for i, (tag, ref, (hh, ww)) in enumerate((("bl", (y_bl, u_bl, v_bl), (h_bl, w_bl)), ("el", (y_el, u_el, v_el), (h_el, w_el)))):
            for j, (got, want) in enumerate(zip(prep.rgb_to_yuv420(hat[tag], hh, ww), ref)):
                prep.sqdiff_planes(got, want, 2 + 3 * i + j)
        sq = prep.fetch()
        rec["rgb_psnr_bl"] = psnr_from_sum(sq[0], 3 * h_bl * w_bl) if sq[0] > 0 else float("inf")
        rec["rgb_psnr_el"] = psnr_from_sum(sq[1], 3 * h_el * w_el) if sq[1] > 0 else float("inf")
        for i, (tag, (hh, ww)) in enumerate((("bl", (h_bl, w_bl)), ("el", (h_el, w_el)))):
            n_y, n_c = hh * ww, (hh // 2) * (ww // 2)
            rec["yuv_" + tag] = (psnr_from_sum(sq[2 + 3 * i], n_y), psnr_from_sum(sq[3 + 3 * i], n_c), psnr_from_sum(sq[4 + 3 * i], n_c))
        hat_bl = _crop(dpb["ref_frame_bl"], pad["P_LR"])
        hat_el = _crop(dpb["ref_frame_el"], pad["P_HR"])
        if png_folder is not None:
            from PIL import Image
            for tag, hat in (("BL", hat_bl), ("EL", hat_el)):
                os.makedirs(os.path.join(png_folder, ratio, tag), exist_ok=True)
                img = hat[0].permute(1, 2, 0).mul(255).round_().clamp_(0, 255).byte().cpu().numpy()
                Image.fromarray(img).save(os.path.join(png_folder, ratio, tag, "%d.png" % frame_idx))
        records.append(rec)
    return records, pad


def aggregate(records, pix_bl, pix_el, test_time):
    """The three result dicts of run_test (test.py:329-535) from per-frame records (any order)."""
    records = sorted(records, key=lambda r: r["frame"])
    n = len(records)
    i_rec = [r for r in records if r["type"] == 0]
    p_rec = [r for r in records if r["type"] == 1]

    def layer(tag, pix, fl=False):
        def tot(rs, f):
            return float(sum(f(r) for r in rs))
        bits = (lambda r: r["bits_bl"] + r["bits_el"]) if fl else (lambda r: r["bits_" + tag])
        yuv = lambda r: (6 * r["yuv_" + tag][0] + r["yuv_" + tag][1] + r["yuv_" + tag][2]) / 8
        out = {"frame_pixel_num": pix, "i_frame_num": len(i_rec), "p_frame_num": len(p_rec), "frame_type": [r["type"] for r in records],
               "test_time": test_time}
        if not fl:
            out["frame_bpp"] = [r["bits_" + tag] / pix for r in records]
        for name, rs in (("i", i_rec), ("p", p_rec)):
            k = len(rs)
            out["ave_%s_frame_bpp" % name] = tot(rs, bits) / k / pix if k else 0
            out["ave_%s_frame_psnr" % name] = tot(rs, yuv) / k if k else 0
            out["ave_%s_frame_rgb_psnr" % name] = tot(rs, lambda r: r["rgb_psnr_" + tag]) / k if k else 0
            if not fl:
                out["ave_%s_frame_YUV_psnr" % name] = [tot(rs, lambda r, c=c: r["yuv_" + tag][c]) / k if k else 0 for c in range(3)]
            out["ave_%s_frame_msssim" % name] = None          # not computed (below): null, so RD scripts fail loudly
            out["ave_%s_frame_rgb_msssim" % name] = None
        out["ave_all_frame_bpp"] = tot(records, bits) / (n * pix)
        out["ave_all_frame_psnr"] = tot(records, yuv) / n
        out["ave_all_frame_rgb_psnr"] = tot(records, lambda r: r["rgb_psnr_" + tag]) / n
        if not fl:
            out["ave_all_frame_YUV_psnr"] = [tot(records, lambda r, c=c: r["yuv_" + tag][c]) / n for c in range(3)]
        out["ave_all_frame_msssim"] = None
        out["ave_all_frame_rgb_msssim"] = None
        kp = max(len(p_rec), 1)
        if fl:
            out["encoding_time"] = tot(p_rec, lambda r: r["enc_bl"] + r["enc_el"]) / kp
            out["decoding_time"] = tot(p_rec, lambda r: r["dec_bl"] + r["dec_el"]) / kp
        else:
            out["encoding_time"] = tot(p_rec, lambda r: r["enc_" + tag]) / kp
            out["decoding_time"] = tot(p_rec, lambda r: r["dec_" + tag]) / kp
        return out

    return layer("bl", pix_bl), layer("el", pix_el), layer("el", pix_el, fl=True)


def filter_dict(result):
    return {k: v for k, v in result.items() if k in RESULT_KEYS}


# ------------------------------------------------------------------------------------------------- jobs / workers
def build_jobs(args, config):
    """(dataset, ratio, sequence, model) units of test.py:682-743, cut further into GOPs."""
    jobs = []
    for ds_name, ds in config.items():
        if ds.get("test", 0) == 0:
            continue
        for ratio in RATIO_LIST:
            if ratio not in ds:
                continue
            for seq, info in ds["sequences"].items():
                for model_idx in range(len(args.model_path)):
                    gop = 1 if args.force_intra else (args.force_intra_period if args.force_intra_period > 0 else info["gop"])
                    frames = args.force_frame_num if args.force_frame_num > 0 else info["frames"]
                    for first, count in split_gops(frames, gop):
                        jobs.append({"ds_name": ds_name, "ratio": ratio, "seq": seq, "model_idx": model_idx, "gop": gop,
                                     "first": first, "count": count, "width": ds["x1"]["width"], "height": ds["x1"]["height"],
                                     "yuv": os.path.join(ds["base_path"], seq, "x1.yuv"),
                                     "i_path": args.i_frame_model_path[model_idx], "p_path": args.model_path[model_idx],
                                     "force_intra": args.force_intra, "write_stream": args.write_stream,
                                     "bin_folder": os.path.join(args.stream_path, seq, str(model_idx)) if args.write_stream else None,
                                     "png_folder": os.path.join("%s_%s_LSSVC" % (args.decoded_frame_path, args.i_frame_model_name), seq,
                                                                str(model_idx)) if args.save_decoded_frame else None,     # test.py:577-579,727-728
                                     "precision": args.precision})
    return jobs


_NETS = {}
_CKPT = {}          # path -> state dict: filled by the rank-0 broadcast in the torchrun mode, else loaded on demand


def _checkpoint(path):
    if path not in _CKPT:
        sd = torch.load(path, map_location="cpu")
        _CKPT[path] = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
    return _CKPT[path]


def _load_nets(job, device):
    """encode_one's model set-up (test.py:541-564), cached per worker process."""
    key = (job["i_path"], job["p_path"], job["force_intra"], job["write_stream"], str(device))
    if key not in _NETS:
        from . import IntraSS, LSSVC_extend, hip_ops
        if not _NETS:
            hip_ops.reserve_device_memory(torch.device(device))
        i_net = IntraSS.from_state_dict(_checkpoint(job["i_path"])).to(device).eval()
        p_net = None
        if not job["force_intra"]:
            p_net = LSSVC_extend()
            p_net.load_dict(_checkpoint(job["p_path"]))
            p_net = p_net.to(device).eval()
        if os.environ.get("LSSVC_GRAPH", "0") == "1" and not job["write_stream"]:
            i_net.set_graph_mode(True, alias_outputs=True)               # hipGraph frame plans (intra.FramePlan)
            if p_net is not None:
                p_net.set_graph_mode(True, alias_outputs=True)
        if job["write_stream"]:
            if p_net is not None:
                p_net.update(force=True)
            i_net.update(force=True)
        _NETS[key] = (i_net, p_net)
    return _NETS[key]


def run_job(job, device=None):
    """One GOP on this process's GPU."""
    if device is None:
        name = multiprocessing.current_process().name
        idx = int(name[name.rfind("-") + 1:]) if "-" in name and name[name.rfind("-") + 1:].isdigit() else 0
        device = "cuda:%d" % (idx % max(torch.cuda.device_count(), 1))          # test.py:648-656
    if job.get("precision"):
        from . import hip_ops
        hip_ops.set_conv_precision(job["precision"])
    i_net, p_net = _load_nets(job, device)
    reader = YUV420Reader(job["yuv"], job["width"], job["height"], start_frame=job["first"])
    t0 = time.time()
    try:
        with torch.no_grad():
            recs, pad = code_frames(i_net, p_net, reader, job["first"], job["count"], job["gop"], job["ratio"], device,
                                    bin_folder=job["bin_folder"], png_folder=job["png_folder"])
    finally:
        reader.close()
    torch.cuda.synchronize()
    return {"key": (job["ds_name"], job["ratio"], job["seq"], job["model_idx"]), "records": recs, "seconds": time.time() - t0,
            "pix_bl": pad["LR_size"][0] * pad["LR_size"][1], "pix_el": pad["HR_size"][0] * pad["HR_size"][1]}


def collect(args, config, job_results):
    """Merge GOP results per (dataset, ratio, sequence, model) and lay them out as test.py:756-789 writes them."""
    merged = {}
    for r in job_results:
        m = merged.setdefault(r["key"], {"records": [], "seconds": 0.0, "pix_bl": r["pix_bl"], "pix_el": r["pix_el"]})
        m["records"].extend(r["records"])
        m["seconds"] += r["seconds"]
    out = {}
    for ratio in RATIO_LIST:
        logs = ({}, {}, {})
        for ds_name, ds in config.items():
            if ds.get("test", 0) == 0:
                continue
            for lg in logs:
                lg[ds_name] = {}
            for seq in ds["sequences"]:
                for lg in logs:
                    lg[ds_name][seq] = {}
                for model_idx, model in enumerate(args.model_path):
                    m = merged.get((ds_name, ratio, seq, model_idx))
                    if m is None:
                        continue
                    frames = [r["frame"] for r in m["records"]]
                    if len(set(frames)) != len(frames):
                        raise RuntimeError("%s/%s: a frame was coded twice" % (ds_name, seq))
                    res = aggregate(m["records"], m["pix_bl"], m["pix_el"], m["seconds"])
                    for lg, one in zip(logs, res):
                        lg[ds_name][seq][os.path.basename(model)] = filter_dict(one)
        out[ratio] = logs
    return out


def main(argv=None):
    args = parse_args(argv)
    if not args.cuda:
        raise SystemExit("lssvc_amd has no CPU path: run with --cuda 1 on a ROCm device")
    if args.cuda_device:
        os.environ["HIP_VISIBLE_DEVICES"] = args.cuda_device
    with open(args.test_config) as f:
        config = json.load(f)
    jobs = build_jobs(args, config)
    begin = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # one process per GPU under `python -m torch.distributed.run --nproc-per-node N test.py ...`: GOP jobs are dealt
        # round-robin over the ranks, rank 0 alone reads the checkpoints and broadcasts them (RCCL), per-frame records are
        # gathered at the end, rank 0 writes the result files. No data-path collective.
        import torch.distributed as dist
        from .shard import broadcast_state_dicts, run_sharded
        local = int(os.environ.get("LOCAL_RANK", "0"))
        device = "cuda:%d" % local
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=torch.device(device))
        _CKPT.update(broadcast_state_dicts(list(args.i_frame_model_path) + list(args.model_path), dist, device))
        # a work queue, longest jobs first (round 6): frames x EL pixels is what a job costs
        results = run_sharded(jobs, lambda j: run_job(j, device=device), dist, cost=lambda j: j["count"] * j["width"] * j["height"])
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return None
    elif args.worker <= 1:
        results = [run_job(j, device="cuda:0") for j in jobs]                   # in-process: no child interpreter
    else:
        ctx = multiprocessing.get_context("spawn")                              # as test.py:676; before any GPU call here
        with concurrent.futures.ProcessPoolExecutor(max_workers=args.worker, mp_context=ctx) as pool:
            results = list(pool.map(run_job, jobs))
    logs = collect(args, config, results)
    os.makedirs(args.output_path, exist_ok=True)
    for ratio, (bl, el, fl) in logs.items():
        for tag, lg in (("BL", bl), ("EL", el), ("FL", fl)):
            with open(os.path.join(args.output_path, "%s_%s.json" % (ratio, tag)), "w") as fp:
                json.dump(lg, fp, indent=2)
    frames = sum(j["count"] for j in jobs)
    print("Test finished: %d frames in %d GOP jobs, %.1f s (%.2f frames/s)" % (frames, len(jobs), time.time() - begin,
                                                                               frames / max(time.time() - begin, 1e-9)))
    return logs


if __name__ == "__main__":
    main()
