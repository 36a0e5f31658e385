"""Bit-exact checks of the INTEGER work on the write_stream path (round-1 verdict "What's weak" 3):
lssvc_build_indexes, lssvc_export_symbols, lssvc_import_symbols on FIXED sigma / symbol tensors -- not downstream of a
float network -- against the oracle's restatement of the reference's build_indexes (video_entropy_models.py:309-313,
img_entropy_models.py:687-691) and of the 4-step fold (LSSVC_net.py:432-442). Equality is exact (np.array_equal)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sigma_cases(spec, n_random, seed):
    """sigma values that exercise every branch of the index map for one table flavour: non-positive and tiny values
    (the 1e-5 floor), values far below / above the table, every level boundary b_k = min * exp(k * step) nudged just
    inside and just outside (an off-by-one at a table edge moves these), the boundaries themselves, mid-levels, and a
    dense log-uniform random set."""
    lo, hi, levels = math.log(spec["min"]), math.log(spec["max"]), spec["levels"]
    step = (hi - lo) / (levels - 1)
    k = torch.arange(-2, levels + 2, dtype=torch.float64)
    edges = torch.exp(lo + k * step)
    parts = [torch.tensor([-3.0, -1e-6, 0.0, 1e-12, 1e-6, 9.9e-6, 1e-5, 1.1e-5, 1e-3, 1e4, 1e8, 3e38], dtype=torch.float64),
             edges * (1 - 2.0 ** -18), edges * (1 + 2.0 ** -18), edges * (1 - 2.0 ** -12), edges * (1 + 2.0 ** -12),
             torch.exp(lo + (k + 0.5) * step)]
    g = torch.Generator().manual_seed(seed)
    parts.append(torch.exp((torch.rand(n_random, generator=g, dtype=torch.float64) * (hi - lo + 6) + lo - 3)))
    return torch.cat(parts).float()


def _as_view(hip, flat, C_):
    """Pad a flat value list to an (H, W, C) NHWC tensor."""
    n = flat.numel()
    W = 16
    H = (n + W * C_ - 1) // (W * C_)
    H += H % 2                                    # even H, W for the 2x2 fold
    x = torch.ones(H * W * C_)
    x[:n] = flat
    x = x.view(1, H, W, C_).permute(0, 3, 1, 2).contiguous()          # NCHW tensor whose NHWC order is `flat`
    return x, hip.T.from_nchw(x.to(DEV))


@pytest.mark.parametrize("flavour", ["laplace", "gaussian"])
def test_build_indexes_bit_exact(flavour):
    from lssvc_amd import hip_ops as hip, tables
    from lssvc_amd._lib import lib, check
    from lssvc_oracle import entropy as E
    spec, add, fn = ((tables.LAPLACE, 0.0, E.laplace_indexes) if flavour == "laplace" else (tables.GAUSSIAN, 1.0, E.gaussian_indexes))
    sig = _sigma_cases(spec, 400_000, 11)
    x, t = _as_view(hip, sig, 32)
    want_nchw = fn(x)                                                 # int32 (1, C, H, W)
    lo, stp, add_, levels = tables.index_params(spec, add)
    # lssvc_build_indexes: NHWC plane
    out = torch.empty(t.H * t.W * t.C, dtype=torch.int32, device=DEV)
    check(lib.lssvc_build_indexes(t.ref, lo, stp, add_, levels, C.c_void_p(out.data_ptr()), hip.stream_ptr()))
    got = out.cpu().view(t.H, t.W, t.C).permute(2, 0, 1).numpy()
    assert np.array_equal(got, want_nchw[0].numpy()), "%d of %d indexes differ" % ((got != want_nchw[0].numpy()).sum(), got.size)
    assert got.min() == 0 and got.max() == levels - 1                 # both clamps were exercised
    # lssvc_export_symbols: the same map, flat NCHW
    _, idx = hip.export_symbols(None, t, (lo, stp, add_, levels))
    assert np.array_equal(idx, want_nchw.reshape(-1).numpy())


def test_export_import_symbols_bit_exact():
    """Symbols: int32(q) of an integer-valued fp32 plane incl. negative, zero, large (escape-range) values, flattened in
    NCHW order; import adds means / per-channel medians back in fp32 exactly as the reference's dequantiser does
    (symbols.float() + means, video_entropy_models.py:234-236; img_entropy_models.py:316-321)."""
    from lssvc_amd import hip_ops as hip
    g = torch.Generator().manual_seed(3)
    C_, H, W = 24, 10, 14
    q = torch.round(torch.randn(1, C_, H, W, generator=g) * 40)
    q[0, 0, 0, :6] = torch.tensor([0.0, -0.0, 70000.0, -70000.0, 32767.0, -32768.0])
    mean = torch.randn(1, C_, H, W, generator=g) * 3
    med = torch.randn(C_, generator=g)
    sym, idx = hip.export_symbols(hip.T.from_nchw(q.to(DEV)), None)
    assert sym.dtype == np.int32 and np.array_equal(sym, q.int().reshape(-1).numpy())
    assert np.array_equal(idx, np.repeat(np.arange(C_, dtype=np.int32), H * W))          # per-channel tables
    out = hip.T.zeros(H, W, C_, DEV)
    hip.import_symbols(sym, out, mean=hip.T.from_nchw(mean.to(DEV)))
    assert torch.equal(out.to_nchw().cpu(), q.int().float() + mean)
    out2 = hip.T.zeros(H, W, C_, DEV)
    hip.import_symbols(sym, out2, channel_add=med.to(DEV))
    assert torch.equal(out2.to_nchw().cpu(), q.int().float() + med.view(1, -1, 1, 1))


@pytest.mark.parametrize("step", [0, 1, 2, 3])
def test_four_step_fold_bit_exact(step):
    """The folded planes of spatial-prior step `step` (y_q_w_k / scales_w_k, LSSVC_net.py:432-442): output channel j at 2x2
    position m carries channel chunk CHUNK_OF_MASK[step][m]; symbols, Laplace indexes and the decoder-side unfold."""
    from lssvc_amd import hip_ops as hip, tables
    from lssvc_amd.inter import CHUNK_OF_MASK, MASK_OF_CHUNK
    from lssvc_oracle import entropy as E
    g = torch.Generator().manual_seed(20 + step)
    C_, H, W = 128, 12, 20
    q = torch.round(torch.randn(1, C_, H, W, generator=g) * 9)
    sig = _sigma_cases(tables.LAPLACE, C_ * H * W, 5 + step)[torch.randperm(C_ * H * W, generator=g)].view(1, C_, H, W)
    mean = torch.randn(1, C_, H, W, generator=g)
    fold_q, fold_s = torch.zeros(1, C_ // 4, H, W), torch.zeros(1, C_ // 4, H, W)
    pos = ((0, 0), (0, 1), (1, 0), (1, 1))
    for m, (r, c) in enumerate(pos):
        ch = CHUNK_OF_MASK[step][m]
        assert MASK_OF_CHUNK[step][ch] == m
        fold_q[:, :, r::2, c::2] = q[:, ch * 32:(ch + 1) * 32, r::2, c::2]
        fold_s[:, :, r::2, c::2] = sig[:, ch * 32:(ch + 1) * 32, r::2, c::2]
    lap = tables.index_params(tables.LAPLACE, 0.0)
    sym, idx = hip.export_symbols(hip.T.from_nchw(q.to(DEV)), hip.T.from_nchw(sig.to(DEV)), lap, chunk_of_mask=CHUNK_OF_MASK[step])
    assert np.array_equal(sym, fold_q.int().reshape(-1).numpy())
    assert np.array_equal(idx, E.laplace_indexes(fold_s).reshape(-1).numpy())
    # decoder side: only the coded chunk of each position is written, the rest of `out` is left alone
    out = hip.T.from_nchw(torch.full((1, C_, H, W), -7.0).to(DEV))
    hip.import_symbols(sym, out, mean=hip.T.from_nchw(mean.to(DEV)), chunk_of_mask=CHUNK_OF_MASK[step])
    want = torch.full((1, C_, H, W), -7.0)
    for m, (r, c) in enumerate(pos):
        ch = CHUNK_OF_MASK[step][m]
        sl = (slice(None), slice(ch * 32, (ch + 1) * 32), slice(r, None, 2), slice(c, None, 2))
        want[sl] = q[sl].int().float() + mean[sl]
    assert torch.equal(out.to_nchw().cpu(), want)


def test_staged_int16_planes_equal_the_int32_planes():
    """The product path's hand-off (hip_ops.SymbolStage: int16 planes written by lssvc_export_symbols_i16 into one device
    buffer, one asynchronous copy into pinned memory, lssvc_import_symbols_i16 on the way back) carries exactly the
    integers of the int32 planes, plain and folded; a symbol that does not fit 16 bits is refused loudly."""
    from lssvc_amd import hip_ops as hip, tables
    from lssvc_amd.inter import CHUNK_OF_MASK
    g = torch.Generator().manual_seed(11)
    C_, H, W = 128, 12, 20
    q = torch.round(torch.randn(1, C_, H, W, generator=g) * 30)
    q[0, 0, 0, :2] = torch.tensor([32767.0, -32768.0])
    sig = _sigma_cases(tables.LAPLACE, C_ * H * W, 2)[torch.randperm(C_ * H * W, generator=g)].view(1, C_, H, W)
    mean = torch.randn(1, C_, H, W, generator=g)
    lap = tables.index_params(tables.LAPLACE, 0.0)
    tq, ts, tm = hip.T.from_nchw(q.to(DEV)), hip.T.from_nchw(sig.to(DEV)), hip.T.from_nchw(mean.to(DEV))
    st = hip.SymbolStage(torch.device(DEV)).begin(8 * C_ * H * W)
    refs = [hip.export_symbols(tq, ts, lap, stage=st), hip.export_symbols(tq, None, stage=st)]
    refs += [hip.export_symbols(tq, ts, lap, chunk_of_mask=CHUNK_OF_MASK[s], stage=st) for s in range(4)]
    st.download(0, st.used)
    want = [hip.export_symbols(tq, ts, lap), hip.export_symbols(tq, None)]
    want += [hip.export_symbols(tq, ts, lap, chunk_of_mask=CHUNK_OF_MASK[s]) for s in range(4)]
    for (r_sym, r_idx), (w_sym, w_idx) in zip(refs, want):
        assert st.numpy(r_sym).dtype == np.int16
        assert np.array_equal(st.numpy(r_sym).astype(np.int32), w_sym) and np.array_equal(st.numpy(r_idx).astype(np.int32), w_idx)
    # back up: plain and folded, through the pinned buffer
    out16, out32 = hip.T.zeros(H, W, C_, DEV), hip.T.zeros(H, W, C_, DEV)
    hip.import_symbols(st.numpy(refs[0][0]), out16, mean=tm, stage=st)
    hip.import_symbols(want[0][0], out32, mean=tm)
    assert torch.equal(out16.to_nchw(), out32.to_nchw())
    for s in range(4):
        hip.import_symbols(st.numpy(refs[2 + s][0]), out16, mean=tm, chunk_of_mask=CHUNK_OF_MASK[s], stage=st)
        hip.import_symbols(want[2 + s][0], out32, mean=tm, chunk_of_mask=CHUNK_OF_MASK[s])
    assert torch.equal(out16.to_nchw(), out32.to_nchw())
    # out of range: flagged on the device, refused on the host
    q[0, 3, 2, 1] = 40000.0
    st.begin(8 * C_ * H * W)
    hip.export_symbols(hip.T.from_nchw(q.to(DEV)), None, stage=st)
    with pytest.raises(RuntimeError):
        st.download(0, st.used)
