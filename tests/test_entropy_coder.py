"""Host entropy coder (CPU-only): the product's C++ coder in liblssvc_hip.so against (a) the reference's own
pmf_to_quantized_cdf outputs (tests/golden/cdf_vectors.json, generated from the reference's ops.cpp), and
(b) the oracle's plain-C restatement of the reference rANS wrapper: identical bytes, exact round trips,
escapes, multi-call cursor, corrupt/short streams rejected."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oracle_rans():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/librans_oracle.so"], stdout=subprocess.DEVNULL)
    from lssvc_oracle import rans
    return rans


def test_cdf_quantiser_matches_reference(oracle_rans):
    from lssvc_amd.entropy_coder import pmf_to_quantized_cdf
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "cdf_vectors.json")))
    assert len(vec) >= 10
    for v in vec:
        assert [int(x) for x in pmf_to_quantized_cdf(v["pmf"], v["precision"])] == v["cdf"]
        assert [int(x) for x in oracle_rans.pmf_to_quantized_cdf(v["pmf"], v["precision"])] == v["cdf"]


def _tables(rng, n_tab=6):
    from lssvc_amd.entropy_coder import Tables
    lengths = rng.integers(3, 40, n_tab)
    pmf = rng.random((n_tab, 40)).astype(np.float32) ** 3
    for i in range(n_tab):
        pmf[i, lengths[i]:] = 0
        pmf[i] /= pmf[i].sum()
    tail = np.full((n_tab, 1), 1e-5, dtype=np.float32)
    return Tables.from_pmfs(pmf, tail, lengths, -rng.integers(0, 20, n_tab))


@pytest.mark.parametrize("seed,n,spread", [(0, 1, 3), (1, 7, 3), (2, 20000, 10), (3, 50000, 300)])
def test_bytes_equal_oracle_and_round_trip(oracle_rans, seed, n, spread):
    from lssvc_amd.entropy_coder import RansEncoder, RansDecoder
    rng = np.random.default_rng(seed)
    t = _tables(rng)
    idx = rng.integers(0, t.cdfs.shape[0], n).astype(np.int32)
    sym = rng.integers(-spread, spread + 30, n).astype(np.int32)        # includes symbols outside every table -> escapes
    cut = n // 3
    enc, ref = RansEncoder(), oracle_rans.Encoder()
    for e in (enc, ref):
        e.reset()
    enc.encode_with_indexes(sym[:cut], idx[:cut], t)
    enc.encode_with_indexes(sym[cut:], idx[cut:], t)
    ref.encode_with_indexes(sym[:cut], idx[:cut], t.cdfs, t.sizes, t.offsets)
    ref.encode_with_indexes(sym[cut:], idx[cut:], t.cdfs, t.sizes, t.offsets)
    data = enc.flush()
    assert data == ref.flush() and len(data) % 4 == 0
    dec = RansDecoder()
    dec.set_stream(data)
    half = n // 2
    got = np.concatenate([dec.decode_stream(idx[:half], t), dec.decode_stream(idx[half:], t)])
    assert np.array_equal(got, sym)
    rdec = oracle_rans.Decoder()
    rdec.set_stream(data)
    assert np.array_equal(rdec.decode_stream(idx, t.cdfs, t.sizes, t.offsets), sym)


def test_empty_and_error_paths(oracle_rans):
    from lssvc_amd.entropy_coder import RansEncoder, RansDecoder
    from lssvc_amd._lib import LssvcHipError
    rng = np.random.default_rng(9)
    t = _tables(rng)
    enc = RansEncoder()
    empty = enc.flush()
    assert len(empty) == 8 and empty == oracle_rans.Encoder().flush()     # just the flushed state
    with pytest.raises(LssvcHipError):
        enc.encode_with_indexes([0], [99], t)                              # index outside the tables
    dec = RansDecoder()
    with pytest.raises(LssvcHipError):
        dec.set_stream(b"\x00\x01\x02")                                    # not a rANS64 stream
    enc.reset()
    enc.encode_with_indexes(rng.integers(0, 10, 4000), rng.integers(0, 6, 4000), t)
    data = enc.flush()
    dec.set_stream(data[:len(data) // 2 // 4 * 4])
    with pytest.raises(LssvcHipError):
        dec.decode_stream(rng.integers(0, 6, 4000), t)                     # truncated stream is detected, not over-read


def test_cdf_tables_match_reference_update():
    """lssvc_amd.tables against the tables the REFERENCE's update(force=True) built for the same synthetic
    weights (tests/golden/cdf_tables.json from tests/golden/make_tables_golden.py): every row, length, offset."""
    import hashlib
    from lssvc_amd import tables
    from lssvc_amd.synth import synth_state_dict
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "cdf_tables.json")))
    sd_i = synth_state_dict("intra_ss", g["seed"], g["gain"])
    sd_p = synth_state_dict("lssvc_extend", g["seed"], g["gain"])
    got = {"el.bit_estimator_z": tables.bit_estimator_tables(sd_p, "bit_estimator_z"),
           "el.bit_estimator_z_mv": tables.bit_estimator_tables(sd_p, "bit_estimator_z_mv"),
           "bl.bit_estimator_z": tables.bit_estimator_tables(sd_p, "base_layer_model.bit_estimator_z"),
           "bl.bit_estimator_z_mv": tables.bit_estimator_tables(sd_p, "base_layer_model.bit_estimator_z_mv"),
           "laplace": tables.laplace_tables(), "gaussian": tables.gaussian_tables(),
           "intra.entropy_bottleneck": tables.bottleneck_tables(sd_i, "entropy_bottleneck")[0],
           "intra.bl.entropy_bottleneck": tables.bottleneck_tables(sd_i, "base_layer_model.entropy_bottleneck")[0]}
    for name, t in got.items():
        want = g[name]
        assert list(t.cdfs.shape) == want["shape"], name
        assert [int(v) for v in t.sizes] == want["lengths"] and [int(v) for v in t.offsets] == want["offsets"], name
        assert [int(v) for v in t.cdfs[0]] == want["row0"] and [int(v) for v in t.cdfs[-1]] == want["row_last"], name
        assert hashlib.sha1(t.cdfs.tobytes()).hexdigest() == want["sha1"], name


def test_stream_length_is_the_tables_ideal_code_length():
    """rANS codes within a whisker of -sum log2(freq / 2^16) of the tables it is given (+ the 64-bit final state)."""
    from lssvc_amd.entropy_coder import RansEncoder
    rng = np.random.default_rng(5)
    t = _tables(rng)
    n = 200000
    idx = rng.integers(0, t.cdfs.shape[0], n).astype(np.int32)
    sym = np.empty(n, np.int32)
    ideal = 0.0
    for i in range(t.cdfs.shape[0]):                  # draw in-table symbols from each table's own distribution
        sel = np.where(idx == i)[0]
        freq = np.diff(t.cdfs[i, :t.sizes[i]]).astype(np.float64)[:-1]          # last slot = escape, never drawn here
        v = rng.choice(freq.size, size=sel.size, p=freq / freq.sum())
        sym[sel] = v + t.offsets[i]
        ideal += -np.log2(freq[v] / 65536.0).sum()
    enc = RansEncoder()
    enc.encode_with_indexes(sym, idx, t)
    bits = len(enc.flush()) * 8
    assert ideal <= bits <= ideal * 1.0005 + 96, (bits, ideal)


@pytest.mark.parametrize("seed,n,spread", [(5, 9, 3), (6, 30000, 40), (7, 30000, 20000)])
def test_int16_planes_give_the_same_stream(seed, n, spread):
    """The 16-bit entry points (lssvc_rans_encode_with_indexes_i16 / lssvc_rans_decode_stream_i16: what the product path
    feeds from its pinned int16 staging buffer) write byte-identical streams to the int32 ones and decode them back, incl.
    escapes up to the edges of the 16-bit range; a decoded symbol that does not fit 16 bits is an error, not a wrap."""
    from lssvc_amd.entropy_coder import RansEncoder, RansDecoder
    from lssvc_amd._lib import LssvcHipError
    rng = np.random.default_rng(seed)
    t = _tables(rng)
    idx = rng.integers(0, t.cdfs.shape[0], n).astype(np.int32)
    sym = rng.integers(-spread, spread + 30, n).astype(np.int32)
    sym[:4] = (32767, -32768, 0, -1)[:min(4, n)]
    e32, e16 = RansEncoder(), RansEncoder()
    e32.reset(), e16.reset()
    e32.encode_with_indexes(sym, idx, t)
    e16.encode_with_indexes(sym.astype(np.int16), idx.astype(np.int16), t)
    data = e32.flush()
    assert data == e16.flush()
    dec = RansDecoder()
    dec.set_stream(data)
    out = np.empty(n, dtype=np.int16)
    got = dec.decode_stream(idx.astype(np.int16), t, out=out)
    assert got is out and got.dtype == np.int16 and np.array_equal(got.astype(np.int32), sym)
    # a symbol beyond 16 bits codes fine as int32 but must be refused by the 16-bit decoder
    e32.reset()
    big = np.array([40000], dtype=np.int32)
    e32.encode_with_indexes(big, idx[:1], t)
    dec.set_stream(e32.flush())
    with pytest.raises(LssvcHipError):
        dec.decode_stream(idx[:1].astype(np.int16), t)
