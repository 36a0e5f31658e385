"""Golden CDF tables from the REFERENCE's update(force=True) (build container only):

    make -C oracle && python tests/golden/make_tables_golden.py

The reference's `update()` imports its native modules; MLCodec_CXX is the reference's own ops.cpp built by
oracle/Makefile (oracle/_ref), MLCodec_rans (not buildable: rans64.h is not vendored) is replaced by an inert
stand-in because update() only instantiates the coder objects and never codes with them."""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
from ref_import import import_reference  # noqa: E402
from lssvc_amd.synth import synth_state_dict  # noqa: E402

IntraSS, LSSVC_extend = import_reference()
import MLCodec_CXX  # noqa: E402
stub = types.ModuleType("src.entropy_models.MLCodec_rans")
for n in ("RansEncoder", "RansDecoder", "BufferedRansEncoder"):
    setattr(stub, n, type(n, (), {}))
sys.modules["src.entropy_models.MLCodec_rans"] = stub
sys.modules["src.entropy_models.MLCodec_CXX"] = MLCodec_CXX

seed, gain = 7, 0.6
inet = IntraSS.from_state_dict(dict(synth_state_dict("intra_ss", seed, gain))).eval()
pnet = LSSVC_extend()
pnet.load_dict(synth_state_dict("lssvc_extend", seed, gain))
pnet.eval()
inet.update(force=True)
pnet.update(force=True)


def rec(cdf, length, offset):
    cdf = np.asarray(cdf, dtype=np.int32)
    return {"shape": list(cdf.shape), "sha1": hashlib.sha1(cdf.tobytes()).hexdigest(),
            "lengths": [int(v) for v in np.asarray(length).reshape(-1)], "offsets": [int(v) for v in np.asarray(offset).reshape(-1)],
            "row0": [int(v) for v in cdf[0]], "row_last": [int(v) for v in cdf[-1]]}


def helper(h):
    c, l, o = h.get_cdf_info_list()
    return rec(c, l, o)


def em(m):
    return rec(m._quantized_cdf.numpy(), m._cdf_length.numpy(), m._offset.numpy())


out = {"seed": seed, "gain": gain,
       "el.bit_estimator_z": helper(pnet.bit_estimator_z.cdf_helper), "el.bit_estimator_z_mv": helper(pnet.bit_estimator_z_mv.cdf_helper),
       "bl.bit_estimator_z": helper(pnet.base_layer_model.bit_estimator_z.cdf_helper),
       "bl.bit_estimator_z_mv": helper(pnet.base_layer_model.bit_estimator_z_mv.cdf_helper),
       "laplace": helper(pnet.gaussian_encoder.cdf_helper),
       "gaussian": em(inet.gaussian_conditional), "intra.entropy_bottleneck": em(inet.entropy_bottleneck),
       "intra.bl.entropy_bottleneck": em(inet.base_layer_model.entropy_bottleneck)}
import json
with open(os.path.join(HERE, "cdf_tables.json"), "w") as f:
    json.dump(out, f)
print({k: v["shape"] for k, v in out.items() if isinstance(v, dict)})
