"""Stand-ins for the reference's two pybind11 extension modules, built on liblssvc_hip.so's C ABI:

    lssvc_amd.compat.MLCodec_rans   (src/cpp/rans/rans_interface.cpp:246-261 + the names the Python calls)
    lssvc_amd.compat.MLCodec_CXX    (src/cpp/ops/ops.cpp:84-91)

A maintainer of the reference drops them in as `src/entropy_models/MLCodec_rans.py` / `MLCodec_CXX.py`
(INTEGRATION.md section 2), or calls install() to register them under the names the reference imports."""
import sys


def install(package="src.entropy_models"):
    """Register the stand-ins in sys.modules as `<package>.MLCodec_rans` / `<package>.MLCodec_CXX` (and bare
    `MLCodec_rans` / `MLCodec_CXX`), so `from .MLCodec_rans import BufferedRansEncoder, RansDecoder`
    (video_entropy_models.py:12, img_entropy_models.py:19, priors.py:627,700) resolves to them."""
    from . import MLCodec_CXX, MLCodec_rans
    for name, mod in (("MLCodec_rans", MLCodec_rans), ("MLCodec_CXX", MLCodec_CXX)):
        sys.modules[name] = mod
        if package:
            sys.modules[package + "." + name] = mod
    return MLCodec_rans, MLCodec_CXX
