"""lssvc_amd -- MI355X-native (gfx950) engine for LSSVC's per-frame encode/decode hot path.

Public surface mirrors the reference's model API:
    from lssvc_amd import IntraSS, LSSVC_extend
Importing the model classes loads liblssvc_hip.so; a missing library is an ImportError, never a
silent CPU fallback. `lssvc_amd.synth` (synthetic checkpoints/clips) has no native dependency.
"""
import os as _os

# The frame plans keep up to nine streams busy (a frame's parallel chains, the next frame's base layer and its chains, the copy stream);
# the HIP runtime maps streams onto 4 hardware queues per process by default and streams that share a queue run one after the other.
# With 8 queues the bench's GOP is 0.6 ... 1.8 % faster (6: 3.7 % slower; 10 / 12 / 16: +1.5 / -0.3 / +0.7 %: profiles/r06_hw_queues_ab.txt);
# results do not depend on it. Read by the runtime when it initialises (the first HIP call of the process), so it is set here, at import,
# unless the caller has set it. (A property of THIS front end's streams: the C++ plan runtime, driven without Python, is 1.3 % slower with
# 8 queues than with the default and is left alone -- INTEGRATION.md.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def __getattr__(name):
    if name == "IntraSS":
        from .intra import IntraSS
        return IntraSS
    if name == "LSSVC_extend":
        from .inter import LSSVC_extend
        return LSSVC_extend
    raise AttributeError(name)
