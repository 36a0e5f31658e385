"""lssvc_amd -- MI355X-native (gfx950) engine for LSSVC's per-frame encode/decode hot path.

Public surface mirrors the reference's model API:
    from lssvc_amd import IntraSS, LSSVC_extend
Importing the model classes loads liblssvc_hip.so; a missing library is an ImportError, never a
silent CPU fallback. `lssvc_amd.synth` (synthetic checkpoints/clips) has no native dependency.
"""


def __getattr__(name):
    if name == "IntraSS":
        from .intra import IntraSS
        return IntraSS
    if name == "LSSVC_extend":
        from .inter import LSSVC_extend
        return LSSVC_extend
    raise AttributeError(name)
