"""cpc_audio_amd -- MI355X-native (gfx950) CPC-audio train-step hot path.

Drop-in replacements for the reference's hot-path modules (same class names, constructor
signatures, attributes and state-dict keys as facebookresearch/CPC_audio cpc/model.py and
cpc/criterion/criterion.py), backed by hand-written HIP kernels behind a C ABI
(include/cpc_hip.h, cpc_audio_amd/lib/libcpc_hip.so).  No CPU fallback.
"""
__version__ = "0.1.0"
