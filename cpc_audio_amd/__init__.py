"""cpc_audio_amd -- MI355X-native (gfx950) CPC-audio train-step hot path.

Drop-in replacements for the reference's hot-path modules (same class names, constructor
signatures, attributes and state-dict keys as facebookresearch/CPC_audio cpc/model.py and
cpc/criterion/criterion.py), backed by hand-written HIP kernels behind a C ABI
(include/cpc_hip.h, cpc_audio_amd/lib/libcpc_hip.so).  No CPU fallback.
"""
__version__ = "0.1.0"


def set_activation_storage(kind="fp32"):
    """Storage of the encoder's activations, saved tensors and gradient tensors (process-wide, cpc_set_mfma_mode):

    ``"fp32"`` (default)  the fp32-accurate path: results within 1e-4 of the reference's CPU path;
    ``"bf16"``            the bf16-storage variant of BASELINE.json configs[1]: y0..y3, xhat1..4 and the encoder's gradient
                          tensors as bf16, weights rounded to bf16 per step, one bf16 MFMA per product, fp32 accumulation and
                          ChannelNorm statistics; encoder output z, the GRU and the criterion stay fp32.  Parity bound (tests/
                          test_gpu_bf16.py): |z - z_ref| < 6e-2 (2e-2 relative), gradients within 6e-2 relative.
    """
    from . import _lib
    mode = {"fp32": _lib.DEFAULT_MFMA_MODE, "bf16": 4}[kind]
    _lib.get().check(_lib.get().cpc_set_mfma_mode(mode), "set_mfma_mode")
