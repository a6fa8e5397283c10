"""ctypes binding of the C ABI declared in include/cpc_hip.h.

The product path loads exactly one library: ``cpc_audio_amd/lib/libcpc_hip.so``, built
by hipcc for gfx950 (``python -m cpc_audio_amd.build``).  If it is missing or fails to
load, every op raises -- there is NO CPU / eager fallback.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcpc_hip.so")
DEFAULT_GRU_MODE = 2       # cpc_set_gru_mode: persistent recurrence, forward products on the fp16 split
DEFAULT_GRU_XCD_LOCAL = 1  # cpc_set_gru_xcd_local: forward hand-over through one XCD's L2 for batches of at most half the device (round 6)
DEFAULT_GRU_POLL_PLAIN = 15  # cpc_set_gru_poll_plain: every first look of the persistent recurrence through the XCD's L2 (round 6)
DEFAULT_MFMA_MODE = 3      # what libcpc_hip starts in (cpc_set_mfma_mode): mode 2's arithmetic (two fp16 pieces, 3 MFMAs per
#                            product) with conv1 / its gradients on the DMA-fed kernels reading H2-stored activations
EXPECTED_ABI = 15          # cpc_abi_version() of the library these signatures were written for: a stale or variant build that
#                            exports every symbol with OLDER argument lists would corrupt memory instead of raising -- bind() refuses it
DEFAULT_DMA_PIPELINE = 2       # cpc_set_dma_pipeline: the tap-pair walk where the shape allows, two 32-k stages elsewhere
DEFAULT_WGRAD_DMA_STAGES = 4   # cpc_set_wgrad_dma_stages
DEFAULT_CONV_SMALL_PIPE = 1    # cpc_set_conv_small_pipe
DEFAULT_GRU_WGRAD_STREAM = 1   # cpc_set_gru_wgrad_stream: the recurrence's weight gradients on the preparation stream (single-rank composite steps)
DEFAULT_NCE_FUSED = 2          # cpc_set_nce_fused: the one-pass criterion with its scoring kernel on fp16 pieces (round 6; 1 = exact-f32 MFMAs)
DEFAULT_INDEX_PREP_GROUPS = -1 # cpc_set_index_prep_groups: one workgroup per CU and launch of the criterion's index preparation
DEFAULT_FWD_NSPLIT = 0         # cpc_set_fwd_nsplit (built and measured in round 5: no gain at B = 64, off)
DEFAULT_DGRAD_NSPLIT = 256     # cpc_set_dgrad_nsplit: the short layers' data gradients on 128 x 128 tiles where that gives >= 256 workgroups
DEFAULT_STEP_SCHEDULE = (1, 0)  # cpc_set_step_schedule: index preparation behind conv0, dz path beside the recurrence

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float

# name -> (restype, argtypes); mirrors include/cpc_hip.h one to one
SIGNATURES = {
    "cpc_abi_version": (_I, []),
    "cpc_release_stream": (_I, [_P]),
    "cpc_streams_overlap": (_I, [_P, _P, _P]),
    "cpc_set_mfma_mode": (_I, [_I]),
    "cpc_get_mfma_mode": (_I, []),
    "cpc_device_error_flags": (_I, [_I]),
    "cpc_conv0_forward": (_I, [_P] * 8 + [_I, _I, _P]),
    "cpc_conv0_forward_h2": (_I, [_P] * 9 + [_I, _I, _P]),
    "cpc_h2_encode": (_I, [_P, _P, _L, _P, _P]),
    "cpc_h2_decode": (_I, [_P, _P, _L, _P, _P]),
    "cpc_conv_weight_relayout_h2": (_I, [_P, _P, _I, _P]),
    "cpc_conv_gemm_forward_h2": (_I, [_P] * 11 + [_I] * 6 + [_P]),
    "cpc_set_dma_tile": (_I, [_I]),
    "cpc_set_conv0_tuning": (_I, [_I, _I]),
    "cpc_set_h2_layers": (_I, [_I]),
    "cpc_set_h2_dx": (_I, [_I]),
    "cpc_set_wgrad1_early": (_I, [_I]),
    "cpc_set_dma_layer2": (_I, [_I]),
    "cpc_set_conv_small_tile": (_I, [_I]),
    "cpc_set_dgrad_nsplit": (_I, [_I]),
    "cpc_set_conv_small_pipe": (_I, [_I]),
    "cpc_set_wgrad_dma_groups": (_I, [_I]),
    "cpc_set_wgrad_dma_stages": (_I, [_I]),
    "cpc_set_wgrad_dma_min_rows": (_I, [_I]),
    "cpc_set_gemm_split": (_I, [_I]),
    "cpc_set_gemm_dma": (_I, [_I]),
    "cpc_set_attn_fwd": (_I, [_I]),
    "cpc_set_gemm_tail_cus": (_I, [_I]),
    "cpc_set_dma_wave_rows": (_I, [_I]),
    "cpc_set_gemm_fuse": (_I, [_I]),
    "cpc_set_gru_xcd_pack": (_I, [_I]),
    "cpc_set_gru_poll_plain": (_I, [_I]),
    "cpc_set_gru_xcd_local": (_I, [_I]),
    "cpc_set_gru_chunk_tiles": (_I, [_I]),
    "cpc_set_gru_tiles_per_wg": (_I, [_I]),
    "cpc_set_gru_poll_pacing": (_I, [_I, _I]),
    "cpc_set_dma_rotation": (_I, [_I]),
    "cpc_set_dma_pipeline": (_I, [_I]),
    "cpc_conv0_backward_scratch_floats": (_L, [_I, _I]),
    "cpc_conv0_backward": (_I, [_P] * 13 + [_I, _I, _P]),
    "cpc_conv_layer_forward": (_I, [_P] * 9 + [_I] * 5 + [_P]),
    "cpc_conv_weight_relayout": (_I, [_P, _P, _I, _P]),
    "cpc_conv_gemm_forward": (_I, [_P] * 9 + [_I] * 5 + [_P]),
    "cpc_absmax": (_I, [_P, ctypes.c_long, _P, _P]),
    "cpc_norm_backward": (_I, [_P] * 10 + [_I, _P]),
    "cpc_conv_layer_dgrad": (_I, [_P] * 3 + [_I] + [_P] * 10 + [_I] * 5 + [_P]),
    "cpc_conv_layer_wgrad": (_I, [_P] * 6 + [_I] * 7 + [_P]),
    "cpc_encoder_layout": (_I, [_I, _I, _P]),
    "cpc_encoder_saved_activation": (_I, [_P, _I, _P, _I, _I, _P]),
    "cpc_encoder_forward": (_I, [_P] * 5 + [_I, _I, _P]),
    "cpc_encoder_prepare_weights": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cpc_encoder_backward": (_I, [_P] * 7 + [_I, _I, _P]),
    "cpc_encoder_backward_streams": (_I, [_P] * 7 + [_I, _I, _P, _P]),
    "cpc_set_conv_tile": (_I, [_I]),
    "cpc_set_gru_mode": (_I, [_I]),
    "cpc_set_gru_spin_limit": (_I, [_I]),
    "cpc_gemm_nt": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "cpc_gemm_tn_scratch_floats": (_L, [_I, _I, _I]),
    "cpc_gemm_tn": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "cpc_nce_scores_forward": (_I, [_P] * 7 + [_I, _I, _I, _I, _P]),
    "cpc_nce_scores_backward": (_I, [_P] * 10 + [_I, _I, _I, _I, _P]),
    "cpc_transformer_layout": (_I, [_I, _I, _P]),
    "cpc_transformer_hidden": (_I, [_P, _P, _I, _I, _P]),
    "cpc_transformer_layer_forward": (_I, [_P] * 5 + [_I, _I, _P]),
    "cpc_transformer_layer_backward": (_I, [_P] * 7 + [_I, _I, _P]),
    "cpc_transformer_layer_forward_dropout": (_I, [_P] * 5 + [_I, _I, _F, ctypes.c_ulonglong, _P]),
    "cpc_transformer_layer_backward_dropout": (_I, [_P] * 7 + [_I, _I, _F, ctypes.c_ulonglong, _P]),
    "cpc_transformer_group_forward": (_I, [_P] * 5 + [_I, _I, _I, _F, ctypes.c_ulonglong, _P]),
    "cpc_transformer_group_backward": (_I, [_P] * 7 + [_I, _I, _I, _F, ctypes.c_ulonglong, _P]),
    "cpc_dropout_keep_mask": (_I, [_P, _L, _I, _I, _F, ctypes.c_ulonglong, _P]),
    "cpc_gru_layout": (_I, [_I, _I, _I, _P]),
    "cpc_gru_forward": (_I, [_P] * 7 + [_I, _I, _I, _P]),
    "cpc_gru_forward_coef": (_I, [_P] * 8 + [_I, _I, _I, _P]),
    "cpc_gru_forward_prepare": (_I, [_P, _I, _I, _I, _P]),
    "cpc_gru_forward_coef_prepared": (_I, [_P] * 8 + [_I, _I, _I, _P]),
    "cpc_gru_backward": (_I, [_P] * 9 + [_I, _I, _I, _P]),
    "cpc_gru_coef_floats": (_L, [_I, _I, _I]),
    "cpc_gru_backward_coef": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "cpc_gru_backward_with_coef": (_I, [_P] * 10 + [_I, _I, _I, _P]),
    "cpc_gru_backward_streams": (_I, [_P] * 10 + [_I, _I, _I, _P, _P]),
    "cpc_nce_layout": (_I, [_I, _I, _I, _I, _P]),
    "cpc_nce_prepare": (_I, [_P] * 6 + [_I, _I, _I, _I, _P]),
    "cpc_nce_forward": (_I, [_P] * 8 + [_I, _I, _I, _I, _P]),
    "cpc_nce_forward_prepared": (_I, [_P] * 8 + [_I, _I, _I, _I, _P]),
    "cpc_nce_bounds": (_I, [_P, ctypes.c_float, _P, _P, _I, _I, _I, _I, _P]),
    "cpc_nce_forward_streams": (_I, [_P] * 8 + [_I, _I, _I, _I, _I, _P, _P]),
    "cpc_nce_backward_prepare": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "cpc_nce_backward_prepared": (_I, [_P] * 10 + [_I, _I, _I, _I, _P]),
    "cpc_nce_backward": (_I, [_P] * 12 + [_I, _I, _I, _I, _P]),
    "cpc_nce_backward_streams": (_I, [_P] * 12 + [_I, _I, _I, _I, _P, _P]),
    "cpc_nce_backward_dz": (_I, [_P] * 7 + [_I, _I, _I, _I, _P]),
    "cpc_nce_backward_dwall": (_I, [_P] * 4 + [_I, _I, _I, _I, _P]),
    "cpc_set_nce_fused": (_I, [_I]),
    "cpc_get_nce_fused": (_I, []),
    "cpc_set_nce_grid": (_I, [_I]),
    "cpc_set_nce_debug": (_I, [_I]),
    "cpc_set_nce_rows_apart": (_I, [_I]),
    "cpc_set_nce_heads_dma": (_I, [_I]),
    "cpc_nce_prepare_z": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cpc_set_index_prep_groups": (_I, [_I]),
    "cpc_set_gru_wgrad_stream": (_I, [_I]),
    "cpc_nce_padded_negatives": (_I, [_I]),
    "cpc_nce_head_group": (_I, [_I, _I]),
    "cpc_set_step_schedule": (_I, [_I, _I]),
    "cpc_train_step_layout": (_I, [_I, _I, _I, _I, _P]),
    "cpc_train_step_prefetch": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cpc_train_step": (_I, [_P, _P, _P, _P, _F] + [_P] * 7 + [_I] * 5 + [_P] * 4),
    "cpc_train_step_tail": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "cpc_train_step_wait": (_I, [_P, _I, _P]),
    "cpc_set_fwd_nsplit": (_I, [_I, _I]),
    "cpc_set_step_timing": (_I, [_I]),
    "cpc_set_tail_schedule": (_I, [_I]),
    "cpc_get_step_timing": (_I, [_P]),
    "cpc_adam_step": (_I, [_P] * 5 + [_I] + [ctypes.c_double] * 6 + [_P]),
    "cpc_adam_step_capturable": (_I, [_P] * 5 + [_I] + [ctypes.c_double] * 4 + [_P, _P, _P]),
}


class CpcHipError(RuntimeError):
    pass


class Bound:
    """Typed view of a loaded library.  ``check`` turns non-zero status into exceptions."""

    def __init__(self, cdll, path):
        self.path = path
        self._cdll = cdll
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        if missing:
            raise CpcHipError(f"{path} lacks symbols declared in include/cpc_hip.h: {missing}")
        have = int(self.cpc_abi_version())
        if have != EXPECTED_ABI:
            raise CpcHipError(f"{path} reports ABI version {have}, this package binds version {EXPECTED_ABI}: rebuild it "
                              "(`python -m cpc_audio_amd.build`)")

    @staticmethod
    def check(status, what=""):
        if status == 0:
            return
        if status == 1:
            raise ValueError(f"cpc_hip {what}: unsupported or inconsistent shape (CPC_ERR_SHAPE)")
        if status == 2:
            raise ValueError(f"cpc_hip {what}: bad argument (CPC_ERR_ARG)")
        raise CpcHipError(f"cpc_hip {what}: HIP error {status - 1000}")


def bind(path):
    return Bound(ctypes.CDLL(path), path)


_lock = threading.Lock()
_bound = None


def get():
    """The product library.  Raises if libcpc_hip.so has not been built."""
    global _bound
    if _bound is None:
        with _lock:
            if _bound is None:
                path = os.environ.get("CPC_HIP_LIB", LIB_PATH)      # developer override: another build of the same ABI
                if not os.path.exists(path):
                    raise CpcHipError(
                        f"{path} not found: build it with `python -m cpc_audio_amd.build` "
                        "(hipcc, gfx950).  There is no CPU fallback.")
                import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)
                _bound = bind(path)
    return _bound


def ptr(t):
    """Device (or host, in emulator tests) pointer of a contiguous tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "cpc_hip expects contiguous tensors"
    return t.data_ptr()


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    return arr
