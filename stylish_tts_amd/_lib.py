"""ctypes binding of libstylish_hip.so (include/stylish_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this module raises,
and every operator raises RuntimeError on a non-zero status with the library's error text.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STTS_LIB: another build of the library next to the product one (diagnostic builds: STTS_BUILD_TAG=_wntrace STTS_HIPCC_FLAGS=-DSTTS_WN_TRACE ...)
LIB_PATH = os.path.abspath(os.environ["STTS_LIB"]) if os.environ.get("STTS_LIB") else os.path.join(_HERE, "libstylish_hip.so")


class ModelDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_fft", "win_length", "hop_length", "sample_rate",
        "style_dim", "inter_dim",
        "dec_hidden", "dec_residual",
        "gen_input", "gen_hidden", "gen_inter", "gen_io_kernel",
        "tokens", "te_hidden", "te_filter", "te_heads", "te_layers", "te_kernel",
        "style_layers",
        "dur_layers", "dur_classes", "dur_max",
        "pe_inter",
    )]


class CfmDims(C.Structure):
    """include/stylish_hip.h: stts_cfm_dims (the constructor keywords of the reference's CfmMelDecoder)."""
    _fields_ = [(n, C.c_int32) for n in ("feat_dim", "asr_dim", "spk_dim", "hidden_dim", "emb_dim", "depth", "enc_blocks", "dec_blocks",
                                         "prev_depth", "post_depth", "head_dim")]


def dims_from_config(cfg) -> ModelDims:
    g, te, du = cfg.generator, cfg.text_encoder, cfg.duration_predictor
    return ModelDims(
        cfg.n_fft, cfg.win_length, cfg.hop_length, cfg.sample_rate,
        cfg.style_dim, cfg.inter_dim,
        cfg.decoder.hidden_dim, cfg.decoder.residual_dim,
        g.input_dim, g.hidden_dim, g.conv_intermediate_dim, g.io_conv_kernel_size,
        te.tokens, te.hidden_dim, te.filter_channels, te.heads, te.layers, te.kernel_size,
        cfg.style_encoder.layers,
        du.n_layer, du.duration_classes, du.max_duration,
        cfg.pitch_energy_predictor.inter_dim,
    )


_P = C.c_void_p
_I = C.c_int
_SZ = C.c_size_t
_I64 = C.c_int64

# name -> (restype, argtypes); mirrors include/stylish_hip.h one to one
SIGNATURES = {
    "stts_last_error": (C.c_char_p, []),
    "stts_version": (_I, []),
    "stts_ctx_create": (_I, [C.POINTER(ModelDims), _I, C.POINTER(_P)]),
    "stts_ctx_destroy": (None, [_P]),
    "stts_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I]),
    "stts_finalize_weights": (_I, [_P, _I]),
    "stts_check_status": (_I, [_P, _P]),
    "stts_frame_workspace_bytes": (_SZ, [_P, _I64, _I, _I]),
    "stts_har_ld": (_I, [_P]),
    "stts_decoder_forward": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ]),
    "stts_prior_flow_forward": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _SZ]),
    "stts_harmonic_stft": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _SZ]),
    "stts_vocoder_forward": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _SZ]),
    "stts_frame_path": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _SZ, _I]),
    "stts_frame_offsets": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "stts_phoneme_workspace_bytes": (_SZ, [_P, _I64, _I64, _I]),
    "stts_text_encoder_forward": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _SZ]),
    "stts_text_style_forward": (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _SZ]),
    "stts_duration_forward": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ]),
    "stts_pitch_energy_forward": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _SZ, _I]),
    "stts_duration_decode": (_I, [_P, _P, _I, _I, _P]),
    "stts_duration_to_alignment": (_I, [_P, _P, _I, _I, _P]),
    "stts_length_regulate": (_I, [_P, _P, _I, _P, _P, _P, _I64, _I, _P, _I, _I, _P, _I, _P]),
    "stts_set_precision": (_I, [_P, _I]),
    "stts_upsample4": (_I, [_P, _P, _I, _P, _P, _P, _P, _P]),
    "stts_euler_step": (_I, [_P, _P, _P, C.c_float, C.c_int64]),
    "stts_cfm_finalize": (_I, [_P, _P]),
    "stts_cfm_workspace_bytes": (_SZ, [_P, _I64, _I]),
    "stts_cfm_estimator": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ]),
    "stts_to_time_major": (_I, [_P, _P, _I, _I, _I, _P, _I]),
    "stts_to_channel_major": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "stts_conv_stft_transform": (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I]),
    "stts_conv_stft_inverse": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _SZ]),
    "stts_profile_begin": (_I, []),
    "stts_profile_end": (_I, [_P, C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "stts_profile_report": (_I, [_P, C.c_char_p, _SZ]),
    "stts_op_mrf_block": (_I, [_P, _P, C.c_char_p, _I, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _SZ]),
}
# the test surface (include/stylish_hip.h under STTS_TEST_OPS): present only in a library built with -DSTTS_TEST_OPS
TEST_SIGNATURES = {
    "stts_bench_gemm": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_double), _I]),
    "stts_op_conv1d": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I]),
    "stts_op_adain_block": (_I, [_P, _P, C.c_char_p, _I, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _SZ]),
    "stts_op_attention": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME, libamdhip64.so.7, as
    # the system one this library is linked against).  If this library were loaded first the system copy would come in,
    # torch would then add its own, and whichever initialises second reports "no ROCm-capable device".  Importing torch
    # first makes the dynamic loader resolve our dependency to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure C-ABI use without PyTorch: the system runtime is the only one
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in TEST_SIGNATURES.items():  # a product-only build (STTS_PRODUCT_ONLY=1) has none of these
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("stylish_hip: " + (load().stts_last_error() or b"?").decode("utf-8", "replace"))
