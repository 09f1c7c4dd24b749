"""ctypes binding of the C-ABI library (include/sahs_nerf.h).

There is NO fallback: if libsahs_nerf.so is missing or a symbol is absent this raises, and every
op in ops.py goes through here.  Build with ``python sahs-deformable-nerf_amd/build.py`` (hipcc,
gfx950; cross-compiles without a GPU).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAHS_NERF_LIB") or os.path.join(_HERE, "libsahs_nerf.so")   # override: ablation builds (tools/ablate.py)
SAHS_F32, SAHS_BF16, SAHS_BF16X3 = 0, 1, 3

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_F = ctypes.c_float

# name -> (restype, argtypes); must list every function declared in include/sahs_nerf.h
SIGNATURES = {
    "sahs_abi_version": (_I, []),
    "sahs_last_error": (ctypes.c_char_p, []),
    "sahs_param_count": (_L, []),
    "sahs_packed_words": (_L, [_I]),
    "sahs_frame_words": (_L, []),
    "sahs_pack_weights": (_I, [_P, _P, _I, _P]),
    "sahs_fold_conditioning": (_I, [_P, _P, _P, _I, _P, _P]),
    "sahs_get_ray_bundle": (_I, [_I, _I, _F, _F, _F, _F, _P, _I, _P, _P, _P]),
    "sahs_ray_uniforms": (_I, [ctypes.c_uint64, _I, _L, _L, _I, _P, _P]),
    "sahs_stratified_depths": (_I, [_L, _I, _P, _I, _I, _P, _P, _P]),
    "sahs_field_forward": (_I, [_P, _P, _I, _L, _I, _P, _I, _P, _P, _P, _I, _P]),
    "sahs_composite_forward": (_I, [_L, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "sahs_resample": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sahs_sample_pdf": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sahs_render_rays": (_I, [_P, _P, _I, _L, _P, _I, _I, _I, _I, _I] + [_P] * 18),
    "sahs_act_words_per_sample": (_L, []),
    "sahs_field_backward_workspace_words": (_L, [_L]),
    "sahs_field_forward_save": (_I, [_P, _P, _I, _L, _I, _P, _I, _P, _P, _P, _P]),
    "sahs_field_backward": (_I, [_P, _P, _I, _L, _P, _P, _P, _P, _P, _P]),
    "sahs_composite_backward": (_I, [_L, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sahs_conditioning_backward": (_I, [_P, _P, _P, _P, _P, _P]),
    "sahs_model_param_count": (_L, [_I]),
    "sahs_model_packed_words": (_L, [_I, _I]),
    "sahs_model_frame_words": (_L, [_I]),
    "sahs_model_pack_weights": (_I, [_I, _P, _P, _I, _P]),
    "sahs_model_fold_conditioning": (_I, [_I, _P, _P, _P, _I, _P, _P]),
    "sahs_model_field_forward": (_I, [_I, _P, _P, _I, _L, _I, _P, _I, _P, _P, _P, _I, _P]),
    "sahs_model_render_rays": (_I, [_I, _P, _P, _I, _L, _P, _I, _I, _I, _I, _I] + [_P] * 18),
    "sahs_composite_forward_rows": (_I, [_L, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _P]),
    "sahs_model_executed_macs_per_sample": (_L, [_I, _I]),
    "sahs_model_executed_macs_part": (_L, [_I, _I, _I]),
    "sahs_model_render_rays_rows": (_I, [_I, _P, _P, _I, _L, _P, _I, _I, _I, _I, _I] + [_P] * 10 + [_I, _P, _P, _P, _P]),
    "sahs_resample_merge": (_I, [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sahs_model_act_words_part": (_L, [_I, _I]),
    "sahs_model_field_forward_split_save": (_I, [_I, _P, _P, _I, _I, _L, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    "sahs_model_field_backward_split": (_I, [_I, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sahs_route_xw_grad": (_I, [_L, _I, _I, _P, _P, _P, _P, _P]),
    "sahs_bf16_exact_leaky": (_I, [_I]),
    "sahs_model_bits_words_part": (_L, [_I, _I]),
    "sahs_model_field_forward_split_save_bits": (_I, [_I, _P, _P, _I, _I, _L, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "sahs_model_field_forward_split_save_bits_x3": (_I, [_I, _P, _P, _I, _I, _L, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "sahs_model_field_backward_fused_workspace_words": (_L, [_I, _I, _L]),
    "sahs_model_field_backward_fused": (_I, [_I, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sahs_stage1_loss_forward": (_I, [_L, _P, _P, _P, _I, _P, _P, _P, _P]),
    "sahs_composite_backward_loss": (_I, [_L, _I, _P, _P, _P, _I, _P, _P, _I] + [_P] * 5 + [_P, _P, _I, _P, _P, _P, _P, _P]),
    "sahs_model_field_forward_split": (_I, [_I, _P, _P, _I, _I, _I, _L, _I, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    "sahs_model_act_words_per_sample": (_L, [_I]),
    "sahs_model_field_backward_workspace_words": (_L, [_I, _L]),
    "sahs_model_field_forward_save": (_I, [_I, _P, _P, _I, _L, _I, _P, _I, _P, _P, _P, _P]),
    "sahs_model_field_backward": (_I, [_I, _P, _P, _I, _L, _P, _P, _P, _P, _P, _P]),
    "sahs_backward_gemm_precision": (_I, [_I]),
    "sahs_spade_modulate_workspace_words": (_L, [_L]),
    "sahs_spade_modulate": (_I, [_L, _L, _P, _P, _P, _F, _F, _P, _P, _P]),
    "sahs_probe_arm": (_I, [_I]),
    "sahs_probe_disarm": (_I, []),
    "sahs_probe_count": (_I, []),
    "sahs_probe_dropped": (_I, []),
    "sahs_probe_read": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_L), ctypes.POINTER(_F)]),
}

_lib = None


class SahsError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SahsError("%s not found: the HIP extension is not built (run `python sahs-deformable-nerf_amd/build.py`); "
                            "there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if L.sahs_abi_version() != 1:
            raise SahsError("libsahs_nerf.so ABI version %d, expected 1" % L.sahs_abi_version())
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        raise SahsError("%s failed (%d): %s" % (what, code, lib().sahs_last_error().decode()))
