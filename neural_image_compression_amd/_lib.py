"""ctypes binding of liblic_hip.so (C ABI declared in include/lic.h).

There is NO fallback: if the shared library is missing or a launch fails this raises.  The
library is built in-tree by `__graft_entry__.build()` / `make -C neural_image_compression_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblic_hip.so")

# enum lic_epilogue
EPI_NONE, EPI_LEAKY, EPI_MUL_LEAKY_MASK, EPI_GDN, EPI_IGDN, EPI_GDN_BWD, EPI_IGDN_BWD, EPI_CONV_GDN, EPI_CONV_IGDN = range(9)
FE_NPARAM = 43

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class IgemmDesc(C.Structure):
    _fields_ = [("in_", _vp), ("w", _vp), ("bias", _vp), ("out", _vp), ("out2", _vp), ("aux", _vp),
                ("aux2", _vp), ("aux3", _vp), ("res", _vp),
                ("in_ld", _i64), ("out_ld", _i64), ("out2_ld", _i64), ("aux_ld", _i64),
                ("aux2_ld", _i64), ("aux3_ld", _i64), ("res_ld", _i64),
                ("B", _i32), ("Hi", _i32), ("Wi", _i32), ("Cin", _i32),
                ("Ho", _i32), ("Wo", _i32), ("Cout", _i32),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("transposed", _i32), ("prologue", _i32), ("epilogue", _i32),
                ("tap_mask", C.c_uint32), ("slope", _f32), ("workspace", _vp), ("workspace_bytes", _sz),
                ("out3", _vp), ("out3_ld", _i64),
                ("force_bm", _i32), ("force_tn", _i32), ("force_split", _i32), ("reserved0", _i32)]


class WgradDesc(C.Structure):
    _fields_ = [("p", _vp), ("g", _vp), ("dst", _vp),
                ("p_ld", _i64), ("g_ld", _i64),
                ("dst_sm", _i64), ("dst_sn", _i64), ("dst_stap", _i64),
                ("B", _i32), ("Hs", _i32), ("Ws", _i32), ("Cp", _i32),
                ("Hl", _i32), ("Wl", _i32), ("Cg", _i32),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("g_is_row", _i32), ("sq_p", _i32), ("sq_g", _i32), ("scale", _f32),
                ("force_tm", _i32), ("force_tn", _i32), ("force_split", _i32)]


PREP_PACK_F32, PREP_PACK_BF16, PREP_MAP, PREP_MASK_INPLACE, PREP_PACK_BF16_KPERM, PREP_PACK_BF16_STEM = range(6)


class PrepJob(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("mask", _vp),
                ("s_tap", _i64), ("s_kq", _i64), ("s_kr", _i64), ("s_nq", _i64), ("s_nr", _i64),
                ("kind", _i32), ("taps", _i32), ("K", _i32), ("N", _i32), ("kdiv", _i32), ("ndiv", _i32),
                ("transform", _i32), ("bound", _f32), ("pedestal", _f32),
                ("cpt", _i32), ("npad", _i32), ("tiled", _i32), ("v4", _i32), ("block0", _i32), ("nblocks", _i32),
                ("total", _i64)]


REDUCE_SLABS, REDUCE_COLUMNS = 0, 1
REDUCE_EPI_NONE, REDUCE_EPI_REPARAM = 0, 1
REDUCE_MAX_JOBS = 32


class ReduceJob(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("param", _vp),
                ("sm", _i64), ("smr", _i64), ("sn", _i64), ("snr", _i64), ("stap", _i64),
                ("kind", _i32), ("epilogue", _i32),
                ("splitk", _i32), ("ntaps", _i32), ("Cm", _i32), ("Cn", _i32), ("Mvalid", _i32), ("Nvalid", _i32),
                ("mdiv", _i32), ("ndiv", _i32),
                ("scale", _f32), ("bound", _f32),
                ("block0", _i32), ("nblocks", _i32)]


class AdamJob(C.Structure):
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("n", _i64), ("block0", _i32), ("nblocks", _i32)]


# name -> (restype, argtypes); every symbol declared in include/lic.h
SIGNATURES = {
    "lic_igemm": (C.c_int, [C.POINTER(IgemmDesc), _vp]),
    "lic_igemm_workspace_bytes": (_sz, [C.POINTER(IgemmDesc)]),
    "lic_igemm_kernel_name": (C.c_int, [C.POINTER(IgemmDesc), C.c_char_p, _sz]),
    "lic_wgrad_stage": (C.c_int, [C.POINTER(WgradDesc), _vp, _sz, _i32, _vp]),
    "lic_wgrad_kernel_name": (C.c_int, [C.POINTER(WgradDesc), C.c_char_p, _sz]),
    "lic_igemm_plan": (C.c_int, [C.POINTER(IgemmDesc), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "lic_packed_weight_floats": (_i64, [_i32, _i32, _i32]),
    "lic_pack_weight": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    "lic_wgrad_workspace_bytes": (_sz, [C.POINTER(WgradDesc)]),
    "lic_wgrad": (C.c_int, [C.POINTER(WgradDesc), _vp, _sz, _vp]),
    "lic_wgrad_plan": (C.c_int, [C.POINTER(WgradDesc), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "lic_colsum_workspace_bytes": (_sz, [_i64, _i32]),
    "lic_colsum": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _sz, _vp]),
    "lic_permute3": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "lic_im2col": (C.c_int, [_vp, _vp] + [_i32] * 11 + [_vp]),
    "lic_col2im": (C.c_int, [_vp, _vp, _vp] + [_i32] * 11 + [_vp]),
    "lic_mul_inplace": (C.c_int, [_vp, _vp, _i64, _vp]),
    "lic_leaky_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp]),
    "lic_gdn_reparam": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "lic_gdn_reparam_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp]),
    "lic_gdn_reparam_bwd2": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _i64, _f32, _vp]),
    "lic_gdn_dnorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "lic_quantize": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "lic_entropy_params_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_entropy_params_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_gmm_likelihood_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "lic_gmm_likelihood_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "lic_factorized_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "lic_factorized_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "lic_factorized_channel_logits": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _vp]),
    "lic_fe_pack": (C.c_int, [_vp, _vp, _i32, _vp]),
    "lic_fe_unpack": (C.c_int, [_vp, _vp, _i32, _vp]),
    "lic_rd_loss_workspace_bytes": (_sz, [_i32]),
    "lic_rd_loss_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i64, _f32, _vp, _vp, _sz, _vp]),
    "lic_rd_loss_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i64, _f32, _vp, _vp, _vp, _vp, _vp]),
    "lic_packed_weight_bf16_elems": (_i64, [_i32, _i32, _i32]),
    "lic_pack_weight_bf16": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    "lic_pack_weight_bf16_kperm": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    "lic_igemm_bf16": (C.c_int, [C.POINTER(IgemmDesc), _i32, _vp]),
    "lic_igemm_bf16_kernel_name": (C.c_int, [C.POINTER(IgemmDesc), C.c_char_p, _sz]),
    "lic_igemm_bf16_workspace_bytes": (_sz, [C.POINTER(IgemmDesc)]),
    "lic_igemm_bf16_fused_gdn_supported": (C.c_int, [_i32, _i32]),
    "lic_stem_gdn_bf16_supported": (C.c_int, [_i32] * 6),
    "lic_stem_weight_bf16_elems": (C.c_int64, [_i32]),
    "lic_pack_stem_weight_bf16": (C.c_int, [_vp, _vp, _i32, _vp]),
    "lic_stem_gdn_bf16": (C.c_int, [_vp] * 8 + [_i32] * 5 + [_vp]),
    "lic_wgrad_bf16_kernel_name": (C.c_int, [C.POINTER(WgradDesc), C.c_char_p, _sz]),
    "lic_wgrad_bf16_workspace_bytes": (_sz, [C.POINTER(WgradDesc)]),
    "lic_wgrad_bf16": (C.c_int, [C.POINTER(WgradDesc), _vp, _sz, _vp]),
    "lic_im2col_bf16": (C.c_int, [_vp, _vp] + [_i32] * 11 + [_vp]),
    "lic_col2im_bf16": (C.c_int, [_vp, _vp, _vp] + [_i32] * 11 + [_vp]),
    "lic_colsum_bf16_workspace_bytes": (_sz, [_i64, _i32]),
    "lic_colsum_bf16": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _sz, _vp]),
    "lic_quantize_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "lic_wgrad_bf16_partial": (C.c_int, [C.POINTER(WgradDesc), _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_wgrad_partial": (C.c_int, [C.POINTER(WgradDesc), _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_colsum_partial": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_colsum_bf16_partial": (C.c_int, [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_colsum2_bf16_partial": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_reduce_batch": (C.c_int, [C.POINTER(ReduceJob), _i32, _vp]),
    "lic_colsum2_bf16": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "lic_gdn_dnorm_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "lic_gdn_bwd_bf16_supported": (C.c_int, [_i32]),
    "lic_gdn_bwd_bf16_partial_rows": (_i64, [_i64]),
    "lic_gdn_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_gdn_bwd_bf16_recompute": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_leaky_bwd_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp]),
    "lic_leaky_bwd_colsum_bf16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    "lic_igemm_fused_gdn_supported": (C.c_int, [_i32, _i32]),
    "lic_igemm_fused_gdn_preferred": (C.c_int, [C.POINTER(IgemmDesc)]),
    "lic_gdn_supported": (C.c_int, [_i32]),
    "lic_gdn_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_gdn_bwd_partial_rows": (_i64, [_i64]),
    "lic_gdn_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lic_factorized_cdf_tables": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "lic_gmm_cdf_tables": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "lic_msssim_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "lic_msssim": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _sz,
                             _vp]),
    "lic_u8_to_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "lic_tensor_stats_workspace_bytes": (_sz, []),
    "lic_tensor_stats": (C.c_int, [_vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "lic_prep_plan": (_i64, [C.POINTER(PrepJob), _i32]),
    "lic_prep_run": (C.c_int, [_vp, _i32, _i64, _vp]),
    "lic_adam_plan": (_i64, [C.POINTER(AdamJob), _i32]),
    "lic_adam_run": (C.c_int, [_vp, _i32, _i64, _vp] + [C.c_double] * 7 + [_vp]),
    "lic_head_convt_bf16_supported": (C.c_int, [_i32] * 7),
    "lic_head_convt_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "lic_stem_conv_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "lic_plan_create": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "lic_plan_info": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "lic_plan_replay": (C.c_int, [_vp, _vp, C.POINTER(C.c_void_p), _i32]),
    "lic_plan_tune": (C.c_int, [_vp, _vp, C.POINTER(C.c_void_p), _i32, C.POINTER(C.c_double)]),
    "lic_plan_destroy": (None, [_vp]),
    "lic_plan_last_error": (C.c_char_p, []),
    "lic_version": (C.c_int, []),
    "lic_last_hip_error": (C.c_int, []),
    "lic_arch": (C.c_char_p, []),
}

_lib = None


class LicError(RuntimeError):
    pass


def load():
    """Load liblic_hip.so (raises LicError when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LicError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C "
                "neural_image_compression_amd/csrc). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the header diverge
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_STATUS = {-1: "LIC_ERR_INVALID", -2: "LIC_ERR_UNSUPPORTED", -3: "LIC_ERR_LAUNCH", -4: "LIC_ERR_WORKSPACE"}


def check(rc: int, what: str):
    if rc != 0:
        extra = f" (hipError {load().lic_last_hip_error()})" if rc == -3 else ""
        raise LicError(f"{what} failed: {_STATUS.get(rc, rc)}{extra}")
