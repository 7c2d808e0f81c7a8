"""ctypes binding of libbmc_hip.so (C ABI: include/bmc_hip.h).

There is no CPU fallback: if the shared library is missing the import of this
module raises, and every call checks the return code and raises RuntimeError
with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libbmc_hip.so has to bind to the HIP runtime torch already loaded
              # (torch ships its own libamdhip64; two runtimes in one process do not share the device context)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMC_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libbmc_hip.so")  # override: tools/ experiments only

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C bmcnet-esr_amd/csrc). "
        "The BMCNet MI355X path has no CPU fallback.")

_lib = C.CDLL(LIB_PATH)

MAX_SRC = 6
c_fp = C.c_void_p  # device pointers travel as integers


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("batch_stride", C.c_longlong), ("pix_stride", C.c_int), ("nch", C.c_int),
                ("batch_shift", C.c_int), ("batch_mod", C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [("nsrc", C.c_int), ("src", Src * MAX_SRC), ("wpacked", C.c_void_p), ("bias", C.c_void_p),
                ("w_group_stride", C.c_longlong), ("bias_group_stride", C.c_int), ("batch_per_group", C.c_int),
                ("out", C.c_void_p), ("out_batch_stride", C.c_longlong), ("out_pix_stride", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cout", C.c_int), ("Coutpad", C.c_int),
                ("taps", C.c_int), ("relu", C.c_int), ("residual", Src), ("mask", Src), ("accumulate", C.c_int),
                ("math", C.c_int)]


class PgemmArgs(C.Structure):
    _fields_ = [("a", Src), ("nsrc", C.c_int), ("src", Src * MAX_SRC), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("taps", C.c_int), ("batch_per_group", C.c_int), ("slabs", C.c_void_p), ("nsplit", C.c_int),
                ("zeros", C.c_void_p), ("bias_slabs", C.c_void_p), ("math", C.c_int), ("tap_groups", C.c_int)]


class ChainFwdArgs(C.Structure):
    _fields_ = [("s0", Src), ("s1", Src), ("wstream", C.c_void_p), ("bias_f", C.c_void_p), ("bias_c", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("yhat", C.c_void_p), ("rstd", C.c_void_p),
                ("centre", C.c_void_p), ("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class ChainBwdArgs(C.Structure):
    _fields_ = [("dcentre", Src), ("wstream", C.c_void_p), ("gamma", C.c_void_p), ("yhat", C.c_void_p), ("rstd", C.c_void_p),
                ("dz", C.c_void_p), ("ds1", C.c_void_p), ("ds0", C.c_void_p), ("ds0_add", Src), ("n", C.c_int), ("C", C.c_int),
                ("H", C.c_int), ("W", C.c_int)]


class MmTerm(C.Structure):
    _fields_ = [("a", C.c_void_p), ("a_sb", C.c_longlong), ("a_sg", C.c_longlong), ("a_si", C.c_int), ("a_sk", C.c_int),
                ("b", C.c_void_p), ("b_sb", C.c_longlong), ("b_sg", C.c_longlong), ("b_sk", C.c_int), ("b_sj", C.c_int),
                ("w", C.c_void_p), ("w_sb", C.c_longlong), ("w_sg", C.c_longlong), ("w_sk", C.c_int)]


class SmallMmArgs(C.Structure):
    _fields_ = [("nterms", C.c_int), ("t", MmTerm * 2), ("nbatch", C.c_int), ("batch_per_group", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("alpha", C.c_float),
                ("u", C.c_void_p), ("u_sb", C.c_longlong), ("u_sg", C.c_longlong),
                ("v", C.c_void_p), ("v_sb", C.c_longlong), ("v_sg", C.c_longlong),
                ("c", C.c_void_p), ("c_sb", C.c_longlong), ("c_sg", C.c_longlong), ("c_si", C.c_int), ("c_sj", C.c_int),
                ("vec_out", C.c_void_p), ("vo_sb", C.c_longlong), ("vo_sg", C.c_longlong), ("accumulate", C.c_int)]


def _sig(name, argtypes, restype=C.c_int):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


_ll, _i, _f, _p = C.c_longlong, C.c_int, C.c_float, C.c_void_p

bmc_version = _sig("bmc_version", [])
bmc_last_error = _sig("bmc_last_error", [], C.c_char_p)
_events = _sig("bmc_events_to_channels", [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p])
_voxel = _sig("bmc_events_to_voxel", [_p, _p, _p, _p, _p, _ll, _i, _i, _i, _i, _p, _i, _p, _p])
_stack_pol = _sig("bmc_events_to_stack_polarity", [_p, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _p, _p, _i, _p])
_mask = _sig("bmc_events_to_mask", [_p, _p, _p, _ll, _i, _i, _p, _p, _i, _p])
_stack = _sig("bmc_events_to_stack", [_p, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _p, _p, _i, _p])
_enc_raw = _sig("bmc_encode_raw_events", [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p])
_events_ws = _sig("bmc_events_binned_ws_bytes", [_ll, _i, _i, _i], C.c_longlong)
_events_binned = _sig("bmc_events_to_channels_binned", [_p, _p, _p, _p, _ll, _i, _i, _i, _p, _i, _p, _ll, _p])
_enc_raw_binned = _sig("bmc_encode_raw_events_binned", [_p, _p, _p, _p, _p, _ll, _i, _i, _i, _p, _p, _ll, _p])
_ev_torch_ws = _sig("bmc_events_torch_ws_ints", [_ll, _i, _i], C.c_longlong)
_ev_img_torch = _sig("bmc_events_to_image_torch", [_p, _p, _p, _ll, _i, _i, _i, _i, _i, _p, _p, _p])
_ev_vox_torch = _sig("bmc_events_to_voxel_torch", [_p, _p, _p, _p, _ll, _i, _i, _i, _p, _p, _p])
_pack_w = _sig("bmc_pack_weight", [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p])
_pack_wt = _sig("bmc_pack_weight_t", [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p])
_pack_wino = _sig("bmc_pack_weight_wino", [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p])
_pack_wino4 = _sig("bmc_pack_weight_wino4", [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p])
_ww_nsplit = _sig("bmc_wgrad_wino_nsplit", [_i, _i, _i])
_stream_low = _sig("bmc_stream_create_low_priority", [C.POINTER(C.c_void_p)])
_small_mm = _sig("bmc_small_mm", [C.POINTER(SmallMmArgs), _p])
_ww = _sig("bmc_wgrad_wino", [C.POINTER(Src), C.POINTER(Src), _i, _i, _i, _i, _p, _p, _p])
_ww_red = _sig("bmc_wgrad_wino_reduce", [_p, _i, _p, _i, _i, _i, _p, _p, _p])
_ww_multi = _sig("bmc_wgrad_wino_multi", [C.POINTER(Src), C.POINTER(Src), C.POINTER(C.c_int), _i, _i, _i, _i, _p, _p, _p])
_ptr_table = _sig("bmc_ptr_table", [C.POINTER(C.c_ulonglong), _i, _p, _p])
_wino_rows = _sig("bmc_conv_wino_rows", [_i, _i, _i, _i, _i])
_ww4_nsplit = _sig("bmc_wgrad_wino4_nsplit", [_i, _i, _i])
_ww4 = _sig("bmc_wgrad_wino4", [C.POINTER(Src), C.POINTER(Src), _i, _i, _i, _i, _p, _p, _p])
_ww4_red = _sig("bmc_wgrad_wino4_reduce", [_p, _i, _p, _i, _i, _i, _p, _p, _p])
_split_w = _sig("bmc_split_weight", [_p, _p, _ll, _i, _i, _p])
_conv = _sig("bmc_conv", [C.POINTER(ConvArgs), _p])
_pgemm = _sig("bmc_pgemm", [C.POINTER(PgemmArgs), _p])
_red_w = _sig("bmc_pgemm_reduce_weight", [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p])
_red_wg = _sig("bmc_pgemm_reduce_weight_groups", [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p])
_red_p = _sig("bmc_pgemm_reduce_plain", [_p, _i, _i, _i, _i, _f, _p, _p])
_colsum = _sig("bmc_colsum", [_p, _ll, _i, _i, _p, _p, _i, _p])
_relu_bwd = _sig("bmc_relu_bwd", [_p, _p, _p, _ll, _p])
_ln_fwd = _sig("bmc_layernorm_fwd", [_p, _p, _p, _ll, _i, _f, _p, _p, _p])
_ln_bwd = _sig("bmc_layernorm_bwd", [_p, _p, _p, _p, _ll, _i, _p, _p, _p, _p, _i, _p])
_sm_fwd = _sig("bmc_softmax_fwd", [_p, _ll, _i, _p, _p])
_sm_bwd = _sig("bmc_softmax_bwd", [_p, _p, _ll, _i, _f, _p, _p])
_pack_in = _sig("bmc_pack_inputs", [_p, _ll, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _p, _p, _p])
_unshuffle = _sig("bmc_unshuffle_to_nhwc", [_p, _i, _i, _i, _i, _i, _p, _i, _p])
_shuffle = _sig("bmc_shuffle_to_hr", [_p, _i, _i, _i, _i, _i, _p, _ll, _ll, _ll, _ll, _p, _i, _p])
_head_mse_fwd = _sig("bmc_head_mse_fwd", [_p, _i, _i, _i, _i, _i, _p, _ll, _ll, _ll, _ll, _p, _ll, _p, _p, _p, _p])
_head_mse_bwd = _sig("bmc_head_mse_bwd", [_p, _p, _p, _ll, _p, _i, _i, _i, _i, _i, _p, _p])
_group_sum = _sig("bmc_group_sum", [_p, _i, _ll, _p, _p])
_bicubic_fwd = _sig("bmc_bicubic_resize_fwd", [_p, _ll, _i, _i, _i, _i, _p, _p])
_bicubic_bwd = _sig("bmc_bicubic_resize_bwd", [_p, _ll, _i, _i, _i, _i, _p, _p])
_chain_fwd = _sig("bmc_chain_fwd", [C.POINTER(ChainFwdArgs), _p])
_chain_bwd = _sig("bmc_chain_bwd", [C.POINTER(ChainBwdArgs), _p])
_chain_affine = _sig("bmc_chain_affine_grads", [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p])

EXPORTS = ["bmc_version", "bmc_last_error", "bmc_events_to_channels", "bmc_events_to_voxel", "bmc_events_to_stack", "bmc_encode_raw_events", "bmc_pack_weight", "bmc_pack_weight_t", "bmc_split_weight", "bmc_conv",
           "bmc_pgemm", "bmc_pgemm_reduce_weight", "bmc_pgemm_reduce_plain", "bmc_colsum", "bmc_relu_bwd",
           "bmc_layernorm_fwd", "bmc_layernorm_bwd", "bmc_softmax_fwd", "bmc_softmax_bwd", "bmc_pack_inputs",
           "bmc_unshuffle_to_nhwc", "bmc_shuffle_to_hr", "bmc_bicubic_resize_fwd", "bmc_bicubic_resize_bwd",
           "bmc_chain_fwd", "bmc_chain_bwd", "bmc_chain_affine_grads", "bmc_group_sum",
           "bmc_head_mse_fwd", "bmc_head_mse_bwd", "bmc_events_to_stack_polarity", "bmc_events_to_mask",
           "bmc_pgemm_reduce_weight_groups", "bmc_events_binned_ws_bytes", "bmc_events_to_channels_binned",
           "bmc_encode_raw_events_binned", "bmc_pack_weight_wino", "bmc_pack_weight_wino4", "bmc_wgrad_wino_nsplit", "bmc_wgrad_wino",
           "bmc_wgrad_wino_reduce", "bmc_events_torch_ws_ints", "bmc_events_to_image_torch",
           "bmc_events_to_voxel_torch", "bmc_stream_create_low_priority", "bmc_small_mm",
           "bmc_wgrad_wino4_nsplit", "bmc_wgrad_wino4", "bmc_wgrad_wino4_reduce", "bmc_wgrad_wino_multi", "bmc_ptr_table",
           "bmc_conv_wino_rows"]


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {bmc_last_error().decode()}")


def call(fn, what, *args):
    check(fn(*args), what)


def has_symbol(name: str) -> bool:
    return hasattr(_lib, name)
