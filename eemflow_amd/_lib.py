"""ctypes binding of libeemflow_hip.so (C ABI: include/eemflow_hip.h).

There is no fallback: if the library is missing or a call fails this raises.  The product path
never computes on the CPU and never imports oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EEM_LIB_PATH") or os.path.join(_HERE, "libeemflow_hip.so")   # (EEM_LIB_PATH: a diagnostic build of the same library)

_c_float_p = ctypes.c_void_p          # device/host pointers travel as integers


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("ms", ctypes.c_float), ("blocks", ctypes.c_int), ("pipe", ctypes.c_int), ("reserved", ctypes.c_int)]


_SIGNATURES = {
    "eemflow_abi_version": (ctypes.c_int, []),
    "eemflow_last_error": (ctypes.c_char_p, []),
    "eemflow_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "eemflow_destroy": (None, [ctypes.c_void_p]),
    "eemflow_load_weights": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "eemflow_update_weights": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_size_t, ctypes.c_void_p]),
    "eemflow_forward_train": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "eemflow_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                        ctypes.c_void_p]),
    "eemflow_sequence_loss": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             _c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "eemflow_set_image_size": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int * 4)]),
    "eemflow_use_graph": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eemflow_set_frames_in_flight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eemflow_set_deferred_input_norm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eemflow_graph_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong * 3)]),
    "eemflow_forward": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eemflow_forward_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]),
    "eemflow_time_kernels": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(KernelStat),
                                            ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "eemflow_get_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_float_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_int * 4), ctypes.c_void_p]),
    "eemflow_decoder": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _c_float_p, ctypes.c_void_p]),
    "eemflow_local_corr53": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            _c_float_p, ctypes.c_void_p]),
    "eemflow_upsample_bilinear": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eraft_corr_lookup_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p,
                                             _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eraft_corr_pyramid_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eraft_convex_upsample_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemplus_warp_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemflow_flow_error": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "eemflow_flow_error_many": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "eemflow_voxelize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "eemflow_voxelize_pair": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemflow_voxelize_many": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "eemflow_forward_backward": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _c_float_p,
                                                _c_float_p, ctypes.POINTER(ctypes.c_double * 5), ctypes.c_void_p]),
    "eemflow_optimizer_step": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_void_p]),
    "eemflow_train_stats_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "eemflow_train_stats_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "eemflow_optimizer_skipped_steps": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "eemflow_get_weights": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_size_t, ctypes.c_void_p]),
    "eraft_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "eraft_destroy": (None, [ctypes.c_void_p]),
    "eraft_load_weights": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_size_t, ctypes.c_int]),
    "eraft_forward": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int * 4), ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eraft_keep_stages": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eraft_set_frames_in_flight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eraft_forward_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.c_void_p]),
    "eraft_set_alternate_corr": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eraft_set_final_only": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eraft_get_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_float_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_int * 4), ctypes.c_void_p]),
    "eraft_corr_lookup": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "eraft_convex_upsample": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             _c_float_p, ctypes.c_void_p]),
    "eemplus_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "eemplus_destroy": (None, [ctypes.c_void_p]),
    "eemplus_load_weights": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "eemplus_forward_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "eemplus_forward": (ctypes.c_int, [ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_int * 4), _c_float_p, ctypes.c_void_p]),
    "eemplus_get_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, _c_float_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_int * 4), ctypes.c_void_p]),
    "eemplus_level": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemplus_set_frames_in_flight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "eemplus_warp": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    _c_float_p, ctypes.c_void_p]),
    "eemplus_upsample_flow_as": (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "eemop_conv2d_fwd": (ctypes.c_int, [_c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p, _c_float_p]
                         + [ctypes.c_int] * 10 + [ctypes.c_float, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eemop_pack_hint": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_longlong]),
    "eemop_pack_forget": (ctypes.c_int, [ctypes.c_longlong]),
    "eemop_pack_cache_bytes": (ctypes.c_longlong, []),
    "eemop_conv2d_bwd_data": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 12 + [_c_float_p, ctypes.c_void_p]),
    "eemop_conv2d_bwd_weight": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 12 + [_c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemop_conv2d_bwd_weight_cat": (ctypes.c_int, [_c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_int, _c_float_p]
                                    + [ctypes.c_int] * 9 + [_c_float_p, _c_float_p, ctypes.c_void_p]),
    "eemop_act_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_float, _c_float_p, ctypes.c_void_p]),
    "eemop_binary": (ctypes.c_int, [ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_float, ctypes.c_longlong, _c_float_p, ctypes.c_void_p]),
    "eemop_sum_n": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, _c_float_p, ctypes.c_void_p]),
    "eemop_gru_blend": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_longlong, _c_float_p, ctypes.c_void_p]),
    "eemop_gru_blend_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_longlong, _c_float_p, _c_float_p,
                                           _c_float_p, ctypes.c_void_p]),
    "eemop_copy_channels": (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "eemop_coords_init": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eemop_replicate_pad": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    "eemop_instnorm_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "eemop_instnorm_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p,
                                          ctypes.c_void_p]),
    "eemop_batchnorm_train_fwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_float, ctypes.c_int]
                                  + [_c_float_p] * 3 + [ctypes.c_void_p]),
    "eemop_batchnorm_train_bwd": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 4 + [_c_float_p] * 3 + [ctypes.c_void_p]),
    "eemop_batchnorm_eval_fwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "eemop_batchnorm_eval_bwd": (ctypes.c_int, [_c_float_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [_c_float_p] * 3 + [ctypes.c_void_p]),
    "eemop_corr_pyramid_fwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 4 + [_c_float_p] * 4 + [ctypes.c_void_p]),
    "eemop_corr_lookup_fwd": (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 3 + [_c_float_p, ctypes.c_void_p]),
    "eemop_convex_upsample_fwd": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [_c_float_p, ctypes.c_void_p]),
    "eemop_act_fwd": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "eemop_shuffle_channels": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "eemop_scale_flow": (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _c_float_p, ctypes.c_void_p]),
    "eemop_pool2_fwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eemop_pool2_bwd": (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "eemop_resize_ac_fwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "eemop_resize_ac_bwd": (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "eemop_local_corr53_bwd": (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 4 + [_c_float_p, _c_float_p, ctypes.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)
_lib = None


class EEMFlowHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EEMFlowHipError(
                f"{LIB_PATH} is missing: build it with `python -m eemflow_amd.build` (needs hipcc). "
                "There is no CPU fallback for the EEMFlow hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise EEMFlowHipError(lib().eemflow_last_error().decode("utf-8", "replace"))


def current_stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
