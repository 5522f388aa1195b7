"""ctypes binding of libcapnet_hip.so (C ABI declared in include/capnet.h).

The library is the product: there is NO CPU fallback. If the shared object is missing, or a
call returns a non-zero status, this module raises -- it never routes around the HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcapnet_hip.so")

_f = C.POINTER(C.c_float)
_vp = C.c_void_p
_i = C.c_int
_l = C.c_long
_sz = C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol of include/capnet.h
SIGNATURES = {
    "capnet_last_error": (C.c_char_p, []),
    "capnet_abi_version": (_i, []),
    "capnet_sgemm": (_i, [_i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _l, _l, _l,
                          _l, _i, _vp]),
    "capnet_sgemm_b3": (_i, [_i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _i, _l, _l, _l, _l, _vp, _sz, _vp]),
    "capnet_sgemm_b3_eligible": (_i, [_i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _l, _l, _l, _l]),
    "capnet_sgemm_splitk": (_i, [_i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp, _sz, _vp]),
    "capnet_sgemm_splitk_fused": (_i, [_i, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp, _sz, _vp, _sz, _vp]),
    "capnet_colsum": (_i, [_vp, _l, _i, _i, _vp, _i, _vp]),
    "capnet_argmax_rows": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "capnet_resize_u8": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "capnet_crop_flip_normalize": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _vp]),
    "capnet_topk_correct": (_i, [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "capnet_beam_topk": (_i, [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "capnet_beam_topk_batched": (_i, [_vp, _l, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "capnet_att_step_fwd": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i,
                                 _vp, _vp, _l, _vp, _vp]),
    "capnet_trunk_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "capnet_trunk_destroy": (None, [_vp]),
    "capnet_trunk_workspace_bytes": (_sz, [_vp]),
    "capnet_trunk_num_convs": (_i, [_vp]),
    "capnet_trunk_final_side": (_i, [_vp]),
    "capnet_trunk_flops": (C.c_double, [_vp]),
    "capnet_trunk_conv_flops": (C.c_double, [_vp, _i]),
    "capnet_trunk_conv_shape": (_i, [_vp, _i, _ip, _ip, _ip, _ip, _ip]),
    "capnet_trunk_forward": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                  C.POINTER(_vp), C.POINTER(_vp), _i, C.c_float, C.c_float, _vp,
                                  _vp, _vp, _ip, _vp, _vp]),
    "capnet_trunk_set_tail_balance": (_i, [_vp, _i]),
    "capnet_trunk_update_running": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.c_float, _vp]),
    "capnet_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "capnet_adaptive_pool_replicate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "capnet_conv2d_fwd": (_i, [_vp, _l, _l, _l, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i,
                               _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "capnet_trunk_conv_kmajor": (_i, [_vp, _i]),
    "capnet_trunk_conv_tile_n": (_i, [_vp, _i]),
    "capnet_pack_conv_weight_kmajor": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "capnet_conv2d_fwd_kmajor": (_i, [_vp, _l, _l, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i,
                                      _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "capnet_sgemm_nt_dma_eligible": (_i, [_i, _i, _i, _vp, _l, _vp, _l, _vp, _l]),
    "capnet_sgemm_nt_dma": (_i, [_i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp]),
    "capnet_conv1x1_tiles_m": (_i, [_l]),
    "capnet_conv1x1_fwd_dma": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp,
                                    _vp, _i, _vp]),
    "capnet_conv1x1_f16x3_weight_words": (_sz, [_i, _i]),
    "capnet_conv1x1_f16x3_bn": (_i, [_l, _i]),
    "capnet_conv1x1_f16x3_pack": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "capnet_conv1x1_fwd_f16x3": (_i, [_vp, _l, _l, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i,
                                      _i, _vp, _vp, _vp, _i, _vp]),
    "capnet_conv3x3_fwd_patch": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "capnet_conv1x1_fwd_tail": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "capnet_conv2d_fwd_f16x3_scaled": (_i, [_vp, _l, _l, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i,
                                            _i, _i, _i, _i, _vp]),
    "capnet_conv3x3_fwd_patch_scaled": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "capnet_conv1x1_fwd_tail_scaled": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "capnet_conv_stem_f16x3_weight_words": (_sz, []),
    "capnet_conv_stem_f16x3_part_rows": (_i, [_i, _i, _i]),
    "capnet_conv_stem_f16x3_pack": (_i, [_vp, _vp, _vp]),
    "capnet_conv_stem_fwd_f16x3": (_i, [_vp, _l, _l, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "capnet_conv_f16x3_weight_words": (_sz, [_i, _i, _i]),
    "capnet_conv_f16x3_pack": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "capnet_conv2d_fwd_f16x3": (_i, [_vp, _l, _l, _l, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i,
                                     _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "capnet_conv_kmajor_slab_floats": (_sz, [_i, _i, _i, _i]),
    "capnet_conv_kmajor_plan": (None, [_i, _i, _i, _i, _ip]),
    "capnet_conv_kmajor_tiles_m": (_i, [_i, _i, _i, _i]),
    "capnet_conv_tiles_m": (_i, [_i, _i, _i]),
    "capnet_bn_finalize": (_i, [_vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp, C.c_float, C.c_float,
                                _vp, _vp, _vp]),
    "capnet_bn_add_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp]),
    "capnet_bn_relu_maxpool": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "capnet_global_avgpool": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "capnet_bn1d_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, C.c_float, C.c_float, _vp, _vp,
                             _vp, _vp]),
    "capnet_bn1d_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capnet_seq_saved_floats": (_sz, [_ip]),
    "capnet_seq_saved_ints": (_sz, [_ip]),
    "capnet_seq_fwd_scratch_floats": (_sz, [_ip]),
    "capnet_seq_bwd_scratch_floats": (_sz, [_ip]),
    "capnet_seq_forward": (_i, [_ip, _ip, C.POINTER(C.c_ubyte), _vp, _vp, _vp, C.POINTER(_vp),
                                _vp, _vp, C.c_float, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp,
                                _vp]),
    "capnet_seq_backward": (_i, [_ip, _ip, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), C.c_float,
                                 C.c_ulonglong, _i, _vp]),
    "capnet_seq_forward_stacked": (_i, [_ip, _i, _ip, _vp, _vp, _vp, _vp, C.POINTER(_vp), _vp, _vp, C.c_float, C.c_ulonglong, _i,
                                        C.POINTER(_vp), C.POINTER(_vp), _vp, C.POINTER(_vp), _vp, _vp]),
    "capnet_seq_backward_stacked": (_i, [_ip, _i, _ip, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp, C.POINTER(_vp),
                                         C.POINTER(_vp), C.c_float, C.c_ulonglong, _i, _vp]),
    "capnet_att_saved_floats": (_sz, [_ip]),
    "capnet_att_saved_ints": (_sz, [_ip]),
    "capnet_att_fwd_scratch_floats": (_sz, [_ip]),
    "capnet_att_bwd_scratch_floats": (_sz, [_ip]),
    "capnet_att_seq_forward": (_i, [_ip, _ip, C.POINTER(C.c_ubyte), _vp, _vp, _vp, C.POINTER(_vp),
                                    _vp, _vp, C.c_float, C.c_ulonglong, _i, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp]),
    "capnet_att_seq_backward": (_i, [_ip, _ip, _vp, _vp, _vp, _vp, C.POINTER(_vp), _vp, _vp, _vp,
                                     C.POINTER(_vp), C.c_float, C.c_ulonglong, _i, _vp]),
    "capnet_embedding_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "capnet_lstm_pointwise_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "capnet_lstm_wfrag_floats": (_sz, [_i]),
    "capnet_lstm_pack_wfrag": (_i, [_vp, _vp, _i, _i, _vp]),
    "capnet_lstm_step_fused": (_i, [_vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "capnet_lstm_step_fused_stamped": (_i, [_vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "capnet_lstm_step_fused_supported": (_i, [_i, _i]),
    "capnet_lstm_persist_supported": (_i, [_i, _i]),
    "capnet_lstm_persist_w_floats": (_sz, []),
    "capnet_lstm_persist_ctl_ints": (_sz, []),
    "capnet_lstm_persist_pack": (_i, [_vp, _vp, _i, _vp]),
    "capnet_lstm_persist_run": (_i, [_vp, _vp, _vp, _vp, _ip, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "capnet_xent_fwd": (_i, [_vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "capnet_xent_bwd": (_i, [_vp, _l, _i, _i, _vp, _vp, _vp, _vp, _l, _vp]),
    "capnet_att_loss_fwd": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _vp, _vp, _vp]),
    "capnet_att_loss_bwd": (_i, [_vp, _vp, _i, _i, _i, C.c_float, _vp, _vp]),
    "capnet_clamp_adam": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                               C.POINTER(_vp), C.POINTER(_l), _ip, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_float, _i, _vp, _vp]),
    "capnet_lstm_persist_set_mode": (_i, [_i]),
    "capnet_att_set_chain_mode": (_i, [_i]),
    "capnet_lstm_pointwise_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "capnet_conv1x1_fwd_areg": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _l, _i, _i, _i, _vp]),
    "capnet_trunk_set_timing": (_i, [_vp, _i]),
    "capnet_trunk_time_next_pass": (_i, [_vp]),
    "capnet_trunk_collect_timing": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_l),
                                         C.POINTER(C.c_double)]),
    "capnet_packed_targets": (_i, [_vp, _i, _i, _ip, _vp, _vp]),
    "capnet_fused_block_weight_words": (_sz, [_i, _i, _i]),
    "capnet_fused_block_pack": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "capnet_fused_block_stats_floats": (_sz, [_l, _i]),
    "capnet_fused_block_stats": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp,
                                      _vp, _vp]),
    "capnet_fused_block_tiles": (_i, [_l, _i]),
    "capnet_fused_block_forward": (_i, [_vp] * 14 + [_l, _i, _i, _i, _vp, _vp]),
    "capnet_comm_unique_id": (_i, [_vp]),
    "capnet_comm_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "capnet_comm_destroy": (_i, [_vp]),
    "capnet_allreduce_grads": (_i, [_vp, _vp, _l, _vp]),
    "capnet_err_word_exchange": (_i, [_vp, _vp, _i, _vp]),
    "capnet_count_skipped": (_i, [_vp, _vp, _vp]),
    "capnet_pack_tensors": (_i, [_i, C.POINTER(_vp), C.POINTER(_l), _vp, _i, C.c_float, _vp]),
    "capnet_clamp": (_i, [_vp, _l, C.c_float, C.c_float, _vp]),
}


class CapnetError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libcapnet_hip.so once; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CapnetError(
                "libcapnet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C image-caption-emotion-indonesia_amd/csrc`. There is no "
                "CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().capnet_last_error()
        raise CapnetError("%s failed (status %d): %s" %
                          (what or "capnet call", status, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = None if t is None else t.data_ptr()
    return arr


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
