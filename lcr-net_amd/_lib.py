"""ctypes binding of liblcr_hip.so (C ABI: include/lcr_hip.h).  Loud failure if the library is absent."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "liblcr_hip.so")

_lib = None

c_f32p = ctypes.c_void_p
c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_p = ctypes.POINTER(ctypes.c_size_t)
c_size_t = ctypes.c_size_t

_SIGS = {
    "lcr_last_error": (ctypes.c_char_p, []),
    "lcr_version": (c_int, []),
    "lcr_support_grid_ws_bytes": (c_int, [c_i64, c_int, c_size_p]),
    "lcr_support_grid_build": (c_int, [c_vp, c_vp, c_int, c_i64, c_float, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_precompute_layout": (c_int, [c_i64, c_int, c_int, c_vp, c_int, c_i64, c_vp]),
    "lcr_precompute_batch": (c_int, [c_vp, c_vp, c_vp, c_float, c_float, c_float, c_int, c_vp, c_size_t, c_vp, c_size_t, c_vp, c_vp, c_vp]),
    "lcr_precompute_batch_rows": (c_int, [c_vp, c_int, c_vp, c_vp, c_float, c_float, c_float, c_int, c_vp, c_size_t, c_vp, c_size_t, c_vp, c_vp, c_vp]),
    "lcr_radius_query_multi": (c_int, [c_vp, c_int, c_int, c_vp]),
    "lcr_radius_query": (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_float, c_int, c_vp, c_vp, c_vp, c_vp]),
    "lcr_stream_spin": (c_int, [c_int, c_vp]),
    "lcr_radius_query_ordered": (c_int, [c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_float, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "lcr_radius_search_ws_bytes": (c_int, [c_i64, c_i64, c_int, c_size_p]),
    "lcr_radius_search": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_float, c_int, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, ctypes.c_size_t, c_vp]),
    "lcr_grid_subsample_ws_bytes": (c_int, [c_i64, c_int, c_size_p]),
    "lcr_grid_subsample": (c_int, [c_vp, c_vp, c_int, c_i64, c_float, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_grid_subsample_ex": (c_int, [c_vp, c_vp, c_int, c_i64, c_float, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_grid_subsample_rows": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_float, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_hashmap_order_host": (c_int, [c_vp, c_i64, c_vp]),
    "lcr_gemm_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "lcr_gemm_f32_anorm": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_float, c_float, c_vp, c_int,
                                   c_int, c_vp, c_vp]),
    "lcr_ktimer_enable": (None, [c_int]),
    "lcr_ktimer_sample": (None, [c_int]),
    "lcr_ktimer_kinds": (None, [ctypes.c_uint]),
    "lcr_ktimer_read": (c_int, [c_int, c_int, c_vp, c_vp]),
    "lcr_ktimer_read2": (c_int, [c_int, c_int, c_vp, c_vp, c_vp]),
    "lcr_encoder_ws_bytes": (c_int, [c_vp, c_vp, c_int, c_size_p]),
    "lcr_encoder_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_size_t, c_vp]),
    "lcr_roformer_ws_bytes": (c_int, [c_vp, c_i64, c_size_p]),
    "lcr_roformer_forward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_size_t, c_vp]),
    "lcr_encoder_forward_ex": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_uint, c_vp, c_size_t, c_vp]),
    "lcr_kpconv_aggregate": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_vp, c_float, c_vp, c_vp, c_vp, c_vp]),
    "lcr_kpconv_fused": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_int,
                                 c_int, c_vp, c_vp, c_vp]),
    "lcr_kpconv_cin1": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_float, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "lcr_maxpool": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "lcr_support_grid_order": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "lcr_row_positive": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "lcr_groupnorm_stats": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "lcr_groupnorm_apply": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_int,
                                    c_float, c_float, c_int, c_vp, c_vp]),
    "lcr_rotary_embed": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp]),
    "lcr_attention_f32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp]),
    "lcr_attention_seg_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "lcr_attention_topk_f32": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "lcr_add_layernorm": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_float, c_vp, c_vp]),
    "lcr_relu_inplace": (c_int, [c_vp, c_i64, c_vp]),
    "lcr_retrieval_ws_bytes": (c_int, [c_i64, c_i64, c_size_p]),
    "lcr_retrieval_topk": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_gemm_f32_strided_batched": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_int, c_vp]),
    "lcr_vote_shift": (c_int, [c_vp, c_vp, c_i64, c_float, c_vp, c_vp]),
    "lcr_greedy_nms_ws_bytes": (c_int, [c_i64, c_size_p]),
    "lcr_greedy_nms": (c_int, [c_vp, c_vp, c_int, c_i64, c_float, c_vp, c_vp, c_vp, c_vp]),
    "lcr_neighbor_mean": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_i64, c_vp, c_vp]),
    "lcr_point_to_node_ws_bytes": (c_int, [c_i64, c_int, c_size_p]),
    "lcr_point_to_node_partition": (c_int, [c_vp, c_i64, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_point_to_node_partition_stack": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_build_padded_scores": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_float, c_vp, c_float, c_vp, c_vp]),
    "lcr_log_sinkhorn": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_float, c_vp, c_vp]),
    "lcr_log_sinkhorn_ws_floats": (c_int, [c_i64, c_int, c_int, c_size_p]),
    "lcr_log_sinkhorn_form": (c_int, [c_i64, c_int, c_int, ctypes.POINTER(c_int)]),
    "lcr_split_bf16x3": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "lcr_split_bf16x3_tiles": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "lcr_gemm_f32_bsplit": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "lcr_log_sinkhorn_ex": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_float, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_top1_matching_ws_bytes": (c_int, [c_i64, c_int, c_int, c_size_p]),
    "lcr_top1_matching": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_top1_matching_ex": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_topk_matching_ws_bytes": (c_int, [c_i64, c_int, c_int, c_size_p]),
    "lcr_topk_matching": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "lcr_patch_scores": (c_int, [c_vp, c_i64, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_float, c_vp, c_float, c_vp, c_vp]),
    "lcr_topk_matching_ex": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t,
                                     c_vp]),
    "lcr_local_global_registration_ex": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_float, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, ctypes.c_size_t, c_vp]),
    "lcr_upsample_concat": (c_int, [c_vp, c_i64, c_int, c_vp, c_int, c_int, c_vp, c_int, c_i64, c_vp, c_vp]),
    "lcr_gather_rows": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_vp]),
    "lcr_procrustes_batched": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_float, c_vp, c_vp]),
    "lcr_inlier_count": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_float, c_vp, c_int, c_vp, c_vp, c_vp]),
    "lcr_inlier_weights": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_float, c_vp, c_vp]),
    "lcr_lgr_ws_bytes": (c_int, [c_i64, c_int, c_int, c_size_p]),
    "lcr_local_global_registration": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_float, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              ctypes.c_size_t, c_vp]),
    "lcr_netvlad_ws_bytes": (c_int, [c_i64, c_int, c_size_p]),
    "lcr_netvlad_forward": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
}


# entry points that synchronise, compute on the host or issue long launch sequences: called with the interpreter lock RELEASED
_RELEASES_GIL = {"lcr_roformer_forward", "lcr_precompute_batch", "lcr_precompute_batch_rows", "lcr_encoder_forward", "lcr_encoder_forward_ex", "lcr_ktimer_read",
                 "lcr_ktimer_read2", "lcr_hashmap_order_host", "lcr_netvlad_forward", "lcr_log_sinkhorn", "lcr_log_sinkhorn_ex",
                 "lcr_local_global_registration", "lcr_local_global_registration_ex", "lcr_retrieval_topk", "lcr_stream_spin"}


def build():
    """Compile the library in-tree (hipcc, gfx950)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_lcr_build", os.path.join(_PKG, "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def lib():
    """The loaded library.  Raises RuntimeError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liblcr_hip.so not found at %s — build it with `python lcr-net_amd/csrc/build.py` "
                "(there is no CPU fallback for the lcr-net_amd hot path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if not hasattr(L, name):
                continue  # symbol presence is enforced by tests/test_cabi_symbols.py against include/lcr_hip.h
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("LCR_CTYPES_KEEP_GIL", "1") != "0":
            # Entry points that only enqueue a few launches (5-10 us) are bound a second time through PyDLL, i.e. called WITHOUT releasing
            # the interpreter lock: with two host threads each issuing ~1000 such calls per registration pair, the release / re-acquire
            # around every call turned into a lock hand-off per launch (a futex wake each) and two workers ran SLOWER than one (100 vs
            # 136 pairs/s at one pair per call, profiles/r06_pair_host_profile.log).  Calls that block or issue long launch sequences
            # (_RELEASES_GIL) keep releasing it — those are what the pipelines' threads overlap on.
            K = ctypes.PyDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                if name in _RELEASES_GIL or not hasattr(K, name):
                    continue
                fn = getattr(K, name)
                fn.restype = res
                fn.argtypes = args
                setattr(L, name, fn)
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().lcr_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a tensor (or NULL for None)."""
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` as a void pointer.  Straight from the C binding: torch.cuda.current_stream
    builds a Stream object per call (4.5 us, ~200 calls per registration pair)."""
    idx = device.index if isinstance(device, torch.device) else None
    if idx is None:
        idx = torch.cuda.current_device() if not isinstance(device, int) else device
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lcr-net_amd ops run on the GPU only: got a %s tensor (no CPU fallback)" % t.device)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
