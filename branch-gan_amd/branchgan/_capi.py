"""ctypes binding of libbranchgan_hip.so (the C-ABI declared in include/branchgan.h).

There is no fallback: if the library is missing or a call fails, this raises.  torch is used only for
device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbranchgan_hip.so")

BG_F32, BG_BF16 = 0, 1
MAX_TOPN, MAX_BLOCKED, MAX_SLOTS, MAX_GATHER = 64, 8, 64, 80
MAX_RANKS = 8
RL_DIAG_COUNT = 20

c_i64, c_i32, c_f32, c_vp, c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
c_f64 = ctypes.c_double


class AttnShape(ctypes.Structure):
    _fields_ = [("dtype", c_i32), ("B", c_i32), ("H", c_i32), ("NQ", c_i32), ("K", c_i32), ("Dh", c_i32), ("S_text", c_i32),
                ("n_slots", c_i32), ("max_path", c_i32), ("scale", c_f32),
                ("q_sb", c_i64), ("q_sh", c_i64), ("q_sn", c_i64), ("q_sk", c_i64),
                ("o_sb", c_i64), ("o_sh", c_i64), ("o_sn", c_i64), ("o_sk", c_i64),
                ("t_sb", c_i64), ("t_sh", c_i64), ("t_ss", c_i64),
                ("s_sb", c_i64), ("s_sh", c_i64), ("s_sk", c_i64),
                ("dq_sb", c_i64), ("dq_sh", c_i64), ("dq_sn", c_i64), ("dq_sk", c_i64),
                ("p_drop", c_f32), ("drop_call", ctypes.c_uint32), ("drop_state", c_vp),
                ("drop_row0", ctypes.c_uint32), ("drop_reserved", ctypes.c_uint32)]


class RowDropout(ctypes.Structure):
    _fields_ = [("p", c_f32), ("site", ctypes.c_uint32), ("row0", ctypes.c_uint32), ("text_rows", c_i32), ("rows_per_batch", c_i32),
                ("reserved", c_i32), ("drop_state", c_vp)]


class DiscLossShape(ctypes.Structure):
    _fields_ = [("B", c_i32), ("S", c_i32), ("N", c_i32), ("K", c_i32), ("n_update", c_i32), ("n_root", c_i32),
                ("max_child", c_i32)]


class RlLossShape(ctypes.Structure):
    _fields_ = [("B", c_i32), ("S", c_i32), ("N", c_i32), ("K", c_i32), ("wA", c_f32), ("wB", c_f32), ("wC", c_f32),
                ("ignore_input", c_i32), ("ignore_alt", c_i32), ("use_depth_mask", c_i32), ("probs_input", c_i32)]


class ReduceDesc(ctypes.Structure):
    _fields_ = [("partial", c_vp), ("out", c_vp), ("groups", c_i32), ("stride", c_i32), ("n", c_i32), ("reserved", c_i32)]


class StackDesc(ctypes.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("rows", c_i64), ("row_bytes", c_i64), ("src_row_stride", c_i64)]


# name -> (restype, argtypes); every symbol include/branchgan.h declares
SIGNATURES = {
    "bg_last_error": (ctypes.c_char_p, []),
    "bg_version": (c_i32, []),
    "bg_build_id": (ctypes.c_char_p, []),
    "bg_device_arch": (c_i32, [ctypes.c_char_p, c_sz]),
    "bg_topn_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "bg_topn_select": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_i32, c_f32, c_i32, c_vp, c_vp,
                               c_vp, c_sz, c_vp]),
    "bg_topn_select_static": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_i32, c_f32, c_i32, c_vp, c_vp,
                                      c_vp]),
    "bg_topp_workspace_bytes": (c_sz, [c_i64, c_i64, c_i32]),
    "bg_topp_select": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i32, c_f32, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "bg_softmax_stats_fwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "bg_softmax_stats_bwd": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "bg_rank_stats": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, ctypes.POINTER(c_i32), c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "bg_branch_attn_fwd": (c_i32, [ctypes.POINTER(AttnShape), c_vp, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                   c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "bg_branch_attn_fwd_workspace_bytes": (c_sz, [ctypes.POINTER(AttnShape)]),
    "bg_branch_attn_fwd_ws": (c_i32, [ctypes.POINTER(AttnShape), c_vp, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "bg_branch_attn_bwd": (c_i32, [ctypes.POINTER(AttnShape), c_vp, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                   c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                   ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp, c_vp]),
    "bg_branch_attn_weights": (c_i32, [ctypes.POINTER(AttnShape), c_vp, c_vp, ctypes.POINTER(c_vp), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "bg_attn_dropout_mask": (c_i32, [ctypes.POINTER(AttnShape), c_vp, c_vp]),
    "bg_dropout_advance": (c_i32, [c_vp, c_vp]),
    "bg_dropout_rows": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i32, ctypes.c_uint32, c_f32, ctypes.c_uint32, c_vp, c_vp]),
    "bg_neox_qkv_rope_fwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "bg_neox_qkv_rope_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "bg_add_layernorm_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "bg_add_layernorm_pair_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "bg_layernorm_bwd_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "bg_layernorm_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_sz,
                                 c_vp]),
    "bg_add_layernorm_dropout_fwd": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                             ctypes.POINTER(RowDropout), c_vp]),
    "bg_layernorm_dropout_bwd": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp,
                                         c_sz, ctypes.POINTER(RowDropout), c_vp]),
    "bg_layernorm_bwd_groups": (c_i32, [c_i64]),
    "bg_bias_act_bwd_groups": (c_i32, [c_i64]),
    "bg_partial_reduce": (c_i32, [ctypes.POINTER(ReduceDesc), c_i32, c_vp]),
    "bg_stack_rows": (c_i32, [ctypes.POINTER(StackDesc), c_i32, c_vp]),
    "bg_bias_act_bwd_workspace_bytes": (c_sz, [c_i64, c_i32]),
    "bg_bias_act_bwd": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "bg_embed_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "bg_embed_bwd_workspace_bytes": (c_sz, [c_i64]),
    "bg_embed_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "bg_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_f64, c_f64, c_f64, c_f64, c_f64, c_i64, c_f64, c_i32, c_vp]),
    "bg_disc_loss_fwd": (c_i32, [ctypes.POINTER(DiscLossShape)] + [c_vp] * 9 + [c_vp, c_vp]),
    "bg_disc_loss_bwd": (c_i32, [ctypes.POINTER(DiscLossShape)] + [c_vp] * 9 + [c_vp] + [c_vp] * 5 + [c_vp]),
    "bg_rl_loss_fwd": (c_i32, [ctypes.POINTER(RlLossShape)] + [c_vp] * 7 + [c_vp, c_vp, c_vp]),
    "bg_dual_head_advantages": (c_i32, [c_vp] * 4 + [c_i32] * 3 + [c_f32] * 3 + [c_i32] * 2 + [c_f32] * 2 + [c_vp] * 6 + [c_vp]),
    "bg_reinforce_terms_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp]),
    "bg_reinforce_terms_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp]),
    "bg_graph_node_census": (c_i32, [c_vp] + [ctypes.POINTER(c_i32)] * 4),
    "bg_graph_replace_memset_nodes": (c_i32, [c_vp, ctypes.POINTER(c_i32)]),
    "bg_rl_loss_bwd": (c_i32, [ctypes.POINTER(RlLossShape)] + [c_vp] * 7 + [c_vp, c_vp, c_vp, c_vp]),
}

_lib = None


def source_build_id():
    """sha256 over the library's sources in the order csrc/Makefile uses (None when the sources are not beside the .so)."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    files = (sorted(glob.glob(os.path.join(csrc, "*.hip"))) + sorted(glob.glob(os.path.join(csrc, "*.h")))
             + [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "branchgan.h"), os.path.join(csrc, "Makefile")])
    if not all(os.path.exists(f) for f in files) or len(files) < 4:
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:32]


def lib():
    """Loads the shared library once.  Raises if it is not built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU or PyTorch fallback for the HIP kernels)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:                  # calling it later raises AttributeError; tests/test_capi.py is strict
                continue
            fn.restype, fn.argtypes = res, args
        want = source_build_id()
        have = handle.bg_build_id().decode() if hasattr(handle, "bg_build_id") else "<none>"
        if want is not None and have != want and os.environ.get("BRANCHGAN_ALLOW_STALE_LIB") != "1":
            raise RuntimeError(f"{LIB_PATH} is stale: built from sources {have}, the sources beside it hash to {want}; "
                               "rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = handle
    return _lib


def exported_symbols():
    """Names from SIGNATURES the built library really exports (used by the no-GPU load test)."""
    handle = ctypes.CDLL(LIB_PATH)
    return [n for n in SIGNATURES if hasattr(handle, n)]


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed (code {code}): {lib().bg_last_error().decode()}")


def ptr(t):
    return None if t is None else c_vp(t.data_ptr())


def stream_ptr():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return BG_F32
    if t.dtype == torch.bfloat16:
        return BG_BF16
    raise TypeError(f"unsupported dtype {t.dtype}: the HIP kernels take float32 or bfloat16")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("branchgan HIP ops need tensors on an MI355X (cuda device); there is no CPU path")
