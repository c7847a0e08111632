"""ctypes binding of csrc/libttdg_mgm.so (the C ABI of include/ttdg_mgm.h).

No torch types cross the boundary: tensors are passed as raw device pointers
(``tensor.data_ptr()``) plus sizes, and every call is enqueued on torch's
current HIP stream.  There is NO fallback: a missing library, a CPU tensor or
a non-zero status raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libttdg_mgm.so")

GAGM_INFO_WORDS = 24      # TTDG_GAGM_INFO_WORDS (include/ttdg_mgm.h): int32 words ttdg_gagm_solve writes into `info`
MAX_GRAPHS = 64
MAX_LEVELS = 8
UNIV = 32


class Graphs(C.Structure):
    _fields_ = [("G", C.c_int32), ("off", C.c_int32 * (MAX_GRAPHS + 1))]


GAGM_LDS_PROJECTORS, GAGM_FORCE_LARGE, GAGM_FORCE_SINGLE, GAGM_256_THREADS, GAGM_COLUMN_PROJECTOR, GAGM_SCIPY_ORDER_LAP, GAGM_ONE_LAUNCH, GAGM_NO_INT_LAP = 1, 2, 4, 8, 16, 32, 64, 128    # ttdg_gagm_cfg_t.variant


class GagmCfg(C.Structure):
    _fields_ = [("tau0", C.c_float), ("gamma", C.c_float), ("min_tau", C.c_float), ("tol", C.c_float),
                ("quad_weight", C.c_float), ("max_iter", C.c_int32), ("sk_iter", C.c_int32),
                ("max_stages", C.c_int32), ("start_hungarian", C.c_int32), ("no_cycle_skip", C.c_int32), ("profile", C.c_int32),
                ("variant", C.c_int32)]


class Levels(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32 * MAX_LEVELS), ("w", C.c_int32 * MAX_LEVELS),
                ("stride", C.c_int32 * MAX_LEVELS), ("lo", C.c_float * MAX_LEVELS), ("hi", C.c_float * MAX_LEVELS)]


class Fpn(C.Structure):
    _fields_ = [("n", C.c_int32), ("C", C.c_int32), ("h", C.c_int32 * MAX_LEVELS), ("w", C.c_int32 * MAX_LEVELS),
                ("feat", C.c_void_p * MAX_LEVELS)]


class SgdTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("buf", C.c_void_p), ("n", C.c_int64), ("wd", C.c_float),
                ("first", C.c_int32)]


class GemmDesc(C.Structure):
    """ttdg_gemm_desc_t: one product of a grouped launch (strides in elements)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("bias", C.c_void_p), ("A2", C.c_void_p), ("B2", C.c_void_p), ("C", C.c_void_p),
                ("sam", C.c_int64), ("sak", C.c_int64), ("sbn", C.c_int64), ("sbk", C.c_int64), ("scm", C.c_int64), ("scn", C.c_int64),
                ("sam2", C.c_int64), ("sak2", C.c_int64), ("sbn2", C.c_int64), ("sbk2", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("K2", C.c_int32), ("alpha", C.c_float), ("beta", C.c_float)]


class RowScale(C.Structure):
    """ttdg_row_scale_t: one tensor of a multi-tensor row scaling."""
    _fields_ = [("inp", C.c_void_p), ("scale", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("rowlen", C.c_int32)]


class RpnLevel(C.Structure):
    """ttdg_rpn_level_t: one FPN level of the fused RPN selection."""
    _fields_ = [("logits", C.c_void_p), ("deltas", C.c_void_p), ("anchors", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32),
                ("k", C.c_int32), ("col0", C.c_int32)]


class Mm(C.Structure):
    """ttdg_mm_t: one streaming product with fused input activation / epilogue (csrc/pointwise.hip)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("bias2", C.c_void_p),
                ("pbias", C.c_void_p), ("ws", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldres", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("a_layout", C.c_int32), ("b_layout", C.c_int32),
                ("a_stride", C.c_int32), ("a_h", C.c_int32), ("a_w", C.c_int32),
                ("res_up", C.c_int32), ("res_h", C.c_int32), ("res_w", C.c_int32),
                ("relu", C.c_int32), ("prelu", C.c_int32), ("kslices", C.c_int32), ("tile", C.c_int32),
                ("A2", C.c_void_p), ("B2", C.c_void_p), ("lda2", C.c_int64), ("ldb2", C.c_int64), ("K2", C.c_int32),
                ("a2_stride", C.c_int32), ("a2_h", C.c_int32), ("a2_w", C.c_int32),
                ("C2", C.c_void_p), ("ldc2", C.c_int64), ("nsplit", C.c_int32)]


GEMM_GROUP_MAX = 8
ROW_SCALE_MAX = 64
RPN_LEVELS_MAX = 8
RPN_SELECT_MAX_K = 2048
_P, _I, _L, _F, _S = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_void_p

# name -> (restype, argtypes); one entry per symbol declared in include/ttdg_mgm.h
SIGNATURES = {
    "ttdg_version": (C.c_int, []),
    "ttdg_last_error": (C.c_char_p, []),
    "ttdg_gemm_f32": (C.c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _F, _F, _S]),
    "ttdg_gemm_splitk_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "ttdg_gemm_f32_splitk": (C.c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _F, _F, _I, _P, _S]),
    "ttdg_gemm_f32_grouped": (C.c_int, [C.POINTER(GemmDesc), _I, _S]),
    "ttdg_colsum_f32": (C.c_int, [_P, _L, _P, _I, _I, _S]),
    "ttdg_row_scale_multi": (C.c_int, [C.POINTER(RowScale), _I, _S]),
    "ttdg_affinity_pairwise_fwd": (C.c_int, [_P, _P, _P, _I, Graphs, _I, _P, _S]),
    "ttdg_affinity_bwd_workspace_bytes": (C.c_size_t, [_I, _I]),
    "ttdg_affinity_pairwise_bwd": (C.c_int, [_P, _P, _P, _P, _I, Graphs, _P, _P, _P, _P, _P, _S]),
    "ttdg_sinkhorn_pairs_fwd": (C.c_int, [_P, _I, _P, Graphs, _F, _I, _P, _P, _S]),
    "ttdg_sinkhorn_pairs_bwd": (C.c_int, [_P, _I, _P, _P, _P, Graphs, _F, _I, _P, _S]),
    "ttdg_pair_stage_fwd": (C.c_int, [_P, _P, _P, _P, _I, Graphs, _F, _I, _P, _P, _P, _S]),
    "ttdg_pair_stage_bwd": (C.c_int, [_P, _P, _P, _P, Graphs, _F, _I, _P, _S]),
    "ttdg_sinkhorn_batched_fwd": (C.c_int, [_P, _L, _L, _L, _I, _I, _I, _P, _P, _I, _F, _I, _P, _P, _S]),
    "ttdg_sinkhorn_batched_bwd": (C.c_int, [_P, _L, _L, _L, _I, _I, _I, _P, _P, _I, _F, _I, _P, _P, _P, _S]),
    "ttdg_mha_adjacency": (C.c_int, [_P, _P, _I, Graphs, _F, _F, C.c_uint64, _I, _P, _S]),
    "ttdg_gagm_workspace_bytes": (C.c_size_t, [_I]),
    "ttdg_gagm_solve": (C.c_int, [_P, _P, _P, Graphs, GagmCfg, _P, _P, _P, _S]),
    "ttdg_lap_batched": (C.c_int, [_P, _I, _I, _I, _P, _S]),
    "ttdg_debug_set_lap_variant": (C.c_int, [_I]),
    "ttdg_debug_project": (C.c_int, [_P, _I, _I, _F, _I, _I, _I, _P, _P, _S]),
    "ttdg_perm_loss_workspace_bytes": (C.c_size_t, [Graphs]),
    "ttdg_perm_loss_fwd_bwd": (C.c_int, [_P, _P, Graphs, _F, _F, _P, _P, _P, _P, _S]),
    "ttdg_node_labels": (C.c_int, [_P, _P, _P, _I, _I, Levels, _P, _S]),
    "ttdg_node_select": (C.c_int, [_P, _I, Levels, _I, _I, _P, _P, _P, _S]),
    "ttdg_node_gather_fwd": (C.c_int, [Fpn, _P, _P, _I, _P, _S]),
    "ttdg_node_gather_bwd": (C.c_int, [Fpn, _P, _P, _I, _P, _S]),
    "ttdg_node_gather_fwd_nhwc": (C.c_int, [Fpn, _P, _P, _I, _P, _S]),
    "ttdg_node_gather_bwd_nhwc": (C.c_int, [Fpn, _P, _P, _I, _P, _S]),
    "ttdg_sgd_multi_tensor": (C.c_int, [_P, _P, _P, _I, _I, _F, _F, _S]),
    "ttdg_roi_align_fwd": (C.c_int, [_P, _I, _I, _I, _I, _P, _I, _F, _I, _P, _S]),
    "ttdg_nms": (C.c_int, [_P, _P, _I, _F, _P, _P, _P, _S]),
    "ttdg_nms_grouped": (C.c_int, [_P, _P, _I, _I, _I, _F, _P, _P, _S]),
    "ttdg_rpn_decode": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _S]),
    "ttdg_rpn_select": (C.c_int, [C.POINTER(RpnLevel), _I, _I, _I, _P, _I, _P, _P, _S]),
    "ttdg_rpn_select_nhwc": (C.c_int, [C.POINTER(RpnLevel), _I, _I, _I, _P, _I, _P, _P, _S]),
    "ttdg_box_inference": (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _P, _P, _S]),
    "ttdg_roi_align_multilevel": (C.c_int, [Fpn, Levels, _P, _I, _I, _F, _I, _I, _P, _S]),
    "ttdg_debug_set_roi_align_sliced": (C.c_int, [_I]),
    "ttdg_resize_u8_workspace_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "ttdg_resize_bilinear_u8": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _S]),
    "ttdg_nchw_to_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _S]),
    "ttdg_roi_align_multilevel_nhwc": (C.c_int, [Fpn, Levels, _P, _I, _I, _F, _I, _I, _P, _S]),
    "ttdg_bias_act": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _S]),
    "ttdg_bias_act_nhwc": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _S]),
    "ttdg_relu_bwd": (C.c_int, [_P, _P, _P, C.c_size_t, _S]),
    "ttdg_mm_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "ttdg_mm_f32": (C.c_int, [C.POINTER(Mm), _S]),
    "ttdg_mm_f32_grouped": (C.c_int, [C.POINTER(Mm), _I, _S]),
    "ttdg_paste_masks": (C.c_int, [_P, _P, _I, _I, _I, _I, _F, _P, _S]),
    "ttdg_mask_pair_counts": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _S]),
    "ttdg_mask_measures": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, C.c_double, _P, _S]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ttdg_mgm_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def graphs(sizes):
    sizes = [int(s) for s in sizes]
    if not 1 <= len(sizes) <= MAX_GRAPHS:
        raise ValueError("between 1 and %d graphs are supported, got %d" % (MAX_GRAPHS, len(sizes)))
    g = Graphs()
    g.G = len(sizes)
    acc = 0
    for i, s in enumerate(sizes):
        if s <= 0:
            raise ValueError("empty graph")
        g.off[i] = acc
        acc += s
    g.off[len(sizes)] = acc
    return g


def ptr(t):
    """Device pointer of a tensor the kernels may touch: fp32/int32/int64/uint8 on a HIP device, contiguous unless
    the callee takes strides."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ttdg_mgm_amd operators run on the GPU only (got a %s tensor); no CPU fallback" % t.device)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.ttdg_last_error().decode()))


def check_f32(*ts):
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("expected contiguous float32 tensors")
