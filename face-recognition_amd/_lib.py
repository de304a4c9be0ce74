"""ctypes binding of libfr_hip.so (include/fr_hip.h).  The product path has NO fallback: if the
library is missing or a call fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfr_hip.so")

FR_PREC_BF16, FR_PREC_F32, FR_PREC_F24 = 0, 1, 2
FR_PLAN_NO_FUSE = 0x100
FR_OP_CONV, FR_OP_SE, FR_OP_LINEAR = 0, 1, 2
FR_ACT_NONE, FR_ACT_PRELU, FR_ACT_HSWISH, FR_ACT_RELU, FR_ACT_HSIGMOID = 0, 1, 2, 3, 4
PRECISIONS = {"bf16": FR_PREC_BF16, "f32": FR_PREC_F32, "f24": FR_PREC_F24}

_fp = C.POINTER(C.c_float)


class LayerDesc(C.Structure):
    _fields_ = [
        ("op", C.c_int32), ("src", C.c_int32), ("residual", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("kernel", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("groups", C.c_int32),
        ("act", C.c_int32),
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("bn_weight", C.c_void_p), ("bn_bias", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p),
        ("bn_eps", C.c_float),
        ("prelu", C.c_void_p), ("weight2", C.c_void_p), ("bias2", C.c_void_p),
    ]


_SIGS = {
    "fr_last_error": (C.c_char_p, []),
    "fr_plan_create": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p)]),
    "fr_plan_destroy": (C.c_int, [C.c_void_p]),
    "fr_plan_workspace_bytes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "fr_plan_layer_dims": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fr_plan_num_launches": (C.c_int, [C.c_void_p]),
    "fr_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fr_embed_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fr_embed_graph_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fr_embed_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fr_embed_graph_nodes": (C.c_int, [C.c_void_p]),
    "fr_embed_graph_destroy": (C.c_int, [C.c_void_p]),
    "fr_debug_fail_after": (C.c_int, [C.c_int]),
    "fr_align_faces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fr_embed_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                               C.c_int, C.c_void_p]),
    "fr_embed_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_float)]),
    "fr_plan_layer_kernel": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]),
    "fr_l2norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fr_l2norm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fr_cosine_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "fr_synth_rows": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_uint64, C.c_int64,
                                C.c_int64, C.c_void_p]),
    "fr_topk_workspace_bytes": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_size_t)]),
    "fr_topk_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fr_topk_search_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "fr_topk_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "fr_topk_merge_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib() -> C.CDLL:
    """Load libfr_hip.so (built by ``__graft_entry__.build()`` / ``make -C face-recognition_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; "
                "g.build()'); there is no CPU fallback for the hot path")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError -> a symbol of include/fr_hip.h is not exported
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fr_last_error()
        raise RuntimeError(f"{what} failed (fr_status {rc}): {msg.decode() if msg else ''}")
