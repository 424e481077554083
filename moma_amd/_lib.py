"""ctypes binding of libmoma_hip.so (the C ABI declared in include/moma_hip.h).

This is the binding a maintainer of the reference would add.  There is no fallback:
if the shared library is missing or a symbol is absent, importing a kernel raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOMA_HIP_LIB", os.path.join(_HERE, "lib", "libmoma_hip.so"))

PREC_F32, PREC_BF16 = 0, 1
DT_F32, DT_BF16 = 0, 1
MHA_SAVE_PROBS, MHA_SAVE_LSE = 0, 1
ABI_VERSION = 4
EMA_BLOCK_ELEMS = 4096

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_z = C.c_size_t

# symbol -> (restype, argtypes); must list every function of include/moma_hip.h
SIGNATURES = {
    "moma_version": (_i, []),
    "moma_error_string": (C.c_char_p, [_i]),
    "moma_debug_set_k2_target_wg": (_i, [_i]),
    "moma_ema_multi": (_i, [_p, _i, _l, _f, _f, _p]),
    "moma_enqueue": (_i, [_p, _p, _i, _l, _i, _i, _i, _p]),
    "moma_enqueue_mirror": (_i, [_p, _p, _p, _i, _l, _i, _i, _p]),
    "moma_queue_prefetch": (_i, [_p, _z, _p]),
    "moma_infonce_logits": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p]),
    "moma_infonce_logits_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p]),
    "moma_infonce_logits_bwd_workspace_bytes": (_z, [_i, _i, _i]),
    "moma_infonce_logits_bwd_ws": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p, _z, _p]),
    "moma_infonce_logits_bwd_kq": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "moma_infonce_fused_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "moma_infonce_fused": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _z, _i, _i, _p]),
    "moma_infonce_fused_ex": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p]),
    "moma_infonce_qpack_bytes": (_z, [_i, _i]),
    "moma_infonce_fused_q": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),
    "moma_infonce_fused_enqueue": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _z, _i, _i, _p, _i, _l, _p, _p, _p, _p, _p]),
    "moma_infonce_fused_multi_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "moma_infonce_fused_multi": (_i, [_p, _i, _i, _i, _i, _f, _p, _z, _i, _i, _p]),
    "moma_mha_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "moma_mha_saved_state": (_i, [_i, _i, _i, _i]),
    "moma_mha_pack_bytes": (_z, [_i]),
    "moma_mha_pack_weights": (_i, [_p, _p, _p, _i, _i, _p]),
    "moma_mha_fwd_fast": (_i, [_p, _i, _i, _i, _i, _p]),
    "moma_mha_bwd_fast_workspace_bytes": (_z, [_i, _i, _i]),
    "moma_mha_bwd_fast": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _p]),
    "moma_dwconv_workspace_bytes": (_z, [_i, _i]),
    "moma_dwconv_fwd": (_i, [_p, _p, _p] + [_i] * 11 + [_p]),
    "moma_dwconv_bwd_data": (_i, [_p, _p, _p] + [_i] * 11 + [_p]),
    "moma_dwconv_bwd_weight": (_i, [_p, _p, _p, _p, _z] + [_i] * 11 + [_p]),
    "moma_plane_mean": (_i, [_p, _p, _i, _i, _i, _p]),
    "moma_se_gate_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "moma_se_gate_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "moma_bn_workspace_bytes": (_z, [_i]),
    "moma_bn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p]),
    "moma_bn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _p, _p]),
    "moma_mha_bwd_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "moma_mha_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _p]),
}



class InfoNCETerm(C.Structure):
    """moma_infonce_term_t of include/moma_hip.h (one term of a moma_infonce_fused_multi call)."""
    _fields_ = [("q", _p), ("k", _p), ("queue", _p), ("loss_rows", _p), ("lse", _p), ("top1", _p), ("dq", _p)]


class MhaModule(C.Structure):
    """moma_mha_module_t of include/moma_hip.h (one module of a grouped moma_mha_fwd_fast call)."""
    _fields_ = [("x", _p), ("pack", _p), ("b_qkv", _p), ("b_proj", _p), ("y", _p), ("qkv16", _p), ("attn16", _p),
                ("lse", _p), ("qpack", _p), ("qpack_scale", _f), ("x_dtype", _i)]


_lib = None


class MomaHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise (never fall back) when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64.so.7; it must be the HIP runtime of the process (streams and
    # device pointers come from torch).  Importing torch first makes the loader bind our library to it.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MomaHipError(
            f"libmoma_hip.so not found at {LIB_PATH}: build it with `python -m moma_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the MoMA hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MomaHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.moma_version()
    if v != ABI_VERSION:
        raise MomaHipError(f"libmoma_hip ABI version {v} != binding version {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().moma_error_string(rc)
        raise MomaHipError(f"{what} failed: rc={rc} ({msg.decode() if msg else '?'})")
