"""ctypes binding of libhpe_hip.so (include/hpe.h).  No torch types cross this boundary: only raw
device/host pointers, sizes and a hipStream_t.  The library must exist (``hpe_amd.build.build()``);
there is no CPU fallback -- a missing library or a non-gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhpe_hip.so")

NUM_CONV = 53
NUM_DENSE = 3
NUM_VERTS = 6890
THETA_DIM = 85
FEATURE_DIM = 2048


class HpeError(RuntimeError):
    pass


class HpeConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),  # sizeof(HpeConfig); hpe_config_init writes it, hpe_create checks it
        ("device", C.c_int),
        ("max_batch", C.c_int),
        ("num_stage", C.c_int),
        ("bn_eps", C.c_float),
        ("encoder_dtype", C.c_int),
        # plan options (-1 = default: environment variable, else built-in); see include/hpe.h
        ("n_streams", C.c_int),
        ("dual_gemm", C.c_int),
        ("stem_fused", C.c_int),
        ("wino_min_c", C.c_int),
        ("wino_min_items", C.c_int),
        ("wino_fused", C.c_int),
        ("wino_fused_min_hw", C.c_int),
        ("mesh_a2b", C.c_int),
        ("wino_f4", C.c_int),
        ("wino4_fused", C.c_int),
        ("bf16_p8", C.c_int),
        ("wino4_ksplit", C.c_int),
        ("chain_fuse", C.c_int),
        ("halo3", C.c_int),
    ]


PLAN_OPTIONS = ("n_streams", "dual_gemm", "stem_fused", "wino_min_c", "wino_min_items", "wino_fused", "wino_fused_min_hw", "mesh_a2b", "wino_f4", "wino4_fused", "bf16_p8", "wino4_ksplit", "chain_fuse", "halo3")


class HpeSmplModel(C.Structure):
    _fields_ = [
        ("v_template", C.c_void_p),
        ("shapedirs", C.c_void_p),
        ("posedirs", C.c_void_p),
        ("J_regressor", C.c_void_p),
        ("weights", C.c_void_p),
        ("kp_regressor", C.c_void_p),
        ("parents", C.c_void_p),
        ("num_kp", C.c_int),
    ]


OUTPUT_FIELDS = ("verts", "joints", "cams", "theta", "J_transformed", "kp2d", "verts2d", "Rs")


class HpeOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in OUTPUT_FIELDS]


_PROTOS = {
    "hpe_last_error": (C.c_char_p, []),
    "hpe_version": (C.c_char_p, []),
    "hpe_conv_layer_name": (C.c_char_p, [C.c_int]),
    "hpe_bn_layer_name": (C.c_char_p, [C.c_int]),
    "hpe_conv_layer_geometry": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "hpe_config_init": (None, [C.POINTER(HpeConfig)]),
    "hpe_create": (C.c_int, [C.POINTER(HpeConfig), C.POINTER(C.c_void_p)]),
    "hpe_destroy": (C.c_int, [C.c_void_p]),
    "hpe_load_smpl": (C.c_int, [C.c_void_p, C.POINTER(HpeSmplModel)]),
    "hpe_load_conv": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "hpe_load_dense": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_load_mean_theta": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpe_finalize": (C.c_int, [C.c_void_p]),
    "hpe_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(HpeOutputs), C.c_int, C.c_void_p]),
    "hpe_forward_pipelined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(HpeOutputs), C.c_int, C.c_void_p]),
    "hpe_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpe_tail": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(HpeOutputs), C.c_int, C.c_void_p]),
    "hpe_tail_stream": (C.c_void_p, [C.c_void_p]),
    "hpe_encoder": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_regress_stage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_smpl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(HpeOutputs), C.c_void_p]),
    "hpe_orth_proj": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_reproject_vertices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "hpe_preprocess_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "hpe_preprocess_u8_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                          C.c_void_p, C.c_void_p]),
    "hpe_get_original": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_float, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpe_kp_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_mesh_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_val_losses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_device_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpe_debug_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_debug_chain": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "hpe_debug_stem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_debug_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_debug_set_dbg": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpe_debug_maxpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_debug_avgpool": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_debug_joint_regress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hpe_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "hpe_get_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hpe_get_span_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "hpe_get_loss_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hpe_debug_set_loss_counter": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpe_get_conv_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every symbol of include/hpe.h.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HpeError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH
        )
    # The HIP runtime must be the one torch already mapped (same SONAME libamdhip64.so.7), so that torch
    # device pointers / streams are valid in our launches: import torch first if it is going to be used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - the library itself does not need torch
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    return sorted(_PROTOS)


def check(rc):
    if rc != 0:
        msg = load().hpe_last_error()
        raise HpeError("libhpe_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def f32(a):
    """contiguous float32 host array + its pointer (keeps the array alive via the return value)"""
    arr = np.ascontiguousarray(a, dtype=np.float32)
    return arr, arr.ctypes.data_as(C.c_void_p)
