"""ctypes binding of libposegen_hip.so (include/posegen_hip.h).

north_star asks for a "thin C-ABI cffi layer"; cffi is not installed in the build
image (and cannot be: no network), so the same C ABI is bound with ctypes from the
standard library.  Nothing here computes: it declares prototypes, loads the library
and turns negative return codes into exceptions.  There is NO CPU fallback -- if
the library is missing every entry point raises `HipLibraryError`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libposegen_hip.so")

PG_OK, PG_EINVAL, PG_ENOMEM, PG_EHIP, PG_ESTATE = 0, -1, -2, -3, -4
PG_FLAG_LINDISP = 1
PG_ACT_RELU, PG_ACT_SOFTPLUS = 0, 1
PG_ABI_VERSION = 10


class HipLibraryError(RuntimeError):
    """libposegen_hip.so is missing or unusable (no CPU fallback exists)."""


class PgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"posegen_hip error {code}: {msg}")
        self.code = code


class PgConfig(C.Structure):
    _fields_ = [("n_joints", C.c_int32), ("multires", C.c_int32), ("multires_views", C.c_int32),
                ("multires_bones", C.c_int32), ("net_depth", C.c_int32), ("net_width", C.c_int32),
                ("skip_layer", C.c_int32), ("view_width", C.c_int32), ("framecode_ch", C.c_int32),
                ("n_framecodes", C.c_int32), ("chunk", C.c_int32), ("precision", C.c_int32),
                ("cutoff_dist", C.c_float), ("density_scale", C.c_float), ("rgb_eps", C.c_float),
                ("softplus_shift", C.c_float), ("density_act", C.c_int32), ("reserved0", C.c_int32)]


_FP = C.c_void_p  # device float*


class PgTrainDraws(C.Structure):
    """pg_train_draws: the caller's random numbers of one training-mode call."""
    _fields_ = [(k, _FP) for k in ("t_rand", "u_rand", "noise0", "noise1", "ray_noise")]


class PgNetParams(C.Structure):
    """pg_net_params: device pointers of one net's 24 tensors (+ frame codes with the mean row appended)."""
    _fields_ = [("w", _FP * 24), ("codes", _FP), ("n_codes", C.c_int32)]


class PgNetGrads(C.Structure):
    _fields_ = [("w", _FP * 24), ("codes", _FP)]


class PgOutputs(C.Structure):
    _fields_ = [(k, _FP) for k in ("rgb_map", "disp_map", "acc_map", "alpha", "rgb0", "disp0", "acc0",
                                   "alpha0", "near_far", "z_coarse", "z_fine", "raw_coarse", "raw_fine",
                                   "weights0")]


# every symbol include/posegen_hip.h declares: (restype, argtypes)
PROTOTYPES = {
    "pg_abi_version": (C.c_int, []),
    "pg_create": (C.c_int, [C.POINTER(PgConfig), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "pg_destroy": (None, [C.c_void_p]),
    "pg_last_error": (C.c_char_p, [C.c_void_p]),
    "pg_load_weights": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int]),
    "pg_set_embedder": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float]),
    "pg_set_framecodes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "pg_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_set_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_set_far_skip": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_set_onchip": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_load_weights_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]),
    "pg_set_train_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_render_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _FP, _FP, C.c_int64, _FP, C.c_int64, _FP,
                                 C.c_int, C.c_int, C.c_int, C.POINTER(PgOutputs)]),
    "pg_calibrate_mfma": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pg_render_rays_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _FP, _FP, C.c_int64, _FP, C.c_int64, _FP,
                                       C.c_int, C.c_int, C.c_int, C.POINTER(PgTrainDraws), C.POINTER(PgOutputs)]),
    "pg_stage_sample_coarse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _FP, _FP, C.c_int64, C.c_int,
                                         C.c_int, _FP, _FP]),
    "pg_stage_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, _FP, _FP, _FP,
                                C.c_int64, _FP, _FP, _FP, C.c_int]),
    "pg_stage_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, _FP, _FP, _FP, _FP, _FP, _FP,
                                     _FP, _FP, C.c_int, _FP]),
    "pg_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "pg_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pg_profile_read_aux": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pg_debug_pack": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int32)]),
    "pg_render_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.POINTER(C.c_int), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "pg_pose_kinematics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_double),
                                     C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_query_density": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_debug_pack_map": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "pg_debug_pack_vy": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int64, C.POINTER(C.c_int64)]),
    "pg_device_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pg_query": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pg_device_count": (C.c_int, [C.c_void_p]),
    "pg_pose_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                C.c_void_p]),
    "pg_render_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_render_frame_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_int), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_compose_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pg_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _FP, _FP, C.c_int64, _FP, C.c_int64, _FP, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(PgTrainDraws), C.POINTER(PgNetParams), C.POINTER(PgNetParams), C.POINTER(PgOutputs),
                                   C.POINTER(C.c_int64)]),
    "pg_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _FP, _FP, _FP, _FP, C.POINTER(PgNetGrads), C.POINTER(PgNetGrads)]),
    "pg_plan_frames": (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                 C.POINTER(C.c_int)]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load (once) and type the shared library; raises HipLibraryError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("POSEGEN_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise HipLibraryError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C posegen_amd/csrc` (there is no CPU fallback for the render path)")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # pragma: no cover - depends on the box
        raise HipLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.pg_abi_version() != PG_ABI_VERSION:
        raise HipLibraryError(f"{p}: ABI version {lib.pg_abi_version()} != {PG_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, handle, rc: int):
    if rc != PG_OK:
        msg = lib.pg_last_error(handle)
        raise PgError(rc, msg.decode() if msg else "unknown error")
