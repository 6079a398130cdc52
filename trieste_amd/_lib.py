"""ctypes binding of libtgp.so (the C-ABI in include/tgp.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, creating an
engine raises.  (The numpy oracle under oracle/ is test infrastructure and is never imported
from this package.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGP_LIB") or os.path.join(_HERE, "libtgp.so")  # TGP_LIB: an experimental build (tools/)

TGP_OK, TGP_ERR_SHAPE, TGP_ERR_NOT_PD, TGP_ERR_ALLOC, TGP_ERR_HIP, TGP_ERR_STATE, TGP_ERR_ARG = range(7)
HOST, DEVICE = 0, 1
KERNELS = {"rbf": 0, "squared_exponential": 0, "matern12": 1, "matern32": 2, "matern52": 3}
PENALIZERS = {"none": 0, "soft": 1, "hard": 2}
ACQ = {"ei": 0, "pi": 1, "nlcb": 2, "aei": 3, "mes": 4, "gibbon": 5}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/tgp.h
SIGNATURES = {
    "tgp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tgp_destroy": (C.c_int, [_vp]),
    "tgp_last_error": (C.c_char_p, [_vp]),
    "tgp_version": (C.c_char_p, []),
    "tgp_set_stream": (C.c_int, [_vp, _vp]),
    "tgp_use_private_stream": (C.c_int, [_vp]),
    "tgp_set_hyper": (C.c_int, [_vp, C.c_double, _vp, C.c_double, C.c_double]),
    "tgp_set_data": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int]),
    "tgp_append_data": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int]),
    "tgp_clone_from": (C.c_int, [_vp, _vp]),
    "tgp_set_penalization": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int64]),
    "tgp_set_min_value_samples": (C.c_int, [_vp, _vp, C.c_int]),
    "tgp_set_repulsion": (C.c_int, [_vp, _vp, C.c_double]),
    "tgp_penalization_values": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int]),
    "tgp_get_sizes": (C.c_int, [_vp, _ip, C.POINTER(C.c_int)]),
    "tgp_nlml": (C.c_int, [_vp, _dp, _vp]),
    "tgp_nlml_trial": (C.c_int, [_vp, _dp]),
    "tgp_nlml_trial_batch": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "tgp_release_scratch": (C.c_int, [C.c_int]),
    "tgp_update_is_persistent": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int)]),
    "tgp_get_factor": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "tgp_predict": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_int]),
    "tgp_predict_mean": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int]),
    "tgp_predict_joint": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int]),
    "tgp_eta": (C.c_int, [_vp, _dp]),
    "tgp_acq_values": (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int64, _vp, C.c_int]),
    "tgp_acq_value_grad": (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int64, _vp, _vp, C.c_int]),
    "tgp_acq_argmax": (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int64, C.c_int64, _dp, _ip, _vp,
                                 C.c_int]),
    "tgp_acq_topk": (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp,
                               C.c_int]),
    "tgp_sample_box": (C.c_int, [_vp, C.c_uint64, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "tgp_qei": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp,
                          C.c_int]),
    "tgp_reparam_samples": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_double, _vp, C.c_int]),
    "tgp_traj_create": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "tgp_traj_create_rff": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.POINTER(_vp)]),
    "tgp_traj_get_theta": (C.c_int, [_vp, _vp]),
    "tgp_traj_destroy": (C.c_int, [_vp]),
    "tgp_traj_get_v": (C.c_int, [_vp, _vp]),
    "tgp_traj_eval": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int]),
    "tgp_traj_value_grad": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, C.c_int]),
    "tgp_sample_joint": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_double, _vp, C.c_int]),
    "tgp_cov_between": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int]),
    "tgp_joint_forward": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int]),
    "tgp_joint_vjp": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int]),
    "tgp_qei_value_grad": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp, _vp, C.c_int]),
    "tgp_traj_argmin": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int]),
    "tgp_acq_argmax_async": (C.c_int, [_vp, C.c_int, C.c_double, _vp, C.c_int64, C.c_int64, _vp]),
    "tgp_traj_argmin_async": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp]),
    "tgp_merge_winners_async": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "tgp_stream_synchronize": (C.c_int, [_vp]),
    "tgp_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "tgp_group_destroy": (C.c_int, [_vp]),
    "tgp_group_last_error": (C.c_char_p, [_vp]),
    "tgp_group_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tgp_group_member": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "tgp_group_set_hyper": (C.c_int, [_vp, C.c_double, _vp, C.c_double, C.c_double]),
    "tgp_group_set_data": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "tgp_group_append_data": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "tgp_group_set_candidates": (C.c_int, [_vp, _vp, C.c_int64]),
    "tgp_group_sample_candidates": (C.c_int, [_vp, C.c_uint64, C.c_int64, _vp, _vp]),
    "tgp_group_acq_argmax": (C.c_int, [_vp, C.c_int, C.c_double, _dp, _ip, _vp]),
    "tgp_group_acq_topk": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int, _vp, _vp]),
    "tgp_group_qei": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int, C.c_double, C.c_double, _vp]),
    "tgp_group_traj_create": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "tgp_group_traj_destroy": (C.c_int, [_vp]),
    "tgp_group_traj_argmin": (C.c_int, [_vp, _vp, _vp]),
    "tgp_group_last_kernel_ms": (C.c_int, [_vp, _dp]),
    "tgp_last_kernel_ms": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "tgp_set_variant": (C.c_int, [_vp, C.c_int]),
    "tgp_set_update_concurrency": (C.c_int, [_vp, C.c_int]),
    "tgp_set_precision": (C.c_int, [_vp, C.c_int]),
    "tgp_get_precision": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp]),
    "tgp_set_auto_sigma": (C.c_int, [_vp, C.c_double]),
    "tgp_get_auto_report": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp, C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "tgp_get_auto_strata": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp, C.POINTER(C.c_int64)]),
    "tgp_dag_plan": (C.c_int, [C.c_int, C.c_int64, _vp, C.c_int64, _ip, _ip, _vp, _vp, C.c_int]),
}

_lib = None


class TgpError(RuntimeError):
    """Any non-OK status from libtgp that has no closer Python analogue."""


class NotPositiveDefiniteError(TgpError, ArithmeticError):
    """Cholesky of K + noise*I (or of a q x q joint covariance) failed.  The reference surfaces
    this as tf.errors.InvalidArgumentError (trieste/models/gpflow/models.py:312-315)."""


def load():
    """Load libtgp.so (once).  Raises ImportError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP engine first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C trieste_amd/csrc).  trieste_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; importing it first makes libtgp bind to the SAME HIP
    # runtime instance so device pointers / streams can be shared with torch tensors.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


MERGES = {"rccl": 0, "peer": 1}
PRECISIONS = {"f64": 0, "i8x4": 1, "i8x5": 2, "auto": 3}


def check(lib, handle, rc, group=False):
    if rc == TGP_OK:
        return
    msg = (lib.tgp_group_last_error if group else lib.tgp_last_error)(handle)
    msg = msg.decode() if msg else ""
    if rc in (TGP_ERR_SHAPE, TGP_ERR_ARG):
        raise ValueError(msg)
    if rc == TGP_ERR_NOT_PD:
        raise NotPositiveDefiniteError(msg)
    if rc == TGP_ERR_ALLOC:
        raise MemoryError(msg)
    if rc == TGP_ERR_STATE:
        raise RuntimeError(msg)
    raise TgpError(f"libtgp status {rc}: {msg}")
