"""ctypes binding of libpcgmix_hip.so (the C ABI declared in include/pcgmix_hip.h).

There is no fallback: if the library is missing or an entry point fails, the caller gets
an exception.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C <package>/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpcgmix_hip.so")
ABI_VERSION = 17

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/pcgmix_hip.h one to one
SIGNATURES = {
    "pcgmix_abi_version": (_c_int, []),
    "pcgmix_error_string": (ctypes.c_char_p, [_c_int]),
    "pcgmix_spline_operator_size": (_c_int, [_c_int]),
    "pcgmix_spline_operator_f64": (_c_int, [_c_int, _c_int, _ptr]),
    "pcgmix_partner_permutation_i64": (_c_int, [_ptr, _c_int, _c_int, ctypes.c_uint64, _ptr]),
    "pcgmix_py_uniform01": (ctypes.c_double, [ctypes.c_uint64]),
    "pcgmix_py_randint0": (ctypes.c_int64, [ctypes.c_uint64, ctypes.c_int64]),
    "pcgmix_np_seed": (_c_int, [_ptr, ctypes.c_uint32]),
    "pcgmix_np_beta": (_c_int, [_ptr, ctypes.c_double, ctypes.c_double, _ptr]),
    "pcgmix_np_normal_fill": (_c_int, [_ptr, ctypes.c_double, ctypes.c_double, ctypes.c_longlong, _ptr]),
    "pcgmix_npdraw_create": (_c_int, [_ptr, _c_int]),
    "pcgmix_npdraw_destroy": (None, [_ptr]),
    "pcgmix_npdraw_step": (_c_int, [_ptr, ctypes.c_uint32, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_longlong, _ptr, _ptr, _ptr, _ptr]),
    "pcgmix_npdraw_stats": (ctypes.c_longlong, [_ptr, _ptr]),
    "pcgmix_pack_plan_i32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr]),
    "pcgmix_mix_warp_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _ptr, _ptr, _c_int,
                                     _ptr, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_saliency_post_f32": (_c_int, [_ptr, _ptr, _ptr, _c_int, ctypes.c_double, _c_int, _c_int,
                                          _c_int, _ptr]),
    "pcgmix_saliency_post2d_f32": (_c_int, [_ptr, _ptr, _ptr, _c_int, ctypes.c_double, _c_int, _c_int,
                                            _c_int, _ptr]),
    "pcgmix_potes_saliency_pass_f32": (_c_int, [_ptr] * 18 + [_c_int, ctypes.c_double, _c_int, _c_int, _c_int, _c_int,
                                                 _ptr]),
    "pcgmix_salopt_workspace_bytes": (ctypes.c_longlong, [_c_int]),
    "pcgmix_salopt_disp_f32": (_c_int, [_ptr, _ptr, _ptr, _c_float, _c_int, _ptr, _ptr, _c_int, _c_int,
                                        _c_int, _ptr]),
    "pcgmix_salopt_disp_hosted_f32": (_c_int, [_ptr, _ptr, _ptr, _c_float, _c_int, _ptr, _ptr, _c_int, _c_int,
                                               _c_int, _ptr, _ptr, _ptr]),
    "pcgmix_salopt_plan": (_c_int, [_ptr, _ptr, _c_int, _c_int, _c_int, _ptr, _c_int]),
    "pcgmix_salopt_mix_warp_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_int, _ptr, _ptr,
                                            _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_potes_head_saliency_f32": (_c_int, [_ptr] * 8 + [_c_int, _c_int, _c_int, _ptr]),
    "pcgmix_logmel_hostframes_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int,
                                              _c_float, _c_float, _c_int, _c_int, _ptr]),
    "pcgmix_logmel_tables_size": (ctypes.c_longlong, [_c_int, _c_int]),
    "pcgmix_logmel_tables": (_c_int, [_c_int, _c_int, _c_float, _c_float, _c_float, _ptr]),
    "pcgmix_logmel_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int,
                                   _c_int, _c_float, _c_float, _c_int, _c_int, _ptr]),
    "pcgmix_logmel_tile_frames": (_c_int, []),
    "pcgmix_logmel_recordings_f32": (_c_int, [_ptr, _ptr, _ptr, _c_int, _ptr, _c_int, _ptr, _c_int, _ptr,
                                              _ptr, ctypes.c_longlong, _ptr, _ptr, _c_int, _c_int,
                                              _c_int, _c_float, _c_float, _c_int, _c_int, _ptr]),
    "pcgmix_potes_out_len": (_c_int, [_c_int]),
    "pcgmix_potes_bwd_blocks": (_c_int, [_c_int, _c_int]),
    "pcgmix_potes_stack_fwd_f32": (_c_int, [_ptr] * 6 + [_c_int, _c_int, _ptr]),
    "pcgmix_potes_stack_bwd_f32": (_c_int, [_ptr] * 8 + [_c_int, _c_int, _ptr]),
    "pcgmix_potes_stack_input_grad_f32": (_c_int, [_ptr] * 7 + [_c_int, _c_int, _ptr]),
    "pcgmix_potes_mask_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "pcgmix_potes_stack_fwd_save_f32": (_c_int, [_ptr] * 8 + [_c_int, _c_int, _ptr, ctypes.c_longlong,
                                                 _ptr, ctypes.c_uint64, _ptr]),
    "pcgmix_potes_stack_bwd_mask_f32": (_c_int, [_ptr] * 9 + [_c_int, _c_int, _ptr]),
    "pcgmix_potes_stack_input_grad_mask_f32": (_c_int, [_ptr] * 6 + [_c_int, _c_int, _ptr]),
    "pcgmix_skinny_linear_splits": (_c_int, [_c_int, _c_int]),
    "pcgmix_skinny_linear_fwd_f32": (_c_int, [_ptr] * 5 + [_c_int, _c_int, _c_int, _ptr]),
    "pcgmix_adam_clip_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, ctypes.c_longlong, _c_float, _c_float,
                                      _c_float, _c_float, _c_float, _c_float, ctypes.c_longlong,
                                      _ptr]),
    "pcgmix_adam_clip_multi_f32": (_c_int, [_c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float,
                                            _c_float, _c_float, _c_float, _c_float,
                                            ctypes.c_longlong, _ptr]),
    "pcgmix_adam_hyper": (_c_int, [_c_float] * 6 + [ctypes.c_longlong, _ptr]),
    "pcgmix_adam_clip_multi_dev_f32": (_c_int, [_c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "pcgmix_adam_clip_multi_reduce_dev_f32": (_c_int, [_c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                       _c_int, _ptr]),
    "pcgmix_potes_reduce_f32": (_c_int, [_ptr, _ptr, _c_int, _ptr]),
    "pcgmix_potes_head_fwd_f32": (_c_int, [_ptr, _ptr, _c_float, _c_int, _c_int, _ptr, _ptr, _ptr, _c_float,
                                           _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int,
                                           _ptr]),
    "pcgmix_potes_head_bwd_f32": (_c_int, [_ptr, _ptr, _ptr, _c_float, _c_int, _ptr, _ptr, _ptr, _c_float,
                                           _c_int, _c_int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                           _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_potes_head_loss_workspace_floats": (ctypes.c_longlong, [_c_int]),
    "pcgmix_potes_head_loss_fwd_f32": (_c_int, [_ptr, _ptr, _c_float, _c_int, _c_int, _ptr, _ptr, _ptr,
                                                _c_float, _c_int, _ptr, _ptr, _ptr] + [_ptr] * 8 +
                                       [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_potes_head_loss_bwd_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_float, _c_int, _c_int, _ptr, _ptr,
                                                _ptr, _ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_soft_ce_fwd_f32": (_c_int, [_ptr, _ptr, _ptr, _c_int, _c_int, _ptr]),
    "pcgmix_soft_ce_bwd_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr]),
    "pcgmix_splice_same_label_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, ctypes.c_uint64, _c_float, _ptr,
                                              _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int,
                                              _ptr]),
    "pcgmix_splice_same_label_ohe_f32": (_c_int, [_ptr, _ptr, _ptr, _c_int, _ptr, _ptr, ctypes.c_uint64,
                                                  _c_float, _ptr, _ptr, _c_int, _ptr, _ptr, _ptr,
                                                  _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_splice_staging_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "pcgmix_ctx_create": (_c_int, [_c_int, ctypes.POINTER(_ptr)]),
    "pcgmix_ctx_destroy": (None, [_ptr]),
    "pcgmix_fetch_h2d": (_c_int, [_ptr, _ptr, ctypes.c_size_t, _ptr]),
    "pcgmix_mix_karg_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_float, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_mix_karg_variant": (_c_int, [_c_int, _c_int, _c_int, _ptr]),
    "pcgmix_ctx_gate": (ctypes.c_double, [_ptr, ctypes.c_uint64]),
    "pcgmix_ctx_set_payload": (_c_int, [_ptr, _ptr, ctypes.c_size_t, _ptr]),
    "pcgmix_ctx_flush_payload": (_c_int, [_ptr, _ptr]),
    "pcgmix_ctx_salopt_begin": (_c_int, [_ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int, _ptr]),
    "pcgmix_ctx_salopt_begin_labels": (_c_int, [_ptr, _ptr, _c_int, _ptr, _ptr, _ptr, _c_int, _c_int,
                                                _ptr]),
    "pcgmix_ctx_salopt_finish": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_uint64, _c_float,
                                          _c_int, _ptr, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_ctx_phase_times": (ctypes.c_longlong, [_ptr, _ptr]),
    "pcgmix_ctx_armed_stats": (_c_int, [_ptr, _ptr]),
    "pcgmix_augment_plain_begin": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_augment_plain_finish": (_c_int, [_ptr, _ptr, ctypes.c_uint64, _c_float, _ptr]),
    "pcgmix_ctx_armed_debug": (_c_int, [_ptr, ctypes.c_uint64, _c_int]),
    "pcgmix_ctx_labels_begin": (_c_int, [_ptr, _ptr, _c_int, _c_int, _ptr, _ptr]),
    "pcgmix_ctx_labels_wait": (_c_int, [_ptr, _ptr, _c_int, _ptr]),
    "pcgmix_augment_plain_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _c_int, _ptr, _ptr, ctypes.c_uint64,
                                          _c_float, _ptr, _c_int, _ptr, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_mix_kernel_name": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_char_p,
                                        _c_int]),
    "pcgmix_mix_variant": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_int),
                                    ctypes.POINTER(_c_int)]),
    "pcgmix_bnrp_workspace_floats": (ctypes.c_longlong, [_c_int, _c_int, _c_int, _c_int]),
    "pcgmix_bnrp_fwd_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_float, _c_float, _ptr, _ptr, _ptr, _ptr,
                                     _ptr, _ptr, _ptr, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr]),
    "pcgmix_bnrp_bwd_f32": (_c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_int,
                                     _c_int, _c_int, _c_int, _c_int, _c_int, _ptr]),
}

_lib = None
_lock = threading.Lock()


class PcgmixLibraryError(RuntimeError):
    pass


TAPE = None      # a list while a captured step records its library launches (train_model.GraphedTrainStep)


class _Recorder:
    """``load()``'s return value while ``TAPE`` is a list: every call whose last argument is a stream
    handle is appended as (name, function, arguments) and then made."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            if TAPE is not None and a and isinstance(a[-1], ctypes.c_void_p):
                TAPE.append((name, fn, a))
            return fn(*a)
        return call


def load() -> ctypes.CDLL:
    """Load (once) and type the library.  Raises PcgmixLibraryError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib if TAPE is None else _Recorder(_lib)
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PcgmixLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}`. "
                "There is no CPU fallback for the PCGmix kernels.")
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime, wrong arch, ...
            raise PcgmixLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise PcgmixLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = restype
            fn.argtypes = argtypes
        got = lib.pcgmix_abi_version()
        if got != ABI_VERSION:
            raise PcgmixLibraryError(f"ABI version mismatch: library {got}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(err: int, what: str) -> None:
    """Raise RuntimeError carrying the hipError_t name when an entry point failed."""
    if err != 0:
        msg = load().pcgmix_error_string(err)
        raise RuntimeError(f"{what} failed: hipError_t {err} "
                           f"({msg.decode() if msg else 'unknown'})")


import contextlib as _contextlib


@_contextlib.contextmanager
def capture_without_gc():
    """Around a stream capture: collect now, then keep Python's cyclic collector off until the capture
    has ended.  A collection that fires inside the capture runs finalizers of unrelated garbage on the
    capturing thread — torch objects whose release makes HIP calls a capture does not allow — and the
    process aborts inside torch (seen as a bare SIGABRT under ``Garbage-collecting`` in the second
    ``GraphedTrainStep`` of a long test session; whether it happens depends on the allocation count at
    that moment).  ``torch.cuda.graph`` collects before the capture for the same reason but leaves the
    collector running."""
    import gc
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
