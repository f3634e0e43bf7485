"""ctypes binding of libcdlrm_hip.so (include/cdlrm_hip.h).

The library is the product path: there is no CPU fallback.  `lib()` raises `CdlrmLibraryError` when the
shared object has not been built (`python -c "import __graft_entry__ as g; g.build()"` or
`make -C cdlrm_amd/csrc`), and every compute entry point of the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcdlrm_hip.so")

c_i32, c_i64, c_u64, c_f32 = C.c_int32, C.c_int64, C.c_uint64, C.c_float
vp = C.c_void_p


class CdlrmLibraryError(RuntimeError):
    pass


class CdlrmError(RuntimeError):
    def __init__(self, rc: int, msg: str):
        super().__init__("libcdlrm_hip: rc=%d: %s" % (rc, msg))
        self.rc = rc


class Geometry(C.Structure):
    _fields_ = [("num_tables", c_i32), ("dim", c_i32), ("num_ways", c_i32), ("aux_rows", c_i32),
                ("table_rows", C.POINTER(c_i64)), ("cache_sets", C.POINTER(c_i64)),
                ("device", c_i32), ("aux_phases", c_i32)]


class Plan(C.Structure):
    _fields_ = [("bitmap", vp), ("uniq", vp), ("uniq_off", vp), ("cap_uniq", c_i64),
                ("prot", vp), ("hit", vp), ("kept", vp), ("kept_off", vp), ("way", vp), ("flags", vp),
                ("winner", vp), ("win_claim", vp), ("win_idx", vp), ("win_row", vp), ("win_tag", vp),
                ("win_off", vp), ("cap_win", c_i64), ("stage", vp), ("ev_tag", vp)]


class Victims(C.Structure):
    _fields_ = [("pos", vp), ("idx", vp), ("off", vp), ("rows", vp), ("cap", c_i64)]


# name -> (restype, argtypes); every symbol include/cdlrm_hip.h declares
PROTOTYPES = {
    "cdlrm_abi_version": (C.c_int, []),
    "cdlrm_last_error": (C.c_char_p, []),
    "cdlrm_ctx_create": (C.c_int, [C.POINTER(Geometry), C.POINTER(vp)]),
    "cdlrm_ctx_destroy": (C.c_int, [vp]),
    "cdlrm_ctx_bind_cache": (C.c_int, [vp, vp, vp]),
    "cdlrm_ctx_bind_host_tables": (C.c_int, [vp, C.POINTER(vp)]),
    "cdlrm_host_register": (C.c_int, [vp, c_u64, C.POINTER(vp)]),
    "cdlrm_host_unregister": (C.c_int, [vp]),
    "cdlrm_ctx_check_sync": (C.c_int, [vp, vp]),
    "cdlrm_embbag_probe": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, vp, c_i32, vp]),
    "cdlrm_window_resolve": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, vp, vp, vp]),
    "cdlrm_embbag_take": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, c_i64, vp, c_i32, vp]),
    "cdlrm_embbag_fwd": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, vp]),
    "cdlrm_embbag_bwd_work_bytes": (c_u64, [c_i32, c_i64, c_i32]),
    "cdlrm_embbag_bwd_sgd": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp]),
    "cdlrm_embbag_bwd_prepare": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "cdlrm_embbag_bwd_apply": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp]),
    "cdlrm_qr_embbag_fwd": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, c_i64, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp]),
    "cdlrm_qr_embbag_bwd": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, vp, c_i64, c_i32, c_i32, c_i32, vp, vp, vp]),
    "cdlrm_bag_fwd": (C.c_int, [vp, vp, c_i64, c_i64, vp, c_i64, c_i32, vp, vp, vp]),
    "cdlrm_bag_bwd": (C.c_int, [vp, vp, c_i64, c_i64, vp, c_i64, c_i32, vp, vp]),
    "cdlrm_window_unique": (C.c_int, [vp, C.POINTER(Plan), vp, c_i64, c_i64, vp]),
    "cdlrm_window_unique_add": (C.c_int, [vp, C.POINTER(Plan), vp, c_i64, c_i64, vp]),
    "cdlrm_window_unique_finish": (C.c_int, [vp, C.POINTER(Plan), vp]),
    "cdlrm_plan_probe": (C.c_int, [vp, C.POINTER(Plan), vp]),
    "cdlrm_plan_offsets_sync": (C.c_int, [vp, C.POINTER(Plan), vp, vp, vp, vp]),
    "cdlrm_plan_assign": (C.c_int, [vp, C.POINTER(Plan), vp, c_u64, vp]),
    "cdlrm_plan_fetch": (C.c_int, [vp, C.POINTER(Plan), C.POINTER(vp), C.c_int, vp]),
    "cdlrm_plan_commit": (C.c_int, [vp, C.POINTER(Plan), vp]),
    "cdlrm_plan_writeback": (C.c_int, [vp, C.POINTER(Plan), C.POINTER(vp), C.c_int, vp]),
    "cdlrm_plan_victims": (C.c_int, [vp, C.POINTER(Plan), C.POINTER(Victims), vp]),
    "cdlrm_plan_victims_list": (C.c_int, [vp, C.POINTER(Plan), C.POINTER(Victims), vp]),
    "cdlrm_host_gather_rows": (C.c_int, [C.POINTER(vp), vp, vp, c_i32, c_i32, vp, c_i32]),
    "cdlrm_ctx_bind_victims": (C.c_int, [vp, C.POINTER(Victims)]),
    "cdlrm_gather_rows": (C.c_int, [vp, vp, c_i64, c_i32, vp, vp]),
    "cdlrm_agg_compact": (C.c_int, [vp, vp, c_i64, vp, c_i64, vp, vp]),
    "cdlrm_agg_gather": (C.c_int, [vp, vp, vp, c_f32, vp, c_i64, c_i64, vp]),
    "cdlrm_agg_scatter": (C.c_int, [vp, vp, vp, vp, c_i64, c_i64, vp]),
    "cdlrm_interact_fwd": (C.c_int, [vp, c_i64, c_i32, c_i32, c_i32, vp, c_i64, vp]),
    "cdlrm_interact_bwd": (C.c_int, [vp, vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, vp, vp]),
    "cdlrm_linear_fwd": (C.c_int, [vp, c_i64, vp, vp, vp, c_i64, c_i64, c_i32, c_i32, c_i32, vp]),
    "cdlrm_linear_bwd_work_bytes": (c_u64, [c_i64, c_i32, c_i32]),
    "cdlrm_linear_bwd": (C.c_int, [vp, c_i64, vp, vp, c_i64, vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i32, c_i32,
                                   c_i32, c_i32, vp, vp]),
    "cdlrm_mlp_wgrad_work_bytes": (c_u64, [c_i32, c_i64, vp, vp]),
    "cdlrm_mlp_wgrad": (C.c_int, [c_i32, vp, vp, vp, vp, vp, vp, c_i64, vp, vp, vp, vp]),
    "cdlrm_bce_fwd_bwd": (C.c_int, [vp, vp, c_i64, vp, vp, c_i32, vp]),
    "cdlrm_loss_fwd_bwd": (C.c_int, [vp, vp, c_i64, c_i32, C.c_float, C.c_float, C.c_float, vp, vp, vp, c_i32, vp]),
    "cdlrm_head_scratch_floats": (c_i64, []),
    "cdlrm_head_fwd_bwd": (C.c_int, [vp, c_i64, vp, vp, vp, c_i64, c_i32, c_i32, C.c_float, C.c_float, C.c_float, c_i32,
                                     vp, vp, vp, vp, c_i64, vp, vp, c_i32, vp]),
    "cdlrm_head_finish": (C.c_int, [vp, c_i64, vp, vp]),
    "cdlrm_sgd_step2": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, c_i64, C.c_float, vp]),
    "cdlrm_act_bwd": (C.c_int, [vp, c_i64, vp, c_i64, c_i64, c_i32, c_i32, vp]),
    "cdlrm_sgd_step": (C.c_int, [vp, vp, c_i64, c_f32, vp]),
    "cdlrm_scale_div": (C.c_int, [vp, c_i64, c_f32, vp]),
    "cdlrm_scatter_rows": (C.c_int, [vp, vp, vp, c_i64, c_i32, C.c_int, vp]),
    "cdlrm_blend_rows": (C.c_int, [vp, vp, vp, c_i64, c_i32, vp, vp]),
    "cdlrm_mark_rows": (C.c_int, [vp, vp, c_i64, vp, vp]),
}

_lib: Optional[C.CDLL] = None
_proxy = None
# Recording state is PER THREAD: while the trainer thread records a step, the window plan's background thread makes
# library calls of its own (host gather, offsets), and the garbage collector may run a context's destructor -- none of
# those may land on the step's tape (a recorded cdlrm_ctx_destroy would be replayed on a dangling handle every step).
import threading as _threading

_rec = _threading.local()           # _rec.tape: a list while this thread records, else absent / None


class _Recording:
    """Thin view of the library whose calls can be recorded: the engine records the launch sequence of a training
    step once and replays it (same functions, same argument objects) without the Python around each launch."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)          # AttributeError for an unknown symbol

        def call(*args):
            tape = getattr(_rec, "tape", None)
            if tape is not None:
                tape.append((fn, args))
            return fn(*args)

        call.__name__ = name
        self.__dict__[name] = call
        return call


def start_recording(tape: list) -> None:
    _rec.tape = tape


def stop_recording() -> None:
    _rec.tape = None


def record(fn, *args):
    """Run a non-library call (event record / stream wait) and, while recording, put it on the tape as well."""
    tape = getattr(_rec, "tape", None)
    if tape is not None:
        tape.append((fn, args))
    return fn(*args)


def raw():
    """The library WITHOUT the recording view: destructors and other calls that must never land on a step's tape."""
    lib()
    return _lib


def lib():
    """The loaded library; raises loudly when it is missing (no fallback path exists)."""
    global _lib, _proxy
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CdlrmLibraryError(
                "%s not found: build it with `make -C cdlrm_amd/csrc` (or __graft_entry__.build()). "
                "cdlrm_amd has no CPU fallback for the cached training path." % LIB_PATH)
        # torch first: it ships its own copy of the HIP runtime, and the process must end up with ONE runtime.  With
        # libcdlrm_hip.so (linked against /opt/rocm's libamdhip64) loaded before torch, the library's first
        # hipSetDevice() reported "no ROCm-capable device" (seen with build() followed by smoke() in one process).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.cdlrm_abi_version() != 1:
            raise CdlrmLibraryError("ABI version mismatch")
        _lib = l
        _proxy = _Recording(l)
    return _proxy


def check(rc: int) -> None:
    if rc != 0:
        raise CdlrmError(rc, lib().cdlrm_last_error().decode("utf-8", "replace"))


def ptr(t) -> Optional[int]:
    """data_ptr of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream
