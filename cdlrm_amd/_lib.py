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
    "cdlrm_debug_set": (C.c_int, [c_i32, c_i32]),
    "cdlrm_last_error": (C.c_char_p, []),
    "cdlrm_ctx_create": (C.c_int, [C.POINTER(Geometry), C.POINTER(vp)]),
    "cdlrm_ctx_destroy": (C.c_int, [vp]),
    "cdlrm_ctx_bind_cache": (C.c_int, [vp, vp, vp]),
    "cdlrm_ctx_bind_host_tables": (C.c_int, [vp, C.POINTER(vp)]),
    "cdlrm_host_register": (C.c_int, [vp, c_u64, C.POINTER(vp)]),
    "cdlrm_host_unregister": (C.c_int, [vp]),
    "cdlrm_ctx_check_sync": (C.c_int, [vp, vp]),
    "cdlrm_embbag_probe": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, vp, c_i32, vp]),
    "cdlrm_window_resolve": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, c_i64, vp, vp, vp]),
    "cdlrm_embbag_take": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, c_i64, vp, c_i32, vp]),
    "cdlrm_victim_writeback_work_bytes": (c_u64, [c_i32, c_i64]),
    "cdlrm_victim_writeback": (C.c_int, [vp, vp, c_i64, c_i64, vp, vp, c_i64, c_i32, vp, vp]),
    "cdlrm_embbag_fwd": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, vp]),
    "cdlrm_embbag_bwd_work_bytes": (c_u64, [c_i32, c_i64, c_i32]),
    "cdlrm_embbag_bwd_sgd": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp]),
    "cdlrm_embbag_bwd_prepare": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "cdlrm_embbag_bwd_apply": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp]),
    "cdlrm_embbag_bwd_apply_rest": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp]),
    "cdlrm_embbag_bwd_once_flags": (C.c_int, [vp, vp, c_i64, vp]),
    "cdlrm_embbag_bwd_sorted_bytes": (C.c_uint64, [c_i32, c_i32, c_i64]),
    "cdlrm_embbag_bwd_prepare_window": (C.c_int, [vp, vp, c_i64, c_i64, c_i32, c_i64, c_i32, c_i32, vp, vp]),
    "cdlrm_embbag_bwd_sorted_views": (C.c_int, [vp, vp, c_i32, c_i64, c_i32, vp, vp, vp]),
    "cdlrm_embbag_bwd_apply_sorted": (C.c_int, [vp, c_i64, vp, c_i64, c_i64, c_f32, vp, vp, vp, c_i64, c_i32, c_i32, vp, vp]),
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
    "cdlrm_agg_mark_tier": (C.c_int, [vp, vp, c_i64, c_i64, c_i32, vp, vp]),
    "cdlrm_agg_split": (C.c_int, [vp, vp, c_i64, vp, c_i32, vp, vp, vp]),
    "cdlrm_interact_fwd": (C.c_int, [vp, c_i64, c_i32, c_i32, c_i32, vp, c_i64, vp]),
    "cdlrm_interact_bwd": (C.c_int, [vp, vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, vp, vp]),
    "cdlrm_gather_interact_supported": (C.c_int, [vp]),
    "cdlrm_gather_interact_fwd": (C.c_int, [vp, vp, c_i64, vp, c_i64, c_i64, c_i32, vp, c_i64, vp]),
    "cdlrm_gather_interact_bwd": (C.c_int, [vp, vp, c_i64, vp, c_i64, vp, c_i64, c_i64, c_i32, c_i32, vp, vp]),
    "cdlrm_gather_interact_bwd_sgd": (C.c_int, [vp, vp, c_i64, vp, c_i64, vp, c_i64, c_i64, c_i32, c_i32, vp, vp, c_i64, c_f32, vp]),
    "cdlrm_linear_fwd": (C.c_int, [vp, c_i64, vp, vp, vp, c_i64, c_i64, c_i32, c_i32, c_i32, vp]),
    "cdlrm_linear_bwd_work_bytes": (c_u64, [c_i64, c_i32, c_i32]),
    "cdlrm_linear_bwd": (C.c_int, [vp, c_i64, vp, vp, c_i64, vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i32, c_i32,
                                   c_i32, c_i32, vp, vp]),
    "cdlrm_mlp_wgrad_work_bytes": (c_u64, [c_i32, c_i64, vp, vp]),
    "cdlrm_mlp_wgrad": (C.c_int, [c_i32, vp, vp, vp, vp, vp, vp, c_i64, vp, vp, vp, vp]),
    "cdlrm_mlp_wgrad_sgd": (C.c_int, [c_i32, vp, vp, vp, vp, vp, vp, vp, vp, c_f32, c_i64, vp, vp, vp, vp]),
    "cdlrm_bce_fwd_bwd": (C.c_int, [vp, vp, c_i64, vp, vp, c_i32, vp]),
    "cdlrm_loss_fwd_bwd": (C.c_int, [vp, vp, c_i64, c_i32, C.c_float, C.c_float, C.c_float, vp, vp, vp, c_i32, vp]),
    "cdlrm_head_scratch_floats": (c_i64, []),
    "cdlrm_head_fwd_bwd": (C.c_int, [vp, c_i64, vp, vp, vp, c_i64, c_i32, c_i32, C.c_float, C.c_float, C.c_float, c_i32,
                                     vp, vp, vp, vp, c_i64, vp, vp, c_i32, vp]),
    "cdlrm_head_finish": (C.c_int, [vp, c_i64, vp, vp, vp]),
    "cdlrm_sgd_step2": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, c_i64, C.c_float, vp]),
    "cdlrm_act_bwd": (C.c_int, [vp, c_i64, vp, c_i64, c_i64, c_i32, c_i32, vp]),
    "cdlrm_sgd_step": (C.c_int, [vp, vp, c_i64, c_f32, vp]),
    "cdlrm_scale_div": (C.c_int, [vp, c_i64, c_f32, vp]),
    "cdlrm_scatter_rows": (C.c_int, [vp, vp, vp, c_i64, c_i32, C.c_int, vp]),
    "cdlrm_blend_rows": (C.c_int, [vp, vp, vp, c_i64, c_i32, vp, vp]),
    "cdlrm_mark_rows": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "cdlrm_synth_indices": (C.c_int, [vp, c_i64, c_i64, c_i64, C.c_double, c_u64, vp]),
    "cdlrm_tape_create": (vp, [c_i32]),
    "cdlrm_tape_destroy": (None, [vp]),
    "cdlrm_tape_add": (C.c_int, [vp, vp, c_i32, vp, vp, c_i32, vp]),
    "cdlrm_tape_cells": (vp, [vp]),
    "cdlrm_tape_length": (c_i64, [vp]),
    "cdlrm_tape_stream_arg": (c_i32, [vp]),
    "cdlrm_tape_op_info": (C.c_char_p, [vp, c_i64, vp]),
    "cdlrm_tape_probe": (C.c_int, [c_f32, c_i64, c_f32, c_i32, vp, c_i64, c_f32, c_i32, c_i64, c_i64, vp, c_i32, c_f32, c_i64]),
    "cdlrm_tape_probe_log": (C.c_int, [c_i64, c_i64]),
    "cdlrm_tape_probe_log_take": (c_i64, [vp, c_i64]),
    "cdlrm_tape_replay": (C.c_int, [vp]),
    "cdlrm_tape_set_lanes": (C.c_int, [vp, vp, vp, c_i64]),
    "cdlrm_tape_selftest": (C.c_int, []),
    "cdlrm_event_record": (C.c_int, [vp, vp]),
    "cdlrm_stream_wait_event": (C.c_int, [vp, vp]),
    "cdlrm_event_attach_next": (C.c_int, [vp, vp]),
    "cdlrm_stream_create": (vp, [c_i32]),
    "cdlrm_stream_destroy": (C.c_int, [vp]),
    "cdlrm_event_create": (vp, [c_i32]),
    "cdlrm_event_destroy": (C.c_int, [vp]),
    "cdlrm_event_elapsed_us": (C.c_int, [vp, vp, vp]),
    "cdlrm_ctx_time_next_gather": (C.c_int, [vp, vp, vp]),
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
        # size / geometry queries (cdlrm_*_work_bytes, ...: anything that does not return a status code) issue nothing: a
        # recorded step that happens to allocate its scratch does not put them on its tape
        is_call = fn.restype is C.c_int

        def call(*args):
            tape = getattr(_rec, "tape", None)
            if tape is not None and is_call:
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


def require_gpu(what: str = "cdlrm_amd") -> None:
    """The cached training path has no CPU fallback: say so, instead of a torch traceback from the first device call."""
    import torch
    if not torch.cuda.is_available():
        raise CdlrmLibraryError(
            "%s needs a HIP device (MI355X / gfx950) and found none: cdlrm_amd has no CPU fallback for the cached training "
            "path -- every compute entry point is a hand-written HIP kernel behind csrc/libcdlrm_hip.so" % what)


HW_QUEUES_TUNED = 4


def hw_queues() -> int:
    """GPU_MAX_HW_QUEUES as the HIP runtime of this process read it (its default is 4)."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", str(HW_QUEUES_TUNED)))
    except ValueError:
        return -1


_warned_queues = False


def warn_hw_queues() -> None:
    """The training step's five streams are scheduled for the runtime's default of 4 hardware queues per process (with 6 or
    8 the c3 step measured 1.39-1.42 ms instead of 0.79).  bench.py and the CLI pin the variable before
    HIP starts; a caller that imports cdlrm_amd with another value gets a warning, once."""
    global _warned_queues
    if not _warned_queues and hw_queues() != HW_QUEUES_TUNED:
        _warned_queues = True
        import warnings
        warnings.warn("GPU_MAX_HW_QUEUES=%s: cdlrm_amd's step schedule is tuned for %d hardware queues per process (set "
                      "GPU_MAX_HW_QUEUES=%d before the first HIP call)"
                      % (os.environ.get("GPU_MAX_HW_QUEUES"), HW_QUEUES_TUNED, HW_QUEUES_TUNED), RuntimeWarning, stacklevel=3)


class TapeUnsupported(Exception):
    """A recorded call the native tape cannot hold (the engine then replays that step's tape from Python)."""


_native_ok: Optional[bool] = None


def native_tape_ok() -> bool:
    """csrc/tape.hip's typed call passes its self-test."""
    global _native_ok
    if _native_ok is None:
        _native_ok = raw().cdlrm_tape_selftest() == 0
    return _native_ok


class NativeTape:
    """A recorded step as a C-side tape: `prog` is the engine's list of (function, arguments, is_library_call) with the
    per-step pointers already replaced by shared ctypes cells; replay() re-issues all of it with ONE library call.
    Library calls are stored by address with their arguments split by register class (ctypes argtypes tell which);
    torch stream / event calls become cdlrm_stream_wait_event / cdlrm_event_record on the raw HIP handles."""

    def __init__(self, prog, cells: dict, main_stream: Optional[int] = None, max_lanes: int = 2):
        """main_stream: raw handle of the training queue's stream.  Given (and not the null stream), the tape is replayed in
        up to `max_lanes` (<= 4) lanes: ops on that stream by the replaying thread, every other stream's by helper threads of
        the library (one lane per stream in order of appearance, the last lane takes the rest), ordered against each other on
        the host the way their events order them on the GPU (cdlrm_tape_set_lanes)."""
        import torch
        L = raw()
        order = list(cells.values())
        index = {id(c): i for i, c in enumerate(order)}
        self._cell_ids = set(index)
        self._cells_py = order
        self._keep = []                          # torch events created for wait_stream: must outlive the tape
        h = L.cdlrm_tape_create(len(order))
        if not h:
            raise CdlrmError(-22, L.cdlrm_last_error().decode("utf-8", "replace"))
        self._h = h
        ops = []
        try:
            for fn, args, _ in prog:
                for target, cargs in self._translate(fn, args, torch):
                    self._add(L, target, cargs, index)
                    ops.append((target, cargs))
            self.lanes = 1
            if main_stream and max_lanes > 1:
                self.lanes = self._set_lanes(L, ops, int(main_stream), min(int(max_lanes), 4))
        except Exception:
            L.cdlrm_tape_destroy(h)
            self._h = None
            raise
        addr = L.cdlrm_tape_cells(h) if order else None     # (a tape without cells has no cell array)
        self._cells = (C.c_int64 * len(order)).from_address(addr) if addr else None
        self._replay = L.cdlrm_tape_replay

    def op_times(self):
        """Development (tools/host_time.py): per recorded op (name, lane, replays timed, us per replay inside the call, us per
        replay waiting for another lane, us of the longest call) -- the clocks run while `cdlrm_debug_set(3, 1)` is on."""
        L, out, res = raw(), (C.c_int64 * 5)(), []
        for k in range(int(L.cdlrm_tape_length(self._h))):
            name = L.cdlrm_tape_op_info(self._h, k, out)
            n = max(1, int(out[1]))
            res.append((name.decode() if name else "?", int(out[0]), int(out[1]), out[2] / n / 1e3, out[3] / n / 1e3, out[4] / 1e3))
        return res

    def _translate(self, fn, args, torch):
        if getattr(fn, "argtypes", None) is not None:
            return [(fn, args)]
        L = raw()
        obj, name = getattr(fn, "__self__", None), getattr(fn, "__name__", "")
        if isinstance(obj, torch.cuda.Stream) and name == "wait_event":
            return [(L.cdlrm_stream_wait_event, (obj.cuda_stream, self._event(args[0])))]
        if isinstance(obj, torch.cuda.Event) and name == "record":
            return [(L.cdlrm_event_record, (self._event(obj), args[0].cuda_stream))]
        if isinstance(obj, torch.cuda.Stream) and name == "wait_stream":
            ev = torch.cuda.Event()
            ev.record(args[0])                    # creates the HIP event (torch makes them lazily)
            self._keep.append(ev)
            return [(L.cdlrm_event_record, (ev.cuda_event, args[0].cuda_stream)),
                    (L.cdlrm_stream_wait_event, (obj.cuda_stream, ev.cuda_event))]
        raise TapeUnsupported(repr(fn))

    def _set_lanes(self, L, ops, main_stream: int, max_lanes: int) -> int:
        """Split the recorded ops by stream and order the lanes on every event both of them touch."""
        names = [getattr(fn, "__name__", "") for fn, _ in ops]

        def val(a, what):
            # Lanes and cross-lane dependencies are fixed at build time.  A STREAM that is patched per replay (a cell) would be
            # laned by whatever handle it held when the step was recorded: refused.  An EVENT cell (the resolver's ring-slot
            # event) is keyed by the cell itself and accepted only if every op that names it ends up in ONE lane, where program
            # order orders its touches (checked below).
            if isinstance(a, C.c_void_p) and id(a) in self._cell_ids:
                if what == "stream":
                    raise TapeUnsupported("a stream argument that is a per-step cell cannot be assigned to a lane")
                return ("cell", id(a))
            return int((a.value or 0) if isinstance(a, C._SimpleCData) else (0 if a is None else a))

        streams, events = [], []
        for (fn, args), name in zip(ops, names):
            # where the stream sits in the parameter list is the library's knowledge (csrc/tape.hip: tape_registry), not a guess
            pos = int(L.cdlrm_tape_stream_arg(C.cast(fn, C.c_void_p)))
            if pos == -2:
                raise TapeUnsupported("%s is not a registered tape entry point" % name)
            streams.append(val(args[pos], "stream") if pos >= 0 else None)     # (None: the call issues nothing itself)
            if name == "cdlrm_stream_wait_event":
                events.append(val(args[1], "event"))
            elif name in ("cdlrm_event_record", "cdlrm_event_attach_next"):
                events.append(val(args[0], "event"))
            else:
                events.append(None)
        lane_of = {main_stream: 0}
        for st in streams:                      # lanes in order of appearance; the last lane takes every further stream
            if st is not None and st not in lane_of:
                lane_of[st] = min(len(lane_of), max_lanes - 1)
        lane = [0] * len(ops)
        nxt = 0
        for k in reversed(range(len(ops))):
            if streams[k] is not None:
                nxt = lane_of[streams[k]]
            lane[k] = nxt
        dep, last = [-1] * len(ops), {}
        for k in range(len(ops)):
            e = events[k]
            if e is None:
                continue
            if isinstance(e, tuple) and e in last and lane[last[e]] != lane[k]:
                raise TapeUnsupported("an event that is a per-step cell is touched from two lanes")
            if e in last and lane[last[e]] != lane[k]:
                dep[k] = last[e]
            last[e] = k
            if names[k] == "cdlrm_event_attach_next":
                # the event is recorded by the NEXT call of this lane (the launch that carries it), not by the attach itself:
                # a wait of the other lane has to be held back until THAT call has been issued
                if k + 1 >= len(ops) or lane[k + 1] != lane[k] or events[k + 1] is not None:
                    raise TapeUnsupported("cdlrm_event_attach_next is not followed by the launch that carries its event")
                last[e] = k + 1
        if max(lane) == 0:
            return 1
        n = len(ops)
        check(L.cdlrm_tape_set_lanes(self._h, (C.c_int32 * n)(*lane), (C.c_int32 * n)(*dep), n))
        return max(lane) + 1

    @staticmethod
    def _event(ev) -> int:
        h = ev.cuda_event
        if not h:
            raise TapeUnsupported("an event that was never recorded has no HIP handle yet")
        return int(h)

    def _add(self, L, fn, args, index):
        types = fn.argtypes
        if len(types) != len(args):
            raise TapeUnsupported("argument count of %r" % fn)
        ia, ci, fa = [], [], []
        for a, ty in zip(args, types):
            if ty is C.c_float:
                fa.append(float(a))
                continue
            if ty is C.c_double:
                raise TapeUnsupported("double argument")
            cell = -1
            if a is None:
                v = 0
            elif isinstance(a, C.c_void_p) and id(a) in index:
                cell, v = index[id(a)], 0
            elif isinstance(a, C.Array):
                v = C.addressof(a)
            elif isinstance(a, C._SimpleCData):
                v = a.value or 0
            elif isinstance(a, (bytes, str)):
                raise TapeUnsupported("string argument")
            else:
                v = int(a)
            ia.append(v)
            ci.append(cell)
        n = len(ia)
        IA = (C.c_int64 * max(1, n))(*ia)
        CI = (C.c_int32 * max(1, n))(*ci)
        FA = (C.c_float * max(1, len(fa)))(*fa)
        rc = L.cdlrm_tape_add(self._h, C.cast(fn, C.c_void_p), n, IA, CI, len(fa), FA)
        if rc:
            raise TapeUnsupported("%s: %s" % (getattr(fn, "__name__", fn), L.cdlrm_last_error().decode("utf-8", "replace")))

    def replay(self) -> int:
        """Patch the cells from the shared ctypes cells the Python tape uses, re-issue the step."""
        cs = self._cells
        for i, c in enumerate(self._cells_py):
            cs[i] = c.value or 0
        return self._replay(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cdlrm_tape_destroy(h)


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
