"""ctypes binding of ``libmkb_hip.so`` (C ABI: ``include/mkb_hip.h``).

The library is the ONLY compute backend of ``mkb_amd``: if it cannot be loaded, or a tensor is not on a
ROCm device, the call raises -- there is no CPU fallback (the CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here).
"""
import ctypes
import pathlib
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint16, c_uint32, c_void_p

import torch

import os

# MKB_HIP_LIB selects an experimental build of the same ABI (tools/kbench.py); default = the in-tree product library
_LIB_PATH = pathlib.Path(os.environ.get("MKB_HIP_LIB") or (pathlib.Path(__file__).resolve().parent / "libmkb_hip.so"))
ABI_VERSION = 6  # == MKB_ABI_VERSION of include/mkb_hip.h (bumped whenever a symbol or a signature changes)

MODEL_IDS = {"TransE": 0, "RotatE": 1, "ComplEx": 2, "DistMult": 3, "pRotatE": 4}
MODE_DEFAULT, MODE_HEAD, MODE_TAIL = 0, 1, 2
ERR_KEY, ERR_EMPTY = -3, -4


def mode_id(mode):
    """models/base.py:153-164: only the two batch strings select a candidate side."""
    return MODE_HEAD if mode == "head-batch" else MODE_TAIL if mode == "tail-batch" else MODE_DEFAULT


class Tables(Structure):
    _fields_ = [("model", c_int32), ("hidden_dim", c_int32), ("n_entity", c_int64), ("n_relation", c_int64),
                ("entity_dim", c_int64), ("relation_dim", c_int64), ("ent", c_void_p), ("rel", c_void_p),
                ("modulus", c_void_p), ("gamma", c_float), ("phase_div", c_float)]


class Grads(Structure):  # mkb_grads_t; rows_clear (default 0): see include/mkb_hip.h
    _fields_ = [("g_ent", c_void_p), ("g_rel", c_void_p), ("g_modulus", c_void_p), ("rows_clear", c_int32)]


class AdamDense(Structure):  # mkb_adam_dense_t: a small dense tensor stepped inside mkb_adam_rows_step's launch
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("n", c_int64), ("step", c_int64)]


class RowSeg(Structure):  # mkb_row_seg_t: rows of a table shard listed by shard index (world == 0) or global entity id
    _fields_ = [("ids", c_void_p), ("n", c_int64), ("rows", c_void_p), ("world", c_int32), ("rank", c_int32),
                ("local_ids", c_void_p)]


class HipLibraryError(RuntimeError):
    pass


_lib = None

_SIGNATURES = {
    "mkb_abi_version": (c_int, []),
    "mkb_last_error": (c_char_p, []),
    "mkb_score_fwd": (c_int, [POINTER(Tables), c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "mkb_score_bwd_workspace_bytes": (c_int64, [POINTER(Tables), c_int64, c_int64, c_int]),
    "mkb_score_bwd": (c_int, [POINTER(Tables), POINTER(Grads), c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                              c_void_p, c_void_p]),
    "mkb_adversarial": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_sampler_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_uint32, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mkb_sampler_generate": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "mkb_sampler_status": (c_int, [c_void_p, c_void_p]),
    "mkb_sampler_set_rng": (c_int, [c_void_p, c_int, ctypes.c_uint64, ctypes.c_uint64]),
    "mkb_sampler_get_rng": (c_int, [c_void_p, POINTER(c_int), POINTER(ctypes.c_uint64), POINTER(ctypes.c_uint64)]),
    "mkb_sampler_get_state": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_void_p]),
    "mkb_sampler_set_state": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "mkb_sampler_destroy": (None, [c_void_p]),
    "mkb_pool_supported": (c_int, [POINTER(Tables), c_int64, c_int64]),
    "mkb_pool_step_workspace_bytes": (c_int64, [POINTER(Tables), c_int64, c_int64]),
    "mkb_pool_step": (c_int, [POINTER(Tables), POINTER(Grads), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                              c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_pool_step_fwd": (c_int, [POINTER(Tables), c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "mkb_pool_step_bwd": (c_int, [POINTER(Tables), POINTER(Grads), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                  c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_pool_score_fwd": (c_int, [POINTER(Tables), c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    "mkb_pool_score_bwd": (c_int, [POINTER(Tables), POINTER(Grads), c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "mkb_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_float,
                              c_float, c_int, c_void_p]),
    "mkb_adam_step_multi": (c_int, [c_void_p, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "mkb_profile_enable": (c_int, [c_int, c_int]),
    "mkb_profile_read": (c_int, [c_int, POINTER(c_int64), POINTER(ctypes.c_double)]),
    "mkb_adam_rows_catchup": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                      c_int64, c_float, c_float, c_float, c_void_p, c_void_p]),
    "mkb_adam_rows_catchup_generate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                               c_float, c_float, c_float, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_adam_rows_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                   c_int64, c_int64, c_float, c_float, c_float, c_float, POINTER(AdamDense), c_void_p]),
    "mkb_adam_rows_advance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
                                      c_int64, c_int64, c_float, c_float, c_float, c_float, POINTER(AdamDense), c_void_p,
                                      c_void_p]),
    "mkb_adam_rows_advance_generate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                               c_int64, c_float, c_float, c_float, c_float, POINTER(AdamDense), c_void_p,
                                               c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p]),
    "mkb_rows_route": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "mkb_rows_gather": (c_int, [c_void_p, c_int64, c_int64, POINTER(RowSeg), c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                c_int64, c_void_p, c_void_p, c_void_p]),
    "mkb_rows_scatter_add": (c_int, [c_void_p, c_int64, c_int64, POINTER(RowSeg), c_int, c_void_p, c_void_p, c_int64,
                                     c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mkb_rows_blocks_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "mkb_rows_blocks_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_void_p]),
    "mkb_rows_comm_available": (c_int, []),
    "mkb_rows_comm_unique_id": (c_int, [c_void_p]),
    "mkb_rows_comm_create": (c_int, [c_void_p, c_int, c_int, c_int64, POINTER(c_void_p)]),
    "mkb_rows_comm_destroy": (None, [c_void_p]),
    "mkb_rows_loop_hub_create": (c_int, [c_int, POINTER(c_void_p)]),
    "mkb_rows_loop_hub_destroy": (None, [c_void_p]),
    "mkb_rows_comm_create_loopback": (c_int, [c_void_p, c_int, c_int64, POINTER(c_void_p)]),
    "mkb_rows_comm_plan": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_void_p, c_void_p, c_void_p]),
    "mkb_rows_comm_take": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mkb_rows_comm_exchange": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mkb_rows_comm_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "mkb_adam_rows_advance_sharded": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                              c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_int64, c_float, c_float,
                                              c_float, c_float, POINTER(AdamDense), c_void_p, c_void_p]),
    "mkb_adam_rows_advance_sharded_generate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                                       c_int, c_int, c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float,
                                                       POINTER(AdamDense), c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_check_ids": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "mkb_kl_divergence": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mkb_debug_sclk_mhz": (c_int, [c_void_p, c_void_p]),
    "mkb_rank_workspace_bytes": (c_int64, [POINTER(Tables), c_int64]),
    "mkb_rank": (c_int, [POINTER(Tables), c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                         c_void_p]),
    "mkb_rank_scores": (c_int, [POINTER(Tables), c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """Load (once) and return the ctypes handle; raises HipLibraryError if the .so is missing."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise HipLibraryError(
                f"{_LIB_PATH} not found: build it with `python -m mkb_amd.csrc.build` "
                "(mkb_amd has no CPU fallback; the HIP library is the only compute path)")
        handle = ctypes.CDLL(str(_LIB_PATH))
        handle.mkb_abi_version.restype = c_int
        if handle.mkb_abi_version() != ABI_VERSION:  # first: a stale build must say so, not fail on a missing symbol later
            raise HipLibraryError(f"ABI version mismatch: library {handle.mkb_abi_version()}, binding {ABI_VERSION} "
                                  "(rebuild with `python -m mkb_amd.csrc.build`)")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


PROF_KINDS = {"pool_bwd_q": 0, "pool_fwd": 1, "adam": 2, "sampler": 3, "loss": 4, "general_fwd": 5, "general_bwd": 6,
              "pool_bwd_x": 7}


def profile_enable(kind, on=True):
    check(lib().mkb_profile_enable(PROF_KINDS[kind], int(on)), "mkb_profile_enable")


def profile_read(kind):
    """-> (launches, total_ms) of the bracketed launches since the last read (synchronises)."""
    n, ms = c_int64(), ctypes.c_double()
    check(lib().mkb_profile_read(PROF_KINDS[kind], ctypes.byref(n), ctypes.byref(ms)), "mkb_profile_read")
    return n.value, ms.value


def check(rc, what):
    if rc != 0:
        msg = lib().mkb_last_error().decode("utf-8", "replace")
        raise HipLibraryError(f"{what} failed (status {rc}): {msg}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "mkb_amd computes on ROCm devices only (no CPU fallback): move the model and its inputs to 'cuda'. "
                f"Got a tensor on {t.device}.")


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def aligned_bytes(n, device):
    """A 256-byte aligned uint8 view of ``n`` bytes of fresh device memory (torch's caching allocator: stream-ordered)."""
    raw = torch.empty(n + 256, dtype=torch.uint8, device=device)
    off = (-raw.data_ptr()) % 256
    return raw[off: off + n]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_exchange = getattr(torch._C, "_cuda_exchangeDevice", None)


def stream_ptr(device=None):
    """torch's current stream on ``device`` (default: the current device) as the ``hipStream_t`` the C ABI takes.  Through
    torch's raw-stream query when it is there: ``torch.cuda.current_stream()`` builds a Stream object per call (~4 us -- a
    step makes four to six library calls, and small configurations are bound by exactly this host time)."""
    if _raw_stream is not None:
        idx = device.index if device is not None and getattr(device, "index", None) is not None else torch.cuda.current_device()
        return c_void_p(_raw_stream(idx))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class on_device:
    """``with on_device(tensor.device):`` -- ``torch.cuda.device`` without its per-call argument parsing (the library calls
    sit inside one each: launches go to the device of their operands)."""

    __slots__ = ("idx", "prev", "ctx")

    def __init__(self, device):
        self.idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
        self.prev, self.ctx = -1, None

    def __enter__(self):
        if _exchange is not None:
            self.prev = _exchange(self.idx)
        else:
            self.ctx = torch.cuda.device(self.idx)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if _exchange is not None:
            if self.prev >= 0 and self.prev != self.idx:
                _exchange(self.prev)
        else:
            self.ctx.__exit__(*exc)
        return False


def contiguous(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()
