"""ctypes binding of libgripnet_hip.so (the C ABI declared in include/gripnet_hip.h).

Nothing here depends on libtorch's ABI: tensors cross the boundary as raw device pointers,
sizes, leading dimensions and the current HIP stream.  There is no CPU fallback: if the
library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

# GN_HIP_LIBRARY selects another build of the same library (the diagnostic `make STAMPS=1` one)
_LIB_PATH = os.environ.get("GN_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                            "libgripnet_hip.so")
_lib = None
_lock = threading.Lock()

GN_OK, GN_ERR_INVALID_ARG, GN_ERR_HIP, GN_ERR_INDEX_RANGE, GN_ERR_UNSUPPORTED, GN_ERR_EDGE_COUNT = range(6)
GN_RGCN_PARTIAL, GN_RGCN_ARITH_FAST, GN_RGCN_PAIR_SUMS_ONLY, GN_RGCN_PAIR_SUMS_READY, GN_RGCN_BASIS_TRANSPOSED = 1, 4, 16, 32, 64        # flags of gn_rgcn_forward_f32
GN_RGCN_PATH_SHIFT = 8
RGCN_PATHS = {"auto": 0, "pair": 1, "lds": 3, "general": 4, "table": 5}                  # kernel choice (tests, measurements)
GN_GEMM_RELU, GN_GEMM_ARITH_FAST, GN_GEMM_B_TRANSPOSED, GN_GEMM_ACCUMULATE, GN_GEMM_A_TRANSPOSED, GN_GEMM_JOIN_BATCH, GN_GEMM_OUT_BF16 = 1, 2, 4, 8, 16, 32, 64                                    # flags of gn_gemm_f32
GN_DM_TYPES_SORTED = 1                                 # flags of gn_distmult_backward_ex_f32
GN_DM_TYPE_TASKS = 2
ABI_VERSION = 154                                       # GN_VERSION of include/gripnet_hip.h this module binds

_p, _i64, _int, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t

# name -> (restype, argtypes); mirrors include/gripnet_hip.h one to one
SIGNATURES = {
    "gn_version": (_int, []),
    "gn_last_error": (C.c_char_p, []),
    "gn_time_next_launch": (_int, [_p, _p]),
    "gn_stream_order": (_int, [_p, _p]),
    "gn_host_scratch_release": (_sz, []),
    "gn_time_launch_pending": (_int, []),
    "gn_gcn_plan_create": (_int, [_p, _p, _p, _i64, _i64, _int, _p, C.POINTER(_p)]),
    "gn_bipartite_plan_create": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_sum_plan_create": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_graph_plan_destroy": (None, [_p]),
    "gn_graph_plan_input_edges": (_i64, [_p]),
    "gn_graph_plan_nnz": (_i64, [_p]),
    "gn_graph_plan_export": (_int, [_p, _p, _p, _p]),
    "gn_graph_aggregate_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _int, _p, _i64, _p, _p, _p]),
    "gn_split_planes_bytes": (_sz, [_i64, _int]),
    "gn_split_planes_f32": (_int, [_p, _i64, _i64, _i64, _i64, _p, _p]),
    "gn_transform_fusable": (_int, [_i64, _i64]),
    "gn_graph_transform_fusable": (_int, [_p, _i64, _i64]),
    "gn_graph_plan_build_blocked": (_int, [_p, _i64, _p]),
    "gn_graph_plan_blocked_cols": (_i64, [_p]),
    "gn_graph_plan_build_transpose": (_int, [_p, _p]),
    "gn_graph_aggregate_t_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _p]),
    "gn_xtg_wide_supported": (_int, [_i64, _i64, _i64]),
    "gn_gemm_f32": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _int, _p]),
    "gn_gemm_addend_f32": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p, _i64, _int, _p]),
    "gn_merge_f32": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _int, _p]),
    "gn_softmax_rows_f32": (_int, [_p, _i64, _i64, _i64, _p]),
    "gn_softmax_rows_backward_f32": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p]),
    "gn_class_scores_f32": (_int, [_p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _int, _p, _i64, _p]),
    "gn_rgcn_plan_create": (_int, [_p, _p, _p, _int, _i64, _i64, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_rgcn_plan_create_ex": (_int, [_p, _p, _p, _int, _i64, _i64, _i64, _i64, _i64, _int, _p, C.POINTER(_p)]),
    "gn_rgcn_plan_destroy": (None, [_p]),
    "gn_rgcn_plan_input_edges": (_i64, [_p]),
    "gn_rgcn_workspace_bytes": (_sz, [_p, _i64, _i64, _i64, _int]),
    "gn_rgcn_forward_path": (_int, [_p, _i64, _i64, _i64, _int]),
    "gn_cast_bf16": (_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "gn_graph_aggregate_bf16": (_int, [_p, _p, _i64, _i64, _p, _int, _p, _i64, _p, _p]),
    "gn_rgcn_forward_f32": (_int, [_p, _p, _i64, _i64, _p, _p, _i64, _p, _p, _i64, _int, _int, _p, _i64, _p, _p, _p, _sz, _p]),
    "gn_rgcn_finalize_f32": (_int, [_p, _p, _i64, _p, _i64, _i64, _p, _p, _i64, _int, _p, _i64, _p, _p]),
    "gn_distmult_forward_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "gn_distmult_packed_forward_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _int, _p, _p, _p]),
    "gn_distmult_plan_create": (_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_distmult_plan_destroy": (None, [_p]),
    "gn_distmult_plan_edges": (_i64, [_p]),
    "gn_distmult_plan_forward_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _int, _p, _p]),
    "gn_distmult_plan_forward_cols_f32": (_int, [_p, _p, _i64, _i64, _i64, _i64, _p, _i64, _int, _p, _p]),
    "gn_xtg_workspace_bytes": (_sz, [_i64, _i64]),
    "gn_xtg_f32": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _p, _i64, _p, _sz, _int, _p]),
    "gn_distmult_backward_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "gn_dense_batch_begin": (_int, []),
    "gn_dense_batch_end": (_int, [_p]),
    "gn_distmult_type_tasks_bytes": (_sz, [_i64, _i64]),
    "gn_distmult_type_tasks": (_int, [_p, _i64, _i64, _p, _sz, _p]),
    "gn_distmult_backward_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "gn_distmult_backward_ex_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _i64, _int, _p, _p, _p, _sz, _p]),
    "gn_distmult_backward_packed_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _i64, _int, _p, _p, _p, _sz, _p]),
    "gn_distmult_bwd_plan_create": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_distmult_bwd_plan_destroy": (None, [_p]),
    "gn_distmult_bwd_plan_workspace_bytes": (_sz, [_p, _i64]),
    "gn_distmult_backward_planned_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "gn_distmult_backward_loss_packed_f32": (_int, [_p, _i64, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _i64, _p, _i64, _int, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "gn_distmult_backward_loss_planned_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _sz, _p]),
    "gn_negative_sampler_create": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_negative_sampler_destroy": (None, [_p]),
    "gn_negative_sampler_sample": (_int, [_p, C.c_uint64, _p, _p, _p, _p]),
    "gn_negative_sampler_sample_packed": (_int, [_p, C.c_uint64, _p, _p, _p, _p, _p]),
    "gn_negative_sampler_sample_stepped": (_int, [_p, C.c_uint64, _p, _p, _p, _p, _p, _p]),
    "gn_rgcn_weight_grad_workspace_bytes": (_sz, [_p, _i64, _i64]),
    "gn_rgcn_weight_grad_supported": (_int, [_p, _i64, _i64]),
    "gn_rgcn_weight_grad_f32": (_int, [_p, _p, _p, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _sz, _p]),
    "gn_rel_grad_plan_create": (_int, [_p, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_rel_grad_plan_destroy": (None, [_p]),
    "gn_rel_weight_grad_supported": (_int, [_p, _i64, _i64]),
    "gn_rel_weight_grad_f32": (_int, [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _p]),
    "gn_grad_prologue_workspace_bytes": (_sz, []),
    "gn_grad_prologue_f32": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _sz, _p]),
    "gn_adam_step_f32": (_int, [_p, _int, _p, _p, _sz, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _p]),
    "gn_class_loss_forward_f32": (_int, [_p, _i64, _p, _i64, _i64, C.c_float, _p, _p, _p]),
    "gn_class_loss_backward_f32": (_int, [_p, _i64, _p, _i64, _i64, C.c_float, _p, _p, _i64, _p]),
    "gn_link_loss_workspace_bytes": (_sz, []),
    "gn_link_loss_forward_f32": (_int, [_p, _i64, _p, _i64, C.c_float, _p, _p, _sz, _p]),
    "gn_link_loss_backward_f32": (_int, [_p, _i64, _p, _i64, C.c_float, _p, _p, _p, _p]),
    "gn_exchange_buffer_bytes": (_sz, [_i64, _int]),
    "gn_exchange_push_f32": (_int, [_p, _i64, _p, _p, _p, _int, _int, _int, _p]),
    "gn_exchange_wait_sum_f32": (_int, [_p, _p, _i64, _int, _int, _p, _int, _p, _p]),
    "gn_link_metrics_plan_create": (_int, [_p, _i64, _i64, _p, C.POINTER(_p)]),
    "gn_link_metrics_plan_destroy": (None, [_p]),
    "gn_link_metrics_plan_workspace_bytes": (_sz, [_p]),
    "gn_link_metrics_planned_f32": (_int, [_p, _p, _p, _p, _p, _sz, _p]),
    "gn_link_metrics_workspace_bytes": (_sz, [_i64, _i64]),
    "gn_link_metrics_f32": (_int, [_p, _p, _p, _i64, _i64, _p, _p, _sz, _p]),
}


class AdamTensor(C.Structure):
    """gn_adam_tensor of include/gripnet_hip.h."""
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("numel", _i64)]


class SideCopy(C.Structure):
    """gn_side_copy: a concat slot filled by the launch that produces its neighbour slot."""
    _fields_ = [("src", _p), ("ld_src", _i64), ("dst", _p), ("ld_dst", _i64), ("rows", _i64), ("cols", _i64),
                ("mode", _int)]


def side_copy(spec):
    """spec = (src, dst, mode) of 2-D fp32 tensors with equal shapes, or None."""
    if spec is None:
        return None
    src, dst, mode = spec
    src = f32_rows(src)
    if tuple(src.shape) != tuple(dst.shape):
        raise ValueError("side copy shapes differ: {} vs {}".format(tuple(src.shape), tuple(dst.shape)))
    sc = SideCopy(src.data_ptr(), ld(src), dst.data_ptr(), ld(dst), src.shape[0], src.shape[1], int(mode))
    sc._keep = (src, dst)
    return sc


class SplitPlanesDesc(C.Structure):
    """gn_split_planes: where a launch leaves the bf16 split planes of what it writes."""
    _fields_ = [("planes", _p), ("rows", _i64), ("nt", _int), ("col_main", _int), ("col_side", _int)]


class SplitPlanes:
    """The bf16 split planes of a row-major fp32 matrix X [rows, 16 nt] (x = hi + mid + lo exactly): what the
    relational layer's matrix products run on, left behind by the layer that produces X (gn_split_planes in
    include/gripnet_hip.h).  Owns the buffer; its last row stays zero."""

    def __init__(self, rows: int, nt: int, device):
        self.rows, self.nt, self.device = int(rows), int(nt), device
        self.buf = torch.zeros((int(load().gn_split_planes_bytes(self.rows, self.nt)),), dtype=torch.uint8, device=device)
        self.generation = 0          # bumped by every launch that rewrites the planes

    def desc(self, col_main=0, col_side=0):
        d = SplitPlanesDesc(self.buf.data_ptr(), self.rows, self.nt, int(col_main), int(col_side))
        d._keep = self.buf
        return d

    def fill_from(self, x: torch.Tensor):
        """Stand-alone split of an existing matrix (callers without a producing layer)."""
        d = self.desc()
        _call("gn_split_planes_f32", ptr(x), ld(x), x.shape[0], x.shape[1], 0, C.byref(d), stream_ptr(x.device))
        self.generation += 1
        return self

    def tag(self, x: torch.Tensor):
        """Remember on `x` that these planes hold its current contents."""
        x._gn_planes = (self, self.generation, x._version)
        return x

    @staticmethod
    def of(x: torch.Tensor, nt: int):
        """The planes tagged on `x`, if they still describe it (same tensor version, not rewritten since), else None."""
        t = getattr(x, "_gn_planes", None)
        if t is None:
            return None
        planes, gen, ver = t
        if planes.generation != gen or x._version != ver or planes.nt != nt or planes.rows != x.shape[0] or planes.device != x.device:
            return None
        return planes


def _ref(sc):
    return None if sc is None else C.byref(sc)


class GripNetHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise RuntimeError(
                    "gripnet_amd: {} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C gripnet_amd/csrc`). There is no CPU fallback.".format(_LIB_PATH))
            lib = C.CDLL(_LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            if lib.gn_version() < ABI_VERSION:
                raise RuntimeError("gripnet_amd: libgripnet_hip.so is older than this package")
            _lib = lib
    return _lib


def check(status: int):
    if status == GN_OK:
        return
    msg = load().gn_last_error().decode("utf-8", "replace")
    if status == GN_ERR_INDEX_RANGE:
        raise IndexError(msg)
    if status == GN_ERR_INVALID_ARG:
        raise ValueError(msg)
    raise GripNetHipError(status, msg)


class KernelTimer:
    """Optional per-entry-point device timing: while active, every launch made through this
    module is bracketed by HIP events on the launching stream (used by bench.py's roofline)."""

    def __init__(self, only=None, pool=0, every=1, records_only=False):
        """`pool`: events created (and recorded once, which is when HIP allocates them) up front, so that a timed
        region pays two event records per bracketed launch and nothing else.  `every`: bracket only every n-th launch
        of an entry point (an event record costs ~4.5 us of stream time on this stack: two per step are 9 % of a 95 us
        step; a sample of the launches gives the same average duration)."""
        self.events = {}
        self.records_only = bool(records_only)      # never the kernels' own dispatch stamps (gn_time_next_launch)
        self.every, self._seen = max(1, int(every)), {}
        self.only = None if only is None else set(only)
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]
        for e in self._pool:
            e.record()

    def event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        global _timer
        self._prev, _timer = _timer, self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = self._prev

    def add(self, name, start, stop):
        self.events.setdefault(name, []).append((start, stop))

    def summary(self):
        """name -> (calls, total_ms); synchronises."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.events.items()}


_timer = None

# Entry points whose kernel can carry its own start / stop events (gn_time_next_launch): no marker packets in the stream.  An
# entry point that turns out to launch another kernel (the events stay pending) is timed with event records from then on.
_STAMPED = {"gn_rgcn_forward_f32", "gn_distmult_plan_forward_f32"}


def _timed_call(t, fn, args, name, tag):
    """One bracketed call under KernelTimer `t`: the kernel's own dispatch stamps where the entry point supports them,
    else an event record in front of and behind the launch (each ~4.5 us of stream time on this stack)."""
    start, stop = t.event(), t.event()
    if name in _STAMPED and not t.records_only and start.cuda_event and stop.cuda_event:      # (a pooled event: its handle exists)
        lib = load()
        lib.gn_time_next_launch(start.cuda_event, stop.cuda_event)
        status = fn(*args)
        if lib.gn_time_launch_pending():                  # another kernel served the call: untimed this once, records from now on
            _STAMPED.discard(name)
            t._pool.extend((start, stop))
        else:
            t.add(tag or name, start, stop)
        return status
    start.record()
    status = fn(*args)
    stop.record()
    t.add(tag or name, start, stop)
    return status


class Recorder:
    """While active, every entry-point call made through this module is also written down - (bound C function, arguments,
    name, tag) - and the tensors whose pointers it was given are held, so that `replay` can make the same calls again straight
    from a loop, without the Python layers above them (module forward, plan look-up, pointer extraction: ~15 us per entry
    point against ~4 us for the call itself).  The calls launch on the stream that was current when they were recorded and
    read / write the buffers they were recorded with, like a captured hipGraph - but a replay is ordinary launches: it
    pipelines behind whatever the stream is still running, where a graph launch on this stack leaves the GPU idle for
    ~9 us in front of its first kernel (tools/launch_modes.py, profiles/r04_counters.md)."""

    def __init__(self):
        self.calls, self.keep = [], []

    def __enter__(self):
        global _recorder
        self._prev, _recorder = _recorder, self
        return self

    def __exit__(self, *exc):
        global _recorder
        _recorder = self._prev


_recorder = None


def replay(calls):
    """Make recorded calls again, in order; an active KernelTimer brackets the ones it names (sampled as in _call)."""
    t = _timer
    for fn, args, name, tag in calls:
        if t is not None and (t.only is None or name in t.only):
            seen = t._seen.get(name, 0)
            t._seen[name] = seen + 1
            if seen % t.every == 0:
                status = _timed_call(t, fn, args, name, tag)
                if status:
                    check(status)
                continue
        status = fn(*args)
        if status:
            check(status)


# Test hooks the library reads from the environment at call time (common.h, aggregate.cuh, gcn_blocked.hip): part of every
# memo key, so that a shortcut recorded under one setting is not replayed under another.
_ENV_HOOKS = ("GN_DISABLE_FAST", "GN_DISABLE_QUAD", "GN_DISABLE_BLOCKED", "GN_BLOCKED_ANY", "GN_DISABLE_LDS_TABLE", "GN_RGCN_BASIS_ORDER")
# hooks the C side reads that can NOT change what a memoised inference forward launches: the host threads of the plan builders
# (plans do not depend on them), the sampler's kernel choice (same draws; no module forward calls the sampler) and the slab size of the
# general relational path (read inside the entry point at every call: a slab is a range of rows, the bits do not depend on it).
# tests/test_abi.py::test_env_hooks_cover_the_library holds the two lists to the getenv calls of csrc/.
_ENV_NEUTRAL = ("GN_PLAN_THREADS", "GN_SAMPLER_TASKS", "GN_RGCN_SLAB_MB")
_ENV_DATA = getattr(os.environ, "_data", None)
_ENV_KEYS = tuple(k.encode() for k in _ENV_HOOKS) if isinstance(_ENV_DATA, dict) and all(isinstance(k, bytes) for k in list(_ENV_DATA)[:1]) else None


def release_host_scratch() -> int:
    """Give the plan builders' kept block of host memory back to the system (gn_host_scratch_release); the bytes freed."""
    return int(load().gn_host_scratch_release())


def env_stamp():
    """The current values of the library's environment hooks (plain dict look-ups on CPython's POSIX `os.environ`)."""
    if _ENV_KEYS is not None:
        d = _ENV_DATA
        return tuple([d.get(k) for k in _ENV_KEYS])
    return tuple([os.environ.get(k) for k in _ENV_HOOKS])


def launch_context(device):
    """(current device, raw stream) a memoised call sequence is bound to."""
    index = device.index
    return torch.cuda.current_device(), (_raw_stream(index) if _raw_stream is not None else torch.cuda.current_stream(index).cuda_stream)


class CallMemo:
    """Steady-state shortcut of ONE module's inference forward (the reference-shaped API of layers.py / decoder.py as a
    caller of GripNet-pose.py:117-138 uses it): the C-ABI calls a forward makes are a function of a few pointers, shapes
    and versions - the module computes that key, and when the key repeats the calls are made again straight from the
    recording (`replay`), without the Python between the module's `forward` and the library (~20 us per entry point:
    module dispatch, plan look-ups, layout checks, ctypes structures).  The OUTPUT is allocated fresh by the module on
    every call (the reference returns fresh tensors) and its address is part of the key: under torch's caching allocator
    a loop's outputs cycle through one or two addresses.  A key is recorded the second time in a row it is seen (one-shot
    inputs - fresh negative samples - never fill the table).  An entry keeps alive what its calls read besides the
    caller's input and the output (plans' scratch, temporaries the slow path allocated, the by-reference structures), and
    the objects whose `id` is part of the key."""

    DEPTH = 8

    def __init__(self):
        self.entries = {}                    # key -> (calls, keep, post)
        self.last = None

    def get(self, key):
        return self.entries.get(key)

    def second_sighting(self, guard):
        seen, self.last = self.last == guard, guard
        return seen

    def record(self, key, run, drop, hold=(), post=None):
        """Run `run()` with every entry-point call written down, store the recording under `key`; `drop`: tensors (the
        caller's input, the fresh output) the entry must NOT keep alive; `hold`: objects it must."""
        global _recorder
        if _recorder is not None:
            return run()
        drop_ptrs = {t.untyped_storage().data_ptr() for t in drop if t is not None}
        with Recorder() as rec:
            result = run()
        keep = list(hold)
        for item in rec.keep:
            if torch.is_tensor(item):
                if item.untyped_storage().data_ptr() not in drop_ptrs:
                    keep.append(item)
                continue
            if isinstance(item, tuple):      # an argument tuple: its by-reference structures stay, what THEY held is filtered too
                for a in item:
                    obj = getattr(a, "_obj", None)
                    held = getattr(obj, "_keep", None) if obj is not None else None
                    if held is not None:
                        for t in (held if isinstance(held, tuple) else (held,)):
                            if torch.is_tensor(t) and t.untyped_storage().data_ptr() not in drop_ptrs:
                                keep.append(t)
                        obj._keep = None
            keep.append(item)
        while len(self.entries) >= self.DEPTH:
            self.entries.pop(next(iter(self.entries)))
        self.entries[key] = (rec.calls, keep, post)
        return result


def _call(name, *args, tag=None):
    """Call an entry point.  Every caller passes `stream_ptr(<device of its operands>)` among the arguments; when that
    device is not the current one the call is made with it current (a launch on a foreign device's stream fails or
    lands on the wrong device)."""
    global _stream_device
    fn = getattr(load(), name)
    dev, _stream_device = _stream_device, None
    if dev is not None and dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            return _call(name, *args, tag=tag)
    t = _timer
    timed = t is not None and (t.only is None or name in t.only)
    if timed and t.every > 1:
        seen = t._seen.get(name, 0)
        t._seen[name] = seen + 1
        timed = seen % t.every == 0
    if timed:
        status = _timed_call(t, fn, args, name, tag)
    else:
        status = fn(*args)
    check(status)
    if _recorder is not None:                        # (calls that were refused - a path that does not apply - are not part of the recording)
        _recorder.calls.append((fn, args, name, tag))
        _recorder.keep.append(args)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "gripnet_amd runs on an MI355X only: got a {} tensor on {} (no CPU fallback; "
                "move the model and the data to 'cuda')".format(tuple(t.shape), t.device))


_side_streams = {}


def side_stream(device) -> "torch.cuda.Stream":
    """One extra HIP stream per device for work that overlaps the main chain (the relational weights)."""
    key = torch.device(device).index
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_stream_device = None       # device index of the last stream_ptr(): read (and cleared) by _call


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (the raw getter costs a fraction of building a Stream object;
    eager launching of the ~120 us forward is close to host-bound)."""
    global _stream_device
    index = device.index if isinstance(device, torch.device) else device
    if index is None:
        index = torch.cuda.current_device()
    _stream_device = index
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def f32_rows(t: torch.Tensor) -> torch.Tensor:
    """fp32 2-D view whose rows are contiguous (any row stride); copies only if it must."""
    if t.dtype != torch.float32:
        raise TypeError("gripnet_amd computes in fp32; got {}".format(t.dtype))
    if t.dim() != 2:
        raise ValueError("expected a 2-D feature matrix, got shape {}".format(tuple(t.shape)))
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def i64_vec(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int64:
        raise TypeError("indices must be int64 (torch.long), got {}".format(t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def edge_rows(edge_index: torch.Tensor):
    """(tensor kept alive, src pointer, dst pointer, E) of a [2, E] int64 edge_index."""
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("edge_index must have shape [2, E], got {}".format(tuple(edge_index.shape)))
    ei = i64_vec(edge_index)
    e = int(ei.shape[1])
    base = ei.data_ptr()
    return ei, base, base + 8 * e, e


def transform_fusable(fin: int, fout: int, x: torch.Tensor) -> bool:
    """True when gn_graph_aggregate_f32 can contract with W in its epilogue for this input."""
    return bool(load().gn_transform_fusable(int(fin), int(fout))) and ld(x) % 4 == 0 and x.data_ptr() % 16 == 0


def e_count(edge_index):
    return int(edge_index.shape[1])


def ptr(t):
    if t is None:
        return None
    if _recorder is not None:
        _recorder.keep.append(t)             # a recorded call's operands stay alive as long as the recording
    return t.data_ptr()


# ---- thin typed wrappers ---------------------------------------------------------------------
def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, bias=None, relu=False, a_rows=None,
         batch=1, stride_a=0, stride_b=0, stride_c=0, m=None, n=None, k=None, lda=None, ldb=None, ldc=None, fast=False,
         b_transposed=False, accumulate=False, a_transposed=False, join_batch=False, addend=None, out_bf16=False):
    """`join_batch`: inside ``with dense_batch(...)`` the product may be queued and leave with the batch (it must not depend
    on another queued product).  `fast`: two-term bf16 splits (<= 2^-16 per product) instead of the default fp32-faithful arithmetic.
    `b_transposed`: b is [n, k] (out = a b^T); `a_transposed`: a is [k, m] (out = a^T b; m <= 64 or n <= 32 only);
    `accumulate`: out += a b; `addend`: an [m, n] fp32 matrix (rows contiguous, any row stride) added to what is stored
    (gn_gemm_addend_f32) - the other gradient of a tensor with two consumers, see `addend_ok`.  `out_bf16`: `out` is a bf16
    table, every value rounded once where it is stored (GN_GEMM_OUT_BF16; raises GripNetHipError with GN_ERR_UNSUPPORTED for
    shapes outside the tall-skinny kernel)."""
    if a_transposed:
        m = a.shape[1] if m is None else m
        k = a.shape[0] if k is None else k
    m = (a.shape[0] if a_rows is None else a_rows.shape[0]) if m is None else m
    k = a.shape[1] if k is None else k
    n = (b.shape[0] if b_transposed else b.shape[-1]) if n is None else n
    join_batch = join_batch and getattr(_batch_tls, 'open', None) is not None
    if join_batch:                                           # a queued product reads its operands when the batch leaves
        _batch_tls.open.keep.extend((a, b, out, bias, a_rows, addend))
    flags = ((GN_GEMM_RELU if relu else 0) | (GN_GEMM_ARITH_FAST if fast else 0) | (GN_GEMM_B_TRANSPOSED if b_transposed else 0) |
             (GN_GEMM_ACCUMULATE if accumulate else 0) | (GN_GEMM_A_TRANSPOSED if a_transposed else 0) | (GN_GEMM_JOIN_BATCH if join_batch else 0) |
             (GN_GEMM_OUT_BF16 if out_bf16 else 0))
    head = (ptr(a), ld(a) if lda is None else lda, stride_a, ptr(a_rows), a.shape[0],
            ptr(b), ld(b) if ldb is None else ldb, stride_b,
            ptr(out), ld(out) if ldc is None else ldc, stride_c, m, n, k, batch, ptr(bias))
    if addend is None:
        _call("gn_gemm_f32", *head, flags, stream_ptr(a.device))
    else:
        if not addend_ok(addend, m, n) or batch != 1:
            raise ValueError("addend: an fp32 [m, n] matrix with contiguous rows, single product")
        _call("gn_gemm_addend_f32", *head, ptr(addend), ld(addend), flags, stream_ptr(a.device))
    return out


def addend_ok(t, m, n) -> bool:
    """Can `t` ride on a product's store as its addend?  (fp32, [m, n], unit column stride - a column slice of a concat's gradient is)"""
    return (t is not None and t.dtype == torch.float32 and t.dim() == 2 and tuple(t.shape) == (m, n) and t.is_cuda and
            (n == 1 or t.stride(1) == 1) and (m == 1 or t.stride(0) >= n))


_xtg_ws = {}
GN_XTG_TICKET_ZEROED, GN_XTG_JOIN_BATCH = 1, 2


def xtg(x: torch.Tensor, g: torch.Tensor, join_batch=False):
    """x^T g for a tall x [m, k1] and g [m, k2] (weight gradients) on gn_xtg_f32: up to 64 x 32 outputs in one launch; wider
    products (the 128 x 64 ... 256 x 128 layers of the node-classification models) as wide products of up to 256 x 128 outputs
    (a launch + a fold) where the sizes allow, else in tiles of 64 x 32 outputs over column slices of x and g; every block
    with a workspace of its own.  `join_batch`: as in `gemm`."""
    k1, k2 = x.shape[1], g.shape[1]
    out = torch.empty((k1, k2), dtype=torch.float32, device=x.device)
    if k1 * k2 == 0:
        return out
    if x.shape[0] == 0:
        return out.zero_()
    if k1 <= 64 and k2 <= 32:
        tiles = [(0, k1, 0, k2)]
    else:
        # blocks of up to 256 x 128 outputs that gn_xtg_f32 takes as ONE wide product (x and g read once); what is left, in
        # tiles of 64 x 32
        tiles, lib, m = [], load(), x.shape[0]
        for c0 in range(0, k1, 256):
            for d0 in range(0, k2, 128):
                w1, w2 = min(256, k1 - c0), min(128, k2 - d0)
                if lib.gn_xtg_wide_supported(m, w1, w2):
                    tiles.append((c0, w1, d0, w2))
                else:
                    tiles.extend((c0 + a, min(64, w1 - a), d0 + b, min(32, w2 - b)) for a in range(0, w1, 64) for b in range(0, w2, 32))
    batched = join_batch and getattr(_batch_tls, 'open', None) is not None
    for t, (c0, w1, d0, w2) in enumerate(tiles):
        need = int(load().gn_xtg_workspace_bytes(w1, w2))
        key = (x.device.index, need, t)
        ws = _xtg_ws.get(key)
        if ws is None:                                         # one zeroed workspace per device, size and tile: its last 64 bytes are the kernel's ticket
            ws = _xtg_ws[key] = torch.zeros((need,), dtype=torch.uint8, device=x.device)
        xs, gs, os_ = (x, g, out) if len(tiles) == 1 else (x[:, c0:c0 + w1], g[:, d0:d0 + w2], out[c0:c0 + w1, d0:d0 + w2])
        if batched:
            _batch_tls.open.keep.extend((xs, gs, os_))
        _call("gn_xtg_f32", ptr(xs), ld(xs), ptr(gs), ld(gs), x.shape[0], w1, w2, ptr(os_), ld(os_), ptr(ws), need,
              GN_XTG_TICKET_ZEROED | (GN_XTG_JOIN_BATCH if batched else 0), stream_ptr(x.device))
    return out


_prologue_ws = {}


def grad_prologue(g, saved_out=None, rowdiv=None, want_masked=True, want_colsum=False):
    """(gm, gd, colsum) of gn_grad_prologue_f32: gm = g masked by saved_out > 0 (None when not wanted), gd = gm / rowdiv
    (None without rowdiv), colsum = the column sums of gm (None when not wanted).  `g` may be a row-strided view."""
    g = f32_rows(g)
    rows, cols = g.shape
    dev = g.device
    gm = torch.empty((rows, cols), dtype=torch.float32, device=dev) if want_masked else None
    gd = torch.empty((rows, cols), dtype=torch.float32, device=dev) if rowdiv is not None else None
    cs = torch.empty((cols,), dtype=torch.float32, device=dev) if want_colsum else None
    ws = None
    if want_colsum:
        key = dev.index
        if key not in _prologue_ws:
            _prologue_ws[key] = torch.zeros((int(load().gn_grad_prologue_workspace_bytes()),), dtype=torch.uint8, device=dev)
        ws = _prologue_ws[key]
    _call("gn_grad_prologue_f32", ptr(g), ld(g), ptr(saved_out), 0 if saved_out is None else ld(saved_out), ptr(rowdiv), rows, cols,
          ptr(gm), cols, ptr(gd), cols, ptr(cs), ptr(ws), 0 if ws is None else ws.numel(), stream_ptr(dev))
    return gm, gd, cs


def merge(dst: torch.Tensor, src: torch.Tensor, mode: int, src2=None):
    _call("gn_merge_f32", ptr(dst), ld(dst), ptr(src), ld(src), ptr(src2), 0 if src2 is None else ld(src2),
          dst.shape[0], dst.shape[1], mode, stream_ptr(dst.device))
    return dst


class GraphPlan:
    """Owner of a gn_graph_plan handle (the cached normalised graph of one GCN-style layer)."""

    def __init__(self, handle, device, kind):
        self._h, self.device, self.kind = handle, device, kind
        self._export = None

    @classmethod
    def gcn(cls, edge_index, num_nodes, edge_weight=None, improved=False):
        lib = load()
        require_gpu(edge_index, edge_weight)
        ei, src, dst, e = edge_rows(edge_index)
        w = None if edge_weight is None else edge_weight.to(torch.float32).contiguous()
        if w is not None and w.numel() != e:
            raise ValueError("edge_weight has {} entries for {} edges".format(w.numel(), e))
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_gcn_plan_create(src, dst, ptr(w), e, int(num_nodes), int(bool(improved)),
                                         stream_ptr(ei.device), C.byref(h)))
        plan = cls(h, ei.device, "gcn")
        plan.n_rows = plan.n_table = int(num_nodes)
        return plan

    @classmethod
    def bipartite(cls, edge_index, num_sources, num_targets, edge_weight=None):
        lib = load()
        require_gpu(edge_index, edge_weight)
        ei, src, dst, e = edge_rows(edge_index)
        w = None if edge_weight is None else edge_weight.to(torch.float32).contiguous()
        if w is not None and w.numel() != e:
            raise ValueError("edge_weight has {} entries for {} edges".format(w.numel(), e))
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_bipartite_plan_create(src, dst, ptr(w), e, int(num_sources), int(num_targets),
                                               stream_ptr(ei.device), C.byref(h)))
        plan = cls(h, ei.device, "bipartite")
        plan.n_rows, plan.n_table = int(num_targets), int(num_sources)
        return plan

    @classmethod
    def plain_sum(cls, edge_index, num_sources, num_targets):
        """out[t] = sum of table[s] over the edges s -> t (no normalisation)."""
        lib = load()
        require_gpu(edge_index)
        ei, src, dst, e = edge_rows(edge_index)
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_sum_plan_create(src, dst, None, e, int(num_sources), int(num_targets),
                                         stream_ptr(ei.device), C.byref(h)))
        plan = cls(h, ei.device, "sum")
        plan.n_rows, plan.n_table = int(num_targets), int(num_sources)
        return plan

    def build_blocked(self, cols: int):
        """Add the source-blocked encoding (LDS-staged gathers) for rows of up to `cols` floats; a no-op for
        graphs that do not qualify.  Returns the width the plan covers (0: wave-per-row kernels)."""
        if self.kind == "gcn" and cols <= 32:
            with torch.cuda.device(self.device):
                check(load().gn_graph_plan_build_blocked(self._h, 16 if cols <= 16 else 32, stream_ptr(self.device)))
        return self.blocked_cols

    @property
    def blocked_cols(self) -> int:
        return int(load().gn_graph_plan_blocked_cols(self._h))

    def blocked_ok(self, fin: int, fout: int, x: torch.Tensor) -> bool:
        """True when gn_graph_aggregate_f32(weight=W) runs on the source-blocked kernels for this input."""
        return (fout in (16, 32) and fout <= self.blocked_cols and fin in (16, 32, 64) and ld(x) % 4 == 0
                and x.data_ptr() % 16 == 0)

    def transform_ok(self, fin: int, fout: int, x: torch.Tensor) -> bool:
        """True when gn_graph_aggregate_f32(weight=W) contracts with W in the gather's launch for this plan and input."""
        return (bool(load().gn_graph_transform_fusable(self._h, int(fin), int(fout))) and ld(x) % 4 == 0
                and x.data_ptr() % 16 == 0)

    @property
    def input_edges(self) -> int:
        return int(load().gn_graph_plan_input_edges(self._h))

    @property
    def nnz(self) -> int:
        return int(load().gn_graph_plan_nnz(self._h))

    def export(self):
        """(edge_index' [2,E'] int64, norm [E'] fp32) in the reference's order."""
        if self._export is None:
            n = self.nnz
            ei = torch.empty((2, n), dtype=torch.int64, device=self.device)
            nrm = torch.empty((n,), dtype=torch.float32, device=self.device)
            check(load().gn_graph_plan_export(self._h, ptr(ei), ptr(nrm), stream_ptr(self.device)))
            self._export = (ei, nrm)
        return self._export

    def __iter__(self):          # lets `edge_index, norm = conv.cached_result` keep working
        return iter(self.export())

    def aggregate(self, xw: torch.Tensor, bias, relu: bool, out: torch.Tensor, side=None, weight=None, planes=None):
        """out = act(A_norm xw + b), or with `weight` act((A_norm xw) weight + b) (xw is then the layer input).
        `planes` = (SplitPlanes, col_main, col_side): the launch also leaves the bf16 split planes of its output and of
        its side copy."""
        sc = side_copy(side)
        pd = None
        if planes is not None:
            pd = planes[0].desc(planes[1], planes[2])
            planes[0].generation += 1
        _call("gn_graph_aggregate_f32", self._h, ptr(xw), ld(xw), xw.shape[1], ptr(weight),
              0 if weight is None else weight.shape[1], ptr(bias), int(bool(relu)),
              ptr(out), ld(out), _ref(sc), _ref(pd), stream_ptr(xw.device), tag="gn_graph_aggregate_f32[{}]".format(self.kind))
        return out

    def aggregate_bf16(self, xw: torch.Tensor, bias, relu: bool, out: torch.Tensor, side=None):
        """out = act(A_norm bf16(xw) + b): the gathered table is rounded to bf16 once and read at half the bytes;
        sums, bias, activation and `out` are fp32.  `xw` is the fp32 product (rounded here by gn_cast_bf16) or already the
        bf16 table (written by the product's own store, GN_GEMM_OUT_BF16: no extra pass)."""
        if xw.dtype == torch.bfloat16:
            table = xw
        else:
            table = torch.empty(xw.shape, dtype=torch.bfloat16, device=xw.device)
            _call("gn_cast_bf16", ptr(xw), ld(xw), ptr(table), ld(table), xw.shape[0], xw.shape[1], stream_ptr(xw.device))
        sc = side_copy(side)
        _call("gn_graph_aggregate_bf16", self._h, ptr(table), ld(table), xw.shape[1], ptr(bias), int(bool(relu)),
              ptr(out), ld(out), _ref(sc), stream_ptr(xw.device), tag="gn_graph_aggregate_bf16[{}]".format(self.kind))
        return out

    def aggregate_t(self, g: torch.Tensor, out: torch.Tensor):
        """out[s] = sum over edges leaving s of coef * g[dst]: gradient of `aggregate` w.r.t. its table."""
        if not getattr(self, "_has_t", False):
            check(load().gn_graph_plan_build_transpose(self._h, stream_ptr(g.device)))
            self._has_t = True
        _call("gn_graph_aggregate_t_f32", self._h, ptr(g), ld(g), g.shape[1], ptr(out), ld(out), stream_ptr(g.device))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_graph_plan_destroy(h)


class RgcnPlan:
    """Owner of a gn_rgcn_plan handle (static multi-relational graph of one supervertex)."""

    def __init__(self, edge_index, range_list, num_nodes, edge_lo=None, edge_hi=None, light=False):
        """`light`: only what the general O(E) kernel reads (GN_RGCN_PLAN_LIGHT) - no host-built schedules: the plan of an edge
        list that will not be seen again."""
        lib = load()
        require_gpu(edge_index)
        ei, src, dst, e = edge_rows(edge_index)
        rl = torch.as_tensor(range_list).to("cpu", torch.int64).contiguous()   # tiny; host copy once
        if rl.dim() != 2 or rl.shape[1] != 2:
            raise ValueError("range_list must have shape [R, 2], got {}".format(tuple(rl.shape)))
        lo = 0 if edge_lo is None else int(edge_lo)
        hi = e if edge_hi is None else int(edge_hi)
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_rgcn_plan_create_ex(src, dst, rl.data_ptr(), 1, rl.shape[0], e, int(num_nodes), lo, hi, 1 if light else 0,
                                             stream_ptr(ei.device), C.byref(h)))
        self.light = bool(light)
        self._h, self.device = h, ei.device
        self.num_nodes, self.num_relations, self.num_edges = int(num_nodes), int(rl.shape[0]), e
        self.edge_lo, self.edge_hi = lo, hi
        self._ws = None
        self._sums, self._sums_stream = None, None            # the pair sums of the split launches (start_pair_sums)
        self._edge_index, self._range_list, self._grad, self._wgrad = ei, rl, None, None

    def grad_plans(self):
        """(reversed-graph relational plan, (relation, source)-major sum plan, in-degree divisor) for the
        backward pass; built on first use.  For a shard (edge_lo / edge_hi) the two plans cover the shard's edges only -
        its share of the sums over edges - while the divisor is the in-degree over the FULL graph, as in the forward."""
        if self._grad is None:
            ei, n, R = self._edge_index, self.num_nodes, self.num_relations
            lo, hi = self.edge_lo, self.edge_hi
            rev = RgcnPlan(ei.flip(0).contiguous(), self._range_list, n, lo, hi)   # same edge positions, endpoints swapped
            sizes = (self._range_list[:, 1] - self._range_list[:, 0]).to(self.device)
            rel = torch.repeat_interleave(torch.arange(R, device=self.device), sizes)      # relation of every edge
            keyed = torch.stack([ei[1], rel * n + ei[0]])[:, lo:hi].contiguous()           # dst -> (relation, src) row
            pairs = GraphPlan.plain_sum(keyed, n, R * n)
            deg = torch.zeros(n, dtype=torch.float32, device=self.device)
            deg.index_add_(0, ei[1], torch.ones(e_count(ei), dtype=torch.float32, device=self.device))
            self._grad = (rev, pairs, deg.clamp_(min=1.0))
        return self._grad

    def weight_grad_plan(self):
        """RelGradPlan of this layer's (shard's) edges, or None where the fused kernel does not apply (> 65534 nodes)."""
        if self._wgrad is None:
            try:
                self._wgrad = RelGradPlan(self.grad_plans()[1], self.num_nodes, self.num_relations)
            except GripNetHipError as err:
                if err.status != GN_ERR_UNSUPPORTED:
                    raise
                self._wgrad = False
        return self._wgrad or None

    def general_weight_grad(self, x, gm):
        """dw [R, fin * fout] = sum over each relation's edges of x[src]^T gm[dst] on gn_rgcn_weight_grad_f32 (any number of
        nodes), or None where it does not cover the shapes."""
        fin, fout = x.shape[1], gm.shape[1]
        lib = load()
        if not lib.gn_rgcn_weight_grad_supported(self._h, fin, fout):
            return None
        ei = self._edge_index
        dw = torch.empty((self.num_relations, fin * fout), dtype=torch.float32, device=x.device)
        need = int(lib.gn_rgcn_weight_grad_workspace_bytes(self._h, fin, fout))
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
        base = ei.data_ptr()
        _call("gn_rgcn_weight_grad_f32", self._h, base, base + 8 * self.num_edges, ptr(x), ld(x), fin, ptr(gm), ld(gm), fout, ptr(dw),
              ptr(ws), need, stream_ptr(x.device))
        if _recorder is not None:
            _recorder.keep.append(ei)
        return dw

    def _workspace(self, fin, fout, bases, flags=0):
        need = int(load().gn_rgcn_workspace_bytes(self._h, fin, fout, bases, flags))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=self.device)
        return self._ws, need

    @staticmethod
    def mode_flags(fast=False, path="auto"):
        """Arithmetic and kernel-choice bits of gn_rgcn_forward_f32."""
        return (GN_RGCN_ARITH_FAST if fast else 0) | (RGCN_PATHS[path] << GN_RGCN_PATH_SHIFT)

    def path(self, fin, fout, bases, fast=False, path="auto"):
        """Name of the kernel a forward with these shapes and flags takes."""
        code = int(load().gn_rgcn_forward_path(self._h, fin, fout, bases, self.mode_flags(fast, path)))
        return {v: k for k, v in RGCN_PATHS.items()}.get(code, "?")

    def forward(self, x, basis, att, root, bias, relu, out, partial=False, side=None, fast=False, path="auto", x_planes=None,
                pair_sums=False, basis_transposed=False):
        """`x_planes`: SplitPlanes of x left by its producer (the destination-major kernel then skips its own split of x;
        the other kernels ignore them).  `pair_sums`: the sums of this step are on their way (start_pair_sums): the launch is
        ordered behind them and contracts them (GN_RGCN_PAIR_SUMS_READY).  `basis_transposed`: `basis` is [bases, out, in] (the
        forward's parameter seen from the reversed layer of its backward; destination-major kernel only: GN_RGCN_BASIS_TRANSPOSED)."""
        mode = self.mode_flags(fast, path)
        if basis_transposed:
            if pair_sums:
                raise ValueError("basis_transposed does not combine with pair_sums")
            fout = basis.shape[1]
            ws, need = self._workspace(x.shape[1], fout, basis.shape[0], mode)
            sc = side_copy(side)
            flags = (GN_RGCN_PARTIAL if partial else 0) | mode | GN_RGCN_BASIS_TRANSPOSED
            _call("gn_rgcn_forward_f32", self._h, ptr(x), ld(x), x.shape[1], ptr(basis), ptr(att), basis.shape[0],
                  ptr(root), ptr(bias), fout, int(bool(relu)), flags,
                  ptr(out), ld(out), _ref(sc), None if x_planes is None else x_planes.buf.data_ptr(), ptr(ws), need,
                  stream_ptr(x.device))
            return out
        if pair_sums:
            sc = side_copy(side)
            flags = (GN_RGCN_PARTIAL if partial else 0) | mode | GN_RGCN_PAIR_SUMS_READY
            _call("gn_stream_order", self._sums_stream, stream_ptr(x.device))
            _call("gn_rgcn_forward_f32", self._h, ptr(x), ld(x), x.shape[1], ptr(basis), ptr(att), basis.shape[0],
                  ptr(root), ptr(bias), basis.shape[2], int(bool(relu)), flags,
                  ptr(out), ld(out), _ref(sc), x_planes.buf.data_ptr(), ptr(self._sums), self._sums.numel(),
                  stream_ptr(x.device))
            return out
        ws, need = self._workspace(x.shape[1], basis.shape[2], basis.shape[0], mode)
        sc = side_copy(side)
        flags = (GN_RGCN_PARTIAL if partial else 0) | mode
        _call("gn_rgcn_forward_f32", self._h, ptr(x), ld(x), x.shape[1], ptr(basis), ptr(att), basis.shape[0],
              ptr(root), ptr(bias), basis.shape[2], int(bool(relu)), flags,
              ptr(out), ld(out), _ref(sc), None if x_planes is None else x_planes.buf.data_ptr(), ptr(ws), need,
              stream_ptr(x.device))
        return out

    def pair_sums_supported(self, x, basis, x_planes, fast=False, path="auto"):
        """Can this layer run as two launches (the x-independent pair sums, then the contraction)?"""
        if x_planes is None or fast or path not in ("auto", "pair"):
            return False
        return self.path(x.shape[1], basis.shape[2], basis.shape[0], False, path) == "pair"

    def start_pair_sums(self, att, fin, fout, bases, stream):
        """The x-INDEPENDENT half of the layer on `stream` (a torch.cuda.Stream; GN_RGCN_PAIR_SUMS_ONLY): reads `att` and the
        plan only.  Ordered behind what the current stream has been given so far (att may just have been written there);
        `forward(..., pair_sums=True)` orders itself behind it.  Every step: the sums are never kept across steps."""
        flags = GN_RGCN_PAIR_SUMS_ONLY
        need = int(load().gn_rgcn_workspace_bytes(self._h, fin, fout, bases, flags))
        if self._sums is None or self._sums.numel() < need:
            self._sums = torch.empty((max(need, 1),), dtype=torch.uint8, device=self.device)
        main = stream_ptr(att.device)
        side = stream.cuda_stream
        _call("gn_stream_order", main, side)
        _call("gn_rgcn_forward_f32", self._h, None, fin, fin, None, ptr(att), bases, None, None, fout, 0, flags,
              None, fout, None, None, ptr(self._sums), need, side, tag="gn_rgcn_forward_f32[pair sums]")
        self._sums_stream = side
        return self._sums

    def finalize(self, summed, x, root, bias, relu, out, side=None):
        sc = side_copy(side)
        _call("gn_rgcn_finalize_f32", self._h, ptr(summed), ld(summed), ptr(x), ld(x), x.shape[1], ptr(root),
              ptr(bias), root.shape[1], int(bool(relu)), ptr(out), ld(out), _ref(sc), stream_ptr(x.device))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_rgcn_plan_destroy(h)


_error_flags = {}


def error_flag(device) -> torch.Tensor:
    """Per-device int32 word the decoder kernels OR index-range errors into (checked lazily)."""
    key = (device.type, device.index)
    if key not in _error_flags:
        _error_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _error_flags[key]


def raise_if_index_errors(device=None):
    """Synchronising check of the decoder error word; raises IndexError like the reference's
    advanced indexing would (gripnet/decoder.py:20)."""
    for key, flag in list(_error_flags.items()):
        if device is not None and (device.type, device.index) != key:
            continue
        bits = int(flag.item())
        if bits != 0:
            flag.zero_()
            if bits & 4:
                raise RuntimeError("one-shot exchange: a peer's partial sums did not arrive within the timeout")
            if bits & 2:
                raise RuntimeError("negative sampler: a relation's positive pairs leave no pair to draw")
            raise IndexError("DistMult decoder saw an edge endpoint or relation id outside its table")


def distmult(z, u_v, edge_type, weight, sigmoid, out):
    ei, u, v, e = edge_rows(u_v)
    et = i64_vec(edge_type)
    if et.numel() != e:
        raise ValueError("edge_type has {} entries for {} edges".format(et.numel(), e))
    _call("gn_distmult_forward_f32", ptr(z), ld(z), z.shape[0], z.shape[1], u, v, ptr(et), ptr(weight),
          ld(weight), weight.shape[0], e, int(bool(sigmoid)), ptr(out),
          ptr(error_flag(z.device)), stream_ptr(z.device))
    return out


_relation_ids = []          # (edge_type tensor, _version, int16 copy) of the last few edge_type tensors asked about


def relation_ids16(et):
    """The relation ids of a (static) edge_type tensor as 16-bit values, narrowed once per tensor and version."""
    for t, ver, r16 in _relation_ids:
        if t is et and ver == et._version:
            return r16
    r16 = et.to(torch.int16)
    _relation_ids.append((et, et._version, r16))
    del _relation_ids[:-4]
    return r16


def packed_pairs(edge_index):
    """The (uint32 words, as int32) that NegativeSampler.sample left next to an edge list, or None (another list, or
    the list was modified since)."""
    tag = getattr(edge_index, "_gn_packed", None)
    if tag is None or tag[1] != edge_index._version:
        return None
    return tag[0]


def distmult_packed(z, packed, edge_type, weight, sigmoid, out):
    e = int(packed.numel())
    if edge_type.numel() != e:
        raise ValueError("edge_type has {} entries for {} edges".format(edge_type.numel(), e))
    r16 = relation_ids16(edge_type)
    _call("gn_distmult_packed_forward_f32", ptr(z), ld(z), z.shape[0], z.shape[1], ptr(packed), ptr(r16), ptr(weight),
          ld(weight), weight.shape[0], e, int(bool(sigmoid)), ptr(out), ptr(error_flag(z.device)), stream_ptr(z.device))
    return out


def distmult_any(z, u_v, edge_type, weight, sigmoid, out):
    """The plan-less decoder: on the packed pairs a NegativeSampler left with `u_v` when there are any (and ids fit 16
    bits), else on the raw int64 triples."""
    packed = packed_pairs(u_v)
    if packed is not None and z.shape[0] <= 65535 and weight.shape[0] <= 32767:
        try:
            return distmult_packed(z, packed, edge_type, weight, sigmoid, out)
        except GripNetHipError as err:                     # node table too large for the LDS: the general kernels
            if err.status != GN_ERR_UNSUPPORTED:
                raise
    return distmult(z, u_v, edge_type, weight, sigmoid, out)


class DistMultPlan:
    """Owner of a gn_distmult_plan handle: one static (edge_index, edge_type) list, validated, packed and ordered
    for the LDS-resident decoder kernel (the positive edges a training loop scores every epoch)."""

    def __init__(self, u_v, edge_type, num_nodes, num_relations, num_features=0):
        """`num_features` (the decoder's in_dim; 0: unknown) lets the plan add the row-class encoding: whole rows of a
        class of nodes in LDS, one table fill per launch (k_distmult_class)."""
        lib = load()
        require_gpu(u_v, edge_type)
        ei, u, v, e = edge_rows(u_v)
        et = i64_vec(edge_type)
        if et.numel() != e:
            raise ValueError("edge_type has {} entries for {} edges".format(et.numel(), e))
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_distmult_plan_create(u, v, ptr(et), e, int(num_nodes), int(num_relations), int(num_features),
                                              stream_ptr(ei.device), C.byref(h)))
        self._h, self.device, self.num_edges = h, ei.device, e
        self.num_nodes, self.num_relations = int(num_nodes), int(num_relations)

    def forward(self, z, weight, sigmoid, out):
        _call("gn_distmult_plan_forward_f32", self._h, ptr(z), ld(z), z.shape[1], ptr(weight), ld(weight),
              int(bool(sigmoid)), ptr(out), stream_ptr(z.device))
        return out

    def forward_cols(self, z, num_features, col_lo, col_hi, weight, sigmoid, out):
        """Feature columns [col_lo, col_hi) of the scores (z may hold the first col_hi columns only); the launch with
        col_lo = 0 starts the sums in `out`, the one with col_hi = num_features finishes them."""
        _call("gn_distmult_plan_forward_cols_f32", self._h, ptr(z), ld(z), int(num_features), int(col_lo), int(col_hi),
              ptr(weight), ld(weight), int(bool(sigmoid)), ptr(out), stream_ptr(z.device))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_distmult_plan_destroy(h)


class RelGradPlan:
    """Owner of a gn_rel_grad_plan handle: the edge-dependent part of the relational layer's weight gradient
    (dW_r = X^T Q_r in one launch), built from the layer's (relation, source)-major sum plan."""

    def __init__(self, sums: "GraphPlan", num_nodes, num_relations):
        lib = load()
        h = _p()
        with torch.cuda.device(sums.device):
            check(lib.gn_rel_grad_plan_create(sums._h, int(num_nodes), int(num_relations), stream_ptr(sums.device), C.byref(h)))
        self._h, self.device, self.num_relations = h, sums.device, int(num_relations)

    def supported(self, fin, fout) -> bool:
        return bool(load().gn_rel_weight_grad_supported(self._h, int(fin), int(fout)))

    def weight_grad(self, x, gm, out=None):
        """dw [R, fin * fout]: dw[r] = sum over the edges e of relation r of x[src_e]^T gm[dst_e]."""
        fin, fout = x.shape[1], gm.shape[1]
        if out is None:
            out = torch.empty((self.num_relations, fin * fout), dtype=torch.float32, device=x.device)
        _call("gn_rel_weight_grad_f32", self._h, ptr(x), ld(x), fin, ptr(gm), ld(gm), fout, ptr(out), stream_ptr(x.device))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_rel_grad_plan_destroy(h)


class DistMultBwdPlan:
    """Owner of a gn_distmult_bwd_plan handle: the gradient-independent part of the decoder's backward pass for one
    static (edge_index, edge_type) list."""

    def __init__(self, u_v, edge_type, num_nodes, num_relations):
        lib = load()
        require_gpu(u_v, edge_type)
        ei, u, v, e = edge_rows(u_v)
        et = i64_vec(edge_type)
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_distmult_bwd_plan_create(u, v, ptr(et), e, int(num_nodes), int(num_relations),
                                                  stream_ptr(ei.device), C.byref(h)))
        self._h, self.device, self.num_edges = h, ei.device, e
        self.num_nodes, self.num_relations = int(num_nodes), int(num_relations)

    def backward(self, z, weight, grad_logit, dz, dd, probs=None, loss=None):
        """`loss` (a LinkLossGrad; then grad_logit is None and probs the forward's probabilities): the scores feed the link loss
        directly - its derivative is computed where the records are built (gn_distmult_backward_loss_planned_f32)."""
        n_grad = probs.numel() if loss is not None else grad_logit.numel()
        if n_grad != self.num_edges:
            raise ValueError("the plan was built from {} edges, got {} gradients".format(self.num_edges, n_grad))
        need = int(load().gn_distmult_bwd_plan_workspace_bytes(self._h, z.shape[1]))
        ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=z.device)
        if loss is not None:
            _call("gn_distmult_backward_loss_planned_f32", self._h, ptr(z), ld(z), z.shape[1], ptr(weight), ld(weight),
                  loss.ref(), ptr(probs), ptr(dz), ld(dz), ptr(dd), ld(dd), ptr(ws), need, stream_ptr(z.device))
            return dz, dd
        _call("gn_distmult_backward_planned_f32", self._h, ptr(z), ld(z), z.shape[1], ptr(weight), ld(weight),
              ptr(grad_logit), ptr(probs), ptr(dz), ld(dz), ptr(dd), ld(dd), ptr(ws), need, stream_ptr(z.device))
        return dz, dd

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_distmult_bwd_plan_destroy(h)


_sorted_types = []          # (tensor, _version, relations, offsets or None) of the last few edge_type tensors asked about


_batch_tls = threading.local()      # the open batch of THIS thread: the library's queue is thread_local too (autograd runs a
                                    # device's backward on its own thread; two GPUs in one process must not see each other's batch)


class dense_batch:
    """``with dense_batch(device):`` - the deep-and-narrow `gemm` calls and the one-launch `xtg` calls inside leave as ONE
    launch at the end of the block (gn_dense_batch_begin / _end); they must not depend on each other.  The operands of
    every `gemm` / `xtg` call inside are kept alive until the batch has been launched (a temporary freed in between could
    be handed out again and overwritten before the queued product reads it)."""

    def __init__(self, device):
        self.device, self.keep = device, []

    def __enter__(self):
        self.off = os.environ.get("GN_DENSE_BATCH") == "0"     # (development switch: every product launches on its own)
        if self.off:
            return self
        _call("gn_dense_batch_begin")
        _batch_tls.open = self
        return self

    def __exit__(self, *exc):
        if self.off:
            return False
        _batch_tls.open = None
        try:
            with torch.cuda.device(self.device):
                _call("gn_dense_batch_end", stream_ptr(self.device))
        finally:
            self.keep = []
        return False


def type_offsets(et, num_relations):
    """[R + 1] int32 offsets of the relations in a non-decreasing edge_type tensor, or None if it is not sorted.  One
    device reduction and host read per tensor (and version): the reference passes the same `train_et` with the
    positive and the negative edges of every epoch."""
    for t, ver, r, off in _sorted_types:
        if t is et and ver == et._version and r == num_relations:
            return off
    off = None
    if et.numel() < 2 or bool((et[1:] >= et[:-1]).all()):
        bounds = torch.arange(num_relations + 1, device=et.device, dtype=et.dtype)
        first = torch.searchsorted(et.contiguous(), bounds).to(torch.int32)
        # the offsets and, behind them, the task list of the decoder gradient's relation-major pass (GN_DM_TYPE_TASKS)
        lib = load()
        need = int(lib.gn_distmult_type_tasks_bytes(num_relations, et.numel()))
        off = torch.empty((need // 4,), dtype=torch.int32, device=et.device)
        with torch.cuda.device(et.device):
            _call("gn_distmult_type_tasks", ptr(first), num_relations, et.numel(), ptr(off), need, stream_ptr(et.device))
    _sorted_types.append((et, et._version, num_relations, off))
    del _sorted_types[:-4]
    return off


class LinkLossGrad:
    """gn_link_loss_grad: where the decoder's backward gets d loss / d probability from when its scores feed the link loss of
    GripNet-pose.py:140-142 directly (upstream: the loss's one upstream gradient, a device scalar, or None for 1)."""

    class _S(C.Structure):
        _fields_ = [("upstream", C.c_void_p), ("eps", C.c_float), ("negative", C.c_int)]

    def __init__(self, upstream, eps, negative):
        self.upstream = upstream                               # (kept alive with the call's operands)
        self._s = LinkLossGrad._S(None if upstream is None else upstream.data_ptr(), float(eps), 1 if negative else 0)

    def ref(self):
        if _recorder is not None:
            _recorder.keep.append(self)
        return C.addressof(self._s)


def distmult_backward_loss_packed(z, u_v, edge_type, weight, probs, dz, dd, loss, dz_add=None, dd_add=None):
    """The loss-fed backward on the sampler's packed pairs (gn_distmult_backward_loss_packed_f32), or False where that path
    does not apply (no packed pairs, unsorted types, tables beyond the counting-sort path): the caller takes the two-step path.
    `dz_add` / `dd_add`: the other list's gradients of z and D, added where this call stores its sums (they may be dz / dd)."""
    e = e_count(u_v)
    offsets = type_offsets(edge_type, weight.shape[0])
    packed = packed_pairs(u_v)
    if packed is None or offsets is None or z.shape[0] > 65535 or weight.shape[0] > 32767:
        return False
    need = int(load().gn_distmult_backward_workspace_bytes(z.shape[0], z.shape[1], weight.shape[0], e))
    ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=z.device)
    try:
        _call("gn_distmult_backward_loss_packed_f32", ptr(z), ld(z), z.shape[0], z.shape[1], ptr(packed), ptr(relation_ids16(edge_type)),
              ptr(weight), ld(weight), weight.shape[0], e, loss.ref(), ptr(probs), ptr(dz), ld(dz), ptr(dd), ld(dd),
              GN_DM_TYPES_SORTED | GN_DM_TYPE_TASKS, ptr(offsets), ptr(dz_add), 0 if dz_add is None else ld(dz_add),
              ptr(dd_add), 0 if dd_add is None else ld(dd_add), ptr(ws), need, stream_ptr(z.device))
    except GripNetHipError as err:
        if err.status != GN_ERR_UNSUPPORTED:
            raise
        return False
    return True


def distmult_backward(z, u_v, edge_type, weight, grad_logit, dz, dd, probs=None):
    """`probs`: the sigmoid scores of the forward; grad_logit is then the gradient with respect to them."""
    ei, u, v, e = edge_rows(u_v)
    et = i64_vec(edge_type)
    offsets = type_offsets(edge_type, weight.shape[0])
    flags = (GN_DM_TYPES_SORTED | GN_DM_TYPE_TASKS) if offsets is not None else 0
    need = int(load().gn_distmult_backward_workspace_bytes(z.shape[0], z.shape[1], weight.shape[0], e))
    ws = torch.empty((max(need, 1),), dtype=torch.uint8, device=z.device)
    packed = packed_pairs(u_v)
    if packed is not None and offsets is not None and z.shape[0] <= 65535 and weight.shape[0] <= 32767:
        # the sampler's 32-bit pairs + the static edge_type's 16-bit ids: 6 instead of 24 bytes per edge and sort pass
        try:
            _call("gn_distmult_backward_packed_f32", ptr(z), ld(z), z.shape[0], z.shape[1], ptr(packed), ptr(relation_ids16(edge_type)),
                  ptr(weight), ld(weight), weight.shape[0], e, ptr(grad_logit), ptr(dz), ld(dz), ptr(dd), ld(dd), flags, ptr(probs),
                  ptr(offsets), ptr(ws), need, stream_ptr(z.device))
            return dz, dd
        except GripNetHipError as err:                     # tables too large for the counting-sort path: the int64 call
            if err.status != GN_ERR_UNSUPPORTED:
                raise
    _call("gn_distmult_backward_ex_f32", ptr(z), ld(z), z.shape[0], z.shape[1], u, v, ptr(et), ptr(weight), ld(weight),
          weight.shape[0], e, ptr(grad_logit), ptr(dz), ld(dz), ptr(dd), ld(dd), flags, ptr(probs), ptr(offsets), ptr(ws),
          need, stream_ptr(z.device))
    return dz, dd


class NegativeSampler:
    """Device-side typed negative sampling for one static positive edge list (reference:
    gripnet/utils.py:98-119, called once per epoch at GripNet-pose.py:131).

        sampler = NegativeSampler(data.train_idx, n_d_node, data.train_range)
        neg_index = sampler.sample(seed=epoch)          # [2, E] int64 on the GPU, no host round trip
    """

    def __init__(self, pos_edge_index, num_nodes, range_list=None):
        lib = load()
        require_gpu(pos_edge_index)
        ei, u, v, e = edge_rows(pos_edge_index)
        if range_list is None:
            range_list = [[0, e]]
        rl = torch.as_tensor(range_list).to("cpu", torch.int64).contiguous().view(-1, 2)
        h = _p()
        with torch.cuda.device(ei.device):
            check(lib.gn_negative_sampler_create(u, v, rl.data_ptr(), rl.shape[0], e, int(num_nodes),
                                                 stream_ptr(ei.device), C.byref(h)))
        self._h, self.device, self.num_edges, self.num_nodes = h, ei.device, e, int(num_nodes)

    def sample(self, seed: int = 0, out=None, step=None) -> torch.Tensor:
        """[2, E] int64 negative pairs.  For graphs of up to 65,535 nodes the same launch also leaves every pair as one
        32-bit word; it travels with the returned tensor (`_gn_packed`) and the decoder scores the list from it - 6
        instead of 24 bytes per edge - as long as the tensor is not modified.

        `step`: a one-element int64 tensor on the device.  The draw then is the one of ``seed + step`` and the counter is
        advanced by one behind it, on the device: a CAPTURED training step that contains this call draws new negatives at
        every replay (gn_negative_sampler_sample_stepped)."""
        if step is not None and (step.dtype != torch.int64 or step.numel() != 1 or step.device != self.device):
            raise ValueError("`step` must be a one-element int64 tensor on {}".format(self.device))
        if out is None:
            out = torch.empty((2, self.num_edges), dtype=torch.int64, device=self.device)
        elif tuple(out.shape) != (2, self.num_edges) or out.dtype != torch.int64 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("`out` must be a contiguous [2, {}] int64 tensor on {}".format(self.num_edges, self.device))
        else:
            out._gn_volatile = True          # refilled behind torch's back (`_version` does not move): never a static list
        base = out.data_ptr()
        if self.num_nodes <= 65535 and self.num_edges > 0:
            # (a refilled `out` keeps its packed words' buffer: a captured step that scores `out` replays on the new draw)
            held = getattr(out, "_gn_packed", None)
            packed = held[0] if held is not None else torch.empty((self.num_edges,), dtype=torch.int32, device=self.device)
            if step is not None:
                _call("gn_negative_sampler_sample_stepped", self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(step), base,
                      base + 8 * self.num_edges, ptr(packed), ptr(error_flag(self.device)), stream_ptr(self.device))
                out._gn_packed = (packed, out._version)
                return out
            _call("gn_negative_sampler_sample_packed", self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, base, base + 8 * self.num_edges,
                  ptr(packed), ptr(error_flag(self.device)), stream_ptr(self.device))
            out._gn_packed = (packed, out._version)
            return out
        if step is not None:
            _call("gn_negative_sampler_sample_stepped", self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(step), base,
                  base + 8 * self.num_edges, None, ptr(error_flag(self.device)), stream_ptr(self.device))
            return out
        _call("gn_negative_sampler_sample", self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, base, base + 8 * self.num_edges,
              ptr(error_flag(self.device)), stream_ptr(self.device))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_negative_sampler_destroy(h)


class MetricsPlan:
    """Owner of a gn_link_metrics_plan handle: the segments of one range list (what relation_metrics needs besides the scores)."""

    def __init__(self, range_list, device):
        rl = torch.as_tensor(range_list).to("cpu", torch.int64).contiguous().view(-1, 2)
        self.R = int(rl.shape[0])
        self.E = int(rl[-1, 1]) if self.R else 0
        self.device = torch.device(device)
        h = _p()
        with torch.cuda.device(self.device):
            check(load().gn_link_metrics_plan_create(rl.data_ptr(), self.R, self.E, stream_ptr(self.device), C.byref(h)))
        self._h = h
        self.workspace_bytes = int(load().gn_link_metrics_plan_workspace_bytes(h))
        self._ws = None

    def workspace(self):
        if self._ws is None:
            self._ws = torch.empty((max(self.workspace_bytes, 1),), dtype=torch.uint8, device=self.device)
        return self._ws

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.gn_link_metrics_plan_destroy(h)


_metric_plans = []          # (range_list object, _version, device, MetricsPlan) of the last few range lists


def metrics_plan(range_list, device):
    """The MetricsPlan of a range list, kept while the same tensor (unmodified) keeps coming: an epoch loop's train / test ranges."""
    ver = getattr(range_list, "_version", None)
    dev = torch.device(device)
    for rl, v, d, plan in _metric_plans:
        if rl is range_list and v == ver and d == dev:
            return plan
    plan = MetricsPlan(range_list, dev)
    if ver is not None:                                  # (a list / array has no version to watch: not kept)
        _metric_plans.insert(0, (range_list, ver, dev, plan))
        del _metric_plans[4:]
    return plan


def link_metrics(pos_score, neg_score, range_list):
    """(auprc, auroc, ap), each a float64 [R] tensor on the GPU: scikit-learn's three link-prediction
    metrics for every relation block at once (reference: one sklearn call per relation and epoch).  Asynchronous."""
    require_gpu(pos_score, neg_score)
    pos = pos_score.detach().to(torch.float32).contiguous()
    neg = neg_score.detach().to(torch.float32).contiguous()
    if pos.numel() != neg.numel():
        raise ValueError("positive and negative score lists differ in length")
    plan = metrics_plan(range_list, pos.device)
    if plan.E != int(pos.numel()):
        raise ValueError("range_list covers {} edges, {} scores given".format(plan.E, int(pos.numel())))
    out = torch.empty((3, plan.R), dtype=torch.float64, device=pos.device)
    ws = plan.workspace()
    _call("gn_link_metrics_planned_f32", plan._h, ptr(pos), ptr(neg), ptr(out), ptr(ws), ws.numel(), stream_ptr(pos.device))
    return out[0], out[1], out[2]


def class_scores(z, weight, nodes, out, softmax=True):
    """out = softmax?(z[nodes] @ weight) in one launch (gn_class_scores_f32)."""
    _call("gn_class_scores_f32", ptr(z), ld(z), z.shape[0], ptr(nodes), out.shape[0], ptr(weight), ld(weight),
          weight.shape[0], weight.shape[1], int(bool(softmax)), ptr(out), ld(out), stream_ptr(z.device))
    return out


def softmax_rows(x):
    _call("gn_softmax_rows_f32", ptr(x), ld(x), x.shape[0], x.shape[1], stream_ptr(x.device))
    return x
