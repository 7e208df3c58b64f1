"""ctypes binding of ``libmars5_hip.so`` (C ABI declared in ``include/mars5_hip.h``).

The product path has NO CPU fallback: if the library is missing this module raises at
import (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``mars5-tts_amd/csrc/build.sh``).  Every call returns an int status; ``check`` turns a
non-zero status into a Python exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# M5_HIP_TOOLS=1 (set by tools/*.py before importing the package) loads the tools build of the same sources:
# tuning knobs, A/B kernels and probes live only there (csrc/common.h, m5_tool_env)
TOOLS = os.environ.get("M5_HIP_TOOLS") == "1"
LIB_PATH = os.path.join(_HERE, "libmars5_hip_tools.so" if TOOLS else "libmars5_hip.so")
if TOOLS and os.environ.get("M5_HIP_TOOLS_LIB"):          # same-box A/B of two tools builds (tools/*.py only)
    LIB_PATH = os.environ["M5_HIP_TOOLS_LIB"]



def tool_knob(name: str, default: str) -> str:
    """A/B knob of the host engines (tools/nar_step_bench.py, tools/ar_step_bench.py: "KNOB=0" "KNOB=1" in one process).  Knobs exist
    in the TOOLS configuration only (M5_HIP_TOOLS=1, which tools/*.py set before importing the package): the product reads no
    environment variable besides M5_HIP_TOOLS here and MARS5_DTYPE in model.py (tests/test_host_cpu.py checks the sources), so
    no untested configuration can be selected in a deployment."""
    return os.environ.get(name, default) if TOOLS else default


M5_OK, M5_ERR_ARG, M5_ERR_LAUNCH, M5_ERR_UNSUPPORTED = 0, -1, -2, -3
F32, F16, BF16, F32X3 = 0, 1, 2, 3
EPI_F32, EPI_DT, EPI_RESIDUAL, EPI_SWIGLU, EPI_QKV, EPI_SILU_DT = 0, 1, 2, 3, 4, 5
ST_POS, ST_NGEN, ST_DONE, ST_NTOK, ST_LAST, ST_WORDS = 0, 1, 2, 3, 4, 8
PRO_RMS, PRO_DT, PRO_ATTN = 0, 1, 2
GEPI_QKV_ROPE, GEPI_RESIDUAL, GEPI_SWIGLU, GEPI_F32 = 0, 1, 2, 3
ATTN_PART = 66
NAR_CONSTS = 8

_ERR = {M5_ERR_ARG: "invalid argument", M5_ERR_LAUNCH: "HIP launch / runtime error",
        M5_ERR_UNSUPPORTED: "unsupported shape or option"}


class Mars5HipError(RuntimeError):
    pass


def check(status: int, what: str = "") -> None:
    if status != M5_OK:
        raise Mars5HipError(f"libmars5_hip {what}: status {status} ({_ERR.get(status, 'unknown')})")


vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class QkvScatter(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("vt", vp), ("rows_per_batch", i32), ("n_heads", i32), ("head_dim", i32),
                ("q_bs", i64), ("q_hs", i64), ("q_rs", i64), ("k_bs", i64), ("k_hs", i64), ("k_rs", i64),
                ("vt_bs", i64), ("vt_hs", i64), ("vt_ds", i64)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("q_bs", i64), ("q_hs", i64), ("q_rs", i64),
                ("k", vp), ("k_bs", i64), ("k_hs", i64), ("k_rs", i64),
                ("vt", vp), ("vt_bs", i64), ("vt_hs", i64), ("vt_ds", i64),
                ("o", vp), ("o_bs", i64), ("o_rs", i64),
                ("B", i32), ("H", i32), ("Sq", i32), ("Sk", i32),
                ("key_len", vp), ("causal", i32), ("scale", f32),
                ("kv_index", vp), ("kv_index_stride_k", i64), ("kv_index_stride_v", i64), ("q_len", vp)]


class Prefetch(C.Structure):
    _fields_ = [("ptr", vp), ("chunk_bytes", i64), ("first_chunk", i32), ("n_chunks", i32), ("wgs", i32), ("pad_", i32)]


class GemvArgs(C.Structure):
    _fields_ = [("W", vp), ("ldw", i64), ("N", i32), ("K", i32),
                ("x_f32", vp), ("norm_w", vp), ("eps", f32),
                ("x_dt", vp),
                ("part", vp), ("nsplit", i32), ("n_heads", i32),
                ("y_dt", vp), ("y_f32", vp), ("xres", vp),
                ("rope", vp), ("state", vp),
                ("kcache", vp), ("vcache", vp), ("qbuf", vp),
                ("w_alloc", i32), ("window", i32), ("dim", i32), ("dbg", vp), ("pf", Prefetch)]


class ArMegaArgs(C.Structure):
    _fields_ = [("wqkv", vp), ("wo", vp), ("w13", vp), ("w2", vp),
                ("attn_norm", vp), ("ffn_norm", vp), ("eps", f32),
                ("dim", i32), ("hidden", i32), ("n_heads", i32), ("layer0", i32), ("layer1", i32),
                ("xres", vp), ("rope", vp), ("state", vp),
                ("kcache", vp), ("vcache", vp), ("w_alloc", i32), ("window", i32), ("scale", f32),
                ("gran", vp), ("err", vp), ("dbg", vp)]


AR_MEGA_GRANULES = 21760


class AttnDecodeArgs(C.Structure):
    _fields_ = [("qbuf", vp), ("kcache", vp), ("vcache", vp), ("part", vp), ("state", vp),
                ("n_heads", i32), ("w_alloc", i32), ("window", i32), ("nsplit", i32), ("scale", f32),
                ("batch", i32), ("state_bs", i32), ("q_bs", i64), ("cache_bs", i64), ("part_bs", i64), ("pf", Prefetch)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", vp), ("V", i32),
                ("state", vp), ("tokens", vp), ("max_len", i32),
                ("alpha_frequency", f32), ("alpha_presence", f32), ("penalty_window", i32),
                ("n_text", i32), ("eos_idx", i32),
                ("n_est", i32), ("eos_table", vp),
                ("temperature", f32), ("div_mode", i32),
                ("top_k", i32), ("top_p", f32), ("typical_p", f32),
                ("noise", vp), ("noise_stride", i64),
                ("embed", vp), ("dim", i32), ("xres", vp),
                ("batch", i32), ("state_bs", i32),
                ("logits_bs", i64), ("tokens_bs", i64), ("noise_bs", i64), ("xres_bs", i64), ("eos_table_bs", i64),
                ("n_est_b", vp), ("max_len_b", vp),
                ("rng", vp), ("rng_bs", i64), ("noise_inc", C.c_uint32), ("noise_grid", C.c_uint32)]


class NarSampleArgs(C.Structure):
    _fields_ = [("logits_c", vp), ("logits_u", vp), ("ld_row", i64), ("ld_q", i64),
                ("S", i32), ("n_q", i32), ("K", i32), ("row_offset", i32),
                ("x", vp), ("x_known", vp), ("m", vp),
                ("u1", vp), ("u2", vp),
                ("consts", vp), ("step", vp),
                ("guidance_w", f32), ("temperature", f32), ("log_eps", f32),
                ("div_mode", i32), ("q0_override_steps", i32)]


class NarUniformArgs(C.Structure):
    """M5NarUniformArgs (include/mars5_hip.h): the step's uniforms as torch.rand draws them, generated in the library."""
    _fields_ = [("out", vp), ("n", i64), ("K", i32), ("k_magic", C.c_uint32), ("k_shift", C.c_uint32), ("m", vp),
                ("rng", vp), ("inc", C.c_uint32), ("grid_threads", C.c_uint32), ("step", vp), ("consts", vp), ("transform", i32)]


class RowTiles(C.Structure):
    """M5RowTiles (include/mars5_hip.h): the row tiles of a padded batch layout that hold real rows, per tile height 96 / 128 / 192."""
    _fields_ = [("map", vp * 3), ("n", i32 * 3), ("rows_per_seq", i32), ("seq_len", vp)]


class DeferredLN(C.Structure):
    """M5DeferredLN (include/mars5_hip.h): a LayerNorm deferred into the GEMM that consumes it."""
    _fields_ = [("mode", i32), ("np", i32), ("xt", vp), ("ld_xt", i64), ("part", vp), ("cen_in", vp), ("cen_out", vp), ("delta", vp),
                ("s", vp), ("s_bs", i64), ("eps", f32), ("n_feat", i32), ("rows_bs", i32)]


PLAN_MAX_ARGS = 24


class PlanOp(C.Structure):
    """M5PlanOp (include/mars5_hip.h): one launch of a stage plan -- entry point code + argument slots (stream excluded)."""
    _fields_ = [("fn", i32), ("n_args", i32), ("a", i64 * PLAN_MAX_ARGS)]


class StagePlanC(C.Structure):
    """M5StagePlan: ops in launch order + the arena holding copies of the argument structures they point to."""
    _fields_ = [("ops", C.POINTER(PlanOp)), ("n_ops", i32), ("arena", vp), ("arena_bytes", i64), ("failed_op", C.POINTER(i32))]


# entry points a stage plan may hold -> M5_FN_* code (the enum of include/mars5_hip.h, in order)
PLAN_FN = {name: i + 1 for i, name in enumerate([
    "m5_gemm", "m5_gemm_ex", "m5_layernorm", "m5_layernorm_twice", "m5_layernorm_mean", "m5_rmsnorm", "m5_attention", "m5_gather_rows",
    "m5_chunked_embed", "m5_xattn_absorb", "m5_xattn_scores", "m5_xattn_scores_ex", "m5_nar_uniforms", "m5_nar_sample", "m5_add_int",
    "m5_copy_d2d", "m5_ar_gemv", "m5_ar_attn_decode", "m5_ar_layers_persistent", "m5_ar_sample", "m5_ar_rope_cache_batch",
    "m5_ar_qkv_rope_batch", "m5_ar_attn_combine_batch"])}

# name -> (restype, argtypes); also the list the symbol-export test checks against the header
PROTOTYPES = {
    "m5_version": (C.c_int, []),
    "m5_build_info": (C.c_char_p, []),
    "m5_gemm": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int,
                          C.POINTER(QkvScatter), C.c_int, i64, i64, i64, i64, vp]),
    "m5_xattn_absorb": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, f32, vp]),
    "m5_xattn_scores": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, vp, i64, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "m5_gemm_ex": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.POINTER(QkvScatter), C.c_int, i64, i64, i64, i64, C.POINTER(DeferredLN), C.POINTER(RowTiles), vp]),
    "m5_xattn_scores_ex": (C.c_int, [C.c_int, vp, i64, i64, vp, i64, vp, i64, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(DeferredLN), C.POINTER(RowTiles), vp]),
    "m5_layernorm_twice": (C.c_int, [C.c_int, vp, i64, vp, vp, f32, f32, vp, i64, C.c_int, C.c_int, i64, C.c_int, vp]),
    "m5_layernorm_mean": (C.c_int, [C.c_int, vp, i64, vp, vp, f32, vp, i64, C.c_int, C.c_int, vp, vp]),
    "m5_layernorm": (C.c_int, [C.c_int, vp, i64, vp, vp, f32, vp, i64, C.c_int, C.c_int, C.c_int, i64, i64, vp]),
    "m5_rmsnorm": (C.c_int, [C.c_int, vp, i64, vp, f32, vp, i64, C.c_int, C.c_int, vp]),
    "m5_attention": (C.c_int, [C.c_int, C.POINTER(AttnArgs), vp]),
    "m5_gather_rows": (C.c_int, [vp, i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "m5_chunked_embed": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "m5_rope_cache": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, i64, C.c_int, vp, i64, i64, vp]),
    "m5_ar_gemv": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(GemvArgs), vp]),
    "m5_ar_attn_decode": (C.c_int, [C.c_int, C.POINTER(AttnDecodeArgs), vp]),
    "m5_ar_layers_persistent": (C.c_int, [C.c_int, C.POINTER(ArMegaArgs), vp]),
    "m5_ar_sample": (C.c_int, [C.POINTER(SampleArgs), vp]),
    "m5_ar_rope_cache_batch": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, vp, vp, i32, vp, i64, vp, vp, i64, i64, C.c_int, vp]),
    "m5_ar_qkv_rope_batch": (C.c_int, [C.c_int, vp, i64, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp, i32, vp, i64, vp, vp, i64, i64,
                                        C.c_int, vp, vp]),
    "m5_ar_attn_combine_batch": (C.c_int, [C.c_int, vp, i64, C.c_int, C.c_int, C.c_int, vp, i32, vp, i64, vp]),
    "m5_nar_sample": (C.c_int, [C.POINTER(NarSampleArgs), vp]),
    "m5_nar_uniforms": (C.c_int, [C.POINTER(NarUniformArgs), vp]),
    "m5_expand_tokens": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp, vp]),
    "m5_trim_bounds": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, f32, vp, C.c_int, vp, vp]),
    "m5_add_int": (C.c_int, [vp, i32, vp]),
    "m5_copy_d2d": (C.c_int, [vp, vp, i64, vp]),
    "m5_nar_step": (C.c_int, [C.POINTER(StagePlanC), vp]),
    "m5_ar_decode_step": (C.c_int, [C.POINTER(StagePlanC), vp]),
    "m5_stage_run": (C.c_int, [C.POINTER(StagePlanC), vp]),
    "m5_graph_begin": (C.c_int, [vp]),
    "m5_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "m5_graph_launch": (C.c_int, [vp, vp]),
    "m5_graph_destroy": (C.c_int, [vp]),
    "m5_event_create": (C.c_int, [C.POINTER(vp)]),
    "m5_event_record": (C.c_int, [vp, vp]),
    "m5_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(f32)]),
    "m5_event_destroy": (C.c_int, [vp]),
    "m5_clock_stamp": (C.c_int, [vp, vp]),
}
# exported by libmars5_hip_tools.so only (header: #ifdef M5_TOOLS)
TOOLS_PROTOTYPES = {
    "m5_gemm_q_cross_attn": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, f32, vp, i64, vp]),
    "m5_gemm_residual_ln": (C.c_int, [C.c_int, vp, i64, vp, i64, vp, vp, i64, C.c_int, C.c_int, C.c_int, vp, vp, f32, vp, i64, vp, i64,
                                       vp, C.c_int, vp]),
    "m5_debug_census": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "m5_debug_gemm_clock": (C.c_int, [vp]),
    "m5_debug_feed_probe": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "m5_debug_launch_chain": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "m5_debug_grid_barrier": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "m5_debug_l2_touch": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "m5_debug_edge_probe": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, vp, C.c_int, vp, vp, vp]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MARS5 HIP library is not built. There is no CPU fallback. "
            "Build it with mars5-tts_amd/csrc/build.sh (needs hipcc, cross-compiles gfx950).")
    lib = C.CDLL(LIB_PATH)
    protos = dict(PROTOTYPES)
    if TOOLS:
        protos.update(TOOLS_PROTOTYPES)
    override = TOOLS and bool(os.environ.get("M5_HIP_TOOLS_LIB"))
    for name, (res, args) in protos.items():
        if override and not hasattr(lib, name):
            continue                 # same-box A/B against an OLDER tools build (tools/*.py only): newer entry points are absent there
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.m5_version() != 1:
        raise ImportError(f"libmars5_hip ABI version {lib.m5_version()} != 1")
    return lib


_cdll = _load()


class PlanRecorder:
    """Collects the launches issued through ``lib`` on this thread instead of enqueuing them (ops.StagePlan.recording)."""

    def __init__(self):
        self.ops = []                 # (fn code, [slots])
        self.arena = bytearray()

    def add(self, name: str, argtypes, args) -> int:
        if name not in PLAN_FN:
            raise Mars5HipError(f"{name} cannot be part of a stage plan (include/mars5_hip.h, M5_FN_*)")
        assert len(args) == len(argtypes) and len(args) - 1 <= PLAN_MAX_ARGS, name
        slots = []
        for t, v in zip(argtypes[:-1], args[:-1]):          # the trailing stream is supplied when the plan runs
            if t is vp:
                slots.append(int(v) if v else 0)
            elif t is f32:
                slots.append(int.from_bytes(C.c_float(v), "little"))
            elif isinstance(t, type) and issubclass(t, C._Pointer):     # POINTER(Struct): copy the structure into the arena
                if v is None:
                    slots.append(0)
                else:
                    obj = v._obj if hasattr(v, "_obj") else v.contents
                    while len(self.arena) % 16:
                        self.arena.append(0)
                    slots.append(len(self.arena) + 1)
                    self.arena += bytes(obj)
            else:
                slots.append(int(v))
        self.ops.append((PLAN_FN[name], slots))
        return M5_OK


import threading as _threading

_REC = _threading.local()


class _LibProxy:
    """The loaded library.  A call is enqueued at once -- unless this thread is recording a stage plan, in which case the
    plannable launch entry points are appended to the plan instead (everything else still executes)."""

    def __init__(self, cdll):
        object.__setattr__(self, "_cdll", cdll)
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if name not in PLAN_FN:
            return fn
        w = self._cache.get(name)
        if w is None:
            def w(*args, _fn=fn, _name=name):
                rec = getattr(_REC, "plan", None)
                if rec is not None:
                    return rec.add(_name, _fn.argtypes, args)
                return _fn(*args)
            w.argtypes, w.restype = fn.argtypes, fn.restype
            self._cache[name] = w
        return w


lib = _LibProxy(_cdll)
