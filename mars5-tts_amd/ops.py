"""Tensor-level wrappers over the C ABI.  torch is plumbing here: device memory
(``data_ptr``), streams and shapes.  No arithmetic happens in this file; every function
enqueues exactly one HIP kernel on ``stream`` (default: torch's current stream)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import threading

import torch

from . import _lib as L
from ._lib import check, lib

DT_CODE = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16}
CODE_DT = {v: k for k, v in DT_CODE.items()}

# How the fp32 engines multiply (process-wide; read when a launch is ENQUEUED, so a captured graph keeps the mode it was
# captured with): "exact" = v_mfma_f32_16x16x4_f32, bitwise an fmaf chain -- the reference instrument; "f16x3" = the same fp32
# operands split into two f16 halves on the fly and three f16 MFMAs per product (csrc/gemm.hip, M5_F32X3): fp32-grade results
# (operand error 2^-22) at several times the rate -- the fast parity-grade mode.  Set with set_f32_products().
_F32_PRODUCTS = "exact"


def set_f32_products(mode: str) -> str:
    """Select the fp32 engines' GEMM arithmetic ("exact" | "f16x3"); returns the previous mode."""
    global _F32_PRODUCTS
    assert mode in ("exact", "f16x3"), mode
    prev, _F32_PRODUCTS = _F32_PRODUCTS, mode
    return prev


def f32_products() -> str:
    return _F32_PRODUCTS


def _gemm_code(dt: torch.dtype) -> int:
    return L.F32X3 if (dt == torch.float32 and _F32_PRODUCTS == "f16x3") else DT_CODE[dt]
DT_NAME = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "libmars5_hip operates on device memory only"
    return t.data_ptr()


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _s(stream: Optional[int]) -> int:
    return cur_stream() if stream is None else stream


def gemm(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor], epi: int, bias: Optional[torch.Tensor] = None,
         scatter: Optional[L.QkvScatter] = None, M: Optional[int] = None, ldc: Optional[int] = None,
         batch: int = 1, sA: int = 0, sW: int = 0, sC: int = 0, sBias: int = 0, stream: Optional[int] = None) -> None:
    """out = a[M,K] @ w[N,K]^T (+bias) with fused epilogue `epi`.  a/w: 2-D, K contiguous."""
    assert a.dtype == w.dtype and a.stride(-1) == 1 and w.stride(-1) == 1
    Mv = a.shape[-2] if M is None else M
    N, K = w.shape[-2], w.shape[-1]
    assert a.shape[-1] == K
    if bias is not None:
        assert bias.dtype == torch.float32
    ld = ldc if ldc is not None else (out.stride(-2) if out is not None else 0)
    check(lib.m5_gemm(_gemm_code(a.dtype), _p(a), a.stride(-2), _p(w), w.stride(-2), _p(bias), _p(out), ld, Mv, N, K, epi,
                      C.byref(scatter) if scatter is not None else None, batch, sA, sW, sC, sBias, _s(stream)), "m5_gemm")


def gemm_dln(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor], epi: int, dl: Optional[L.DeferredLN], bias: Optional[torch.Tensor] = None,
             scatter: Optional[L.QkvScatter] = None, M: Optional[int] = None, ldc: Optional[int] = None,
             batch: int = 1, sA: int = 0, sW: int = 0, sC: int = 0, sBias: int = 0, stream: Optional[int] = None,
             rt: Optional[L.RowTiles] = None) -> None:
    """``gemm`` with a deferred LayerNorm (include/mars5_hip.h, M5DeferredLN): epi RESIDUAL = the producer (also writes the
    centred copy + row partials), QKV / SWIGLU = a consumer (applies the normalisation in its epilogue); and / or over a
    row-tile list `rt` (M5RowTiles: only the tiles that hold real rows of a padded batch are launched)."""
    assert a.dtype == w.dtype and a.stride(-1) == 1 and w.stride(-1) == 1
    Mv = a.shape[-2] if M is None else M
    N, K = w.shape[-2], w.shape[-1]
    assert a.shape[-1] == K
    if bias is not None:
        assert bias.dtype == torch.float32
    ld = ldc if ldc is not None else (out.stride(-2) if out is not None else 0)
    check(lib.m5_gemm_ex(DT_CODE[a.dtype], _p(a), a.stride(-2), _p(w), w.stride(-2), _p(bias), _p(out), ld, Mv, N, K, epi,
                         C.byref(scatter) if scatter is not None else None, batch, sA, sW, sC, sBias, C.byref(dl) if dl is not None else None,
                         C.byref(rt) if rt is not None else None, _s(stream)), "m5_gemm_ex")


def xattn_scores_dln(x: torch.Tensor, sX: int, a_tab: torch.Tensor, c_tab: torch.Tensor, p_out: torch.Tensor, sP: int, M: int, n_heads: int,
                     Lp: int, batch: int, dl: Optional[L.DeferredLN], stream: Optional[int] = None, rt: Optional[L.RowTiles] = None) -> None:
    """``xattn_scores`` on the centred copy x with the LayerNorm applied in the epilogue (dl.mode = 2) and / or over a row-tile list."""
    N, K = n_heads * Lp, x.shape[-1]
    check(lib.m5_xattn_scores_ex(DT_CODE[x.dtype], _p(x), x.stride(-2), sX, _p(a_tab), N * K, _p(c_tab), N, _p(p_out), p_out.stride(-2), sP,
                                 M, n_heads, Lp, K, batch, C.byref(dl) if dl is not None else None, C.byref(rt) if rt is not None else None,
                                 _s(stream)), "m5_xattn_scores_ex")


def layernorm_mean(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor, mean_out: torch.Tensor,
                   M: Optional[int] = None, stream: Optional[int] = None) -> None:
    """``layernorm`` that also leaves the row means (fp32) in mean_out: the first centre of a deferred-LayerNorm chain."""
    assert x.dtype == torch.float32 and mean_out.dtype == torch.float32 and x.stride(-1) == 1 and out.stride(-1) == 1
    Mv = x.shape[0] if M is None else M
    assert mean_out.numel() >= Mv
    check(lib.m5_layernorm_mean(DT_CODE[out.dtype], _p(x), x.stride(0), _p(gamma), _p(beta), eps, _p(out), out.stride(-2), Mv, x.shape[1],
                                _p(mean_out), _s(stream)), "m5_layernorm_mean")


def layernorm_twice(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, eps2: float, out: torch.Tensor, rows_per_seq: int,
                    n_seq: int = 1, x_seq_stride: int = 0, stream: Optional[int] = None) -> None:
    """out (n_seq * rows_per_seq, D) = normalise(LayerNorm(x; gamma, beta, eps); eps2), no second affine; run s of x starts
    x_seq_stride rows after run s - 1 (include/mars5_hip.h)."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and out.stride(-1) == 1
    check(lib.m5_layernorm_twice(DT_CODE[out.dtype], _p(x), x.stride(0), _p(gamma), _p(beta), eps, eps2, _p(out), out.stride(-2), rows_per_seq, n_seq,
                                 x_seq_stride, x.shape[1], _s(stream)), "m5_layernorm_twice")


def gemm_q_cross_attn(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], n_heads: int, mem_table: torch.Tensor, max_le: int,
                      rows_per_seq: int, step: torch.Tensor, scale: float, out: torch.Tensor, stream: Optional[int] = None) -> bool:
    """out = cross-attention of the projected queries a @ w^T + bias against short pre-projected memories, one launch.
    mem_table (n_seq, 6) int64 device (include/mars5_hip.h).  False if the shape is not eligible (nothing launched)."""
    assert a.dtype == w.dtype == out.dtype and mem_table.dtype == torch.int64 and mem_table.is_contiguous()
    if L.tool_knob("M5_GEMM_XATTN", "0") != "1":    # tools build only (the product library does not export it): measured slower inside the NAR step (csrc/gemm16.hip)
        return False
    st = lib.m5_gemm_q_cross_attn(DT_CODE[a.dtype], _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), a.shape[0], n_heads, w.shape[1],
                                  _p(mem_table), max_le, rows_per_seq, _p(step), scale, _p(out), out.stride(0), _s(stream))
    if st == L.M5_ERR_UNSUPPORTED:
        return False
    check(st, "m5_gemm_q_cross_attn")
    return True


def xattn_absorb(dtype: torch.dtype, tab_seq: torch.Tensor, tab_layer: torch.Tensor, n_layers: int, n_seq: int, n_heads: int, D: int, Lp: int,
                 step: torch.Tensor, scale: float, stream: Optional[int] = None) -> None:
    """Build the absorbed cross-attention operands (A, c, B^T) of every (layer, sequence) for the memory block of step *step."""
    assert tab_seq.dtype == torch.int64 and tab_layer.dtype == torch.int64 and tab_seq.is_contiguous() and tab_layer.is_contiguous()
    check(lib.m5_xattn_absorb(DT_CODE[dtype], _p(tab_seq), _p(tab_layer), n_layers, n_seq, n_heads, D, Lp, _p(step), scale, _s(stream)),
          "m5_xattn_absorb")


def xattn_scores(x: torch.Tensor, sX: int, a_tab: torch.Tensor, c_tab: torch.Tensor, p_out: torch.Tensor, sP: int, M: int, n_heads: int,
                 Lp: int, batch: int, stream: Optional[int] = None) -> None:
    """p_out[b] = per-head softmax(x[b] a_tab[b]^T + c_tab[b]): x (rows, K) with batch stride sX rows*K elements given in elements."""
    N, K = n_heads * Lp, x.shape[-1]
    check(lib.m5_xattn_scores(DT_CODE[x.dtype], _p(x), x.stride(-2), sX, _p(a_tab), N * K, _p(c_tab), N, _p(p_out), p_out.stride(-2), sP,
                              M, n_heads, Lp, K, batch, _s(stream)), "m5_xattn_scores")


def gemm_residual_ln(a: torch.Tensor, w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor,
                     beta: torch.Tensor, eps: float, xn: torch.Tensor, scratch: torch.Tensor, tag: int = 0,
                     tag_step: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> bool:
    """x += a @ w^T + bias and xn = LayerNorm(x) in one launch.  False if the shape is not eligible for the fused kernel
    (nothing was launched; run gemm + layernorm instead).  Consecutive calls on one `scratch` need different launch tags
    (tag, or the device counter `tag_step` under graph replay): see include/mars5_hip.h."""
    assert a.dtype == w.dtype == xn.dtype and x.dtype == torch.float32 and scratch.dtype == torch.uint8
    if L.tool_knob("M5_GEMM_LN", "0") != "1":       # tools build only (the product library does not export it): measured slower than GEMM + LayerNorm launches (csrc/gemm16.hip)
        return False
    N, K = w.shape
    st = lib.m5_gemm_residual_ln(DT_CODE[a.dtype], _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x), x.stride(0), a.shape[0], N, K,
                                 _p(gamma), _p(beta), eps, _p(xn), xn.stride(0), _p(scratch), scratch.numel(), _p(tag_step), tag, _s(stream))
    if st == L.M5_ERR_UNSUPPORTED:
        return False
    check(st, "m5_gemm_residual_ln")
    return True


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor,
              n_affine: int = 1, affine_stride: int = 0, y_affine_stride: int = 0, M: Optional[int] = None,
              stream: Optional[int] = None) -> None:
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and out.stride(-1) == 1
    Mv = x.shape[0] if M is None else M
    ldy = out.stride(-2)
    check(lib.m5_layernorm(DT_CODE[out.dtype], _p(x), x.stride(0), _p(gamma), _p(beta), eps, _p(out), ldy, Mv, x.shape[1],
                           n_affine, affine_stride, y_affine_stride, _s(stream)), "m5_layernorm")


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: torch.Tensor, stream: Optional[int] = None) -> None:
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    check(lib.m5_rmsnorm(DT_CODE[out.dtype], _p(x), x.stride(0), _p(w), eps, _p(out), out.stride(0), x.shape[0], x.shape[1],
                         _s(stream)), "m5_rmsnorm")


def attention(dtype: torch.dtype, args: L.AttnArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_attention(_gemm_code(dtype), C.byref(args), _s(stream)), "m5_attention")


def gather_rows(out: torch.Tensor, table: torch.Tensor, idx: torch.Tensor, alpha: Optional[torch.Tensor] = None,
                pe: Optional[torch.Tensor] = None, pos: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
                add_idx: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> None:
    assert out.dtype == torch.float32 and table.dtype == torch.float32 and idx.dtype == torch.int64
    R, D = out.shape
    assert table.shape[1] == D and table.is_contiguous()
    check(lib.m5_gather_rows(_p(out), out.stride(0), R, D, _p(table), _p(idx), _p(alpha), _p(pe), _p(pos), _p(add),
                             _p(add_idx), _s(stream)), "m5_gather_rows")


def chunked_embed(out: torch.Tensor, tables: torch.Tensor, codes: Optional[torch.Tensor], lead_row: Optional[torch.Tensor],
                  alpha: Optional[torch.Tensor], pe: Optional[torch.Tensor], add: Optional[torch.Tensor] = None,
                  add_index: Optional[torch.Tensor] = None, rows: Optional[int] = None, stream: Optional[int] = None) -> None:
    """out (n_rep, Rr, D) fp32, the first R = `rows` (default Rr) rows of every rep are written;
    tables (n_q, n_codes, D/n_q); codes (R-lead, n_q) int64."""
    assert out.dtype == torch.float32 and out.is_contiguous() and tables.is_contiguous()
    n_rep, Rr, D = out.shape
    R = Rr if rows is None else rows
    assert R <= Rr
    n_q, n_codes, _ = tables.shape
    if codes is not None:
        assert codes.dtype == torch.int64 and codes.is_contiguous()
    check(lib.m5_chunked_embed(_p(out), Rr * D, n_rep, R, D, n_q, n_codes, _p(tables), _p(codes), _p(lead_row), _p(alpha),
                               _p(pe), _p(add), _p(add_index), _s(stream)), "m5_chunked_embed")


def rope_cache(qkv: torch.Tensor, n_heads: int, pos0: int, rope: torch.Tensor, q_out: torch.Tensor, kcache: torch.Tensor,
               vcache: torch.Tensor, cache_hs: int, window: int, vt_out: torch.Tensor, vt_hs: int, vt_ds: int,
               stream: Optional[int] = None) -> None:
    check(lib.m5_rope_cache(DT_CODE[qkv.dtype], _p(qkv), qkv.shape[0], n_heads, pos0, _p(rope), _p(q_out), _p(kcache),
                            _p(vcache), cache_hs, window, _p(vt_out), vt_hs, vt_ds, _s(stream)), "m5_rope_cache")


def ar_gemv(dtype: torch.dtype, pro: int, epi: int, args: L.GemvArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_ar_gemv(DT_CODE[dtype], pro, epi, C.byref(args), _s(stream)), "m5_ar_gemv")


def ar_layers_persistent(dtype: torch.dtype, args: L.ArMegaArgs, stream: Optional[int] = None) -> int:
    """All Mistral layers of one decode step as one persistent launch (csrc/ar_mega.hip).  Returns the C status: M5_OK, or
    M5_ERR_UNSUPPORTED when the geometry / device does not allow it (the caller then enqueues the per-launch form)."""
    rc = lib.m5_ar_layers_persistent(DT_CODE[dtype], C.byref(args), _s(stream))
    if rc not in (L.M5_OK, L.M5_ERR_UNSUPPORTED):
        check(rc, "m5_ar_layers_persistent")
    return rc


def ar_rope_cache_batch(qkv: torch.Tensor, n_heads: int, rope: torch.Tensor, state: torch.Tensor, qbuf: torch.Tensor,
                        kcache: torch.Tensor, vcache: torch.Tensor, cache_bs: int, cache_hs: int, window: int,
                        stream: Optional[int] = None) -> None:
    """qkv (B, 3D) dtype; state (B, ST_WORDS) int32; qbuf (B, D); k/vcache: this layer's [h][W][64] block of
    sequence 0, sequence b at + b*cache_bs elements."""
    B = qkv.shape[0]
    check(lib.m5_ar_rope_cache_batch(DT_CODE[qkv.dtype], _p(qkv), B, n_heads, _p(rope), _p(state), state.stride(0), _p(qbuf),
                                     qbuf.stride(0), _p(kcache), _p(vcache), cache_bs, cache_hs, window, _s(stream)),
          "m5_ar_rope_cache_batch")


def ar_qkv_rope_batch(xn: torch.Tensor, wqkv: torch.Tensor, n_heads: int, rope: torch.Tensor, state: torch.Tensor, qbuf: torch.Tensor,
                      kcache: torch.Tensor, vcache: torch.Tensor, cache_bs: int, cache_hs: int, window: int, qkv_tmp: torch.Tensor,
                      stream: Optional[int] = None) -> None:
    """Batched decode QKV projection + RoPE + cache write: xn (B, K) dtype, wqkv (3D, K)."""
    check(lib.m5_ar_qkv_rope_batch(DT_CODE[xn.dtype], _p(xn), xn.stride(0), _p(wqkv), wqkv.stride(0), xn.shape[0], n_heads, xn.shape[1],
                                   _p(rope), _p(state), state.stride(0), _p(qbuf), qbuf.stride(0), _p(kcache), _p(vcache), cache_bs,
                                   cache_hs, window, _p(qkv_tmp), _s(stream)), "m5_ar_qkv_rope_batch")


def ar_attn_combine_batch(part: torch.Tensor, n_heads: int, nsplit: int, state: torch.Tensor, out: torch.Tensor,
                          stream: Optional[int] = None) -> None:
    """part (B, H, nsplit, ATTN_PART) fp32 -> out (B, D) dtype."""
    B = part.shape[0]
    check(lib.m5_ar_attn_combine_batch(DT_CODE[out.dtype], _p(part), part.stride(0), B, n_heads, nsplit, _p(state), state.stride(0),
                                       _p(out), out.stride(0), _s(stream)), "m5_ar_attn_combine_batch")


def ar_attn_decode(dtype: torch.dtype, args: L.AttnDecodeArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_ar_attn_decode(DT_CODE[dtype], C.byref(args), _s(stream)), "m5_ar_attn_decode")


def ar_sample(args: L.SampleArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_ar_sample(C.byref(args), _s(stream)), "m5_ar_sample")


def nar_sample(args: L.NarSampleArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_nar_sample(C.byref(args), _s(stream)), "m5_nar_sample")


def nar_uniforms(args: L.NarUniformArgs, stream: Optional[int] = None) -> None:
    check(lib.m5_nar_uniforms(C.byref(args), _s(stream)), "m5_nar_uniforms")


def expand_tokens(tokens: torch.Tensor, n_text: int, off: torch.Tensor, vals: torch.Tensor, max_run: int,
                  stream: Optional[int] = None) -> torch.Tensor:
    """AR token ids (n,) int64 -> codebook-0 frames (G,) int64 through the CSR expansion table (off int32 (V+1,), vals int64):
    one launch, then a 4-byte read-back of G (the NAR buffers are sized by it)."""
    assert tokens.dtype == torch.int64 and off.dtype == torch.int32 and vals.dtype == torch.int64 and tokens.is_contiguous()
    n = int(tokens.shape[0])
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=tokens.device)
    cap = max(n * max(max_run, 1), 1)
    out = torch.empty(cap, dtype=torch.int64, device=tokens.device)
    total = torch.zeros(1, dtype=torch.int32, device=tokens.device)
    check(lib.m5_expand_tokens(_p(tokens), n, n_text, _p(off), _p(vals), off.shape[0] - 1, _p(out), cap, _p(total), _s(stream)), "m5_expand_tokens")
    return out[: int(total.item())]


def trim_bounds(y: torch.Tensor, top_db: float, frame_length: int = 2048, hop_length: int = 512, stream: Optional[int] = None) -> torch.Tensor:
    """[start, end) (int32 (2,), device) of the non-silent part of the mono fp32 device signal y (n,): two launches."""
    assert y.dtype == torch.float32 and y.dim() == 1 and y.is_contiguous()
    n = int(y.shape[0])
    nf = 1 + n // hop_length
    power = torch.empty(nf, dtype=torch.float32, device=y.device)
    bounds = torch.zeros(2, dtype=torch.int32, device=y.device)
    check(lib.m5_trim_bounds(_p(y), n, frame_length, hop_length, float(top_db), _p(power), nf, _p(bounds), _s(stream)), "m5_trim_bounds")
    return bounds


def copy_d2d(dst: torch.Tensor, src: torch.Tensor, stream: Optional[int] = None) -> None:
    """dst = src (contiguous device tensors of the same size and dtype) as a stream-ordered device copy of this library --
    capturable, and plannable (a torch ``copy_`` is neither visible to a stage plan nor free of an ATen launch)."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
    check(lib.m5_copy_d2d(_p(dst), _p(src), dst.numel() * dst.element_size(), _s(stream)), "m5_copy_d2d")


class StagePlan:
    """A stage of the hot path as ONE C call (include/mars5_hip.h: m5_nar_step / m5_ar_decode_step / m5_stage_run).  The
    host engine runs its enqueue code once under ``recording()`` -- every launch that would have gone to the library is
    appended to the plan instead -- and from then on ``run(stream)`` enqueues the whole stage from C.  The tensors the
    recorded launches point to must stay alive and in place (the engines' sessions own them); `keep` holds extra references."""

    KINDS = {"nar_step": "m5_nar_step", "ar_decode_step": "m5_ar_decode_step", "stage": "m5_stage_run"}

    def __init__(self, kind: str):
        assert kind in self.KINDS, kind
        self.kind, self._rec, self._c, self.keep = kind, None, None, []

    def recording(self):
        plan = self

        class _Ctx:
            def __enter__(self_c):
                assert getattr(L._REC, "plan", None) is None, "a stage plan is already being recorded on this thread"
                plan._rec = L.PlanRecorder()
                L._REC.plan = plan._rec
                return plan

            def __exit__(self_c, *exc):
                L._REC.plan = None
                if exc[0] is None:
                    plan._finish()
                return False
        return _Ctx()

    def _finish(self) -> None:
        rec = self._rec
        n = len(rec.ops)
        arr = (L.PlanOp * max(n, 1))()
        for i, (fn, slots) in enumerate(rec.ops):
            arr[i].fn, arr[i].n_args = fn, len(slots)
            for j, v in enumerate(slots):
                arr[i].a[j] = v if v < (1 << 63) else v - (1 << 64)
        arena = (C.c_ubyte * max(len(rec.arena), 1)).from_buffer_copy(bytes(rec.arena) or b"\0")
        self._failed = L.i32(-1)
        self._arr, self._arena = arr, arena
        self._c = L.StagePlanC(ops=arr, n_ops=n, arena=C.cast(arena, C.c_void_p), arena_bytes=len(rec.arena), failed_op=C.pointer(self._failed))
        self.n_ops = n

    def run(self, stream: Optional[int] = None, raise_on_error: bool = True) -> int:
        assert self._c is not None, "record the plan first"
        rc = getattr(lib, self.KINDS[self.kind])(C.byref(self._c), _s(stream))
        if rc != L.M5_OK and raise_on_error:
            check(rc, f"{self.KINDS[self.kind]} (op {int(self._failed.value)} of {self.n_ops})")
        return rc


def add_int(p: torch.Tensor, delta: int, stream: Optional[int] = None) -> None:
    assert p.dtype == torch.int32
    check(lib.m5_add_int(_p(p), delta, _s(stream)), "m5_add_int")


def mark(label: str, stream: Optional[int] = None) -> None:
    """Hook in front of launches that do not go through this module (torch copies inside a captured step): a no-op in the
    product; bench.py's in-graph timing replaces it with a clock stamp so that those launches get their own interval."""
    return None


def clock_stamp(slots: torch.Tensor, i: int, stream: Optional[int] = None) -> None:
    """slots[i] (int64, device) = the 100 MHz wall clock when this one-lane launch runs (in-graph timing, bench roofline leg)."""
    assert slots.dtype == torch.int64 and slots.is_cuda and 0 <= i < slots.numel()
    check(lib.m5_clock_stamp(slots.data_ptr() + 8 * i, _s(stream)), "m5_clock_stamp")


def use_on(t: Optional[torch.Tensor], stream: torch.cuda.Stream) -> Optional[torch.Tensor]:
    """`t` is about to be read by launches on `stream` although another stream's allocation may back it (a caller's temporary made on
    the current stream and dropped as soon as the call returns): tell torch's caching allocator, which otherwise hands the block to
    the next allocation of ITS stream while `stream` has not read it yet.  Round 5 found exactly that: the text ids of `tts()` -- a
    temporary of the default stream, gathered on the NAR stream -- were overwritten once the host no longer waited for the
    conditioning, and the gather read through garbage indices (a memory fault)."""
    if t is not None and t.is_cuda:
        t.record_stream(stream)
    return t


_SESSION_TLS = threading.local()      # per host thread: {(device index, role): stream}; the entries die with their thread


def session_stream(dev, role: str) -> torch.cuda.Stream:
    """The long-lived stream of one role ("ar", "nar", "nar_lane1", ...) on one device for the calling host thread.
    Sessions used to make a fresh ``torch.cuda.Stream()`` each: torch hands those out of a pool of 32, and its caching
    allocator keeps freed blocks PER STREAM, so the ~1.2 GB of per-utterance state (hoisted conditioning, workspaces, KV cache)
    was hipMalloc'ed again for every utterance until the pool wrapped around -- 50 ms of host time per utterance in front of the
    decode (tools/host_profile.py).  With one stream per role the blocks of the previous utterance are reused.  Roles keep the
    concurrency that exists (AR decode beside NAR conditioning; two batch groups in flight); host threads keep their own streams.
    Consequence to know about: two sessions of the SAME role made on one host thread without an explicit `stream=` share this
    stream and therefore run one after the other -- pass each its own ``torch.cuda.Stream`` (as ``tts_batch_from_codes`` does
    for its groups in flight) to overlap them.  The table is thread-local (no lock, no growth in thread-per-request servers:
    a dead thread's entries go with it); every stream that stays alive keeps its own pool of cached allocator blocks
    (``torch.cuda.empty_cache()`` returns them)."""
    d = torch.device(dev)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    tab = getattr(_SESSION_TLS, "streams", None)
    if tab is None:
        tab = _SESSION_TLS.streams = {}
    st = tab.get((idx, role))
    if st is None:
        st = tab[(idx, role)] = torch.cuda.Stream(device=idx)
    return st


class Graph:
    """A captured hipGraph of libmars5_hip launches (m5_graph_* helpers)."""

    def __init__(self):
        self.exec = C.c_void_p(None)

    @staticmethod
    def begin(stream: int) -> None:
        check(lib.m5_graph_begin(stream), "m5_graph_begin")

    def end(self, stream: int) -> "Graph":
        check(lib.m5_graph_end(stream, C.byref(self.exec)), "m5_graph_end")
        return self

    def launch(self, stream: int) -> None:
        check(lib.m5_graph_launch(self.exec, stream), "m5_graph_launch")

    def __del__(self):
        try:
            if self.exec and self.exec.value:
                lib.m5_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = C.c_void_p(None)
        check(lib.m5_event_create(C.byref(self.ev)), "m5_event_create")

    def record(self, stream: int) -> None:
        check(lib.m5_event_record(self.ev, stream), "m5_event_record")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib.m5_event_elapsed_ms(self.ev, stop.ev, C.byref(ms)), "m5_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.ev and self.ev.value:
                lib.m5_event_destroy(self.ev)
        except Exception:
            pass
