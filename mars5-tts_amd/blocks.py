"""Pre-LN transformer encoder / decoder layers of the reference (``nn.TransformerEncoderLayer``
/ ``nn.TransformerDecoderLayer`` with ``linear1 = Identity`` and ``activation = FNNSwiGLU``,
reference model.py:61-67,179-203) as launch sequences over libmars5_hip kernels.

Host code only sequences kernels and owns buffers; weight repacking (dtype cast, W/V row
interleave for the fused SwiGLU epilogue) happens once at load.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops

LAYERNORM_EPS = 4e-5        # reference model.py:13


def interleave_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """rows (a_0, b_0, a_1, b_1, ...): lets one GEMM produce silu(a_i x) * (b_i x) in its epilogue."""
    return torch.stack([a, b], dim=1).reshape(a.shape[0] * 2, a.shape[1]).contiguous()


@dataclass
class EncLayerW:
    in_w: torch.Tensor; in_b: torch.Tensor
    out_w: torch.Tensor; out_b: torch.Tensor
    act_w: torch.Tensor                      # (2*FF, D) interleaved (W_i, V_i)
    l2_w: torch.Tensor; l2_b: torch.Tensor
    n1_w: torch.Tensor; n1_b: torch.Tensor
    n2_w: torch.Tensor; n2_b: torch.Tensor
    # decoder only
    ca_q_w: Optional[torch.Tensor] = None; ca_q_b: Optional[torch.Tensor] = None
    ca_kv_w: Optional[torch.Tensor] = None; ca_kv_b: Optional[torch.Tensor] = None
    ca_out_w: Optional[torch.Tensor] = None; ca_out_b: Optional[torch.Tensor] = None
    ca_q_wT: Optional[torch.Tensor] = None
    n3_w: Optional[torch.Tensor] = None; n3_b: Optional[torch.Tensor] = None


def pack_layer(sd: Dict[str, torch.Tensor], p: str, dt: torch.dtype, dev, cross: bool = False) -> EncLayerW:
    def W(name):
        return sd[name].to(device=dev, dtype=dt).contiguous()

    def Fv(name):
        return sd[name].to(device=dev, dtype=torch.float32).contiguous()

    act = interleave_rows(sd[f"{p}.activation.W.weight"].float(), sd[f"{p}.activation.V.weight"].float())
    lw = EncLayerW(
        in_w=W(f"{p}.self_attn.in_proj_weight"), in_b=Fv(f"{p}.self_attn.in_proj_bias"),
        out_w=W(f"{p}.self_attn.out_proj.weight"), out_b=Fv(f"{p}.self_attn.out_proj.bias"),
        act_w=act.to(device=dev, dtype=dt).contiguous(),
        l2_w=W(f"{p}.linear2.weight"), l2_b=Fv(f"{p}.linear2.bias"),
        n1_w=Fv(f"{p}.norm1.weight"), n1_b=Fv(f"{p}.norm1.bias"),
        n2_w=Fv(f"{p}.norm2.weight"), n2_b=Fv(f"{p}.norm2.bias"))
    if cross:
        D = lw.out_w.shape[0]
        cw, cb = sd[f"{p}.multihead_attn.in_proj_weight"], sd[f"{p}.multihead_attn.in_proj_bias"]
        lw.ca_q_w = cw[:D].to(device=dev, dtype=dt).contiguous()
        lw.ca_q_b = cb[:D].to(device=dev, dtype=torch.float32).contiguous()
        lw.ca_kv_w = cw[D:].to(device=dev, dtype=dt).contiguous()
        lw.ca_kv_b = cb[D:].to(device=dev, dtype=torch.float32).contiguous()
        lw.ca_out_w = W(f"{p}.multihead_attn.out_proj.weight")
        # absorbed cross-attention (csrc/xattn_absorb.hip): Wq as [H][D][64] (Wq[h*64+d][n] at [h][n][d]), K and V weights apart
        if dt != torch.float32:
            H = D // 64
            lw.ca_q_wT = lw.ca_q_w.view(H, 64, D).permute(0, 2, 1).contiguous()
            lw.ca_k_w, lw.ca_v_w = lw.ca_kv_w[:D], lw.ca_kv_w[D:]
        lw.ca_out_b = Fv(f"{p}.multihead_attn.out_proj.bias")
        lw.n3_w, lw.n3_b = Fv(f"{p}.norm3.weight"), Fv(f"{p}.norm3.bias")
        if DeferredLN.eligible(D, dt):
            fold_layer_dln(sd, p, lw, act, dt, dev)
    return lw


def fold_layer_dln(sd: Dict[str, torch.Tensor], p: str, lw: EncLayerW, act: torch.Tensor, dt: torch.dtype, dev) -> None:
    """Weights of one decoder layer for the deferred-LayerNorm path (include/mars5_hip.h, M5DeferredLN): for each Linear behind
    a LayerNorm (norm1 -> in_proj, norm2 -> cross-attention query projection, norm3 -> SwiGLU pair)
    W' = dtype(W diag gamma), b' = b + W beta (fp32, from the fp32 master weights), s = row sums of W' AS ROUNDED (fp32)."""
    f = lambda n: sd[n].float()                                      # noqa: E731
    D = lw.out_w.shape[0]
    H = D // 64

    def fold(W, b, g, be):
        Wf = (W * g[None, :]).to(dt)
        bf = W @ be + (b if b is not None else 0.0)
        return Wf.to(dev).contiguous(), bf.to(dev, torch.float32).contiguous(), Wf.float().sum(dim=1).to(dev).contiguous()

    lw.in_w_f, lw.in_b_f, lw.in_s = fold(f(f"{p}.self_attn.in_proj_weight"), f(f"{p}.self_attn.in_proj_bias"), f(f"{p}.norm1.weight"), f(f"{p}.norm1.bias"))
    lw.act_w_f, lw.act_b_f, lw.act_s = fold(act, None, f(f"{p}.norm3.weight"), f(f"{p}.norm3.bias"))
    cw, cb = f(f"{p}.multihead_attn.in_proj_weight")[:D], f(f"{p}.multihead_attn.in_proj_bias")[:D]
    lw.ca_q_w_f, lw.ca_q_b_f, lw.ca_q_rs = fold(cw, cb, f(f"{p}.norm2.weight"), f(f"{p}.norm2.bias"))
    lw.ca_q_wT_f = lw.ca_q_w_f.view(H, 64, D).permute(0, 2, 1).contiguous()


class DeferredLN:
    """Buffers of a deferred-LayerNorm chain over the rows of a SeqWorkspace (csrc/gemm16.hip, DLN): per row and 128-column
    tile the partial sums {sum, sum of squares} of (x - centre) left by the residual GEMM that produced x, and the row
    centres.  The centred 16-bit copy of x lives in ws.xn (the LayerNorm output it replaces)."""

    def __init__(self, ws: "SeqWorkspace", dev):
        self.np = ws.D // 128
        self.D = ws.D
        self.part = torch.zeros(ws.M, self.np, 2, dtype=torch.float32, device=dev)
        # A producer centres its rows by cen[k] + delta (the previous producer's centre + the d the consumer in between
        # measured = the row's mean at that point) and leaves that centre in cen[1 - k]: two buffers, because the other column
        # tiles of the same launch are still reading cen[k].  `start` = the explicit LayerNorm that begins a chain wrote the
        # row means into cen[0]; the first producer after it takes them as they are.
        self.cen = torch.zeros(2, ws.M, dtype=torch.float32, device=dev)
        self.delta = torch.zeros(ws.M, dtype=torch.float32, device=dev)
        self.k = 0
        self.fresh = True

    def start(self) -> torch.Tensor:
        """The buffer the chain-starting LayerNorm (ops.layernorm_mean) writes the row means to."""
        self.k, self.fresh = 0, True
        return self.cen[0]

    @staticmethod
    def eligible(D: int, dt: torch.dtype) -> bool:
        return dt != torch.float32 and D % 256 == 0 and D // 128 <= 8

    def producer(self, ws: "SeqWorkspace", r0: int = 0, rows_bs: Optional[int] = None, advance: bool = True) -> L.DeferredLN:
        """The residual GEMM over rows [r0, ...) also writes the centred copy (ws.xn) and the partials.  `advance`: this is the
        last (or only) launch of the producing step -- a step split over several launches (one per run of utterances) passes
        False for all but the last, so that every launch of the step reads and writes the same pair of buffers."""
        es = ws.xn.element_size()
        k, fresh = self.k, self.fresh
        if advance:
            self.k, self.fresh = 1 - k, False
        return L.DeferredLN(mode=1, np=self.np, xt=ws.xn.data_ptr() + r0 * ws.D * es, ld_xt=ws.D, part=self.part.data_ptr() + r0 * self.np * 8,
                            cen_in=self.cen[k].data_ptr() + r0 * 4, cen_out=self.cen[1 - k].data_ptr() + r0 * 4,
                            delta=None if fresh else self.delta.data_ptr() + r0 * 4, s=None, s_bs=0, eps=0.0, n_feat=self.D,
                            rows_bs=rows_bs if rows_bs is not None else ws.M)

    def consumer(self, s: torch.Tensor, r0: int = 0, rows_bs: Optional[int] = None, s_bs: int = 0, M: Optional[int] = None) -> L.DeferredLN:
        """A GEMM whose A operand is the centred copy applies LayerNorm in its epilogue (s: row sums of the folded weights) and
        moves the row centres to the rows' means."""
        assert s.dtype == torch.float32 and s.is_contiguous()
        return L.DeferredLN(mode=2, np=self.np, xt=None, ld_xt=0, part=self.part.data_ptr() + r0 * self.np * 8, cen_in=None, cen_out=None,
                            delta=self.delta.data_ptr() + r0 * 4, s=s.data_ptr(), s_bs=s_bs, eps=LAYERNORM_EPS, n_feat=self.D,
                            rows_bs=rows_bs if rows_bs is not None else (M or 1 << 30))


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class SeqWorkspace:
    """Buffers for running encoder/decoder layers over B sequences of S rows (dim D, H heads
    of 64).  Each sequence occupies Sr = round_up(S, row_pad) rows of the row-major [B*Sr, D]
    activation buffers: with row_pad = 64 every sequence starts on a tile boundary, so the GEMM
    epilogues and the V^T scatter (4 consecutive s per 8-byte store) stay vector-aligned for
    B > 1.  The pad rows are ordinary FINITE rows (zero-initialised in a workspace that owns its buffers; a workspace laid
    `inside` another one inherits whatever finite values the larger one last wrote there), never attended to: keys >= S are
    masked by index; V^T rows are padded to whole 64-key tiles, and pad key columns of the last tile are multiplied by
    p = 0 -- which is why every buffer must stay free of Inf / NaN (16-bit engines: an overflow anywhere would leak into
    real rows of a later sub-problem).  No buffer of a workspace is live across a launch sequence that runs on a workspace
    laid inside it (tests/test_host_cpu.py::test_workspace_laid_over_a_larger_one_shares_its_front)."""

    def __init__(self, B: int, S: int, D: int, FF: int, dt: torch.dtype, dev, row_pad: int = 1, fuse_ln: bool = False,
                 inside: Optional["SeqWorkspace"] = None):
        """`inside`: lay this workspace over the front of a larger one's buffers instead of allocating (a sub-problem that
        runs strictly between two uses of the larger workspace -- the one-branch layer-0 block, the generated-rows-only last
        layer: the bytes it touches are then the ones the surrounding launches keep warm in L2 / the Infinity Cache)."""
        self.B, self.S, self.D, self.FF, self.dt = B, S, D, FF, dt
        self.H = D // 64
        self.Sr = round_up(S, row_pad)
        self.Sp = round_up(self.Sr, 64)
        M = B * self.Sr
        self.M = M
        if inside is not None:
            assert inside.dt == dt and inside.D == D and inside.FF == FF and M <= inside.M and B * self.Sp <= inside.B * inside.Sp
            front = lambda t, *shape: t.view(-1)[: int(torch.Size(shape).numel())].view(*shape)       # noqa: E731
        else:
            front = lambda t, *shape: torch.zeros(*shape, dtype=dt, device=dev)                       # noqa: E731
        src = inside
        self.xn = front(src.xn if src else None, M, D)
        self.q = front(src.q if src else None, B, self.H, self.Sr, 64)
        self.k = front(src.k if src else None, B, self.H, self.Sr, 64)
        self.vt = front(src.vt if src else None, B, self.H, 64, self.Sp)
        self.att = front(src.att if src else None, M, D)
        self.hff = front(src.hff if src else None, M, FF)
        # scratch of the fused residual + LayerNorm GEMM (partials + counters, zero between launches); None = never fuse
        self.ln_scratch = None
        if fuse_ln and dt != torch.float32:
            tiles_m = (M + 95) // 96
            self.ln_scratch = torch.zeros(256 + 768 * tiles_m * 16, dtype=torch.uint8, device=dev)
        self.ln_tag = 0                    # launch tag of the next fused call (consecutive calls on ln_scratch must differ)
        self.ln_tag_step = None            # device int32 that changes between graph replays (the DDPM step counter)

    def scatter(self, q=True, k=True, v=True) -> L.QkvScatter:
        H, S, Sp = self.H, self.Sr, self.Sp
        return L.QkvScatter(q=self.q.data_ptr() if q else None, k=self.k.data_ptr() if k else None,
                            vt=self.vt.data_ptr() if v else None, rows_per_batch=S, n_heads=H, head_dim=64,
                            q_bs=H * S * 64, q_hs=S * 64, q_rs=64, k_bs=H * S * 64, k_hs=S * 64, k_rs=64,
                            vt_bs=H * 64 * Sp, vt_hs=64 * Sp, vt_ds=Sp)

    def self_attn_args(self, key_len: Optional[torch.Tensor], causal: bool = False) -> L.AttnArgs:
        H, S, Sr, Sp, D = self.H, self.S, self.Sr, self.Sp, self.D
        return L.AttnArgs(q=self.q.data_ptr(), q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64,
                          k=self.k.data_ptr(), k_bs=H * Sr * 64, k_hs=Sr * 64, k_rs=64,
                          vt=self.vt.data_ptr(), vt_bs=H * 64 * Sp, vt_hs=64 * Sp, vt_ds=Sp,
                          o=self.att.data_ptr(), o_bs=Sr * D, o_rs=D, B=self.B, H=H, Sq=S, Sk=S,
                          key_len=key_len.data_ptr() if key_len is not None else None, causal=1 if causal else 0,
                          scale=64 ** -0.5, kv_index=None, kv_index_stride_k=0, kv_index_stride_v=0,
                          q_len=key_len.data_ptr() if key_len is not None else None)     # self-attention: a sequence's queries = its keys


def residual_gemm(a: torch.Tensor, w: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor], ws: SeqWorkspace,
                  next_ln, stream=None) -> bool:
    """x += a @ w^T + bias.  With `next_ln` = (gamma, beta) of the LayerNorm that consumes x next, the fused kernel
    also leaves ws.xn = LayerNorm(x) when the shape is eligible; returns whether ws.xn was produced."""
    if next_ln is not None and ws.ln_scratch is not None:
        if ops.gemm_residual_ln(a, w, x, bias, next_ln[0], next_ln[1], LAYERNORM_EPS, ws.xn, ws.ln_scratch, tag=ws.ln_tag,
                                tag_step=ws.ln_tag_step, stream=stream):
            ws.ln_tag = (ws.ln_tag + 1) % 64
            return True
    ops.gemm(a, w, x, L.EPI_RESIDUAL, bias=bias, stream=stream)
    return False


def self_attn_block(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, key_len: Optional[torch.Tensor], stream=None,
                    normed: bool = False, next_ln=None) -> bool:
    """x = x + out_proj(SDPA(in_proj(LN1(x))))   (x fp32 [B*S, D], updated in place).
    normed: ws.xn already holds LN1(x) (left there by the previous block's fused epilogue).  Returns whether ws.xn
    holds next_ln(x) on exit."""
    if not normed:
        ops.layernorm(x, lw.n1_w, lw.n1_b, LAYERNORM_EPS, ws.xn, stream=stream)
    ops.gemm(ws.xn, lw.in_w, None, L.EPI_QKV, bias=lw.in_b, scatter=ws.scatter(), stream=stream)
    ops.attention(ws.dt, ws.self_attn_args(key_len), stream=stream)
    return residual_gemm(ws.att, lw.out_w, x, lw.out_b, ws, next_ln, stream)


def ff_block(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, norm_w: torch.Tensor, norm_b: torch.Tensor, stream=None,
             normed: bool = False, next_ln=None) -> bool:
    """x = x + linear2(silu(W LN(x)) * (V LN(x)))."""
    if not normed:
        ops.layernorm(x, norm_w, norm_b, LAYERNORM_EPS, ws.xn, stream=stream)
    ops.gemm(ws.xn, lw.act_w, ws.hff, L.EPI_SWIGLU, stream=stream)
    return residual_gemm(ws.hff, lw.l2_w, x, lw.l2_b, ws, next_ln, stream)


def encoder_layer(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, key_len: Optional[torch.Tensor], stream=None) -> None:
    self_attn_block(x, lw, ws, key_len, stream)
    ff_block(x, lw, ws, lw.n2_w, lw.n2_b, stream)


@dataclass
class CrossMemory:
    """Pre-projected cross-attention memory of one decoder layer for every reverse step:
    k (T*Bm, H, Le, 64), vt (T*Bm, H, 64, Lep)."""
    k: torch.Tensor
    vt: torch.Tensor
    Le: int
    Lep: int
    Bm: int          # memory batch entries per step (2 = cond, uncond)
    v_rows: Optional[torch.Tensor] = None     # V as rows (T*Bm, H, Le, 64), for the absorbed path (16-bit engines)


def cross_memory_table(mems, dev) -> tuple:
    """(table, max_le) for ops.gemm_q_cross_attn: one row {K base, V^T base, Le, Lep, K step stride, V^T step stride} per
    sequence of the workspace, in workspace order (`mems`: one CrossMemory or a list, each covering Bm sequences)."""
    if isinstance(mems, CrossMemory):
        mems = [mems]
    rows = []
    for mem in mems:
        H = mem.k.shape[1]
        esz = mem.k.element_size()
        for b in range(mem.Bm):
            rows.append([mem.k.data_ptr() + b * H * mem.Le * 64 * esz, mem.vt.data_ptr() + b * H * 64 * mem.Lep * esz, mem.Le, mem.Lep,
                         mem.Bm * H * mem.Le * 64, mem.Bm * H * 64 * mem.Lep])
    return torch.tensor(rows, dtype=torch.int64, device="cpu").to(dev), max(m.Le for m in mems)


class AbsorbedCross:
    """Operands of the absorbed cross-attention (csrc/xattn_absorb.hip) for all decoder layers and a CONTIGUOUS run of
    workspace sequences [s0, s0 + n_seq) whose memories share the padded length Lp (48 if Le <= 48, else 64): built for the
    current reverse step by ONE launch (`build`), then each layer's block is a scores GEMM with per-head softmax and a P.B
    GEMM with the residual epilogue, both batched over the sequences.  Lp is a function of the utterance's own Le, so an
    utterance is computed with the same arithmetic alone and inside any batch."""

    def __init__(self, layers, mems_per_layer, s0: int, D: int, dt: torch.dtype, dev, dln: bool = False):
        H = D // 64
        self.H, self.D, self.dt, self.s0 = H, D, dt, s0
        self.n_seq = sum(mem.Bm for mem in mems_per_layer[0])
        self.n_layers = len(layers)
        self.Lp = self.lp_of(mems_per_layer[0][0].Le, H)
        assert self.Lp and all(self.lp_of(m.Le, H) == self.Lp for m in mems_per_layer[0])
        N = H * self.Lp
        self.A = torch.zeros(self.n_layers, self.n_seq, N, D, dtype=dt, device=dev)
        self.c = torch.zeros(self.n_layers, self.n_seq, N, dtype=torch.float32, device=dev)
        self.Bt = torch.zeros(self.n_layers, self.n_seq, D, N, dtype=dt, device=dev)
        # deferred LayerNorm (`dln`): layers >= 1 take the gamma-folded query weights and also get s = row sums of A
        self.dln = dln and all(hasattr(lw, "ca_q_wT_f") for lw in layers)
        self.sA = torch.zeros(self.n_layers, self.n_seq, N, dtype=torch.float32, device=dev) if self.dln else None
        rows, lrows = [], []
        es = self.A.element_size()
        for l, (lw, mems) in enumerate(zip(layers, mems_per_layer)):
            s = 0
            for mem in mems:
                for b in range(mem.Bm):
                    folded = self.dln and l >= 1
                    rows.append([mem.k.data_ptr() + b * H * mem.Le * 64 * es, mem.v_rows.data_ptr() + b * H * mem.Le * 64 * es, mem.Le,
                                 mem.Bm * H * mem.Le * 64, self.A[l, s].data_ptr(), self.c[l, s].data_ptr(), self.Bt[l, s].data_ptr(),
                                 self.sA[l, s].data_ptr() if folded else 0])
                    s += 1
            if self.dln and l >= 1:
                lrows.append([lw.ca_q_wT_f.data_ptr(), lw.ca_out_w.data_ptr(), lw.ca_q_b_f.data_ptr(), lw.ca_q_rs.data_ptr()])
            else:
                lrows.append([lw.ca_q_wT.data_ptr(), lw.ca_out_w.data_ptr(), lw.ca_q_b.data_ptr() if lw.ca_q_b is not None else 0, 0])
        self.tab_seq = torch.tensor(rows, dtype=torch.int64, device="cpu").to(dev)
        self.tab_layer = torch.tensor(lrows, dtype=torch.int64, device="cpu").to(dev)

    @staticmethod
    def lp_of(Le: int, H: int) -> int:
        """padded memory length of the absorbed path for a model with H heads; 0 = not eligible (longer memories keep the
        unfused path: beyond 64 keys the absorbed operands are wider than the projections they replace).  H * Lp is the K
        of the P.B GEMM (whole 64-deep K-steps) and two heads share a workgroup tile of the scores GEMM."""
        if H % 2:
            return 0
        for lp in (48, 64):
            if Le <= lp and (H * lp) % 64 == 0:
                return lp
        return 0

    def build(self, step_ptr: torch.Tensor, stream=None) -> None:
        ops.xattn_absorb(self.dt, self.tab_seq, self.tab_layer, self.n_layers, self.n_seq, self.H, self.D, self.Lp, step_ptr, 64 ** -0.5,
                         stream=stream)

    def block(self, l: int, x: torch.Tensor, lw: EncLayerW, ws: "SeqWorkspace", stream=None) -> None:
        """x[rows of my sequences] += cross-attention of ws.xn (= LN2(x)) against layer l's memory: two launches."""
        N = self.H * self.Lp
        r0, rows = self.s0 * ws.Sr, self.n_seq * ws.Sr
        assert N <= ws.FF
        P = ws.hff.view(-1)[r0 * N: (r0 + rows) * N].view(rows, N)        # the feed-forward scratch is free here (N <= FF)
        ops.xattn_scores(ws.xn[r0:], ws.Sr * self.D, self.A[l], self.c[l], P, ws.Sr * N, ws.Sr, self.H, self.Lp, self.n_seq, stream=stream)
        ops.gemm(P, self.Bt[l, 0], x[r0:], L.EPI_RESIDUAL, bias=lw.ca_out_b, M=ws.Sr, batch=self.n_seq, sA=ws.Sr * N, sW=self.D * N,
                 sC=ws.Sr * self.D, sBias=0, stream=stream)


    def block_dln(self, l: int, x: torch.Tensor, lw: EncLayerW, ws: "SeqWorkspace", dl: DeferredLN, consume: bool, stream=None,
                  last_segment: bool = True, rt: Optional["RowTiles"] = None) -> None:
        """``block`` inside a deferred-LayerNorm chain: `consume` = ws.xn holds the centred copy of x (the scores GEMM applies
        norm2 in its epilogue; layers >= 1), else ws.xn = LN2(x); the P.B GEMM always leaves the centred copy + partials of
        the updated rows for norm3."""
        N = self.H * self.Lp
        r0, rows = self.s0 * ws.Sr, self.n_seq * ws.Sr
        assert N <= ws.FF and self.dln
        P = ws.hff.view(-1)[r0 * N: (r0 + rows) * N].view(rows, N)
        rc = _rt_sub(rt, self.s0, self.n_seq)      # the row tiles of THIS run of sequences (numbered from the run's first row)
        if consume:
            assert l >= 1
            ops.xattn_scores_dln(ws.xn[r0:], ws.Sr * self.D, self.A[l], self.c[l], P, ws.Sr * N, ws.Sr, self.H, self.Lp, self.n_seq,
                                 dl.consumer(self.sA[l], r0=r0, rows_bs=ws.Sr, s_bs=N), stream=stream, rt=rc)
        elif rc is not None:
            ops.xattn_scores_dln(ws.xn[r0:], ws.Sr * self.D, self.A[l], self.c[l], P, ws.Sr * N, ws.Sr, self.H, self.Lp, self.n_seq, None, stream=stream, rt=rc)
        else:
            ops.xattn_scores(ws.xn[r0:], ws.Sr * self.D, self.A[l], self.c[l], P, ws.Sr * N, ws.Sr, self.H, self.Lp, self.n_seq, stream=stream)
        ops.gemm_dln(P, self.Bt[l, 0], x[r0:], L.EPI_RESIDUAL, dl.producer(ws, r0=r0, rows_bs=ws.Sr, advance=last_segment), bias=lw.ca_out_b, M=ws.Sr, batch=self.n_seq,
                     sA=ws.Sr * N, sW=self.D * N, sC=ws.Sr * self.D, sBias=0, stream=stream, rt=rc)


def make_cross_plan(layers, mems_per_layer, D: int, dt: torch.dtype, dev, dln: bool = False) -> list:
    """Split the workspace's sequences (in order; mems_per_layer[l] = one CrossMemory per utterance) into maximal runs of
    utterances that take the same cross-attention path: ("absorbed", AbsorbedCross) for 16-bit engines and memories of
    <= 64 rows (one run per padded length), ("plain", s0, s1, utterance indices) otherwise."""
    plan, s = [], 0
    mems0 = mems_per_layer[0]
    i = 0
    while i < len(mems0):
        H = D // 64
        cls_of = lambda m: AbsorbedCross.lp_of(m.Le, H) if (dt != torch.float32 and m.v_rows is not None) else 0     # noqa: E731
        cls = cls_of(mems0[i])
        j = i
        while j < len(mems0) and cls_of(mems0[j]) == cls:
            j += 1
        n = sum(m.Bm for m in mems0[i:j])
        if cls:
            plan.append(("absorbed", AbsorbedCross(layers, [ml[i:j] for ml in mems_per_layer], s, D, dt, dev, dln=dln)))
        else:
            plan.append(("plain", s, s + n, list(range(i, j))))
        s += n
        i = j
    return plan


def _plain_cross(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, mems, s0: int, s1: int, step_ptr: torch.Tensor, stream=None) -> None:
    """q-projection, attention per utterance against its pre-projected memory, out-projection (+residual) for the
    workspace sequences [s0, s1) -- the reference's operation order."""
    H, S, Sr, D = ws.H, ws.S, ws.Sr, ws.D
    esz = ws.q.element_size()
    r0, rows = s0 * Sr, (s1 - s0) * Sr
    sc = ws.scatter(True, False, False)
    sc.q = ws.q.data_ptr() + s0 * H * Sr * 64 * esz
    ops.gemm(ws.xn[r0:r0 + rows], lw.ca_q_w, None, L.EPI_QKV, bias=lw.ca_q_b, scatter=sc, stream=stream)
    b0 = s0
    for mem in mems:
        a = L.AttnArgs(q=ws.q.data_ptr() + b0 * H * Sr * 64 * esz, q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64,
                       k=mem.k.data_ptr(), k_bs=H * mem.Le * 64, k_hs=mem.Le * 64, k_rs=64,
                       vt=mem.vt.data_ptr(), vt_bs=H * 64 * mem.Lep, vt_hs=64 * mem.Lep, vt_ds=mem.Lep,
                       o=ws.att.data_ptr() + b0 * Sr * D * esz, o_bs=Sr * D, o_rs=D, B=mem.Bm, H=H, Sq=S, Sk=mem.Le, key_len=None,
                       causal=0, scale=64 ** -0.5, kv_index=step_ptr.data_ptr(),
                       kv_index_stride_k=mem.Bm * H * mem.Le * 64, kv_index_stride_v=mem.Bm * H * 64 * mem.Lep)
        ops.attention(ws.dt, a, stream=stream)
        b0 += mem.Bm
    assert b0 == s1
    ops.gemm(ws.att[r0:r0 + rows], lw.ca_out_w, x[r0:r0 + rows], L.EPI_RESIDUAL, bias=lw.ca_out_b, stream=stream)


def cross_attn_block(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, mems, step_ptr: torch.Tensor, stream=None,
                     normed: bool = False, next_ln=None, xa=None, plan=None, layer: int = 0) -> bool:
    """x = x + out_proj(SDPA(q_proj(LN2(x)), memory K/V of step *step_ptr)).
    `mems`: one CrossMemory covering all ws.B sequences, or a list of them (one per utterance, each covering its Bm
    consecutive sequences of the workspace).  `plan` (``make_cross_plan``): per run of utterances either the absorbed
    form (two batched GEMMs; operands built at the top of the step) or the reference's operation order."""
    H, S, Sr, D = ws.H, ws.S, ws.Sr, ws.D
    if not normed:
        ops.layernorm(x, lw.n2_w, lw.n2_b, LAYERNORM_EPS, ws.xn, stream=stream)
    if isinstance(mems, CrossMemory):
        mems = [mems]
    if plan is not None:
        for seg in plan:
            if seg[0] == "absorbed":
                seg[1].block(layer, x, lw, ws, stream)
            else:
                _plain_cross(x, lw, ws, [mems[u] for u in seg[3]], seg[1], seg[2], step_ptr, stream)
        return False
    # xa = (memory table, longest memory) built before any graph capture: query projection + attention in one launch
    if xa is not None and ws.dt != torch.float32 and \
            ops.gemm_q_cross_attn(ws.xn, lw.ca_q_w, lw.ca_q_b, H, xa[0], xa[1], Sr, step_ptr, 64 ** -0.5, ws.att, stream=stream):
        return residual_gemm(ws.att, lw.ca_out_w, x, lw.ca_out_b, ws, next_ln, stream)
    ops.gemm(ws.xn, lw.ca_q_w, None, L.EPI_QKV, bias=lw.ca_q_b, scatter=ws.scatter(True, False, False), stream=stream)
    b0 = 0
    esz = ws.q.element_size()
    for mem in mems:
        a = L.AttnArgs(q=ws.q.data_ptr() + b0 * H * Sr * 64 * esz, q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64,
                       k=mem.k.data_ptr(), k_bs=H * mem.Le * 64, k_hs=mem.Le * 64, k_rs=64,
                       vt=mem.vt.data_ptr(), vt_bs=H * 64 * mem.Lep, vt_hs=64 * mem.Lep, vt_ds=mem.Lep,
                       o=ws.att.data_ptr() + b0 * Sr * D * esz, o_bs=Sr * D, o_rs=D, B=mem.Bm, H=H, Sq=S, Sk=mem.Le, key_len=None,
                       causal=0, scale=64 ** -0.5, kv_index=step_ptr.data_ptr(),
                       kv_index_stride_k=mem.Bm * H * mem.Le * 64, kv_index_stride_v=mem.Bm * H * 64 * mem.Lep)
        ops.attention(ws.dt, a, stream=stream)
        b0 += mem.Bm
    assert b0 == ws.B
    return residual_gemm(ws.att, lw.ca_out_w, x, lw.ca_out_b, ws, next_ln, stream)


def decoder_layer(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, mems, step_ptr: torch.Tensor, stream=None,
                  key_len: Optional[torch.Tensor] = None, normed: bool = False, next_ln=None, xa=None, plan=None, layer: int = 0) -> bool:
    """One pre-LN decoder layer.  Each residual GEMM tries to leave the NEXT LayerNorm's output in ws.xn (fused
    epilogue); `normed` says the caller (previous layer) already did that for norm1, the return value says whether
    `next_ln` (the following layer's norm1) has been applied on exit."""
    n = self_attn_block(x, lw, ws, key_len, stream, normed=normed, next_ln=(lw.n2_w, lw.n2_b))
    n = cross_attn_block(x, lw, ws, mems, step_ptr, stream, normed=n, next_ln=(lw.n3_w, lw.n3_b), xa=xa, plan=plan, layer=layer)
    return ff_block(x, lw, ws, lw.n3_w, lw.n3_b, stream, normed=n, next_ln=next_ln)


class RowTiles:
    """Row-tile lists of a padded batch layout (include/mars5_hip.h, M5RowTiles): sequence b occupies rows b * Sr .. of the
    workspace, its first lens[b] rows are real; for the tile heights 96 / 128 / 192 the tiles that hold a real row.  GEMMs
    launched with `.c` run only those, so padding a group of utterances to its longest member costs (almost) nothing."""

    ALIGN = 384                                   # lcm(96, 128, 192): every sequence starts on a tile boundary of every tile height

    def __init__(self, lens, Sr: int, dev):
        assert Sr % self.ALIGN == 0 and all(0 < n <= Sr for n in lens)
        self.maps, ns = [], []
        for bm in (96, 128, 192):
            tpb = Sr // bm
            e = [b * tpb + t for b, n in enumerate(lens) for t in range((n + bm - 1) // bm)]
            self.maps.append(torch.tensor(e, dtype=torch.int32, device="cpu").to(dev))
            ns.append(len(e))
        self.n, self.Sr, self.lens = ns, Sr, list(lens)
        self._subs = {}
        # the sequences' own lengths: the three lists cover different pad rows (ceil(len / BM) * BM), so the deferred-LayerNorm
        # consumers give the pad rows of their tiles d = r = 0 (include/mars5_hip.h, M5RowTiles.seq_len)
        self.len_dev = torch.tensor(self.lens, dtype=torch.int32, device="cpu").to(dev)
        self.c = L.RowTiles(map=(L.vp * 3)(*[m.data_ptr() for m in self.maps]), n=(L.i32 * 3)(*ns), rows_per_seq=Sr,
                            seq_len=self.len_dev.data_ptr())

    def prebuild(self, runs) -> None:
        """Make the sub-lists of the runs [(s0, n), ...] of sequences now (a small host -> device copy each), so that the launch
        sequence itself -- which may be under stream capture -- allocates and copies nothing."""
        for s0, n in runs:
            _rt_sub(self, s0, n)


def _rt_sub(rt: Optional[RowTiles], s0: int, n: int):
    """The ctypes lists of the sequences [s0, s0 + n) of `rt`, numbered from s0 (a launch over that run of the workspace)."""
    if rt is None:
        return None
    if s0 == 0 and n == len(rt.lens):
        return rt.c
    key = (s0, n)
    if key not in rt._subs:
        rt._subs[key] = RowTiles(rt.lens[s0: s0 + n], rt.Sr, rt.maps[0].device)
    return rt._subs[key].c


def plan_allows_dln(plan) -> bool:
    """Every run of utterances can take part in a deferred-LayerNorm chain: absorbed runs built with the folded operands,
    plain runs (memories of more than 64 rows) through the folded query projection (blocks._plain_cross_dln)."""
    return plan is not None and len(plan) > 0 and all((seg[0] == "absorbed" and seg[1].dln) or seg[0] == "plain" for seg in plan)


def _plain_cross_dln(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, mems, s0: int, s1: int, step_ptr: torch.Tensor, dl: DeferredLN,
                     consume: bool, last_segment: bool, rt: Optional[RowTiles], stream=None) -> None:
    """``_plain_cross`` inside a deferred-LayerNorm chain: the query projection is the consumer of norm2 (gamma-folded weights;
    layer 0: ws.xn = LN2(x) and the plain weights), the out-projection the producer for norm3."""
    H, S, Sr, D = ws.H, ws.S, ws.Sr, ws.D
    esz = ws.q.element_size()
    r0, rows = s0 * Sr, (s1 - s0) * Sr
    rc = _rt_sub(rt, s0, s1 - s0)
    sc = ws.scatter(True, False, False)
    sc.q = ws.q.data_ptr() + s0 * H * Sr * 64 * esz
    if consume:
        ops.gemm_dln(ws.xn[r0:r0 + rows], lw.ca_q_w_f, None, L.EPI_QKV, dl.consumer(lw.ca_q_rs, r0=r0, M=rows), bias=lw.ca_q_b_f, scatter=sc,
                     stream=stream, rt=rc)
    elif rc is not None:
        ops.gemm_dln(ws.xn[r0:r0 + rows], lw.ca_q_w, None, L.EPI_QKV, None, bias=lw.ca_q_b, scatter=sc, stream=stream, rt=rc)
    else:
        ops.gemm(ws.xn[r0:r0 + rows], lw.ca_q_w, None, L.EPI_QKV, bias=lw.ca_q_b, scatter=sc, stream=stream)
    b0 = s0
    for mem in mems:
        a = L.AttnArgs(q=ws.q.data_ptr() + b0 * H * Sr * 64 * esz, q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64,
                       k=mem.k.data_ptr(), k_bs=H * mem.Le * 64, k_hs=mem.Le * 64, k_rs=64,
                       vt=mem.vt.data_ptr(), vt_bs=H * 64 * mem.Lep, vt_hs=64 * mem.Lep, vt_ds=mem.Lep,
                       o=ws.att.data_ptr() + b0 * Sr * D * esz, o_bs=Sr * D, o_rs=D, B=mem.Bm, H=H, Sq=(rt.lens[b0] if rt is not None else S),
                       Sk=mem.Le, key_len=None, causal=0, scale=64 ** -0.5, kv_index=step_ptr.data_ptr(),
                       kv_index_stride_k=mem.Bm * H * mem.Le * 64, kv_index_stride_v=mem.Bm * H * 64 * mem.Lep)
        ops.attention(ws.dt, a, stream=stream)
        b0 += mem.Bm
    assert b0 == s1
    ops.gemm_dln(ws.att[r0:r0 + rows], lw.ca_out_w, x[r0:r0 + rows], L.EPI_RESIDUAL, dl.producer(ws, r0=r0, rows_bs=rows, advance=last_segment),
                 bias=lw.ca_out_b, stream=stream, rt=rc)


def decoder_layer_dln(x: torch.Tensor, lw: EncLayerW, ws: SeqWorkspace, step_ptr: torch.Tensor, dl: DeferredLN, plan, layer: int,
                      chain_in: bool, chain_out: bool, stream=None, key_len: Optional[torch.Tensor] = None,
                      skip_self: bool = False, rt: Optional[RowTiles] = None, mems=None) -> None:
    """One pre-LN decoder layer with its LayerNorms DEFERRED into the GEMMs that consume them (include/mars5_hip.h,
    M5DeferredLN; reference model.py:179-203, same mathematics): every residual GEMM leaves a centred 16-bit copy of the rows
    it updated (ws.xn) + per-tile row partials, the next projection applies the normalisation in its epilogue -- 3 LayerNorm
    launches per layer less.  `chain_in`: ws.xn / dl already describe x (left by the previous layer's linear2), else norm1
    is an explicit launch.  Layer 0 keeps explicit norm1 / norm2 launches (its self-attention block runs once for both guidance
    branches; norm2 starts the chain and leaves the row means as the first centres), so the per-row arithmetic is the same
    for a lone utterance and inside any batch.  `chain_out`: linear2 leaves the copy for the next layer's norm1.
    `skip_self`: the caller already ran the self-attention block (layer 0 of a guided lone utterance).
    `rt` (a batch of utterances of different lengths): every GEMM runs over the row tiles that hold real rows only."""
    first = layer == 0
    assert first or chain_in
    rc = rt.c if rt is not None else None
    if not skip_self:
        if chain_in and not first:
            ops.gemm_dln(ws.xn, lw.in_w_f, None, L.EPI_QKV, dl.consumer(lw.in_s, M=ws.M), bias=lw.in_b_f, scatter=ws.scatter(), stream=stream, rt=rc)
        else:
            ops.layernorm(x, lw.n1_w, lw.n1_b, LAYERNORM_EPS, ws.xn, stream=stream)
            if rc is None:
                ops.gemm(ws.xn, lw.in_w, None, L.EPI_QKV, bias=lw.in_b, scatter=ws.scatter(), stream=stream)
            else:
                ops.gemm_dln(ws.xn, lw.in_w, None, L.EPI_QKV, None, bias=lw.in_b, scatter=ws.scatter(), stream=stream, rt=rc)
        ops.attention(ws.dt, ws.self_attn_args(key_len), stream=stream)
        if first:
            if rc is None:
                ops.gemm(ws.att, lw.out_w, x, L.EPI_RESIDUAL, bias=lw.out_b, stream=stream)
            else:
                ops.gemm_dln(ws.att, lw.out_w, x, L.EPI_RESIDUAL, None, bias=lw.out_b, stream=stream, rt=rc)
        else:
            ops.gemm_dln(ws.att, lw.out_w, x, L.EPI_RESIDUAL, dl.producer(ws), bias=lw.out_b, stream=stream, rt=rc)
    if first:
        ops.layernorm_mean(x, lw.n2_w, lw.n2_b, LAYERNORM_EPS, ws.xn, dl.start(), stream=stream)
    for i, seg in enumerate(plan):
        if seg[0] == "absorbed":
            seg[1].block_dln(layer, x, lw, ws, dl, consume=not first, stream=stream, last_segment=i + 1 == len(plan), rt=rt)
        else:
            _plain_cross_dln(x, lw, ws, [mems[u] for u in seg[3]], seg[1], seg[2], step_ptr, dl, consume=not first,
                             last_segment=i + 1 == len(plan), rt=rt, stream=stream)
    ops.gemm_dln(ws.xn, lw.act_w_f, ws.hff, L.EPI_SWIGLU, dl.consumer(lw.act_s, M=ws.M), bias=lw.act_b_f, stream=stream, rt=rc)
    if chain_out:
        ops.gemm_dln(ws.hff, lw.l2_w, x, L.EPI_RESIDUAL, dl.producer(ws), bias=lw.l2_b, stream=stream, rt=rc)
    elif rc is None:
        ops.gemm(ws.hff, lw.l2_w, x, L.EPI_RESIDUAL, bias=lw.l2_b, stream=stream)
    else:
        ops.gemm_dln(ws.hff, lw.l2_w, x, L.EPI_RESIDUAL, None, bias=lw.l2_b, stream=stream, rt=rc)


class SpeakerEncoder:
    """The reference's speaker-reference encoder (CodecLM: model.py:109-127, ResidualTransformer:
    model.py:298-310): [identity row, chunked-embedding(codes)] + sine positions -> pre-LN
    encoder layers -> final LayerNorm -> row 0."""

    def __init__(self, sd: Dict[str, torch.Tensor], emb_prefix: str, alpha_name: str, n_layers: int, dt: torch.dtype, dev,
                 max_frames: int = 4000):
        self.dt, self.dev = dt, dev
        self.layers = [pack_layer(sd, f"spk_encoder.layers.{l}", dt, dev) for l in range(n_layers)]
        self.norm_w = sd["spk_encoder.norm.weight"].to(dev, torch.float32).contiguous()
        self.norm_b = sd["spk_encoder.norm.bias"].to(dev, torch.float32).contiguous()
        self.tables = torch.stack([sd[f"{emb_prefix}.embs.{q}.weight"].float() for q in range(8)]).to(dev).contiguous()
        self.identity = sd["spk_identity_emb.weight"].to(dev, torch.float32).contiguous()
        self.alpha = sd[alpha_name].to(dev, torch.float32).contiguous()
        self.D = self.identity.shape[1]
        self.FF = self.layers[0].l2_w.shape[1]
        from .tables import sine_pe
        self.pe = sine_pe(max_frames + 1, self.D).to(dev)

    def __call__(self, codes: Optional[torch.Tensor], stream=None) -> torch.Tensor:
        """codes (Lc, 8) int64 device, or None for the unconditional vector (all-pad codes,
        length 0: only position 0 is attended, so the sequence collapses to the identity
        row -- a per-model constant).  Returns (D,) fp32."""
        R = 1 if codes is None else 1 + codes.shape[0]
        assert R <= self.pe.shape[0]
        x = torch.empty(1, R, self.D, dtype=torch.float32, device=self.dev)
        ops.chunked_embed(x, self.tables, codes, self.identity, self.alpha, self.pe, stream=stream)
        x = x[0]
        ws = SeqWorkspace(1, R, self.D, self.FF, self.dt, self.dev)
        for lw in self.layers:
            encoder_layer(x, lw, ws, None, stream)
        out = torch.empty(1, self.D, dtype=torch.float32, device=self.dev)
        ops.layernorm(x, self.norm_w, self.norm_b, LAYERNORM_EPS, out, M=1, stream=stream)
        return out[0]
