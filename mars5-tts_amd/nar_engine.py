"""NAR stage on MI355X: multinomial-DDPM refinement of the 7 remaining Encodec codebooks.

Replaces the device work of reference ``mars5/diffuser.py:345-472`` +
``mars5/model.py:264-343`` (``ResidualTransformer.forward``).

MI355X-first restructuring (all exact re-orderings of the reference computation):
  * everything that does not depend on x_t is hoisted out of the 200-step loop and batched
    over ALL reverse steps at once: the two speaker vectors (t-independent; the
    unconditional one is a model constant), both timestep MLPs, the whole 8-layer text
    encoder for (step, cond/uncond) pairs, and the cross-attention K / V^T projections of all
    16 decoder layers (~1 GB bf16 per utterance, resident in HBM; 288 GB makes this free).
    The loop body is then only the 16-layer decoder + heads;
  * cond / uncond (classifier-free guidance) run as one batch of 2;
  * codebook 0 and the prompt frames are never sampled from the model (m = 1 there), so the
    heads skip them;
  * one step = one captured hipGraph (chunked embedding -> 16 decoder layers -> LayerNorms ->
    7-head batched GEMM) + torch RNG draws + the fused posterior/sample kernel.  The step
    index lives in device memory so the same graph serves all steps.
RNG: the uniforms come from ``torch.rand`` in the reference's order and shapes
((1,S,8,K) twice per step, once at t = 0) so a seeded run consumes the generator exactly
like the reference on the same device.
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from .blocks import (LAYERNORM_EPS, AbsorbedCross, CrossMemory, DeferredLN, RowTiles, SeqWorkspace, SpeakerEncoder, cross_attn_block,
                     cross_memory_table, decoder_layer, decoder_layer_dln, encoder_layer, ff_block, make_cross_plan, pack_layer,
                     plan_allows_dln, residual_gemm, round_up, self_attn_block)
from .synth import NARShape
from .tables import log_eps, nar_step_consts, reverse_schedule, sine_pe, timestep_inputs


LAST_STATS: dict = {}     # filled by NARSession.run: HIP-event timings of the last utterance


class NARModel:
    def __init__(self, sd: Dict[str, torch.Tensor], shape: NARShape, dtype: torch.dtype, device, max_frames: int = 6000):
        self.shape, self.dt, self.dev = shape, dtype, torch.device(device)
        dev, dt = self.dev, dtype
        D, Q, K = shape.dim, shape.n_codebooks, shape.n_quant
        f32 = lambda n: sd[n].to(dev, torch.float32).contiguous()
        wdt = lambda n: sd[n].to(device=dev, dtype=dt).contiguous()
        self.enc = [pack_layer(sd, f"tfm.encoder.layers.{l}", dt, dev) for l in range(shape.enc_layers)]
        self.dec = [pack_layer(sd, f"tfm.decoder.layers.{l}", dt, dev, cross=True) for l in range(shape.dec_layers)]
        self.enc_norm = (f32("tfm.encoder.norm.weight"), f32("tfm.encoder.norm.bias"))
        self.dec_norm = (f32("tfm.decoder.norm.weight"), f32("tfm.decoder.norm.bias"))
        self.te = [(wdt(f"timestep_{w}_emb.0.weight"), f32(f"timestep_{w}_emb.0.bias"),
                    wdt(f"timestep_{w}_emb.2.weight"), f32(f"timestep_{w}_emb.2.bias")) for w in ("encoder", "decoder")]
        self.text_embed = f32("text_embed.weight")
        self.res_tables = torch.stack([sd[f"residual_encoder.embs.{q}.weight"].float() for q in range(Q)]).to(dev).contiguous()
        self.cond_alpha, self.pos_alpha = f32("cond_pos_embedding.alpha"), f32("pos_embedding.alpha")
        # heads 1..Q-1 only: codebook 0 is always taken from the known branch (m[...,0] = 1)
        self.head_g = torch.stack([sd[f"residual_decoder.{q}.0.weight"].float() for q in range(1, Q)]).to(dev).contiguous()
        self.head_b = torch.stack([sd[f"residual_decoder.{q}.0.bias"].float() for q in range(1, Q)]).to(dev).contiguous()
        self.head_w = torch.stack([sd[f"residual_decoder.{q}.1.weight"].float() for q in range(1, Q)]).to(device=dev, dtype=dt).contiguous()
        self.head_bias = torch.stack([sd[f"residual_decoder.{q}.1.bias"].float() for q in range(1, Q)]).to(dev).contiguous()
        # 16-bit engines: the heads' own LayerNorm (gamma_q, beta_q) folded into their Linear -- all seven heads normalise the same
        # rows with the same statistics, so ONE affine-free normalised copy (m5_layernorm_twice) feeds ONE GEMM of N = 7 x Kp
        # columns (Kp = K rounded up to 4: zero rows, so a head's logits start on a 16-byte boundary as before):
        # W'_q = dtype(W_q diag gamma_q), b'_q = b_q + W_q beta_q.  The fp32 parity engine keeps the reference's order.
        self.head_wf = self.head_bf = None
        if dt != torch.float32 and D % 256 == 0:                # (m5_layernorm_twice has the vector form only)
            Kp = round_up(K, 4)
            wf = torch.zeros(Q - 1, Kp, D, dtype=dt)
            bf = torch.zeros(Q - 1, Kp, dtype=torch.float32)
            for q in range(1, Q):
                W, b = sd[f"residual_decoder.{q}.1.weight"].float(), sd[f"residual_decoder.{q}.1.bias"].float()
                g, be = sd[f"residual_decoder.{q}.0.weight"].float(), sd[f"residual_decoder.{q}.0.bias"].float()
                wf[q - 1, :K] = (W * g[None, :]).to(dt)
                bf[q - 1, :K] = b + W @ be
            self.head_wf, self.head_bf = wf.view((Q - 1) * Kp, D).to(dev).contiguous(), bf.view(-1).to(dev).contiguous()
        self.spk = SpeakerEncoder(sd, "ref_embedder", "ref_pos_embedding.alpha", shape.n_spk_layers, dt, dev)
        self.pe = sine_pe(max_frames, D).to(dev)
        self.spk_uncond: Optional[torch.Tensor] = None      # model constant, computed on first use
        self._tvec_cache: Dict[tuple, tuple] = {}           # schedule -> (t_enc, t_dec): both timestep MLPs over every scheduled t

    def timestep_vectors(self, times: List[int], stream: torch.cuda.Stream) -> tuple:
        """(t_enc, t_dec), each (len(times), D) fp32: timestep_embedding -> Linear -> SiLU -> Linear (model.py:18-35,
        :281-283) for every scheduled t.  Depends on the schedule only, so it is computed once per schedule."""
        key = tuple(times)
        hit = self._tvec_cache.get(key)
        if hit is not None:
            return hit
        s, dev, dt = self.shape, self.dev, self.dt
        T, D = len(times), s.dim
        st = stream.cuda_stream
        with torch.cuda.stream(stream):
            tin = timestep_inputs(list(times), s.t_emb_dim).to(device=dev, dtype=dt)
            tvec = []
            for (w0, b0, w2, b2) in self.te:
                h = torch.empty(T, D, dtype=dt, device=dev)
                o = torch.empty(T, D, dtype=torch.float32, device=dev)
                ops.gemm(tin, w0, h, L.EPI_SILU_DT, bias=b0, stream=st)
                ops.gemm(h, w2, o, L.EPI_F32, bias=b2, stream=st)
                tvec.append(o)
            stream.synchronize()                 # cached across sessions (and their streams): make it plainly complete
        if len(self._tvec_cache) > 8:
            self._tvec_cache.clear()
        self._tvec_cache[key] = (tvec[0], tvec[1])
        return self._tvec_cache[key]

    def uncond_speaker(self, stream=None) -> torch.Tensor:
        if self.spk_uncond is None:
            self.spk_uncond = self.spk(None, stream=stream)
        return self.spk_uncond

    def flops_per_step(self, S: int, Le: int, s_out: int, guidance: bool = True) -> float:
        """Algorithmic MFMA flops of one reverse step's loop body (decoder + heads), SURVEY §8d."""
        s = self.shape
        D, FF = s.dim, s.dim_ff
        nb = 2 if guidance else 1
        per_row = 2 * (4 * D * D + 2 * D * D + 3 * D * FF)          # self in/out, cross q/out, SwiGLU + linear2
        att = 4 * (S + Le) * D                                        # QK^T + PV, self + cross
        dec = s.dec_layers * S * (per_row + att)
        heads = s_out * (s.n_codebooks - 1) * 2 * D * s.n_quant
        return float(nb * (dec + heads))


@dataclass
class NARConfig:
    """The DSH fields that reach the shipped inference path (reference diffuser.py:302-315,
    inference.py:289-294) plus T."""
    T: int = 200
    x_0_temp: float = 0.7
    guidance_w: float = 3.0
    deep_clone: bool = True
    q0_override_steps: int = 20
    div_mode: int = 0


def _build_cross_operands(plan, step_ptr: torch.Tensor, st: int) -> None:
    """Enqueue the launch that rebuilds the step's absorbed cross-attention operands (A, c, B^T of all 16 layers for this
    step's memory block: HBM-bound, ~65 us, needed first by layer 0's CROSS-attention).  (As a parallel graph branch beside
    layer 0's self-attention block it measured 45 us per step SLOWER, profiles/r3w_*: on this stack every cross-stream
    dependency inside a step costs more than it hides; that variant and the second-stream uniform draws are gone.)"""
    for seg in plan:
        if seg[0] == "absorbed":
            seg[1].build(step_ptr, st)


GEN_LOCK = threading.Lock()                # serialises the get_offset / set_offset pairs that reserve a generator range (PhiloxDraws.reserve, ar_generate)
_PROBE_CACHE: Dict[tuple, bool] = {}       # (device index, n) -> m5_nar_uniforms reproduces torch.rand of n floats on this torch build


def _philox_geometry(n: int, dev) -> tuple:
    """(G, inc): threads of torch's `uniform_` launch for n floats on this device and the generator offset one draw consumes."""
    prop = torch.cuda.get_device_properties(dev)
    per_mp = max(int(getattr(prop, "max_threads_per_multi_processor", 2048)) // 256, 1)
    G = 256 * min(int(prop.multi_processor_count) * per_mp, (n + 255) // 256)
    return G, ((n - 1) // (4 * G) + 1) * 4


def philox_matches_torch(n: int, dev) -> bool:
    """Once per (device, n) and process: does m5_nar_uniforms reproduce `torch.rand(n)` on THIS torch / ROCm build?  Compared are
    the VALUES of one draw (a private generator; m = NULL form = draw 1 everywhere) and the offset the draw consumes -- the
    offset alone cannot tell launch widths apart (inc = ceil(n / 4G) * 4 is the same for many G, and always 4 when the draw fits
    one grid pass).  False: the caller keeps the eager `torch.rand` path (same results, two ATen launches per step)."""
    key = (torch.device(dev).index, n)
    ok = _PROBE_CACHE.get(key)
    if ok is None:
        ok = False
        try:
            G, inc = _philox_geometry(n, dev)
            gen = torch.Generator(device=dev)
            gen.manual_seed(0x5EED5EED)
            gen.set_offset(8)
            ref = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=gen)
            if int(gen.get_offset()) - 8 == inc:
                out = torch.empty(n, dtype=torch.float32, device=dev)
                rng = torch.tensor([0x5EED5EED, 8], dtype=torch.int64, device=dev)
                a = L.NarUniformArgs(out=out.data_ptr(), n=n, K=1, k_magic=0, k_shift=0, m=None, rng=rng.data_ptr(), inc=inc, grid_threads=G,
                                     step=None, consts=None)
                ops.nar_uniforms(a, stream=torch.cuda.current_stream(dev).cuda_stream)
                ok = bool(torch.equal(out, ref))
        except Exception:          # an unsupported size / another generator implementation: the eager path serves it
            ok = False
        _PROBE_CACHE[key] = ok
    return ok


def _magic_div(d: int, n_max: int = 1 << 32) -> tuple:
    """(m, s) with e // d == (e * m) >> (32 + s) for every 0 <= e < n_max, m a 32-bit multiplier -- or (0, 0) when there is none
    (the kernel then divides).  m = floor(2^(32+s) / d) + 1 overshoots 2^(32+s) / d by r / d with r = m d - 2^(32+s) in (0, d]; the
    quotient stays exact while e r < 2^(32+s)."""
    for s in (max(d.bit_length() - 1, 0), d.bit_length()):
        m = (1 << (32 + s)) // d + 1
        if m < (1 << 32) and (m * d - (1 << (32 + s))) * max(n_max - 1, 0) < (1 << (32 + s)):
            return m, s
    return 0, 0


class PhiloxDraws:
    """The reverse steps' uniforms generated INSIDE the step graph (csrc/nar_sample.hip, m5_nar_uniforms) instead of by two
    ``torch.rand`` launches per step: bit-identical values (torch's Philox4x32-10 launch geometry on this device) and the same
    advance of the caller's generator, which is moved past ALL the run's draws when the run is enqueued (so a pipelined caller
    that draws from the same generator afterwards sees the reference's stream).  One merged buffer: first draw on the rows
    sampled from the model, second draw on the known rows -- what m5_nar_sample reads."""

    def __init__(self, gen: torch.Generator, S: int, n_q: int, K: int, m_mask: torch.Tensor, consts: torch.Tensor, step_ptr: torch.Tensor,
                 times: List[int], dev):
        assert all(t > 0 for t in times[:-1]), "only the last reverse step may have t = 0 (one draw)"
        self.gen, self.n = gen, S * n_q * K
        self.grid_threads, self.inc = _philox_geometry(self.n, dev)
        self.buf = torch.empty(S, n_q, K, dtype=torch.float32, device=dev)
        self.rng = torch.zeros(2, dtype=torch.int64, device=dev)             # {seed, offset0}: read by the kernel, so the step graph outlives a run
        km, ks = _magic_div(K, self.n)
        self.m_mask, self.consts, self.step_ptr = m_mask, consts, step_ptr
        self.args = L.NarUniformArgs(out=self.buf.data_ptr(), n=self.n, K=K, k_magic=km, k_shift=ks, m=m_mask.data_ptr(), rng=self.rng.data_ptr(),
                                     inc=self.inc, grid_threads=self.grid_threads, step=step_ptr.data_ptr(), consts=consts.data_ptr())

    def reserve(self, times: List[int], first_step: int, stream: torch.cuda.Stream) -> None:
        """Bind the steps first_step .. of the session (`times`: their t) to the generator's present state and move the generator
        past their draws (two per step, one at t = 0).  Stream-ordered: the state words reach the device behind whatever the
        stream already holds (an earlier run's steps)."""
        wrap = lambda v: v - (1 << 64) if v >= (1 << 63) else v                 # noqa: E731  (uint64 bit pattern in an int64 tensor)
        with GEN_LOCK:        # get_offset / set_offset is not atomic: two host threads sharing a generator must not bind overlapping ranges
            seed, off = int(self.gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF, int(self.gen.get_offset())
            self.gen.set_offset(off + sum(2 if t > 0 else 1 for t in times) * self.inc)
        with torch.cuda.stream(stream):
            self.rng.copy_(torch.tensor([wrap(seed), wrap((off - 2 * first_step * self.inc) & 0xFFFFFFFFFFFFFFFF)], dtype=torch.int64), non_blocking=True)

    def enqueue(self, st: int) -> None:
        ops.nar_uniforms(self.args, stream=st)


class NARSession:
    """One utterance.  ``prepare`` = everything x_t-independent; ``step`` = one reverse step."""

    def __init__(self, model: NARModel, cfg: NARConfig, stream: Optional[torch.cuda.Stream] = None, diff_tables=None):
        """`diff_tables`: the four fp32 log-tables of the caller's MultinomialDiffusion (None = default schedule)."""
        self.m, self.cfg = model, cfg
        self.stream = stream if stream is not None else ops.session_stream(model.dev, "nar")
        self.graph: Optional[ops.Graph] = None
        self.graph_step: Optional[ops.Graph] = None
        self._ph: Optional[PhiloxDraws] = None
        self.diff_tables = diff_tables
        self.step_plan = None
        self.use_c_plan = True                      # False: the step's launches are composed in this file (tests compare the two)

    def _enter(self) -> None:
        """Order this session's stream behind whatever the caller has already enqueued on its current stream (inputs
        produced there: codec codes, the AR hand-off, memsets of fresh allocations)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.m.dev))

    # -------------------------------------------------------------------------- prepare
    def prepare(self, c_text: torch.Tensor, c_codes: torch.Tensor, x: torch.Tensor, x_known: torch.Tensor, m_mask: torch.Tensor,
                row_offset: int, times: Optional[List[int]] = None) -> None:
        """c_text (Lt,), c_codes (Lc, 8): conditioning.  x / x_known (S, 8) int64, m_mask (S, 8)
        uint8: the inpainting state built by ``perform_simple_inference`` (diffuser.py:405-436).
        row_offset: prompt frames prepended in deep-clone mode (never sampled from the model)."""
        self.prepare_state(x, x_known, m_mask, row_offset)
        self.prepare_cond(c_text, c_codes, times)
        self.prepare_loop()

    def prepare_state(self, x: torch.Tensor, x_known: torch.Tensor, m_mask: torch.Tensor, row_offset: int) -> None:
        """The inpainting state of one utterance (depends on the AR output)."""
        dev = self.m.dev
        self._enter()
        with torch.cuda.stream(self.stream):
            self.x = ops.use_on(x.to(dev), self.stream).contiguous().clone()
            self.x_known = ops.use_on(x_known.to(dev), self.stream).contiguous()
            self.m_mask = ops.use_on(m_mask.to(device=dev), self.stream).to(dtype=torch.uint8).contiguous()
        self.S, self.row_offset = int(self.x.shape[0]), int(row_offset)
        self.s_out = self.S - self.row_offset
        assert self.S <= self.m.pe.shape[0]

    def adopt_cond(self, other: "NARSession") -> None:
        """Take over the conditioning state another session computed for the SAME (text ids, reference codes, schedule,
        guidance setting): per-layer cross-attention memories, timestep vectors, schedule constants.  Read-only in the
        loop, so sessions may share it; the donor's work must be complete (its stream synchronised)."""
        assert other.cfg == self.cfg and other.m is self.m
        self.times, self.nb, self.t_dec, self.mems, self.consts = other.times, other.nb, other.t_dec, other.mems, other.consts
        self._keep = getattr(other, "_keep", [])
        # order this session's stream behind the donor's conditioning work (recorded at the end of its prepare_cond): the donor
        # may still be running on another stream / worker thread
        ev = getattr(other, "cond_ready", None)
        if ev is not None:
            self.stream.wait_event(ev)
        self.cond_ready = ev

    def prepare_cond(self, c_text: torch.Tensor, c_codes: torch.Tensor, times: Optional[List[int]] = None,
                     spk_vec: Optional[torch.Tensor] = None) -> None:
        """Everything that depends only on the conditioning (text ids, reference codes) and the
        schedule - independent of the AR output, so a server can run it beside the AR decode:
        speaker vectors, timestep MLPs, the text encoder for every (step, cond/uncond) pair and the
        cross-attention K / V^T of every decoder layer."""
        mdl, s, cfg = self.m, self.m.shape, self.cfg
        dev, dt = mdl.dev, mdl.dt
        st = self.stream.cuda_stream
        self.times = reverse_schedule(cfg.T) if times is None else list(times)
        T = len(self.times)
        D, H, FF, K, Q = s.dim, s.dim // 64, s.dim_ff, s.n_quant, s.n_codebooks
        guided = cfg.guidance_w != 1
        nb = 2 if guided else 1
        self.nb = nb
        self._enter()
        with torch.cuda.stream(self.stream):
            # Every host -> device copy of this function happens HERE, in front of the launches: a pageable copy blocks the host
            # until the stream has executed it -- behind the text encoder and the 16 K / V projections that used to be 6 ms in which
            # the host could not prepare the AR stage (tools/host_profile.py).  (The caller orders the AR stream behind `cond_ready`:
            # see inference._tts_core for why the two stages do not run concurrently.)
            c_text = ops.use_on(c_text.to(dev), self.stream)
            c_codes = ops.use_on(c_codes.to(dev).contiguous(), self.stream)
            ops.use_on(spk_vec, self.stream)
            self.consts = nar_step_consts(self.times, K, tables=self.diff_tables).to(dev)
            Lt = int(c_text.shape[0])
            Le = Lt + 1
            # -- speaker vectors (t-independent) and timestep MLPs for every scheduled t
            spk_c = mdl.spk(c_codes, stream=st) if spk_vec is None else spk_vec.to(dev)
            rows = [spk_c]
            if guided:
                rows.append(mdl.uncond_speaker(stream=st))
            t_enc, self.t_dec = mdl.timestep_vectors(self.times, self.stream)
            # -- encoder input for every (step, cond/uncond): [spk, text] + pos + t_enc[step]
            table = torch.cat([mdl.text_embed] + [r[None] for r in rows], dim=0)
            nt = mdl.text_embed.shape[0]
            one = torch.cat([torch.zeros(1, dtype=c_text.dtype, device=dev), c_text])      # row 0 placeholder (device-side fill: no host copy)
            idx = one[None, None, :].repeat(T, nb, 1)
            for b in range(nb):
                idx[:, b, 0] = nt + b
            pos = torch.arange(Le, device=dev, dtype=torch.int32)[None, None].expand(T, nb, Le).contiguous()
            aidx = torch.arange(T, device=dev, dtype=torch.int32)[:, None, None].expand(T, nb, Le).contiguous()
            Me = T * nb * Le
            c = torch.empty(Me, D, dtype=torch.float32, device=dev)
            ops.gather_rows(c, table, idx.reshape(-1).contiguous(), mdl.cond_alpha, mdl.pe, pos.reshape(-1), t_enc,
                            aidx.reshape(-1), stream=st)
            wse = SeqWorkspace(T * nb, Le, D, FF, dt, dev)
            for lw in mdl.enc:
                encoder_layer(c, lw, wse, None, st)
            mem = torch.empty(Me, D, dtype=dt, device=dev)
            ops.layernorm(c, mdl.enc_norm[0], mdl.enc_norm[1], LAYERNORM_EPS, mem, stream=st)
            # -- cross-attention K / V^T of every decoder layer for every step
            Lep = round_up(Le, 64)
            self.mems: List[CrossMemory] = []
            # short memory: K and V as rows for the absorbed form (blocks.AbsorbedCross); M5_NAR_ABSORB=0: A/B knob (tools/nar_step_bench.py)
            absorbed = dt != torch.float32 and AbsorbedCross.lp_of(Le, H) > 0 and H * 64 <= FF and L.tool_knob("M5_NAR_ABSORB", "1") != "0"
            for lw in mdl.dec:
                k = torch.empty(T * nb, H, Le, 64, dtype=dt, device=dev)
                if absorbed:
                    v_rows = torch.empty(T * nb, H, Le, 64, dtype=dt, device=dev)
                    for w, b, dst in ((lw.ca_k_w, lw.ca_kv_b[:D], k), (lw.ca_v_w, lw.ca_kv_b[D:], v_rows)):
                        sc = L.QkvScatter(q=None, k=dst.data_ptr(), vt=None, rows_per_batch=Le, n_heads=H, head_dim=64,
                                          q_bs=0, q_hs=0, q_rs=0, k_bs=H * Le * 64, k_hs=Le * 64, k_rs=64, vt_bs=0, vt_hs=0, vt_ds=0)
                        ops.gemm(mem, w, None, L.EPI_QKV, bias=b, scatter=sc, stream=st)
                    self.mems.append(CrossMemory(k, None, Le, Lep, nb, v_rows=v_rows))
                    continue
                vt = torch.zeros(T * nb, H, 64, Lep, dtype=dt, device=dev)
                sc = L.QkvScatter(q=None, k=k.data_ptr(), vt=vt.data_ptr(), rows_per_batch=Le, n_heads=H, head_dim=64,
                                  q_bs=0, q_hs=0, q_rs=0, k_bs=H * Le * 64, k_hs=Le * 64, k_rs=64,
                                  vt_bs=H * 64 * Lep, vt_hs=64 * Lep, vt_ds=Lep)
                ops.gemm(mem, lw.ca_kv_w, None, L.EPI_QKV, bias=lw.ca_kv_b, scatter=sc, stream=st)
                self.mems.append(CrossMemory(k, vt, Le, Lep, nb))
            self._keep = [table, t_enc, mem]
            self.cond_ready = torch.cuda.Event()
            self.cond_ready.record(self.stream)

    def prepare_loop(self) -> None:
        """Loop-body buffers of a single-utterance session."""
        mdl, s = self.m, self.m.shape
        dev, dt = mdl.dev, mdl.dt
        D, FF, K, Q = s.dim, s.dim_ff, s.n_quant, s.n_codebooks
        S, nb = self.S, self.nb
        self._enter()
        with torch.cuda.stream(self.stream):
            self.ws = SeqWorkspace(nb, S, D, FF, dt, dev, row_pad=64, fuse_ln=True)
            self.Sr = Sr = self.ws.Sr
            # the two guidance branches enter the decoder with the SAME rows (x_t embedding + timestep vector) and first
            # differ in layer 0's cross-attention, so layer 0's self-attention block runs once (one-sequence workspace)
            share0 = L.tool_knob("M5_NAR_SHARE0", "1") != "0"          # A/B knob (tools/nar_step_bench.py)
            over = self.ws if L.tool_knob("M5_NAR_ALIAS", "1") != "0" else None     # A/B knob: private buffers for the two sub-problems
            self.ws0 = SeqWorkspace(1, S, D, FF, dt, dev, row_pad=64, inside=over) if (nb == 2 and share0) else None
            self.h = torch.zeros(nb, Sr, D, dtype=torch.float32, device=dev)
            self.hf = torch.zeros(nb * Sr, D, dtype=torch.float32, device=dev)
            self.Kp = round_up(K, 4)
            self.logits = torch.empty(nb * self.s_out, Q - 1, self.Kp, dtype=torch.float32, device=dev)
            self.step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
            self.step_i = 0
            # cross-attention path of this utterance: absorbed operands (rebuilt per step by one launch) or the reference order
            want_dln = DeferredLN.eligible(D, dt) and L.tool_knob("M5_NAR_DLN", "1") != "0"      # A/B knob (tools/nar_step_bench.py)
            self.plan = make_cross_plan(mdl.dec, [[mem] for mem in self.mems], D, dt, dev, dln=want_dln)
            # LayerNorms deferred into the consuming GEMMs (blocks.decoder_layer_dln) when every utterance takes the absorbed path
            self.dl = DeferredLN(self.ws, dev) if (want_dln and plan_allows_dln(self.plan)) else None
            self.xa = [cross_memory_table(mem, dev) if mem.vt is not None else None for mem in self.mems]   # opt-in fused q-proj + attention
            # Deep clone: the prompt frames (row_offset of S rows) are never sampled, so in the LAST decoder layer their
            # rows are needed only as keys / values.  That layer's queries, projections and feed-forward run on the
            # s_out generated rows of each branch, gathered into a compact workspace (exact: every kernel is row-wise).
            self.ws_l = None
            so_r = round_up(self.s_out, 64)
            # (not with deferred LayerNorms: the compact layer would cost a row copy + three explicit LayerNorm launches to save
            # 0.1 % of the step)
            if self.dl is None and self.row_offset > 0 and 4 * so_r <= 3 * Sr and L.tool_knob("M5_NAR_LASTROWS", "1") != "0":
                self.ws_l = SeqWorkspace(nb, self.s_out, D, FF, dt, dev, row_pad=64, inside=over)
                self.x_l = torch.zeros(nb, so_r, D, dtype=torch.float32, device=dev)
                self.hf_l = torch.zeros(nb * so_r, D, dtype=torch.float32, device=dev)
            # folded heads (one normalised copy feeds ONE GEMM over all heads) unless the compact last layer is in use; the
            # reference order keeps one normalised slab per head
            self.fold_heads = mdl.head_wf is not None and self.ws_l is None and L.tool_knob("M5_NAR_HEADFOLD", "1") != "0"
            self.hn = torch.empty(1 if self.fold_heads else Q - 1, nb * self.s_out, D, dtype=dt, device=dev)
        self.graph = None
        self.graph_step, self._ph = None, None      # whole-step graph and in-graph uniform generator (made at the first run that can use them)
        self.step_plan = None                       # (ops.StagePlan of one reverse step, the PhiloxDraws it was recorded with)

    # ----------------------------------------------------------------------------- step
    def _last_layer_compact(self, lw, mem, normed: bool, st: int) -> None:
        """The last decoder layer + final LayerNorm for the generated rows only (see prepare_loop): keys / values come
        from ALL rows, queries and everything after the attention from rows row_offset..S-1 of each branch."""
        mdl, s = self.m, self.m.shape
        S, Sr, nb, D, H = self.S, self.Sr, self.nb, s.dim, s.dim // 64
        ws, wl = self.ws, self.ws_l
        so, so_r, off = self.s_out, wl.Sr, self.row_offset
        hx = self.h.view(nb * Sr, D)
        if not normed:
            ops.layernorm(hx, lw.n1_w, lw.n1_b, LAYERNORM_EPS, ws.xn, stream=st)
        ops.gemm(ws.xn, lw.in_w, None, L.EPI_QKV, bias=lw.in_b, scatter=ws.scatter(), stream=st)
        a = ws.self_attn_args(None)
        a.q = ws.q.data_ptr() + off * 64 * ws.q.element_size()          # queries row_offset.. of every (branch, head)
        a.Sq = so
        a.o, a.o_bs = wl.att.data_ptr(), so_r * D
        ops.attention(ws.dt, a, stream=st)
        ops.mark("torch copy: generated rows -> compact workspace", st)
        with torch.cuda.stream(self.stream):                              # same stream: captured into the step graph
            self.x_l[:, :so].copy_(self.h[:, off:S])
            if so_r > so:
                self.x_l[:, so:].zero_()                                  # pad rows: finite, never read as keys
        xl = self.x_l.view(nb * so_r, D)
        residual_gemm(wl.att, lw.out_w, xl, lw.out_b, wl, None, st)
        cross_attn_block(xl, lw, wl, mem, self.step_ptr, st, plan=self.plan, layer=len(mdl.dec) - 1)
        ff_block(xl, lw, wl, lw.n3_w, lw.n3_b, st)
        ops.layernorm(xl, mdl.dec_norm[0], mdl.dec_norm[1], LAYERNORM_EPS, self.hf_l, stream=st)

    def _enqueue_decoder_dln(self, st: int) -> None:
        """The 16 decoder layers with deferred LayerNorms (blocks.decoder_layer_dln)."""
        mdl, s = self.m, self.m.shape
        S, Sr, nb, D = self.S, self.Sr, self.nb, s.dim
        hx = self.h.view(nb * Sr, D)
        nl = len(mdl.dec)
        _build_cross_operands(self.plan, self.step_ptr, st)
        skip0 = False
        if self.ws0 is not None:
            # layer 0's self-attention block once for both guidance branches (they enter the decoder with the same rows)
            ops.chunked_embed(self.h[:1], mdl.res_tables, self.x, None, mdl.pos_alpha, mdl.pe, add=self.t_dec, add_index=self.step_ptr,
                              rows=S, stream=st)
            self_attn_block(hx[:Sr], mdl.dec[0], self.ws0, None, st)
            ops.mark("copy: branch 0 -> branch 1", st)
            ops.copy_d2d(self.h[1], self.h[0], stream=st)         # (a library copy: capturable AND visible to a stage plan)
            skip0 = True
        else:
            ops.chunked_embed(self.h, mdl.res_tables, self.x, None, mdl.pos_alpha, mdl.pe, add=self.t_dec, add_index=self.step_ptr,
                              rows=S, stream=st)
        for l, lw in enumerate(mdl.dec):
            decoder_layer_dln(hx, lw, self.ws, self.step_ptr, self.dl, self.plan, l, chain_in=l > 0, chain_out=l + 1 < nl, stream=st,
                              skip_self=(skip0 and l == 0), mems=[self.mems[l]])

    def enqueue_forward(self, st: int) -> None:
        """x_t -> logits for both guidance branches (the loop body's GEMM/attention work)."""
        mdl, s = self.m, self.m.shape
        S, Sr, nb, D, Q = self.S, self.Sr, self.nb, s.dim, s.n_codebooks
        hx = self.h.view(nb * Sr, D)
        if self.dl is not None:
            self._enqueue_decoder_dln(st)
            self._enqueue_heads(hx, st)
            return
        layers = list(zip(mdl.dec, self.mems))
        self.ws.ln_tag, self.ws.ln_tag_step = 0, self.step_ptr      # fused LN launches: tag = f(step counter, call index)
        _build_cross_operands(self.plan, self.step_ptr, st)
        if self.ws0 is not None:
            ops.chunked_embed(self.h[:1], mdl.res_tables, self.x, None, mdl.pos_alpha, mdl.pe, add=self.t_dec, add_index=self.step_ptr,
                              rows=S, stream=st)
            lw, mem = layers[0]
            self_attn_block(hx[:Sr], lw, self.ws0, None, st)
            ops.mark("copy: branch 0 -> branch 1", st)
            ops.copy_d2d(self.h[1], self.h[0], stream=st)         # same stream (captured into the step graph)
            nxt = (mdl.dec[1].n1_w, mdl.dec[1].n1_b) if len(mdl.dec) > 1 else None
            normed = cross_attn_block(hx, lw, self.ws, mem, self.step_ptr, st, next_ln=(lw.n3_w, lw.n3_b), xa=self.xa[0], plan=self.plan, layer=0)
            normed = ff_block(hx, lw, self.ws, lw.n3_w, lw.n3_b, st, normed=normed, next_ln=nxt)
            layers = layers[1:]
            l0 = 1
        else:
            ops.chunked_embed(self.h, mdl.res_tables, self.x, None, mdl.pos_alpha, mdl.pe, add=self.t_dec, add_index=self.step_ptr,
                              rows=S, stream=st)
            normed, l0 = False, 0
        compact = self.ws_l is not None and len(layers) >= 1
        for k, (lw, mem) in enumerate(layers):
            l = l0 + k                                         # every LayerNorm but the first rides on the residual GEMM before it
            if compact and l == len(mdl.dec) - 1:
                self._last_layer_compact(lw, mem, normed, st)
                break
            nxt = (mdl.dec[l + 1].n1_w, mdl.dec[l + 1].n1_b) if l + 1 < len(mdl.dec) else None
            normed = decoder_layer(hx, lw, self.ws, mem, self.step_ptr, st, normed=normed, next_ln=nxt, xa=self.xa[l], plan=self.plan, layer=l)
        self._enqueue_heads(hx, st, compact)

    def _enqueue_heads(self, hx: torch.Tensor, st: int, compact: bool = False) -> None:
        """Final decoder LayerNorm + the 7 codebook heads on the generated rows of each branch."""
        mdl, s = self.m, self.m.shape
        Sr, nb, D, Q = self.Sr, self.nb, s.dim, s.n_codebooks
        so = self.s_out
        if self.fold_heads:
            # final LayerNorm + the heads' (folded) LayerNorm of the generated rows of both branches in one launch, then ONE GEMM
            ops.layernorm_twice(hx[self.row_offset:], mdl.dec_norm[0], mdl.dec_norm[1], LAYERNORM_EPS, 1e-5, self.hn[0], so, n_seq=nb, x_seq_stride=Sr,
                                stream=st)
            ops.gemm(self.hn[0], mdl.head_wf, self.logits.view(nb * so, (Q - 1) * self.Kp), L.EPI_F32, bias=mdl.head_bf, stream=st)
            return
        if compact:
            hf, hrow = self.hf_l, [b * self.ws_l.Sr for b in range(nb)]
        else:
            ops.layernorm(hx, mdl.dec_norm[0], mdl.dec_norm[1], LAYERNORM_EPS, self.hf, stream=st)
            hf, hrow = self.hf, [b * Sr + self.row_offset for b in range(nb)]
        for b in range(nb):
            ops.layernorm(hf[hrow[b]:], mdl.head_g, mdl.head_b, 1e-5, self.hn[:, b * so:], n_affine=Q - 1,
                          affine_stride=D, y_affine_stride=nb * so * D, M=so, stream=st)
        ops.gemm(self.hn[0], mdl.head_w[0], self.logits, L.EPI_F32, bias=mdl.head_bias, ldc=(Q - 1) * self.Kp, batch=Q - 1,
                 sA=nb * so * D, sW=s.n_quant * D, sC=self.Kp, sBias=s.n_quant, stream=st)

    def enqueue_sample(self, u1: torch.Tensor, u2: Optional[torch.Tensor], st: int) -> None:
        s, cfg = self.m.shape, self.cfg
        so, Q = self.s_out, s.n_codebooks
        ld_row = (Q - 1) * self.Kp
        lu = self.logits[so:] if self.nb == 2 else None
        a = L.NarSampleArgs(logits_c=self.logits.data_ptr(), logits_u=lu.data_ptr() if lu is not None else None,
                            ld_row=ld_row, ld_q=self.Kp, S=self.S, n_q=Q, K=s.n_quant, row_offset=self.row_offset,
                            x=self.x.data_ptr(), x_known=self.x_known.data_ptr(), m=self.m_mask.data_ptr(),
                            u1=u1.data_ptr(), u2=(u2 if u2 is not None else u1).data_ptr(), consts=self.consts.data_ptr(),
                            step=self.step_ptr.data_ptr(), guidance_w=cfg.guidance_w, temperature=cfg.x_0_temp,
                            log_eps=log_eps(), div_mode=cfg.div_mode, q0_override_steps=cfg.q0_override_steps)
        ops.nar_sample(a, stream=st)
        ops.add_int(self.step_ptr, 1, stream=st)

    def _philox(self, uniform) -> Optional[PhiloxDraws]:
        """The in-graph generator for this session's draws, if `uniform` is the generator-backed draw of diffuser.py (it carries
        its generator) and the schedule has its only t = 0 step last; else None (explicit uniforms: parity tests)."""
        gen = getattr(uniform, "gen", None)
        if gen is None or not all(t > 0 for t in self.times[:-1]) or L.tool_knob("M5_NAR_PHILOX", "1") == "0":      # A/B knob (tools/nar_step_bench.py)
            return None
        ph = self._ph
        if ph is None or ph.gen is not gen:
            s = self.m.shape
            with torch.cuda.stream(self.stream):
                if not philox_matches_torch(self.S * s.n_codebooks * s.n_quant, self.m.dev):
                    return None                     # another torch build's launch geometry: the eager torch.rand path serves it
                ph = self._ph = PhiloxDraws(gen, self.S, s.n_codebooks, s.n_quant, self.m_mask, self.consts, self.step_ptr, self.times, self.m.dev)
            self.graph_step = None
        return ph

    def _step_philox(self, ph: PhiloxDraws, use_graph: bool) -> None:
        """One reverse step as ONE launch sequence -- forward, the step's uniforms (generated here, bit-identical to torch.rand's),
        posterior / sample, step counter -- replayed as one hipGraph from the second step on."""
        st = self.stream.cuda_stream

        def compose(s_):
            self.enqueue_forward(s_)
            ph.enqueue(s_)
            self.enqueue_sample(ph.buf, ph.buf, s_)

        def body():
            # The whole reverse step through ONE C call (include/mars5_hip.h m5_nar_step): the ~190 launches are recorded once
            # per session into a stage plan -- all their arguments are pointers / strides / sizes, the step index lives in
            # device memory -- and composed by the library from then on.  (Deferred-LayerNorm engines; M5_NAR_CPLAN=0 is the
            # tools A/B knob that keeps the composition in this file.)
            if self.dl is None or not self.use_c_plan or L.tool_knob("M5_NAR_CPLAN", "1") == "0":
                return compose(st)
            if self.step_plan is None or self.step_plan[1] is not ph:
                pl = ops.StagePlan("nar_step")
                with pl.recording():
                    compose(0)
                self.step_plan = (pl, ph)
            self.step_plan[0].run(st)
        if not use_graph or (self.graph_step is None and self.step_i == 0 and len(self.times) > 1):
            body()                                  # first step launch by launch: the capture below runs while the GPU works on it
        else:
            if self.graph_step is None:
                ops.Graph.begin(st)
                body()
                self.graph_step = ops.Graph().end(st)
            self.graph_step.launch(st)
        self.step_i += 1

    def step(self, uniform: Callable[[tuple], torch.Tensor], use_graph: bool = True) -> None:
        """One reverse step t = times[step_i]: forward (graph), uniform draws, sample."""
        st = self.stream.cuda_stream
        t = self.times[self.step_i]
        shape = (1, self.S, self.m.shape.n_codebooks, self.m.shape.n_quant)
        if use_graph and self.graph is None and self.step_i == 0 and len(self.times) > 1:
            # The first step goes out launch by launch, and the step graph is captured (a few ms of host time) while the GPU works on
            # it: with the capture in front of the first step the GPU sat idle for its duration (round 4).
            self.enqueue_forward(st)
        elif use_graph:
            if self.graph is None:
                ops.Graph.begin(st)
                self.enqueue_forward(st)
                self.graph = ops.Graph().end(st)
            self.graph.launch(st)
        else:
            self.enqueue_forward(st)
        with torch.cuda.stream(self.stream):
            u1 = uniform(shape)
            u2 = uniform(shape) if t > 0 else None
            self.enqueue_sample(u1[0], u2[0] if u2 is not None else None, st)
        self.step_i += 1

    def run(self, uniform: Callable[[tuple], torch.Tensor], use_graph: bool = True, n_steps: Optional[int] = None,
            on_step: Optional[Callable[[dict], None]] = None, wait: bool = True) -> torch.Tensor:
        """`on_step` (parity tests only; synchronises every step): called after each reverse step with
        {i, t, x_t, x_tm1, u1, u2} so a checker can replay the step.  `wait=False`: enqueue all steps and return without
        synchronising (pipelined serving: the host goes on to the next request's AR decode); call ``finish()`` for x."""
        n = len(self.times) if n_steps is None else n_steps
        st = self.stream.cuda_stream
        ev0, ev1 = ops.Event(), ops.Event()
        ev0.record(st)
        ph = self._philox(uniform) if on_step is None else None
        if ph is not None:
            ph.reserve(self.times[self.step_i: self.step_i + n], self.step_i, self.stream)
        for i in range(n):
            if ph is not None:
                self._step_philox(ph, use_graph)
                continue
            if on_step is None:
                self.step(uniform, use_graph)
                continue
            with torch.cuda.stream(self.stream):
                x_t = self.x.clone()
            drawn: List[torch.Tensor] = []

            def spy(shape, _d=drawn):
                _d.append(uniform(shape))
                return _d[-1]

            t = self.times[self.step_i]
            self.step(spy, use_graph)
            self.stream.synchronize()
            on_step(dict(i=i, t=t, x_t=x_t, x_tm1=self.x.clone(), u1=drawn[0][0], u2=drawn[1][0] if len(drawn) > 1 else None))
        ev1.record(st)
        self._pending = (ev0, ev1, n)
        return self.finish() if wait else self.x

    def finish(self) -> torch.Tensor:
        """Wait for the steps ``run`` enqueued; returns x (S, 8)."""
        ev0, ev1, n = self._pending
        self.stream.synchronize()
        if self.ws.ln_scratch is not None and int(self.ws.ln_scratch[:4].view(torch.int32)[0]) != 0:
            raise RuntimeError("fused residual+LayerNorm GEMM: a row-tile wait timed out (grid not co-resident?)")
        LAST_STATS.update(loop_ms=ev0.elapsed_ms(ev1), steps=n, S=self.S, s_out=self.s_out, Le=self.mems[0].Le, nb=self.nb)
        return self.x


class NARBatchSession:
    """U utterances of mixed lengths refined together (BASELINE config 3: a batch of independent
    requests on one GPU).  The reference refines one utterance per ``tts()`` call; nothing couples
    utterances, so batching is an exact re-ordering: per reverse step ONE decoder pass runs over the
    2U row-concatenated (cond, uncond) sequences - every projection / SwiGLU / head GEMM sees
    M = 2U * Sr rows instead of 2 * Sr, which is what fills the 256 CUs - self-attention masks each
    sequence to its own length (``key_len``), cross-attention is launched per utterance against that
    utterance's pre-projected memory, and the posterior/sample kernel runs per utterance with that
    utterance's own uniforms, so every utterance consumes its RNG stream exactly as it would alone.
    Rows between an utterance's length and the common padded length hold finite junk that no real
    row ever attends to and that is never sampled."""

    def __init__(self, model: NARModel, cfg: NARConfig, stream: Optional[torch.cuda.Stream] = None, diff_tables=None):
        self.m, self.cfg = model, cfg
        self.stream = stream if stream is not None else ops.session_stream(model.dev, "nar")
        self.graph: Optional[ops.Graph] = None
        self.graph_step: Optional[ops.Graph] = None
        self._phs: Optional[List[PhiloxDraws]] = None
        self.step_plan = None
        self.use_c_plan = True
        self.subs: List[NARSession] = []
        self.diff_tables = diff_tables

    def prepare(self, items: List[dict], times: Optional[List[int]] = None) -> None:
        """items: dicts with the arguments of ``NARSession.prepare`` (c_text, c_codes, x, x_known,
        m_mask, row_offset), one per utterance."""
        mdl, s = self.m, self.m.shape
        dev, dt = mdl.dev, mdl.dt
        D, FF, K, Q = s.dim, s.dim_ff, s.n_quant, s.n_codebooks
        assert len(items) >= 1
        # utterances that take the same cross-attention path sit next to each other in the workspace (one batched launch
        # pair per path); results go back in the caller's order
        cls = lambda it: AbsorbedCross.lp_of(int(it["c_text"].shape[0]) + 1, D // 64) if dt != torch.float32 else 0     # noqa: E731
        self._order = sorted(range(len(items)), key=lambda i: cls(items[i]))
        items = [items[i] for i in self._order]
        self.subs = []
        for it in items:
            sub = NARSession(mdl, self.cfg, self.stream, self.diff_tables)
            sub.prepare_state(it["x"], it["x_known"], it["m_mask"], it["row_offset"])
            sub.prepare_cond(it["c_text"], it["c_codes"], times)
            self.subs.append(sub)
        self.times = self.subs[0].times
        self.nb = nb = self.subs[0].nb
        U = len(self.subs)
        S_max = max(sub.S for sub in self.subs)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            # Deferred LayerNorms + row-tile lists (every utterance on the absorbed cross-attention path): sequences start on
            # 384-row boundaries and the GEMMs / attention skip the tiles that hold only padding -- a group no longer pays for
            # being padded to its longest member.  M5_NAR_ROWTILES=0: A/B knob (tools/nar_batch_bench.py).
            want_dln = DeferredLN.eligible(D, dt) and L.tool_knob("M5_NAR_DLN", "1") != "0"
            use_rt = want_dln and L.tool_knob("M5_NAR_ROWTILES", "1") != "0"
            self.ws = SeqWorkspace(U * nb, S_max, D, FF, dt, dev, row_pad=RowTiles.ALIGN if use_rt else 64)
            self.Sr = Sr = self.ws.Sr
            self.rt = RowTiles([sub.S for sub in self.subs for _ in range(nb)], Sr, dev) if use_rt else None
            self.key_len = torch.tensor([sub.S for sub in self.subs for _ in range(nb)], dtype=torch.int32, device=dev)
            self.h = torch.zeros(U * nb, Sr, D, dtype=torch.float32, device=dev)
            self.hf = torch.zeros(U * nb * Sr, D, dtype=torch.float32, device=dev)
            self.row0 = []
            r = 0
            for sub in self.subs:
                self.row0.append(r)
                r += nb * sub.s_out
            self.R = r
            self.fold_heads = mdl.head_wf is not None and L.tool_knob("M5_NAR_HEADFOLD", "1") != "0"
            self.hn = torch.empty(1 if self.fold_heads else Q - 1, self.R, D, dtype=dt, device=dev)
            self.Kp = round_up(K, 4)
            self.logits = torch.empty(self.R, Q - 1, self.Kp, dtype=torch.float32, device=dev)
            self.step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
            self.step_i = 0
            self.plan = make_cross_plan(mdl.dec, [[sub.mems[l] for sub in self.subs] for l in range(len(mdl.dec))], D, dt, dev, dln=want_dln)
            self.dl = DeferredLN(self.ws, dev) if (want_dln and plan_allows_dln(self.plan)) else None
            if self.dl is None:
                assert self.rt is None, "row-tile lists need the deferred-LayerNorm path"
            if self.rt is not None:
                # the per-run sub-lists now (host -> device copies), not inside the first -- possibly captured -- launch sequence
                self.rt.prebuild([(seg[1].s0, seg[1].n_seq) if seg[0] == "absorbed" else (seg[1], seg[2] - seg[1]) for seg in self.plan])
        self.graph = None
        self.graph_step, self._phs = None, None

    def enqueue_forward(self, st: int) -> None:
        mdl, s = self.m, self.m.shape
        Sr, nb, D, Q = self.Sr, self.nb, s.dim, s.n_codebooks
        t_dec = self.subs[0].t_dec                           # depends on the schedule only
        for u, sub in enumerate(self.subs):
            ops.chunked_embed(self.h[u * nb:(u + 1) * nb], mdl.res_tables, sub.x, None, mdl.pos_alpha, mdl.pe, add=t_dec,
                              add_index=self.step_ptr, rows=sub.S, stream=st)
        hx = self.h.view(-1, D)
        _build_cross_operands(self.plan, self.step_ptr, st)
        nl = len(mdl.dec)
        for l, lw in enumerate(mdl.dec):
            if self.dl is not None:
                decoder_layer_dln(hx, lw, self.ws, self.step_ptr, self.dl, self.plan, l, chain_in=l > 0, chain_out=l + 1 < nl, stream=st,
                                  key_len=self.key_len, rt=self.rt, mems=[sub.mems[l] for sub in self.subs])
                continue
            decoder_layer(hx, lw, self.ws, [sub.mems[l] for sub in self.subs], self.step_ptr, st, key_len=self.key_len, plan=self.plan, layer=l)
        if self.fold_heads:
            for u, sub in enumerate(self.subs):
                ops.layernorm_twice(hx[u * nb * Sr + sub.row_offset:], mdl.dec_norm[0], mdl.dec_norm[1], LAYERNORM_EPS, 1e-5, self.hn[0, self.row0[u]:],
                                    sub.s_out, n_seq=nb, x_seq_stride=Sr, stream=st)
            ops.gemm(self.hn[0], mdl.head_wf, self.logits.view(self.R, (Q - 1) * self.Kp), L.EPI_F32, bias=mdl.head_bf, stream=st)
            return
        ops.layernorm(hx, mdl.dec_norm[0], mdl.dec_norm[1], LAYERNORM_EPS, self.hf, stream=st)
        for u, sub in enumerate(self.subs):
            so = sub.s_out
            for b in range(nb):
                ops.layernorm(self.hf[(u * nb + b) * Sr + sub.row_offset:], mdl.head_g, mdl.head_b, 1e-5,
                              self.hn[:, self.row0[u] + b * so:], n_affine=Q - 1, affine_stride=D, y_affine_stride=self.R * D, M=so,
                              stream=st)
        ops.gemm(self.hn[0], mdl.head_w[0], self.logits, L.EPI_F32, bias=mdl.head_bias, ldc=(Q - 1) * self.Kp, batch=Q - 1,
                 sA=self.R * D, sW=s.n_quant * D, sC=self.Kp, sBias=s.n_quant, stream=st)

    def step(self, uniforms: List[Callable[[tuple], torch.Tensor]], use_graph: bool = True) -> None:
        st = self.stream.cuda_stream
        s, cfg = self.m.shape, self.cfg
        t = self.times[self.step_i]
        Q = s.n_codebooks
        shapes = [(1, sub.S, Q, s.n_quant) for sub in self.subs]
        draws = [uniforms[self._order[u]] for u in range(len(self.subs))]         # uniforms are in the caller's order
        if use_graph and self.graph is None and self.step_i == 0 and len(self.times) > 1:
            self.enqueue_forward(st)                # first step eager, the graph is captured behind it (NARSession.step)
        elif use_graph:
            if self.graph is None:
                ops.Graph.begin(st)
                self.enqueue_forward(st)
                self.graph = ops.Graph().end(st)
            self.graph.launch(st)
        else:
            self.enqueue_forward(st)
        with torch.cuda.stream(self.stream):
            for u, sub in enumerate(self.subs):
                u1 = draws[u](shapes[u])
                u2 = draws[u](shapes[u]) if t > 0 else u1
                ops.nar_sample(self._sample_args(u, sub, u1, u2), stream=st)
            ops.add_int(self.step_ptr, 1, stream=st)
        self.step_i += 1

    def _sample_args(self, u: int, sub: NARSession, u1: torch.Tensor, u2: torch.Tensor) -> L.NarSampleArgs:
        s, cfg = self.m.shape, self.cfg
        Q = s.n_codebooks
        lc = self.logits[self.row0[u]:]
        lu = self.logits[self.row0[u] + sub.s_out:] if self.nb == 2 else None
        return L.NarSampleArgs(logits_c=lc.data_ptr(), logits_u=lu.data_ptr() if lu is not None else None,
                               ld_row=(Q - 1) * self.Kp, ld_q=self.Kp, S=sub.S, n_q=Q, K=s.n_quant, row_offset=sub.row_offset,
                               x=sub.x.data_ptr(), x_known=sub.x_known.data_ptr(), m=sub.m_mask.data_ptr(),
                               u1=u1.data_ptr(), u2=u2.data_ptr(), consts=sub.consts.data_ptr(),
                               step=self.step_ptr.data_ptr(), guidance_w=cfg.guidance_w, temperature=cfg.x_0_temp,
                               log_eps=log_eps(), div_mode=cfg.div_mode, q0_override_steps=cfg.q0_override_steps)

    def _philox(self, uniforms) -> Optional[List[PhiloxDraws]]:
        """Per utterance the in-graph generator of its draws (NARSession._philox), or None if any utterance's draw is not
        generator-backed."""
        draws = [uniforms[self._order[u]] for u in range(len(self.subs))]
        if not all(getattr(d, "gen", None) is not None for d in draws) or not all(t > 0 for t in self.times[:-1]) or L.tool_knob("M5_NAR_PHILOX", "1") == "0":
            return None
        phs = self._phs
        if phs is None or any(p.gen is not d.gen for p, d in zip(phs, draws)):
            s = self.m.shape
            with torch.cuda.stream(self.stream):
                if not all(philox_matches_torch(sub.S * s.n_codebooks * s.n_quant, self.m.dev) for sub in self.subs):
                    return None
                phs = self._phs = [PhiloxDraws(d.gen, sub.S, s.n_codebooks, s.n_quant, sub.m_mask, sub.consts, self.step_ptr, self.times, self.m.dev)
                                   for d, sub in zip(draws, self.subs)]
            self.graph_step = None
        return phs

    def _step_philox(self, phs: List[PhiloxDraws], use_graph: bool) -> None:
        """One reverse step of the group as ONE launch sequence / hipGraph: the batched forward, then per utterance its uniforms
        (its own generator's stream) and its posterior / sample launch, then the step counter."""
        st = self.stream.cuda_stream

        def compose(s_):
            self.enqueue_forward(s_)
            for u, (sub, ph) in enumerate(zip(self.subs, phs)):
                ph.enqueue(s_)
                ops.nar_sample(self._sample_args(u, sub, ph.buf, ph.buf), stream=s_)
            ops.add_int(self.step_ptr, 1, stream=s_)

        def body():          # the group's step as one m5_nar_step call (NARSession._step_philox)
            if self.dl is None or not self.use_c_plan or L.tool_knob("M5_NAR_CPLAN", "1") == "0":
                return compose(st)
            if self.step_plan is None or self.step_plan[1] is not phs:
                pl = ops.StagePlan("nar_step")
                with pl.recording():
                    compose(0)
                self.step_plan = (pl, phs)
            self.step_plan[0].run(st)
        if not use_graph or (self.graph_step is None and self.step_i == 0 and len(self.times) > 1):
            body()
        else:
            if self.graph_step is None:
                ops.Graph.begin(st)
                body()
                self.graph_step = ops.Graph().end(st)
            self.graph_step.launch(st)
        self.step_i += 1

    def run(self, uniforms: List[Callable[[tuple], torch.Tensor]], use_graph: bool = True, n_steps: Optional[int] = None,
            wait: bool = True) -> Optional[List[torch.Tensor]]:
        """`wait=False`: enqueue every step and return without synchronising (the caller keeps a second group in flight on
        another stream: inference.tts_batch_from_codes); ``finish()`` then returns the results."""
        n = len(self.times) if n_steps is None else n_steps
        st = self.stream.cuda_stream
        ev0, ev1 = ops.Event(), ops.Event()
        ev0.record(st)
        phs = self._philox(uniforms)
        if phs is not None:
            for ph in phs:
                ph.reserve(self.times[self.step_i: self.step_i + n], self.step_i, self.stream)
        for _ in range(n):
            if phs is not None:
                self._step_philox(phs, use_graph)
                continue
            self.step(uniforms, use_graph)
        ev1.record(st)
        self._pending = (ev0, ev1, n)
        return self.finish() if wait else None

    def finish(self) -> List[torch.Tensor]:
        """Wait for the steps ``run`` enqueued; x (S_u, 8) per utterance, in the caller's order."""
        ev0, ev1, n = self._pending
        self.stream.synchronize()
        LAST_STATS.update(loop_ms=ev0.elapsed_ms(ev1), steps=n, S=[sub.S for sub in self.subs], s_out=[sub.s_out for sub in self.subs],
                          Le=[sub.mems[0].Le for sub in self.subs], nb=self.nb, batch=len(self.subs), rows=self.ws.M)
        out = [None] * len(self.subs)
        for u, sub in enumerate(self.subs):
            out[self._order[u]] = sub.x
        return out
