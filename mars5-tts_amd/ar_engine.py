"""AR stage on MI355X: CodecLM prefill + hipGraph-captured decode loop.

Replaces the device work of reference ``mars5/ar_generate.py:62-157`` +
``mars5/model.py:95-141`` + ``mars5/nn_future.py:235-398`` + ``mars5/samplers.py``.

What is different from the reference by design (all exact):
  * the speaker vector is computed once per utterance, not once per token;
  * only the new token is embedded per step;
  * sampling, EOS/max_len handling and the next-token embedding happen on device, with the
    position / counters in device memory, so ONE captured graph (5 launches per layer +
    head + sampler) is replayed per token with no host sync; the host polls the done flag
    every ``poll`` steps;
  * KV cache is [layer][head][slot][64] (head-major: a head's keys are contiguous, so decode
    attention streams them with fully coalesced 1 KiB wave loads), slot = pos % 3000.
"""
from __future__ import annotations

import contextlib
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops
from .blocks import SpeakerEncoder, interleave_rows, round_up
from .synth import ARShape
from .tables import eos_penalty_table, rope_table

NSPLIT = 8
LAST_STATS: dict = {}     # filled by ARSession.decode: HIP-event timings of the last utterance


@dataclass
class ARSamplingConfig:
    """Arguments of reference ``ar_generate`` (ar_generate.py:15-22) that steer sampling."""
    temperature: float = 1.0
    topk: Optional[int] = None
    top_p: float = 1.0
    alpha_frequency: float = 0.0
    alpha_presence: float = 0.0
    penalty_window: int = 100
    typical_p: float = 1.0
    eos_penalty_factor: float = 1.0
    eos_penalty_decay: float = 0.0
    n_phones_gen: Optional[int] = None
    div_mode: int = 0            # 0: logits / T (reference CPU kernel), 1: logits * (1/T) (reference GPU kernel)


# Two persistent decode steps must never share the GPU: each needs all 256 CUs for its co-resident workgroups, and two of
# them dispatched at the same time from two streams could each hold half of the CUs and spin for the other half (the spins
# are bounded, so the result would be an error, not a hang).  Sessions therefore enqueue their persistent launches under
# one per-device lock, each batch of launches behind the event that closed the previous session's batch.
_MEGA_LOCKS: Dict[int, threading.Lock] = {}
_MEGA_LOCKS_GUARD = threading.Lock()
_MEGA_LAST: Dict[int, "torch.cuda.Event"] = {}


GRAPH_GROUP = 8          # decode steps per replayed hipGraph (ARSession.capture)


@contextlib.contextmanager
def _mega_exclusive(stream: "torch.cuda.Stream", dev: torch.device):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _MEGA_LOCKS_GUARD:
        lock = _MEGA_LOCKS.setdefault(key, threading.Lock())       # one lock per device: sessions on different GPUs do not serialise
    with lock:
        last = _MEGA_LAST.get(key)
        if last is not None:
            stream.wait_event(last)
        try:
            yield
        finally:
            ev = torch.cuda.Event()
            ev.record(stream)
            _MEGA_LAST[key] = ev


def _mega_default() -> bool:
    """Persistent-layers form of the batch-1 decode step (csrc/ar_mega.hip): on unless M5_AR_MEGA=0.  It computes the same
    bits as the per-launch form (tests/test_gpu_parity16.py) and applies to the 16-bit CodecLM geometry on a device with
    at least 256 CUs; anything else runs the per-launch form."""
    return L.tool_knob("M5_AR_MEGA", "1") != "0"


class ARModel:
    """Packed CodecLM weights on one GPU.  dtype: GEMM operand type ('f32' | 'f16' | 'bf16')."""

    def __init__(self, sd: Dict[str, torch.Tensor], shape: ARShape, dtype: torch.dtype, device, max_pos: int = 8192):
        self.shape, self.dt, self.dev = shape, dtype, torch.device(device)
        dev, dt = self.dev, dtype
        Lr, D, F, V = shape.n_layers, shape.dim, shape.hidden_dim, shape.n_vocab
        assert shape.head_dim == 64 and D % 64 == 0 and F % 8 == 0

        def stack(fmt, f=lambda t: t):
            return torch.stack([f(sd[fmt.format(l)].float()) for l in range(Lr)]).to(device=dev, dtype=dt).contiguous()

        self.wqkv = torch.stack([torch.cat([sd[f"ar.layers.{l}.attention.wq.weight"], sd[f"ar.layers.{l}.attention.wk.weight"],
                                            sd[f"ar.layers.{l}.attention.wv.weight"]]).float() for l in range(Lr)]
                                ).to(device=dev, dtype=dt).contiguous()                      # (L, 3D, D)
        self.wo = stack("ar.layers.{}.attention.wo.weight")                                   # (L, D, D)
        self.w13 = torch.stack([interleave_rows(sd[f"ar.layers.{l}.feed_forward.w1.weight"].float(),
                                                sd[f"ar.layers.{l}.feed_forward.w3.weight"].float()) for l in range(Lr)]
                               ).to(device=dev, dtype=dt).contiguous()                        # (L, 2F, D)
        self.w2 = stack("ar.layers.{}.feed_forward.w2.weight")                                # (L, D, F)
        self.attn_norm = torch.stack([sd[f"ar.layers.{l}.attention_norm.weight"].float() for l in range(Lr)]).to(dev).contiguous()
        self.ffn_norm = torch.stack([sd[f"ar.layers.{l}.ffn_norm.weight"].float() for l in range(Lr)]).to(dev).contiguous()
        self.final_norm = sd["ar.norm.weight"].float().to(dev).contiguous()
        self.w_out = sd["ar.output.weight"].to(device=dev, dtype=dt).contiguous()             # (V, D)
        self.embed = sd["embed.weight"].float().to(dev).contiguous()                          # (V, D) fp32
        self.rope = rope_table(64, max_pos).to(dev)
        self.max_pos = max_pos
        self.spk = SpeakerEncoder(sd, "ref_chunked_emb", "pos_embedding.alpha", shape.n_spk_layers, dt, dev)

    def weight_bytes_per_token(self) -> int:
        """Algorithmic HBM bytes one decode step must stream."""
        es = torch.tensor([], dtype=self.dt).element_size()
        n = self.wqkv.numel() + self.wo.numel() + self.w13.numel() + self.w2.numel() + self.w_out.numel()
        return n * es + (self.attn_norm.numel() + self.ffn_norm.numel() + self.final_norm.numel()) * 4


class ARSession:
    """One utterance: prefill + decode.  Owns KV cache and step buffers (caller = this host code)."""

    def __init__(self, model: ARModel, max_len: int, stream: Optional[torch.cuda.Stream] = None,
                 w_alloc: Optional[int] = None, buffers: Optional[Dict[str, torch.Tensor]] = None):
        """`buffers` (kc, vc, xdec, logits, state, tokens): views into a batch session's arrays, so the same
        prefill code fills sequence b of a batch; `w_alloc` then is the batch's common cache allocation."""
        self.m = model
        s, dev, dt = model.shape, model.dev, model.dt
        self.max_len = max_len
        self.window = s.sliding_window
        self.w_alloc = min(s.sliding_window, max_len + 1) if w_alloc is None else w_alloc
        assert max_len + 1 <= model.max_pos
        H, D, F, V = s.nhead, s.dim, s.hidden_dim, s.n_vocab
        bf = buffers or {}
        self.stream = stream if stream is not None else ops.session_stream(dev, "ar")
        # the buffers are zero-filled ON the session stream (behind the caller's pending work): a memset left on the
        # caller's stream could otherwise land after this stream has started writing state / KV rows
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            self.kc = bf["kc"] if "kc" in bf else torch.zeros(s.n_layers, H, self.w_alloc, 64, dtype=dt, device=dev)
            self.vc = bf["vc"] if "vc" in bf else torch.zeros(s.n_layers, H, self.w_alloc, 64, dtype=dt, device=dev)
            self.xdec = bf["xdec"] if "xdec" in bf else torch.zeros(D, dtype=torch.float32, device=dev)
            self.qbuf = torch.zeros(D, dtype=dt, device=dev)
            self.hbuf = torch.zeros(F, dtype=dt, device=dev)
            self.part = torch.zeros(H, NSPLIT, L.ATTN_PART, dtype=torch.float32, device=dev)
            self.logits = bf["logits"] if "logits" in bf else torch.zeros(V, dtype=torch.float32, device=dev)
            self.state = bf["state"] if "state" in bf else torch.zeros(L.ST_WORDS, dtype=torch.int32, device=dev)
            self.tokens = bf["tokens"] if "tokens" in bf else torch.zeros(max_len + 1, dtype=torch.int64, device=dev)
            # persistent-layers form of the decode step (csrc/ar_mega.hip): granule buffer + sticky error word
            self.gran = torch.zeros(L.AR_MEGA_GRANULES, dtype=torch.int64, device=dev)
            self.mega_err = torch.zeros(4, dtype=torch.int32, device=dev)
        self.mega_dbg: Optional[torch.Tensor] = None           # tools/ar_mega_clock.py: (32, 16, 8) int64 phase stamps
        self.mega = _mega_default() and model.dt != torch.float32 and (D, F, H) == (1536, 3584, 24) and not buffers
        self.graph: Optional[ops.Graph] = None
        self.graph_group: Optional[ops.Graph] = None           # GRAPH_GROUP consecutive steps as one graph (capture)
        self.group = 1
        self.step_plan: Optional[ops.StagePlan] = None         # the recorded decode step (enqueue_step)
        self.use_c_plan = True                                 # False: the step's launches are composed in this file (tests compare the two)
        self._sample_args: Optional[L.SampleArgs] = None
        self._keep: List[torch.Tensor] = []
        self.ended_on_eos = False

    # ------------------------------------------------------------------ prefill
    def prefill(self, prompt: torch.Tensor, ref_codes: torch.Tensor, spk_vec: Optional[torch.Tensor] = None) -> None:
        """prompt (P,) int64 global ids; ref_codes (Lc, 8) int64.  Runs the speaker encoder and
        the 26-layer stack over [spk_vec, tok_0..tok_{P-1}] (positions 0..P), filling the cache
        and leaving the last row's residual in ``xdec``."""
        m, s = self.m, self.m.shape
        dev, dt = m.dev, m.dt
        st = self.stream.cuda_stream
        P = int(prompt.shape[0])
        M = P + 1
        assert M <= self.window, "prefill longer than the sliding window is not supported"
        H, D, F = s.nhead, s.dim, s.hidden_dim
        self.stream.wait_stream(torch.cuda.current_stream(dev))       # prompt / ref_codes may have been produced there
        with torch.cuda.stream(self.stream):
            prompt = ops.use_on(prompt.to(dev), self.stream)
            ref_codes = ops.use_on(ref_codes.to(dev).contiguous(), self.stream)
            ops.use_on(spk_vec, self.stream)
            # Every host -> device copy of this function happens HERE, in front of the launches: a pageable copy blocks the host
            # until the stream has executed it, i.e. until everything enqueued before it is done (the state words used to go last:
            # the host then sat out the whole prefill before it could enqueue the first decode step).
            self.tokens[:P].copy_(prompt)
            self.state.copy_(torch.tensor([P, 0, 0, P, -1, 0, 0, 0], dtype=torch.int32, device="cpu"), non_blocking=False)
            # (D,) fp32; a cached vector of the same reference (Mars5TTS.prepare_reference) is the same tensor by construction
            spk = m.spk(ref_codes, stream=st) if spk_vec is None else spk_vec.to(dev)
            table = torch.cat([m.embed, spk[None]], dim=0)                       # plumbing: one extra row
            idx = torch.cat([torch.full((1,), s.n_vocab, device=dev, dtype=torch.int64), prompt])      # (a device-side fill: no host copy)
            x = torch.empty(M, D, dtype=torch.float32, device=dev)
            ops.gather_rows(x, table, idx, stream=st)
            Mp = round_up(M, 64)
            xn = torch.empty(M, D, dtype=dt, device=dev)
            qkv = torch.empty(M, 3 * D, dtype=dt, device=dev)
            q = torch.empty(H, M, 64, dtype=dt, device=dev)
            vt = torch.zeros(H, 64, Mp, dtype=dt, device=dev)
            att = torch.empty(M, D, dtype=dt, device=dev)
            hb = torch.empty(M, F, dtype=dt, device=dev)
            for l in range(s.n_layers):
                ops.rmsnorm(x, m.attn_norm[l], s.norm_eps, xn, stream=st)
                ops.gemm(xn, m.wqkv[l], qkv, L.EPI_DT, stream=st)
                ops.rope_cache(qkv, H, 0, m.rope, q, self.kc[l], self.vc[l], self.w_alloc * 64, self.window, vt, 64 * Mp, Mp, stream=st)
                a = L.AttnArgs(q=q.data_ptr(), q_bs=0, q_hs=M * 64, q_rs=64,
                               k=self.kc[l].data_ptr(), k_bs=0, k_hs=self.w_alloc * 64, k_rs=64,
                               vt=vt.data_ptr(), vt_bs=0, vt_hs=64 * Mp, vt_ds=Mp,
                               o=att.data_ptr(), o_bs=0, o_rs=D, B=1, H=H, Sq=M, Sk=M, key_len=None, causal=1,
                               scale=64 ** -0.5, kv_index=None, kv_index_stride_k=0, kv_index_stride_v=0)
                ops.attention(dt, a, stream=st)
                ops.gemm(att, m.wo[l], x, L.EPI_RESIDUAL, stream=st)
                ops.rmsnorm(x, m.ffn_norm[l], s.norm_eps, xn, stream=st)
                ops.gemm(xn, m.w13[l], hb, L.EPI_SWIGLU, stream=st)
                ops.gemm(hb, m.w2[l], x, L.EPI_RESIDUAL, stream=st)
            self.xdec.copy_(x[M - 1])
            self.gran.zero_()                                  # granule tags are unique within one utterance only
            self.mega_err.zero_()
            self._keep = [table, x]
        self.P = P

    # ------------------------------------------------------------------ decode
    def _gemv_args(self, **kw) -> L.GemvArgs:
        a = L.GemvArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return a

    def enqueue_head_and_sample(self, st: int) -> None:
        m, s = self.m, self.m.shape
        a = self._gemv_args(W=m.w_out, ldw=s.dim, N=s.n_vocab, K=s.dim, x_f32=self.xdec, norm_w=m.final_norm, eps=s.norm_eps,
                            y_f32=self.logits, state=self.state)
        ops.ar_gemv(m.dt, L.PRO_RMS, L.GEPI_F32, a, stream=st)
        ops.ar_sample(self._sample_args, stream=st)

    @staticmethod
    def _pf(w: Optional[torch.Tensor], wgs: int, first: int = 0, n: int = 256) -> L.Prefetch:
        """Prefetch descriptor for weight matrix `w` as the streaming GEMV will read it: 256 consuming workgroups, each a
        contiguous block of rows (include/mars5_hip.h, M5Prefetch).  None / wgs 0 = off."""
        if w is None or wgs <= 0:
            return L.Prefetch()
        chunk = w.numel() * w.element_size() // 256
        if chunk % 64 or w.numel() * w.element_size() % 256:
            return L.Prefetch()
        return L.Prefetch(ptr=w.data_ptr(), chunk_bytes=chunk, first_chunk=first, n_chunks=n, wgs=wgs)

    def enqueue_layers(self, st: int) -> None:
        m, s = self.m, self.m.shape
        D, F, H = s.dim, s.hidden_dim, s.nhead
        if self.mega:
            a = L.ArMegaArgs(wqkv=m.wqkv.data_ptr(), wo=m.wo.data_ptr(), w13=m.w13.data_ptr(), w2=m.w2.data_ptr(),
                             attn_norm=m.attn_norm.data_ptr(), ffn_norm=m.ffn_norm.data_ptr(), eps=s.norm_eps,
                             dim=D, hidden=F, n_heads=H, layer0=0, layer1=s.n_layers, xres=self.xdec.data_ptr(),
                             rope=m.rope.data_ptr(), state=self.state.data_ptr(), kcache=self.kc.data_ptr(), vcache=self.vc.data_ptr(),
                             w_alloc=self.w_alloc, window=self.window, scale=64 ** -0.5, gran=self.gran.data_ptr(),
                             err=self.mega_err.data_ptr(), dbg=self.mega_dbg.data_ptr() if self.mega_dbg is not None else None)
            if ops.ar_layers_persistent(m.dt, a, stream=st) == L.M5_OK:
                return
            self.mega = False                                  # geometry / device not eligible: per-launch form from here on
        # same-stream prefetch plan (M5Prefetch): 0 off; 1 every launch pulls the NEXT launch's weights; 2 only the two
        # bandwidth-idle launches (cache scan, Wo) pull the two halves of W1|W3.  16-bit streaming geometry only.
        plan = int(L.tool_knob("M5_AR_PREFETCH", "0")) if (m.dt != torch.float32 and D == 1536 and F == 3584) else 0
        npf = int(L.tool_knob("M5_AR_PREFETCH_WGS", "64"))
        for l in range(s.n_layers):
            nxt = m.wqkv[l + 1] if l + 1 < s.n_layers else m.w_out
            a = self._gemv_args(W=m.wqkv[l], ldw=D, N=3 * D, K=D, x_f32=self.xdec, norm_w=m.attn_norm[l], eps=s.norm_eps,
                                rope=m.rope, state=self.state, kcache=self.kc[l], vcache=self.vc[l], qbuf=self.qbuf,
                                w_alloc=self.w_alloc, window=self.window, dim=D, pf=self._pf(m.wo[l] if plan == 1 else None, npf))
            ops.ar_gemv(m.dt, L.PRO_RMS, L.GEPI_QKV_ROPE, a, stream=st)
            d = L.AttnDecodeArgs(qbuf=self.qbuf.data_ptr(), kcache=self.kc[l].data_ptr(), vcache=self.vc[l].data_ptr(),
                                 part=self.part.data_ptr(), state=self.state.data_ptr(), n_heads=H, w_alloc=self.w_alloc,
                                 window=self.window, nsplit=NSPLIT, scale=64 ** -0.5,
                                 pf=self._pf(m.w13[l] if plan else None, (npf + H - 1) // H * H, 0, 128))
            ops.ar_attn_decode(m.dt, d, stream=st)
            a = self._gemv_args(W=m.wo[l], ldw=D, N=D, K=D, part=self.part, nsplit=NSPLIT, n_heads=H, xres=self.xdec, state=self.state,
                                pf=self._pf(m.w13[l] if plan else None, npf, 128, 128))
            ops.ar_gemv(m.dt, L.PRO_ATTN, L.GEPI_RESIDUAL, a, stream=st)
            a = self._gemv_args(W=m.w13[l], ldw=D, N=2 * F, K=D, x_f32=self.xdec, norm_w=m.ffn_norm[l], eps=s.norm_eps,
                                y_dt=self.hbuf, state=self.state, pf=self._pf(m.w2[l] if plan == 1 else None, npf))
            ops.ar_gemv(m.dt, L.PRO_RMS, L.GEPI_SWIGLU, a, stream=st)
            a = self._gemv_args(W=m.w2[l], ldw=F, N=D, K=F, x_dt=self.hbuf, xres=self.xdec, state=self.state,
                                pf=self._pf(nxt if plan == 1 else None, npf))
            ops.ar_gemv(m.dt, L.PRO_DT, L.GEPI_RESIDUAL, a, stream=st)

    def configure_sampler(self, cfg: ARSamplingConfig, n_text: int, eos_idx: int, noise: Optional[torch.Tensor], rng: Optional[tuple] = None,
                          n_steps: Optional[int] = None) -> None:
        """noise: (n_steps, V) fp32 device tensor of Exp(1) draws (one row per sampler call) -- or None with
        rng = (seed, offset0, generator offset per draw, torch's launch width for V values): the sampler then generates the row
        of call i itself, bit-identical to the i-th ``torch.empty(V).exponential_(1)`` of a generator that stood at (seed, offset0)
        (include/mars5_hip.h, M5SampleArgs.rng): no noise buffer, no ATen launch in the decode loop."""
        m, s = self.m, self.m.shape
        self._rng = None
        if noise is None:
            assert rng is not None and n_steps is not None
            seed, off0, inc, grid = rng
            wrap = lambda v: v - (1 << 64) if v >= (1 << 63) else v                 # noqa: E731
            with torch.cuda.stream(self.stream):
                self._rng = torch.tensor([wrap(int(seed) & 0xFFFFFFFFFFFFFFFF), int(off0)], dtype=torch.int64).to(m.dev)
        eos_tab = None
        n_est = 0
        if cfg.n_phones_gen is not None:
            n_est = int(cfg.n_phones_gen)
            eos_tab = eos_penalty_table(n_est, cfg.eos_penalty_decay, cfg.eos_penalty_factor).to(m.dev)
        self._eos_tab, self._noise = eos_tab, noise
        assert noise is None or (noise.dtype == torch.float32 and noise.shape[1] == s.n_vocab and noise.is_contiguous())
        a = L.SampleArgs(logits=self.logits.data_ptr(), V=s.n_vocab, state=self.state.data_ptr(), tokens=self.tokens.data_ptr(),
                         max_len=self.max_len, alpha_frequency=cfg.alpha_frequency, alpha_presence=cfg.alpha_presence,
                         penalty_window=cfg.penalty_window, n_text=n_text, eos_idx=eos_idx, n_est=n_est,
                         eos_table=eos_tab.data_ptr() if eos_tab is not None else None, temperature=cfg.temperature,
                         div_mode=cfg.div_mode, top_k=int(cfg.topk or 0), top_p=cfg.top_p, typical_p=float(cfg.typical_p),
                         noise=noise.data_ptr() if noise is not None else None,
                         noise_stride=s.n_vocab, embed=m.embed.data_ptr(), dim=s.dim, xres=self.xdec.data_ptr(),
                         rng=self._rng.data_ptr() if self._rng is not None else None, rng_bs=0,
                         noise_inc=int(rng[2]) if noise is None else 0, noise_grid=int(rng[3]) if noise is None else 0)
        self._sample_args = a
        self.step_plan = None                                    # (the recorded step holds a copy of the sampler arguments)
        self.n_noise = noise.shape[0] if noise is not None else int(n_steps)

    def enqueue_step(self, st: int) -> None:
        """One decode step (layers + head + sampler) through ONE C call (include/mars5_hip.h m5_ar_decode_step): the launches
        are recorded once per session and sampler configuration into a stage plan (positions, counters and RNG state live in
        device memory) and composed by the library from then on.  (M5_AR_CPLAN=0: tools A/B knob, composition stays here.)"""
        if not self.use_c_plan or L.tool_knob("M5_AR_CPLAN", "1") == "0":
            self.enqueue_layers(st)
            self.enqueue_head_and_sample(st)
            return
        for _ in range(2):
            if self.step_plan is None:
                pl = ops.StagePlan("ar_decode_step")
                with pl.recording():
                    self.enqueue_layers(0)
                    self.enqueue_head_and_sample(0)
                self.step_plan = pl
            rc = self.step_plan.run(st, raise_on_error=False)
            if rc == L.M5_OK:
                return
            if rc == L.M5_ERR_UNSUPPORTED and self.mega and int(self.step_plan._failed.value) == 0:
                # the persistent layer kernel declined (device / geometry): nothing was enqueued; per-launch form from here on
                self.mega, self.step_plan = False, None
                continue
            L.check(rc, f"m5_ar_decode_step (op {int(self.step_plan._failed.value)})")

    def capture(self) -> None:
        """Capture one decode step (layers + head + sampler) as a hipGraph, and GRAPH_GROUP consecutive steps as a second one:
        positions, counters and RNG state live in device memory, so a graph of k steps IS k replays of the one-step graph --
        without the ~10 us the GPU idles between two graph launches (profiles/r6ae_utterance_gpu_idle_gaps.txt: 5 ms per
        utterance).  A finished sequence's steps return at once (state[DONE]), as they do between two polls."""
        st = self.stream.cuda_stream
        self.stream.synchronize()
        ops.Graph.begin(st)
        self.enqueue_step(st)
        self.graph = ops.Graph().end(st)
        self.graph_group, self.group = None, int(L.tool_knob("M5_AR_GROUP", str(GRAPH_GROUP)))      # A/B knob (tools/ar_step_bench.py)
        if self.group > 1 and self.mega:                        # (the per-launch form is 133 launches per step: one step per graph)
            ops.Graph.begin(st)
            for _ in range(self.group):
                self.enqueue_step(st)
            self.graph_group = ops.Graph().end(st)

    def _launch_steps(self, n: int, use_graph: bool, st: int) -> None:
        """Enqueue n decode steps (persistent launches under the per-device exclusive lock)."""
        with (_mega_exclusive(self.stream, self.m.dev) if self.mega else contextlib.nullcontext()):
            if use_graph and self.graph_group is not None:
                while n >= self.group:
                    self.graph_group.launch(st)
                    n -= self.group
            for _ in range(n):
                if use_graph:
                    self.graph.launch(st)
                else:
                    self.enqueue_step(st)

    def decode(self, use_graph: bool = True, poll: int = 32, noise_fill=None) -> torch.Tensor:
        """Run sampler for the prefill logits, then decode steps until EOS / max_len.
        Returns the token sequence (prompt + generated) like ``ar_generate`` (EOS not appended).
        `noise_fill(lo, hi)`: called (host side, before the launches that read them) to draw noise rows lo..hi-1
        on this session's stream -- sampler call i reads row i -- so an utterance that stops early never pays for the
        rows of the steps it does not run; None = the noise tensor is already complete."""
        st = self.stream.cuda_stream
        if self.P >= self.max_len:
            return self.tokens[: self.P].clone()
        filled = 0

        def need(hi):
            nonlocal filled
            hi = min(hi, self.n_noise)
            if noise_fill is not None and hi > filled:
                filled = noise_fill(filled, hi) or hi

        need(1)
        self.enqueue_head_and_sample(st)                       # token P from the prefill's last row
        budget = min(self.max_len - self.P - 1, self.n_noise - 1)
        if use_graph and self.graph is None and budget > 0:
            self.capture()
        ev0, ev1 = ops.Event(), ops.Event()
        ev0.record(st)
        done = 0
        # Recovery point of the persistent form: (steps done, state words, residual-stream input of the next step) at the last
        # poll whose error word was clean.  If a persistent launch gives up (grid not co-resident: a foreign workgroup held a
        # CU past the spin bound) the error word is sticky, the remaining launches of the batch return at once and the sampler
        # runs on a stale residual stream; the host then restores this point, switches the session to the per-launch form
        # (bit-identical arithmetic, tests/test_gpu_parity16.py) and replays from there -- the request is never lost.
        snap = None
        kv_snap = None            # past the wrap of the rotating cache: the rows the next batch overwrites (see below)
        self.mega_recovered = 0
        pos_host = self.P         # the host's copy of state[ST_POS] at the last poll (the cache row the next launch writes, or the one before)
        if self.mega:
            with torch.cuda.stream(self.stream):
                snap = (0, self.state.clone(), self.xdec.clone())
        while done < budget:
            n = min(poll, budget - done)
            need(done + n + 1)                                 # the n steps below read rows done+1 .. done+n
            if self.mega and self.w_alloc == self.window and pos_host + n + 1 > self.window:
                # The cache has wrapped (or wraps inside this batch): the batch overwrites rows that still hold positions
                # pos - window .., which a replay from the recovery point attends to.  Keep those rows (<= poll per layer and
                # head, a few MB) so that a recovery restores the cache too, not only the state words and the residual stream.
                with torch.cuda.stream(self.stream):
                    slots = torch.tensor(sorted({(pos_host + i) % self.window for i in range(n + 2)}), dtype=torch.long, device="cpu").to(self.m.dev)
                    kv_snap = (slots, self.kc.index_select(2, slots), self.vc.index_select(2, slots))
            else:
                kv_snap = None
            self._launch_steps(n, use_graph, st)
            done += n
            with torch.cuda.stream(self.stream):
                flag = self.state.cpu()                        # syncs this stream only
                bad = bool(int(self.mega_err.cpu()[0])) if self.mega else False
            if bad:
                done0, st0, x0 = snap
                with torch.cuda.stream(self.stream):
                    self.state.copy_(st0)
                    self.xdec.copy_(x0)
                    self.mega_err.zero_()
                    if kv_snap is not None:
                        self.kc.index_copy_(2, kv_snap[0], kv_snap[1])
                        self.vc.index_copy_(2, kv_snap[0], kv_snap[2])
                self.mega, self.step_plan = False, None       # (the recorded step held the persistent launch)
                self.mega_recovered += 1
                if use_graph:
                    self.capture()                             # the per-launch form of the step
                done = done0
                continue
            pos_host = int(flag[L.ST_POS])
            if self.mega:
                with torch.cuda.stream(self.stream):
                    snap = (done, self.state.clone(), self.xdec.clone())
            if int(flag[L.ST_DONE]):
                break
        ev1.record(st)
        self.stream.synchronize()
        final = self.state.cpu()
        n_tok = int(final[L.ST_NTOK])
        self.ended_on_eos = bool(int(final[L.ST_DONE])) and int(final[L.ST_LAST]) == int(self._sample_args.eos_idx)
        LAST_STATS.update(decode_ms=ev0.elapsed_ms(ev1), decode_steps_launched=done, n_generated=n_tok - self.P,
                          prefill_len=self.P + 1, final_len=n_tok, persistent=bool(self.mega), persistent_recoveries=self.mega_recovered)
        return self.tokens[:n_tok].clone()


class ARBatchSession:
    """B independent sequences decoded together (BASELINE config 3).  The decode step is HBM-bound on the
    1.36 GB of weights, so advancing B sequences per step reads them once for all: the projections become
    M = B row GEMMs (``m5_gemm``'s M <= 32 weight-streaming path), while everything per sequence -- RoPE
    position, KV cache, cache scan, sampler chain, EOS / max_len, RNG rows -- keeps its own device state,
    so sequence b produces what it would produce alone up to the summation order of the GEMMs.  Prefill
    runs sequence by sequence through ``ARSession.prefill`` on views of the batch arrays.  One captured
    hipGraph (8 launches per layer + head + sampler) serves every step; finished sequences idle."""

    def __init__(self, model: ARModel, max_lens: List[int], stream: Optional[torch.cuda.Stream] = None):
        self.m = model
        s, dev, dt = model.shape, model.dev, model.dt
        self.B = B = len(max_lens)
        assert 1 <= B <= 32, "the skinny GEMM path covers up to 32 sequences per step"
        self.max_lens = list(max_lens)
        self.window = s.sliding_window
        self.w_alloc = min(s.sliding_window, max(max_lens) + 1)
        assert max(max_lens) + 1 <= model.max_pos
        H, D, F, V, Lr = s.nhead, s.dim, s.hidden_dim, s.n_vocab, s.n_layers
        self.stream = stream if stream is not None else ops.session_stream(dev, "ar")
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        # key-range splits of the cache scan: enough workgroups to fill the chip, no more (each costs a partial + merge)
        self.nsplit = max(1, min(NSPLIT, 1 << max(0, (1024 // (H * B)).bit_length() - 1)))
        with torch.cuda.stream(self.stream):                           # zero-fills ordered on the session stream (see ARSession)
            self.kc = torch.zeros(B, Lr, H, self.w_alloc, 64, dtype=dt, device=dev)
            self.vc = torch.zeros(B, Lr, H, self.w_alloc, 64, dtype=dt, device=dev)
            self.x = torch.zeros(B, D, dtype=torch.float32, device=dev)
            self.xn = torch.zeros(B, D, dtype=dt, device=dev)
            self.qkv = torch.zeros(B, 3 * D, dtype=dt, device=dev)
            self.qbuf = torch.zeros(B, D, dtype=dt, device=dev)
            self.att = torch.zeros(B, D, dtype=dt, device=dev)
            self.hbuf = torch.zeros(B, F, dtype=dt, device=dev)
            self.part = torch.zeros(B, H, self.nsplit, L.ATTN_PART, dtype=torch.float32, device=dev)
            self.logits = torch.zeros(B, V, dtype=torch.float32, device=dev)
            self.state = torch.zeros(B, L.ST_WORDS, dtype=torch.int32, device=dev)
            self.tokens = torch.zeros(B, max(max_lens) + 1, dtype=torch.int64, device=dev)
        self.subs = [ARSession(model, max_lens[b], self.stream, w_alloc=self.w_alloc,
                               buffers=dict(kc=self.kc[b], vc=self.vc[b], xdec=self.x[b], logits=self.logits[b], state=self.state[b],
                                            tokens=self.tokens[b])) for b in range(B)]
        self.graph: Optional[ops.Graph] = None
        self.P: List[int] = [0] * B

    def prefill(self, prompts: List[torch.Tensor], ref_codes: List[torch.Tensor]) -> None:
        for b, sub in enumerate(self.subs):
            sub.prefill(prompts[b], ref_codes[b])
            self.P[b] = sub.P

    def configure_sampler(self, cfg: ARSamplingConfig, n_text: int, eos_idx: int, noise: Optional[torch.Tensor],
                          n_phones_gen: Optional[List[Optional[int]]] = None, rng: Optional[tuple] = None, n_steps: Optional[int] = None) -> None:
        """noise (B, n_steps, V) fp32 device: row [b][i] feeds sequence b's i-th sampler call -- or None with
        rng = ([(seed_b, offset0_b)], generator offset per draw, torch's launch width for V values): every sequence's sampler
        generates its own Exp(1) values (ARSession.configure_sampler).
        n_phones_gen: per-sequence EOS-penalty length estimates (``cfg.n_phones_gen`` for all when None)."""
        m, s, B = self.m, self.m.shape, self.B
        rng_dev = None
        if noise is None:
            assert rng is not None and n_steps is not None and len(rng[0]) == B
            wrap = lambda v: v - (1 << 64) if v >= (1 << 63) else v                 # noqa: E731
            with torch.cuda.stream(self.stream):
                rng_dev = torch.tensor([[wrap(int(sd) & 0xFFFFFFFFFFFFFFFF), int(of)] for sd, of in rng[0]], dtype=torch.int64).to(m.dev)
        else:
            assert noise.dtype == torch.float32 and noise.shape[0] == B and noise.shape[2] == s.n_vocab and noise.is_contiguous()
        ests = n_phones_gen if n_phones_gen is not None else [cfg.n_phones_gen] * B
        eos_tab = n_est_b = None
        if any(e is not None for e in ests):
            assert all(e is not None for e in ests)
            n_max = max(int(e) for e in ests)
            eos_tab = torch.zeros(B, n_max + 1, dtype=torch.float32, device="cpu")
            for b, e in enumerate(ests):
                eos_tab[b, : int(e) + 1] = eos_penalty_table(int(e), cfg.eos_penalty_decay, cfg.eos_penalty_factor)
            eos_tab = eos_tab.to(m.dev)
            n_est_b = torch.tensor([int(e) for e in ests], dtype=torch.int32, device=m.dev)
        max_len_b = torch.tensor(self.max_lens, dtype=torch.int32, device=m.dev)
        self._keep = [eos_tab, n_est_b, max_len_b, noise, rng_dev]
        self.eos_idx = eos_idx
        self.n_noise = noise.shape[1] if noise is not None else int(n_steps)
        self._sample_args = L.SampleArgs(
            logits=self.logits.data_ptr(), V=s.n_vocab, state=self.state.data_ptr(), tokens=self.tokens.data_ptr(), max_len=max(self.max_lens),
            alpha_frequency=cfg.alpha_frequency, alpha_presence=cfg.alpha_presence, penalty_window=cfg.penalty_window, n_text=n_text,
            eos_idx=eos_idx, n_est=0, eos_table=eos_tab.data_ptr() if eos_tab is not None else None, temperature=cfg.temperature,
            div_mode=cfg.div_mode, top_k=int(cfg.topk or 0), top_p=cfg.top_p, typical_p=float(cfg.typical_p),
            noise=noise.data_ptr() if noise is not None else None,
            noise_stride=s.n_vocab, embed=m.embed.data_ptr(), dim=s.dim, xres=self.x.data_ptr(),
            batch=B, state_bs=L.ST_WORDS, logits_bs=s.n_vocab, tokens_bs=self.tokens.stride(0), noise_bs=noise.stride(0) if noise is not None else 0, xres_bs=s.dim,
            rng=rng_dev.data_ptr() if rng_dev is not None else None, rng_bs=2, noise_inc=int(rng[1]) if rng_dev is not None else 0,
            noise_grid=int(rng[2]) if rng_dev is not None else 0,
            eos_table_bs=eos_tab.stride(0) if eos_tab is not None else 0,
            n_est_b=n_est_b.data_ptr() if n_est_b is not None else None, max_len_b=max_len_b.data_ptr())

    def enqueue_head_and_sample(self, st: int) -> None:
        m, s = self.m, self.m.shape
        ops.rmsnorm(self.x, m.final_norm, s.norm_eps, self.xn, stream=st)
        ops.gemm(self.xn, m.w_out, self.logits, L.EPI_F32, stream=st)
        ops.ar_sample(self._sample_args, stream=st)

    def enqueue_layers(self, st: int) -> None:
        m, s, B = self.m, self.m.shape, self.B
        D, H = s.dim, s.nhead
        cache_hs = self.w_alloc * 64
        cache_bs = s.n_layers * H * cache_hs
        for l in range(s.n_layers):
            ops.rmsnorm(self.x, m.attn_norm[l], s.norm_eps, self.xn, stream=st)
            ops.ar_qkv_rope_batch(self.xn, m.wqkv[l], H, m.rope, self.state, self.qbuf, self.kc[0, l], self.vc[0, l], cache_bs, cache_hs,
                                  self.window, self.qkv, stream=st)
            d = L.AttnDecodeArgs(qbuf=self.qbuf.data_ptr(), kcache=self.kc[0, l].data_ptr(), vcache=self.vc[0, l].data_ptr(),
                                 part=self.part.data_ptr(), state=self.state.data_ptr(), n_heads=H, w_alloc=self.w_alloc,
                                 window=self.window, nsplit=self.nsplit, scale=64 ** -0.5, batch=B, state_bs=L.ST_WORDS, q_bs=D,
                                 cache_bs=cache_bs, part_bs=self.part.stride(0))
            ops.ar_attn_decode(m.dt, d, stream=st)
            ops.ar_attn_combine_batch(self.part, H, self.nsplit, self.state, self.att, stream=st)
            ops.gemm(self.att, m.wo[l], self.x, L.EPI_RESIDUAL, stream=st)
            ops.rmsnorm(self.x, m.ffn_norm[l], s.norm_eps, self.xn, stream=st)
            ops.gemm(self.xn, m.w13[l], self.hbuf, L.EPI_SWIGLU, stream=st)
            ops.gemm(self.hbuf, m.w2[l], self.x, L.EPI_RESIDUAL, stream=st)

    def decode(self, use_graph: bool = True, poll: int = 32, noise_fill=None) -> List[torch.Tensor]:
        """Sampler for the prefill logits of every sequence, then batched steps until every sequence has hit
        EOS or its max_len.  Returns the B token sequences (prompt + generated, EOS not appended).
        `noise_fill(lo, hi)`: draws noise rows lo..hi-1 of every sequence before the launches that read them (see ARSession)."""
        st = self.stream.cuda_stream
        filled = 0

        def need(hi):
            nonlocal filled
            hi = min(hi, self.n_noise)
            if noise_fill is not None and hi > filled:
                filled = noise_fill(filled, hi) or hi

        need(1)
        self.enqueue_head_and_sample(st)
        budget = min(max(ml - p - 1 for ml, p in zip(self.max_lens, self.P)), self.n_noise - 1)
        if use_graph and self.graph is None and budget > 0:
            self.stream.synchronize()
            ops.Graph.begin(st)
            self.enqueue_layers(st)
            self.enqueue_head_and_sample(st)
            self.graph = ops.Graph().end(st)
        ev0, ev1 = ops.Event(), ops.Event()
        ev0.record(st)
        done = 0
        while done < budget:
            n = min(poll, budget - done)
            need(done + n + 1)
            for _ in range(n):
                if use_graph:
                    self.graph.launch(st)
                else:
                    self.enqueue_layers(st)
                    self.enqueue_head_and_sample(st)
            done += n
            with torch.cuda.stream(self.stream):
                flags = self.state.cpu()                       # syncs this stream only
            if bool((flags[:, L.ST_DONE] != 0).all()):
                break
        ev1.record(st)
        self.stream.synchronize()
        final = self.state.cpu()
        outs = []
        self.ended_on_eos = []
        for b in range(self.B):
            n_tok = int(final[b, L.ST_NTOK])
            outs.append(self.tokens[b, :n_tok].clone())
            self.ended_on_eos.append(bool(int(final[b, L.ST_DONE])) and int(final[b, L.ST_LAST]) == int(self.eos_idx))
        LAST_STATS.update(decode_ms=ev0.elapsed_ms(ev1), decode_steps_launched=done, batch=self.B,
                          n_generated=[int(final[b, L.ST_NTOK]) - self.P[b] for b in range(self.B)])
        return outs
