"""Drop-in for the inference part of reference ``mars5/diffuser.py``: ``MultinomialDiffusion``
(schedule tables, :62-95), ``DSH`` (:302-315), ``get_schedule`` (:318-333) and
``perform_simple_inference`` (:398-472).  The reverse-diffusion loop runs on the MI355X
engine (``nar_engine``).  Training-only pieces (compute_Lt, kl_prior, ...) and the RePaint
forward-jump branch (dead at the shipped jump_len = jump_n_sample = 1) are out of scope."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import torch
from torch import Tensor

from .nar_engine import NARBatchSession, NARConfig, NARSession
from .tables import diffusion_log_tables


def _table_key(tensors):
    """Identity of the four schedule tensors: object, storage and the in-place version counter.  None when torch tracks no
    version for one of them (tensors made under inference_mode): an in-place edit could then not be seen, so the caller
    must not trust a host copy made earlier (None never compares equal to a key below)."""
    key = []
    for t in tensors:
        try:
            key.append((id(t), t.data_ptr(), t._version))
        except RuntimeError:
            return None
    return tuple(key)


class MultinomialDiffusion:
    def __init__(self, num_classes, timesteps=100, diffusion_s=0.008, loss_type='vb_stochastic', parametrization='x0',
                 dtype=torch.float32, device='cpu'):
        assert loss_type in ('vb_stochastic',)
        assert parametrization in ('x0', 'direct')
        self.num_classes = num_classes
        self.loss_type = loss_type
        self.num_timesteps = timesteps
        self.parametrization = parametrization
        la, l1ma, lca, l1mca = diffusion_log_tables(timesteps, diffusion_s)
        self.log_alpha = la.to(dtype).to(device)
        self.log_1_min_alpha = l1ma.to(dtype).to(device)
        self.log_cumprod_alpha = lca.to(dtype).to(device)
        self.log_1_min_cumprod_alpha = l1mca.to(dtype).to(device)
        # host copies of the four tables as they were made: the engine derives its per-step constants on the host, and reading
        # the device tensors back cost four synchronising copies (6 ms) in front of every utterance's AR decode.  Used only
        # while the attributes are still the tensors made here, unmodified (``_tables`` checks identity and version counters).
        self._host_tables = tuple(t.to(dtype).to("cpu") for t in (la, l1ma, lca, l1mca))
        self._host_key = _table_key((self.log_alpha, self.log_1_min_alpha, self.log_cumprod_alpha, self.log_1_min_cumprod_alpha))


@dataclass
class DSH():
    jump_len: int = 1
    jump_n_sample: int = 1
    last_greedy: bool = False          # never forwarded by the reference loop (diffuser.py:461)
    x_0_temp: float = 1.0
    guidance_w: float = 1.0
    enable_kevin_scaled_inference: bool = True
    T_override: Union[None, int] = None
    deep_clone: bool = False
    q0_override_steps: int = 0
    progress: bool = False


def get_schedule(t_T, jump_len=10, jump_n_sample=10):
    jumps = {j: jump_n_sample - 1 for j in range(0, t_T - jump_len, jump_len)}
    t, ts = t_T, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if jumps.get(t, 0) > 0:
            jumps[t] -= 1
            for _ in range(jump_len):
                t += 1
                ts.append(t)
    ts.append(-1)
    return ts


def _inpaint_state(batch: tuple, K: int, dsh, dev, randint: Optional[Callable], generator: Optional[torch.Generator]):
    """The state ``perform_simple_inference`` builds before its loop (diffuser.py:405-436): random
    codes with codebook 0 pinned to the AR output, the known-value / mask tensors, and in deep-clone
    mode the reference codes prepended as fully known frames.  Returns (xr, x_known, m, offset)."""
    c_text, c_codes, c_text_lengths, c_codes_lengths, x, x_padding_mask = batch
    assert c_text.shape[0] == 1, "batch size 1 per call (the reference breaks for bs > 1, SURVEY App. B-9)"
    cur = torch.cuda.current_stream(dev) if torch.device(dev).type == "cuda" else None      # the session's stream (the callers enter it)
    x = x.to(dev)
    c_codes = c_codes.to(dev)
    if cur is not None:
        from . import ops
        ops.use_on(x, cur)
        ops.use_on(c_codes, cur)
    assert int(x.max()) < K, f'Error: {int(x.max())} >= {K}'           # diffuser.py:36
    x_quant0 = x[0, :, 0].clone()
    if randint is None:
        xr = torch.randint(0, K, x.shape, dtype=x.dtype, device=dev, generator=generator)[0]
    else:
        xr = randint(tuple(x.shape)).to(dev)[0]
    xr[:, 0] = x_quant0
    x_known = torch.zeros_like(xr)
    x_known[:, 0] = xr[:, 0]
    m = torch.zeros_like(xr, dtype=torch.uint8)
    m[:, 0] = 1
    offset = 0
    if dsh.deep_clone:
        prompt = c_codes[0]
        xr = torch.cat((prompt, xr), dim=0)
        x_known = torch.cat((prompt, x_known), dim=0)
        m = torch.cat((torch.ones_like(prompt, dtype=torch.uint8), m), dim=0)
        offset = int(c_codes_lengths[0])
    return xr, x_known, m, offset


def _nar_config(T, dsh, div_mode: int) -> NARConfig:
    if dsh.jump_len != 1 or dsh.jump_n_sample != 1:
        raise NotImplementedError("RePaint resampling (jump_len/jump_n_sample != 1) is outside the shipped inference path")
    return NARConfig(T=T, x_0_temp=float(dsh.x_0_temp), guidance_w=float(dsh.guidance_w), deep_clone=bool(dsh.deep_clone),
                     q0_override_steps=int(dsh.q0_override_steps), div_mode=div_mode)


@torch.inference_mode()
def _tables(diff) -> Optional[tuple]:
    """The four log-tables of a MultinomialDiffusion (q_pred / q_posterior read them from `diff`, reference
    diffuser.py:118-206), or None for the default schedule."""
    if diff is None:
        return None
    cur = (diff.log_alpha, diff.log_1_min_alpha, diff.log_cumprod_alpha, diff.log_1_min_cumprod_alpha)
    host = getattr(diff, "_host_tables", None)
    key = _table_key(cur)
    if host is not None and key is not None and getattr(diff, "_host_key", None) == key:
        return host                                    # the tables as constructed, already on the host (no device read-back)
    return cur


def _same_tables(a, b) -> bool:
    if a is None or b is None:
        return a is None and b is None
    return all(torch.equal(x.detach().cpu().float(), y.detach().cpu().float()) for x, y in zip(a, b))


def begin_inference(model, c_text: Tensor, c_codes: Tensor, T, dsh=DSH, div_mode: int = 0, diff=None,
                    spk_vec: Optional[Tensor] = None, cond_from: Optional[NARSession] = None, stream=None) -> NARSession:
    """Start the part of ``perform_simple_inference`` that depends only on the conditioning (text ids (1,Lt),
    reference codes (1,Lc,8)) and the schedule -- speaker vector, text encoder for every step and guidance branch,
    cross-attention K / V -- on the session's own stream and return without waiting.  ``tts()`` calls this BEFORE the
    AR decode (which leaves most of the chip idle) and hands the session to ``perform_simple_inference``."""
    cfg = _nar_config(T, dsh, div_mode)
    eng = model.engine()
    times = get_schedule(T, jump_n_sample=dsh.jump_n_sample, jump_len=dsh.jump_len)[:-1]
    sess = NARSession(eng, cfg, stream=stream, diff_tables=_tables(diff))
    if cond_from is not None and cond_from.cfg == cfg and cond_from.times == list(times) and _same_tables(cond_from.diff_tables, sess.diff_tables):
        sess.adopt_cond(cond_from)          # same text, reference, schedule: nothing to recompute (Mars5TTS.prepare_reference)
    else:
        sess.prepare_cond(c_text[0], c_codes[0].to(eng.dev), times, spk_vec=spk_vec)
    return sess


@torch.inference_mode()
def perform_simple_inference(model, batch: tuple, diff: MultinomialDiffusion, T, dtype=torch.float16,
                             retain_quant0: bool = True, dsh=DSH,
                             uniform: Optional[Callable[[tuple], Tensor]] = None, randint: Optional[Callable] = None,
                             use_graph: bool = True, div_mode: int = 0, n_steps: Optional[int] = None,
                             generator: Optional[torch.Generator] = None, session: Optional[NARSession] = None,
                             on_step: Optional[Callable[[dict], None]] = None, wait: bool = True):
    """batch = (c_text (1,Lt), c_codes (1,Lc,8), c_text_lengths, c_codes_lengths, x (1,Lx,8),
    x_padding_mask); returns (1, S - offset, 8) int64.  RNG draws follow the reference order:
    randint(0,K,(1,Lx,8)) then per step rand (1,S,8,K) x2 (x1 at t = 0).
    `uniform(shape)` / `randint(shape)` override the device generator (parity tests);
    `generator` draws from a private device generator instead of the global one; `session`: the result of
    ``begin_inference`` for the same conditioning, T and dsh (its conditioning work is then not repeated);
    `on_step`: per-step observer for parity tests (``NARSession.run``); `wait=False` (pipelined serving): all steps are
    enqueued and a zero-argument callable is returned that waits for them and yields the result."""
    c_text, c_codes = batch[0], batch[1]
    assert retain_quant0, "retain_quant0=False is not a shipped configuration (inference.py:298)"
    cfg = _nar_config(T, dsh, div_mode)
    eng = model.engine()
    dev = eng.dev
    K = diff.num_classes
    assert K == eng.shape.n_quant
    times = get_schedule(T, jump_n_sample=dsh.jump_n_sample, jump_len=dsh.jump_len)[:-1]
    assert all(0 <= t < diff.num_timesteps for t in times), "T exceeds the diffusion's number of timesteps"
    sess = session if session is not None else NARSession(eng, cfg, diff_tables=_tables(diff))
    sess._enter()
    with torch.cuda.stream(sess.stream):
        xr, x_known, m, offset = _inpaint_state(batch, K, dsh, dev, randint, generator)
        if uniform is None:
            uniform = _generator_uniform(dev, generator)
    if session is None:
        sess.prepare(c_text[0], c_codes[0].to(dev), xr, x_known, m, offset, times)
    else:
        assert sess.times == list(times) and sess.cfg == cfg, "session was begun with a different schedule / DSH"
        assert _same_tables(sess.diff_tables, _tables(diff)) or (sess.diff_tables is None and _same_tables(
            _tables(diff), _tables(MultinomialDiffusion(diff.num_classes, timesteps=200)))), "session was begun with another diffusion"
        sess.prepare_state(xr, x_known, m, offset)
        sess.prepare_loop()
    out = sess.run(uniform, use_graph=use_graph, n_steps=n_steps, on_step=on_step, wait=wait)
    if not wait:
        return lambda: sess.finish()[None, offset:].clone()
    return out[None, offset:].clone()


def _generator_uniform(dev, generator: Optional[torch.Generator]):
    """The step's uniform draw as the reference makes it (``torch.rand(shape, device=...)`` on the current / given generator).
    ``out=``: fill a preallocated buffer instead -- torch.rand IS empty(shape).uniform_(0, 1, generator), so the values and the
    generator's advance are the same; ``out_ok`` tells the engine it may draw on its second stream (nar_engine._UniformRing)."""
    def uniform(shape, out: Optional[Tensor] = None) -> Tensor:
        if out is None:
            return torch.rand(shape, dtype=torch.float32, device=dev, generator=generator)
        assert tuple(out.shape) == tuple(shape) and out.dtype == torch.float32
        return out.uniform_(0.0, 1.0, generator=generator)
    uniform.out_ok = True
    # the engine may generate these draws inside its own step graph (nar_engine.PhiloxDraws: same values, same generator advance)
    uniform.gen = generator if generator is not None else torch.cuda.default_generators[
        torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()]
    return uniform


@torch.inference_mode()
def perform_batch_inference(model, batches: List[tuple], diff: MultinomialDiffusion, T, dsh=DSH,
                            generators: Optional[List[Optional[torch.Generator]]] = None,
                            uniforms: Optional[List[Callable[[tuple], Tensor]]] = None,
                            randints: Optional[List[Callable]] = None,
                            use_graph: bool = True, div_mode: int = 0, n_steps: Optional[int] = None, wait: bool = True,
                            stream: Optional[torch.cuda.Stream] = None):
    """``perform_simple_inference`` for several independent utterances at once (BASELINE config 3):
    one batched decoder pass per reverse step over all of them (``NARBatchSession``).  Utterance i
    draws its random numbers from ``generators[i]`` in the order a lone call would (randint, then
    per step two rand), so result i equals ``perform_simple_inference(batches[i], generator=generators[i])``
    whatever else is in the batch.  With ``generators=None`` (or the same generator given to several utterances) the draws of
    the shared generator are bound per utterance, contiguously, in the engine's sorted utterance order -- all of utterance a's
    steps, then utterance b's (since round 5, when the draws moved into the step graph) -- not interleaved step by step as
    an eager loop would draw them: results are deterministic for a seed, but differ from the per-step interleaving; give every
    utterance its own generator for placement-independent results.  A generator must not be shared between host THREADS
    that run inference concurrently: each run reserves its range with a get_offset / set_offset pair (serialised by
    nar_engine.GEN_LOCK for the NAR; the AR decode holds its range for the whole decode).  Returns a list of (1, S_i - offset_i, 8) int64 tensors; with `wait=False` the steps
    are only enqueued (on the session's own stream) and a callable that waits and returns that list comes back instead."""
    cfg = _nar_config(T, dsh, div_mode)
    eng = model.engine()
    dev = eng.dev
    K = diff.num_classes
    assert K == eng.shape.n_quant
    U = len(batches)
    generators = generators if generators is not None else [None] * U
    times = get_schedule(T, jump_n_sample=dsh.jump_n_sample, jump_len=dsh.jump_len)[:-1]
    sess = NARBatchSession(eng, cfg, stream=stream, diff_tables=_tables(diff))
    items, offsets, us = [], [], []
    sess.stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(sess.stream):
        for i, batch in enumerate(batches):
            g = generators[i]
            xr, x_known, m, offset = _inpaint_state(batch, K, dsh, dev, randints[i] if randints else None, g)
            items.append(dict(c_text=batch[0][0], c_codes=batch[1][0].to(dev), x=xr, x_known=x_known, m_mask=m, row_offset=offset))
            offsets.append(offset)
            if uniforms is not None:
                us.append(uniforms[i])
            else:
                us.append(_generator_uniform(dev, g))
    sess.prepare(items, times)
    outs = sess.run(us, use_graph=use_graph, n_steps=n_steps, wait=wait)
    if not wait:
        return lambda: [o[None, off:].clone() for o, off in zip(sess.finish(), offsets)]
    return [o[None, off:].clone() for o, off in zip(outs, offsets)]
