"""Drop-in for reference ``mars5/ar_generate.py:15-165``: same name, arguments, return value
and error behaviour; the token loop runs on the MI355X engine (``ar_engine``)."""
from __future__ import annotations

import logging
from typing import List, Optional, Union

import torch
from torch import Tensor

from . import _lib as L
from .ar_engine import ARBatchSession, ARSamplingConfig, ARSession


@torch.inference_mode()
def ar_generate(texttok, speechtok, codeclm, xx: Tensor, ss_gen: Tensor, first_codex_idx: int,
                max_len: int = 1500, fp16: bool = True, temperature: float = 1.0, topk: int = None,
                top_p=1.0, alpha_frequency=0, alpha_presence=0, penalty_window=100,
                typical_p=1.0, eos_penalty_factor=1.0, eos_penalty_decay=0, n_phones_gen=None, vocode=True,
                beam_width: int = 1, beam_length_penalty=2, use_kv_cache: bool = True,
                noise: Optional[Tensor] = None, use_graph: bool = True, div_mode: int = 0,
                generator: Optional[torch.Generator] = None, spk_vec: Optional[Tensor] = None, stream=None) -> Tensor:
    """Autoregressively complete `xx` (seq_len,) with the `codeclm` language model; `ss_gen`
    (ref_len, 8) is the speaker reference.  Returns the full sequence (prompt + generated),
    EOS not appended.  `fp16`, `beam_length_penalty` and `use_kv_cache` are accepted for
    signature compatibility (the engine's operand dtype is a property of `codeclm`; the KV
    cache is always on -- "disabling/enabling kv caching won't affect output", inference.py:66).

    Extensions (keyword-only in spirit): `noise` (n_steps, V) Exp(1) draws to use instead of
    the device generator (parity tests), `use_graph`, `div_mode`, `generator` (a private device
    generator instead of the global one: per-utterance RNG streams for batched serving), `spk_vec` (the speaker
    vector of `ss_gen` computed earlier, ``codeclm.get_spk_embedding``: skips the speaker encoder).
    """
    assert xx.dim() == 1, "Only batch size of 1 is currently supported."
    assert beam_width == 1, "Only beam size of 1 is currently supported."
    if vocode:
        raise AssertionError()
    eng = codeclm.engine()
    dev = eng.dev
    n_text = len(texttok.vocab)
    n_vocab = n_text + len(speechtok.vocab)
    assert n_vocab == eng.shape.n_vocab, (n_vocab, eng.shape.n_vocab)
    eos_idx = n_text + speechtok.special_tokens['<|endofspeech|>']
    logging.info(f"Starting beam decoding with beam_width={beam_width}")
    P = int(xx.shape[-1])
    if P >= max_len:
        logging.warning(f"[ar_generate] prompt of {P} tokens leaves no room under max_len = {max_len}: nothing generated")
        return xx.to(dev)

    sess = ARSession(eng, max_len, stream=stream)
    n_steps = max_len - P
    gen = None
    fill = None
    rng = None
    with torch.cuda.stream(sess.stream):
        if noise is None:
            # One Exp(1) vector per sampler call, drawn call-by-call like torch.multinomial does (ar_generate.py:115).
            # Rows are drawn in chunks just ahead of the graph replays that read them (ARSession.decode asks for them
            # every poll interval), so an utterance that ends on EOS does not pay for max_len - P draws; the generator is
            # then rewound to where the reference leaves it (one draw per executed loop iteration), so what follows (the
            # NAR stage) sees the same stream.
            gen = generator if generator is not None else torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            off0 = gen.get_offset()
            probe = torch.empty(n_vocab, dtype=torch.float32, device=dev)
            probe.exponential_(1, generator=generator)                       # the generator offset one (V,) draw consumes
            per_draw = gen.get_offset() - off0
            gen.set_offset(off0)
            # Round 5: the sampler generates each call's Exp(1) values itself, bit-identical to these exponential_ draws (torch's
            # Philox4x32-10 launch geometry for V values: csrc/philox.h, m5_torch_draw_bits) -- provided this torch build advances the
            # generator the way that geometry implies; otherwise (or M5_AR_PHILOX=0, tools A/B) the rows are drawn by torch, in
            # chunks just ahead of the graph replays that read them.
            prop = torch.cuda.get_device_properties(dev)
            grid = 256 * min(int(prop.multi_processor_count) * max(int(getattr(prop, "max_threads_per_multi_processor", 2048)) // 256, 1),
                             (n_vocab + 255) // 256)
            if per_draw == ((n_vocab - 1) // (4 * grid) + 1) * 4 and L.tool_knob("M5_AR_PHILOX", "1") != "0":
                rng, noise_d = (int(gen.initial_seed()), int(off0), int(per_draw), int(grid)), None
            else:
                noise_d = torch.empty(n_steps, n_vocab, dtype=torch.float32, device=dev)

                def fill(lo, hi, _chunk=64):
                    hi = min(n_steps, (hi + _chunk - 1) // _chunk * _chunk)      # whole chunks: fewer host round trips
                    with torch.cuda.stream(sess.stream):
                        for i in range(lo, hi):
                            noise_d[i].exponential_(1, generator=generator)
                    return hi
        else:
            noise_d = noise.to(device=dev, dtype=torch.float32).contiguous()
    cfg = ARSamplingConfig(temperature=float(temperature), topk=topk, top_p=float(top_p), alpha_frequency=float(alpha_frequency),
                           alpha_presence=float(alpha_presence), penalty_window=int(penalty_window), typical_p=float(typical_p),
                           eos_penalty_factor=float(eos_penalty_factor), eos_penalty_decay=float(eos_penalty_decay),
                           n_phones_gen=n_phones_gen, div_mode=div_mode)
    sess.configure_sampler(cfg, n_text, eos_idx, noise_d, rng=rng, n_steps=n_steps)
    sess.prefill(xx, ss_gen, spk_vec=spk_vec)
    out = sess.decode(use_graph=use_graph, noise_fill=fill)
    if gen is not None:
        n_iter = (int(out.shape[-1]) - P) + (1 if sess.ended_on_eos else 0)      # loop iterations the reference executes
        gen.set_offset(off0 + n_iter * per_draw)
    if out.shape[-1] >= max_len - 1:
        logging.warning(f"[ar_generate] stopped by max_len ({out.shape[-1]} tokens) rather than by the end-of-speech token")
    return out


@torch.inference_mode()
def ar_generate_batch(texttok, speechtok, codeclm, xxs: List[Tensor], ss_gens: List[Tensor], first_codex_idxs: List[int],
                      max_len: Union[int, List[int]] = 1500, temperature: float = 1.0, topk: int = None, top_p=1.0, alpha_frequency=0,
                      alpha_presence=0, penalty_window=100, typical_p=1.0, eos_penalty_factor=1.0, eos_penalty_decay=0,
                      n_phones_gens: Optional[List[Optional[int]]] = None, generators: Optional[List[Optional[torch.Generator]]] = None,
                      noises: Optional[List[Tensor]] = None, use_graph: bool = True, div_mode: int = 0) -> List[Tensor]:
    """``ar_generate`` for B <= 32 independent requests decoded together (BASELINE config 3): request i
    completes prompt ``xxs[i]`` with speaker reference ``ss_gens[i]`` exactly as a lone call would, drawing
    its Exp(1) noise from ``generators[i]`` (or ``noises[i]``, (n_steps, V)); the decode step reads the
    weights once for all requests.  `max_len` may be a list (one cap per request).
    Returns the B full sequences (prompt + generated, EOS not appended)."""
    B_all = len(xxs)
    max_lens_all = [int(max_len)] * B_all if not isinstance(max_len, (list, tuple)) else [int(v) for v in max_len]
    assert len(max_lens_all) == B_all
    assert len(ss_gens) == B_all and len(first_codex_idxs) == B_all
    eng = codeclm.engine()
    dev = eng.dev
    n_text = len(texttok.vocab)
    n_vocab = n_text + len(speechtok.vocab)
    assert n_vocab == eng.shape.n_vocab, (n_vocab, eng.shape.n_vocab)
    eos_idx = n_text + speechtok.special_tokens['<|endofspeech|>']
    results: List[Optional[Tensor]] = [None] * B_all
    live = []
    for i, (x, ml) in enumerate(zip(xxs, max_lens_all)):
        assert x.dim() == 1
        if int(x.shape[-1]) >= ml:          # no room to generate: what a lone call does (ar_generate.py:62,160-161)
            logging.warning(f"[ar_generate] prompt of {int(x.shape[-1])} tokens leaves no room under max_len = {ml}: nothing generated")
            results[i] = x.to(dev)
        else:
            live.append(i)
    if not live:
        return results
    xxs, ss_gens = [xxs[i] for i in live], [ss_gens[i] for i in live]
    max_lens = [max_lens_all[i] for i in live]
    n_phones_gens = [n_phones_gens[i] for i in live] if n_phones_gens is not None else None
    generators = [generators[i] for i in live] if generators is not None else None
    noises = [noises[i] for i in live] if noises is not None else None
    B = len(live)
    Ps = [int(x.shape[-1]) for x in xxs]
    sess = ARBatchSession(eng, max_lens)
    n_steps = max(ml - P for ml, P in zip(max_lens, Ps))
    gens, offs, pers = [], [], []
    with torch.cuda.stream(sess.stream):
        noise_d = torch.ones(B, n_steps, n_vocab, dtype=torch.float32, device=dev) if noises is not None else None
        probe = torch.empty(n_vocab, dtype=torch.float32, device=dev)
        for b in range(B):
            nb = max_lens[b] - Ps[b]
            if noises is not None:
                noise_d[b, :nb] = noises[b].to(device=dev, dtype=torch.float32)[:nb]
                gens.append(None)
                offs.append(0)
                pers.append(0)
                continue
            g = generators[b] if generators is not None and generators[b] is not None else \
                torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
            off0 = g.get_offset()
            probe.exponential_(1, generator=g)                 # the generator offset one (V,) draw consumes
            gens.append(g)
            offs.append(off0)
            pers.append(g.get_offset() - off0)
            g.set_offset(off0)
    shared = len({id(g) for g in gens if g is not None}) < sum(g is not None for g in gens)
    if shared:                              # one generator for several requests: request b's rows follow request b-1's in full
        run = {}
        for b in range(B):
            if gens[b] is not None:
                offs[b] = offs[b] + run.get(id(gens[b]), 0)
                run[id(gens[b])] = run.get(id(gens[b]), 0) + (max_lens[b] - Ps[b]) * pers[b]

    # Round 5: every sequence's sampler generates its own Exp(1) values from its generator's Philox stream (M5SampleArgs.rng, one
    # {seed, offset0} pair per sequence; see ar_generate above) -- no (B, n_steps, V) noise tensor, no exponential_ launches.
    prop = torch.cuda.get_device_properties(dev)
    grid = 256 * min(int(prop.multi_processor_count) * max(int(getattr(prop, "max_threads_per_multi_processor", 2048)) // 256, 1), (n_vocab + 255) // 256)
    inc = ((n_vocab - 1) // (4 * grid) + 1) * 4
    rng = None
    if noises is None and all(p == inc for p in pers) and L.tool_knob("M5_AR_PHILOX", "1") != "0":
        rng = ([(int(g.initial_seed()), int(o)) for g, o in zip(gens, offs)], inc, grid)
    elif noises is None:
        with torch.cuda.stream(sess.stream):
            noise_d = torch.ones(B, n_steps, n_vocab, dtype=torch.float32, device=dev)

    def fill(lo, hi, _chunk=64):
        """rows lo..hi-1 of every sequence that draws from a generator, in whole chunks just ahead of the replays that read
        them.  Requests that share ONE generator (the global one) would interleave their draws chunk by chunk; a lone call
        draws a request's rows consecutively, so in that case everything is drawn up front, request by request."""
        hi = n_steps if shared else min(n_steps, (hi + _chunk - 1) // _chunk * _chunk)
        with torch.cuda.stream(sess.stream):
            for b in range(B):
                if gens[b] is None:
                    continue
                for i in range(lo, min(hi, max_lens[b] - Ps[b])):
                    noise_d[b, i].exponential_(1, generator=gens[b])
        return hi

    cfg = ARSamplingConfig(temperature=float(temperature), topk=topk, top_p=float(top_p), alpha_frequency=float(alpha_frequency),
                           alpha_presence=float(alpha_presence), penalty_window=int(penalty_window), typical_p=float(typical_p),
                           eos_penalty_factor=float(eos_penalty_factor), eos_penalty_decay=float(eos_penalty_decay),
                           n_phones_gen=None, div_mode=div_mode)
    sess.configure_sampler(cfg, n_text, eos_idx, noise_d, n_phones_gen=n_phones_gens, rng=rng, n_steps=n_steps)
    sess.prefill(xxs, ss_gens)
    outs = sess.decode(use_graph=use_graph, noise_fill=fill if (rng is None and any(g is not None for g in gens)) else None)
    for b in range(B):                      # leave every generator where a lone reference call leaves it
        if gens[b] is not None:
            n_iter = (int(outs[b].shape[-1]) - Ps[b]) + (1 if sess.ended_on_eos[b] else 0)
            gens[b].set_offset(offs[b] + n_iter * pers[b])
    for i, o in zip(live, outs):
        results[i] = o
    return results
