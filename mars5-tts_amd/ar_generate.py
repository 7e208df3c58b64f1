"""Drop-in for reference ``mars5/ar_generate.py:15-165``: same name, arguments, return value
and error behaviour; the token loop runs on the MI355X engine (``ar_engine``)."""
from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor

from .ar_engine import ARSamplingConfig, ARSession


@torch.inference_mode()
def ar_generate(texttok, speechtok, codeclm, xx: Tensor, ss_gen: Tensor, first_codex_idx: int,
                max_len: int = 1500, fp16: bool = True, temperature: float = 1.0, topk: int = None,
                top_p=1.0, alpha_frequency=0, alpha_presence=0, penalty_window=100,
                typical_p=1.0, eos_penalty_factor=1.0, eos_penalty_decay=0, n_phones_gen=None, vocode=True,
                beam_width: int = 1, beam_length_penalty=2, use_kv_cache: bool = True,
                noise: Optional[Tensor] = None, use_graph: bool = True, div_mode: int = 0,
                generator: Optional[torch.Generator] = None) -> Tensor:
    """Autoregressively complete `xx` (seq_len,) with the `codeclm` language model; `ss_gen`
    (ref_len, 8) is the speaker reference.  Returns the full sequence (prompt + generated),
    EOS not appended.  `fp16`, `beam_length_penalty` and `use_kv_cache` are accepted for
    signature compatibility (the engine's operand dtype is a property of `codeclm`; the KV
    cache is always on -- "disabling/enabling kv caching won't affect output", inference.py:66).

    Extensions (keyword-only in spirit): `noise` (n_steps, V) Exp(1) draws to use instead of
    the device generator (parity tests), `use_graph`, `div_mode`, `generator` (a private device
    generator instead of the global one: per-utterance RNG streams for batched serving).
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
        logging.warning(f"[autoregressive generation] output length = {P} -- inference likely failed or input too long!")
        return xx.to(dev)

    sess = ARSession(eng, max_len)
    n_steps = max_len - P
    gen = None
    with torch.cuda.stream(sess.stream):
        if noise is None:
            # One Exp(1) vector per sampler call, drawn call-by-call like torch.multinomial does
            # (ar_generate.py:115).  All max_len - P rows are drawn up front so the decode loop never
            # touches the host; the generator is then rewound to where the reference leaves it (one
            # draw per executed loop iteration), so what follows (the NAR stage) sees the same stream.
            gen = generator if generator is not None else torch.cuda.default_generators[dev.index]
            off0 = gen.get_offset()
            noise_d = torch.empty(n_steps, n_vocab, dtype=torch.float32, device=dev)
            for i in range(n_steps):
                noise_d[i].exponential_(1, generator=generator)
            per_draw = (gen.get_offset() - off0) // n_steps
        else:
            noise_d = noise.to(device=dev, dtype=torch.float32).contiguous()
    cfg = ARSamplingConfig(temperature=float(temperature), topk=topk, top_p=float(top_p), alpha_frequency=float(alpha_frequency),
                           alpha_presence=float(alpha_presence), penalty_window=int(penalty_window), typical_p=float(typical_p),
                           eos_penalty_factor=float(eos_penalty_factor), eos_penalty_decay=float(eos_penalty_decay),
                           n_phones_gen=n_phones_gen, div_mode=div_mode)
    sess.configure_sampler(cfg, n_text, eos_idx, noise_d)
    sess.prefill(xx, ss_gen)
    out = sess.decode(use_graph=use_graph)
    if gen is not None:
        n_iter = (int(out.shape[-1]) - P) + (1 if sess.ended_on_eos else 0)      # loop iterations the reference executes
        gen.set_offset(off0 + n_iter * per_draw)
    if out.shape[-1] >= max_len - 1:
        logging.warning(f"[autoregressive generation] output length = {out.shape[-1]} -- inference likely failed or input too long!")
    return out
