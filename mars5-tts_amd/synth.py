"""Deterministic synthetic checkpoints, tokenizers and inputs in the reference's own formats.

No MARS5 checkpoints exist offline, so tests, goldens and the bench use
seeded random weights laid out exactly like the reference's ckpt dicts
(``{'vocab': {'texttok.model', 'speechtok.model'}, 'model': state_dict}``, reference
``inference.py:80-112``; state-dict names SURVEY §8b).  Every tensor is drawn from its own
generator seeded by (seed, crc32(name)), so the result does not depend on iteration order
and is reproducible on any host with the same torch build (CPU mt19937 stream).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch

from .minbpe import GPT4_SPLIT_PATTERN, train_merges, write_model_text


# --------------------------------------------------------------------------- configs
@dataclass(frozen=True)
class ARShape:
    """CodecLM geometry (reference model.py:44-58)."""
    n_vocab: int
    dim: int = 1536
    nhead: int = 24
    n_layers: int = 26
    n_spk_layers: int = 2
    hidden_dim: int = 3584          # int(dim * 7/3)
    sliding_window: int = 3000
    head_dim: int = 64
    norm_eps: float = 1e-5

    @property
    def spk_ff(self) -> int:        # int(dim*4*(3/4)), reference model.py:57
        return int(self.dim * 4 * (3 / 4))


@dataclass(frozen=True)
class NARShape:
    """ResidualTransformer geometry (reference model.py:165-244)."""
    n_text_vocab: int
    n_quant: int = 1025
    dim: int = 1024
    nhead: int = 16
    enc_layers: int = 8
    dec_layers: int = 16
    n_spk_layers: int = 3
    t_emb_dim: int = 1024
    n_codebooks: int = 8

    @property
    def dim_ff(self) -> int:
        return int(self.dim * 4 * (3 / 4))


def tiny_ar_shape(n_vocab: int) -> ARShape:
    return ARShape(n_vocab=n_vocab, dim=192, nhead=3, n_layers=2, n_spk_layers=1,
                   hidden_dim=int(192 * 7 / 3))


def tiny_nar_shape(n_text_vocab: int) -> NARShape:
    return NARShape(n_text_vocab=n_text_vocab, dim=128, nhead=2, enc_layers=2, dec_layers=2,
                    n_spk_layers=1, t_emb_dim=128)


def full_ar_shape(n_vocab: int) -> ARShape:
    return ARShape(n_vocab=n_vocab)


def full_nar_shape(n_text_vocab: int) -> NARShape:
    return NARShape(n_text_vocab=n_text_vocab)


# --------------------------------------------------------------------------- tensors
def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) % (2 ** 63))
    return g


def _uniform(seed, name, shape, bound):
    return (torch.rand(shape, generator=_gen(seed, name), dtype=torch.float32) * 2 - 1) * bound


def _normal(seed, name, shape, std=1.0):
    return torch.randn(shape, generator=_gen(seed, name), dtype=torch.float32) * std


def _linear(sd, seed, name, out_f, in_f, bias=False, wname="weight"):
    bound = in_f ** -0.5
    sd[f"{name}.{wname}" if wname else name] = _uniform(seed, name + ".w", (out_f, in_f), bound)
    if bias:
        sd[f"{name}.bias"] = _uniform(seed, name + ".b", (out_f,), bound)


def _norm(sd, seed, name, dim, bias=True):
    # weights near one / biases near zero, but not exactly, so affine bugs are visible
    sd[f"{name}.weight"] = 1.0 + _uniform(seed, name + ".w", (dim,), 0.1)
    if bias:
        sd[f"{name}.bias"] = _uniform(seed, name + ".b", (dim,), 0.05)


def _encoder_layer(sd, seed, prefix, dim, ff, cross=False):
    sd[f"{prefix}.self_attn.in_proj_weight"] = _uniform(seed, prefix + ".sa.w", (3 * dim, dim), dim ** -0.5)
    sd[f"{prefix}.self_attn.in_proj_bias"] = _uniform(seed, prefix + ".sa.b", (3 * dim,), 0.02)
    _linear(sd, seed, f"{prefix}.self_attn.out_proj", dim, dim, bias=True)
    if cross:
        sd[f"{prefix}.multihead_attn.in_proj_weight"] = _uniform(seed, prefix + ".ca.w", (3 * dim, dim), dim ** -0.5)
        sd[f"{prefix}.multihead_attn.in_proj_bias"] = _uniform(seed, prefix + ".ca.b", (3 * dim,), 0.02)
        _linear(sd, seed, f"{prefix}.multihead_attn.out_proj", dim, dim, bias=True)
    _linear(sd, seed, f"{prefix}.linear2", dim, ff, bias=True)
    _norm(sd, seed, f"{prefix}.norm1", dim)
    _norm(sd, seed, f"{prefix}.norm2", dim)
    if cross:
        _norm(sd, seed, f"{prefix}.norm3", dim)
    _linear(sd, seed, f"{prefix}.activation.V", ff, dim)
    _linear(sd, seed, f"{prefix}.activation.W", ff, dim)


def make_ar_state_dict(shape: ARShape, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 state dict with the reference CodecLM parameter names."""
    sd: Dict[str, torch.Tensor] = {}
    D, F, V = shape.dim, shape.hidden_dim, shape.n_vocab
    assert shape.nhead * shape.head_dim == D
    for l in range(shape.n_layers):
        p = f"ar.layers.{l}"
        for w in ("wq", "wk", "wv", "wo"):
            _linear(sd, seed, f"{p}.attention.{w}", D, D)
        _linear(sd, seed, f"{p}.feed_forward.w1", F, D)
        _linear(sd, seed, f"{p}.feed_forward.w2", D, F)
        _linear(sd, seed, f"{p}.feed_forward.w3", F, D)
        _norm(sd, seed, f"{p}.attention_norm", D, bias=False)
        _norm(sd, seed, f"{p}.ffn_norm", D, bias=False)
    _norm(sd, seed, "ar.norm", D, bias=False)
    _linear(sd, seed, "ar.output", V, D)
    # logits ~ N(0, ~1.3): scale the head up so sampling is not uniform
    sd["ar.output.weight"] *= 4.0
    sd["embed.weight"] = _normal(seed, "embed", (V, D))
    sd["pos_embedding.alpha"] = torch.tensor([0.75])
    for q in range(8):
        sd[f"ref_chunked_emb.embs.{q}.weight"] = _normal(seed, f"ref_chunked_emb.{q}", (1025, D // 8))
    sd["spk_identity_emb.weight"] = _normal(seed, "spk_identity_emb", (1, D))
    for l in range(shape.n_spk_layers):
        _encoder_layer(sd, seed, f"spk_encoder.layers.{l}", D, shape.spk_ff)
    _norm(sd, seed, "spk_encoder.norm", D)
    return sd


def make_nar_state_dict(shape: NARShape, seed: int = 1) -> Dict[str, torch.Tensor]:
    """fp32 state dict with the reference ResidualTransformer parameter names."""
    sd: Dict[str, torch.Tensor] = {}
    D, FF, Q = shape.dim, shape.dim_ff, shape.n_codebooks
    sd["cond_pos_embedding.alpha"] = torch.tensor([0.9])
    sd["pos_embedding.alpha"] = torch.tensor([1.1])
    sd["ref_pos_embedding.alpha"] = torch.tensor([0.8])
    for l in range(shape.enc_layers):
        _encoder_layer(sd, seed, f"tfm.encoder.layers.{l}", D, FF)
    _norm(sd, seed, "tfm.encoder.norm", D)
    for l in range(shape.dec_layers):
        _encoder_layer(sd, seed, f"tfm.decoder.layers.{l}", D, FF, cross=True)
    _norm(sd, seed, "tfm.decoder.norm", D)
    for which in ("timestep_encoder_emb", "timestep_decoder_emb"):
        _linear(sd, seed, f"{which}.0", D, shape.t_emb_dim, bias=True)
        _linear(sd, seed, f"{which}.2", D, D, bias=True)
    sd["text_embed.weight"] = _normal(seed, "text_embed", (shape.n_text_vocab, D))
    for q in range(Q):
        sd[f"ref_embedder.embs.{q}.weight"] = _normal(seed, f"ref_embedder.{q}", (shape.n_quant, D // Q))
        sd[f"residual_encoder.embs.{q}.weight"] = _normal(seed, f"residual_encoder.{q}", (shape.n_quant, D // Q))
    sd["spk_identity_emb.weight"] = _normal(seed, "nar.spk_identity_emb", (1, D))
    for l in range(shape.n_spk_layers):
        _encoder_layer(sd, seed, f"spk_encoder.layers.{l}", D, FF)
    _norm(sd, seed, "spk_encoder.norm", D)
    for q in range(Q):
        _norm(sd, seed, f"residual_decoder.{q}.0", D)
        _linear(sd, seed, f"residual_decoder.{q}.1", shape.n_quant, D, bias=True)
        sd[f"residual_decoder.{q}.1.weight"] *= 3.0
    return sd


# --------------------------------------------------------------------------- tokenizers
_CORPUS = (
    "the quick brown fox jumps over the lazy dog. we actually have not managed to meet demand. "
    "speech synthesis turns written text into natural sounding audio for every speaker. "
    "a reference recording of a few seconds is enough to clone the voice and the prosody. "
    "the weather today is bright and clear with a light breeze coming in from the north east. "
    "please remember to bring the documents that were discussed during the meeting yesterday. "
)


def make_text_tokenizer_model(n_merges: int, seed: int = 3) -> str:
    """minbpe-v1 text with 256 bytes + n_merges + <|startoftext|>, <|endoftext|> (names
    required by reference inference.py:223)."""
    words = GPT4_split(_CORPUS * 2)
    merges = train_merges([list(w.encode("utf-8")) for w in words], n_merges, 256)
    g = _gen(seed, "text-merges")
    seen = set(merges)
    while len(merges) < n_merges:       # filler merges that never fire on real text
        hi = 256 + len(merges)
        a, b = (int(x) for x in torch.randint(128, hi, (2,), generator=g))
        if (a, b) not in seen:
            seen.add((a, b))
            merges.append((a, b))
    specials = {"<|startoftext|>": 256 + n_merges, "<|endoftext|>": 256 + n_merges + 1}
    return write_model_text(GPT4_SPLIT_PATTERN, specials, merges)


def GPT4_split(text: str) -> List[str]:
    import regex
    return regex.findall(regex.compile(GPT4_SPLIT_PATTERN), text)


def speech_corpus_codes(n: int, seed: int = 5, alphabet: int = 12) -> List[int]:
    """L0 code sequence over a small alphabet so that BPE merges exist and fire."""
    g = _gen(seed, "speech-corpus")
    sym = torch.randint(0, 1024, (alphabet,), generator=g)
    idx = torch.randint(0, alphabet, (n,), generator=g)
    return [int(sym[i]) for i in idx]


def make_speech_tokenizer_model(n_merges: int, seed: int = 5) -> str:
    """minbpe-v1 text with 1024 codes + n_merges + <|endofspeech|> (reference
    ar_generate.py:47)."""
    merges: List[Tuple[int, int]] = []
    if n_merges > 0:
        merges = train_merges([speech_corpus_codes(4000, seed)], n_merges, 1024)
        g = _gen(seed, "speech-merges")
        seen = set(merges)
        while len(merges) < n_merges:
            hi = 1024 + len(merges)
            a, b = (int(x) for x in torch.randint(0, hi, (2,), generator=g))
            if (a, b) not in seen:
                seen.add((a, b))
                merges.append((a, b))
    specials = {"<|endofspeech|>": 1024 + n_merges}
    return write_model_text(GPT4_SPLIT_PATTERN, specials, merges)


# --------------------------------------------------------------------------- bundles
@dataclass
class SynthBundle:
    ar_ckpt: dict
    nar_ckpt: dict
    ar_shape: ARShape
    nar_shape: NARShape
    n_text: int = field(default=0)
    n_speech: int = field(default=0)


def make_vocab(text_merges: int, speech_merges: int) -> Dict[str, str]:
    """The checkpoint's ``vocab`` entry alone (tokenizer model texts), without any weights."""
    return {"texttok.model": make_text_tokenizer_model(text_merges), "speechtok.model": make_speech_tokenizer_model(speech_merges)}


def make_bundle(size: str = "tiny", seed: int = 0, text_merges: int = None, speech_merges: int = None,
                dtype_round: str = None, sliding_window: int = None) -> SynthBundle:
    """size 'tiny' (CPU-test scale) or 'full' (the real MARS5 geometry, n_vocab 4096).

    full: text vocab 256+2813+2 = 3071, speech vocab 1024+0+1 = 1025 (merge-free: one AR
    token = one 75 Hz frame) -> n_vocab 4096 as in SURVEY §8(d).
    dtype_round: 'f16'/'bf16' rounds every weight to that grid first (the real checkpoints
    are fp16-valued), so a reduced-precision engine sees exactly the oracle's weights."""
    if size == "tiny":
        tm = 30 if text_merges is None else text_merges
        sm = 63 if speech_merges is None else speech_merges
    elif size == "full":
        tm = 2813 if text_merges is None else text_merges
        sm = 0 if speech_merges is None else speech_merges
    else:
        raise ValueError(size)
    n_text, n_speech = 256 + tm + 2, 1024 + sm + 1
    vocab = {"texttok.model": make_text_tokenizer_model(tm), "speechtok.model": make_speech_tokenizer_model(sm)}
    n_vocab = n_text + n_speech
    if size == "tiny":
        a, n = tiny_ar_shape(n_vocab), tiny_nar_shape(n_text + 1)
    else:
        a, n = full_ar_shape(n_vocab), full_nar_shape(n_text + 1)
    if sliding_window is not None:      # rotating-KV-cache tests: same weights, smaller window (reference model.py:44)
        import dataclasses
        a = dataclasses.replace(a, sliding_window=int(sliding_window))
    ar_sd, nar_sd = make_ar_state_dict(a, seed), make_nar_state_dict(n, seed + 1)
    if dtype_round is not None:
        dt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype_round]
        for sd in (ar_sd, nar_sd):
            for k in sd:
                sd[k] = sd[k].to(dt).to(torch.float32)
    return SynthBundle({"vocab": vocab, "model": ar_sd}, {"vocab": dict(vocab), "model": nar_sd}, a, n, n_text, n_speech)


def make_ref_codes(n_frames: int, seed: int = 7, merge_friendly: bool = False) -> torch.Tensor:
    """(1, 8, n_frames) int64 Encodec codes in [0, 1024) (stand-in for codec.encode)."""
    g = _gen(seed, "ref-codes")
    codes = torch.randint(0, 1024, (1, 8, n_frames), generator=g, dtype=torch.long)
    if merge_friendly:
        codes[0, 0] = torch.tensor(speech_corpus_codes(n_frames, seed=5), dtype=torch.long)
    return codes
