"""CPU oracle for the MARS5 hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain torch-CPU fp32, function-by-function restatement of the reference algorithm for
the AR decode loop and the multinomial-DDPM NAR refinement.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
the product path (``mars5-tts_amd/``) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference itself, generated in the build container by
``oracle/gen_golden.py`` (imports ``/root/reference`` unchanged) and committed under
``tests/golden/``;  ``tests/test_oracle_golden.py`` replays them.

Every function cites the reference lines it follows (paths relative to the reference root).
Everything operates on a flat ``state_dict`` (reference parameter names) so no reference
module is needed at run time.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
LAYERNORM_EPS = 4e-5            # mars5/model.py:13
MIN_LOG_ARG = 1e-7              # mars5/diffuser.py:18


# ---- reduced-precision emulation ---------------------------------------------------------
# On a GPU the reference runs the AR stage under ``torch.autocast(dtype=float16)``
# (mars5/ar_generate.py:59,67): every nn.Linear casts its input to the autocast dtype,
# accumulates in fp32 and returns the autocast dtype; RMSNorm / LayerNorm / softmax and the
# residual stream stay fp32 (nn_future.py:307-312, x + r promotes).  ``dt`` (None = the CPU path,
# all fp32) reproduces those rounding points on fp32 containers, so the 16-bit engines can be
# compared with what the reference computes in that dtype.  Weights must already be
# dt-representable (``round_linear_weights``): autocast's weight cast is then the identity.  Biases
# (speaker encoder only on this stage) are added in fp32 before the output rounding; autocast
# would first round them to dt -- at most half a dt-ulp of the bias, far below the output rounding.
def _r(x: Tensor, dt: Optional[torch.dtype]) -> Tensor:
    return x if dt is None else x.to(dt).to(torch.float32)


def _lin(x: Tensor, w: Tensor, b: Optional[Tensor] = None, dt: Optional[torch.dtype] = None) -> Tensor:
    if dt is None:
        return F.linear(x, w, b)
    return _r(F.linear(_r(x, dt), w, b), dt)


def round_linear_weights(sd: Dict[str, Tensor], dt: torch.dtype) -> Dict[str, Tensor]:
    """Copy of ``sd`` with every matrix that feeds an nn.Linear rounded to ``dt`` (what autocast /
    a 16-bit engine sees); embedding tables, norms, biases and scalars keep their fp32 values."""
    out = {}
    for k, v in sd.items():
        is_table = k in ("embed.weight", "text_embed.weight", "spk_identity_emb.weight") or ".embs." in k
        out[k] = v.to(dt).to(torch.float32) if (v.dim() == 2 and not is_table) else v
    return out


# ======================================================================================
# shared building blocks
# ======================================================================================
def sine_positional_table(n: int, dim: int) -> Tensor:
    """mars5/nn_future.py:51-76 (non-reversed): pe[:,0::2]=sin, pe[:,1::2]=cos."""
    pe = torch.zeros(n, dim)
    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def sine_positional_embedding(x: Tensor, alpha: Tensor) -> Tensor:
    """mars5/nn_future.py:78-83: x * 1.0 + alpha * pe[:len]; x is (L, dim).  The reference
    builds the table for max(4000, L) rows -- values do not depend on the table length."""
    pe = sine_positional_table(max(4000, x.shape[0]), x.shape[1])
    return x * 1.0 + alpha * pe[: x.shape[0]]


def chunked_embedding(sd: Dict[str, Tensor], prefix: str, codes: Tensor) -> Tensor:
    """mars5/model.py:154-159: codes (L, 8) -> (L, dim), concat of 8 per-codebook gathers."""
    return torch.cat([sd[f"{prefix}.embs.{i}.weight"][codes[:, i]] for i in range(codes.shape[-1])], dim=-1)


def precompute_freqs_cis(head_dim: int, end: int, theta: float = 10000.0) -> Tensor:
    """mars5/nn_future.py:194-198."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: (head_dim // 2)].float() / head_dim))
    t = torch.arange(end)
    freqs = torch.outer(t, freqs).float()
    return torch.polar(torch.ones_like(freqs), freqs)


def apply_rotary(x: Tensor, freqs_cis: Tensor) -> Tensor:
    """mars5/nn_future.py:181-191; x (L, H, hd), freqs_cis (L, hd/2) complex."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    out = torch.view_as_real(xc * freqs_cis[:, None, :]).flatten(2)
    return out.type_as(x)


def rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """mars5/nn_future.py:307-312."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x) * w


def mha(q_in: Tensor, kv_in: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor,
        nhead: int, key_mask: Optional[Tensor], dt: Optional[torch.dtype] = None) -> Tensor:
    """torch.nn.MultiheadAttention forward as used by the reference encoder/decoder layers
    (model.py:61-67,179-203): packed in-projection with bias, SDPA with scale 1/sqrt(hd),
    boolean key-padding mask (True = ignore), out-projection with bias.
    q_in (Lq, D), kv_in (Lk, D)."""
    D = q_in.shape[-1]
    hd = D // nhead
    q = _lin(q_in, w_in[:D], b_in[:D], dt)
    k = _lin(kv_in, w_in[D:2 * D], b_in[D:2 * D], dt)
    v = _lin(kv_in, w_in[2 * D:], b_in[2 * D:], dt)
    q = q.view(-1, nhead, hd).transpose(0, 1)
    k = k.view(-1, nhead, hd).transpose(0, 1)
    v = v.view(-1, nhead, hd).transpose(0, 1)
    scores = (q @ k.transpose(1, 2)) / math.sqrt(hd)
    if key_mask is not None:
        scores = scores.masked_fill(key_mask[None, None, :], float("-inf"))
    att = _r(torch.softmax(scores, dim=-1) @ v, dt)
    return _lin(att.transpose(0, 1).reshape(-1, D), w_out, b_out, dt)


def swiglu_ff(x: Tensor, sd: Dict[str, Tensor], p: str, dt: Optional[torch.dtype] = None) -> Tensor:
    """linear1 = Identity, activation = FNNSwiGLU (nn_future.py:21-29), then linear2+bias."""
    h = _r(_r(F.silu(_lin(x, sd[f"{p}.activation.W.weight"], None, dt)), dt) * _lin(x, sd[f"{p}.activation.V.weight"], None, dt), dt)
    return _lin(h, sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"], dt)


def encoder_layer(x: Tensor, sd: Dict[str, Tensor], p: str, nhead: int, key_mask: Optional[Tensor],
                  dt: Optional[torch.dtype] = None) -> Tensor:
    """nn.TransformerEncoderLayer, norm_first=True, eps 4e-5 (model.py:61-67)."""
    h = F.layer_norm(x, x.shape[-1:], sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], LAYERNORM_EPS)
    x = x + mha(h, h, sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"],
                sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"], nhead, key_mask, dt)
    h = F.layer_norm(x, x.shape[-1:], sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], LAYERNORM_EPS)
    return x + swiglu_ff(h, sd, p, dt)


def decoder_layer(x: Tensor, mem: Tensor, sd: Dict[str, Tensor], p: str, nhead: int,
                  tgt_mask: Optional[Tensor], mem_mask: Optional[Tensor], dt: Optional[torch.dtype] = None) -> Tensor:
    """nn.TransformerDecoderLayer, norm_first=True (model.py:187-193): self-attn, cross-attn
    (queries from tgt, keys/values from memory), SwiGLU feed-forward."""
    h = F.layer_norm(x, x.shape[-1:], sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], LAYERNORM_EPS)
    x = x + mha(h, h, sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"],
                sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"], nhead, tgt_mask, dt)
    h = F.layer_norm(x, x.shape[-1:], sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], LAYERNORM_EPS)
    x = x + mha(h, mem, sd[f"{p}.multihead_attn.in_proj_weight"], sd[f"{p}.multihead_attn.in_proj_bias"],
                sd[f"{p}.multihead_attn.out_proj.weight"], sd[f"{p}.multihead_attn.out_proj.bias"], nhead, mem_mask, dt)
    h = F.layer_norm(x, x.shape[-1:], sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], LAYERNORM_EPS)
    return x + swiglu_ff(h, sd, p, dt)


def _count_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def speaker_encode(sd: Dict[str, Tensor], codes: Tensor, nhead: int, emb_prefix: str, pos_alpha: str,
                   valid_len: Optional[int] = None, dt: Optional[torch.dtype] = None) -> Tensor:
    """Speaker-reference encoder shared by CodecLM (model.py:109-127) and
    ResidualTransformer (model.py:298-310): [spk_identity, chunked_emb(codes)] + sine pos,
    pre-LN encoder layers, final LayerNorm, take position 0.  codes (Lc, 8).
    ``valid_len``: number of valid frames (keys beyond 1+valid_len are masked); None = all
    (AR: padding only where code == 1024, model.py:119; absent for real codes)."""
    seq = torch.cat([sd["spk_identity_emb.weight"], chunked_embedding(sd, emb_prefix, codes)], dim=0)
    seq = sine_positional_embedding(seq, sd[pos_alpha])
    key_mask = None
    if valid_len is not None:
        key_mask = torch.arange(seq.shape[0]) >= (valid_len + 1)
    for l in range(_count_layers(sd, "spk_encoder.layers")):
        seq = encoder_layer(seq, sd, f"spk_encoder.layers.{l}", nhead, key_mask, dt)
    seq = F.layer_norm(seq, seq.shape[-1:], sd["spk_encoder.norm.weight"], sd["spk_encoder.norm.bias"], LAYERNORM_EPS)
    return seq[0]


# ======================================================================================
# AR: CodecLM + Mistral stack with KV cache   (model.py:95-141, nn_future.py:235-398)
# ======================================================================================
@dataclass
class ARState:
    k: List[Tensor] = field(default_factory=list)   # per layer (L_cached, H, hd)
    v: List[Tensor] = field(default_factory=list)
    spk: Optional[Tensor] = None


def ar_spk_vector(sd: Dict[str, Tensor], ref_codes: Tensor, nhead: int, dt: Optional[torch.dtype] = None) -> Tensor:
    """model.py:109-127: ref_codes (Lc, 8).  AR padding mask = cumsum(code0 == 1024) > 0."""
    pad = (ref_codes[:, 0] == 1024).cumsum(0) > 0
    valid = None if not bool(pad.any()) else int((~pad).sum())
    return speaker_encode(sd, ref_codes, nhead, "ref_chunked_emb", "pos_embedding.alpha", valid, dt)


def mistral_forward(sd: Dict[str, Tensor], h: Tensor, positions: Tensor, st: ARState, nhead: int,
                    norm_eps: float = 1e-5, sliding_window: int = 3000, dt: Optional[torch.dtype] = None) -> Tensor:
    """nn_future.py:369-398 + Attention.forward :235-274 + FeedForward :297-298.
    h (M, D) rows at RoPE ``positions`` (M,).  M > 1 = prefill (attends over the fresh k/v
    with the causal band mask, :380-392, while filling the cache); M == 1 = decode against
    the cache.  The rotating buffer (slot = pos % sliding_window, :249) is order-agnostic
    because RoPE is applied before caching; this oracle keeps a plain list and drops the
    oldest entry beyond the window, which is the same set of keys."""
    M, D = h.shape
    hd = D // nhead
    freqs = precompute_freqs_cis(hd, int(positions.max()) + 1)[positions]
    n_layers = _count_layers(sd, "ar.layers")
    mask = None
    if M > 1:
        band = torch.triu(torch.tril(torch.ones(M, M)), diagonal=-sliding_window)
        mask = torch.log(band)
    for l in range(n_layers):
        p = f"ar.layers.{l}"
        a = rmsnorm(h, sd[f"{p}.attention_norm.weight"], norm_eps)
        q = _lin(a, sd[f"{p}.attention.wq.weight"], None, dt).view(M, nhead, hd)
        k = _lin(a, sd[f"{p}.attention.wk.weight"], None, dt).view(M, nhead, hd)
        v = _lin(a, sd[f"{p}.attention.wv.weight"], None, dt).view(M, nhead, hd)
        q, k = _r(apply_rotary(q, freqs), dt), _r(apply_rotary(k, freqs), dt)      # .type_as(x): back to the autocast dtype
        if len(st.k) <= l:
            st.k.append(k[-sliding_window:].clone())
            st.v.append(v[-sliding_window:].clone())
        else:
            st.k[l] = torch.cat([st.k[l], k])[-sliding_window:]
            st.v[l] = torch.cat([st.v[l], v])[-sliding_window:]
        if M > 1:
            key, val = k, v
        else:
            key, val = st.k[l], st.v[l]
        scores = torch.einsum("mhd,nhd->hmn", q, key) / math.sqrt(hd)
        if mask is not None:
            scores = scores + mask[None]
        o = _r(torch.einsum("hmn,nhd->mhd", torch.softmax(scores, dim=-1), val).reshape(M, D), dt)
        h = h + _lin(o, sd[f"{p}.attention.wo.weight"], None, dt)
        f = rmsnorm(h, sd[f"{p}.ffn_norm.weight"], norm_eps)
        g = _r(_r(F.silu(_lin(f, sd[f"{p}.feed_forward.w1.weight"], None, dt)), dt) * _lin(f, sd[f"{p}.feed_forward.w3.weight"], None, dt), dt)
        h = h + _lin(g, sd[f"{p}.feed_forward.w2.weight"], None, dt)
    return _lin(rmsnorm(h, sd["ar.norm.weight"], norm_eps), sd["ar.output.weight"], None, dt)


def codeclm_step(sd: Dict[str, Tensor], tokens: Tensor, ref_codes: Tensor, st: ARState, counter: int,
                 nhead: int, recompute_spk: bool = False, sliding_window: int = 3000,
                 dt: Optional[torch.dtype] = None) -> Tensor:
    """CodecLM.forward with a KV cache (model.py:95-141): internal sequence is
    [spk_vec, tok_0 .. tok_{L-1}] so token i sits at position i+1; counter == 1 runs the
    whole prefix (prefill) and strips the speaker position; later steps feed the last
    token only.  Returns the last-position logits (V,).
    The reference recomputes the speaker vector every step (bit-identical, SURVEY App.B-12);
    ``recompute_spk`` reproduces that *cost* for the CPU baseline."""
    if st.spk is None or recompute_spk:
        st.spk = ar_spk_vector(sd, ref_codes, nhead, dt)
    L = tokens.shape[0]
    if counter == 1:
        x = torch.cat([st.spk[None], sd["embed.weight"][tokens]], dim=0)
        positions = torch.arange(0, L + 1)
    else:
        x = sd["embed.weight"][tokens[-1:]]
        positions = torch.tensor([L])
    return mistral_forward(sd, x, positions, st, nhead, sliding_window=sliding_window, dt=dt)[-1]


# ---- sampler chain  (samplers.py + ar_generate.py:74-115) ----------------------------
@dataclass
class ARSamplingParams:
    temperature: float = 0.7
    top_k: int = 200
    top_p: float = 0.2
    typical_p: float = 1.0
    alpha_frequency: float = 3.0
    alpha_presence: float = 0.4
    penalty_window: int = 80
    eos_penalty_decay: float = 0.5
    eos_penalty_factor: float = 1.0
    n_phones_gen: Optional[int] = None


def filter_logits(logits: Tensor, prev_ids: Sequence[int], p: ARSamplingParams, n_text: int, eos_idx: int) -> Tensor:
    """ar_generate.py:74-98 in order; logits (V,) fp32 -> filtered logits (V,).
    NB the text-id mask is [0, n_text-1): the last text id stays live (App. B-1)."""
    z = logits.clone()
    if len(prev_ids) > 1:                                   # ar_generate.py:77, samplers.py:20-36
        window = torch.tensor(prev_ids[-p.penalty_window:], dtype=torch.long)
        c = torch.zeros_like(z, dtype=torch.long)
        vals, cnts = window.unique(return_counts=True)
        c[vals] = cnts
        z = z - c * p.alpha_frequency - (c > 0).to(z.dtype) * p.alpha_presence
    z[: n_text - 1] = float("-inf")                         # ar_generate.py:82
    if p.n_phones_gen is not None:                          # samplers.py:39-56
        n_gen = len(prev_ids)
        if not n_gen > p.n_phones_gen:
            penalty = max(p.n_phones_gen - n_gen, 1)
            z[eos_idx] -= p.eos_penalty_factor * (penalty ** p.eos_penalty_decay)
    z = z / p.temperature                                   # ar_generate.py:91
    if p.top_k is not None and p.top_k > 0:                 # samplers.py:70-74
        k = min(max(p.top_k, 1), z.numel())
        z[z < torch.topk(z, k)[0][-1]] = float("-inf")
    if p.top_p < 1.0:                                       # samplers.py:76-91
        s, idx = torch.sort(z, descending=True)
        cum = torch.cumsum(F.softmax(s, dim=-1), dim=-1)
        rm = cum > p.top_p
        rm[1:] = rm[:-1].clone()
        rm[0] = False
        z[idx[rm]] = float("-inf")
    if not p.typical_p > 0.999:                             # samplers.py:96-122
        normalized = F.log_softmax(z, dim=-1)
        pr = torch.exp(normalized)
        ent = -(normalized * pr).nansum(-1, keepdim=True)
        shifted = torch.abs((-normalized) - ent)
        ss, si = torch.sort(shifted, descending=False)
        cp = z[si].softmax(dim=-1).cumsum(dim=-1)
        last = int((cp < p.typical_p).sum())
        z = z.masked_fill(shifted > ss[last], float("-inf"))
    z[: n_text - 1] = float("-inf")                         # ar_generate.py:96
    return z


def draw_token(z: Tensor, q: Tensor) -> int:
    """ar_generate.py:102,115: log_softmax -> exp -> torch.multinomial(num_samples=1), which
    for one sample is argmax(p / q) with q ~ Exp(1) (ATen multinomial fast path)."""
    return int(torch.argmax(z.log_softmax(dim=-1).exp() / q))


def ar_generate_oracle(sd: Dict[str, Tensor], nhead: int, n_text: int, n_speech: int, eos_special: int,
                       prompt: Tensor, ref_codes: Tensor, max_len: int, params: ARSamplingParams,
                       generator: Optional[torch.Generator] = None, noise: Optional[Tensor] = None,
                       recompute_spk: bool = False, return_logits: bool = False, sliding_window: int = 3000,
                       dt: Optional[torch.dtype] = None, forced: Optional[Tensor] = None):
    """ar_generate.py:15-165 for bs = beam = 1 with the KV cache.  prompt (P,) int64 (global
    ids), ref_codes (Lc, 8).  Returns the full sequence (prompt + generated, EOS not appended,
    ar_generate.py:121-131).  RNG: one Exp(1) vector of size V per step, from ``noise[step]``
    when given, else ``torch.empty(V).exponential_(1, generator)`` (what multinomial draws).
    ``dt``: autocast dtype of the reference's GPU path (None = the CPU path, fp32).
    ``forced`` (teacher forcing, for logit comparisons): a full token sequence whose generated part is
    fed back instead of the oracle's own draws; ``choices`` then records what the oracle WOULD have drawn
    at each step given that history.  Returns (forced, logits, choices) in that mode."""
    V = n_text + n_speech
    eos_idx = n_text + eos_special
    tokens = prompt.clone()
    st = ARState()
    prev: List[int] = []
    counter = 0
    all_logits = []
    choices: List[int] = []
    if forced is not None:
        max_len = min(max_len, int(forced.shape[0]) + 1)     # one step per forced token, plus the step after the last
    while tokens.shape[0] < max_len:
        counter += 1
        logits = codeclm_step(sd, tokens, ref_codes, st, counter, nhead, recompute_spk, sliding_window, dt).float()
        if return_logits or forced is not None:
            all_logits.append(logits.clone())
        z = filter_logits(logits, prev, params, n_text, eos_idx)
        if noise is not None:
            q = noise[counter - 1]
        else:
            q = torch.empty(V).exponential_(1, generator=generator)
        tok = draw_token(z, q)
        if forced is not None:
            choices.append(tok)
            if tokens.shape[0] >= forced.shape[0]:
                break
            tok = int(forced[tokens.shape[0]])
        elif tok == eos_idx:
            break
        prev.append(tok)
        tokens = torch.cat([tokens, torch.tensor([tok])])
    if forced is not None:
        return tokens, all_logits, choices
    return (tokens, all_logits) if return_logits else tokens


# ======================================================================================
# NAR: ResidualTransformer forward   (model.py:264-343)
# ======================================================================================
def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    """model.py:18-35: cos || sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def nar_spk_vector(sd: Dict[str, Tensor], c_codes: Tensor, nhead: int, drop_cond: bool, dt: Optional[torch.dtype] = None) -> Tensor:
    """model.py:295-310: with drop_cond the codes become pad (1024) and the length 0, so
    only position 0 is attended: a per-model constant (App. B-13)."""
    if drop_cond:
        return speaker_encode(sd, torch.full_like(c_codes, 1024), nhead, "ref_embedder", "ref_pos_embedding.alpha", 0, dt)
    return speaker_encode(sd, c_codes, nhead, "ref_embedder", "ref_pos_embedding.alpha", c_codes.shape[0], dt)


def nar_forward(sd: Dict[str, Tensor], nhead: int, c_text: Tensor, c_codes: Tensor, x: Tensor, t: int,
                drop_cond: bool = False, spk_vec: Optional[Tensor] = None, dt: Optional[torch.dtype] = None) -> Tensor:
    """ResidualTransformer.forward for one utterance (bs = 1, no padding).
    c_text (Lt,), c_codes (Lc, 8), x (S, 8), t int -> logits (S, 8, K) (already in the
    permuted layout of diffuser.py:359).  The reference runs this stage in fp32 on every device
    (SURVEY App. B-4); ``dt`` is NOT a reference mode here: it applies the 16-bit engines' operand
    rounding (Linear inputs / outputs in dt, fp32 accumulate, fp32 norms and residual) so the engine's
    implementation can be checked tightly, beside the fp32 comparison that bounds the dtype's own error."""
    D = sd["text_embed.weight"].shape[1]
    t_dim = sd["timestep_encoder_emb.0.weight"].shape[1]
    Q = x.shape[-1]
    if spk_vec is None:
        spk_vec = nar_spk_vector(sd, c_codes, nhead, drop_cond, dt)
    t_emb = timestep_embedding(torch.tensor([t]), t_dim)

    def mlp(name):
        h = _r(F.silu(_lin(t_emb, sd[f"{name}.0.weight"], sd[f"{name}.0.bias"], dt)), dt)
        return F.linear(h, sd[f"{name}.2.weight"], sd[f"{name}.2.bias"])[0]

    t_enc, t_dec = mlp("timestep_encoder_emb"), mlp("timestep_decoder_emb")
    c = torch.cat([spk_vec[None], sd["text_embed.weight"][c_text]], dim=0)          # model.py:320-326
    c = sine_positional_embedding(c, sd["cond_pos_embedding.alpha"]) + t_enc[None]   # :329,:337
    xe = chunked_embedding(sd, "residual_encoder", x)                                # :332
    xe = sine_positional_embedding(xe, sd["pos_embedding.alpha"]) + t_dec[None]      # :334-336
    mem = c
    for l in range(_count_layers(sd, "tfm.encoder.layers")):
        mem = encoder_layer(mem, sd, f"tfm.encoder.layers.{l}", nhead, None, dt)
    mem = F.layer_norm(mem, (D,), sd["tfm.encoder.norm.weight"], sd["tfm.encoder.norm.bias"], LAYERNORM_EPS)
    h = xe
    for l in range(_count_layers(sd, "tfm.decoder.layers")):
        h = decoder_layer(h, mem, sd, f"tfm.decoder.layers.{l}", nhead, None, None, dt)
    h = F.layer_norm(h, (D,), sd["tfm.decoder.norm.weight"], sd["tfm.decoder.norm.bias"], LAYERNORM_EPS)
    outs = []
    for q in range(Q):                                                               # :342 (LN eps 1e-5)
        hn = F.layer_norm(h, (D,), sd[f"residual_decoder.{q}.0.weight"], sd[f"residual_decoder.{q}.0.bias"], 1e-5)
        outs.append(F.linear(_r(hn, dt), sd[f"residual_decoder.{q}.1.weight"], sd[f"residual_decoder.{q}.1.bias"]))
    return torch.stack(outs, dim=1)        # (S, Q, K)


# ---- multinomial diffusion  (diffuser.py) ---------------------------------------------
@dataclass
class DiffusionTables:
    log_alpha: Tensor
    log_1_min_alpha: Tensor
    log_cumprod_alpha: Tensor
    log_1_min_cumprod_alpha: Tensor
    num_classes: int


def diffusion_tables(num_classes: int = 1025, timesteps: int = 200, s: float = 0.008) -> DiffusionTables:
    """diffuser.py:63-109: cosine schedule evaluated in fp32, sqrt (!) of the clamped alpha
    ratios, then float64 logs/cumsum, cast back to fp32."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    ac = ac / ac[0]
    alphas = torch.sqrt(torch.clamp(ac[1:] / ac[:-1], 0.001, 1.0)).to(torch.float64)
    la = alphas.log()
    lca = torch.cumsum(la, dim=-1)
    l1ma = torch.log((1 - la.exp()).clamp_(min=1e-30))
    l1mca = torch.log((1 - lca.exp()).clamp_(min=1e-30))
    return DiffusionTables(la.float(), l1ma.float(), lca.float(), l1mca.float(), num_classes)


def log_add_exp(a: Tensor, b: Tensor) -> Tensor:
    """diffuser.py:22-24."""
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def index_to_log_onehot(x: Tensor, K: int) -> Tensor:
    """diffuser.py:34-47: log(clamp(onehot, 1e-7)) -> {0, log(1e-7)}."""
    assert int(x.max()) < K, f"Error: {int(x.max())} >= {K}"
    return torch.log(F.one_hot(x, K).to(torch.float32).clamp(min=MIN_LOG_ARG))


def gumbel_scores(logp: Tensor, u: Tensor) -> Tensor:
    """diffuser.py:219-228 with the uniforms supplied: the perturbed scores whose arg-max is the sample."""
    g = -torch.log((-torch.log(u.clamp(min=MIN_LOG_ARG))).clamp(min=MIN_LOG_ARG))
    return g + logp


def gumbel_argmax(logp: Tensor, u: Tensor) -> Tensor:
    return gumbel_scores(logp, u).argmax(dim=-1)


def reverse_step(tb: DiffusionTables, logits_c: Tensor, logits_u: Optional[Tensor], x_t: Tensor, x_known: Tensor,
                 m: Tensor, t: int, u1: Tensor, u2: Optional[Tensor], guidance_w: float, temperature: float,
                 return_scores: bool = False):
    """diffuser.py:345-394 for one utterance given the model outputs.  logits (S, 8, K),
    x_t / x_known / m (S, 8), u1/u2 uniforms (S, 8, K).  The 'ensemble' block (:373-378) is
    an exact identity at bs = 1; last_greedy never reaches here (App. B-3).
    ``return_scores``: also return the (S, 8, K) perturbed scores of the unknown and the known branch (None at
    t = 0), so a test can tell a wrong id from a tie that float rounding may legally break either way."""
    K = tb.num_classes
    lnK = math.log(K)      # np.log(num_classes): python double, subtracted from fp32 tensors
    x0 = logits_c
    if guidance_w != 1:
        x0 = guidance_w * logits_c + (1 - guidance_w) * logits_u
    x0 = x0 / temperature
    l0 = F.log_softmax(x0, dim=-1)
    log_x_t = index_to_log_onehot(x_t, K)
    # q_posterior (:176-206)
    tm1 = max(t - 1, 0)
    ev = log_add_exp(l0 + tb.log_cumprod_alpha[tm1], tb.log_1_min_cumprod_alpha[tm1] - lnK)
    if t == 0:
        ev = l0
    one = log_add_exp(log_x_t + tb.log_alpha[t], tb.log_1_min_alpha[t] - lnK)
    un = ev + one
    logp = un - torch.logsumexp(un, dim=-1, keepdim=True)
    s_unk = gumbel_scores(logp, u1)
    unk = s_unk.argmax(dim=-1)
    s_kn = None
    if t == 0:
        kn = x_known
    else:
        lk = index_to_log_onehot(x_known, K)
        s_kn = gumbel_scores(log_add_exp(lk + tb.log_cumprod_alpha[t], tb.log_1_min_cumprod_alpha[t] - lnK), u2)
        kn = s_kn.argmax(dim=-1)
    out = kn * m.long() + unk * (1 - m.long())
    return (out, s_unk, s_kn) if return_scores else out


@dataclass
class NARParams:
    T: int = 200
    x_0_temp: float = 0.7
    guidance_w: float = 3.0
    deep_clone: bool = True
    q0_override_steps: int = 20


def perform_simple_inference_oracle(sd: Dict[str, Tensor], nhead: int, c_text: Tensor, c_codes: Tensor, x_l0: Tensor,
                                    p: NARParams, generator: Optional[torch.Generator] = None,
                                    n_steps: Optional[int] = None, record: Optional[list] = None,
                                    hoist: bool = True) -> Tensor:
    """diffuser.py:398-472 at the shipped jump_len = jump_n_sample = 1 (schedule T-1..0,
    every step a reverse step).  c_text (Lt,), c_codes (Lc, 8), x_l0 (Lx,) L0 codes handed
    over by the AR stage.  Returns (S - offset, 8).  RNG order (App. C): randint(0, K,
    (1, Lx, 8)); per step rand_like (1,S,8,K) for the unknown branch then (t > 0 only) for
    the known branch.  ``hoist``: compute the t-independent speaker vectors once (exact);
    hoist=False recomputes them every forward like the reference (CPU-baseline cost)."""
    tb = diffusion_tables(1025, 200)
    K = tb.num_classes
    Lx = x_l0.shape[0]
    x_quant0 = x_l0.clone()
    x = torch.randint(0, K, (1, Lx, 8), dtype=torch.long, generator=generator)[0]
    x[:, 0] = x_quant0
    x_known = torch.zeros_like(x)
    x_known[:, 0] = x[:, 0]
    m = torch.zeros_like(x).bool()
    m[:, 0] = True
    offset = 0
    if p.deep_clone:                                        # diffuser.py:423-436
        x = torch.cat([c_codes, x], dim=0)
        x_known = torch.cat([c_codes, x_known], dim=0)
        m = torch.cat([torch.ones_like(c_codes).bool(), m], dim=0)
        x_quant0 = torch.cat([c_codes[:, 0], x_quant0], dim=0)
        offset = c_codes.shape[0]
    S = x.shape[0]
    spk_c = nar_spk_vector(sd, c_codes, nhead, False) if hoist else None
    spk_u = nar_spk_vector(sd, c_codes, nhead, True) if hoist else None
    times = list(range(p.T - 1, -1, -1))
    if n_steps is not None:
        times = times[:n_steps]
    for t in times:
        lc = nar_forward(sd, nhead, c_text, c_codes, x, t, False, spk_c)
        lu = nar_forward(sd, nhead, c_text, c_codes, x, t, True, spk_u) if p.guidance_w != 1 else None
        u1 = torch.rand((1, S, 8, K), generator=generator)[0]
        u2 = torch.rand((1, S, 8, K), generator=generator)[0] if t > 0 else None
        x_prev = x
        x, s_unk, s_kn = reverse_step(tb, lc, lu, x, x_known, m, t, u1, u2, p.guidance_w, p.x_0_temp, return_scores=True)
        if p.q0_override_steps < t:
            x[:, 0] = x_quant0
        if record is not None:
            record.append({"t": t, "x_t": x_prev.clone(), "x_tm1": x.clone(), "u1": u1, "u2": u2, "s_unk": s_unk, "s_kn": s_kn,
                           "m": m, "x_known": x_known})
    return x[offset:]


# ======================================================================================
# tts_core: the index bookkeeping of inference.py:235-301 without Encodec / Vocos
# ======================================================================================
@dataclass
class Prompt:
    prompt: Tensor            # (P,) AR prompt, global ids
    first_codec_idx: int
    text_tokens: List[int]
    n_speech_inp: int


def build_prompt(text_ids: List[int], text_ids_full: List[int], speech_tokens: List[int], n_text: int, deep_clone: bool) -> Prompt:
    """inference.py:243-255."""
    offset_codes = [s + n_text for s in speech_tokens]
    if not deep_clone:
        offset_codes, n_speech_inp, text_tokens = [], 0, text_ids
    else:
        n_speech_inp, text_tokens = len(offset_codes), text_ids_full
    prompt = torch.tensor(text_tokens + offset_codes, dtype=torch.long)
    return Prompt(prompt, prompt.shape[-1] - n_speech_inp + 1, text_tokens, n_speech_inp)


def parse_ar_output(ar_codes: Tensor, n_text: int, first_codec_idx: int, expansion: List[List[int]]) -> Tensor:
    """inference.py:272-275: subtract the text offset, clamp >= 0, drop through
    first_codec_idx (this also drops the FIRST speech token, App. B-2), BPE-decode to L0
    codes, keep ints (special tokens decode to strings and are filtered out)."""
    toks = (ar_codes - n_text).clamp(min=0)[first_codec_idx:].tolist()
    frames: List[int] = []
    for tk in toks:
        frames.extend(expansion[tk])
    return torch.tensor(frames, dtype=torch.long)


# ----------------------------------------------------------------------------- trim test inputs
def trim_test_waves() -> List[Tensor]:
    """Deterministic 24 kHz waveforms for the silence-trim fixture (``tests/golden/trim_cases.npz`` holds what the
    reference's ``mars5/trim.py:110-178`` returns for them; generator: ``oracle/gen_golden.py --only trim``)."""
    g = torch.Generator().manual_seed(77)
    sr = 24000
    t = torch.arange(sr * 2) / sr
    waves = []
    w = torch.zeros(sr * 2)
    w[9000:30000] = 0.5 * torch.sin(2 * math.pi * 220 * t[9000:30000])
    waves.append(w)                                                            # silence - tone - silence
    w = 1e-3 * torch.randn(sr * 2, generator=g)
    w[20000:26000] += 0.8 * torch.sin(2 * math.pi * 440 * t[20000:26000])
    waves.append(w)                                                            # noise floor 58 dB below a burst
    waves.append(torch.randn(sr * 2, generator=g) * torch.exp(-t * 6))          # decaying noise (trailing trim only)
    waves.append(torch.zeros(5000))                                            # all zeros -> empty
    waves.append(0.3 * torch.randn(3000, generator=g))                          # shorter than two frames
    waves.append(torch.stack([waves[0], 0.5 * waves[1]]))                      # stereo: mean over channels decides
    return waves


# ----------------------------------------------------------------------------- tokenizer test inputs
TOKENIZER_TEST_STRINGS = [
    "The quick brown rat.",
    "We actually haven't managed to meet demand this year, I'm told; they'd've tried!",
    "  leading and   multiple   spaces\tand\ttabs\nnew lines\r\n  ",
    "Numbers 1 22 333 4444 55555 and 3.14159, 1,000,000; 2024-08-07 at 12:30pm.",
    "Unicode: naïve café — “quotes” … ellipsis, ß, Ωmega, 北京, 🙂 emoji, e\u0301 combining.",
    "CamelCaseWords and snake_case_words and SHOUTING and x86_64 and C++/C# #hashtag @user http://a.b/c?d=e&f=g",
    "'s 't 're 've 'm 'll 'd contractions at start; O'Neil's it's IT'S",
    "<|startoftext|>special tokens <|endoftext|> inside <|startoftext|> text<|endoftext|>",
    "",
    " ",
    "a",
    "!!!???...,,,;;;:::---___***(((]]]}}}",
]


TOKENIZER_TEST_VOCABS = {"tiny": (30, 63), "full": (2813, 0), "merged": (1790, 1023)}     # (text merges, speech merges)


def tokenizer_test_code_strings() -> List[str]:
    """Codebook-tokenizer inputs: space-separated code strings (what inference.py:237-238 builds from the L0 codes)."""
    g = torch.Generator().manual_seed(31)
    out = []
    for n in (1, 2, 7, 40, 150):
        out.append(" ".join(str(int(v)) for v in torch.randint(0, 1024, (n,), generator=g)))
    from mars5_tts_amd import synth                       # merge-friendly sequences exercise the BPE merges
    out.append(" ".join(str(v) for v in synth.speech_corpus_codes(120, seed=5)))
    out.append(" ".join(str(v) for v in synth.speech_corpus_codes(33, seed=9)))
    return out
