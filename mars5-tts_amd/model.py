"""Parameter containers with the reference's class names, constructor arguments and
state-dict contract (reference ``mars5/model.py:42-141`` ``CodecLM``, ``:163-343``
``ResidualTransformer``).  They hold the checkpoint tensors and lazily build the packed
MI355X engines (``ARModel`` / ``NARModel``); no arithmetic happens here.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .ops import DT_NAME
from .synth import ARShape, NARShape


SAMPLER_MAX_VOCAB = 8192             # m5_ar_sample (csrc/ar_decode.hip): V2 * 12 B of LDS, V2 = next power of two
SAMPLER_MAX_VOCAB_TYPICAL = 4096     # with typical_p < 1 (a second sort: V2 * 20 B)


def default_engine_dtype() -> torch.dtype:
    """GEMM operand dtype of the engines: env MARS5_DTYPE in {f16 (default), bf16, f32 (exact-fp32 parity mode)}.
    f16 is what the reference computes in on a GPU (autocast, ar_generate.py:59,67) and what the published MARS5
    checkpoints are stored in (reference README.md:163-164), so loading them rounds nothing; bf16 (same MFMA rate,
    3 mantissa bits fewer) is BASELINE.json's stated dtype and what bench.py selects explicitly."""
    return DT_NAME[os.environ.get("MARS5_DTYPE", "f16")]


class _Container:
    def __init__(self):
        self._sd: Dict[str, torch.Tensor] = {}
        self.device = torch.device("cpu")
        self.engine_dtype: Optional[torch.dtype] = None
        self._engine = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        expected = self._expected_shapes()
        if strict:
            missing = [k for k in expected if k not in sd]
            unexpected = [k for k in sd if k not in expected]
            bad = [k for k in expected if k in sd and tuple(sd[k].shape) != tuple(expected[k])]
            if missing or unexpected or bad:
                raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: missing {missing[:5]} "
                                   f"unexpected {unexpected[:5]} size mismatch {bad[:5]}")
        self._sd = {k: v.detach() for k, v in sd.items()}
        self._engine = None
        return self

    def state_dict(self):
        return dict(self._sd)

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
        self._engine = None
        return self

    def eval(self):
        return self

    def set_engine_dtype(self, dtype: torch.dtype):
        self.engine_dtype = dtype
        self._engine = None
        return self


class CodecLM(_Container):
    def __init__(self, n_vocab, dim=1536, nhead=24, n_layers=26, n_spk_layers=2, dim_ff_scale=None, sliding_window=3000) -> None:
        super().__init__()
        hidden = int(dim * 4 * (3 / 4)) if dim_ff_scale is None else int(dim * dim_ff_scale)
        self.shape = ARShape(n_vocab=n_vocab, dim=dim, nhead=nhead, n_layers=n_layers, n_spk_layers=n_spk_layers,
                             hidden_dim=hidden, sliding_window=sliding_window)
        self.cfg = SimpleNamespace(vocab_size=n_vocab, dim=dim, n_layers=n_layers, head_dim=64, hidden_dim=hidden, n_heads=nhead,
                                   n_kv_heads=nhead, sliding_window=sliding_window, norm_eps=1e-5)
        self.ar = SimpleNamespace(args=self.cfg)

    def _expected_shapes(self):
        s = self.shape
        D, F, V, FF = s.dim, s.hidden_dim, s.n_vocab, s.spk_ff
        e = {"ar.norm.weight": (D,), "ar.output.weight": (V, D), "embed.weight": (V, D), "pos_embedding.alpha": (1,),
             "spk_identity_emb.weight": (1, D), "spk_encoder.norm.weight": (D,), "spk_encoder.norm.bias": (D,)}
        for l in range(s.n_layers):
            p = f"ar.layers.{l}"
            for w in ("wq", "wk", "wv", "wo"):
                e[f"{p}.attention.{w}.weight"] = (D, D)
            e[f"{p}.feed_forward.w1.weight"] = (F, D)
            e[f"{p}.feed_forward.w3.weight"] = (F, D)
            e[f"{p}.feed_forward.w2.weight"] = (D, F)
            e[f"{p}.attention_norm.weight"] = (D,)
            e[f"{p}.ffn_norm.weight"] = (D,)
        for q in range(8):
            e[f"ref_chunked_emb.embs.{q}.weight"] = (1025, D // 8)
        for l in range(s.n_spk_layers):
            e.update(_enc_layer_shapes(f"spk_encoder.layers.{l}", D, FF, False))
        return e

    def engine(self):
        if self._engine is None:
            from .ar_engine import ARModel
            if self.device.type != "cuda":
                raise RuntimeError("CodecLM engine needs a ROCm GPU device: there is no CPU fallback in mars5_tts_amd")
            if self.shape.n_vocab > SAMPLER_MAX_VOCAB:
                raise RuntimeError(f"n_vocab = {self.shape.n_vocab} exceeds the on-device sampler's limit of {SAMPLER_MAX_VOCAB} "
                                   f"(m5_ar_sample sorts the whole vocabulary in one workgroup's LDS; typical_p < 1 further needs "
                                   f"n_vocab <= {SAMPLER_MAX_VOCAB_TYPICAL})")
            self._engine = ARModel(self._sd, self.shape, self.engine_dtype or default_engine_dtype(), self.device)
        return self._engine

    @torch.inference_mode()
    def get_spk_embedding(self, spk_reference, c_codes_lengths=None) -> torch.Tensor:
        """reference model.py:70-92: (bs=1, seq_len, n_codebooks) -> (1, dim)."""
        if spk_reference.shape[0] != 1:
            raise AssertionError("Speaker embedding extraction only implemented using for bs=1 currently.")
        eng = self.engine()
        out = eng.spk(spk_reference[0].to(eng.dev).contiguous())
        torch.cuda.current_stream().synchronize()
        return out[None]


def _enc_layer_shapes(p, D, FF, cross):
    e = {f"{p}.self_attn.in_proj_weight": (3 * D, D), f"{p}.self_attn.in_proj_bias": (3 * D,),
         f"{p}.self_attn.out_proj.weight": (D, D), f"{p}.self_attn.out_proj.bias": (D,),
         f"{p}.linear2.weight": (D, FF), f"{p}.linear2.bias": (D,),
         f"{p}.norm1.weight": (D,), f"{p}.norm1.bias": (D,), f"{p}.norm2.weight": (D,), f"{p}.norm2.bias": (D,),
         f"{p}.activation.V.weight": (FF, D), f"{p}.activation.W.weight": (FF, D)}
    if cross:
        e.update({f"{p}.multihead_attn.in_proj_weight": (3 * D, D), f"{p}.multihead_attn.in_proj_bias": (3 * D,),
                  f"{p}.multihead_attn.out_proj.weight": (D, D), f"{p}.multihead_attn.out_proj.bias": (D,),
                  f"{p}.norm3.weight": (D,), f"{p}.norm3.bias": (D,)})
    return e


class ResidualTransformer(_Container):
    def __init__(self, n_text_vocab, n_quant=1024, dim=1024, nhead=16, enc_layers=8, dec_layers=16, n_spk_layers=3,
                 c_quant_levels=8, pred_quant_levels=8, t_emb_dim=1024, norm_first=True, p_cond_drop=0.1, dropout=0) -> None:
        super().__init__()
        assert c_quant_levels == 8 and pred_quant_levels == 8 and norm_first
        self.shape = NARShape(n_text_vocab=n_text_vocab, n_quant=n_quant, dim=dim, nhead=nhead, enc_layers=enc_layers,
                              dec_layers=dec_layers, n_spk_layers=n_spk_layers, t_emb_dim=t_emb_dim)
        self.n_quantizer = pred_quant_levels
        self.p_cond_drop = p_cond_drop
        self.t_emb_dim = t_emb_dim

    def _expected_shapes(self):
        s = self.shape
        D, FF, Kq = s.dim, s.dim_ff, s.n_quant
        e = {"cond_pos_embedding.alpha": (1,), "pos_embedding.alpha": (1,), "ref_pos_embedding.alpha": (1,),
             "tfm.encoder.norm.weight": (D,), "tfm.encoder.norm.bias": (D,), "tfm.decoder.norm.weight": (D,),
             "tfm.decoder.norm.bias": (D,), "text_embed.weight": (s.n_text_vocab, D), "spk_identity_emb.weight": (1, D),
             "spk_encoder.norm.weight": (D,), "spk_encoder.norm.bias": (D,)}
        for l in range(s.enc_layers):
            e.update(_enc_layer_shapes(f"tfm.encoder.layers.{l}", D, FF, False))
        for l in range(s.dec_layers):
            e.update(_enc_layer_shapes(f"tfm.decoder.layers.{l}", D, FF, True))
        for l in range(s.n_spk_layers):
            e.update(_enc_layer_shapes(f"spk_encoder.layers.{l}", D, FF, False))
        for w in ("encoder", "decoder"):
            e[f"timestep_{w}_emb.0.weight"] = (D, s.t_emb_dim)
            e[f"timestep_{w}_emb.0.bias"] = (D,)
            e[f"timestep_{w}_emb.2.weight"] = (D, D)
            e[f"timestep_{w}_emb.2.bias"] = (D,)
        for q in range(8):
            e[f"ref_embedder.embs.{q}.weight"] = (Kq, D // 8)
            e[f"residual_encoder.embs.{q}.weight"] = (Kq, D // 8)
            e[f"residual_decoder.{q}.0.weight"] = (D,)
            e[f"residual_decoder.{q}.0.bias"] = (D,)
            e[f"residual_decoder.{q}.1.weight"] = (Kq, D)
            e[f"residual_decoder.{q}.1.bias"] = (Kq,)
        return e

    def engine(self):
        if self._engine is None:
            from .nar_engine import NARModel
            if self.device.type != "cuda":
                raise RuntimeError("ResidualTransformer engine needs a ROCm GPU device: there is no CPU fallback in mars5_tts_amd")
            self._engine = NARModel(self._sd, self.shape, self.engine_dtype or default_engine_dtype(), self.device)
        return self._engine

    @torch.inference_mode()
    def get_spk_embedding(self, c_codes, c_codes_length) -> torch.Tensor:
        """reference model.py:246-261 for bs = 1 without padding."""
        assert c_codes.shape[0] == 1
        eng = self.engine()
        out = eng.spk(c_codes[0, : int(c_codes_length[0])].to(eng.dev).contiguous())
        torch.cuda.current_stream().synchronize()
        return out[None]
