"""Drop-in for the reference's top-level ``inference.py``: ``InferenceConfig`` (same 21 fields and
defaults, reference inference.py:24-77) and ``Mars5TTS`` (same constructor, ``tts``, ``vocode``,
``get_speaker_embedding``, attributes; reference inference.py:79-307).  The AR decode loop and
the NAR diffusion run on the MI355X engine in ``mars5-tts_amd/``; Encodec analysis and Vocos
synthesis stay third-party modules exactly as in the reference (lazy: only needed by the
audio-in / audio-out methods, and injectable for hosts where they are not installed).
"""
from __future__ import annotations

import dataclasses
import io
import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Tuple, Type, Union

import torch
import torch.nn.functional as F
from torch import Tensor

from mars5_tts_amd import _lib as _L
from mars5_tts_amd import ops
from mars5_tts_amd.ar_generate import ar_generate, ar_generate_batch
from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, begin_inference, perform_batch_inference, perform_simple_inference
from mars5_tts_amd.minbpe import GPT4_SPLIT_PATTERN, CodebookTokenizer, RegexTokenizer
from mars5_tts_amd.model import CodecLM, ResidualTransformer
from mars5_tts_amd.trim import trim, trim_device


@dataclass
class InferenceConfig():
    """Inference settings: the 21 fields, names and defaults of the reference's ``InferenceConfig`` (inference.py:24-77)."""
    ## >>>> AR CONFIG
    temperature: float = 0.7
    top_k: int = 200          # 0 disables it
    top_p: float = 0.2        # 1.0 disables it
    typical_p: float = 1.0
    freq_penalty: float = 3
    presence_penalty: float = 0.4
    rep_penalty_window: int = 80
    eos_penalty_decay: float = 0.5
    eos_penalty_factor: float = 1
    eos_estimated_gen_length_factor: float = 1.0
    ## >>>> NAR CONFIG
    timesteps: int = 200      # unused by the reference too: T is fixed at default_T (inference.py:113,286)
    x_0_temp: float = 0.7
    q0_override_steps: int = 20
    nar_guidance_w: float = 3
    max_prompt_dur: float = 12
    generate_max_len_override: int = -1
    deep_clone: bool = True
    use_kv_cache: bool = True
    trim_db: float = 27
    beam_width: int = 1
    ref_audio_pad: float = 0


@dataclass
class ReferenceHandle:
    """What depends only on the reference audio (and, per text, on the text) -- computed once by
    ``Mars5TTS.prepare_reference`` and reused by every request that clones the same voice (reference
    inference.py:174-199, mars5/model.py:71-92,246-261 recompute all of it per call):
    the reference's BPE speech tokens, the AR and NAR speaker vectors, and a small LRU of NAR conditioning states
    (text encoder output for all 200 steps + cross-attention K / V of 16 layers, ~1 GB each) keyed by the text ids."""
    prompt_codec: Tensor                      # (1, n_q, Lc) on the device
    ref_transcript: Optional[str]
    speech_tokens: List[int]
    ar_spk: Tensor                            # (dim_ar,) fp32
    nar_spk: Tensor                           # (dim_nar,) fp32
    max_cond: int = 2
    cond: Dict[tuple, object] = dataclasses.field(default_factory=dict)     # (text ids, cfg key) -> NARSession holding the state


class Mars5TTS:
    def __init__(self, ar_ckpt, nar_ckpt, device: str = None, codec=None, vocos=None) -> None:
        if device is None:
            device = 'cuda' if torch.cuda.is_available() else 'cpu'
        self.device = torch.device(device)
        self.codec = codec if codec is not None else _load_encodec(self.device)   # pass False to skip loading
        self.texttok = RegexTokenizer(GPT4_SPLIT_PATTERN)
        self.texttok.load(io.BytesIO(ar_ckpt['vocab']['texttok.model'].encode('utf-8')))
        self.speechtok = CodebookTokenizer(GPT4_SPLIT_PATTERN)
        self.speechtok.load(io.BytesIO(ar_ckpt['vocab']['speechtok.model'].encode('utf-8')))
        self.n_vocab = len(self.texttok.vocab) + len(self.speechtok.vocab)
        self.n_text_vocab = len(self.texttok.vocab) + 1
        self.diffusion_n_classes: int = 1025
        self.codeclm = CodecLM(n_vocab=self.n_vocab, dim=1536, dim_ff_scale=7/3)
        self.codeclm.load_state_dict(ar_ckpt['model'])
        self.codeclm = self.codeclm.to(self.device).eval()
        self.codecnar = ResidualTransformer(n_text_vocab=self.n_text_vocab, n_quant=self.diffusion_n_classes,
                                            p_cond_drop=0, dropout=0)
        self.codecnar.load_state_dict(nar_ckpt['model'])
        self.codecnar = self.codecnar.to(self.device).eval()
        self.default_T = 200
        self.sr = 24000
        self.latent_sr = 75
        self.vocos = vocos if vocos is not None else _load_vocos(self.device)
        self._expansion = self.speechtok.expansion_table()
        self._exp_csr = None                 # device copy of the expansion table, built on first use

    # ------------------------------------------------------------------ hub loading
    @classmethod
    def from_pretrained(cls, model_id: str, device: str = None, revision=None, cache_dir=None, force_download=False,
                        proxies=None, local_files_only=False, token=None, **kw) -> "Mars5TTS":
        return cls._from_pretrained(model_id=model_id, revision=revision, cache_dir=cache_dir, force_download=force_download,
                                    proxies=proxies, local_files_only=local_files_only, token=token, device=device, **kw)

    @classmethod
    def _from_pretrained(cls: Type["Mars5TTS"], *, model_id: str, revision: Optional[str], cache_dir: Optional[Union[str, Path]],
                         force_download: bool, proxies: Optional[Dict], local_files_only: bool, token: Optional[Union[str, bool]],
                         device: str = None, **model_kwargs) -> "Mars5TTS":
        from huggingface_hub import hf_hub_download
        from safetensors import safe_open
        ckpts = []
        for fname in ("mars5_ar.safetensors", "mars5_nar.safetensors"):
            path = hf_hub_download(repo_id=model_id, filename=fname, revision=revision, cache_dir=cache_dir,
                                   force_download=force_download, proxies=proxies, local_files_only=local_files_only, token=token)
            ck = {'model': {}}
            with safe_open(path, framework='pt', device='cpu') as f:
                md = f.metadata()
                ck['vocab'] = {'texttok.model': md['texttok.model'], 'speechtok.model': md['speechtok.model']}
                for k in f.keys():
                    ck['model'][k] = f.get_tensor(k)
            ckpts.append(ck)
        return cls(ar_ckpt=ckpts[0], nar_ckpt=ckpts[1], device=device, **model_kwargs)

    # ------------------------------------------------------------------ audio ends (third party)
    @torch.inference_mode()
    def vocode(self, tokens: Tensor) -> Tensor:
        """(seq_len, n_q) Encodec codes -> waveform (1, T), through the injected Vocos model (reference inference.py:160-172)."""
        return self._vocode_device(tokens).cpu().squeeze()[None]

    def _vocode_device(self, tokens: Tensor) -> Tensor:
        _need(self.vocos, "vocos")
        tokens = tokens.T.to(self.device)
        features = self.vocos.codes_to_features(tokens)
        bandwidth_id = torch.tensor([1], device=self.device)
        return self.vocos.decode(features, bandwidth_id=bandwidth_id)

    @torch.inference_mode()
    def get_speaker_embedding(self, ref_audio: Tensor) -> Tensor:
        """Speaker vector (bs, dim) of the AR model's reference encoder for `ref_audio` (bs, T) (reference inference.py:174-199)."""
        _need(self.codec, "encodec")
        if ref_audio.dim() == 1:
            ref_audio = ref_audio[None]
        spk_reference = self.codec.encode(ref_audio[None].to(self.device))[0][0].permute(0, 2, 1)
        return self.codeclm.get_spk_embedding(spk_reference)

    # ------------------------------------------------------------------ per-reference cache
    @torch.inference_mode()
    def prepare_reference(self, prompt_codec: Tensor, ref_transcript: Optional[str] = None, max_cond: int = 2) -> ReferenceHandle:
        """Everything a request needs from the reference alone.  prompt_codec (1, n_q, Lc) Encodec codes of the reference
        audio (``codec.encode(audio)[0][0]``).  Pass the handle to ``tts_from_codes(..., ref_handle=h)``: results are
        identical to calls without it (the same kernels run on the same inputs, only once)."""
        prompt_codec = prompt_codec.to(self.device)
        q0_str = ' '.join([str(t) for t in prompt_codec[0, 0].tolist()])
        speech_tokens = self.speechtok.encode(q0_str.strip())
        codes = prompt_codec[0].T.contiguous()
        ar_spk = self.codeclm.engine().spk(codes)
        nar_spk = self.codecnar.engine().spk(codes)
        torch.cuda.current_stream(self.device).synchronize()
        return ReferenceHandle(prompt_codec, ref_transcript, speech_tokens, ar_spk, nar_spk, max_cond)

    # ------------------------------------------------------------------ the hot path
    def _prompt(self, text: str, prompt_codec: Tensor, ref_transcript: Optional[str], cfg: InferenceConfig,
                ref_handle: Optional[ReferenceHandle] = None) -> dict:
        """Prompt construction of reference inference.py:222-258: tokenise, then ``_prompt_from_ids``."""
        # both tokenisations are built unconditionally, as in the reference (so ref_transcript=None raises TypeError even
        # for a shallow clone)
        text_tokens = self.texttok.encode("<|startoftext|>" + text.strip() + "<|endoftext|>", allowed_special='all')
        text_tokens_full = self.texttok.encode("<|startoftext|>" + ref_transcript + ' ' + str(text).strip() + "<|endoftext|>",
                                               allowed_special='all')
        if cfg.deep_clone:
            text_tokens = text_tokens_full
        return self._prompt_from_ids(text_tokens, prompt_codec, round(cfg.eos_estimated_gen_length_factor * len(text)), cfg, ref_handle)

    def _prompt_from_ids(self, text_tokens: List[int], prompt_codec: Tensor, n_phones_gen: int, cfg: InferenceConfig,
                         ref_handle: Optional[ReferenceHandle] = None) -> dict:
        """The AR prompt from already tokenised text (deep clone: transcript + text, else text alone) and the reference
        codes (1, n_q, Lc): text ids, then (deep clone only) the BPE tokens of the reference's codebook-0 codes offset by
        the text vocabulary (reference inference.py:235-258)."""
        text_tokens = [int(t) for t in text_tokens]
        if ref_handle is not None:
            prompt_codec, speech_tokens = ref_handle.prompt_codec, ref_handle.speech_tokens
        else:
            prompt_codec = prompt_codec.to(self.device)
            q0_str = ' '.join([str(t) for t in prompt_codec[0, 0].tolist()])
            speech_tokens = self.speechtok.encode(q0_str.strip())
        spk_ref_codec = prompt_codec[0, :, :].T
        n_text = len(self.texttok.vocab)
        offset_speech_codes = [p + n_text for p in speech_tokens] if cfg.deep_clone else []
        n_speech_inp = len(offset_speech_codes)
        prompt = torch.tensor(text_tokens + offset_speech_codes, dtype=torch.long, device=self.device)
        return dict(prompt=prompt, first_codec_idx=prompt.shape[-1] - n_speech_inp + 1, spk_ref_codec=spk_ref_codec,
                    text_tokens=text_tokens, prompt_codec=prompt_codec, n_text=n_text, n_phones_gen=int(n_phones_gen), ref_handle=ref_handle)

    def _ar_kwargs(self, cfg: InferenceConfig) -> dict:
        return dict(max_len=cfg.generate_max_len_override if cfg.generate_max_len_override > 1 else 2000,
                    temperature=cfg.temperature, topk=cfg.top_k, top_p=cfg.top_p, typical_p=cfg.typical_p,
                    alpha_frequency=cfg.freq_penalty, alpha_presence=cfg.presence_penalty,
                    penalty_window=cfg.rep_penalty_window, eos_penalty_decay=cfg.eos_penalty_decay,
                    eos_penalty_factor=cfg.eos_penalty_factor)

    def _handoff(self, pr: dict, ar_codes: Tensor, cfg: InferenceConfig):
        """AR -> NAR hand-off (reference inference.py:262-285): token ids -> L0 frames through the BPE
        expansion table on the device (same result as speechtok.decode_int on the id list, inference.py:272-275),
        then the ``perform_simple_inference`` batch tuple."""
        # on device: one kernel through the CSR expansion table + a 4-byte read-back of the frame count (m5_expand_tokens)
        if getattr(self, "_exp_csr", None) is None:
            off, vals, mx = self.speechtok.expansion_csr()
            self._exp_csr = (off.to(self.device), vals.to(self.device), mx)
        off, vals, mx = self._exp_csr
        gen_codes_decoded = ops.expand_tokens(ar_codes[pr["first_codec_idx"]:].to(self.device).contiguous(), pr["n_text"], off, vals, mx)
        text_tokens, prompt_codec = pr["text_tokens"], pr["prompt_codec"]
        c_text = torch.tensor(text_tokens, dtype=torch.long, device=self.device)[None]
        c_codes = prompt_codec.permute(0, 2, 1)
        c_texts_lengths = torch.tensor([len(text_tokens)], dtype=torch.long, device=self.device)
        c_codes_lengths = torch.tensor([c_codes.shape[1]], dtype=torch.long, device=self.device)
        _x = gen_codes_decoded[None, :, None].repeat(1, 1, 8)
        x_padding_mask = torch.zeros((1, _x.shape[1]), dtype=torch.bool, device=_x.device)
        skip_front = prompt_codec.shape[-1] if cfg.deep_clone else 0
        return gen_codes_decoded, (c_text, c_codes, c_texts_lengths, c_codes_lengths, _x, x_padding_mask), skip_front

    def _ar_stage(self, text: str, prompt_codec: Tensor, ref_transcript: Optional[str], cfg: InferenceConfig,
                  ar_noise: Optional[Tensor] = None, generator: Optional[torch.Generator] = None):
        """Prompt construction + AR decode + BPE expansion (reference inference.py:222-285).
        Returns (L0 frames (G,), the ``perform_simple_inference`` batch tuple, frames to skip in front)."""
        return self._ar_stage_pr(self._prompt(text, prompt_codec, ref_transcript, cfg), cfg, ar_noise, generator)

    def _ar_stage_pr(self, pr: dict, cfg: InferenceConfig, ar_noise: Optional[Tensor] = None, generator: Optional[torch.Generator] = None):
        """``_ar_stage`` for a prompt that is already built (``_prompt`` / ``_prompt_from_ids``)."""
        ar_codes = ar_generate(self.texttok, self.speechtok, self.codeclm, pr["prompt"], pr["spk_ref_codec"], pr["first_codec_idx"],
                               fp16=True if torch.cuda.is_available() else False, beam_width=cfg.beam_width, beam_length_penalty=1,
                               n_phones_gen=pr["n_phones_gen"], vocode=False, use_kv_cache=cfg.use_kv_cache, noise=ar_noise,
                               generator=generator, **self._ar_kwargs(cfg))
        return self._handoff(pr, ar_codes, cfg)

    def _dsh(self, cfg: InferenceConfig) -> DSH:
        return DSH(last_greedy=True, x_0_temp=cfg.x_0_temp, guidance_w=cfg.nar_guidance_w, deep_clone=cfg.deep_clone,
                   jump_len=1, jump_n_sample=1, q0_override_steps=cfg.q0_override_steps,
                   enable_kevin_scaled_inference=True, progress=False)

    @torch.inference_mode()
    def tts_from_codes(self, text: str, prompt_codec: Tensor, ref_transcript: Optional[str],
                       cfg: InferenceConfig = InferenceConfig(), ar_noise: Optional[Tensor] = None,
                       generator: Optional[torch.Generator] = None, rng_hooks=None,
                       ref_handle: Optional[ReferenceHandle] = None) -> Tuple[Tensor, Tensor]:
        """``tts`` between the codec and the vocoder (reference inference.py:222-301):
        prompt_codec (1, n_q, seq_len) Encodec codes in -> (AR L0 codes, final (S_out, 8) codes) out.
        `ref_handle` (``prepare_reference``): reuse what was computed for this reference (prompt_codec / ref_transcript
        may then be None).
        `rng_hooks` (parity tests): an object with ar_noise(n_steps, V), after_ar(n_iterations), randint(shape),
        uniform(shape) that supplies every random draw instead of the device generator (oracle/fakes.py), and
        optionally nar_on_step(dict), an observer of every reverse step."""
        if ref_handle is not None:
            prompt_codec = ref_handle.prompt_codec
            ref_transcript = ref_handle.ref_transcript if ref_transcript is None else ref_transcript
        return self._tts_core(self._prompt(text, prompt_codec, ref_transcript, cfg, ref_handle), cfg, ar_noise, generator, rng_hooks)

    @torch.inference_mode()
    def tts_from_ids(self, text_ids, prompt_codec: Tensor, n_phones_gen: int, cfg: InferenceConfig = InferenceConfig(),
                     generator: Optional[torch.Generator] = None) -> Tuple[Tensor, Tensor]:
        """``tts_from_codes`` for a request that arrives already tokenised -- the wire format of the multi-GPU request
        scatter (``mars5_tts_amd.sharding.Request``): `text_ids` = ids of "<|startoftext|>[transcript ]text<|endoftext|>",
        `n_phones_gen` = the EOS-penalty length estimate round(cfg.eos_estimated_gen_length_factor * len(text))."""
        ids = text_ids.tolist() if isinstance(text_ids, Tensor) else list(text_ids)
        return self._tts_core(self._prompt_from_ids(ids, prompt_codec, n_phones_gen, cfg), cfg, None, generator, None)

    @torch.inference_mode()
    def tts_stream_from_codes(self, texts: List[str], prompt_codecs: List[Tensor], ref_transcripts: List[Optional[str]],
                              cfg: InferenceConfig = InferenceConfig(), seeds: Optional[List[int]] = None):
        """Pipelined serving of independent requests on one GPU: a generator that yields (L0 codes, final codes) per
        request, in order.  The two stages of consecutive requests overlap -- request i+1's AR decode (latency-bound
        weight streaming on its own stream) runs while request i's 200 NAR steps (compute-bound) are in flight on the NAR
        stream -- which raises throughput, not the latency of a request.  Every request draws from a private device
        generator seeded seeds[i], so result i equals ``torch.manual_seed(seeds[i]); tts_from_codes(...)`` (same property
        as ``tts_batch_from_codes``)."""
        n = len(texts)
        ar_stream, nar_stream = ops.session_stream(self.device, "ar"), ops.session_stream(self.device, "nar")
        pending = None
        for i in range(n):
            g = torch.Generator(device=self.device)
            g.manual_seed(int(seeds[i]) if seeds is not None else int(torch.randint(0, 2 ** 62, (1,)).item()))
            pr = self._prompt(texts[i], prompt_codecs[i], ref_transcripts[i], cfg)
            gen, fin = self._tts_core(pr, cfg, None, g, None, streams=(ar_stream, nar_stream), wait=False)
            if pending is not None:
                yield pending[0], pending[1]()
            pending = (gen, fin)
        if pending is not None:
            yield pending[0], pending[1]()

    def _tts_core(self, pr: dict, cfg: InferenceConfig, ar_noise, generator, rng_hooks, streams=None, wait: bool = True):
        T = self.default_T
        diff = MultinomialDiffusion(self.diffusion_n_classes, timesteps=T, device=self.device)
        if rng_hooks is not None:
            n_steps = self._ar_kwargs(cfg)["max_len"] - int(pr["prompt"].shape[0])
            ar_noise = rng_hooks.ar_noise(max(n_steps, 1), self.n_vocab)
        # the NAR stage's conditioning work (text encoder for all 200 steps, cross-attention K / V) does not depend on the
        # AR output: enqueue it on the NAR stream now, it runs beside the AR decode
        h = pr.get("ref_handle")
        key = (tuple(pr["text_tokens"]), cfg.nar_guidance_w, cfg.x_0_temp, cfg.deep_clone, cfg.q0_override_steps) if h is not None else None
        nar_sess = begin_inference(self.codecnar, torch.tensor(pr["text_tokens"], dtype=torch.long, device=self.device)[None],
                                   pr["prompt_codec"].permute(0, 2, 1), diff.num_timesteps, dsh=self._dsh(cfg), diff=diff,
                                   spk_vec=h.nar_spk if h is not None else None, cond_from=h.cond.get(key) if h is not None else None,
                                   stream=streams[1] if streams else None)
        cache_cond = h is not None and key not in h.cond          # inserted only once the request has completed (below)
        # The AR stage is ordered BEHIND the conditioning on the GPU (an event wait on the AR stream).  The prefill's first operations
        # are small blocking host -> device copies on that stream (ARSession.prefill: state words, prompt ids), so the host too waits
        # for the conditioning there (~6 ms; it has nothing else to enqueue for this request).  History (round 5): until then a pageable host -> device copy at the END of prepare_cond had
        # kept the host -- and so every AR launch -- waiting for the conditioning by accident.  With the copies hoisted the two stages
        # really overlapped on their streams, and the fp32 `tts()` test died with a GPU memory fault: the text ids, a temporary of the
        # CURRENT stream that this function drops as soon as begin_inference returns, were still to be gathered on the NAR stream when
        # torch's caching allocator handed their block to the next allocation (fixed at the root: ops.use_on / record_stream on every
        # tensor that crosses streams; tools/guard_tts_fp32.py reproduces it with M5_TTS_STAGE_ORDER=0 on the old tree).  The
        # ordering stays because overlapping buys ~1 ms of 850 (the 5-6 ms of conditioning compete with the prefill for the same CUs).
        if streams is None and self.device.type == "cuda" and getattr(nar_sess, "cond_ready", None) is not None and \
                _L.tool_knob("M5_TTS_STAGE_ORDER", "1") != "0":      # (tools configuration: 0 = let the stages overlap, to reproduce that abort)
            ops.session_stream(self.device, "ar").wait_event(nar_sess.cond_ready)
        ar_codes = ar_generate(self.texttok, self.speechtok, self.codeclm, pr["prompt"], pr["spk_ref_codec"], pr["first_codec_idx"],
                               fp16=True if torch.cuda.is_available() else False, beam_width=cfg.beam_width, beam_length_penalty=1,
                               n_phones_gen=pr["n_phones_gen"], vocode=False, use_kv_cache=cfg.use_kv_cache, noise=ar_noise,
                               generator=generator, spk_vec=h.ar_spk if h is not None else None, stream=streams[0] if streams else None,
                               **self._ar_kwargs(cfg))
        gen_codes_decoded, batch, skip_front = self._handoff(pr, ar_codes, cfg)
        hook_kw = {}
        if rng_hooks is not None:
            n_gen = int(ar_codes.shape[0]) - int(pr["prompt"].shape[0])
            rng_hooks.after_ar(n_gen + (1 if int(ar_codes.shape[0]) < self._ar_kwargs(cfg)["max_len"] else 0))
            hook_kw = dict(uniform=rng_hooks.uniform, randint=rng_hooks.randint, on_step=getattr(rng_hooks, "nar_on_step", None))
        final_output = perform_simple_inference(self.codecnar, batch, diff, diff.num_timesteps, torch.float16, dsh=self._dsh(cfg),
                                                retain_quant0=True, generator=generator, session=nar_sess, wait=wait, **hook_kw)
        def remember_cond():       # the conditioning joins the handle's cache only after the request that built it went through
            if cache_cond and key not in h.cond and h.max_cond > 0:
                while len(h.cond) >= max(h.max_cond, 1):
                    h.cond.pop(next(iter(h.cond)))             # oldest first
                h.cond[key] = nar_sess

        if not wait:               # pipelined serving: the NAR steps are in flight; the caller collects them later
            def collect():
                out = final_output()[0, skip_front:].to(self.device)
                remember_cond()
                return out
            return gen_codes_decoded, collect
        final_output = final_output[0, skip_front:].to(self.device)
        remember_cond()
        return gen_codes_decoded, final_output

    @torch.inference_mode()
    def tts_batch_from_codes(self, texts: List[str], prompt_codecs: List[Tensor], ref_transcripts: List[Optional[str]],
                             cfg: InferenceConfig = InferenceConfig(), seeds: Optional[List[int]] = None,
                             nar_batch: int = 32, ar_batch: int = 1, max_lens: Optional[List[int]] = None,
                             nar_in_flight: int = 2, prompts: Optional[List[dict]] = None) -> List[Tuple[Tensor, Tensor]]:
        """Several independent requests on one GPU (BASELINE config 3).  Request i gets a private
        device generator seeded ``seeds[i]`` and consumes it as a lone call would.
        NAR: up to `nar_batch` requests of similar length (default 32 since round 4: a group is laid out padded to its longest
        member but only the row tiles that hold real rows are launched, so large groups are the efficient ones; each request
        holds ~1 GB of hoisted conditioning while its group runs) are refined per decoder pass
        (``perform_batch_inference``) - exact: result i equals ``torch.manual_seed(seeds[i]);
        tts_from_codes(...)`` whatever else is in the batch.
        AR: `ar_batch` = 1 decodes request by request (the batch-1 weight-streaming GEMV path, bit-equal to
        the lone call); `ar_batch` > 1 decodes that many requests per step (``ar_generate_batch``: the weights
        are read once per step for all of them; logits then differ from the lone call by GEMM summation
        order, like any batch-size change does in the reference).
        `nar_in_flight`: how many NAR groups are refined at once, each on its own stream (a group's 200 dependent step
        graphs leave per-launch bubbles that another group's launches fill, and the host prepares the next group while the
        previous ones run; results do not depend on it).
        `prompts`: prompts that are already built (``_prompt_from_ids``: requests that arrive tokenised, ``tts_batch_from_ids``);
        texts / prompt_codecs / ref_transcripts are then unused."""
        n = len(prompts) if prompts is not None else len(texts)
        assert prompts is not None or (len(prompt_codecs) == n and len(ref_transcripts) == n)
        pr_of = (lambda i: prompts[i]) if prompts is not None else (lambda i: self._prompt(texts[i], prompt_codecs[i], ref_transcripts[i], cfg))
        gens = []
        for i in range(n):
            g = torch.Generator(device=self.device)
            # no seeds given: derive them from the global CPU generator (honours torch.manual_seed, and -- unlike
            # torch.seed() -- leaves every global generator seeded as the caller left it)
            g.manual_seed(int(seeds[i]) if seeds is not None else int(torch.randint(0, 2 ** 62, (1,)).item()))
            gens.append(g)
        def cfg_of(i):
            return cfg if max_lens is None else dataclasses.replace(cfg, generate_max_len_override=int(max_lens[i]))

        if ar_batch <= 1:
            staged = [self._ar_stage_pr(pr_of(i), cfg_of(i), None, gens[i]) for i in range(n)]
        else:
            assert cfg.beam_width == 1, "Only beam size of 1 is currently supported."
            prs = [pr_of(i) for i in range(n)]
            staged = [None] * n
            order = sorted(range(n), key=lambda i: prs[i]["prompt"].shape[0])
            for g0 in range(0, n, min(ar_batch, 32)):
                grp = order[g0:g0 + min(ar_batch, 32)]
                kw = self._ar_kwargs(cfg)
                kw["max_len"] = [self._ar_kwargs(cfg_of(i))["max_len"] for i in grp]
                outs = ar_generate_batch(self.texttok, self.speechtok, self.codeclm, [prs[i]["prompt"] for i in grp],
                                         [prs[i]["spk_ref_codec"] for i in grp], [prs[i]["first_codec_idx"] for i in grp],
                                         n_phones_gens=[prs[i]["n_phones_gen"] for i in grp], generators=[gens[i] for i in grp], **kw)
                for i, o in zip(grp, outs):
                    staged[i] = self._handoff(prs[i], o, cfg)
        T = self.default_T
        diff = MultinomialDiffusion(self.diffusion_n_classes, timesteps=T, device=self.device)
        # group requests of similar total NAR length: the batch is padded to its longest member
        order = sorted(range(n), key=lambda i: staged[i][1][4].shape[1] + staged[i][2])
        finals: List[Optional[Tensor]] = [None] * n
        flying: List[tuple] = []                     # (group, callable that waits for it), oldest first
        lanes = ([ops.session_stream(self.device, "nar" if k == 0 else f"nar_lane{k}") for k in range(max(1, nar_in_flight))]
                 if self.device.type == "cuda" else [None])

        def land():
            grp, wait_for = flying.pop(0)
            for i, o in zip(grp, wait_for()):
                finals[i] = o[0, staged[i][2]:].to(self.device)

        for g0 in range(0, n, max(1, nar_batch)):
            grp = order[g0:g0 + max(1, nar_batch)]
            if len(flying) >= max(1, nar_in_flight):
                land()
            lane = lanes[(g0 // max(1, nar_batch)) % len(lanes)]       # a lane's previous group has landed: groups land oldest first
            flying.append((grp, perform_batch_inference(self.codecnar, [staged[i][1] for i in grp], diff, diff.num_timesteps,
                                                        dsh=self._dsh(cfg), generators=[gens[i] for i in grp], wait=False, stream=lane)))
        while flying:
            land()
        return [(staged[i][0], finals[i]) for i in range(n)]

    @torch.inference_mode()
    def tts_batch_from_ids(self, text_ids: List, prompt_codecs: List[Tensor], n_phones_gens: List[int],
                           cfg: InferenceConfig = InferenceConfig(), **kw) -> List[Tuple[Tensor, Tensor]]:
        """``tts_batch_from_codes`` for requests that arrive already tokenised (the wire format of the multi-GPU request scatter,
        ``mars5_tts_amd.sharding.Request``; see ``tts_from_ids``): a rank refines its whole shard in NAR groups instead of request
        by request.  With ``ar_batch=1`` (the default) result i is bit-identical to ``torch.manual_seed(seeds[i]); tts_from_ids(...)``."""
        prompts = [self._prompt_from_ids(t.tolist() if isinstance(t, Tensor) else list(t), prompt_codecs[i], int(n_phones_gens[i]), cfg)
                   for i, t in enumerate(text_ids)]
        return self.tts_batch_from_codes(None, None, None, cfg, prompts=prompts, **kw)

    @torch.inference_mode()
    def tts(self, text: str, ref_audio: Tensor, ref_transcript: Optional[str] = None,
            cfg: Optional[InferenceConfig] = InferenceConfig(), rng_hooks=None) -> Tuple[Tensor, Tensor]:
        """Speak `text` in the voice of `ref_audio` ((T,) samples at 24 kHz; `ref_transcript` = what it says, required
        for deep clone).  Returns (L0 codes of the generated frames (seq_len,), waveform (T_out,) at 24 kHz) like the
        reference ``tts`` (inference.py:201-307), with the same AssertionError / warning conditions."""
        if cfg.deep_clone and ref_transcript is None:
            raise AssertionError("cfg.deep_clone=True needs `ref_transcript` (the words spoken in `ref_audio`); pass it, or use "
                                 "InferenceConfig(deep_clone=False) for a shallow clone")
        ref_dur = ref_audio.shape[-1] / self.sr
        if ref_dur > cfg.max_prompt_dur:
            logging.warning(f"reference audio is {ref_dur:.2f} s long, above cfg.max_prompt_dur = {cfg.max_prompt_dur} s: quality "
                            f"usually drops; a shorter reference is recommended")
        _need(self.codec, "encodec")
        if ref_audio.dim() == 1:
            ref_audio = ref_audio[None]
        if ref_audio.shape[0] != 1:
            ref_audio = ref_audio.mean(dim=0, keepdim=True)
        ref_audio = F.pad(ref_audio, (int(self.sr * cfg.ref_audio_pad), 0))
        prompt_codec = self.codec.encode(ref_audio[None].to(self.device))[0][0]
        gen_codes_decoded, final_output = self.tts_from_codes(text, prompt_codec, ref_transcript, cfg, rng_hooks=rng_hooks)
        # vocoder output stays on the device for the silence trim (m5_trim_bounds); only the trimmed waveform goes to the host
        final_audio = self._vocode_device(final_output).squeeze()
        if final_audio.device.type == "cuda":
            final_audio, _ = trim_device(final_audio.to(torch.float32), top_db=cfg.trim_db)
        else:
            final_audio, _ = trim(final_audio, top_db=cfg.trim_db)
        return gen_codes_decoded, final_audio.cpu()


def _need(obj, name):
    if obj is None or obj is False:
        raise RuntimeError(f"the third-party '{name}' model is not available on this host; install it or pass an instance "
                           f"to Mars5TTS(..., codec=..., vocos=...)")


def _load_encodec(device):
    try:
        from encodec import EncodecModel
    except ImportError:
        logging.warning("encodec is not installed: Mars5TTS.tts()/get_speaker_embedding() need it (tts_from_codes does not)")
        return None
    from mars5_tts_amd.trim import nuke_weight_norm
    codec = EncodecModel.encodec_model_24khz().to(device).eval()
    codec.set_target_bandwidth(6.0)
    nuke_weight_norm(codec)
    return codec


def _load_vocos(device):
    try:
        from vocos import Vocos
    except ImportError:
        logging.warning("vocos is not installed: Mars5TTS.tts()/vocode() need it (tts_from_codes does not)")
        return None
    from mars5_tts_amd.trim import nuke_weight_norm
    vocos = Vocos.from_pretrained("charactr/vocos-encodec-24khz").to(device).eval()
    nuke_weight_norm(vocos)
    return vocos
