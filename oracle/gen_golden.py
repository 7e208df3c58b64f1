"""Generate the golden fixtures in ``tests/golden/`` by running the UNMODIFIED reference
(``/root/reference``) on seeded synthetic checkpoints.  TEST INFRASTRUCTURE.

Run in the build container only (the reference does not exist on the GPU box):
    python oracle/gen_golden.py [--full]

For every fixture the oracle (``oracle/mars5_oracle.py``) is checked against the reference
on the FULL tensors before the (sub-sampled, small) fixture is written, so a committed
fixture certifies "reference == oracle here" at generation time and lets the tests re-check
the oracle anywhere.  Fixtures record torch version + device in ``meta.json``.
"""
from __future__ import annotations

import argparse
import io
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))   # unused import, ar_generate.py:3

import warnings
warnings.filterwarnings("ignore")

from mars5.ar_generate import ar_generate                      # noqa: E402  (reference)
from mars5.diffuser import DSH, MultinomialDiffusion, perform_simple_inference  # noqa: E402
import mars5.diffuser as ref_diffuser                           # noqa: E402
from mars5.minbpe.codebook import CodebookTokenizer as RefCodebookTok  # noqa: E402
from mars5.minbpe.regex import GPT4_SPLIT_PATTERN, RegexTokenizer as RefRegexTok  # noqa: E402
from mars5.model import CodecLM, ResidualTransformer           # noqa: E402
from mars5.samplers import apply_typical_p, early_eos_penalty, freq_rep_penalty, top_k_top_p_filtering  # noqa: E402

import mars5_oracle as O                                        # noqa: E402
from mars5_tts_amd import synth                                 # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TEXT = "The quick brown rat."
TRANSCRIPT = "We actually haven't managed to meet demand."


def ref_tokenizers(vocab):
    tt = RefRegexTok(GPT4_SPLIT_PATTERN)
    tt.load(io.BytesIO(vocab["texttok.model"].encode("utf-8")))
    st = RefCodebookTok(GPT4_SPLIT_PATTERN)
    st.load(io.BytesIO(vocab["speechtok.model"].encode("utf-8")))
    return tt, st


def ref_models(b: synth.SynthBundle):
    a, n = b.ar_shape, b.nar_shape
    lm = CodecLM(n_vocab=a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                 dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    assert lm.cfg.hidden_dim == a.hidden_dim, (lm.cfg.hidden_dim, a.hidden_dim)
    lm.load_state_dict(b.ar_ckpt["model"], strict=True)
    nar = ResidualTransformer(n_text_vocab=n.n_text_vocab, n_quant=n.n_quant, dim=n.dim, nhead=n.nhead,
                              enc_layers=n.enc_layers, dec_layers=n.dec_layers, n_spk_layers=n.n_spk_layers,
                              t_emb_dim=n.t_emb_dim, p_cond_drop=0, dropout=0)
    nar.load_state_dict(b.nar_ckpt["model"], strict=True)
    return lm.eval(), nar.eval()


def build_inputs(tt, st, ref_codes, deep_clone):
    """inference.py:223-255 (tokenisation + prompt assembly), reference objects only."""
    text_tokens = tt.encode("<|startoftext|>" + TEXT.strip() + "<|endoftext|>", allowed_special="all")
    text_full = tt.encode("<|startoftext|>" + TRANSCRIPT + " " + TEXT.strip() + "<|endoftext|>", allowed_special="all")
    q0_str = " ".join(str(t) for t in ref_codes[0, 0].tolist())
    speech_tokens = st.encode(q0_str.strip())
    offs = [p + len(tt.vocab) for p in speech_tokens]
    n_speech_inp = 0
    if not deep_clone:
        offs = offs[:0]
    else:
        text_tokens = text_full
        n_speech_inp = len(offs)
    prompt = torch.tensor(text_tokens + offs, dtype=torch.long)
    return prompt, prompt.shape[-1] - n_speech_inp + 1, text_tokens, speech_tokens


def gen_ar(tag, b, lm, tt, st, n_ref, n_gen, deep_clone, sample_kwargs, seed, save_logits):
    ref_codes = synth.make_ref_codes(n_ref, seed=7, merge_friendly=True)
    prompt, first_idx, text_tokens, speech_tokens = build_inputs(tt, st, ref_codes, deep_clone)
    spk_ref = ref_codes[0].T.contiguous()
    rec = []
    hook = lm.register_forward_hook(lambda m, i, o: rec.append(o[0, -1].float().clone()))
    torch.manual_seed(seed)
    out = ar_generate(tt, st, lm, prompt, spk_ref, first_idx, max_len=prompt.shape[0] + n_gen, fp16=False,
                      vocode=False, use_kv_cache=True, n_phones_gen=round(1.0 * len(TEXT)), **sample_kwargs)
    hook.remove()
    # oracle replay
    p = O.ARSamplingParams(temperature=sample_kwargs["temperature"], top_k=sample_kwargs["topk"], top_p=sample_kwargs["top_p"],
                           typical_p=sample_kwargs["typical_p"], alpha_frequency=sample_kwargs["alpha_frequency"],
                           alpha_presence=sample_kwargs["alpha_presence"], penalty_window=sample_kwargs["penalty_window"],
                           eos_penalty_decay=sample_kwargs["eos_penalty_decay"], eos_penalty_factor=sample_kwargs["eos_penalty_factor"],
                           n_phones_gen=round(1.0 * len(TEXT)))
    g = torch.Generator().manual_seed(seed)
    o_tokens, o_logits = O.ar_generate_oracle(b.ar_ckpt["model"], b.ar_shape.nhead, b.n_text, b.n_speech,
                                              st.special_tokens["<|endofspeech|>"], prompt, spk_ref,
                                              prompt.shape[0] + n_gen, p, generator=g, return_logits=True,
                                              sliding_window=b.ar_shape.sliding_window)
    assert torch.equal(out, o_tokens), f"{tag}: oracle tokens differ from reference\n{out}\n{o_tokens}"
    err = max(float((a - c).abs().max()) for a, c in zip(rec, o_logits))
    print(f"[{tag}] P={prompt.shape[0]} gen={out.shape[0] - prompt.shape[0]} tokens equal; max |logit diff| = {err:.3e}")
    assert err < 5e-4
    fx = dict(prompt=prompt.numpy(), first_codec_idx=np.int64(first_idx), ref_codes=ref_codes.numpy(),
              tokens=out.numpy(), text_tokens=np.array(text_tokens), speech_tokens=np.array(speech_tokens),
              seed=np.int64(seed))
    if save_logits:
        fx["logits"] = torch.stack(rec).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **fx)
    return out, first_idx, ref_codes, text_tokens


def gen_sampler(b, tt, st):
    """Reference function chain of ar_generate.py:74-115 on random logits."""
    V, n_text = b.ar_shape.n_vocab, b.n_text
    eos_idx = n_text + st.special_tokens["<|endofspeech|>"]
    g = torch.Generator().manual_seed(99)
    cases = []
    cfgs = [dict(temperature=0.7, topk=100, top_p=0.2, typical_p=1.0, af=3.0, ap=0.4, win=80, dec=0.5, fac=1.0),
            dict(temperature=1.0, topk=0, top_p=0.9, typical_p=1.0, af=0.0, ap=0.0, win=100, dec=0.0, fac=1.0),
            dict(temperature=0.5, topk=20, top_p=1.0, typical_p=1.0, af=1.5, ap=0.1, win=10, dec=1.0, fac=2.0),
            dict(temperature=0.9, topk=50, top_p=0.95, typical_p=0.6, af=3.0, ap=0.4, win=80, dec=0.5, fac=1.0),
            dict(temperature=0.7, topk=1, top_p=0.2, typical_p=1.0, af=3.0, ap=0.4, win=80, dec=0.5, fac=1.0)]
    for ci, c in enumerate(cfgs):
        for rep in range(3):
            logits = torch.randn(1, V, generator=g) * 3.0
            n_prev = [0, 1, 7, 150][(ci + rep) % 4]
            prev = torch.randint(n_text - 1, V, (n_prev,), generator=g).tolist()
            n_est = 20
            z = logits.clone()
            if len(prev) > 1:
                z = freq_rep_penalty(z, previous=torch.tensor([prev], dtype=torch.long), alpha_frequency=c["af"],
                                     alpha_presence=c["ap"], penalty_window=c["win"])
            z[..., :n_text - 1] = float("-inf")
            z[..., V + 1:] = float("-inf")
            z = early_eos_penalty(z, len(prev), n_est, c["dec"], c["fac"], eos_index=eos_idx)
            z = z / c["temperature"]
            z = top_k_top_p_filtering(z, top_k=c["topk"], top_p=c["top_p"])
            z = apply_typical_p(z, mass=c["typical_p"])
            z[..., :n_text - 1] = float("-inf")
            probs = z.log_softmax(dim=-1).flatten().exp()
            q = torch.empty(V).exponential_(1, generator=g)
            tok = int(torch.argmax(probs / q))
            p = O.ARSamplingParams(c["temperature"], c["topk"], c["top_p"], c["typical_p"], c["af"], c["ap"], c["win"], c["dec"], c["fac"], n_est)
            zo = O.filter_logits(logits[0], prev, p, n_text, eos_idx)
            assert torch.equal(torch.isinf(zo), torch.isinf(z[0])), f"sampler case {ci}/{rep}: kept set differs"
            assert torch.allclose(zo[~torch.isinf(zo)], z[0][~torch.isinf(zo)], rtol=0, atol=1e-6)
            assert O.draw_token(zo, q) == tok
            cases.append(dict(logits=logits[0].numpy(), prev=np.array(prev, dtype=np.int64), cfg=c, n_est=n_est,
                              kept=(~torch.isinf(z[0])).numpy(), probs=probs.numpy(), q=q.numpy(), tok=tok))
    print(f"[sampler] {len(cases)} cases: oracle == reference")
    np.savez_compressed(os.path.join(GOLD, "sampler_cases.npz"),
                        logits=np.stack([c["logits"] for c in cases]), kept=np.stack([c["kept"] for c in cases]),
                        probs=np.stack([c["probs"] for c in cases]), q=np.stack([c["q"] for c in cases]),
                        tok=np.array([c["tok"] for c in cases]), n_est=np.array([c["n_est"] for c in cases]),
                        prev=np.array([c["prev"] for c in cases], dtype=object),
                        cfg=np.array([json.dumps(c["cfg"]) for c in cases]),
                        n_text=np.int64(n_text), eos_idx=np.int64(eos_idx))


def gen_nar(tag, b, nar, tt, ar_tokens, first_idx, ref_codes, text_tokens, st, deep_clone, T_run, seed, save_logits):
    n_text = len(tt.vocab)
    out_tokens = (ar_tokens - n_text).clamp(min=0).squeeze()[first_idx:].tolist()
    dec = st.decode_int(out_tokens)
    gen_codes = torch.tensor([s for s in dec if type(s) == int], dtype=torch.long)
    c_text = torch.tensor(text_tokens, dtype=torch.long)[None]
    c_codes = ref_codes.permute(0, 2, 1).contiguous()
    c_tl = torch.tensor([len(text_tokens)], dtype=torch.long)
    c_cl = torch.tensor([c_codes.shape[1]], dtype=torch.long)
    _x = gen_codes[None, :, None].repeat(1, 1, 8)
    pad = torch.zeros((1, _x.shape[1]), dtype=torch.bool)
    diff = MultinomialDiffusion(1025, timesteps=200, device="cpu")
    dsh = DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=deep_clone, jump_len=1, jump_n_sample=1,
              q0_override_steps=20, enable_kevin_scaled_inference=True, progress=False)
    steps = []
    orig = ref_diffuser.reverse_diffusion

    def spy(diff_, model, batch, *a, **k):
        x_in, t = batch[4].clone(), int(batch[-1][0])
        x_out, x0 = orig(diff_, model, batch, *a, **k)
        steps.append(dict(t=t, x_t=x_in[0], x_tm1_pre=x_out[0].clone()))
        return x_out, x0

    ref_diffuser.reverse_diffusion = spy
    torch.manual_seed(seed)
    # NB: get_schedule(T_run) walks T_run-1..0 of the 200-step tables: a cheap prefix-free
    # way to exercise t = 0 (no known-branch draw) and t > 0.
    final = perform_simple_inference(nar, (c_text, c_codes.clone(), c_tl, c_cl.clone(), _x, pad), diff, T_run,
                                     torch.float16, dsh=dsh, retain_quant0=True)
    ref_diffuser.reverse_diffusion = orig
    # oracle replay (its own loop over the same times)
    p = O.NARParams(T=T_run, x_0_temp=0.7, guidance_w=3.0, deep_clone=deep_clone, q0_override_steps=20)
    rec = []
    g = torch.Generator().manual_seed(seed)
    o_final = O.perform_simple_inference_oracle(b.nar_ckpt["model"], b.nar_shape.nhead, c_text[0], c_codes[0], gen_codes, p,
                                                generator=g, record=rec)
    n_diff = int((o_final != final[0]).sum())
    print(f"[{tag}] S={steps[0]['x_t'].shape[0]} steps={len(steps)} final mismatches oracle vs reference: {n_diff}/{final[0].numel()}")
    assert n_diff == 0, "oracle NAR trajectory differs from the reference"
    # one forward pair for logit pinning
    t0 = steps[0]["t"]
    x0 = steps[0]["x_t"][None]
    tt_ = torch.tensor([t0])
    with torch.inference_mode():
        lc = nar(c_text, c_codes.clone(), c_tl, c_cl.clone(), x0, torch.zeros(1, x0.shape[1], dtype=torch.bool), tt_).permute(0, 1, 3, 2)[0]
        lu = nar(c_text, c_codes.clone(), c_tl, c_cl.clone(), x0, torch.zeros(1, x0.shape[1], dtype=torch.bool), tt_, drop_cond=True).permute(0, 1, 3, 2)[0]
    oc = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text[0], c_codes[0], x0[0], t0, False)
    ou = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text[0], c_codes[0], x0[0], t0, True)
    e1, e2 = float((lc - oc).abs().max()), float((lu - ou).abs().max())
    print(f"[{tag}] nar_forward max |diff| cond {e1:.3e} uncond {e2:.3e} (|logit| max {float(lc.abs().max()):.2f})")
    assert e1 < 2e-4 and e2 < 2e-4
    fx = dict(c_text=c_text[0].numpy(), c_codes=c_codes[0].numpy(), x_l0=gen_codes.numpy(), final=final[0].numpy(),
              steps_t=np.array([s["t"] for s in steps]), steps_x_t=np.stack([s["x_t"].numpy() for s in steps]),
              steps_x_tm1=np.stack([r["x_tm1"].numpy() for r in rec]), seed=np.int64(seed), T_run=np.int64(T_run),
              deep_clone=np.bool_(deep_clone))
    if save_logits:
        fx["logits_c_sub"] = lc[:, :, ::8].numpy()
        fx["logits_u_sub"] = lu[:, :, ::8].numpy()
        fx["logits_c_argmax"] = lc.argmax(-1).numpy()
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **fx)


def gen_tokenizers():
    """Text (regex BPE) and speech (codebook BPE) tokenizers of the reference (mars5/minbpe/{regex,codebook}.py) on
    awkward strings -- contractions, digit runs, whitespace runs, non-ASCII, special tokens in the text -- for three
    synthetic vocabularies (tiny, the bench's, and one with 1023 speech merges); encode, decode round trip, decode_int."""
    out = {}
    for size, (tm, sm) in O.TOKENIZER_TEST_VOCABS.items():
        tt, st = ref_tokenizers(synth.make_vocab(tm, sm))
        for i, sx in enumerate(O.TOKENIZER_TEST_STRINGS):
            ids = tt.encode(sx, allowed_special="all")
            out[f"{size}_text_{i}"] = np.array(ids, dtype=np.int64)
            out[f"{size}_text_{i}_dec"] = np.frombuffer(tt.decode(ids).encode("utf-8"), dtype=np.uint8)
            ids_o = tt.encode_ordinary(sx)
            out[f"{size}_text_{i}_ord"] = np.array(ids_o, dtype=np.int64)
        for i, cs in enumerate(O.tokenizer_test_code_strings()):
            ids = st.encode(cs)
            out[f"{size}_code_{i}"] = np.array(ids, dtype=np.int64)
            out[f"{size}_code_{i}_int"] = np.array(st.decode_int(ids), dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "tokenizer_cases.npz"), **out)
    print("tokenizer fixture:", len(out), "arrays")


def gen_tts():
    """The reference's OWN ``inference.py`` (``Mars5TTS.tts``, inference.py:201-307) end to end on CPU: full-size
    synthetic checkpoints, deep clone, README sampling settings, the global CPU generator seeded once.  Encodec and
    Vocos are replaced by the deterministic stand-ins of ``oracle/fakes.py`` (injected as the ``encodec`` / ``vocos``
    modules the reference imports), everything between them runs unmodified."""
    import importlib.util
    import fakes
    enc_mod, voc_mod = types.ModuleType("encodec"), types.ModuleType("vocos")

    class EncodecModel:
        @staticmethod
        def encodec_model_24khz():
            return fakes.FakeCodec()

    class Vocos:
        @staticmethod
        def from_pretrained(name):
            return fakes.FakeVocos()

    enc_mod.EncodecModel, voc_mod.Vocos = EncodecModel, Vocos
    sys.modules["encodec"], sys.modules["vocos"] = enc_mod, voc_mod
    _np1_shim()
    spec = importlib.util.spec_from_file_location("ref_inference", "/root/reference/inference.py")
    ref_inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_inf)
    b = synth.make_bundle("full", seed=0)
    m = ref_inf.Mars5TTS(b.ar_ckpt, b.nar_ckpt, device="cpu")
    out = {"spk_emb_24": m.get_speaker_embedding(torch.zeros(320 * 24)).numpy()}        # inference.py:174-199
    cases = [dict(text="Hi there.", transcript="We meet.", ref_frames=24, max_len=64, seed=2024, deep=True),
             dict(text="Rats!", transcript="", ref_frames=30, max_len=20, seed=7, deep=False)]
    for i, c in enumerate(cases):
        cfg = ref_inf.InferenceConfig(deep_clone=c["deep"], temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                                      generate_max_len_override=c["max_len"])
        ref_audio = torch.zeros(320 * c["ref_frames"])
        torch.manual_seed(c["seed"])
        gen, wav = m.tts(c["text"], ref_audio, c["transcript"], cfg)
        final = m.vocos.last_tokens.T.contiguous()                          # (S_out, 8): what tts() handed to the vocoder
        out[f"gen_{i}"], out[f"wav_{i}"], out[f"final_{i}"] = gen.numpy(), wav.numpy(), final.numpy()
        print(f"[tts {i}] generated {gen.shape[0]} frames, final {tuple(final.shape)}, audio {wav.shape[-1]} samples")
    np.savez_compressed(os.path.join(GOLD, "tts_full.npz"), cases=np.array([json.dumps(c) for c in cases]), **out)


def _np1_shim():
    """The reference's trim.py targets NumPy 1.x: np.array(x, copy=False) meant "copy only if needed" (= np.asarray in
    2.x).  Give ITS module namespace that meaning so the reference source runs unmodified."""
    import mars5.trim as ref_trim_mod

    class _Np1:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def array(obj, *a, copy=True, subok=False, **k):
            return (np.asanyarray(obj, *a, **k) if subok else np.asarray(obj, *a, **k)) if copy is False else np.array(obj, *a, copy=copy, subok=subok, **k)

    ref_trim_mod.np = _Np1()


def gen_trim():
    """Silence trim after the vocoder (reference mars5/trim.py:110-178, called at inference.py:306 with trim_db = 27):
    synthetic 24 kHz waveforms (silence / tone bursts / decaying noise / all-zero / stereo) through the reference."""
    from mars5.trim import trim as ref_trim
    _np1_shim()
    cases, outs = [], {}
    for i, w in enumerate(O.trim_test_waves()):
        for top_db in (27, 60, 10):
            yt, idx = ref_trim(w.clone(), top_db=top_db)
            outs[f"idx_{i}_{top_db}"] = idx.numpy().astype(np.int64)
            outs[f"sum_{i}_{top_db}"] = np.float64(yt.double().abs().sum())
            cases.append((i, top_db))
    np.savez_compressed(os.path.join(GOLD, "trim_cases.npz"), cases=np.array(cases, dtype=np.int64), **outs)
    print("trim fixture:", len(cases), "cases")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also generate the full-size (1536/1024-dim) fixtures")
    ap.add_argument("--only", default=None, choices=[None, "trim", "tokenizers", "tts"], help="regenerate a single fixture")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if args.only == "trim":
        gen_trim()
        return
    if args.only == "tokenizers":
        gen_tokenizers()
        return
    if args.only == "tts":
        gen_tts()
        return
    torch.set_num_threads(8)
    greedy = dict(temperature=0.7, topk=1, top_p=0.2, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4,
                  penalty_window=80, eos_penalty_decay=0.5, eos_penalty_factor=1.0)
    sampled = dict(greedy, topk=100, top_p=0.9, penalty_window=100)

    b = synth.make_bundle("tiny", seed=0)
    tt, st = ref_tokenizers(b.ar_ckpt["vocab"])
    lm, nar = ref_models(b)
    # tokenizer fixture
    ref_codes = synth.make_ref_codes(40, seed=7, merge_friendly=True)
    prompt, first_idx, text_tokens, speech_tokens = build_inputs(tt, st, ref_codes, True)
    np.savez_compressed(os.path.join(GOLD, "tokenizer.npz"), prompt=prompt.numpy(), first_codec_idx=np.int64(first_idx),
                        text_tokens=np.array(text_tokens), speech_tokens=np.array(speech_tokens), ref_codes=ref_codes.numpy(),
                        decode_int=np.array(st.decode_int(speech_tokens)))
    gen_sampler(b, tt, st)
    out, fi, rc, ttk = gen_ar("ar_tiny_greedy_deep", b, lm, tt, st, 40, 24, True, greedy, 1234, True)
    gen_nar("nar_tiny_deep", b, nar, tt, out, fi, rc, ttk, st, True, 24, 4321, True)
    out, fi, rc, ttk = gen_ar("ar_tiny_sampled_deep", b, lm, tt, st, 40, 24, True, sampled, 1234, False)
    out, fi, rc, ttk = gen_ar("ar_tiny_greedy_shallow", b, lm, tt, st, 40, 24, False, greedy, 1234, False)
    # rotating KV cache (BASELINE config 5's mechanism at test scale): sliding_window 48, prompt 23 tokens,
    # 100 generated tokens -> positions wrap the 48-slot buffer twice (nn_future.py:249-259)
    bw = synth.make_bundle("tiny", seed=0, sliding_window=48)
    lmw, _ = ref_models(bw)
    gen_ar("ar_tiny_window48_shallow", bw, lmw, tt, st, 40, 100, False, dict(greedy, eos_penalty_factor=50.0, eos_penalty_decay=0.0), 1234, True)
    gen_nar("nar_tiny_shallow", b, nar, tt, out, fi, rc, ttk, st, False, 4, 4321, False)
    gen_trim()
    gen_tokenizers()

    if args.full:
        bf = synth.make_bundle("full", seed=0)
        ttf, stf = ref_tokenizers(bf.ar_ckpt["vocab"])
        lmf, narf = ref_models(bf)
        out, fi, rc, ttk = gen_ar("ar_full_greedy_deep", bf, lmf, ttf, stf, 24, 16, True, greedy, 1234, False)
        gen_nar("nar_full_deep", bf, narf, ttf, out, fi, rc, ttk, stf, True, 2, 4321, False)
        del lmf, narf
        gen_tts()

    meta = dict(torch=torch.__version__, device="cpu", reference="Camb-ai/MARS5-TTS @ 2024_08_07",
                text=TEXT, transcript=TRANSCRIPT, note="generated by oracle/gen_golden.py from the unmodified reference")
    with open(os.path.join(GOLD, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
