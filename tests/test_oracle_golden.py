"""Pin the CPU oracle against fixtures produced by the unmodified reference
(oracle/gen_golden.py).  CPU only."""
import io
import json
import os

import numpy as np
import pytest
import torch

import mars5_oracle as O
from mars5_tts_amd import minbpe, synth

TEXT = "The quick brown rat."
TRANSCRIPT = "We actually haven't managed to meet demand."


def _toks(b):
    tt = minbpe.RegexTokenizer()
    tt.load(io.BytesIO(b.ar_ckpt["vocab"]["texttok.model"].encode()))
    st = minbpe.CodebookTokenizer()
    st.load(io.BytesIO(b.ar_ckpt["vocab"]["speechtok.model"].encode()))
    return tt, st


def test_tokenizer_fixture(tiny_bundle, gold_dir):
    fx = np.load(os.path.join(gold_dir, "tokenizer.npz"))
    tt, st = _toks(tiny_bundle)
    text_full = tt.encode("<|startoftext|>" + TRANSCRIPT + " " + TEXT + "<|endoftext|>", allowed_special="all")
    assert text_full == fx["text_tokens"].tolist()
    q0 = " ".join(str(t) for t in fx["ref_codes"][0, 0].tolist())
    sp = st.encode(q0.strip())
    assert sp == fx["speech_tokens"].tolist()
    assert len(sp) < fx["ref_codes"].shape[-1], "fixture should exercise BPE merges"
    assert st.decode_int(sp) == fx["decode_int"].tolist()
    pr = O.build_prompt([], text_full, sp, len(tt.vocab), True)
    assert pr.prompt.tolist() == fx["prompt"].tolist() and pr.first_codec_idx == int(fx["first_codec_idx"])
    # table-based expansion == string-based decode_int
    exp = st.expansion_table()
    flat = [c for t in sp for c in exp[t]]
    assert flat == fx["decode_int"].tolist()


def test_sampler_cases(gold_dir):
    fx = np.load(os.path.join(gold_dir, "sampler_cases.npz"), allow_pickle=True)
    n_text, eos = int(fx["n_text"]), int(fx["eos_idx"])
    for i in range(fx["logits"].shape[0]):
        c = json.loads(str(fx["cfg"][i]))
        p = O.ARSamplingParams(c["temperature"], c["topk"], c["top_p"], c["typical_p"], c["af"], c["ap"], c["win"],
                               c["dec"], c["fac"], int(fx["n_est"][i]))
        z = O.filter_logits(torch.from_numpy(fx["logits"][i]), list(fx["prev"][i]), p, n_text, eos)
        assert np.array_equal((~torch.isinf(z)).numpy(), fx["kept"][i]), f"case {i}"
        probs = z.log_softmax(-1).exp().numpy()
        np.testing.assert_allclose(probs, fx["probs"][i], rtol=1e-5, atol=1e-8)
        assert O.draw_token(z, torch.from_numpy(fx["q"][i])) == int(fx["tok"][i])


@pytest.mark.parametrize("tag,deep,kw", [
    ("ar_tiny_greedy_deep", True, dict(top_k=1, top_p=0.2, penalty_window=80)),
    ("ar_tiny_sampled_deep", True, dict(top_k=100, top_p=0.9, penalty_window=100)),
    ("ar_tiny_greedy_shallow", False, dict(top_k=1, top_p=0.2, penalty_window=80)),
])
def test_ar_golden(tiny_bundle, gold_dir, tag, deep, kw):
    fx = np.load(os.path.join(gold_dir, f"{tag}.npz"))
    b = tiny_bundle
    tt, st = _toks(b)
    p = O.ARSamplingParams(temperature=0.7, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, eos_penalty_decay=0.5,
                           eos_penalty_factor=1.0, n_phones_gen=round(len(TEXT)), **kw)
    prompt = torch.from_numpy(fx["prompt"])
    ref = torch.from_numpy(fx["ref_codes"])[0].T.contiguous()
    g = torch.Generator().manual_seed(int(fx["seed"]))
    out, logits = O.ar_generate_oracle(b.ar_ckpt["model"], b.ar_shape.nhead, b.n_text, b.n_speech,
                                       st.special_tokens["<|endofspeech|>"], prompt, ref, prompt.shape[0] + 24, p,
                                       generator=g, return_logits=True)
    assert out.tolist() == fx["tokens"].tolist()
    if "logits" in fx:
        np.testing.assert_allclose(torch.stack(logits).numpy(), fx["logits"], rtol=0, atol=5e-5)


def test_ar_rotating_window_golden(gold_dir):
    """Reference run with sliding_window = 48 (prompt 18 tokens + 100 generated: positions wrap the
    rotating KV buffer twice, nn_future.py:249-259): the oracle reproduces tokens and logits."""
    from mars5_tts_amd import synth
    fx = np.load(os.path.join(gold_dir, "ar_tiny_window48_shallow.npz"))
    b = synth.make_bundle("tiny", seed=0, sliding_window=48)
    tt, st = _toks(b)
    p = O.ARSamplingParams(temperature=0.7, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, eos_penalty_decay=0.0,
                           eos_penalty_factor=50.0, n_phones_gen=round(len(TEXT)), top_k=1, top_p=0.2, penalty_window=80)
    prompt = torch.from_numpy(fx["prompt"])
    ref = torch.from_numpy(fx["ref_codes"])[0].T.contiguous()
    g = torch.Generator().manual_seed(int(fx["seed"]))
    out, logits = O.ar_generate_oracle(b.ar_ckpt["model"], b.ar_shape.nhead, b.n_text, b.n_speech,
                                       st.special_tokens["<|endofspeech|>"], prompt, ref, prompt.shape[0] + 100, p,
                                       generator=g, return_logits=True, sliding_window=48)
    assert out.shape[0] == prompt.shape[0] + 100 and out.shape[0] > 2 * 48
    assert out.tolist() == fx["tokens"].tolist()
    np.testing.assert_allclose(torch.stack(logits).numpy(), fx["logits"], rtol=0, atol=5e-5)
    # and the window matters: an unbounded cache gives different logits once the buffer has wrapped
    out2, logits2 = O.ar_generate_oracle(b.ar_ckpt["model"], b.ar_shape.nhead, b.n_text, b.n_speech,
                                         st.special_tokens["<|endofspeech|>"], prompt, ref, prompt.shape[0] + 60, p,
                                         generator=torch.Generator().manual_seed(int(fx["seed"])), return_logits=True)
    assert float((torch.stack(logits2)[55] - torch.from_numpy(fx["logits"][55])).abs().max()) > 1e-3


@pytest.mark.parametrize("tag", ["nar_tiny_deep", "nar_tiny_shallow"])
def test_nar_golden(tiny_bundle, gold_dir, tag):
    fx = np.load(os.path.join(gold_dir, f"{tag}.npz"))
    b = tiny_bundle
    p = O.NARParams(T=int(fx["T_run"]), deep_clone=bool(fx["deep_clone"]))
    rec = []
    g = torch.Generator().manual_seed(int(fx["seed"]))
    out = O.perform_simple_inference_oracle(b.nar_ckpt["model"], b.nar_shape.nhead, torch.from_numpy(fx["c_text"]),
                                            torch.from_numpy(fx["c_codes"]), torch.from_numpy(fx["x_l0"]), p,
                                            generator=g, record=rec)
    assert np.array_equal(out.numpy(), fx["final"])
    assert np.array_equal(np.stack([r["x_t"].numpy() for r in rec]), fx["steps_x_t"])
    if "logits_c_sub" in fx:
        t0 = int(fx["steps_t"][0])
        x0 = torch.from_numpy(fx["steps_x_t"][0])
        lc = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, torch.from_numpy(fx["c_text"]),
                           torch.from_numpy(fx["c_codes"]), x0, t0, False)
        lu = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, torch.from_numpy(fx["c_text"]),
                           torch.from_numpy(fx["c_codes"]), x0, t0, True)
        np.testing.assert_allclose(lc[:, :, ::8].numpy(), fx["logits_c_sub"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(lu[:, :, ::8].numpy(), fx["logits_u_sub"], rtol=0, atol=5e-5)


def test_diffusion_tables_sanity():
    tb = O.diffusion_tables(1025, 200)
    # the reference's own asserts (diffuser.py:87-89)
    assert float(O.log_add_exp(tb.log_alpha.double(), tb.log_1_min_alpha.double()).abs().sum()) < 1e-4
    assert float(O.log_add_exp(tb.log_cumprod_alpha.double(), tb.log_1_min_cumprod_alpha.double()).abs().sum()) < 1e-4


def test_trim_matches_reference(gold_dir):
    """Host-side silence trim (the last step of ``tts()``, reference inference.py:306 with trim_db = 27) against what
    the reference's ``mars5/trim.py:110-178`` returned for the same deterministic waveforms: same [start, end]
    interval and the same samples, for mono / stereo / all-zero / very short inputs at three thresholds."""
    import mars5_oracle as O
    from mars5_tts_amd.trim import trim
    fx = np.load(os.path.join(gold_dir, "trim_cases.npz"))
    waves = O.trim_test_waves()
    assert len(fx["cases"]) == 3 * len(waves)
    for i, top_db in fx["cases"].tolist():
        y, idx = trim(waves[i].clone(), top_db=top_db)
        ref_idx = fx[f"idx_{i}_{top_db}"].tolist()
        assert list(np.asarray(idx).tolist()) == ref_idx, f"wave {i} top_db {top_db}: {idx} vs reference {ref_idx}"
        assert torch.equal(y, waves[i][..., ref_idx[0]:ref_idx[1]])
        assert abs(float(y.double().abs().sum()) - float(fx[f"sum_{i}_{top_db}"])) <= 1e-9 * max(1.0, float(fx[f"sum_{i}_{top_db}"]))


def test_tokenizers_match_reference_on_awkward_strings(gold_dir):
    """Host BPE tokenizers (``mars5-tts_amd/minbpe.py``) vs the reference's (mars5/minbpe/{regex,codebook}.py) on
    contractions, digit runs, whitespace runs, non-ASCII text, special tokens inside the text, and code strings with
    and without applicable merges, for three synthetic vocabularies: ids, decode round trip, decode_int, and the
    expansion table the AR -> NAR hand-off uses."""
    import mars5_oracle as O
    from mars5_tts_amd import minbpe, synth
    fx = np.load(os.path.join(gold_dir, "tokenizer_cases.npz"))
    for size, (tm, sm) in O.TOKENIZER_TEST_VOCABS.items():
        vocab = synth.make_vocab(tm, sm)
        tt = minbpe.RegexTokenizer()
        tt.load(io.BytesIO(vocab["texttok.model"].encode()))
        st = minbpe.CodebookTokenizer()
        st.load(io.BytesIO(vocab["speechtok.model"].encode()))
        for i, sx in enumerate(O.TOKENIZER_TEST_STRINGS):
            ids = tt.encode(sx, allowed_special="all")
            assert ids == fx[f"{size}_text_{i}"].tolist(), f"{size} text {i}: {sx!r}"
            assert tt.encode_ordinary(sx) == fx[f"{size}_text_{i}_ord"].tolist(), f"{size} text {i} (ordinary)"
            assert tt.decode(ids).encode("utf-8") == fx[f"{size}_text_{i}_dec"].tobytes(), f"{size} text {i}: decode"
        exp = st.expansion_table()
        for i, cs in enumerate(O.tokenizer_test_code_strings()):
            ids = st.encode(cs)
            assert ids == fx[f"{size}_code_{i}"].tolist(), f"{size} codes {i}"
            assert st.decode_int(ids) == fx[f"{size}_code_{i}_int"].tolist()
            assert [c for t in ids for c in exp[t]] == fx[f"{size}_code_{i}_int"].tolist()
