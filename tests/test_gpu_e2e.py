"""End-to-end parity on a real MI355X through the reference-named seams
(``ar_generate`` / ``perform_simple_inference``): HIP path vs golden fixtures produced by the
unmodified reference, and vs the CPU oracle on the same seeded inputs."""
import io
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXT = "The quick brown rat."


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _toks(b):
    from mars5_tts_amd import minbpe
    tt = minbpe.RegexTokenizer()
    tt.load(io.BytesIO(b.ar_ckpt["vocab"]["texttok.model"].encode()))
    st = minbpe.CodebookTokenizer()
    st.load(io.BytesIO(b.ar_ckpt["vocab"]["speechtok.model"].encode()))
    return tt, st


def _lm(b, dt, dev):
    from mars5_tts_amd import model
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    lm.load_state_dict(b.ar_ckpt["model"])
    return lm.to(dev).set_engine_dtype(dt)


def _nar(b, dt, dev):
    from mars5_tts_amd import model
    n = b.nar_shape
    nar = model.ResidualTransformer(n.n_text_vocab, n_quant=n.n_quant, dim=n.dim, nhead=n.nhead, enc_layers=n.enc_layers,
                                    dec_layers=n.dec_layers, n_spk_layers=n.n_spk_layers, t_emb_dim=n.t_emb_dim, p_cond_drop=0, dropout=0)
    nar.load_state_dict(b.nar_ckpt["model"])
    return nar.to(dev).set_engine_dtype(dt)


SAMPLERS = {
    "ar_tiny_greedy_deep": dict(topk=1, top_p=0.2, penalty_window=80),
    "ar_tiny_sampled_deep": dict(topk=100, top_p=0.9, penalty_window=100),
    "ar_tiny_greedy_shallow": dict(topk=1, top_p=0.2, penalty_window=80),
    "ar_full_greedy_deep": dict(topk=1, top_p=0.2, penalty_window=80),
}


def _run_ar(lm, tt, st, fx, kw, n_gen, use_graph, noise):
    from mars5_tts_amd.ar_generate import ar_generate
    prompt = torch.from_numpy(fx["prompt"])
    ref = torch.from_numpy(fx["ref_codes"])[0].T.contiguous()
    return ar_generate(tt, st, lm, prompt, ref, int(fx["first_codec_idx"]), max_len=prompt.shape[0] + n_gen, fp16=False,
                       temperature=0.7, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, eos_penalty_decay=0.5,
                       eos_penalty_factor=1.0, n_phones_gen=round(len(TEXT)), vocode=False, noise=noise, use_graph=use_graph, **kw)


@pytest.mark.parametrize("tag", ["ar_tiny_greedy_deep", "ar_tiny_sampled_deep", "ar_tiny_greedy_shallow"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_ar_tiny_f32_matches_reference_tokens(dev, tiny_bundle, gold_dir, tag, use_graph):
    """fp32 engine, same Exp(1) noise stream as the reference run (CPU mt19937, seed in the
    fixture): token ids must equal the reference's, bit for bit."""
    fx = np.load(os.path.join(gold_dir, f"{tag}.npz"))
    tt, st = _toks(tiny_bundle)
    lm = _lm(tiny_bundle, torch.float32, dev)
    V = tiny_bundle.ar_shape.n_vocab
    g = torch.Generator().manual_seed(int(fx["seed"]))
    noise = torch.stack([torch.empty(V).exponential_(1, generator=g) for _ in range(24)])
    out = _run_ar(lm, tt, st, fx, SAMPLERS[tag], 24, use_graph, noise)
    assert out.cpu().tolist() == fx["tokens"].tolist()


@pytest.mark.parametrize("use_graph", [False, True])
def test_ar_rotating_window_matches_reference(dev, gold_dir, use_graph):
    """BASELINE config 5's mechanism (long-form decode past the sliding window) at test scale: the
    reference was run with sliding_window = 48 for 100 tokens (positions wrap the rotating buffer
    twice); the engine's slot = pos % window cache + decode attention over min(pos+1, window) keys
    must give the same greedy tokens and the same logits."""
    from mars5_tts_amd import synth
    from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession
    from mars5_tts_amd.ar_generate import ar_generate
    fx = np.load(os.path.join(gold_dir, "ar_tiny_window48_shallow.npz"))
    b = synth.make_bundle("tiny", seed=0, sliding_window=48)
    tt, st = _toks(b)
    lm = _lm(b, torch.float32, dev)
    V = b.ar_shape.n_vocab
    prompt = torch.from_numpy(fx["prompt"])
    ref = torch.from_numpy(fx["ref_codes"])[0].T.contiguous()
    out = ar_generate(tt, st, lm, prompt, ref, int(fx["first_codec_idx"]), max_len=prompt.shape[0] + 100, fp16=False,
                      temperature=0.7, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, eos_penalty_decay=0.0,
                      eos_penalty_factor=50.0, n_phones_gen=round(len(TEXT)), vocode=False, noise=torch.ones(100, V),
                      use_graph=use_graph, topk=1, top_p=0.2, penalty_window=80)
    assert out.cpu().tolist() == fx["tokens"].tolist()
    if use_graph:
        return
    # teacher-forced logits through both wraps of the 48-slot buffer
    eng = lm.engine()
    n_text = len(tt.vocab)
    eos = n_text + st.special_tokens["<|endofspeech|>"]
    sess = ARSession(eng, prompt.shape[0] + 100)
    cfg = ARSamplingConfig(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=80,
                           eos_penalty_factor=50.0, eos_penalty_decay=0.0, n_phones_gen=round(len(TEXT)))
    sess.configure_sampler(cfg, n_text, eos, torch.ones(100, V, device=dev))
    sess.prefill(prompt, ref)
    stv = sess.stream.cuda_stream
    errs = []
    for i in range(100):
        sess.enqueue_head_and_sample(stv)
        sess.stream.synchronize()
        errs.append(float((sess.logits.cpu() - torch.from_numpy(fx["logits"][i])).abs().max()))
        if i < 99:
            sess.enqueue_layers(stv)
    assert max(errs) < 2e-4, (max(errs), errs.index(max(errs)))


def test_ar_tiny_f32_logits_vs_reference(dev, tiny_bundle, gold_dir):
    """Teacher-forced: prefill logits and the first decode steps' logits vs the reference's."""
    from mars5_tts_amd import _lib as L
    from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession
    fx = np.load(os.path.join(gold_dir, "ar_tiny_greedy_deep.npz"))
    tt, st = _toks(tiny_bundle)
    eng = _lm(tiny_bundle, torch.float32, dev).engine()
    V = eng.shape.n_vocab
    prompt = torch.from_numpy(fx["prompt"])
    sess = ARSession(eng, prompt.shape[0] + 24)
    g = torch.Generator().manual_seed(int(fx["seed"]))
    noise = torch.stack([torch.empty(V).exponential_(1, generator=g) for _ in range(24)]).to(dev)
    cfg = ARSamplingConfig(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=80,
                           eos_penalty_factor=1.0, eos_penalty_decay=0.5, n_phones_gen=round(len(TEXT)))
    sess.configure_sampler(cfg, tiny_bundle.n_text, tiny_bundle.n_text + st.special_tokens["<|endofspeech|>"], noise)
    sess.prefill(prompt, torch.from_numpy(fx["ref_codes"])[0].T.contiguous())
    s = sess.stream.cuda_stream
    sess.enqueue_head_and_sample(s)
    sess.stream.synchronize()
    errs = [float((sess.logits.cpu() - torch.from_numpy(fx["logits"][0])).abs().max())]
    for i in range(1, 6):
        sess.enqueue_layers(s)
        sess.enqueue_head_and_sample(s)
        sess.stream.synchronize()
        errs.append(float((sess.logits.cpu() - torch.from_numpy(fx["logits"][i])).abs().max()))
    print("AR f32 logits max|diff| per step:", errs)
    assert max(errs) < 2e-4, errs     # |logits| ~ 5; fp32 accumulation-order noise only


@pytest.fixture
def f32_products(request):
    """Run a test with the fp32 engines' GEMM arithmetic set to the parameter ("exact" | "f16x3", ops.set_f32_products)."""
    from mars5_tts_amd import ops
    prev = ops.set_f32_products(request.param)
    yield request.param
    ops.set_f32_products(prev)


@pytest.mark.parametrize("tag,use_graph,f32_products", [("nar_tiny_deep", False, "exact"), ("nar_tiny_deep", True, "exact"),
                                                        ("nar_tiny_shallow", False, "exact"), ("nar_tiny_shallow", True, "exact"),
                                                        ("nar_tiny_deep", True, "f16x3"), ("nar_tiny_shallow", True, "f16x3")],
                         indirect=["f32_products"])
def test_nar_tiny_f32_matches_reference(dev, tiny_bundle, gold_dir, tag, use_graph, f32_products):
    """fp32 engine with the reference run's RNG stream (CPU generator, seed in the fixture).
    Whole-trajectory equality is the headline statistic; the hard assertion is teacher-forced
    (each step restarted from the reference's x_t) so one libm-ulp near-tie cannot cascade."""
    from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, perform_simple_inference
    fx = np.load(os.path.join(gold_dir, f"{tag}.npz"))
    nar = _nar(tiny_bundle, torch.float32, dev)
    deep = bool(fx["deep_clone"])
    T = int(fx["T_run"])
    c_text = torch.from_numpy(fx["c_text"])[None]
    c_codes = torch.from_numpy(fx["c_codes"])[None]
    x_l0 = torch.from_numpy(fx["x_l0"])
    _x = x_l0[None, :, None].repeat(1, 1, 8)
    batch = (c_text, c_codes, torch.tensor([c_text.shape[1]]), torch.tensor([c_codes.shape[1]]), _x, torch.zeros(1, _x.shape[1], dtype=torch.bool))
    diff = MultinomialDiffusion(1025, timesteps=200, device="cpu")
    dsh = DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=deep, q0_override_steps=20)
    g = torch.Generator().manual_seed(int(fx["seed"]))
    out = perform_simple_inference(nar, batch, diff, T, torch.float16, dsh=dsh, retain_quant0=True,
                                   uniform=lambda shp: torch.rand(shp, generator=g).to(dev),
                                   randint=lambda shp: torch.randint(0, 1025, shp, dtype=torch.long, generator=g), use_graph=use_graph)
    n_bad = int((out[0].cpu() != torch.from_numpy(fx["final"])).sum())
    print(f"{tag} graph={use_graph}: free-running final mismatches {n_bad}/{fx['final'].size}")
    # teacher-forced per step
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    eng = nar.engine()
    off = c_codes.shape[1] if deep else 0
    S = fx["steps_x_t"].shape[1]
    x_known = torch.zeros(S, 8, dtype=torch.long)
    m = torch.zeros(S, 8, dtype=torch.uint8)
    m[:, 0] = 1
    if deep:
        x_known[:off] = c_codes[0]
        m[:off] = 1
        x_known[off:, 0] = x_l0
    else:
        x_known[:, 0] = x_l0
    g = torch.Generator().manual_seed(int(fx["seed"]))
    torch.randint(0, 1025, (1, x_l0.shape[0], 8), dtype=torch.long, generator=g)
    sess = NARSession(eng, NARConfig(T=T, x_0_temp=0.7, guidance_w=3.0, deep_clone=deep, q0_override_steps=20))
    sess.prepare(c_text[0], c_codes[0], torch.from_numpy(fx["steps_x_t"][0]), x_known, m, off, list(range(T - 1, -1, -1)))
    import mars5_oracle as O
    from parity_util import ungated_mismatches
    sd = tiny_bundle.nar_ckpt["model"]
    nh = tiny_bundle.nar_shape.nhead
    tb = O.diffusion_tables(1025, 200)
    spk = [O.nar_spk_vector(sd, c_codes[0], nh, False), O.nar_spk_vector(sd, c_codes[0], nh, True)]
    n_excused = 0
    for i in range(T):
        t = int(fx["steps_t"][i])
        x_t = torch.from_numpy(fx["steps_x_t"][i])
        sess.x.copy_(x_t.to(dev))
        drawn = []

        def uni(shp):
            drawn.append(torch.rand(shp, generator=g))
            return drawn[-1].to(dev)

        sess.step(uni, use_graph=use_graph)
        sess.stream.synchronize()
        got = sess.x.cpu()
        want = torch.from_numpy(fx["steps_x_tm1"][i])
        if not torch.equal(got, want):
            # the oracle's scores at this step (its ids are pinned to this very fixture by tests/test_oracle_golden.py): a
            # differing id is legal only where the two classes' scores are within what the fp32 engine's logit
            # noise (<= 3e-4, test_nar_tiny_logits_vs_reference) times the guidance / temperature gain (7.1) can move
            lc = O.nar_forward(sd, nh, c_text[0], c_codes[0], x_t, t, False, spk[0])
            lu = O.nar_forward(sd, nh, c_text[0], c_codes[0], x_t, t, True, spk[1])
            ref, s_unk, s_kn = O.reverse_step(tb, lc, lu, x_t, x_known, m.bool(), t, drawn[0][0], drawn[1][0] if t > 0 else None, 3.0, 0.7,
                                              return_scores=True)
            if 20 < t:
                ref[:, 0] = x_known[:, 0]
            assert torch.equal(ref, want), "oracle and reference fixture disagree (CPU pinning test should have caught this)"
            n_mis, bad = ungated_mismatches(got, want, s_unk, s_kn, m.bool(), eps=5e-3)
            assert not bad, f"{tag} step {i} (t={t}): ids differ from the reference away from any tie: {bad[:5]}"
            n_excused += n_mis
    print(f"{tag}: teacher-forced steps: {n_excused} tie-excused id differences over {T * S * 8}")
    if n_excused == 0:
        assert n_bad == 0, f"{n_bad} free-running final ids differ although every teacher-forced step is exact"
    else:
        assert n_bad <= 0.02 * fx["final"].size      # a legal tie flip may cascade through the free-running trajectory


@pytest.mark.parametrize("f32_products", ["exact", "f16x3"], indirect=True)
def test_nar_tiny_logits_vs_reference(dev, tiny_bundle, gold_dir, f32_products):
    """fp32 engine logits against the unmodified reference's (fixture) -- in both fp32 product modes: the exact fp32 MFMA
    (an fmaf chain) and the split-f16 mode (three f16 MFMAs per product, operand error 2^-22: VERDICT r5 #4 asked whether
    its logits stay inside the same 3e-4)."""
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    fx = np.load(os.path.join(gold_dir, "nar_tiny_deep.npz"))
    eng = _nar(tiny_bundle, torch.float32, dev).engine()
    c_text, c_codes = torch.from_numpy(fx["c_text"]), torch.from_numpy(fx["c_codes"])
    S = fx["steps_x_t"].shape[1]
    off = c_codes.shape[0]
    T = int(fx["T_run"])
    sess = NARSession(eng, NARConfig(T=T))
    z = torch.zeros(S, 8, dtype=torch.long)
    sess.prepare(c_text, c_codes, torch.from_numpy(fx["steps_x_t"][0]), z, z.to(torch.uint8), off, list(range(T - 1, -1, -1)))
    sess.enqueue_forward(sess.stream.cuda_stream)
    sess.stream.synchronize()
    so = S - off
    lg = sess.logits.cpu()[:, :, :1025]
    ref_c = torch.from_numpy(fx["logits_c_sub"])[off:, 1:]
    ref_u = torch.from_numpy(fx["logits_u_sub"])[off:, 1:]
    ec = float((lg[:so, :, ::8] - ref_c).abs().max())
    eu = float((lg[so:, :, ::8] - ref_u).abs().max())
    print(f"NAR f32 ({f32_products}) logits max|diff| cond {ec:.3e} uncond {eu:.3e}")
    assert ec < 3e-4 and eu < 3e-4
    assert torch.equal(lg[:so].argmax(-1), torch.from_numpy(fx["logits_c_argmax"])[off:, 1:])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_nar_tiny_reduced_precision_logits(dev, dt):
    import mars5_oracle as O
    from mars5_tts_amd import synth
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    b = synth.make_bundle("tiny", seed=0, dtype_round="f16" if dt == torch.float16 else "bf16")
    eng = _nar(b, dt, dev).engine()
    g = torch.Generator().manual_seed(3)
    c_text = torch.randint(0, b.nar_shape.n_text_vocab, (19,), generator=g)
    c_codes = synth.make_ref_codes(30)[0].T.contiguous()
    S, off, t = 77, 30, 150
    x = torch.randint(0, 1025, (S, 8), generator=g)
    lc = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text, c_codes, x, t, False)
    lu = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text, c_codes, x, t, True)
    sess = NARSession(eng, NARConfig(T=200))
    z = torch.zeros(S, 8, dtype=torch.long)
    sess.prepare(c_text, c_codes, x, z, z.to(torch.uint8), off, [t])
    sess.enqueue_forward(sess.stream.cuda_stream)
    sess.stream.synchronize()
    lg = sess.logits.cpu()[:, :, :1025]
    so = S - off
    tol = 0.05 if dt == torch.float16 else 0.3
    ec = float((lg[:so] - lc[off:, 1:]).abs().max())
    eu = float((lg[so:] - lu[off:, 1:]).abs().max())
    # the same forward with the engine's operand rounding (weights here are already dt-valued): a much tighter bound
    lce = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text, c_codes, x, t, False, dt=dt)
    lue = O.nar_forward(b.nar_ckpt["model"], b.nar_shape.nhead, c_text, c_codes, x, t, True, dt=dt)
    ece = float((lg[:so] - lce[off:, 1:]).abs().max())
    eue = float((lg[so:] - lue[off:, 1:]).abs().max())
    print(f"NAR {dt} logits max|diff| vs fp32 oracle cond {ec:.3e} uncond {eu:.3e}; vs operand-rounding oracle cond {ece:.3e} uncond {eue:.3e} "
          f"(|logit| max {float(lc.abs().max()):.2f})")
    assert ec < tol and eu < tol
    assert ece < tol / 2 and eue < tol / 2


@pytest.mark.parametrize("f32_products", ["exact", "f16x3"], indirect=True)
def test_full_size_goldens_f32(dev, gold_dir, full_bundle, f32_products):
    """The real MARS5 geometry (1536-d x 26 layers AR, 1024-d 8+16 layers NAR) with seeded
    weights regenerated on this host: fp32 engine vs tokens produced by the reference -- with exact fp32-MFMA products and
    with split-f16 products (csrc/gemm.hip X3: the prefill GEMMs / attention of the AR stage and the whole NAR forward)."""
    from mars5_tts_amd import synth
    from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, perform_simple_inference
    b = full_bundle
    tt, st = _toks(b)
    fx = np.load(os.path.join(gold_dir, "ar_full_greedy_deep.npz"))
    lm = _lm(b, torch.float32, dev)
    V = b.ar_shape.n_vocab
    g = torch.Generator().manual_seed(int(fx["seed"]))
    noise = torch.stack([torch.empty(V).exponential_(1, generator=g) for _ in range(16)])
    out = _run_ar(lm, tt, st, fx, SAMPLERS["ar_full_greedy_deep"], 16, True, noise)
    assert out.cpu().tolist() == fx["tokens"].tolist()
    del lm
    torch.cuda.empty_cache()
    fx = np.load(os.path.join(gold_dir, "nar_full_deep.npz"))
    nar = _nar(b, torch.float32, dev)
    c_text, c_codes = torch.from_numpy(fx["c_text"])[None], torch.from_numpy(fx["c_codes"])[None]
    _x = torch.from_numpy(fx["x_l0"])[None, :, None].repeat(1, 1, 8)
    batch = (c_text, c_codes, torch.tensor([c_text.shape[1]]), torch.tensor([c_codes.shape[1]]), _x, torch.zeros(1, _x.shape[1], dtype=torch.bool))
    g = torch.Generator().manual_seed(int(fx["seed"]))
    traj = []
    outn = perform_simple_inference(nar, batch, MultinomialDiffusion(1025, timesteps=200), int(fx["T_run"]), torch.float16,
                                    dsh=DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=True, q0_override_steps=20),
                                    uniform=lambda shp: torch.rand(shp, generator=g).to(dev),
                                    randint=lambda shp: torch.randint(0, 1025, shp, dtype=torch.long, generator=g), on_step=traj.append)
    n_bad = int((outn[0].cpu() != torch.from_numpy(fx["final"])).sum())
    # every step of the engine's own trajectory replayed by the oracle (on the GPU: full-size forwards): ids equal up to oracle ties
    import mars5_oracle as O
    from parity_util import gate_trajectory
    off = c_codes.shape[1]
    S = off + _x.shape[1]
    x_known = torch.zeros(S, 8, dtype=torch.long)
    m = torch.zeros(S, 8, dtype=torch.uint8)
    m[:, 0] = 1
    m[:off] = 1
    x_known[:off] = c_codes[0]
    x_known[off:, 0] = torch.from_numpy(fx["x_l0"])
    n_diff, bad = gate_trajectory(O, b.nar_ckpt["model"], b.nar_shape.nhead, c_text[0], c_codes[0], x_known, m, traj, 3.0, 0.7, 20, 5e-3, dev)
    print(f"full-size NAR: {n_bad}/{fx['final'].size} final ids differ from the reference; per-step replay: {n_diff} tie-excused, {len(bad)} unexcused")
    assert not bad, bad[:5]
    assert n_bad == 0 or n_diff > 0, "final ids differ although every step equals the oracle's"
    assert n_bad <= 0.02 * fx["final"].size


# ------------------------------------------------------------------------ batched requests (config 3)
def _nar_batch_tuple(c_text, c_codes, x_l0):
    _x = x_l0[None, :, None].repeat(1, 1, 8)
    return (c_text[None], c_codes[None], torch.tensor([c_text.shape[0]]), torch.tensor([c_codes.shape[0]]), _x,
            torch.zeros(1, _x.shape[1], dtype=torch.bool))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_nar_batch_mixed_lengths_equals_single(dev, tiny_bundle, gold_dir, dt):
    """Config 3 (a batch of mixed-length requests on one GPU): the batched decoder pass must give every
    utterance exactly the codes it gets alone (bit-exact: batching only re-orders independent work), and
    the utterance that is the reference fixture must still match the reference's output."""
    from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, perform_batch_inference, perform_simple_inference
    fx = np.load(os.path.join(gold_dir, "nar_tiny_deep.npz"))
    nar = _nar(tiny_bundle, dt, dev)
    T = int(fx["T_run"])
    c_text, c_codes, x_l0 = torch.from_numpy(fx["c_text"]), torch.from_numpy(fx["c_codes"]), torch.from_numpy(fx["x_l0"])
    gg = torch.Generator().manual_seed(123)
    utts = [(c_text, c_codes, x_l0, int(fx["seed"])),                                       # the reference fixture itself
            (c_text[: max(3, c_text.shape[0] // 2)], c_codes[: c_codes.shape[0] // 3], x_l0[: x_l0.shape[0] // 2], 41),
            (torch.cat([c_text, c_text[1:-1]]), torch.randint(0, 1024, (c_codes.shape[0] + 37, 8), generator=gg),
             torch.randint(0, 1024, (x_l0.shape[0] + 29,), generator=gg), 42),
            (c_text[:5], c_codes[:70], x_l0[:9], 43)]
    diff = MultinomialDiffusion(1025, timesteps=200, device="cpu")
    dsh = DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=True, q0_override_steps=20)

    def streams(seed):
        g = torch.Generator().manual_seed(seed)
        return (lambda shp: torch.rand(shp, generator=g).to(dev)), (lambda shp: torch.randint(0, 1025, shp, dtype=torch.long, generator=g))

    singles = []
    for ct, cc, xl, seed in utts:
        uni, ri = streams(seed)
        singles.append(perform_simple_inference(nar, _nar_batch_tuple(ct, cc, xl), diff, T, torch.float16, dsh=dsh, retain_quant0=True,
                                                uniform=uni, randint=ri).cpu())
    for use_graph in (False, True):
        sr = [streams(seed) for *_, seed in utts]
        outs = perform_batch_inference(nar, [_nar_batch_tuple(ct, cc, xl) for ct, cc, xl, _ in utts], diff, T, dsh=dsh,
                                       uniforms=[s[0] for s in sr], randints=[s[1] for s in sr], use_graph=use_graph)
        for i, (o, s) in enumerate(zip(outs, singles)):
            assert o.shape == s.shape
            assert torch.equal(o.cpu(), s), f"utterance {i} (graph={use_graph}): {int((o.cpu() != s).sum())} codes differ from the lone run"
    if dt == torch.float32:
        n_bad = int((outs[0][0].cpu() != torch.from_numpy(fx["final"])).sum())
        assert n_bad <= 0.02 * fx["final"].size


def _tiny_tts(tiny_bundle, dev, dt):
    """A Mars5TTS around the tiny synthetic checkpoints (the constructor itself fixes the real geometry)."""
    from inference import Mars5TTS
    m = Mars5TTS.__new__(Mars5TTS)
    m.device = dev
    m.codec = m.vocos = False
    m.texttok, m.speechtok = _toks(tiny_bundle)
    m.n_vocab = len(m.texttok.vocab) + len(m.speechtok.vocab)
    m.n_text_vocab = len(m.texttok.vocab) + 1
    m.diffusion_n_classes = 1025
    m.codeclm = _lm(tiny_bundle, dt, dev)
    m.codecnar = _nar(tiny_bundle, dt, dev)
    m.default_T, m.sr, m.latent_sr = 12, 24000, 75
    m._expansion = m.speechtok.expansion_table()
    return m


def test_tts_batch_equals_sequential_seeded_calls(dev, tiny_bundle):
    """``tts_batch_from_codes(seeds=[s_i])`` == ``torch.manual_seed(s_i); tts_from_codes(...)`` per request:
    private per-request generators reproduce the global-generator streams (AR Exp(1) draws, generator
    rewind after the decode, NAR randint / rand), independent of batch composition and order."""
    from inference import InferenceConfig
    from mars5_tts_amd import synth
    m = _tiny_tts(tiny_bundle, dev, torch.bfloat16)
    texts = ["The quick brown rat.", "Hi.", "A somewhat longer sentence, to vary the lengths.", "Rats!"]
    trs = ["We meet.", "Demand is high, we hear.", "Ok.", "Yes yes."]
    refs = [synth.make_ref_codes(n, seed=7 + i) for i, n in enumerate([60, 25, 90, 40])]
    seeds = [1000, 1001, 1002, 1003]
    cfgs = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                           generate_max_len_override=220)
    seq = []
    for i in range(4):
        cfg_i = cfgs
        torch.manual_seed(seeds[i])
        seq.append(m.tts_from_codes(texts[i], refs[i], trs[i], cfg_i))
    for nb in (4, 3):
        out = m.tts_batch_from_codes(texts, refs, trs, cfgs, seeds=seeds, nar_batch=nb)
        for i in range(4):
            assert torch.equal(out[i][0].cpu(), seq[i][0].cpu()), f"request {i}: AR frames differ"
            assert torch.equal(out[i][1].cpu(), seq[i][1].cpu()), f"request {i}: final codes differ (nar_batch={nb})"
    print("generated frames per request:", [int(a.shape[0]) for a, _ in seq])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_in_graph_uniforms_equal_torch_rand_draws(dev, tiny_bundle, gold_dir, dt):
    """The reverse steps' uniforms generated inside the step graph (nar_engine.PhiloxDraws / m5_nar_uniforms: what every
    generator-backed call uses) against the same utterance with explicit ``torch.rand`` draws from an identically seeded
    generator (the reference's own calls, diffuser.py:219-228): identical codes after all steps, and the caller's generator is
    left exactly where the two-draws-per-step loop leaves it -- lone (hipGraph and eager) and as a batch of three."""
    from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, perform_batch_inference, perform_simple_inference
    fx = np.load(os.path.join(gold_dir, "nar_tiny_deep.npz"))
    nar = _nar(tiny_bundle, dt, dev)
    T = 12
    c_text, c_codes, x_l0 = torch.from_numpy(fx["c_text"]), torch.from_numpy(fx["c_codes"]), torch.from_numpy(fx["x_l0"])
    diff = MultinomialDiffusion(1025, timesteps=200, device="cpu")
    dsh = DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=True, q0_override_steps=20)
    utts = [(c_text, c_codes, x_l0), (c_text[:5], c_codes[:33], x_l0[:17]), (torch.cat([c_text, c_text[1:-1]]), c_codes[:50], x_l0[:41])]

    def gen(seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return g

    want, offs = [], []
    for i, (ct, cc, xl) in enumerate(utts):
        g = gen(900 + i)
        legacy = lambda shp, _g=g: torch.rand(shp, generator=_g, device=dev)      # noqa: E731  (no .gen attribute: drawn by torch, passed as buffers)
        want.append(perform_simple_inference(nar, _nar_batch_tuple(ct, cc, xl), diff, T, dsh=dsh, uniform=legacy, generator=g).cpu())
        offs.append(g.get_offset())
    for use_graph in (True, False):
        for i, (ct, cc, xl) in enumerate(utts):
            g = gen(900 + i)
            got = perform_simple_inference(nar, _nar_batch_tuple(ct, cc, xl), diff, T, dsh=dsh, generator=g, use_graph=use_graph).cpu()
            assert torch.equal(got, want[i]), f"utterance {i} (graph={use_graph}): {int((got != want[i]).sum())} codes differ from the torch.rand run"
            assert g.get_offset() == offs[i], "the generator is not where the reference's draws leave it"
    gs = [gen(900 + i) for i in range(len(utts))]
    outs = perform_batch_inference(nar, [_nar_batch_tuple(ct, cc, xl) for ct, cc, xl in utts], diff, T, dsh=dsh, generators=gs)
    for i, o in enumerate(outs):
        assert torch.equal(o.cpu(), want[i]), f"batched, utterance {i}: codes differ from the lone torch.rand run"
        assert gs[i].get_offset() == offs[i]


def test_in_graph_uniforms_survive_a_run_split_in_two(dev, tiny_bundle, gold_dir):
    """A session whose reverse steps are enqueued by two ``run`` calls (4 steps, then the rest -- what tools/nar_step_bench.py does)
    binds each part to the generator's state at that moment: the draw offsets follow the device step counter from the SESSION's
    first step, so the codes and the generator's final state equal those of a single run."""
    from mars5_tts_amd.diffuser import DSH, MultinomialDiffusion, _generator_uniform, _inpaint_state, _tables, get_schedule
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    fx = np.load(os.path.join(gold_dir, "nar_tiny_deep.npz"))
    eng = _nar(tiny_bundle, torch.bfloat16, dev).engine()
    T = 10
    c_text, c_codes, x_l0 = torch.from_numpy(fx["c_text"]), torch.from_numpy(fx["c_codes"]), torch.from_numpy(fx["x_l0"])
    dsh = DSH(x_0_temp=0.7, guidance_w=3, deep_clone=True, q0_override_steps=20)
    diff = MultinomialDiffusion(1025, timesteps=200, device="cpu")
    times = get_schedule(T, jump_n_sample=1, jump_len=1)[:-1]
    outs, offs = [], []
    for split in (None, 4):
        g = torch.Generator(device=dev)
        g.manual_seed(31)
        xr, x_known, m, offset = _inpaint_state(_nar_batch_tuple(c_text, c_codes, x_l0), 1025, dsh, dev, None, g)
        sess = NARSession(eng, NARConfig(T=T, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20), diff_tables=_tables(diff))
        sess.prepare(c_text, c_codes.to(dev), xr, x_known, m, offset, times)
        uni = _generator_uniform(dev, g)
        if split is None:
            x = sess.run(uni)
        else:
            sess.run(uni, n_steps=split)
            mid = g.get_offset()
            x = sess.run(uni, n_steps=len(times) - split)
            assert mid < g.get_offset()
        assert sess._ph is not None, "the generator-backed draw must take the in-graph path"
        outs.append(x.clone().cpu())
        offs.append(g.get_offset())
    assert torch.equal(outs[0], outs[1]) and offs[0] == offs[1]


def test_tensors_that_cross_streams_are_recorded_on_the_consuming_stream(dev, tiny_bundle, monkeypatch):
    """Round 5's GPU memory fault: `tts()` makes the text ids as a temporary of the CURRENT stream, `prepare_cond` gathers them on the
    NAR stream and the caller drops them on return -- without ``Tensor.record_stream`` torch's caching allocator recycles the block
    before the NAR stream has read it.  (i) the allocator contract ``ops.use_on`` relies on: a block recorded on a busy
    stream is not handed out again; (ii) every entry point that consumes caller tensors on a session stream calls it."""
    from inference import InferenceConfig
    from mars5_tts_amd import ops, synth
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        torch.cuda._sleep(400_000_000)                      # keeps `side` busy for ~0.2 s
    t = torch.arange(4096, device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out = t * 2                                         # reads t behind the sleep
    ops.use_on(t, side)
    ptr = t.data_ptr()
    del t
    junk = torch.full((4096,), -1, dtype=torch.long, device=dev)      # same size, same (current) stream: would reuse the block
    assert junk.data_ptr() != ptr, "the caching allocator recycled a block that a busy stream still has to read"
    side.synchronize()
    assert torch.equal(out, torch.arange(4096, device=dev) * 2)
    # (ii) call sites
    seen = []
    real = ops.use_on

    def spy(tensor, stream):
        if tensor is not None and tensor.is_cuda:
            seen.append((tuple(tensor.shape), tensor.dtype, stream))
        return real(tensor, stream)
    monkeypatch.setattr(ops, "use_on", spy)
    m = _tiny_tts(tiny_bundle, dev, torch.bfloat16)
    torch.manual_seed(5)
    m.tts_from_codes("The quick brown rat.", synth.make_ref_codes(40, seed=7).to(dev), "We meet.",
                     InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, generate_max_len_override=120))
    nar, ar = ops.session_stream(dev, "nar"), ops.session_stream(dev, "ar")
    on_nar = [x for x in seen if x[2] is nar]
    on_ar = [x for x in seen if x[2] is ar]
    assert any(len(sh) == 1 and dt == torch.long for sh, dt, _ in on_nar), "prepare_cond must record the text ids on the NAR stream"
    assert any(len(sh) == 2 and sh[1] == 8 for sh, dt, _ in on_nar), "the reference codes / inpainting state must be recorded on the NAR stream"
    assert any(len(sh) == 1 and dt == torch.long for sh, dt, _ in on_ar), "prefill must record the prompt on the AR stream"


def test_tts_stream_equals_sequential_seeded_calls(dev, tiny_bundle):
    """``tts_stream_from_codes`` (request i+1's AR decode overlapped with request i's NAR steps on two streams) must
    return, per request, exactly what ``torch.manual_seed(s_i); tts_from_codes(...)`` returns."""
    from inference import InferenceConfig
    from mars5_tts_amd import synth
    m = _tiny_tts(tiny_bundle, dev, torch.bfloat16)
    texts = ["The quick brown rat.", "Hi.", "A somewhat longer sentence, to vary the lengths.", "Rats!", "One more."]
    trs = ["We meet.", "Demand is high, we hear.", "Ok.", "Yes yes.", "Fine."]
    refs = [synth.make_ref_codes(n, seed=17 + i) for i, n in enumerate([60, 25, 90, 40, 33])]
    seeds = [2000 + i for i in range(5)]
    cfg = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100, generate_max_len_override=220)
    seq = []
    for i in range(5):
        torch.manual_seed(seeds[i])
        seq.append(m.tts_from_codes(texts[i], refs[i], trs[i], cfg))
    outs = list(m.tts_stream_from_codes(texts, refs, trs, cfg, seeds=seeds))
    assert len(outs) == 5
    for i in range(5):
        assert torch.equal(outs[i][0].cpu(), seq[i][0].cpu()), f"request {i}: AR frames differ"
        assert torch.equal(outs[i][1].cpu(), seq[i][1].cpu()), f"request {i}: final codes differ"


def test_reference_handle_gives_identical_codes(dev, tiny_bundle):
    """``prepare_reference`` caches what depends on the reference only (speech tokens, AR / NAR speaker
    vectors) and, per text, the NAR conditioning state.  Exact by construction: seeded calls with the handle -- first use,
    conditioning-cache hit on the same text, a different text, eviction beyond max_cond -- must return the codes of
    the plain call, bit for bit."""
    from inference import InferenceConfig
    from mars5_tts_amd import synth
    m = _tiny_tts(tiny_bundle, dev, torch.bfloat16)
    ref = synth.make_ref_codes(60, seed=9)
    tr = "We meet at noon."
    cfg = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100, generate_max_len_override=200)
    texts = ["The quick brown rat.", "The quick brown rat.", "Another sentence entirely.", "Third one.", "The quick brown rat."]
    plain = []
    for i, t in enumerate(texts):
        torch.manual_seed(70 + i)
        plain.append(m.tts_from_codes(t, ref, tr, cfg))
    h = m.prepare_reference(ref, tr, max_cond=2)
    for i, t in enumerate(texts):
        torch.manual_seed(70 + i)
        gen, final = m.tts_from_codes(t, None, None, cfg, ref_handle=h)
        assert torch.equal(gen.cpu(), plain[i][0].cpu()) and torch.equal(final.cpu(), plain[i][1].cpu()), f"request {i} ({t!r})"
        assert len(h.cond) <= 2
    for deep in (False,):                                    # shallow clone through the same handle (no speech prompt)
        cfg_s = InferenceConfig(deep_clone=deep, temperature=0.7, top_k=100, generate_max_len_override=60)
        torch.manual_seed(5)
        a = m.tts_from_codes("Hello there.", ref, tr, cfg_s)
        torch.manual_seed(5)
        b = m.tts_from_codes("Hello there.", None, None, cfg_s, ref_handle=h)
        assert torch.equal(a[0].cpu(), b[0].cpu()) and torch.equal(a[1].cpu(), b[1].cpu())


def test_ar_generator_left_where_reference_leaves_it(dev, tiny_bundle):
    """After ``ar_generate`` the device generator must sit where the reference leaves it: one
    Exp(1) draw of (V,) per executed loop iteration (ar_generate.py:115), including the iteration that
    samples EOS - not one per pre-drawn row."""
    lm = _lm(tiny_bundle, torch.float32, dev)
    tt, st = _toks(tiny_bundle)
    from mars5_tts_amd import synth
    from mars5_tts_amd.ar_generate import ar_generate
    V = lm.engine().shape.n_vocab
    n_text = len(tt.vocab)
    eos = n_text + st.special_tokens['<|endofspeech|>']
    ref = synth.make_ref_codes(40, seed=3)[0].T.contiguous()
    prompt = torch.tensor(tt.encode("<|startoftext|>hello there<|endoftext|>", allowed_special='all'), dtype=torch.long)
    seen_eos = seen_full = False
    for seed in range(16):
        max_len = prompt.shape[0] + (40 if seed % 2 == 0 else 600)
        torch.manual_seed(seed)
        out = ar_generate(tt, st, lm, prompt, ref, prompt.shape[0] + 1, max_len=max_len, fp16=False, temperature=1.5 if seed % 2 == 0 else 40.0, topk=0, top_p=1.0,
                          n_phones_gen=None, vocode=False)
        after = torch.rand(16, device=dev).cpu()
        n_gen = out.shape[0] - prompt.shape[0]
        ended = out.shape[0] < max_len
        torch.manual_seed(seed)
        for _ in range(n_gen + (1 if ended else 0)):
            torch.empty(V, device=dev).exponential_(1)
        expect = torch.rand(16, device=dev).cpu()
        assert torch.equal(after, expect), f"seed {seed}: generator offset wrong (n_gen {n_gen}, ended_by_eos {ended})"
        assert eos not in out.tolist()
        seen_eos |= ended
        seen_full |= not ended
        if seen_eos and seen_full:
            break
    assert seen_full and seen_eos, f"max_len run seen: {seen_full}, EOS-terminated run seen: {seen_eos}"


def test_ar_in_kernel_noise_equals_torch_exponential_rows(dev, tiny_bundle, gold_dir):
    """The sampler's own Exp(1) values (M5SampleArgs.rng: torch's Philox stream for call i, csrc/philox.h) against the same decode
    fed explicit rows drawn by ``Tensor.exponential_`` from an identically seeded generator (torch.multinomial's draw, reference
    ar_generate.py:115): identical tokens under a sampler setting where the draws decide (top_k 100, top_p 0.95, T 1.0), hipGraph
    and eager, and the generator is left one draw per executed loop iteration further on."""
    from mars5_tts_amd.ar_generate import ar_generate
    fx = np.load(os.path.join(gold_dir, "ar_tiny_sampled_deep.npz"))
    lm = _lm(tiny_bundle, torch.float32, dev)
    tt, st = _toks(tiny_bundle)
    prompt = torch.from_numpy(fx["prompt"])
    ref = torch.from_numpy(fx["ref_codes"])[0].T.contiguous()
    n_gen = 60
    V = tiny_bundle.ar_shape.n_vocab

    def run(**kw):
        return ar_generate(tt, st, lm, prompt, ref, int(fx["first_codec_idx"]), max_len=prompt.shape[0] + n_gen, fp16=False, temperature=1.0,
                           topk=100, top_p=0.95, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, penalty_window=100, eos_penalty_decay=0.5,
                           eos_penalty_factor=1.0, n_phones_gen=round(len(TEXT)), vocode=False, **kw).cpu()

    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    rows = torch.stack([torch.empty(V, device=dev).exponential_(1, generator=g) for _ in range(n_gen)])
    per_draw = g.get_offset() // n_gen
    want = run(noise=rows)
    assert len(set(want[prompt.shape[0]:].tolist())) > 5, "the sampled continuation should not be degenerate"
    for use_graph in (True, False):
        g2 = torch.Generator(device=dev)
        g2.manual_seed(4242)
        got = run(generator=g2, use_graph=use_graph)
        assert torch.equal(got, want), f"graph={use_graph}: tokens differ from the run on torch's exponential_ rows"
        n_iter = int(got.shape[0]) - int(prompt.shape[0]) + (1 if got.shape[0] < prompt.shape[0] + n_gen else 0)
        assert g2.get_offset() in (n_iter * per_draw, (n_iter + 1) * per_draw) and g2.get_offset() <= n_gen * per_draw
    # the batched decode step: one {seed, offset} pair per sequence against explicit per-sequence rows
    from mars5_tts_amd.ar_generate import ar_generate_batch
    prompts = [prompt, prompt[: prompt.shape[0] - 7], torch.cat([prompt, prompt[-5:]])]
    seeds = [11, 12, 13]
    kw = dict(max_len=[int(p.shape[0]) + 40 for p in prompts], temperature=1.0, topk=100, top_p=0.95, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4,
              penalty_window=100, eos_penalty_decay=0.5, eos_penalty_factor=1.0, n_phones_gens=[round(len(TEXT))] * 3)
    gens = []
    rows_b = []
    for sd in seeds:
        gb = torch.Generator(device=dev)
        gb.manual_seed(sd)
        rows_b.append(torch.stack([torch.empty(V, device=dev).exponential_(1, generator=gb) for _ in range(40)]))
        gb2 = torch.Generator(device=dev)
        gb2.manual_seed(sd)
        gens.append(gb2)
    want_b = ar_generate_batch(tt, st, lm, prompts, [ref] * 3, [int(fx["first_codec_idx"])] * 3, noises=rows_b, **kw)
    got_b = ar_generate_batch(tt, st, lm, prompts, [ref] * 3, [int(fx["first_codec_idx"])] * 3, generators=gens, **kw)
    for i, (w, g_) in enumerate(zip(want_b, got_b)):
        assert torch.equal(w.cpu(), g_.cpu()), f"batched decode, sequence {i}: tokens differ from the run on torch's exponential_ rows"


def test_ar_batch_tiny_f32_matches_reference_tokens(dev, tiny_bundle, gold_dir):
    """Batched AR decode (config 3) in fp32: three reference fixtures (greedy deep, sampled deep, greedy
    shallow -- different prompts, lengths and noise streams, ONE sampler configuration is shared per batch, so
    the two top-k settings run as two batches) decoded together must reproduce the reference's tokens."""
    from mars5_tts_amd.ar_generate import ar_generate_batch
    lm = _lm(tiny_bundle, torch.float32, dev)
    tt, st = _toks(tiny_bundle)
    V = lm.engine().shape.n_vocab
    for tags in (["ar_tiny_greedy_deep", "ar_tiny_greedy_shallow", "ar_tiny_greedy_deep"], ["ar_tiny_sampled_deep", "ar_tiny_sampled_deep"]):
        fxs = [np.load(os.path.join(gold_dir, f"{t}.npz")) for t in tags]
        kw = SAMPLERS[tags[0]]
        n_gen = [int(fx["tokens"].shape[0] - fx["prompt"].shape[0]) for fx in fxs]
        prompts = [torch.from_numpy(fx["prompt"]) for fx in fxs]
        max_len = max(p.shape[0] + n for p, n in zip(prompts, n_gen))
        noises = []
        for fx, p in zip(fxs, prompts):
            g = torch.Generator().manual_seed(int(fx["seed"]))
            noises.append(torch.stack([torch.empty(V).exponential_(1, generator=g) for _ in range(max_len - p.shape[0])]))
        for use_graph in (False, True):
            outs = ar_generate_batch(tt, st, lm, prompts, [torch.from_numpy(fx["ref_codes"])[0].T.contiguous() for fx in fxs],
                                     [int(fx["first_codec_idx"]) for fx in fxs], max_len=max_len, temperature=0.7, typical_p=1.0,
                                     alpha_frequency=3, alpha_presence=0.4, eos_penalty_decay=0.5, eos_penalty_factor=1.0,
                                     n_phones_gens=[round(len(TEXT))] * len(fxs), noises=noises, use_graph=use_graph, **kw)
            for i, (o, fx) in enumerate(zip(outs, fxs)):
                ref = torch.from_numpy(fx["tokens"])
                n = min(o.shape[0], ref.shape[0])
                assert o.shape[0] >= ref.shape[0], (o.shape, ref.shape)       # the batch shares one max_len: only the reference's span is compared
                assert torch.equal(o[:n].cpu(), ref[:n]), f"{tags[i]} (slot {i}, graph={use_graph}): first diff at " \
                    f"{next(j for j in range(n) if int(o[j]) != int(ref[j]))} of {n}"


@pytest.mark.parametrize("dt", [torch.bfloat16])
def test_ar_batch_full_size_vs_single(dev, dt, full_bundle):
    """Full-size geometry, 16-bit operands: the batched step (skinny GEMMs) against the batch-1 GEMV path on
    the same prompts and noise.  The two differ only by fp32 summation order inside the projections, so the
    logits of the first steps must agree to dtype tolerance and greedy tokens may first differ only late /
    at near-ties; the statistic is printed, the logit agreement is asserted."""
    from mars5_tts_amd import synth
    from mars5_tts_amd.ar_engine import ARBatchSession, ARSamplingConfig, ARSession
    b = full_bundle
    lm = _lm(b, dt, dev)
    tt, st = _toks(b)
    eng = lm.engine()
    V = eng.shape.n_vocab
    g = torch.Generator().manual_seed(5)
    Ps = [70, 131, 40, 97, 64]
    n_gen = 24
    prompts = [torch.randint(b.n_text, V - 1, (P,), generator=g) for P in Ps]
    refs = [synth.make_ref_codes(60 + 10 * i, seed=20 + i)[0].T.contiguous() for i in range(len(Ps))]
    noise = torch.ones(len(Ps), n_gen + 1, V, device=dev)
    cfg = ARSamplingConfig(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=80)
    eos = b.n_text + st.special_tokens["<|endofspeech|>"]
    from mars5_tts_amd import _lib as L
    n_cmp = 4                                   # the prefill step's logits + 3 decode steps
    single_logits, single_last, single_tok = [], [], []
    for i, P in enumerate(Ps):
        s1 = ARSession(eng, P + n_gen)
        s1.configure_sampler(cfg, b.n_text, eos, noise[i].contiguous())
        s1.prefill(prompts[i], refs[i])
        sv = s1.stream.cuda_stream
        lg, last = [], []
        for k in range(n_cmp):
            if k:
                s1.enqueue_layers(sv)
            s1.enqueue_head_and_sample(sv)
            s1.stream.synchronize()
            lg.append(s1.logits.clone())
            last.append(int(s1.state[L.ST_LAST]))
        single_logits.append(lg)
        single_last.append(last)
        s2 = ARSession(eng, P + n_gen)
        s2.configure_sampler(cfg, b.n_text, eos, noise[i].contiguous())
        s2.prefill(prompts[i], refs[i])
        single_tok.append(s2.decode().cpu())
    bs = ARBatchSession(eng, [P + n_gen for P in Ps])
    bs.configure_sampler(cfg, b.n_text, eos, noise)
    bs.prefill(prompts, refs)
    sv = bs.stream.cuda_stream
    worst, n_compared = 0.0, 0
    same = [True] * len(Ps)                     # sequence i has sampled the same tokens on both paths so far
    for k in range(n_cmp):
        if k:
            bs.enqueue_layers(sv)
        bs.enqueue_head_and_sample(sv)
        bs.stream.synchronize()
        for i in range(len(Ps)):
            if not same[i]:
                continue                        # different history: the logits are not comparable any more
            ref = single_logits[i][k]
            worst = max(worst, float((bs.logits[i] - ref).abs().max() / ref.abs().max()))
            n_compared += 1
            tb, ts = int(bs.state[i, L.ST_LAST]), single_last[i][k]
            if tb != ts:                        # greedy flip: only legitimate at a near-tie of the two candidates
                same[i] = False
                gap = float((ref[ts] - ref[tb]).abs() / ref.abs().max())
                assert gap < 2e-2, f"sequence {i} step {k}: tokens {ts} vs {tb} differ although their logits are {gap} apart"
    assert worst < 3e-2, f"batched vs batch-1 logits: rel diff {worst}"
    assert n_compared >= len(Ps) * 2, "too few comparable steps"
    bs2 = ARBatchSession(eng, [P + n_gen for P in Ps])
    bs2.configure_sampler(cfg, b.n_text, eos, noise)
    bs2.prefill(prompts, refs)
    outs = bs2.decode()
    agree = []
    for i, P in enumerate(Ps):
        o, r = outs[i].cpu(), single_tok[i]
        assert o.shape == r.shape and torch.equal(o[:P], r[:P])
        agree.append(next((j - P for j in range(P, o.shape[0]) if int(o[j]) != int(r[j])), n_gen))
    print(f"batched vs batch-1: worst rel logit diff {worst:.2e}; greedy tokens agree for the first {agree} of {n_gen} steps")


@pytest.mark.parametrize("f32_products", ["exact", "f16x3"], indirect=True)
def test_tts_entry_point_matches_reference_inference(dev, gold_dir, full_bundle, f32_products):
    """(Both fp32 product modes: exact fp32-MFMA and split-f16, ops.set_f32_products.)
    The public ``Mars5TTS.tts()`` against the reference's OWN ``inference.py`` (fixture: the unmodified reference on
    CPU, full-size seeded weights, deep and shallow clone, README sampling settings; Encodec / Vocos replaced on both
    sides by the deterministic stand-ins of oracle/fakes.py).  fp32 engine, the reference's CPU random stream replayed
    draw by draw: the AR frames must be identical and so must the final codes -- unless a step of the engine's own
    trajectory, replayed by the oracle, hit a tie of the oracle's top-2 scores (then the free-running diffusion may
    legally cascade, bounded at 2 %); when the codes are identical so is the trimmed waveform."""
    import json as _json
    import fakes
    from inference import InferenceConfig, Mars5TTS
    from mars5_tts_amd import synth
    fx = np.load(os.path.join(gold_dir, "tts_full.npz"))
    b = full_bundle
    m = Mars5TTS(b.ar_ckpt, b.nar_ckpt, device=str(dev), codec=fakes.FakeCodec(), vocos=fakes.FakeVocos())
    m.codeclm.set_engine_dtype(torch.float32)
    m.codecnar.set_engine_dtype(torch.float32)
    emb = m.get_speaker_embedding(torch.zeros(320 * 24)).cpu()                      # reference inference.py:174-199
    ref_emb = torch.from_numpy(fx["spk_emb_24"])
    assert emb.shape == ref_emb.shape == (1, 1536)
    assert float((emb - ref_emb).abs().max() / ref_emb.abs().max()) < 1e-4
    for i, cj in enumerate(fx["cases"].tolist()):
        c = _json.loads(cj)
        cfg = InferenceConfig(deep_clone=c["deep"], temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                              generate_max_len_override=c["max_len"])
        hooks = fakes.CpuStreamHooks(c["seed"], dev)
        traj = []
        hooks.nar_on_step = traj.append
        gen, wav = m.tts(c["text"], torch.zeros(320 * c["ref_frames"]), c["transcript"], cfg, rng_hooks=hooks)
        assert gen.cpu().tolist() == fx[f"gen_{i}"].tolist(), f"case {i}: AR frames differ from the reference"
        final = m.vocos.last_tokens.T.contiguous().numpy()
        ref_final = fx[f"final_{i}"]
        assert final.shape == ref_final.shape
        n_bad = int((final != ref_final).sum())
        print(f"tts case {i}: {gen.shape[0]} frames, final codes differing from the reference: {n_bad}/{ref_final.size}")
        if True:
            # replay all 200 steps of the engine's own trajectory through the oracle (on the GPU): every id must be the
            # oracle's, except at oracle ties -- and only such a tie can excuse final codes that are not bit-equal
            import mars5_oracle as O
            from parity_util import gate_trajectory
            codes = m.codec.encode(torch.zeros(1, 1, 320 * c["ref_frames"]))[0][0][0].T.contiguous().cpu()     # (Lc, 8)
            tt_ids = m.texttok.encode("<|startoftext|>" + ((c["transcript"] + ' ') if c["deep"] else '') + c["text"].strip() + "<|endoftext|>",
                                      allowed_special='all')
            off = codes.shape[0] if c["deep"] else 0
            S = traj[0]["x_t"].shape[0]
            x_known = torch.zeros(S, 8, dtype=torch.long)
            mk = torch.zeros(S, 8, dtype=torch.uint8)
            mk[:, 0] = 1
            if c["deep"]:
                x_known[:off] = codes
                mk[:off] = 1
            x_known[off:, 0] = gen.cpu()
            n_diff, bad = gate_trajectory(O, b.nar_ckpt["model"], b.nar_shape.nhead, torch.tensor(tt_ids), codes, x_known, mk, traj,
                                          3.0, 0.7, 20, 5e-3, dev)
            print(f"tts case {i}: per-step replay of the engine trajectory: {n_diff} tie-excused, {len(bad)} unexcused id differences")
            assert not bad, bad[:5]
            assert n_bad == 0 or n_diff > 0, "final codes differ although every step equals the oracle's"
        assert n_bad <= 0.02 * ref_final.size
        if n_bad == 0:
            assert wav.shape[-1] == fx[f"wav_{i}"].shape[-1]
            d = wav.cpu() - torch.from_numpy(fx[f"wav_{i}"])        # the stand-in vocoder's sin() runs on the GPU here, on the CPU there
            assert float(d.pow(2).mean().sqrt()) < 1e-4 and float(d.abs().max()) < 1e-4      # BASELINE: waveform RMS within 1e-4


# ------------------------------------------------------------------ the multi-GPU path on ONE GPU
def _bench_cmd(extra, nproc):
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py")] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


def test_c4_request_scatter_and_gather_run_over_rccl_with_one_rank(dev):
    """What a 1-GPU box can prove about BASELINE configs[3]: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1
    --workload c4 --batch 4 --backend nccl --collective-at-1` -- RCCL communicator init, the request scatter and result gather
    on DEVICE tensors (sharding.scatter_requests / gather_results), the timing all-reduces, the rank census, and rank 0's
    re-computation of two gathered results, all through bench.py's own c4 path with the real engines."""
    r = _bench_cmd(["--gpus", "1", "--workload", "c4", "--batch", "4", "--backend", "nccl", "--collective-at-1", "--steps", "1", "--warmup", "0",
                    "--n-gen", "90", "--no-preflight"], 1)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = j["collective"]
    print("c4 over RCCL, one rank:", {k: c[k] for k in ("backend", "ranks_seen", "scatter_bytes", "gather_bytes", "remote_results_rechecked_equal")}, j["value"])
    assert c["backend"] == "nccl" and c["ranks_seen"] == 1 and c["requests_per_rank"] == [4]
    assert c["scatter_bytes"] > 4 * 8 * 150 and c["gather_bytes"] > 4 * 8 * 90 and c["remote_results_rechecked_equal"] == 2
    assert j["n_gpus"] == 1 and j["value"] > 0


def test_two_rccl_ranks_on_the_one_visible_gpu_if_the_library_allows(dev):
    """`bench.py --gpus 2 --launch-check --same-device --backend nccl`: two ranks, both on device 0 -- rendezvous, communicator
    init, the all-gather census and the /dev/shm checkpoint sharing over RCCL.  RCCL may refuse two ranks on one device
    ("Duplicate GPU detected"): then this is a skip, not a failure (the CPU suite covers the same flow over gloo)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the driver's own multi-GPU run covers this")
    r = _bench_cmd(["--gpus", "2", "--launch-check", "--check-bundle", "--same-device", "--backend", "nccl"], 2)
    if r.returncode != 0:
        tail = (r.stderr + r.stdout)[-4000:]
        if "uplicate GPU" in tail or "invalid usage" in tail.lower() or "ncclInvalidUsage" in tail or "NCCL error" in tail or "ncclUnhandledCudaError" in tail:
            pytest.skip("RCCL refuses two ranks on one device: " + tail.strip().splitlines()[-1][:200])
        assert False, tail
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["collective"]["ranks_seen"] == 2 and j["shared_bundle"]["checksums_equal"]
