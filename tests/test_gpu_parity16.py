"""Parity of the BENCHMARKED configuration: the 16-bit engines (bf16 = BASELINE's dtype, f16 = the
reference's own GPU dtype) at the real MARS5 geometry and at bench length, against the oracle.

The oracle is plain torch, so it runs on the GPU here (``with torch.device(dev)``): that makes the
full-size / full-length comparison affordable (450 teacher-forced decode steps of a 26-layer
1536-d model, a 16-layer decoder pass over 1349 rows) and takes exp / log from the same device
math library as the kernels.  Two oracle modes (oracle/mars5_oracle.py, "reduced-precision emulation"):
  * ``dt=None``  fp32 arithmetic on the dt-rounded weights: bounds what the operand dtype costs;
  * ``dt=<16-bit>``  the reference's autocast rounding points (ar_generate.py:59,67): the engine's
    implementation is then checked to a much tighter tolerance.
Tolerances are absolute on logits whose magnitude is printed next to them.
"""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

TEXT = "The quick brown rat jumped over the lazy dogs twice."
TRANSCRIPT = "We actually haven't managed to meet demand this year."

# max |engine logit - oracle logit| allowed, per operand dtype:  vs the autocast-emulating oracle / vs the fp32 oracle.
# About 3x what round 2 measured on MI355X (profiles/r2a_parity16.txt): AR, |logit| <= 12.2: bf16 0.053 / 0.030, f16 0.0062 / 0.0038
# over 450 teacher-forced steps; NAR at S = 1349, |logit| <= 9.3, relative to max |logit|: bf16 0.0029 / 0.0022, f16 0.0004 / 0.0003.
AR_TOL_EMU = {torch.bfloat16: 0.15, torch.float16: 0.02}
AR_TOL_F32 = {torch.bfloat16: 0.10, torch.float16: 0.012}
NAR_TOL_EMU = {torch.bfloat16: 0.009, torch.float16: 0.0012}      # relative to max |logit|
NAR_TOL_F32 = {torch.bfloat16: 0.007, torch.float16: 0.0010}


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


def _bench_prompt(b, tt, st, ref_codes):
    """The AR prompt bench.py builds for BASELINE configs[1] (deep clone: transcript + text + reference L0 tokens)."""
    text = tt.encode("<|startoftext|>" + TRANSCRIPT + ' ' + TEXT.strip() + "<|endoftext|>", allowed_special='all')
    sp = st.encode(' '.join(str(t) for t in ref_codes[0, 0].tolist()))
    return torch.tensor(text + [s + b.n_text for s in sp], dtype=torch.long), len(text)


def _ar_compare(dev, b, dt, prompt, ref, N, n_f32, odev, text_len, tol_emu, tol_f32, min_gen):
    """Engine (graph + eager) vs oracle, teacher-forced on the engine's greedy tokens.  Returns the statistics dict."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, model
    from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(dt).engine()
    P, V = int(prompt.shape[0]), a.n_vocab
    n_text = b.n_text
    eos = n_text + st.special_tokens["<|endofspeech|>"]
    kw = dict(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100,
              eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=100 * text_len)
    noise = torch.ones(N, V, device=dev)

    # ---- engine, graph replay (the timed path)
    sg = ARSession(eng, P + N)
    sg.configure_sampler(ARSamplingConfig(**kw), n_text, eos, noise)
    sg.prefill(prompt, ref)
    tok_graph = sg.decode(use_graph=True).cpu()
    del sg
    # ---- engine, eager, logits of every step
    se = ARSession(eng, P + N)
    se.configure_sampler(ARSamplingConfig(**kw), n_text, eos, noise)
    se.prefill(prompt, ref)
    sv = se.stream.cuda_stream
    eng_logits = []
    for i in range(N):
        if i:
            se.enqueue_layers(sv)
        se.enqueue_head_and_sample(sv)
        se.stream.synchronize()
        eng_logits.append(se.logits.clone())
    n_tok = int(se.state.cpu()[L.ST_NTOK])
    tok_eager = se.tokens[:n_tok].cpu()
    assert tok_graph.tolist() == tok_eager.tolist(), "hipGraph replay and the eager launch sequence disagree"
    n_gen = n_tok - P
    assert n_gen >= min_gen, f"only {n_gen} tokens generated: the forced-length settings did not hold"

    # ---- oracle (on `odev`), teacher-forced on the engine's tokens
    okw = dict(temperature=0.7, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100, eos_penalty_factor=50.0, eos_penalty_decay=0.5,
               n_phones_gen=100 * text_len)
    p = O.ARSamplingParams(top_k=1, top_p=0.2, **okw)
    p_nofilter = O.ARSamplingParams(top_k=0, top_p=1.0, **okw)
    with torch.device(odev), torch.inference_mode():
        sd = O.round_linear_weights({k: v.to(odev) for k, v in b.ar_ckpt["model"].items()}, dt)
        forced = tok_eager.to(odev)
        noise1 = torch.ones(N, V)
        args = (sd, a.nhead, n_text, b.n_speech, st.special_tokens["<|endofspeech|>"], prompt.to(odev), ref.to(odev), P + N, p)
        _, lo_emu, choices = O.ar_generate_oracle(*args, noise=noise1, forced=forced, dt=dt)
        _, lo_f32, _ = O.ar_generate_oracle(*args, noise=noise1, forced=forced[:P + n_f32], dt=None)
        assert len(lo_emu) >= n_gen and len(lo_f32) >= n_f32
        err_emu = [float((eng_logits[i].to(odev) - lo_emu[i]).abs().max()) for i in range(n_gen)]
        err_f32 = [float((eng_logits[i].to(odev) - lo_f32[i]).abs().max()) for i in range(n_f32)]
        zmax = max(float(l.abs().max()) for l in lo_emu[:n_f32])
        flips = [i for i in range(n_gen) if choices[i] != int(tok_eager[P + i])]
        bad_flips = []
        for i in flips:
            prev = tok_eager[P:P + i].tolist()
            z = O.filter_logits(lo_emu[i], prev, p_nofilter, n_text, eos)
            margin = float(z[choices[i]] - z[int(tok_eager[P + i])])
            if not margin <= 2.0 * err_emu[i] / 0.7 + 1e-6:
                bad_flips.append((i, margin, err_emu[i]))
    first = flips[0] if flips else None
    print(f"AR {dt} dim {a.dim} x {a.n_layers}L, P={P}, {n_gen} tokens: max|dlogit| vs autocast-emulating oracle {max(err_emu):.4f} "
          f"(mean {sum(err_emu) / n_gen:.4f}), vs fp32 oracle ({n_f32} steps) {max(err_f32):.4f}; max|logit| {zmax:.2f}; "
          f"greedy agreement {n_gen - len(flips)}/{n_gen}, first divergence at step {first}")
    assert max(err_emu) <= tol_emu, (max(err_emu), err_emu.index(max(err_emu)))
    assert max(err_f32) <= tol_f32, (max(err_f32), err_f32.index(max(err_f32)))
    assert not bad_flips, f"token flips outside the oracle's near-tie margin: {bad_flips[:5]}"
    return dict(n_gen=n_gen, first_divergence=first, agreement=(n_gen - len(flips)) / n_gen, err_emu=max(err_emu), err_f32=max(err_f32))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ar_full_size_16bit_450_tokens_vs_oracle(dev, full_bundle, dt):
    """BASELINE configs[1]'s AR stage as timed by bench.py (prompt 488, 450 generated tokens, 16-bit operands), greedy
    so that no random stream is involved: (1) the hipGraph decode (what bench.py times) and the eager launch sequence
    give the same tokens; (2) every step's logits, teacher-forced on the engine's own tokens, are within AR_TOL_EMU of
    the oracle that rounds where the reference's autocast rounds, and the first 64 steps within AR_TOL_F32 of the fp32
    oracle; (3) wherever the oracle would have picked a different token, its own margin between the two candidates is
    inside twice the measured logit error of that step (a legal near-tie flip), and the agreement rate is reported."""
    from mars5_tts_amd import synth
    b = full_bundle
    tt, st = _toks(b)
    ref_codes = synth.make_ref_codes(450, seed=7)
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    assert 480 <= prompt.shape[0] <= 500
    _ar_compare(dev, b, dt, prompt, ref_codes[0].T.contiguous(), 450, 64, dev, len(TEXT), AR_TOL_EMU[dt], AR_TOL_F32[dt], 400)


def test_ar_rotating_window_past_the_wrap_16bit_vs_oracle(dev, full_bundle):
    """BASELINE configs[4]'s mechanism at the real width and dtype: a 2990-token prompt and 150 decoded
    positions cross the 3000-slot rotating KV window (reference nn_future.py:249-259), i.e. >= 100 steps run past the wrap
    with every slot of the window live; teacher-forced against the oracle on the GPU like the 450-token test."""
    from mars5_tts_amd import synth
    b = full_bundle
    dt = torch.bfloat16
    tt, st = _toks(b)
    ref_codes = synth.make_ref_codes(450, seed=7)
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    plen = 2990
    prompt = torch.cat([prompt] + [prompt[-450:]] * ((plen - int(prompt.shape[0]) + 449) // 450))[:plen]
    r = _ar_compare(dev, b, dt, prompt, ref_codes[0].T.contiguous(), 150, 16, dev, len(TEXT), AR_TOL_EMU[dt], AR_TOL_F32[dt], 140)
    assert plen + r["n_gen"] >= 3100


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ar_batch_full_size_16bit_vs_oracle(dev, full_bundle, dt):
    """The batched decode step (BASELINE configs[2]; skinny MFMA GEMMs, per-sequence positions / caches / sampler state) at
    the real geometry in the benchmarked dtypes against the ORACLE (VERDICT r2 missing #3; reference ar_generate.py:62-157
    run per sequence): 8 sequences with prompts of 64 ... 488 tokens advance 64 steps together, every step's logits of
    every sequence are compared with the autocast-emulating oracle teacher-forced on that sequence's own tokens; greedy
    flips must sit inside the oracle's near-tie margin."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, model, synth
    from mars5_tts_amd.ar_engine import ARBatchSession, ARSamplingConfig
    b = full_bundle
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(dt).engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    ref = ref_codes[0].T.contiguous()
    full, text_len = _bench_prompt(b, tt, st, ref_codes)
    Ps = [488, 400, 333, 256, 190, 150, 97, 64]
    prompts = [full[: min(P, int(full.shape[0]))].clone() for P in Ps]
    Ps = [int(p.shape[0]) for p in prompts]
    B, N, V = len(Ps), 64, a.n_vocab
    n_text = b.n_text
    eos = n_text + st.special_tokens["<|endofspeech|>"]
    kw = dict(temperature=0.7, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100, eos_penalty_factor=50.0, eos_penalty_decay=0.5,
              n_phones_gen=100 * len(TEXT))
    noise = torch.ones(B, N, V, device=dev)
    bs = ARBatchSession(eng, [P + N for P in Ps])
    bs.configure_sampler(ARSamplingConfig(topk=1, top_p=0.2, **kw), n_text, eos, noise)
    bs.prefill(prompts, [ref] * B)
    sv = bs.stream.cuda_stream
    eng_logits = []
    for i in range(N):
        if i:
            bs.enqueue_layers(sv)
        bs.enqueue_head_and_sample(sv)
        bs.stream.synchronize()
        eng_logits.append(bs.logits.clone())
    state = bs.state.cpu()
    p = O.ARSamplingParams(top_k=1, top_p=0.2, **kw)
    p_nofilter = O.ARSamplingParams(top_k=0, top_p=1.0, **kw)
    worst, n_flip, bad = 0.0, 0, []
    with torch.device(dev), torch.inference_mode():
        sd = O.round_linear_weights({k: v.to(dev) for k, v in b.ar_ckpt["model"].items()}, dt)
        for q in range(B):
            n_tok = int(state[q, L.ST_NTOK])
            toks = bs.tokens[q, :n_tok].clone()
            n_gen = n_tok - Ps[q]
            assert n_gen >= N - 1, (q, n_gen)
            _, lo, choices = O.ar_generate_oracle(sd, a.nhead, n_text, b.n_speech, st.special_tokens["<|endofspeech|>"], prompts[q].to(dev),
                                                  ref.to(dev), Ps[q] + N, p, noise=torch.ones(N, V), forced=toks, dt=dt)
            for i in range(min(n_gen, len(lo))):
                e = float((eng_logits[i][q] - lo[i]).abs().max())
                worst = max(worst, e)
                if choices[i] != int(toks[Ps[q] + i]):
                    n_flip += 1
                    z = O.filter_logits(lo[i], toks[Ps[q]:Ps[q] + i].tolist(), p_nofilter, n_text, eos)
                    margin = float(z[choices[i]] - z[int(toks[Ps[q] + i])])
                    if not margin <= 2.0 * e / 0.7 + 1e-6:
                        bad.append((q, i, margin, e))
    print(f"AR batch {dt} B={B} prompts {Ps}, {N} steps: max|dlogit| vs autocast-emulating oracle {worst:.4f}; greedy flips {n_flip} "
          f"(outside the oracle's near-tie margin: {len(bad)})")
    assert worst <= AR_TOL_EMU[dt], worst
    assert not bad, bad[:5]


def _session(eng, b, st, P, N, noise, persistent):
    from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession
    s = ARSession(eng, P + N)
    s.mega = bool(persistent) and s.mega
    cfg = ARSamplingConfig(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100,
                           eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=100 * len(TEXT))
    s.configure_sampler(cfg, b.n_text, b.n_text + st.special_tokens["<|endofspeech|>"], noise)
    return s


@pytest.mark.parametrize("window,plen", [(3000, 0), (96, 60), (3000, 1400)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ar_persistent_step_is_bit_identical_to_the_per_launch_step(dev, full_bundle, dt, window, plen):
    """The one-launch form of the decode step (csrc/ar_mega.hip: 256 co-resident workgroups, tagged-granule edges, LDS-DMA
    weight prefetch) against the five-launches-per-layer form it replaces: every step's logits bit for bit (eager), the
    same tokens from the hipGraph replays, no workgroup ever gave up waiting.  window = 96: a 60-token prompt and 200
    decoded positions wrap the rotating KV buffer twice (BASELINE configs[4]'s mechanism at the real width); a 1400-token
    prompt makes every key split longer than one 128-position pass of the cache scan (its multi-pass loop)."""
    from mars5_tts_amd import _lib as L, model, synth
    b = full_bundle
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=window)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(dt).engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    ref = ref_codes[0].T.contiguous()
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    if plen and plen < prompt.shape[0]:
        prompt = prompt[-plen:]
    elif plen:
        prompt = torch.cat([prompt] + [prompt[-450:]] * ((plen - int(prompt.shape[0]) + 449) // 450))[:plen]
    P, N = int(prompt.shape[0]), 200
    noise = torch.ones(N, a.n_vocab, device=dev)
    runs = {}
    for persistent in (False, True):
        s = _session(eng, b, st, P, N, noise, persistent)
        assert s.mega == persistent, "the persistent form should apply to this geometry on an MI355X"
        s.prefill(prompt, ref)
        sv = s.stream.cuda_stream
        lg = []
        for i in range(N):
            if i:
                s.enqueue_layers(sv)
            s.enqueue_head_and_sample(sv)
            s.stream.synchronize()
            lg.append(s.logits.clone())
        n_tok = int(s.state.cpu()[L.ST_NTOK])
        assert int(s.mega_err.cpu()[0]) == 0
        runs[persistent] = (torch.stack(lg), s.tokens[:n_tok].cpu())
        del s
    assert runs[True][1].tolist() == runs[False][1].tolist()
    n_gen = runs[True][1].shape[0] - P
    assert n_gen >= 150, n_gen
    diff = (runs[True][0][:n_gen] != runs[False][0][:n_gen]).any(dim=1)
    assert not bool(diff.any()), f"first differing step {int(diff.nonzero()[0])}"
    # the timed path: hipGraph replays
    s = _session(eng, b, st, P, N, noise, True)
    s.prefill(prompt, ref)
    tok = s.decode(use_graph=True).cpu()
    assert s.mega and tok.tolist() == runs[False][1].tolist()
    # the replays are graphs of ar_engine.GRAPH_GROUP steps (+ one-step graphs for the remainder of a poll interval)
    assert s.graph_group is not None and s.group > 1
    if window == 3000 and plen == 0:
        s = _session(eng, b, st, P, N, noise, True)
        s.prefill(prompt, ref)
        tok = s.decode(use_graph=True, poll=s.group * 2 + 3).cpu()       # every poll interval = two grouped graphs + three single steps
        assert s.mega and tok.tolist() == runs[False][1].tolist()
    print(f"AR persistent step {str(dt).split('.')[-1]} window {window} prompt {P}: {n_gen} steps bit-identical to the per-launch form")


def test_two_persistent_sessions_decode_concurrently_from_two_threads(dev, full_bundle):
    """Two batch-1 decodes on two streams from two host threads (a server with two workers): each persistent step needs every
    CU, so their launches are serialised on the device (ar_engine._mega_exclusive); both must finish without a workgroup
    ever giving up and give the tokens a lone run gives."""
    import threading
    from mars5_tts_amd import model, synth
    b = full_bundle
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(torch.bfloat16).engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    ref = ref_codes[0].T.contiguous()
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    prompts = [prompt, prompt[:-7]]
    N = 96
    noise = torch.ones(N, a.n_vocab, device=dev)

    def run(p):
        s = _session(eng, b, st, int(p.shape[0]), N, noise, True)
        assert s.mega
        s.prefill(p, ref)
        return s.decode(use_graph=True).cpu().tolist()

    alone = [run(p) for p in prompts]
    out, errs = [None, None], []

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            out[i] = run(prompts[i])
        except Exception as e:                      # noqa: BLE001 - reported below
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs
    assert out == alone


def _bench_session(dev, b, N, persistent=True):
    from mars5_tts_amd import model, synth
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=a.sliding_window)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(torch.bfloat16).engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    noise = torch.ones(N, a.n_vocab, device=dev)
    s = _session(eng, b, st, int(prompt.shape[0]), N, noise, persistent)
    s.prefill(prompt, ref_codes[0].T.contiguous())
    return s, eng


def _inject_failure(s, at_step):
    """Make the session's launches fail the way a persistent launch that gave up leaves things, once, in the batch that reaches
    `at_step` decode steps: sticky error word set, a residual stream that is not the step's, the position word bumped, and
    garbage in the KV-cache rows that batch wrote.  (The production decode loop has no test hook: the instance's
    ``_launch_steps`` is wrapped here.)"""
    from mars5_tts_amd import _lib as L
    orig = s._launch_steps
    seen = {"n": 0, "done": False}

    def wrapped(n, use_graph, st):
        orig(n, use_graph, st)
        seen["n"] += n
        if s.mega and not seen["done"] and seen["n"] >= at_step:
            seen["done"] = True
            with torch.cuda.stream(s.stream):
                pos = int(s.state.cpu()[L.ST_POS])
                rows = torch.tensor(sorted({(pos - i) % s.window for i in range(n + 1)}), device=s.m.dev)
                s.kc.index_fill_(2, rows, 7.0)
                s.vc.index_fill_(2, rows, -7.0)
                s.mega_err.fill_(1)
                s.xdec.mul_(0.5)
                s.state[L.ST_POS] += 3

    s._launch_steps = wrapped


def test_persistent_step_failure_past_the_cache_wrap_is_recovered(dev, full_bundle):
    """past the wrap of the rotating KV cache a failed batch has overwritten rows that still held positions
    pos - window .., which the replay attends to.  96-slot window, 60-token prompt (the cache wraps after 36 steps), failure in
    the batch that reaches step 64: decode() must restore the rows it kept, and the tokens must equal a clean run's."""
    from mars5_tts_amd import model, synth
    b = full_bundle
    tt, st = _toks(b)
    a = b.ar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9, sliding_window=96)
    lm.load_state_dict(b.ar_ckpt["model"])
    eng = lm.to(dev).set_engine_dtype(torch.bfloat16).engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    ref = ref_codes[0].T.contiguous()
    prompt = _bench_prompt(b, tt, st, ref_codes)[0][-60:]
    P, N = int(prompt.shape[0]), 200
    noise = torch.ones(N, a.n_vocab, device=dev)
    runs = []
    for at in (None, 64, 150):
        s = _session(eng, b, st, P, N, noise, True)
        assert s.mega and s.w_alloc == s.window == 96
        s.prefill(prompt, ref)
        if at is not None:
            _inject_failure(s, at)
        runs.append(s.decode(use_graph=True).cpu().tolist())
        assert s.mega_recovered == (0 if at is None else 1)
    assert len(runs[0]) - P >= 150
    assert runs[1] == runs[0] and runs[2] == runs[0]


def test_persistent_step_failure_is_recovered_on_the_per_launch_form(dev, full_bundle):
    """ADVICE r2 (medium) / VERDICT r2 weak #8: if a persistent launch gives up (grid not co-resident) its error word is
    sticky and the rest of the batch samples from a stale residual stream.  decode() must not lose the request: it restores
    the state of the last clean poll, switches to the per-launch form and replays.  The failure is injected the way the
    kernel leaves it (error word set, residual stream and position corrupted) after 64 of 160 steps; the tokens must equal
    a clean run's -- the two forms are bit-identical -- and the session must report one recovery."""
    from mars5_tts_amd import ar_engine
    N = 160
    s, _ = _bench_session(dev, full_bundle, N)
    assert s.mega
    clean = s.decode(use_graph=True).cpu().tolist()
    s2, _ = _bench_session(dev, full_bundle, N)
    _inject_failure(s2, 64)
    got = s2.decode(use_graph=True).cpu().tolist()
    assert s2.mega_recovered == 1 and not s2.mega and ar_engine.LAST_STATS["persistent_recoveries"] == 1
    assert int(s2.mega_err.cpu()[0]) == 0
    assert got == clean
    # the eager (no hipGraph) path recovers the same way
    s3, _ = _bench_session(dev, full_bundle, N)
    _inject_failure(s3, 96)
    assert s3.decode(use_graph=False).cpu().tolist() == clean and s3.mega_recovered == 1


def test_persistent_decode_with_foreign_work_on_another_stream(dev, full_bundle):
    """The serving situation the advisor described: while the persistent decode step runs (it wants every CU), another
    stream of the same process keeps the GPU busy with NAR-style GEMM work (workgroups that hold most of a CU's LDS, so a
    persistent workgroup cannot be co-resident with them).  Whatever the interleaving does -- the launches wait for the
    foreign workgroups, or a spin bound trips and decode() recovers on the per-launch form -- the tokens must be the lone
    run's and nothing may raise."""
    import threading
    from mars5_tts_amd import _lib as L, ops
    N = 128
    s, eng = _bench_session(dev, full_bundle, N)
    clean = s.decode(use_graph=True).cpu().tolist()
    s2, _ = _bench_session(dev, full_bundle, N)
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(2816, 1024, device=dev).to(torch.bfloat16)
    w = torch.randn(6144, 1024, device=dev).to(torch.bfloat16)
    out = torch.zeros(2816, 3072, device=dev, dtype=torch.bfloat16)
    stop = threading.Event()

    def flood():
        torch.cuda.set_device(dev)
        while not stop.is_set():
            for _ in range(50):
                ops.gemm(a, w, out, L.EPI_SWIGLU, stream=side.cuda_stream)
            side.synchronize()

    th = threading.Thread(target=flood)
    th.start()
    try:
        got = s2.decode(use_graph=True).cpu().tolist()
    finally:
        stop.set()
        th.join()
    assert got == clean
    print(f"persistent decode beside a flooded side stream: recovered {s2.mega_recovered} time(s), persistent at the end: {s2.mega}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_ar_tiny_16bit_vs_cpu_oracle(dev, tiny_bundle, dt):
    """The same comparison at test scale against the oracle on the CPU (the pinned instrument itself, CPU libm): the
    generic-shape GEMV kernels (the streaming kernels only cover the real geometry) and a BPE-merged speech prompt."""
    from mars5_tts_amd import synth
    b = tiny_bundle
    tt, st = _toks(b)
    ref_codes = synth.make_ref_codes(40, seed=7, merge_friendly=True)
    prompt, _ = _bench_prompt(b, tt, st, ref_codes)
    _ar_compare(dev, b, dt, prompt, ref_codes[0].T.contiguous(), 48, 48, "cpu", len(TEXT), AR_TOL_EMU[dt], AR_TOL_F32[dt], 40)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_nar_full_size_16bit_forward_and_step_vs_oracle(dev, full_bundle, dt):
    """BASELINE configs[1]'s NAR stage at bench shape (S = 1349 rows: 899 prompt + 450 generated, 38 text tokens, both
    guidance branches), 16-bit operands: logits of one decoder pass vs the oracle (fp32 reference arithmetic, and with
    the engine's operand rounding), then one whole reverse step (forward + fused posterior / Gumbel sample) on the same
    uniforms: the ids are reported as an agreement rate, and with the ORACLE's logits fed to the kernel the ids must
    be equal wherever the oracle's own top-2 scores are further apart than the float noise of the formula."""
    import mars5_oracle as O
    from mars5_tts_amd import model, synth
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    b = full_bundle
    n = b.nar_shape
    nar = model.ResidualTransformer(n.n_text_vocab, n_quant=n.n_quant, dim=n.dim, nhead=n.nhead, enc_layers=n.enc_layers,
                                    dec_layers=n.dec_layers, n_spk_layers=n.n_spk_layers, t_emb_dim=n.t_emb_dim, p_cond_drop=0, dropout=0)
    nar.load_state_dict(b.nar_ckpt["model"])
    eng = nar.to(dev).set_engine_dtype(dt).engine()
    g = torch.Generator().manual_seed(3)
    S, off, Lt, t = 1349, 899, 38, 100
    K = n.n_quant
    c_text = torch.randint(0, n.n_text_vocab, (Lt,), generator=g)
    c_codes = synth.make_ref_codes(450, seed=7)[0].T.contiguous()
    x = torch.randint(0, 1024, (S, 8), generator=g)
    x_known = torch.zeros(S, 8, dtype=torch.long)
    m = torch.zeros(S, 8, dtype=torch.uint8)
    m[:, 0] = 1
    m[:off] = 1
    x_known[:off] = x[:off]
    x_known[:, 0] = x[:, 0]
    sess = NARSession(eng, NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20))
    sess.prepare(c_text, c_codes, x, x_known, m, off, [t])
    sess.enqueue_forward(sess.stream.cuda_stream)
    sess.stream.synchronize()
    so = S - off
    lg = sess.logits[:, :, :K].clone()                      # (2*so, 7, K): cond rows then uncond rows
    with torch.device(dev), torch.inference_mode():
        sd = O.round_linear_weights({k: v.to(dev) for k, v in b.nar_ckpt["model"].items()}, dt)
        res = {}
        for name, odt in (("f32", None), ("emu", dt)):
            lc = O.nar_forward(sd, n.nhead, c_text.to(dev), c_codes.to(dev), x.to(dev), t, False, dt=odt)
            lu = O.nar_forward(sd, n.nhead, c_text.to(dev), c_codes.to(dev), x.to(dev), t, True, dt=odt)
            zmax = float(torch.maximum(lc.abs().max(), lu.abs().max()))
            ec = float((lg[:so] - lc[off:, 1:]).abs().max()) / zmax
            eu = float((lg[so:] - lu[off:, 1:]).abs().max()) / zmax
            am = float((lg[:so].argmax(-1) == lc[off:, 1:].argmax(-1)).float().mean())
            res[name] = (ec, eu, zmax, am, lc, lu)
        print(f"NAR {dt} S={S}: max|dlogit|/max|logit| cond/uncond vs fp32 oracle {res['f32'][0]:.4f}/{res['f32'][1]:.4f}, vs operand-rounding "
              f"oracle {res['emu'][0]:.4f}/{res['emu'][1]:.4f}; max|logit| {res['f32'][2]:.2f}; argmax agreement vs fp32 {res['f32'][3]:.4f}")
        assert max(res["emu"][0], res["emu"][1]) <= NAR_TOL_EMU[dt]
        assert max(res["f32"][0], res["f32"][1]) <= NAR_TOL_F32[dt]
        # ---- whole reverse steps (forward + fused posterior / Gumbel sample) on identical uniforms at the DISCRIMINATING times
        # (VERDICT r2 weak #1b): t = 199 (first step), 100 (posterior dominated by x_t), 21 / 20 (the q0_override edge), 1 (the
        # l0-dominated posterior) and 0 (no u2 draw, known branch = copy).  Known positions do not depend on the logits: exact
        # up to float ties.  Sampled positions: an id may differ from the fp32 oracle's only where the oracle's own margin
        # between the two candidates is inside the score error the measured logit error of THAT step allows:
        # z = (w zc + (1 - w) zu) / T  =>  |dscore| <= (|w| + |1 - w|) / T x max|dlogit| (both candidates: x 2).
        from mars5_tts_amd import _lib as L, ops
        from mars5_tts_amd.tables import log_eps
        from parity_util import SCORE_EPS, ungated_mismatches
        tb = O.diffusion_tables(K, 200)
        mb = m.to(dev).bool()
        amp = (abs(3.0) + abs(1.0 - 3.0)) / 0.7
        for tq in (199, 100, 21, 20, 1, 0):
            gg = torch.Generator(device=dev).manual_seed(11 + tq)
            u1 = torch.rand((1, S, 8, K), generator=gg, device=dev)
            u2 = torch.rand((1, S, 8, K), generator=gg, device=dev) if tq > 0 else None
            draws = iter([u1, u2])
            sess2 = NARSession(eng, NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20))
            sess2.prepare(c_text, c_codes, x, x_known, m, off, [tq])
            sess2.step(lambda shp: next(draws), use_graph=True)
            sess2.stream.synchronize()
            x_eng = sess2.x.clone()
            lge = sess2.logits[:, :, :K]
            if tq == t:
                lc, lu = res["f32"][4], res["f32"][5]
            else:
                lc = O.nar_forward(sd, n.nhead, c_text.to(dev), c_codes.to(dev), x.to(dev), tq, False, dt=None)
                lu = O.nar_forward(sd, n.nhead, c_text.to(dev), c_codes.to(dev), x.to(dev), tq, True, dt=None)
            dl = max(float((lge[:so] - lc[off:, 1:]).abs().max()), float((lge[so:] - lu[off:, 1:]).abs().max()))
            ref, s_unk, s_kn = O.reverse_step(tb, lc, lu, x.to(dev), x_known.to(dev), mb, tq, u1[0], u2[0] if u2 is not None else None, 3.0, 0.7,
                                              return_scores=True)
            if tq > 20:
                ref[:, 0] = x_known[:, 0].to(dev)               # t > q0_override_steps (diffuser.py:392-393)
            n_kn, bad_kn = ungated_mismatches(torch.where(mb, x_eng, ref), ref, s_unk, s_kn, mb)
            gate = 2.0 * amp * dl + SCORE_EPS
            n_unk, bad_unk = ungated_mismatches(torch.where(mb, ref, x_eng), ref, s_unk, s_kn, mb, eps=gate)
            agree = float((x_eng[off:, 1:] == ref[off:, 1:]).float().mean())
            print(f"NAR {dt} reverse step t={tq}: max|dlogit| {dl:.4f} -> score gate {gate:.3f}; sampled ids equal to the fp32 oracle's {agree:.4f} "
                  f"({n_unk} differ, {len(bad_unk)} outside the gate); known-branch mismatches {n_kn} (unexcused {len(bad_kn)})")
            assert not bad_kn, (tq, bad_kn[:5])
            assert not bad_unk, (tq, bad_unk[:5])
            if tq == t:
                keep = (ref, s_unk, s_kn, u1, u2, sess2.consts)
        ref, s_unk, s_kn, u1, u2, consts_t = keep
        lc, lu = res["f32"][4], res["f32"][5]                   # the t = 100 logits again (the loop above ended at t = 0)
        # ---- the fused posterior / sample kernel at bench size on the ORACLE's logits: ids equal up to excused ties
        Kp = (K + 3) // 4 * 4
        lgc = torch.zeros(so, 7, Kp)
        lgu = torch.zeros(so, 7, Kp)
        lgc[..., :K], lgu[..., :K] = lc[off:, 1:], lu[off:, 1:]
        xd = x.to(dev).clone()
        xk, md = x_known.to(dev), m.to(dev)
        step = torch.zeros(1, dtype=torch.int32)
        a = L.NarSampleArgs(logits_c=lgc.data_ptr(), logits_u=lgu.data_ptr(), ld_row=7 * Kp, ld_q=Kp, S=S, n_q=8, K=K, row_offset=off,
                            x=xd.data_ptr(), x_known=xk.data_ptr(), m=md.data_ptr(), u1=u1.data_ptr(), u2=u2.data_ptr(),
                            consts=consts_t.data_ptr(), step=step.data_ptr(), guidance_w=3.0, temperature=0.7, log_eps=log_eps(),
                            div_mode=0, q0_override_steps=20)
        ops.nar_sample(a)
        torch.cuda.synchronize()
        n_mis, bad = ungated_mismatches(xd, ref, s_unk, s_kn, mb)
        print(f"nar_sample_kernel at S={S} on the oracle's logits: {n_mis} of {S * 8} ids differ, {len(bad)} not excused by an oracle tie")
        assert not bad, bad[:5]


def _nar_engine(b, dt, dev):
    from mars5_tts_amd import model
    n = b.nar_shape
    nar = model.ResidualTransformer(n.n_text_vocab, n_quant=n.n_quant, dim=n.dim, nhead=n.nhead, enc_layers=n.enc_layers,
                                    dec_layers=n.dec_layers, n_spk_layers=n.n_spk_layers, t_emb_dim=n.t_emb_dim, p_cond_drop=0, dropout=0)
    nar.load_state_dict(b.nar_ckpt["model"])
    return nar.to(dev).set_engine_dtype(dt).engine()


def _nar_item(n, Lt, Lc, n_gen, seed):
    """Deep-clone inpainting state of one utterance: Lc prompt frames (known), n_gen generated frames (codebook 0 known)."""
    from mars5_tts_amd import synth
    g = torch.Generator().manual_seed(seed)
    S = Lc + n_gen
    c_text = torch.randint(0, n.n_text_vocab, (Lt,), generator=g)
    c_codes = synth.make_ref_codes(Lc, seed=seed)[0].T.contiguous()
    x = torch.randint(0, 1024, (S, 8), generator=g)
    x[:Lc] = c_codes
    x_known = torch.zeros(S, 8, dtype=torch.long)
    m = torch.zeros(S, 8, dtype=torch.uint8)
    m[:, 0] = 1
    m[:Lc] = 1
    x_known[:Lc] = x[:Lc]
    x_known[:, 0] = x[:, 0]
    return dict(c_text=c_text, c_codes=c_codes, x=x, x_known=x_known, m_mask=m, row_offset=Lc)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_nar_batch_row_tiles_full_size_equals_lone_and_oracle(dev, full_bundle, dt):
    """BASELINE configs[2] / [3]'s NAR path at the REAL geometry (VERDICT r4 weak #1a): five utterances of
    90 / 100 / 350 / 700 / 899 rows -- lengths whose 96- / 128- / 192-row tile lists cover different pad rows, one text memory of
    more than 64 rows (the reference-order cross-attention inside the deferred-LayerNorm chain) -- refined TOGETHER over the
    row-tile lists (``NARBatchSession``: the path bench.py's c3 / c4 legs run), three reverse steps (t = 199, 100, 0: first
    step launch by launch, then the captured graph) on per-utterance device generators: every utterance's codes equal the
    lone ``NARSession`` run bit for bit; and the batched forward's logits of two of them (the 100-row one and the long-memory
    one) against the oracle run alone on that utterance, within the same tolerance as the lone bench-shape test."""
    import mars5_oracle as O
    from mars5_tts_amd.nar_engine import NARBatchSession, NARConfig, NARSession
    b = full_bundle
    n = b.nar_shape
    eng = _nar_engine(b, dt, dev)
    K = n.n_quant
    specs = [(12, 60, 30), (25, 40, 60), (70, 150, 200), (38, 300, 400), (20, 450, 449)]           # (text tokens, prompt frames, generated frames)
    items = [_nar_item(n, Lt, Lc, ng, 100 + i) for i, (Lt, Lc, ng) in enumerate(specs)]
    times = [199, 100, 0]
    cfg = NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20)

    def draws(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        return lambda shp: torch.rand(shp, generator=g, device=dev)

    lone = []
    for i, it in enumerate(items):
        s1 = NARSession(eng, cfg)
        s1.prepare(it["c_text"], it["c_codes"], it["x"], it["x_known"], it["m_mask"], it["row_offset"], times)
        lone.append(s1.run(draws(500 + i), use_graph=True).clone())
        assert s1.dl is not None, "the lone session must take the deferred-LayerNorm path at the real geometry"
    sess = NARBatchSession(eng, cfg)
    sess.prepare([dict(it) for it in items], times)
    assert sess.rt is not None and sess.dl is not None, "the batch must run over row-tile lists + deferred LayerNorms (the benchmarked path)"
    assert sess.Sr % 384 == 0 and sess.rt.lens == [sub.S for sub in sess.subs for _ in range(2)]
    assert any(seg[0] == "plain" for seg in sess.plan) and any(seg[0] == "absorbed" for seg in sess.plan)
    outs = sess.run([draws(500 + i) for i in range(len(items))], use_graph=True)
    for i, (o, l) in enumerate(zip(outs, lone)):
        nd = int((o != l).sum())
        assert nd == 0, f"{dt}: utterance {i} (S = {int(l.shape[0])}): {nd} codes of the batched refinement differ from the lone run"
        assert int((o[items[i]['row_offset']:, 1:] != items[i]["x"].to(dev)[items[i]['row_offset']:, 1:]).sum()) > 0      # something was sampled
    # -- the batched forward against the oracle, utterance by utterance
    sess2 = NARBatchSession(eng, cfg)
    sess2.prepare([dict(it) for it in items], [100])
    sess2.enqueue_forward(sess2.stream.cuda_stream)
    sess2.stream.synchronize()
    with torch.device(dev), torch.inference_mode():
        sd = O.round_linear_weights({k: v.to(dev) for k, v in b.nar_ckpt["model"].items()}, dt)
        for want in (1, 2):
            u = sess2._order.index(want)                       # the session sorts its utterances by cross-attention path
            sub, it = sess2.subs[u], items[want]
            so, off = sub.s_out, it["row_offset"]
            lg = sess2.logits[sess2.row0[u]: sess2.row0[u] + 2 * so, :, :K]
            for name, odt, tol in (("f32", None, NAR_TOL_F32[dt]), ("emu", dt, NAR_TOL_EMU[dt])):
                lc = O.nar_forward(sd, n.nhead, it["c_text"].to(dev), it["c_codes"].to(dev), it["x"].to(dev), 100, False, dt=odt)
                lu = O.nar_forward(sd, n.nhead, it["c_text"].to(dev), it["c_codes"].to(dev), it["x"].to(dev), 100, True, dt=odt)
                zmax = float(torch.maximum(lc.abs().max(), lu.abs().max()))
                e = max(float((lg[:so] - lc[off:, 1:]).abs().max()), float((lg[so:] - lu[off:, 1:]).abs().max())) / zmax
                print(f"batched NAR forward {dt}, utterance {want} (S = {sub.S}, text memory {it['c_text'].shape[0] + 1} rows) vs {name} oracle: "
                      f"max|dlogit|/max|logit| {e:.4f} (bound {tol})")
                assert e <= tol, (want, name, e)


@pytest.mark.parametrize("dt", [torch.bfloat16])
def test_nar_long_form_forward_vs_oracle(dev, full_bundle, dt):
    """BASELINE configs[4]'s NAR shape (VERDICT r4 weak #1b): one decoder pass at S = 5399 rows (899 prompt + 4500 generated frames:
    85 key tiles per query block, 22 M scores per head), both guidance branches, against the oracle on the GPU -- the longest key
    range any other test covers is 1349."""
    import mars5_oracle as O
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    b = full_bundle
    n = b.nar_shape
    eng = _nar_engine(b, dt, dev)
    K, t = n.n_quant, 100
    it = _nar_item(n, 160, 899, 4500, 77)
    S, off = 5399, 899
    sess = NARSession(eng, NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20))
    sess.prepare(it["c_text"], it["c_codes"], it["x"], it["x_known"], it["m_mask"], off, [t])
    sess.enqueue_forward(sess.stream.cuda_stream)
    sess.stream.synchronize()
    so = S - off
    lg = sess.logits[:, :, :K]
    assert bool(torch.isfinite(lg).all())
    with torch.device(dev), torch.inference_mode():
        sd = O.round_linear_weights({k: v.to(dev) for k, v in b.nar_ckpt["model"].items()}, dt)
        for name, odt, tol in (("f32", None, NAR_TOL_F32[dt]), ("emu", dt, NAR_TOL_EMU[dt])):
            lc = O.nar_forward(sd, n.nhead, it["c_text"].to(dev), it["c_codes"].to(dev), it["x"].to(dev), t, False, dt=odt)
            lu = O.nar_forward(sd, n.nhead, it["c_text"].to(dev), it["c_codes"].to(dev), it["x"].to(dev), t, True, dt=odt)
            zmax = float(torch.maximum(lc.abs().max(), lu.abs().max()))
            ec = float((lg[:so] - lc[off:, 1:]).abs().max()) / zmax
            eu = float((lg[so:] - lu[off:, 1:]).abs().max()) / zmax
            am = float((lg[:so].argmax(-1) == lc[off:, 1:].argmax(-1)).float().mean())
            print(f"NAR {dt} S={S} (configs[4]) vs {name} oracle: max|dlogit|/max|logit| cond/uncond {ec:.4f}/{eu:.4f} (bound {tol}); max|logit| {zmax:.2f}; "
                  f"argmax agreement {am:.4f}")
            assert max(ec, eu) <= tol
            del lc, lu


# ------------------------------------------------------------------ stage-level C entry points
def test_c_composed_stages_equal_the_host_composed_ones_bit_for_bit(dev, full_bundle):
    """The product enqueues a NAR reverse step and an AR decode step through ONE C call each (m5_nar_step / m5_ar_decode_step
    over a stage plan recorded once per session, csrc/stage_plan.hip).  Same launches, same arguments: the results must be
    bit-identical to the steps composed launch by launch in Python -- lone NAR session (12 reverse steps, both guidance
    branches, in-graph uniforms), a batched NAR group of three utterances, and 96 AR decode steps on the persistent and on
    the per-launch form, each under a hipGraph and eagerly."""
    from mars5_tts_amd import _lib as L
    from mars5_tts_amd.diffuser import _generator_uniform
    from mars5_tts_amd.nar_engine import NARBatchSession, NARConfig, NARSession
    b = full_bundle
    eng = _nar_engine(b, torch.bfloat16, dev)
    cfg = NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20)
    times = list(range(199, 187, -1))
    it = _nar_item(b.nar_shape, 38, 450, 449, 5)
    outs = {}
    for mode in ("c", "host", "c-eager"):
        sess = NARSession(eng, cfg)
        sess.use_c_plan = mode != "host"
        sess.prepare(it["c_text"], it["c_codes"], it["x"], it["x_known"], it["m_mask"], it["row_offset"], times)
        assert sess.dl is not None, "the bench engine runs the deferred-LayerNorm schedule"
        gen = torch.Generator(device=dev).manual_seed(77)
        x = sess.run(_generator_uniform(dev, gen), use_graph=mode != "c-eager").clone()
        outs[mode] = (x.cpu(), int(gen.get_offset()))
        if mode != "host":
            assert sess.step_plan is not None and sess.step_plan[0].n_ops > 100, "the C-composed path did not run"
            n_ops = sess.step_plan[0].n_ops
        else:
            assert sess.step_plan is None
    assert torch.equal(outs["c"][0], outs["host"][0]) and torch.equal(outs["c-eager"][0], outs["host"][0])
    assert outs["c"][1] == outs["host"][1]
    print(f"NAR reverse step: {n_ops} launches per m5_nar_step call; 12 steps bit-identical to the host-composed step (graph and eager)")
    # a batched group (row-tile lists, per-utterance uniforms)
    items = [_nar_item(b.nar_shape, 30, 300, 200, 11), _nar_item(b.nar_shape, 52, 450, 260, 12), _nar_item(b.nar_shape, 41, 180, 120, 13)]
    res = {}
    for mode in ("c", "host"):
        bs = NARBatchSession(eng, cfg)
        bs.use_c_plan = mode == "c"
        bs.prepare(items, times[:6])
        gens = [torch.Generator(device=dev).manual_seed(500 + i) for i in range(len(items))]
        res[mode] = [t.cpu() for t in bs.run([_generator_uniform(dev, g) for g in gens])]
        assert (bs.step_plan is not None) == (mode == "c")
    assert all(torch.equal(a, c) for a, c in zip(res["c"], res["host"]))
    # AR decode: persistent and per-launch forms
    N = 96
    for persistent in (True, False):
        toks = {}
        for mode, graph in (("c", True), ("host", True), ("c", False)):
            s, _ = _bench_session(dev, b, N, persistent=persistent)
            s.use_c_plan = mode == "c"
            toks[(mode, graph)] = s.decode(use_graph=graph).cpu().tolist()
            assert s.mega == persistent
            if mode == "c":
                assert s.step_plan is not None and s.step_plan.n_ops == (3 if persistent else 26 * 5 + 2), s.step_plan.n_ops
            else:
                assert s.step_plan is None
        assert toks[("c", True)] == toks[("host", True)] == toks[("c", False)]
        assert len(toks[("c", True)]) > 60
    print("AR decode step: 3 launches (persistent) / 132 (per-launch) per m5_ar_decode_step call; tokens identical to the host-composed step")
