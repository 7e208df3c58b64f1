"""Host-side logic that needs no GPU: schedule constants taken from the caller's diffusion object, the cross-attention
path plan of a batch, request wire format, reference-handle LRU bookkeeping."""
import os
import math

import numpy as np
import torch


def test_step_consts_follow_the_callers_diffusion():
    """The eight per-step scalars m5_nar_sample consumes must come from the MultinomialDiffusion the caller passed
    (reference q_pred / q_posterior read diff.log_*, diffuser.py:118-206): a 100-step schedule gives other constants than
    the default 200-step one, equal to the oracle's tables for that schedule, and out-of-range steps fail loudly."""
    import mars5_oracle as O
    from mars5_tts_amd.diffuser import MultinomialDiffusion, _tables
    from mars5_tts_amd.tables import nar_step_consts
    K = 1025
    for T in (200, 100, 37):
        diff = MultinomialDiffusion(K, timesteps=T)
        tb = O.diffusion_tables(K, T)
        times = [T - 1, T // 2, 1, 0]
        c = nar_step_consts(times, K, tables=_tables(diff))
        lnK = np.log(K)
        for i, t in enumerate(times):
            tm1 = max(t - 1, 0)
            want = torch.stack([tb.log_cumprod_alpha[tm1], tb.log_1_min_cumprod_alpha[tm1] - lnK, tb.log_alpha[t], tb.log_1_min_alpha[t] - lnK,
                                tb.log_cumprod_alpha[t], tb.log_1_min_cumprod_alpha[t] - lnK]).float()
            assert torch.equal(c[i, :6], want), (T, t)
            assert float(c[i, 6]) == float(t)
    assert not torch.equal(nar_step_consts([50], K, tables=_tables(MultinomialDiffusion(K, timesteps=100))),
                           nar_step_consts([50], K))
    try:
        nar_step_consts([150], K, tables=_tables(MultinomialDiffusion(K, timesteps=100)))
        raise SystemExit("a step outside the diffusion's range must not be accepted")
    except AssertionError:
        pass


def test_cross_plan_groups_utterances_by_path():
    """``make_cross_plan``: consecutive utterances that take the same cross-attention path form one run; the padded memory
    length is a function of the utterance's own memory length (so alone == inside any batch); fp32 engines and memories of
    more than 64 rows keep the reference's operation order; models whose H * Lp is not a whole number of 64-deep K-steps
    fall back to the next padded length."""
    from mars5_tts_amd.blocks import AbsorbedCross, CrossMemory, EncLayerW, make_cross_plan
    assert [AbsorbedCross.lp_of(le, 16) for le in (1, 39, 48, 49, 64, 65, 300)] == [48, 48, 48, 64, 64, 0, 0]
    assert AbsorbedCross.lp_of(39, 2) == 64 and AbsorbedCross.lp_of(39, 3) == 0          # 2 * 48 = 96 is not a multiple of 64; odd head counts
    D, H, T, Bm = 128, 2, 3, 2

    def mem(le):
        k = torch.zeros(T * Bm, H, le, 64, dtype=torch.bfloat16)
        return CrossMemory(k, None, le, 64, Bm, v_rows=torch.zeros_like(k))

    lw = EncLayerW(in_w=None, in_b=None, out_w=None, out_b=None, act_w=None, l2_w=None, l2_b=None, n1_w=None, n1_b=None, n2_w=None, n2_b=None)
    lw.ca_q_wT, lw.ca_out_w, lw.ca_q_b = torch.zeros(H, D, 64, dtype=torch.bfloat16), torch.zeros(D, D, dtype=torch.bfloat16), torch.zeros(D)
    les = [20, 64, 70, 90, 33]
    plan = make_cross_plan([lw, lw], [[mem(le) for le in les]] * 2, D, torch.bfloat16, "cpu")
    kinds = [(seg[0], seg[1].s0, seg[1].n_seq, seg[1].Lp) if seg[0] == "absorbed" else (seg[0], seg[1], seg[2]) for seg in plan]
    assert kinds == [("absorbed", 0, 4, 64), ("plain", 4, 8), ("absorbed", 8, 2, 64)], kinds
    assert plan[0][1].tab_seq.shape == (2 * 4, 8) and plan[0][1].A.shape == (2, 4, H * 64, D)
    plan32 = make_cross_plan([lw, lw], [[mem(le) for le in les]] * 2, D, torch.float32, "cpu")
    assert [seg[0] for seg in plan32] == ["plain"] and plan32[0][1:3] == (0, 10)


def test_request_wire_format_roundtrip():
    from mars5_tts_amd import sharding as sh
    g = torch.Generator().manual_seed(1)
    reqs = [sh.Request(i, torch.randint(0, 3000, (5 + i,), generator=g), torch.randint(0, 1024, (7 * (i + 1), 8), generator=g), seed=10 + i,
                       n_gen_est=100 + i, n_phones_gen=40 + i, max_len=900 + i) for i in range(4)]
    back = sh._unpack(sh._pack(reqs))
    for a, b in zip(reqs, back):
        assert (a.idx, a.seed, a.n_gen_est, a.n_phones_gen, a.max_len) == (b.idx, b.seed, b.n_gen_est, b.n_phones_gen, b.max_len)
        assert torch.equal(a.text_ids, b.text_ids) and torch.equal(a.ref_codes, b.ref_codes)
    assert sh._unpack(sh._pack([])) == []


def test_generator_uniform_fills_a_buffer_with_the_same_draws():
    """diffuser._generator_uniform(out=buf) -- what the engine's second-stream ring calls -- must produce the values of the
    reference's ``torch.rand(shape)`` on the same generator AND leave the generator where that call leaves it (the next
    draw is the same too), for a shape of the step's kind (not a multiple of the Philox unroll)."""
    from mars5_tts_amd.diffuser import _generator_uniform
    dev = torch.device("cpu")
    shape = (1, 37, 8, 1025)
    ga, gb = torch.Generator().manual_seed(123), torch.Generator().manual_seed(123)
    ua, ub = _generator_uniform(dev, ga), _generator_uniform(dev, gb)
    assert getattr(ua, "out_ok", False)
    buf = torch.empty(shape)
    for _ in range(3):
        a = ua(shape)
        b = ub(shape, out=buf)
        assert b.data_ptr() == buf.data_ptr() and torch.equal(a, b)
    assert torch.equal(torch.rand(5, generator=ga), torch.rand(5, generator=gb))


def test_workspace_laid_over_a_larger_one_shares_its_front():
    """SeqWorkspace(inside=ws): the sub-problem workspaces of a NAR step (one-branch layer 0, generated-rows last layer) are
    views of the FRONT of the main workspace's buffers -- same strides per sequence as a private workspace would have, no
    memory of their own -- and a too-large request is refused."""
    import pytest
    from mars5_tts_amd.blocks import SeqWorkspace
    dev, dt = torch.device("cpu"), torch.bfloat16
    ws = SeqWorkspace(2, 150, 128, 384, dt, dev, row_pad=64)
    one = SeqWorkspace(1, 150, 128, 384, dt, dev, row_pad=64, inside=ws)
    part = SeqWorkspace(2, 50, 128, 384, dt, dev, row_pad=64, inside=ws)
    for sub in (one, part):
        priv = SeqWorkspace(sub.B, sub.S, 128, 384, dt, dev, row_pad=64)
        for name in ("xn", "q", "k", "vt", "att", "hff"):
            a, b, big = getattr(sub, name), getattr(priv, name), getattr(ws, name)
            assert a.shape == b.shape and a.stride() == b.stride() and a.is_contiguous()
            assert a.data_ptr() == big.data_ptr() and a.numel() <= big.numel()
        sc, pc = sub.scatter(), priv.scatter()
        assert (sc.q_bs, sc.q_hs, sc.k_bs, sc.vt_bs, sc.vt_hs, sc.vt_ds, sc.rows_per_batch) == (pc.q_bs, pc.q_hs, pc.k_bs, pc.vt_bs, pc.vt_hs, pc.vt_ds, pc.rows_per_batch)
    one.q.fill_(1.0)
    assert float(ws.q[0].float().min()) == 1.0 and float(ws.q[1].float().abs().max()) == 0.0      # branch 0 of the main workspace, nothing else
    with pytest.raises(AssertionError):
        SeqWorkspace(3, 150, 128, 384, dt, dev, row_pad=64, inside=ws)


def test_step_consts_gathered_form_equals_the_scalar_form():
    """tables.nar_step_consts builds its (steps, 8) table with one gather per column; it must be bit-identical to the
    step-by-step scalar construction it replaced (fp32 table entry minus the python double ln K, per step), for the default
    schedule, a jumpy one and a caller's own diffusion tables."""
    from mars5_tts_amd import tables as T

    def scalar_form(times, K, tabs):
        la, l1ma, lca, l1mca = [t.detach().to("cpu", torch.float32) for t in tabs]
        lnK = np.log(K)
        rows = []
        for t in times:
            tm1 = max(t - 1, 0)
            rows.append(torch.stack([lca[tm1], l1mca[tm1] - lnK, la[t], l1ma[t] - lnK, lca[t], l1mca[t] - lnK,
                                     torch.tensor(float(t)), torch.tensor(0.0)]))
        return torch.stack(rows).float().contiguous()

    for steps, times in ((200, list(range(199, -1, -1))), (200, [199, 198, 199, 198, 3, 0, 1, 0]), (37, list(range(36, -1, -1)))):
        tabs = T.diffusion_log_tables(steps)
        got = T.nar_step_consts(times, 1025, tables=tabs)
        ref = scalar_form(times, 1025, tabs)
        assert got.shape == ref.shape == (len(times), 8) and got.dtype == torch.float32
        assert torch.equal(got, ref)
    assert torch.equal(T.nar_step_consts([5, 4], 77), scalar_form([5, 4], 77, T.diffusion_log_tables(200)))


def test_host_tables_ignore_the_ambient_default_device():
    """Every host-side table builder must return the SAME CPU tensor whatever the ambient default device is
    (``with torch.device(dev)`` in the parity tests / ``torch.set_default_device("cuda")`` in a user's program): round 3
    lost its driver records to a ``torch.tensor(list(times))`` that followed the ambient device and then indexed CPU
    tables.  The meta device stands in for "not the CPU" here: a factory call that forgets ``device=`` produces a meta
    tensor, and mixing it with the CPU tables raises (or yields a meta result, caught by the device assert)."""
    import io

    from mars5_tts_amd import minbpe, tables as T
    from mars5_tts_amd.diffuser import MultinomialDiffusion, _tables
    from mars5_tts_amd.trim import trim

    def build():
        diff_tabs = _tables(MultinomialDiffusion(1025, timesteps=200))
        return dict(
            sine=T.sine_pe(33, 64), rope=T.rope_table(64, 50), tse=T.timestep_inputs([199, 100, 0], 64), tse_odd=T.timestep_inputs([5], 63),
            **{f"dlt{i}": t for i, t in enumerate(T.diffusion_log_tables(200))},
            consts=T.nar_step_consts(list(range(199, -1, -1)), 1025), consts_own=T.nar_step_consts([150, 3, 0], 1025, tables=diff_tabs),
            eos=T.eos_penalty_table(40, 0.5, 50.0), log_eps=torch.tensor(T.log_eps(), device="cpu"))

    ref = build()
    with torch.device("meta"):
        got = build()
        y = torch.cat([torch.zeros(4096, device="cpu"), torch.ones(8192, device="cpu"), torch.zeros(4096, device="cpu")])
        yt, idx = trim(y, top_db=30)
        assert idx.device.type == "cpu" and yt.device.type == "cpu" and 0 < int(idx[0]) < int(idx[1]) < 16384
    for k, v in ref.items():
        assert got[k].device.type == "cpu", k
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    # the speech tokenizer's expansion table (device hand-off) is host data too
    st = minbpe.CodebookTokenizer()
    from mars5_tts_amd import synth
    b = synth.make_bundle("tiny", seed=0)
    st.load(io.BytesIO(b.ar_ckpt["vocab"]["speechtok.model"].encode()))
    a = st.expansion_csr()
    with torch.device("meta"):
        c = st.expansion_csr()
    for x_, y_ in zip(a[:2], c[:2]):
        assert y_.device.type == "cpu" and torch.equal(x_, y_)


def test_row_tile_lists_cover_exactly_the_real_rows():
    """blocks.RowTiles: for each tile height the listed tiles are exactly those that contain a real row of some sequence of the
    padded layout, ascending, numbered b * (Sr / BM) + t; the ctypes mirror carries the same counts."""
    from mars5_tts_amd.blocks import RowTiles
    lens, Sr = [835, 1, 384, 385, 2237, 2304], 2304
    rt = RowTiles(lens, Sr, torch.device("cpu"))
    for i, bm in enumerate((96, 128, 192)):
        tpb = Sr // bm
        got = rt.maps[i].tolist()
        want = [b * tpb + t for b in range(len(lens)) for t in range(tpb) if t * bm < lens[b]]
        assert got == want == sorted(got) and rt.n[i] == len(want) == int(rt.c.n[i])
        real_rows = {b * Sr + r for b, n in enumerate(lens) for r in range(n)}
        covered = {e * bm + r for e in got for r in range(bm)}
        assert real_rows <= covered and len(covered) - len(real_rows) < len(lens) * bm       # at most one partial tile per sequence
    assert int(rt.c.rows_per_seq) == Sr
    import pytest
    with pytest.raises(AssertionError):
        RowTiles([10], 320, torch.device("cpu"))                                              # not a multiple of 384


def test_deferred_layernorm_fold_is_the_same_linear_map():
    """blocks.fold_layer_dln + the identity the kernels implement (include/mars5_hip.h, M5DeferredLN), in fp32 on the host:
    LN(x; gamma, beta) W^T + b == r (xt W'^T - d s) + b' with xt = x - c for ANY per-row centre c, for the three folded
    projections of a decoder layer (in_proj behind norm1, cross-attention query behind norm2, the interleaved SwiGLU pair behind
    norm3); and the chain's centre bookkeeping (blocks.DeferredLN: producers alternate between two centre buffers, the first one
    after a chain start takes the LayerNorm's means without a delta)."""
    from mars5_tts_amd import _lib as L
    from mars5_tts_amd.blocks import DeferredLN, EncLayerW, SeqWorkspace, fold_layer_dln, interleave_rows, LAYERNORM_EPS
    torch.manual_seed(0)
    D, FF, M = 256, 384, 37
    p = "tfm.decoder.layers.0"
    sd = {f"{p}.self_attn.in_proj_weight": torch.randn(3 * D, D) / 16, f"{p}.self_attn.in_proj_bias": torch.randn(3 * D),
          f"{p}.multihead_attn.in_proj_weight": torch.randn(3 * D, D) / 16, f"{p}.multihead_attn.in_proj_bias": torch.randn(3 * D),
          f"{p}.activation.W.weight": torch.randn(FF, D) / 16, f"{p}.activation.V.weight": torch.randn(FF, D) / 16}
    for n in (1, 2, 3):
        sd[f"{p}.norm{n}.weight"], sd[f"{p}.norm{n}.bias"] = 1 + 0.3 * torch.randn(D), 0.2 * torch.randn(D)
    lw = EncLayerW(in_w=None, in_b=None, out_w=torch.zeros(D, D), out_b=None, act_w=None, l2_w=None, l2_b=None, n1_w=None, n1_b=None, n2_w=None, n2_b=None)
    act = interleave_rows(sd[f"{p}.activation.W.weight"], sd[f"{p}.activation.V.weight"])
    fold_layer_dln(sd, p, lw, act, torch.float32, "cpu")
    x = 2.0 * torch.randn(M, D) + 5.0 * torch.randn(M, 1)
    c = x.mean(dim=1, keepdim=True) + 0.5 * torch.randn(M, 1)             # a centre near, not at, the mean
    xt = x - c
    d = xt.mean(dim=1, keepdim=True)
    r = 1.0 / torch.sqrt(x.var(dim=1, unbiased=False, keepdim=True) + LAYERNORM_EPS)
    cases = [(sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"], 1, lw.in_w_f, lw.in_b_f, lw.in_s),
             (sd[f"{p}.multihead_attn.in_proj_weight"][:D], sd[f"{p}.multihead_attn.in_proj_bias"][:D], 2, lw.ca_q_w_f, lw.ca_q_b_f, lw.ca_q_rs),
             (act, None, 3, lw.act_w_f, lw.act_b_f, lw.act_s)]
    for W, b, n, Wf, bf, s in cases:
        ln = torch.nn.functional.layer_norm(x, (D,), sd[f"{p}.norm{n}.weight"], sd[f"{p}.norm{n}.bias"], LAYERNORM_EPS)
        want = ln @ W.T + (b if b is not None else 0.0)
        got = r * (xt @ Wf.T - d * s[None, :]) + bf[None, :]
        assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max()), n
    assert torch.equal(lw.ca_q_wT_f, lw.ca_q_w_f.view(D // 64, 64, D).permute(0, 2, 1).contiguous())
    # centre bookkeeping of a chain
    ws = SeqWorkspace(1, M, D, FF, torch.bfloat16, torch.device("cpu"))
    dl = DeferredLN(ws, torch.device("cpu"))
    start = dl.start()
    assert start.data_ptr() == dl.cen[0].data_ptr()
    p1 = dl.producer(ws)                                   # first producer after the start: the LayerNorm's means, no delta
    assert (p1.cen_in, p1.cen_out, p1.delta) == (dl.cen[0].data_ptr(), dl.cen[1].data_ptr(), None) and p1.mode == 1 and p1.np == D // 128
    cns = dl.consumer(lw.in_s, M=M)
    assert cns.mode == 2 and cns.delta == dl.delta.data_ptr() and cns.cen_in is None and cns.n_feat == D
    p2a = dl.producer(ws, r0=8, rows_bs=16, advance=False)     # a step split over two launches reads / writes the same pair of buffers
    p2b = dl.producer(ws, r0=24, rows_bs=16)
    assert (p2a.cen_in - 8 * 4, p2a.cen_out - 8 * 4) == (dl.cen[1].data_ptr(), dl.cen[0].data_ptr()) == (p2b.cen_in - 24 * 4, p2b.cen_out - 24 * 4)
    assert p2a.delta == dl.delta.data_ptr() + 8 * 4 and p2a.xt == ws.xn.data_ptr() + 8 * D * 2 and p2a.part == dl.part.data_ptr() + 8 * (D // 128) * 8
    p3 = dl.producer(ws)
    assert (p3.cen_in, p3.cen_out) == (dl.cen[0].data_ptr(), dl.cen[1].data_ptr())


def test_product_reads_no_environment_variable_but_its_two(monkeypatch):
    """The product (package + inference.py + hubconf.py) may read MARS5_DTYPE (the engines' default operand type, model.py)
    and M5_HIP_TOOLS / M5_HIP_TOOLS_LIB (which library _lib.py loads) and nothing else: every A/B knob of the host engines goes
    through ``_lib.tool_knob``, which answers with the default unless the tools library is loaded.  (The library itself:
    tests/test_abi_cpu.py checks that libmars5_hip.so does not even import getenv.)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "mars5-tts_amd", f) for f in sorted(os.listdir(os.path.join(root, "mars5-tts_amd"))) if f.endswith(".py")]
    files += [os.path.join(root, "inference.py"), os.path.join(root, "hubconf.py"), os.path.join(root, "mars5_tts_amd.py")]
    allowed = {"MARS5_DTYPE", "M5_HIP_TOOLS", "M5_HIP_TOOLS_LIB"}
    reads = []
    for f in files:
        src = open(f).read()
        for m in re.finditer(r"(os\.environ|os\.getenv|getenv\()", src):
            line = src[src.rfind("\n", 0, m.start()) + 1: src.find("\n", m.end())]
            names = set(re.findall(r"[\"']([A-Z][A-Z0-9_]+)[\"']", line))
            if os.path.basename(f) == "_lib.py" and "os.environ.get(name, default) if TOOLS else default" in line:
                continue                                                     # tool_knob itself: gated on the tools library
            reads.append((os.path.basename(f), line.strip(), names))
    assert reads, "the scan found nothing: pattern broken?"
    for f, line, names in reads:
        assert names and names <= allowed, f"{f}: `{line}` reads the environment outside the allowed set {sorted(allowed)}"
    from mars5_tts_amd import _lib as L
    assert not L.TOOLS, "the CPU suite runs on the product library"
    monkeypatch.setenv("M5_NAR_DLN", "0")
    monkeypatch.setenv("M5_AR_MEGA", "0")
    assert L.tool_knob("M5_NAR_DLN", "1") == "1" and L.tool_knob("M5_AR_MEGA", "1") == "1"


def test_row_tile_lists_carry_the_sequence_lengths():
    """M5RowTiles.seq_len: the lists of the three tile heights cover different pad rows (ceil(len / BM) * BM), so the kernels are
    told each sequence's own length (deferred-LayerNorm consumers give pad rows d = r = 0); sub-lists of a run of sequences are
    made ahead of the launch sequence (``prebuild``), not inside it."""
    from mars5_tts_amd.blocks import RowTiles, _rt_sub
    lens = [90, 90, 100, 100, 700, 700]
    rt = RowTiles(lens, 768, torch.device("cpu"))
    assert rt.len_dev.tolist() == lens and rt.c.seq_len == rt.len_dev.data_ptr() and rt.c.rows_per_seq == 768
    cover = [[-(-n // bm) * bm for n in lens] for bm in (96, 128, 192)]
    assert cover[0][0] == 96 and cover[1][0] == 128 and cover[2][0] == 192          # the case the advisor named: rows 96..191 of a 90-row sequence
    assert rt.n == [sum(c // bm for c in cv) for cv, bm in zip(cover, (96, 128, 192))]
    rt.prebuild([(2, 2), (4, 2)])
    assert set(rt._subs) == {(2, 2), (4, 2)}
    sub = rt._subs[(4, 2)]
    assert sub.len_dev.tolist() == [700, 700] and _rt_sub(rt, 4, 2) is sub.c and _rt_sub(rt, 0, 6) is rt.c
    assert sub.maps[1].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]            # 6 tiles of 128 per sequence, numbered from the run's first row


def test_multiply_shift_row_index_of_the_uniform_kernel():
    """m5_nar_uniforms finds the row of element e as (e * m) >> (32 + s) (nar_engine._magic_div): exact for every e below the
    stated bound -- all 32-bit e for K = 1025 classes -- and (0, 0) = "divide" where no 32-bit multiplier is exact (7 over 2^32)."""
    import random
    from mars5_tts_amd.nar_engine import _magic_div
    rnd = random.Random(5)
    for d, n_max in [(1025, 2 ** 32), (1024, 2 ** 32), (7, 2 ** 32), (7, 6000 * 8 * 7), (3, 2 ** 32), (641, 2 ** 32), (1, 2 ** 20), (2, 2 ** 32),
                     (65537, 2 ** 32), (4097, 2 ** 31), (999983, 2 ** 32), (1025, 6000 * 8 * 1025)]:
        m, s = _magic_div(d, n_max)
        if m == 0:
            continue
        assert m < 2 ** 32
        probes = [0, 1, d - 1, d, d + 1, n_max - 1, n_max - d, n_max // 2]
        probes += [k * d + o for k in (1, 2, 1000, (n_max - 1) // d, (n_max - 1) // d - 1, rnd.randrange(1, max(n_max // d, 2))) for o in (-1, 0, 1)]
        probes += [rnd.randrange(0, n_max) for _ in range(20000)]
        for e in probes:
            if 0 <= e < n_max:
                assert (e * m) >> (32 + s) == e // d, (d, n_max, e)
    assert _magic_div(1025)[0] != 0, "the real class count takes the multiply-shift path over the whole 32-bit range"
    assert _magic_div(7) == (0, 0) and _magic_div(7, 6000 * 8 * 7)[0] != 0      # no 32-bit multiplier divides by 7 over 2^32; over a real draw one does
