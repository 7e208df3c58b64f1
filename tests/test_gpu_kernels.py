"""Kernel-level parity on a real MI355X: every HIP kernel (through the C ABI) against a
plain torch-CPU fp32 computation of the same op / the CPU oracle.  Integer outputs are
compared exactly; floating point within the tolerance written in each test."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.float16, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}   # relative to max |ref|


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def _rel(out, ref):
    return float((out.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))


def _q(t, dt):
    return t.to(dt).float()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


# ------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,K", [(200, 300, 192), (130, 1025, 128), (70, 64, 448), (257, 129, 64)])
def test_gemm_epilogues(dev, dt, M, N, K):
    from mars5_tts_amd import _lib as L, ops
    a, w, b = _q(_rand((M, K), 1), dt), _q(_rand((N, K), 2), dt), _rand((N,), 3)
    ref = a @ w.T + b
    ad, wd, bd = a.to(dev, dt), w.to(dev, dt), b.to(dev)
    out = torch.zeros(M, N, device=dev)
    ops.gemm(ad, wd, out, L.EPI_F32, bias=bd)
    torch.cuda.synchronize()
    r = _rel(out.cpu(), ref)
    assert r < TOL[dt], f"EPI_F32 rel err {r}"
    # transpose detection: an asymmetric case must not match the transposed product
    out2 = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm(ad, wd, out2, L.EPI_DT, bias=bd)
    res = _rand((M, N), 4).to(dev)
    res0 = res.clone()
    ops.gemm(ad, wd, res, L.EPI_RESIDUAL, bias=bd)
    out3 = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm(ad, wd, out3, L.EPI_SILU_DT, bias=bd)
    torch.cuda.synchronize()
    assert _rel(out2.float().cpu(), ref) < max(TOL[dt], 8e-3 if dt != torch.float32 else 0)
    assert _rel((res - res0).cpu(), ref) < TOL[dt] * 2
    assert _rel(out3.float().cpu(), torch.nn.functional.silu(ref)) < max(TOL[dt], 8e-3 if dt != torch.float32 else 0)


@pytest.mark.parametrize("M,N,K,scale", [(300, 260, 1024, 1.0), (129, 200, 3072, 1.0), (257, 129, 64, 1.0), (200, 256, 1024, 1e-3), (130, 140, 512, 900.0),
                                         (2816, 1024, 1024, 1.0)])
def test_gemm_f32_split_f16_products(dev, M, N, K, scale):
    """M5_F32X3 (csrc/gemm.hip "X3"): fp32 operands multiplied as three split-f16 MFMA terms.  Against a float64 product of
    the same fp32 operands its error must stay at fp32 level -- within 4x the exact fp32-MFMA kernel's own error (which is
    accumulation-order noise) and below 2e-6 of max |ref| -- over every epilogue the fp32 engines use; small (1e-3)
    and large (900) activations included; the last shape takes the 64-row tiling (the operands are pre-scaled into f16's normal range, csrc/gemm.hip)."""
    from mars5_tts_amd import _lib as L, ops
    a, w, b = _rand((M, K), 11) * scale, _rand((N, K), 12), _rand((N,), 13) * scale
    ref = (a.double() @ w.double().T + b.double())
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    outs = {}
    for mode in ("exact", "f16x3"):
        prev = ops.set_f32_products(mode)
        try:
            o = torch.zeros(M, N, device=dev)
            ops.gemm(ad, wd, o, L.EPI_F32, bias=bd)
            res = (_rand((M, N), 14) * scale).to(dev)
            res0 = res.clone()
            ops.gemm(ad, wd, res, L.EPI_RESIDUAL, bias=bd)
            torch.cuda.synchronize()
            outs[mode] = (o.cpu(), (res - res0).cpu())
        finally:
            ops.set_f32_products(prev)
    e_exact, e_x3 = _rel(outs["exact"][0], ref), _rel(outs["f16x3"][0], ref)
    print(f"f32 GEMM {M}x{N}x{K} scale {scale}: rel err vs float64 exact-MFMA {e_exact:.2e}, split-f16 {e_x3:.2e}")
    assert e_x3 < 2e-6 and e_x3 < max(4 * e_exact, 5e-7)
    assert _rel(outs["f16x3"][1], ref) < 4e-6            # (the residual difference re-rounds against the old C)
    # transposition / operand-order check on an asymmetric case is implied by the float64 comparison above


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 384, 512, 1024, 3072])
def test_gemm_prefetched_fragment_loop_every_pipeline_depth(dev, dt, K):
    """The 96x128 region kernel's K loop (round 3: fragments of the next MFMA block read one block ahead, barrier between the
    blocks, slot refilled NSTAGE K-steps ahead, tails of 3 / 2 / 1 K-steps unrolled): every K-step count from 1 up through
    the prologue-only, tail-only and steady-state regimes, residual and plain 16-bit epilogues, against torch fp32 on the
    same 16-bit operands.  M / N are not tile multiples (row / column clamps of the scalar-base DMA offsets)."""
    from mars5_tts_amd import _lib as L, ops
    M, N = 250, 300
    a, w, b = _q(_rand((M, K), 11), dt), _q(_rand((N, K), 12, scale=K ** -0.5), dt), _rand((N,), 13)
    ref = a @ w.T + b
    ad, wd, bd = a.to(dev, dt), w.to(dev, dt), b.to(dev)
    res = _rand((M, N), 14).to(dev)
    res0 = res.clone()
    ops.gemm(ad, wd, res, L.EPI_RESIDUAL, bias=bd)
    out = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm(ad, wd, out, L.EPI_DT, bias=bd)
    torch.cuda.synchronize()
    assert _rel((res - res0).cpu(), ref) < TOL[dt] * 2, f"residual K={K}"
    assert _rel(out.float().cpu(), ref) < 8e-3, f"plain K={K}"
    # a second launch on the same buffers (graph-style back-to-back reuse of the stage slots / M0)
    ops.gemm(ad, wd, res, L.EPI_RESIDUAL, bias=bd)
    torch.cuda.synchronize()
    assert _rel((res - res0).cpu(), 2 * ref) < TOL[dt] * 2


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 5, 16, 17, 32])
def test_gemm_skinny_decode_shapes(dev, dt, M):
    """The M <= 32 weight-streaming path of m5_gemm (batched AR decode step) at the real projection shapes:
    every epilogue against fp32 torch on the dtype-rounded operands, and against the wide-tile kernel
    (reached by padding the call to 40 rows) to catch layout mistakes that a tolerance could hide."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import interleave_rows
    tol = max(TOL[dt], 8e-3)
    for (N, K, epi) in [(4608, 1536, L.EPI_DT), (1536, 1536, L.EPI_RESIDUAL), (7168, 1536, L.EPI_SWIGLU), (1536, 3584, L.EPI_RESIDUAL),
                        (4096, 1536, L.EPI_F32), (48, 64, L.EPI_F32), (160, 4096, L.EPI_DT)]:
        a = _q(_rand((M, K), 1 + M), dt)
        sc = 1.0 / math.sqrt(K)
        if epi == L.EPI_SWIGLU:
            w1, w3 = _q(_rand((N // 2, K), 2, sc * 4), dt), _q(_rand((N // 2, K), 3, sc * 4), dt)
            w = interleave_rows(w1, w3)
            ref = torch.nn.functional.silu(a @ w1.T) * (a @ w3.T)
        else:
            w = _q(_rand((N, K), 2, sc * 4), dt)
            ref = a @ w.T
        bias = _rand((N,), 4) if epi in (L.EPI_DT, L.EPI_F32) else None
        if bias is not None:
            ref = ref + bias
        ad, wd = a.to(dev, dt), w.to(dev, dt)
        bd = bias.to(dev) if bias is not None else None
        outs = []
        # the same rows through the skinny kernel (M <= 32) and through the wide-tile kernel (forced by padding the
        # call to 40 rows: the row count is what m5_gemm dispatches on; rows are independent)
        for Mc in (M, 40):
            ac = torch.zeros(Mc, K, device=dev, dtype=dt)
            ac[:M] = ad
            if epi == L.EPI_RESIDUAL:
                res0 = _rand((Mc, N), 5).to(dev)
                o = res0.clone()
                ops.gemm(ac, wd, o, epi)
                torch.cuda.synchronize()
                o = (o - res0)[:M].cpu()
            else:
                No = N // 2 if epi == L.EPI_SWIGLU else N
                o = torch.zeros(Mc + 2, No, device=dev, dtype=torch.float32 if epi == L.EPI_F32 else dt)   # 2 guard rows
                ops.gemm(ac, wd, o[:Mc], epi, bias=bd)
                torch.cuda.synchronize()
                assert float(o[Mc:].float().abs().max()) == 0.0, "rows >= M must not be written"
                o = o[:M].float().cpu()
            outs.append(o)
        r = _rel(outs[0], ref)
        assert r < tol * (2 if epi == L.EPI_SWIGLU else 1), f"M={M} N={N} K={K} epi={epi}: rel err {r}"
        assert _rel(outs[0], outs[1]) < tol, f"M={M} N={N} K={K} epi={epi}: skinny vs wide-tile kernel"


def _tools_build():
    from mars5_tts_amd import _lib as L
    return L.TOOLS


TOOLS_ONLY = "an A/B kernel of the tools library (measured slower in the NAR step, not shipped): run with M5_HIP_TOOLS=1"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_residual_layernorm_fused(dev, dt):
    """m5_gemm_residual_ln: x += A W^T + bias and xn = LayerNorm(x) from ONE launch (row statistics exchanged between
    the workgroups of a row tile) vs fp32 torch; repeated launches reuse the same scratch (launch tags);
    ineligible shapes report unsupported without launching."""
    from mars5_tts_amd import ops
    if not _tools_build():
        pytest.skip(TOOLS_ONLY)
    torch.manual_seed(0)
    os.environ["M5_GEMM_LN"] = "1"          # opt-in path (off by default: slower than two launches, see gemm16.hip)
    for (M, N, K) in [(2816, 1024, 1024), (1408, 1024, 3072), (100, 1024, 64), (2816, 256, 128)]:
        a = _q(_rand((M, K), 1), dt)
        w = _q(_rand((N, K), 2, 2.0 / math.sqrt(K)), dt)
        bias, g, b = _rand((N,), 3), 1.0 + 0.3 * _rand((N,), 4), 0.2 * _rand((N,), 5)
        x0 = _rand((M, N), 6, 3.0) + 5.0 * _rand((M, 1), 7)               # rows with very different means
        scratch = torch.zeros(256 + 768 * 30 * 16, dtype=torch.uint8, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        ad, wd = a.to(dev, dt), w.to(dev, dt)
        x = x0.to(dev).clone()
        xn = torch.zeros(M + 1, N, device=dev, dtype=dt)
        ref_x = x0.clone()
        for rep in range(3):                                                # the residual accumulates; counters re-arm
            ok = ops.gemm_residual_ln(ad, wd, x, bias.to(dev), g.to(dev), b.to(dev), 4e-5, xn[:M], scratch, tag=rep % 2,
                                      tag_step=step if rep else None)
            assert ok, (M, N, K)
            step += 1
            ref_x = ref_x + (a @ w.T + bias)
        torch.cuda.synchronize()
        ref_xn = torch.nn.functional.layer_norm(ref_x, (N,), g, b, 4e-5)
        assert int(scratch[:4].view(torch.int32)[0]) == 0, "a row-tile wait timed out"
        assert _rel(x.cpu(), ref_x) < TOL[dt] * 2, (M, N, K, _rel(x.cpu(), ref_x))
        r = float((xn[:M].float().cpu() - ref_xn).abs().max())
        assert r < (2e-2 if dt == torch.float16 else 6e-2), f"{(M, N, K)}: xn max abs err {r}"   # |xn| ~ 3
        assert float(xn[M].float().abs().max()) == 0.0
    # more row tiles than CUs, or N not a multiple of the tile width: the caller must fall back
    big = torch.zeros(96 * 40, 64, device=dev, dtype=dt)
    assert not ops.gemm_residual_ln(big, torch.zeros(1024, 64, device=dev, dtype=dt), torch.zeros(96 * 40, 1024, device=dev), None,
                                    torch.ones(1024, device=dev), torch.zeros(1024, device=dev), 4e-5,
                                    torch.zeros(96 * 40, 1024, device=dev, dtype=dt), torch.zeros(2 ** 22, dtype=torch.uint8, device=dev))
    os.environ.pop("M5_GEMM_LN", None)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [2816, 1408, 200, 16384])
def test_deferred_layernorm_chain(dev, dt, M):
    """M5DeferredLN: a residual GEMM (producer) leaves the centred 16-bit copy + per-tile row partials of the rows it updated,
    and the three kinds of consumer -- QKV scatter (Q / K and the transposed V section), SwiGLU pair, per-head-softmax scores --
    apply LayerNorm in their epilogues, against fp32 torch LayerNorm + Linear; rows with means far from 0 (the centre matters);
    the consumers move the row centres to the row means; a second producer then centres by them.  Sizes: the NAR step's
    M = 2816 / 1408, a ragged 200 and a batched group of 16,384 rows (other tile configurations, same arithmetic per row)."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import interleave_rows
    torch.manual_seed(0)
    D, K0, H, FF, Lp = 1024, 256, 16, 512, 48
    eps = 4e-5
    npart = D // 128
    a0 = _q(_rand((M, K0), 1), dt)
    w0 = _q(_rand((D, K0), 2, 2.0 / math.sqrt(K0)), dt)
    b0 = _rand((D,), 3)
    x0 = _rand((M, D), 4, 2.0) + 6.0 * _rand((M, 1), 5)                   # row means up to 6, spread ~1.2: |mean| / sigma up to ~5
    cen0 = (x0.mean(dim=1) + 0.3 * _rand((M,), 6)).contiguous()           # a centre NEAR the mean, as the chain provides
    x_ref = x0 + a0 @ w0.T + b0
    # -- producer
    x = x0.to(dev).clone()
    xt = torch.zeros(M + 1, D, device=dev, dtype=dt)
    part = torch.zeros(M, npart, 2, device=dev)
    cen = cen0.to(dev).clone()
    cen_b = torch.zeros(M, device=dev)                                      # the centre the producer used, for the next producer
    delta = torch.zeros(M, device=dev)                                      # d of each row, written by a consumer
    dlp = L.DeferredLN(mode=1, np=npart, xt=xt.data_ptr(), ld_xt=D, part=part.data_ptr(), cen_in=cen.data_ptr(), cen_out=cen_b.data_ptr(), delta=None,
                       s=None, s_bs=0, eps=0.0, n_feat=D, rows_bs=M)
    ops.gemm_dln(a0.to(dev, dt), w0.to(dev, dt), x, L.EPI_RESIDUAL, dlp, bias=b0.to(dev))
    torch.cuda.synchronize()
    assert _rel(x.cpu(), x_ref) < TOL[dt], _rel(x.cpu(), x_ref)
    assert torch.equal(cen_b.cpu(), cen0)
    xc = x.cpu() - cen0[:, None]                                            # what the copy and the partials describe (the kernel's own x)
    assert float((xt[:M].float().cpu() - xc).abs().max()) <= float(xc.abs().max()) * (2 ** -8 if dt == torch.bfloat16 else 2 ** -11) * 1.01
    assert float(xt[M].float().abs().max()) == 0.0
    ps = part.cpu()
    xc_t = xc.view(M, npart, 128)
    assert torch.allclose(ps[..., 0], xc_t.sum(-1), rtol=1e-4, atol=2e-3) and torch.allclose(ps[..., 1], (xc_t * xc_t).sum(-1), rtol=1e-4, atol=2e-3)
    # -- consumers: LayerNorm(gamma, beta) + Linear folded the way blocks.fold_layer_dln folds
    g, be = 1.0 + 0.3 * _rand((D,), 7), 0.2 * _rand((D,), 8)
    ln = torch.nn.functional.layer_norm(x.cpu(), (D,), g, be, eps)

    def fold(W, b):
        Wf = (W * g[None, :]).to(dt)
        return Wf.to(dev).contiguous(), (W @ be + (b if b is not None else 0.0)).to(dev).contiguous(), Wf.float().sum(1).to(dev).contiguous()

    def consumer(s_vec, d_out=True, rows_bs=M, s_bs=0):
        return L.DeferredLN(mode=2, np=npart, xt=None, ld_xt=0, part=part.data_ptr(), cen_in=None, cen_out=None, delta=delta.data_ptr() if d_out else None,
                            s=s_vec.data_ptr(), s_bs=s_bs, eps=eps, n_feat=D, rows_bs=rows_bs)

    tol = 1.2e-2 if dt == torch.bfloat16 else 2.5e-3                       # of max |ref|: one operand rounding + one weight rounding
    # QKV scatter (whole sequence = one batch entry; S rows padded to 64 for the transposed V)
    wq, bq = _rand((3 * D, D), 9, 1.5 / math.sqrt(D)), _rand((3 * D,), 10)
    wqf, bqf, sq = fold(wq, bq)
    ref = (ln @ wq.T + bq).view(M, 3, H, 64)
    Sp = (M + 63) // 64 * 64
    q = torch.zeros(1, H, M, 64, device=dev, dtype=dt)
    k = torch.zeros(1, H, M, 64, device=dev, dtype=dt)
    vt = torch.zeros(1, H, 64, Sp, device=dev, dtype=dt)
    sc = L.QkvScatter(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), rows_per_batch=M, n_heads=H, head_dim=64, q_bs=H * M * 64, q_hs=M * 64, q_rs=64,
                      k_bs=H * M * 64, k_hs=M * 64, k_rs=64, vt_bs=H * 64 * Sp, vt_hs=64 * Sp, vt_ds=Sp)
    ops.gemm_dln(xt[:M], wqf, None, L.EPI_QKV, consumer(sq, d_out=False), bias=bqf, scatter=sc)
    torch.cuda.synchronize()
    assert _rel(q[0].float().cpu(), ref[:, 0].permute(1, 0, 2)) < tol
    assert _rel(k[0].float().cpu(), ref[:, 1].permute(1, 0, 2)) < tol
    assert _rel(vt[0].float().cpu()[..., :M], ref[:, 2].permute(1, 2, 0)) < tol
    assert float(delta.abs().max()) == 0.0                                  # delta = NULL: nothing written
    # SwiGLU pair (+ the centres move to the row means)
    w1, w3 = _rand((FF, D), 11, 1.5 / math.sqrt(D)), _rand((FF, D), 12, 1.5 / math.sqrt(D))
    wsf, bsf, ss = fold(interleave_rows(w1, w3), None)
    hff = torch.zeros(M, FF, device=dev, dtype=dt)
    ops.gemm_dln(xt[:M], wsf, hff, L.EPI_SWIGLU, consumer(ss), bias=bsf)
    torch.cuda.synchronize()
    ref = torch.nn.functional.silu(ln @ w1.T) * (ln @ w3.T)
    assert _rel(hff.float().cpu(), ref) < 2 * tol
    assert float((cen0 + delta.cpu() - x.cpu().mean(dim=1)).abs().max()) < 1e-3          # centre + d = the row's mean
    # scores with per-head softmax: two sequences of M / 2 rows with their own A / c / s tables (batched launch, rows_bs)
    if M % 128 == 0:
        Ms = M // 2
        N = H * Lp
        A = _rand((2, N, D), 13, 2.0 / math.sqrt(D))
        c = _rand((2, N), 14)
        c.view(2, H, Lp)[:, :, 40:] = -1e30                                 # padded keys
        Af = (A * g[None, None, :]).to(dt)
        cf = (torch.einsum("bnk,k->bn", A, be) + c).to(dev).contiguous()
        sA = Af.float().sum(-1).to(dev).contiguous()
        P = torch.zeros(M, N, device=dev, dtype=dt)
        delta.zero_()
        ops.xattn_scores_dln(xt[:M], Ms * D, Af.to(dev).contiguous(), cf, P, Ms * N, Ms, H, Lp, 2, consumer(sA, rows_bs=Ms, s_bs=N))
        torch.cuda.synchronize()
        sc_ref = torch.stack([ln[b * Ms:(b + 1) * Ms] @ A[b].T + c[b] for b in range(2)]).view(2, Ms, H, Lp)
        p_ref = torch.softmax(sc_ref, dim=-1).reshape(M, N)
        assert float((P.float().cpu() - p_ref).abs().max()) < (2.5e-2 if dt == torch.bfloat16 else 5e-3)
        assert float(P.float().cpu().view(M, H, Lp)[:, :, 40:].abs().max()) == 0.0
        assert float((cen0 + delta.cpu() - x.cpu().mean(dim=1)).abs().max()) < 1e-3
    # -- a second producer, batched like the P.B GEMM (two sequences, rows_bs), centres by the means the consumer left
    if M % 128 == 0:
        Ms = M // 2
        a1 = _q(_rand((M, K0), 15), dt)
        w1b = _q(_rand((2, D, K0), 16, 2.0 / math.sqrt(K0)), dt)
        x2_ref = x.cpu() + torch.cat([a1[b * Ms:(b + 1) * Ms] @ w1b[b].T for b in range(2)]) + b0
        dlp2 = L.DeferredLN(mode=1, np=npart, xt=xt.data_ptr(), ld_xt=D, part=part.data_ptr(), cen_in=cen_b.data_ptr(), cen_out=cen.data_ptr(),
                            delta=delta.data_ptr(), s=None, s_bs=0, eps=0.0, n_feat=D, rows_bs=Ms)
        cen_now = (cen_b + delta).cpu().clone()                             # = the row means the consumer measured
        ops.gemm_dln(a1.to(dev, dt), w1b.to(dev, dt)[0], x, L.EPI_RESIDUAL, dlp2, bias=b0.to(dev), M=Ms, batch=2, sA=Ms * K0, sW=D * K0, sC=Ms * D, sBias=0)
        torch.cuda.synchronize()
        assert _rel(x.cpu(), x2_ref) < TOL[dt]
        assert torch.equal(cen.cpu(), cen_now)
        xc2 = (x.cpu() - cen_now[:, None]).view(M, npart, 128)
        assert torch.allclose(part.cpu()[..., 0], xc2.sum(-1), rtol=1e-4, atol=2e-3)
        assert float((xt[:M].float().cpu() - xc2.view(M, D)).abs().max()) <= float(xc2.abs().max()) * (2 ** -8 if dt == torch.bfloat16 else 2 ** -11) * 1.01


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_row_tile_lists_equal_the_dense_launch(dev, dt):
    """M5RowTiles through the C ABI (m5_gemm_ex / m5_xattn_scores_ex with `rt`, m5_attention with `q_len`): three sequences of
    90 / 500 / 700 real rows in a 768-row padded layout.  For the residual producer (flat and batched), the QKV and SwiGLU
    consumers, the per-head-softmax scores and self-attention: every REAL row equals the dense launch bit for bit, rows past
    every tile height's coverage of a sequence keep the poison they were filled with (unlisted tiles are neither read nor
    written), and the pad rows INSIDE a consumer's tiles -- whose partials no producer of that tile height wrote (here: poison
    that would give r = 1 / sqrt(eps)) -- come out as the bias alone (M5RowTiles.seq_len: d = r = 0), finite in f16."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import RowTiles, interleave_rows
    torch.manual_seed(0)
    D, K0, H, FF, Lp, Sr = 1024, 256, 16, 512, 48, 768
    lens = [90, 500, 700]
    B = len(lens)
    M = B * Sr
    npart, eps = D // 128, 4e-5
    rt = RowTiles(lens, Sr, dev)
    real = torch.zeros(M, dtype=torch.bool)
    beyond = torch.zeros(M, dtype=torch.bool)                 # rows past the coverage of the tallest tile (192): in no list
    for b, n in enumerate(lens):
        real[b * Sr: b * Sr + n] = True
        beyond[b * Sr + (n + 191) // 192 * 192: (b + 1) * Sr] = True
    pad_all = ~real & ~beyond                                 # pad rows some tile height covers
    pad96 = torch.zeros(M, dtype=torch.bool)                  # pad rows EVERY tile height covers
    for b, n in enumerate(lens):
        pad96[b * Sr + n: b * Sr + min((n + h - 1) // h * h for h in (96, 128, 192))] = True
    POISON = 7.0

    a0 = _q(_rand((M, K0), 1), dt).to(dev, dt)
    w0 = _q(_rand((D, K0), 2, 2.0 / math.sqrt(K0)), dt).to(dev, dt)
    b0 = _rand((D,), 3).to(dev)
    x0 = (_rand((M, D), 4, 2.0) + 6.0 * _rand((M, 1), 5)).to(dev)
    cen0 = (x0.mean(dim=1) + 0.3 * _rand((M,), 6).to(dev)).contiguous()

    def producer(rtc, batched):
        x = x0.clone()
        xt = torch.full((M, D), POISON, device=dev, dtype=dt)
        part = torch.full((M, npart, 2), -3.0, device=dev)   # poison: sum = -3, sumsq = -3 -> var clamps to 0 -> r = 1 / sqrt(eps) without seq_len
        cen_b = torch.full((M,), POISON, device=dev)
        dlp = L.DeferredLN(mode=1, np=npart, xt=xt.data_ptr(), ld_xt=D, part=part.data_ptr(), cen_in=cen0.data_ptr(), cen_out=cen_b.data_ptr(), delta=None,
                           s=None, s_bs=0, eps=0.0, n_feat=D, rows_bs=Sr if batched else M)
        if batched:
            ops.gemm_dln(a0, w0, x, L.EPI_RESIDUAL, dlp, bias=b0, M=Sr, batch=B, sA=Sr * K0, sW=0, sC=Sr * D, sBias=0, rt=rtc)
        else:
            ops.gemm_dln(a0, w0, x, L.EPI_RESIDUAL, dlp, bias=b0, rt=rtc)
        torch.cuda.synchronize()
        return x, xt, part, cen_b

    dense = producer(None, False)
    for batched in (False, True):
        got = producer(rt.c, batched)
        for name, d_, g_ in zip(("x", "centred copy", "partials", "centres"), dense, got):
            assert torch.equal(g_.cpu()[real], d_.cpu()[real]), f"producer (batched={batched}): {name} of real rows differs from the dense launch"
        assert torch.equal(got[0].cpu()[beyond], x0.cpu()[beyond]), "residual stream of unlisted rows was written"
        assert bool((got[1].float().cpu()[beyond] == POISON).all()) and bool((got[2].cpu()[beyond] == -3.0).all()) and bool((got[3].cpu()[beyond] == POISON).all())
    x, xt, part, _ = producer(rt.c, False)                     # the state the consumers see: pad rows past the producer's tiles hold poison
    g, be = 1.0 + 0.3 * _rand((D,), 7), 0.2 * _rand((D,), 8)

    def fold(W, b):
        Wf = (W * g[None, :]).to(dt)
        return Wf.to(dev).contiguous(), (W @ be + (b if b is not None else 0.0)).to(dev).contiguous(), Wf.float().sum(1).to(dev).contiguous()

    def consumer(s_vec, rows_bs=M, s_bs=0):
        return L.DeferredLN(mode=2, np=npart, xt=None, ld_xt=0, part=part.data_ptr(), cen_in=None, cen_out=None, delta=None,
                            s=s_vec.data_ptr(), s_bs=s_bs, eps=eps, n_feat=D, rows_bs=rows_bs)

    # -- QKV scatter consumer
    wqf, bqf, sq = fold(_rand((3 * D, D), 9, 1.5 / math.sqrt(D)), _rand((3 * D,), 10))

    def qkv(rtc):
        q = torch.full((B, H, Sr, 64), POISON, device=dev, dtype=dt)
        k = torch.full((B, H, Sr, 64), POISON, device=dev, dtype=dt)
        vt = torch.full((B, H, 64, Sr), POISON, device=dev, dtype=dt)
        sc = L.QkvScatter(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), rows_per_batch=Sr, n_heads=H, head_dim=64, q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64,
                          k_bs=H * Sr * 64, k_hs=Sr * 64, k_rs=64, vt_bs=H * 64 * Sr, vt_hs=64 * Sr, vt_ds=Sr)
        ops.gemm_dln(xt, wqf, None, L.EPI_QKV, consumer(sq), bias=bqf, scatter=sc, rt=rtc)
        torch.cuda.synchronize()
        # rows as the leading axis: (M, H, 64) each
        return (q.permute(0, 2, 1, 3).reshape(M, H, 64).float().cpu(), k.permute(0, 2, 1, 3).reshape(M, H, 64).float().cpu(),
                vt.permute(0, 3, 1, 2).reshape(M, H, 64).float().cpu()), (q, k, vt)

    (qd, kd, vd), _ = qkv(None)
    (qr, kr, vr), (q_dev, k_dev, vt_dev) = qkv(rt.c)
    bias_rows = [bqf.cpu()[i * D:(i + 1) * D].to(dt).float().view(H, 64) for i in range(3)]
    for name, d_, r_, brow in zip("qkv", (qd, kd, vd), (qr, kr, vr), bias_rows):
        assert torch.equal(r_[real], d_[real]), f"QKV consumer: {name} of real rows differs from the dense launch"
        assert bool((r_[beyond] == POISON).all()), f"QKV consumer wrote {name} rows of unlisted tiles"
        assert bool(torch.isfinite(r_).all())
        pr = r_[pad_all]
        is_bias = (pr == brow[None]).flatten(1).all(1)
        is_poison = (pr == POISON).flatten(1).all(1)
        assert bool((is_bias | is_poison).all()), f"QKV consumer: a pad row of {name} is neither untouched nor the bias alone (stale partials leaked: r != 0)"
        assert bool((r_[pad96] == brow[None]).all()), f"QKV consumer: pad rows inside every tile height's coverage must be the bias alone ({name})"
    # ... whereas the dense launch (no list, no lengths) amplifies the rows whose partials nobody wrote -- rows 128..191 of the
    # 90-row sequence: past the producer's 96- or 128-row tile, inside a 192-row consumer tile -- by 1 / sqrt(eps): the case ADVICE r4 #1 named
    assert float(qd[128:192].abs().max()) > 10 * float(bias_rows[0].abs().max())

    # -- SwiGLU consumer
    wsf, bsf, ss = fold(interleave_rows(_rand((FF, D), 11, 1.5 / math.sqrt(D)), _rand((FF, D), 12, 1.5 / math.sqrt(D))), None)

    def swiglu(rtc):
        hff = torch.full((M, FF), POISON, device=dev, dtype=dt)
        ops.gemm_dln(xt, wsf, hff, L.EPI_SWIGLU, consumer(ss), bias=bsf, rt=rtc)
        torch.cuda.synchronize()
        return hff.float().cpu()
    hd, hr = swiglu(None), swiglu(rt.c)
    assert torch.equal(hr[real], hd[real]) and bool((hr[beyond] == POISON).all()) and bool(torch.isfinite(hr).all())
    assert float(hr[pad_all].abs().max()) <= max(POISON, 1.0)     # silu(b') b' with b' = W beta: small; or untouched poison

    # -- scores with per-head softmax (batched: per-sequence operand tables)
    N = H * Lp
    A = _rand((B, N, D), 13, 2.0 / math.sqrt(D))
    c = _rand((B, N), 14)
    c.view(B, H, Lp)[:, :, 40:] = -1e30
    Af = (A * g[None, None, :]).to(dt).to(dev).contiguous()
    cf = (torch.einsum("bnk,k->bn", A, be) + c).to(dev).contiguous()
    sA = Af.float().sum(-1).contiguous()

    def scores(rtc):
        P = torch.full((M, N), POISON, device=dev, dtype=dt)
        ops.xattn_scores_dln(xt, Sr * D, Af, cf, P, Sr * N, Sr, H, Lp, B, consumer(sA, rows_bs=Sr, s_bs=N), rt=rtc)
        torch.cuda.synchronize()
        return P.float().cpu()
    pd_, pr_ = scores(None), scores(rt.c)
    assert torch.equal(pr_[real], pd_[real]) and bool((pr_[beyond] == POISON).all()) and bool(torch.isfinite(pr_).all())

    # -- self-attention with q_len: query blocks past a sequence's own length are not computed
    kl = torch.tensor(lens, dtype=torch.int32, device=dev)

    def attn(with_q_len):
        o = torch.full((M, D), POISON, device=dev, dtype=dt)
        a = L.AttnArgs(q=q_dev.data_ptr(), q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64, k=k_dev.data_ptr(), k_bs=H * Sr * 64, k_hs=Sr * 64, k_rs=64,
                       vt=vt_dev.data_ptr(), vt_bs=H * 64 * Sr, vt_hs=64 * Sr, vt_ds=Sr, o=o.data_ptr(), o_bs=Sr * D, o_rs=D, B=B, H=H, Sq=Sr, Sk=Sr,
                       key_len=kl.data_ptr(), causal=0, scale=64 ** -0.5, kv_index=None, kv_index_stride_k=0, kv_index_stride_v=0,
                       q_len=kl.data_ptr() if with_q_len else None)
        ops.attention(dt, a)
        torch.cuda.synchronize()
        return o.float().cpu()
    od, orr = attn(False), attn(True)
    assert torch.equal(orr[real], od[real]), "attention with q_len: real query rows differ"
    assert bool((orr[beyond] == POISON).all()) and not bool((od[beyond] == POISON).any())
    assert bool(torch.isfinite(orr[real]).all())


@pytest.mark.parametrize("dt", DTS)
def test_gemm_swiglu_and_qkv(dev, dt):
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import interleave_rows
    M, K, F = 150, 192, 136
    a = _q(_rand((M, K), 1), dt)
    w1, w3 = _q(_rand((F, K), 2), dt), _q(_rand((F, K), 3), dt)
    ref = torch.nn.functional.silu(a @ w1.T) * (a @ w3.T)
    h = torch.zeros(M, F, device=dev, dtype=dt)
    ops.gemm(a.to(dev, dt), interleave_rows(w1, w3).to(dev, dt), h, L.EPI_SWIGLU)
    torch.cuda.synchronize()
    assert _rel(h.float().cpu(), ref) < max(TOL[dt] * 2, 1e-2 if dt != torch.float32 else 0)
    # QKV scatter: B=2 sequences of S rows, H heads
    B, S, H = 2, 75, 3
    D = H * 64
    a = _q(_rand((B * S, D), 5), dt)
    w = _q(_rand((3 * D, D), 6, 0.2), dt)
    bias = _rand((3 * D,), 7)
    ref = (a @ w.T + bias).view(B, S, 3, H, 64)
    Sp = 128
    q = torch.zeros(B, H, S, 64, device=dev, dtype=dt)
    k = torch.zeros(B, H, S, 64, device=dev, dtype=dt)
    vt = torch.zeros(B, H, 64, Sp, device=dev, dtype=dt)
    sc = L.QkvScatter(q=q.data_ptr(), k=k.data_ptr(), vt=vt.data_ptr(), rows_per_batch=S, n_heads=H, head_dim=64,
                      q_bs=H * S * 64, q_hs=S * 64, q_rs=64, k_bs=H * S * 64, k_hs=S * 64, k_rs=64,
                      vt_bs=H * 64 * Sp, vt_hs=64 * Sp, vt_ds=Sp)
    ops.gemm(a.to(dev, dt), w.to(dev, dt), None, L.EPI_QKV, bias=bias.to(dev), scatter=sc)
    torch.cuda.synchronize()
    tol = max(TOL[dt], 8e-3 if dt != torch.float32 else 0)
    assert _rel(q.float().cpu(), ref[:, :, 0].permute(0, 2, 1, 3)) < tol
    assert _rel(k.float().cpu(), ref[:, :, 1].permute(0, 2, 1, 3)) < tol
    assert _rel(vt.float().cpu()[..., :S], ref[:, :, 2].permute(0, 2, 3, 1)) < tol
    assert float(vt[..., S:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTS)
def test_gemm_batched_heads(dev, dt):
    from mars5_tts_amd import _lib as L, ops
    nb, M, K, N = 3, 90, 128, 1025
    a = _q(_rand((nb, M, K), 1), dt)
    w = _q(_rand((nb, N, K), 2), dt)
    b = _rand((nb, N), 3)
    Kp = 1028
    out = torch.zeros(M, nb, Kp, device=dev)
    ad, wd = a.to(dev, dt), w.to(dev, dt)
    ops.gemm(ad[0], wd[0], out, L.EPI_F32, bias=b.to(dev), ldc=nb * Kp, batch=nb, sA=M * K, sW=N * K, sC=Kp, sBias=N)
    torch.cuda.synchronize()
    ref = torch.einsum("bmk,bnk->mbn", a, w) + b[None]
    assert _rel(out.cpu()[..., :N], ref) < TOL[dt]


# ------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("dt", DTS)
def test_norms(dev, dt):
    from mars5_tts_amd import ops
    M, D = 37, 192
    x = _rand((M, D), 1, 3.0) + 0.5
    g, b = 1 + _rand((4, D), 2, 0.2), _rand((4, D), 3, 0.2)
    out = torch.zeros(4, M, D, device=dev, dtype=dt)
    ops.layernorm(x.to(dev), g.to(dev), b.to(dev), 4e-5, out, n_affine=4, affine_stride=D, y_affine_stride=M * D)
    o2 = torch.zeros(M, D, device=dev, dtype=dt)
    ops.rmsnorm(x.to(dev), g[0].to(dev), 1e-5, o2)
    torch.cuda.synchronize()
    for i in range(4):
        ref = torch.nn.functional.layer_norm(x, (D,), g[i], b[i], 4e-5)
        assert _rel(out[i].float().cpu(), ref) < max(TOL[dt], 5e-3 if dt != torch.float32 else 0)
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * g[0]
    assert _rel(o2.float().cpu(), ref) < max(TOL[dt], 5e-3 if dt != torch.float32 else 0)


# ------------------------------------------------------------------------------ attention
def _ref_attention(q, k, v, key_len, causal):
    # q (B,H,Sq,64) k,v (B,H,Sk,64)
    s = (q @ k.transpose(-1, -2)) / 8.0
    Sq, Sk = q.shape[2], k.shape[2]
    mask = torch.zeros(q.shape[0], 1, Sq, Sk, dtype=torch.bool)
    for b, kl in enumerate(key_len):
        mask[b, :, :, kl:] = True
    if causal:
        mask |= torch.triu(torch.ones(Sq, Sk, dtype=torch.bool), diagonal=1)[None, None]
    s = s.masked_fill(mask, float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("dt", DTS)
def test_layernorm_twice(dev, dt):
    """m5_layernorm_twice: normalise(LayerNorm(x; gamma, beta, eps); eps2) in one pass, rows gathered from runs with a stride
    (the generated rows of the two guidance branches), against two torch LayerNorms in fp32."""
    from mars5_tts_amd import ops
    D, rps, n_seq, stride = 1024, 45, 2, 70
    x = _rand((n_seq * stride + 5, D), 1, 3.0) + 2.0 * _rand((n_seq * stride + 5, 1), 2)
    g, b = 1.0 + 0.3 * _rand((D,), 3), 0.2 * _rand((D,), 4)
    out = torch.zeros(n_seq * rps + 1, D, device=dev, dtype=dt)
    ops.layernorm_twice(x.to(dev)[3:], g.to(dev), b.to(dev), 4e-5, 1e-5, out[:n_seq * rps], rps, n_seq=n_seq, x_seq_stride=stride)
    torch.cuda.synchronize()
    rows = torch.cat([x[3 + s_ * stride: 3 + s_ * stride + rps] for s_ in range(n_seq)])
    ref = torch.nn.functional.layer_norm(torch.nn.functional.layer_norm(rows, (D,), g, b, 4e-5), (D,), None, None, 1e-5)
    assert _rel(out[:n_seq * rps].float().cpu(), ref) < max(TOL[dt] / 3, 5e-6), _rel(out[:n_seq * rps].float().cpu(), ref)
    assert float(out[n_seq * rps].float().abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("B,H,Sq,Sk,kls,causal", [(2, 3, 150, 150, [150, 77], False), (1, 2, 130, 130, [130], True),
                                                   (2, 2, 100, 41, [41, 41], False), (1, 1, 70, 300, [300], False),
                                                   # long key ranges (22 / 10 key tiles), ragged query blocks, a key limit inside the first tile
                                                   (2, 2, 700, 700, [700, 530], False), (1, 3, 95, 1349, [1349], False),
                                                   (2, 1, 130, 600, [20, 577], False)])
def test_attention(dev, dt, B, H, Sq, Sk, kls, causal):
    from mars5_tts_amd import _lib as L, ops
    q, k, v = _q(_rand((B, H, Sq, 64), 1, 2.0), dt), _q(_rand((B, H, Sk, 64), 2, 2.0), dt), _q(_rand((B, H, Sk, 64), 3), dt)
    ref = _ref_attention(q, k, v, kls, causal).permute(0, 2, 1, 3).reshape(B, Sq, H * 64)
    Skp = (Sk + 63) // 64 * 64
    vt = torch.zeros(B, H, 64, Skp, dtype=dt)
    vt[..., :Sk] = v.transpose(-1, -2).to(dt)
    qd, kd, vtd = q.to(dev, dt).contiguous(), k.to(dev, dt).contiguous(), vt.to(dev)
    o = torch.zeros(B, Sq, H * 64, device=dev, dtype=dt)
    kl = torch.tensor(kls, dtype=torch.int32, device=dev)
    a = L.AttnArgs(q=qd.data_ptr(), q_bs=H * Sq * 64, q_hs=Sq * 64, q_rs=64, k=kd.data_ptr(), k_bs=H * Sk * 64, k_hs=Sk * 64, k_rs=64,
                   vt=vtd.data_ptr(), vt_bs=H * 64 * Skp, vt_hs=64 * Skp, vt_ds=Skp, o=o.data_ptr(), o_bs=Sq * H * 64, o_rs=H * 64,
                   B=B, H=H, Sq=Sq, Sk=Sk, key_len=kl.data_ptr(), causal=1 if causal else 0, scale=0.125, kv_index=None,
                   kv_index_stride_k=0, kv_index_stride_v=0)
    ops.attention(dt, a)
    torch.cuda.synchronize()
    r = _rel(o.float().cpu(), ref)
    assert r < max(TOL[dt], 1e-2 if dt != torch.float32 else 1e-4), f"attention rel err {r}"
    if dt == torch.float32:
        # the split-f16 product mode of the fp32 engines (csrc/attention.hip attn_x3_kernel): same masks, key order and softmax
        # arithmetic, both products as three f16 MFMA terms -- within 4x the exact kernel's own distance from the fp32 reference
        prev = ops.set_f32_products("f16x3")
        try:
            o3 = torch.zeros_like(o)
            a.o = o3.data_ptr()
            ops.attention(dt, a)
            torch.cuda.synchronize()
        finally:
            ops.set_f32_products(prev)
        r3 = _rel(o3.cpu(), ref)
        print(f"fp32 attention rel err: exact {r:.2e}, split-f16 {r3:.2e}")
        assert r3 < max(4 * r, 2e-6), f"split-f16 attention rel err {r3} (exact {r})"


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Sq,Sk,kl,causal", [(300, 1349, 1349, False), (333, 333, 333, True), (200, 700, 530, False), (97, 64, 64, False),
                                             (130, 200, 129, False)])
def test_attention_forms_bit_identical(dev, dt, Sq, Sk, kl, causal):
    """The two 16-bit attention kernels (csrc/attention16.hip: attn16_kernel, 128-row workgroups that bring their own operands;
    attn16w_kernel, 96-row workgroups with a loader wave and the hand-ordered tile loop) compute the same arithmetic in the same
    order: the library picks by grid size only, so ONE problem (small grid: loader form) and the same problem as every sequence of
    a 36-sequence batch (large grid: attn16_kernel) must agree bit for bit -- which is also what keeps a batched NAR group equal to
    its lone calls whichever side of the rule each lands on."""
    from mars5_tts_amd import _lib as L, ops
    H, NB = 16, 36
    assert (Sq + 127) // 128 * H * 1 <= 512 < (Sq + 127) // 128 * H * NB
    q, k, v = _rand((1, H, Sq, 64), 11, 2.0), _rand((1, H, Sk, 64), 12, 2.0), _rand((1, H, Sk, 64), 13)
    Skp = (Sk + 63) // 64 * 64
    vt = torch.zeros(1, H, 64, Skp)
    vt[..., :Sk] = v.transpose(-1, -2)
    outs = []
    for B in (1, NB):
        qd, kd, vtd = (t.to(dev, dt).expand(B, *t.shape[1:]).contiguous() for t in (q, k, vt))
        o = torch.zeros(B, Sq, H * 64, device=dev, dtype=dt)
        kls = torch.full((B,), kl, dtype=torch.int32, device=dev)
        a = L.AttnArgs(q=qd.data_ptr(), q_bs=H * Sq * 64, q_hs=Sq * 64, q_rs=64, k=kd.data_ptr(), k_bs=H * Sk * 64, k_hs=Sk * 64, k_rs=64,
                       vt=vtd.data_ptr(), vt_bs=H * 64 * Skp, vt_hs=64 * Skp, vt_ds=Skp, o=o.data_ptr(), o_bs=Sq * H * 64, o_rs=H * 64,
                       B=B, H=H, Sq=Sq, Sk=Sk, key_len=kls.data_ptr(), causal=1 if causal else 0, scale=0.125, kv_index=None,
                       kv_index_stride_k=0, kv_index_stride_v=0)
        ops.attention(dt, a)
        torch.cuda.synchronize()
        outs.append(o.cpu())
    ref = _ref_attention(_q(q, dt), _q(k, dt), _q(v, dt), [kl], causal).permute(0, 2, 1, 3).reshape(1, Sq, H * 64)
    assert _rel(outs[0].float(), ref) < 1e-2
    for b in range(NB):
        assert torch.equal(outs[1][b], outs[0][0]), f"sequence {b} of the batch differs from the lone call"


def test_attention_kv_index(dev):
    from mars5_tts_amd import _lib as L, ops
    dt = torch.float32
    T, B, H, Sq, Sk = 3, 2, 2, 50, 20
    q = _rand((B, H, Sq, 64), 1)
    k, v = _rand((T, B, H, Sk, 64), 2), _rand((T, B, H, Sk, 64), 3)
    vt = torch.zeros(T, B, H, 64, 64)
    vt[..., :Sk] = v.transpose(-1, -2)
    qd, kd, vtd = q.to(dev), k.to(dev), vt.to(dev)
    o = torch.zeros(B, Sq, H * 64, device=dev)
    step = torch.tensor([2], dtype=torch.int32, device=dev)
    a = L.AttnArgs(q=qd.data_ptr(), q_bs=H * Sq * 64, q_hs=Sq * 64, q_rs=64, k=kd.data_ptr(), k_bs=H * Sk * 64, k_hs=Sk * 64, k_rs=64,
                   vt=vtd.data_ptr(), vt_bs=H * 64 * 64, vt_hs=64 * 64, vt_ds=64, o=o.data_ptr(), o_bs=Sq * H * 64, o_rs=H * 64,
                   B=B, H=H, Sq=Sq, Sk=Sk, key_len=None, causal=0, scale=0.125, kv_index=step.data_ptr(),
                   kv_index_stride_k=B * H * Sk * 64, kv_index_stride_v=B * H * 64 * 64)
    ops.attention(dt, a)
    torch.cuda.synchronize()
    ref = _ref_attention(q, k[2], v[2], [Sk, Sk], False).permute(0, 2, 1, 3).reshape(B, Sq, H * 64)
    assert _rel(o.cpu(), ref) < 1e-4


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_q_cross_attention_fused(dev, dt):
    """m5_gemm_q_cross_attn: query projection + cross-attention against short pre-projected memories in one launch,
    vs fp32 torch.  Three sequences of 112 rows (not a multiple of the 96-row tile: tiles straddle sequences) attend
    to memories of different lengths (39, 64, 17 keys) inside the block selected by a device step index."""
    from mars5_tts_amd import ops
    from mars5_tts_amd.blocks import CrossMemory, cross_memory_table
    if not _tools_build():
        pytest.skip(TOOLS_ONLY)
    os.environ["M5_GEMM_XATTN"] = "1"       # opt-in path (off by default: no gain inside the NAR step, see gemm16.hip)
    H, Kd, Sr, T, step_i = 16, 1024, 112, 3, 1
    D = H * 64
    les = [39, 64, 17]
    M = Sr * len(les) - 5                                   # ragged tail: the last rows do not exist
    a = _q(_rand((Sr * len(les), Kd), 1), dt)
    w = _q(_rand((D, Kd), 2, 3.0 / math.sqrt(Kd)), dt)
    bias = _rand((D,), 3, 0.5)
    q = _q(a @ w.T + bias, dt).view(len(les), Sr, H, 64)    # the separate kernels round q to the operand type too
    mems, ref = [], torch.zeros(len(les) * Sr, D)
    for s_i, le in enumerate(les):
        lep = (le + 63) // 64 * 64
        k = _q(_rand((T, H, le, 64), 10 + s_i, 2.0), dt)
        v = _q(_rand((T, H, le, 64), 20 + s_i), dt)
        vt = torch.zeros(T, H, 64, lep)
        vt[..., :le] = v.transpose(-1, -2)
        mems.append(CrossMemory(k.to(dev, dt), vt.to(dev, dt), le, lep, 1))
        sc = torch.einsum("shd,hnd->hsn", q[s_i], k[step_i]) * 0.125
        o = torch.einsum("hsn,hnd->shd", torch.softmax(sc, -1), v[step_i])
        ref[s_i * Sr:(s_i + 1) * Sr] = o.reshape(Sr, D)
    tab, max_le = cross_memory_table(mems, dev)
    out = torch.zeros(len(les) * Sr, D, device=dev, dtype=dt)
    step = torch.tensor([step_i], dtype=torch.int32, device=dev)
    ok = ops.gemm_q_cross_attn(a.to(dev, dt)[:M], w.to(dev, dt), bias.to(dev), H, tab, max_le, Sr, step, 0.125, out)
    torch.cuda.synchronize()
    assert ok
    r = _rel(out[:M].float().cpu(), ref[:M])
    assert r < (1e-2 if dt == torch.float16 else 4e-2), f"fused cross-attention rel err {r}"
    assert float(out[M:].float().abs().max()) == 0.0, "rows >= M must not be written"
    # memory longer than 64 keys: not eligible, nothing launched
    assert not ops.gemm_q_cross_attn(a.to(dev, dt)[:M], w.to(dev, dt), bias.to(dev), H, tab, 65, Sr, step, 0.125, out)
    os.environ.pop("M5_GEMM_XATTN", None)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("les", [[39, 20], [57, 64], [48, 70, 39]])
def test_absorbed_cross_attention_vs_reference_order(dev, dt, les):
    """blocks.AbsorbedCross (m5_xattn_absorb + m5_xattn_scores + residual GEMM) against the reference's operation order
    (q-projection, per-head softmax(q k^T / 8) v, out-projection + bias, residual add) in fp32 torch on the same 16-bit
    operands: two layers, utterances with different memory lengths in one workspace (padded lengths 48 and 64, and a
    memory of 70 rows that must take the plain path), two guidance branches each, the memory block of device step 1."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import CrossMemory, EncLayerW, SeqWorkspace, cross_attn_block, make_cross_plan
    H, D, FF, S, T, step_i, Bm, NL = 4, 256, 768, 150, 3, 1, 2, 2
    U = len(les)
    ws = SeqWorkspace(U * Bm, S, D, FF, dt, dev, row_pad=64)
    Sr = ws.Sr
    layers, mems_per_layer, refs = [], [], []
    x0 = _rand((U * Bm * Sr, D), 1, 2.0)
    xn = _q(_rand((U * Bm * Sr, D), 2), dt)
    for l in range(NL):
        wq = _q(_rand((D, D), 10 + l, 3.0 / math.sqrt(D)), dt)
        wo = _q(_rand((D, D), 20 + l, 3.0 / math.sqrt(D)), dt)
        bq, bo = _rand((D,), 30 + l, 0.5), _rand((D,), 40 + l, 0.5)
        lw = EncLayerW(in_w=None, in_b=None, out_w=None, out_b=None, act_w=None, l2_w=None, l2_b=None, n1_w=None, n1_b=None, n2_w=None, n2_b=None)
        lw.ca_q_w, lw.ca_q_b = wq.to(dev, dt), bq.to(dev)
        lw.ca_out_w, lw.ca_out_b = wo.to(dev, dt), bo.to(dev)
        lw.ca_q_wT = lw.ca_q_w.view(H, 64, D).permute(0, 2, 1).contiguous()
        layers.append(lw)
        mems, ref = [], x0.clone()
        q = _q(xn @ wq.T + bq, dt).view(U * Bm, Sr, H, 64)
        for u, le in enumerate(les):
            lep = (le + 63) // 64 * 64
            k = _q(_rand((T * Bm, H, le, 64), 100 + 10 * l + u, 2.0), dt)
            v = _q(_rand((T * Bm, H, le, 64), 200 + 10 * l + u), dt)
            vt = torch.zeros(T * Bm, H, 64, lep)
            vt[..., :le] = v.transpose(-1, -2)
            mems.append(CrossMemory(k.to(dev, dt), vt.to(dev, dt), le, lep, Bm, v_rows=v.to(dev, dt)))
            for b in range(Bm):
                sq = u * Bm + b
                sc = torch.einsum("shd,hnd->hsn", q[sq], k[step_i * Bm + b]) * 0.125
                o = _q(torch.einsum("hsn,hnd->shd", torch.softmax(sc, -1), v[step_i * Bm + b]).reshape(Sr, D), dt)
                ref[sq * Sr:(sq + 1) * Sr] += o @ wo.T + bo
        mems_per_layer.append(mems)
        refs.append(ref)
    plan = make_cross_plan(layers, mems_per_layer, D, dt, dev)
    kinds = [seg[0] for seg in plan]
    assert kinds.count("plain") == sum(1 for le in les if le > 64) and "absorbed" in kinds
    step = torch.tensor([step_i], dtype=torch.int32, device=dev)
    for seg in plan:
        if seg[0] == "absorbed":
            seg[1].build(step)
    ws.xn.copy_(xn.to(dev, dt))
    for l in range(NL):
        x = x0.clone().to(dev)
        cross_attn_block(x, layers[l], ws, mems_per_layer[l], step, None, normed=True, plan=plan, layer=l)
        torch.cuda.synchronize()
        for sq in range(U * Bm):
            got, want = x[sq * Sr: sq * Sr + S].cpu(), refs[l][sq * Sr: sq * Sr + S]
            r = _rel(got - x0[sq * Sr: sq * Sr + S], want - x0[sq * Sr: sq * Sr + S])
            assert r < (1e-2 if dt == torch.float16 else 5e-2), f"layer {l} sequence {sq} (Le {les[sq // Bm]}): rel err {r}"


# ------------------------------------------------------------------------------ gathers / rope
def test_gather_and_chunked(dev):
    from mars5_tts_amd import ops
    from mars5_tts_amd.tables import sine_pe
    D, R = 128, 9
    table = _rand((50, D), 1)
    idx = torch.tensor([3, 49, 0, 7, 7, 12, 1, 2, 30])
    pe = sine_pe(40, D)
    alpha = torch.tensor([0.7])
    add = _rand((4, D), 2)
    pos = torch.tensor([0, 1, 2, 0, 1, 2, 5, 6, 7], dtype=torch.int32)
    aidx = torch.tensor([0, 0, 0, 1, 1, 1, 3, 3, 3], dtype=torch.int32)
    out = torch.zeros(R, D, device=dev)
    ops.gather_rows(out, table.to(dev), idx.to(dev), alpha.to(dev), pe.to(dev), pos.to(dev), add.to(dev), aidx.to(dev))
    torch.cuda.synchronize()
    ref = table[idx] * 1.0 + alpha * pe[pos.long()] + add[aidx.long()]
    assert torch.equal(out.cpu(), ref), float((out.cpu() - ref).abs().max())
    tables = _rand((8, 1025, D // 8), 3)
    codes = torch.randint(0, 1025, (R - 1, 8), generator=torch.Generator().manual_seed(4))
    lead = _rand((1, D), 5)
    o2 = torch.zeros(2, R, D, device=dev)
    si = torch.tensor([2], dtype=torch.int32, device=dev)
    ops.chunked_embed(o2, tables.to(dev), codes.to(dev), lead.to(dev), alpha.to(dev), pe.to(dev), add.to(dev), si)
    torch.cuda.synchronize()
    body = torch.cat([tables[q][codes[:, q]] for q in range(8)], dim=-1)
    ref2 = torch.cat([lead, body]) * 1.0 + alpha * pe[:R] + add[2]
    assert torch.equal(o2[0].cpu(), ref2) and torch.equal(o2[1].cpu(), ref2)


@pytest.mark.parametrize("dt", DTS)
def test_rope_cache(dev, dt):
    import mars5_oracle as O
    from mars5_tts_amd import ops
    from mars5_tts_amd.tables import rope_table
    M, H, W = 21, 3, 32
    D = H * 64
    qkv = _q(_rand((M, 3 * D), 1), dt)
    rope = rope_table(64, 64)
    fc = O.precompute_freqs_cis(64, M)
    qr = O.apply_rotary(qkv[:, :D].view(M, H, 64), fc)
    kr = O.apply_rotary(qkv[:, D:2 * D].view(M, H, 64), fc)
    q = torch.zeros(H, M, 64, device=dev, dtype=dt)
    kc = torch.zeros(H, W, 64, device=dev, dtype=dt)
    vc = torch.zeros(H, W, 64, device=dev, dtype=dt)
    vt = torch.zeros(H, 64, 64, device=dev, dtype=dt)
    ops.rope_cache(qkv.to(dev, dt), H, 0, rope.to(dev), q, kc, vc, W * 64, W, vt, 64 * 64, 64)
    torch.cuda.synchronize()
    tol = 1e-6 if dt == torch.float32 else 8e-3
    assert _rel(q.float().cpu(), qr.permute(1, 0, 2)) < tol
    assert _rel(kc.float().cpu()[:, :M], kr.permute(1, 0, 2)) < tol
    assert torch.equal(vc.float().cpu()[:, :M], qkv[:, 2 * D:].view(M, H, 64).permute(1, 0, 2))
    assert torch.equal(vt.float().cpu()[:, :, :M], qkv[:, 2 * D:].view(M, H, 64).permute(1, 2, 0))


# ------------------------------------------------------------------------------ decode kernels
@pytest.mark.parametrize("dt", DTS)
def test_ar_gemv_variants(dev, dt):
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.blocks import interleave_rows
    from mars5_tts_amd.tables import rope_table
    H, F = 3, 448
    D = H * 64
    tol = max(TOL[dt] * 2, 1e-2 if dt != torch.float32 else 0)
    x = _rand((D,), 1, 2.0)
    nw = 1 + _rand((D,), 2, 0.1)
    xn = _q(O.rmsnorm(x[None], nw, 1e-5)[0], dt)
    state = torch.tensor([5, 0, 0, 5, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    # --- RMS + QKV + RoPE + cache write at pos 5
    wqkv = _q(_rand((3 * D, D), 3, 0.1), dt)
    rope = rope_table(64, 16)
    W = 12
    kc = torch.zeros(H, W, 64, device=dev, dtype=dt)
    vc = torch.zeros(H, W, 64, device=dev, dtype=dt)
    qb = torch.zeros(D, device=dev, dtype=dt)
    keep = [x.to(dev), nw.to(dev), wqkv.to(dev, dt), rope.to(dev)]
    a = L.GemvArgs(W=keep[2].data_ptr(), ldw=D, N=3 * D, K=D, x_f32=keep[0].data_ptr(), norm_w=keep[1].data_ptr(), eps=1e-5,
                   rope=keep[3].data_ptr(), state=state.data_ptr(), kcache=kc.data_ptr(), vcache=vc.data_ptr(), qbuf=qb.data_ptr(),
                   w_alloc=W, window=3000, dim=D)
    ops.ar_gemv(dt, L.PRO_RMS, L.GEPI_QKV_ROPE, a)
    torch.cuda.synchronize()
    y = _q(wqkv @ xn, dt)
    fc = O.precompute_freqs_cis(64, 6)[5:6]
    qr = O.apply_rotary(y[:D].view(1, H, 64), fc)[0]
    kr = O.apply_rotary(y[D:2 * D].view(1, H, 64), fc)[0]
    assert _rel(qb.float().cpu().view(H, 64), qr) < tol
    assert _rel(kc.float().cpu()[:, 5], kr) < tol
    assert _rel(vc.float().cpu()[:, 5], y[2 * D:].view(H, 64)) < tol
    assert float(kc[:, :5].abs().max()) == 0 and float(kc[:, 6:].abs().max()) == 0
    # --- RMS + SwiGLU
    w1, w3 = _q(_rand((F, D), 4, 0.1), dt), _q(_rand((F, D), 5, 0.1), dt)
    hb = torch.zeros(F, device=dev, dtype=dt)
    w13 = interleave_rows(w1, w3).to(dev, dt)
    a = L.GemvArgs(W=w13.data_ptr(), ldw=D, N=2 * F, K=D, x_f32=keep[0].data_ptr(), norm_w=keep[1].data_ptr(), eps=1e-5,
                   y_dt=hb.data_ptr(), state=state.data_ptr())
    ops.ar_gemv(dt, L.PRO_RMS, L.GEPI_SWIGLU, a)
    torch.cuda.synchronize()
    href = torch.nn.functional.silu(w1 @ xn) * (w3 @ xn)
    assert _rel(hb.float().cpu(), href) < tol
    # --- DT + residual  (w2)
    w2 = _q(_rand((D, F), 6, 0.1), dt)
    hq = _q(href, dt)
    xr = x.clone().to(dev)
    w2d, hd = w2.to(dev, dt), hq.to(dev, dt)
    a = L.GemvArgs(W=w2d.data_ptr(), ldw=F, N=D, K=F, x_dt=hd.data_ptr(), xres=xr.data_ptr(), state=state.data_ptr())
    ops.ar_gemv(dt, L.PRO_DT, L.GEPI_RESIDUAL, a)
    torch.cuda.synchronize()
    assert _rel(xr.cpu() - x, w2 @ hq) < tol
    # --- RMS + logits
    V = 1376
    wo = _q(_rand((V, D), 7, 0.2), dt)
    lg = torch.zeros(V, device=dev)
    wod = wo.to(dev, dt)
    a = L.GemvArgs(W=wod.data_ptr(), ldw=D, N=V, K=D, x_f32=keep[0].data_ptr(), norm_w=keep[1].data_ptr(), eps=1e-5,
                   y_f32=lg.data_ptr(), state=state.data_ptr())
    ops.ar_gemv(dt, L.PRO_RMS, L.GEPI_F32, a)
    torch.cuda.synchronize()
    assert _rel(lg.cpu(), wo @ xn) < TOL[dt] * 2
    # --- done flag makes every step kernel a no-op
    state[L.ST_DONE] = 1
    lg.zero_()
    ops.ar_gemv(dt, L.PRO_RMS, L.GEPI_F32, a)
    torch.cuda.synchronize()
    assert float(lg.abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("pos,W,window", [(0, 40, 3000), (37, 64, 3000), (700, 800, 3000), (95, 32, 32)])
def test_attn_decode_and_combine(dev, dt, pos, W, window):
    from mars5_tts_amd import _lib as L, ops
    H, NS = 3, 8
    D = H * 64
    n_valid = min(pos + 1, window)
    q = _q(_rand((H, 64), 1, 2.0), dt)
    k, v = _q(_rand((H, W, 64), 2, 2.0), dt), _q(_rand((H, W, 64), 3), dt)
    s = torch.einsum("hd,hnd->hn", q, k[:, :n_valid]) / 8.0
    ref = torch.einsum("hn,hnd->hd", torch.softmax(s, -1), v[:, :n_valid]).reshape(D)
    state = torch.tensor([pos, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    qd, kd, vd = q.reshape(D).to(dev, dt), k.to(dev, dt), v.to(dev, dt)
    part = torch.zeros(H, NS, L.ATTN_PART, device=dev)
    a = L.AttnDecodeArgs(qbuf=qd.data_ptr(), kcache=kd.data_ptr(), vcache=vd.data_ptr(), part=part.data_ptr(), state=state.data_ptr(),
                         n_heads=H, w_alloc=W, window=window, nsplit=NS, scale=0.125)
    ops.ar_attn_decode(dt, a)
    # combine through the wo-GEMV prologue with W = identity
    eye = torch.eye(D).to(dev, dt)
    xr = torch.zeros(D, device=dev)
    g = L.GemvArgs(W=eye.data_ptr(), ldw=D, N=D, K=D, part=part.data_ptr(), nsplit=NS, n_heads=H, xres=xr.data_ptr(), state=state.data_ptr())
    ops.ar_gemv(dt, L.PRO_ATTN, L.GEPI_RESIDUAL, g)
    torch.cuda.synchronize()
    r = _rel(xr.cpu(), ref)
    assert r < max(TOL[dt], 5e-3 if dt != torch.float32 else 1e-5), f"decode attention rel err {r}"


@pytest.mark.parametrize("dt", DTS)
def test_ar_qkv_rope_batch(dev, dt):
    """Batched decode QKV projection with fused RoPE + cache write (16-bit: skinny-GEMM epilogue; fp32: GEMM +
    rope kernel) vs torch: every sequence rotates at ITS position and writes ITS cache slot (pos % window,
    wrapped for one of them); a finished sequence writes nothing."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import rope_table
    B, H, K, W, window = 5, 24, 1536, 128, 100
    D = H * 64
    nL = 2                                          # cache layout [B][layers][H][W][64]; layer 1 is the one written
    xn = _q(_rand((B, K), 1), dt)
    w = _q(_rand((3 * D, K), 2, 4.0 / math.sqrt(K)), dt)
    pos = [0, 37, 99, 250, 64]
    done = [0, 0, 0, 0, 1]
    rope = rope_table(64, 512)
    qkv = _q(xn @ w.T, dt).view(B, 3, H, 32, 2)
    cs, sn = rope[pos, :, 0], rope[pos, :, 1]        # (B, 32)
    def rot(t):
        a, b = t[..., 0], t[..., 1]
        return torch.stack([a * cs[:, None, :] - b * sn[:, None, :], a * sn[:, None, :] + b * cs[:, None, :]], -1).reshape(B, H, 64)
    q_ref, k_ref, v_ref = rot(qkv[:, 0]), rot(qkv[:, 1]), qkv[:, 2].reshape(B, H, 64)
    state = torch.zeros(B, L.ST_WORDS, dtype=torch.int32)
    state[:, L.ST_POS] = torch.tensor(pos, dtype=torch.int32)
    state[:, L.ST_DONE] = torch.tensor(done, dtype=torch.int32)
    state = state.to(dev)
    qbuf = torch.full((B, D), 7.0, device=dev, dtype=dt)
    kc = torch.full((B, nL, H, W, 64), 7.0, device=dev, dtype=dt)
    vc = torch.full((B, nL, H, W, 64), 7.0, device=dev, dtype=dt)
    tmp = torch.zeros(B, 3 * D, device=dev, dtype=dt)
    ops.ar_qkv_rope_batch(xn.to(dev, dt), w.to(dev, dt), H, rope.to(dev), state, qbuf, kc[0, 1], vc[0, 1], nL * H * W * 64, W * 64, window, tmp)
    torch.cuda.synchronize()
    tol = max(TOL[dt], 1e-2 if dt != torch.float32 else 0)
    scale = float(q_ref.abs().max())
    for b in range(B):
        slot = pos[b] % window
        if done[b]:
            assert float((qbuf[b].float() - 7.0).abs().max()) == 0.0 and float((kc[b].float() - 7.0).abs().max()) == 0.0
            continue
        assert float((qbuf[b].float().cpu().view(H, 64) - q_ref[b]).abs().max()) / scale < tol
        assert float((kc[b, 1, :, slot].float().cpu() - k_ref[b]).abs().max()) / scale < tol
        assert float((vc[b, 1, :, slot].float().cpu() - v_ref[b]).abs().max()) / scale < tol
        untouched = torch.ones(W, dtype=torch.bool)
        untouched[slot] = False
        assert float((kc[b, 1][:, untouched].float() - 7.0).abs().max()) == 0.0 and float((kc[b, 0].float() - 7.0).abs().max()) == 0.0
        assert float((vc[b, 1][:, untouched].float() - 7.0).abs().max()) == 0.0


def test_ar_sampler_golden_cases(dev, gold_dir):
    """Device sampler chain vs the reference function chain (fixtures from the reference)."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import eos_penalty_table
    fx = np.load(os.path.join(gold_dir, "sampler_cases.npz"), allow_pickle=True)
    n_text, eos = int(fx["n_text"]), int(fx["eos_idx"])
    V = fx["logits"].shape[1]
    D = 64
    embed = _rand((V, D), 1).to(dev)
    n_checked = 0
    for i in range(fx["logits"].shape[0]):
        c = json.loads(str(fx["cfg"][i]))
        prev = [int(t) for t in fx["prev"][i]]
        n_est = int(fx["n_est"][i])
        P = 5
        tokens = torch.zeros(P + len(prev) + 4, dtype=torch.int64)
        tokens[P:P + len(prev)] = torch.tensor(prev, dtype=torch.int64)
        tokens = tokens.to(dev)
        state = torch.tensor([P + len(prev), len(prev), 0, P + len(prev), -1, 0, 0, 0], dtype=torch.int32, device=dev)
        noise = torch.ones(len(prev) + 1, V)
        noise[len(prev)] = torch.from_numpy(fx["q"][i])
        noise = noise.to(dev)
        tab = eos_penalty_table(n_est, c["dec"], c["fac"]).to(dev)
        logits = torch.from_numpy(fx["logits"][i]).to(dev)
        xres = torch.zeros(D, device=dev)
        a = L.SampleArgs(logits=logits.data_ptr(), V=V, state=state.data_ptr(), tokens=tokens.data_ptr(), max_len=10 ** 6,
                         alpha_frequency=c["af"], alpha_presence=c["ap"], penalty_window=c["win"], n_text=n_text, eos_idx=eos,
                         n_est=n_est, eos_table=tab.data_ptr(), temperature=c["temperature"], div_mode=0, top_k=c["topk"],
                         top_p=c["top_p"], typical_p=c["typical_p"], noise=noise.data_ptr(), noise_stride=V, embed=embed.data_ptr(), dim=D,
                         xres=xres.data_ptr())
        ops.ar_sample(a)
        torch.cuda.synchronize()
        st = state.cpu()
        tok = int(fx["tok"][i])
        assert int(st[L.ST_LAST]) == tok, f"case {i}: device token {int(st[L.ST_LAST])} != reference {tok}"
        if tok == eos:
            assert int(st[L.ST_DONE]) == 1
        else:
            assert int(st[L.ST_NGEN]) == len(prev) + 1 and int(tokens[P + len(prev)]) == tok
            assert torch.equal(xres.cpu(), embed[tok].cpu())
        n_checked += 1
    assert n_checked >= 12          # incl. the typical_p = 0.6 cases


def test_ar_sampler_random_vs_oracle(dev):
    """More sampler cases than the fixtures hold: random logits vs the CPU oracle, incl. V = 4096."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import eos_penalty_table
    g = torch.Generator().manual_seed(5)
    for V, n_text in [(4096, 3071), (1376, 288), (3000, 500)]:
        eos = V - 1
        embed = torch.zeros(V, 64, device=dev)
        for trial in range(9):
            cfg = [dict(t=0.7, k=100, p=0.2), dict(t=1.0, k=0, p=1.0), dict(t=0.7, k=100, p=1.0), dict(t=0.9, k=5, p=0.5),
                   dict(t=0.7, k=1, p=0.2), dict(t=1.3, k=4000, p=0.97),
                   # typical-p (samplers.py:96-122): alone, after top-k, after top-k + top-p
                   dict(t=1.0, k=0, p=1.0, ty=0.6), dict(t=0.8, k=200, p=1.0, ty=0.3), dict(t=0.9, k=100, p=0.9, ty=0.9)][trial]
            ty = cfg.get("ty", 1.0)
            logits = torch.randn(V, generator=g) * 2.5
            n_prev = [0, 3, 90, 120, 2, 40, 10, 0, 77][trial]
            prev = torch.randint(n_text - 1, V - 1, (n_prev,), generator=g).tolist()
            q = torch.empty(V).exponential_(1, generator=g)
            p = O.ARSamplingParams(cfg["t"], cfg["k"], cfg["p"], ty, 3.0, 0.4, 100, 0.5, 1.0, 30)
            z = O.filter_logits(logits, prev, p, n_text, eos)
            tok = O.draw_token(z, q)
            P = 3
            tokens = torch.zeros(P + n_prev + 2, dtype=torch.int64)
            tokens[P:P + n_prev] = torch.tensor(prev, dtype=torch.int64)
            tokens = tokens.to(dev)
            state = torch.tensor([P + n_prev, n_prev, 0, P + n_prev, -1, 0, 0, 0], dtype=torch.int32, device=dev)
            noise = torch.ones(n_prev + 1, V)
            noise[n_prev] = q
            noise = noise.to(dev)
            tab = eos_penalty_table(30, 0.5, 1.0).to(dev)
            ld = logits.to(dev)
            xres = torch.zeros(64, device=dev)
            a = L.SampleArgs(logits=ld.data_ptr(), V=V, state=state.data_ptr(), tokens=tokens.data_ptr(), max_len=10 ** 6,
                             alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100, n_text=n_text, eos_idx=eos, n_est=30,
                             eos_table=tab.data_ptr(), temperature=cfg["t"], div_mode=0, top_k=cfg["k"], top_p=cfg["p"], typical_p=ty,
                             noise=noise.data_ptr(), noise_stride=V, embed=embed.data_ptr(), dim=64, xres=xres.data_ptr())
            ops.ar_sample(a)
            torch.cuda.synchronize()
            assert int(state[L.ST_LAST]) == tok, f"V={V} trial {trial}: {int(state[L.ST_LAST])} != {tok}"


def test_ar_sampler_topk_ties_and_masked_threshold(dev):
    """top-k radix-select path: ties at the threshold (all kept, torch `logits < kth` semantics), tie sets
    larger than the select buffer, and a threshold that falls into the masked (-inf) region."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import eos_penalty_table
    g = torch.Generator().manual_seed(17)
    cases = [  # (V, n_text, quantum, top_k, top_p)
        (4096, 3071, 0.5, 100, 1.0), (4096, 3071, 0.5, 100, 0.6), (4096, 3071, 2.0, 50, 1.0), (4096, 300, 4.0, 20, 0.9),
        (4096, 4000, 0.0, 200, 1.0), (1376, 1300, 0.25, 100, 0.8), (3000, 500, 1.0, 256, 1.0), (3000, 500, 8.0, 3, 1.0)]
    for ci, (V, n_text, quantum, k, top_p) in enumerate(cases):
        eos = V - 1
        embed = torch.zeros(V, 64, device=dev)
        logits = torch.randn(V, generator=g) * 2.5
        if quantum > 0:
            logits = torch.round(logits / quantum) * quantum
        q = torch.empty(V).exponential_(1, generator=g)
        p = O.ARSamplingParams(1.0, k, top_p, 1.0, 0.0, 0.0, 100, 0.5, 1.0, 30)
        z = O.filter_logits(logits, [], p, n_text, eos)
        tok = O.draw_token(z, q)
        tokens = torch.zeros(8, dtype=torch.int64, device=dev)
        state = torch.tensor([3, 0, 0, 3, -1, 0, 0, 0], dtype=torch.int32, device=dev)
        noise = q.reshape(1, V).to(dev)
        tab = eos_penalty_table(30, 0.5, 1.0).to(dev)
        ld = logits.to(dev)
        xres = torch.zeros(64, device=dev)
        a = L.SampleArgs(logits=ld.data_ptr(), V=V, state=state.data_ptr(), tokens=tokens.data_ptr(), max_len=10 ** 6,
                         alpha_frequency=0.0, alpha_presence=0.0, penalty_window=100, n_text=n_text, eos_idx=eos, n_est=30,
                         eos_table=tab.data_ptr(), temperature=1.0, div_mode=0, top_k=k, top_p=top_p, typical_p=1.0,
                         noise=noise.data_ptr(), noise_stride=V, embed=embed.data_ptr(), dim=64, xres=xres.data_ptr())
        ops.ar_sample(a)
        torch.cuda.synchronize()
        assert int(state[L.ST_LAST]) == tok, f"case {ci}: {int(state[L.ST_LAST])} != {tok}"


def test_nar_sample_vs_oracle(dev):
    """Fused posterior/sample kernel vs the oracle's reverse_step on identical logits and uniforms: integer
    outputs, EQUAL -- a differing id is accepted only where the oracle's own scores of the two classes are
    within float noise of each other (tests/parity_util.py), i.e. a tie either side may break."""
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import log_eps, nar_step_consts
    S, Q, K, off = 70, 8, 1025, 20
    g = torch.Generator().manual_seed(11)
    tb = O.diffusion_tables(K, 200)
    times = [199, 150, 21, 20, 1, 0]
    consts = nar_step_consts(times, K).to(dev)
    total_bad = 0
    for si, t in enumerate(times):
        lc = torch.randn(S - off, Q - 1, K, generator=g) * 2
        lu = torch.randn(S - off, Q - 1, K, generator=g) * 2
        x_t = torch.randint(0, K, (S, Q), generator=g)
        x_known = torch.randint(0, 1024, (S, Q), generator=g)
        m = torch.zeros(S, Q, dtype=torch.bool)
        m[:, 0] = True
        m[:off] = True
        u1, u2 = torch.rand(S, Q, K, generator=g), torch.rand(S, Q, K, generator=g)
        # oracle wants full-shape logits; rows it never uses (m = 1) are filled with zeros
        fc = torch.zeros(S, Q, K)
        fu = torch.zeros(S, Q, K)
        fc[off:, 1:], fu[off:, 1:] = lc, lu
        ref, s_unk, s_kn = O.reverse_step(tb, fc, fu, x_t, x_known, m, t, u1, u2 if t > 0 else None, 3.0, 0.7, return_scores=True)
        if 20 < t:
            ref[:, 0] = x_known[:, 0]
        Kp = 1028
        lgc = torch.zeros(S - off, Q - 1, Kp)
        lgu = torch.zeros(S - off, Q - 1, Kp)
        lgc[..., :K], lgu[..., :K] = lc, lu
        xd = x_t.clone().to(dev)
        keep = [lgc.to(dev), lgu.to(dev), x_known.to(dev), m.to(torch.uint8).to(dev), u1.to(dev), u2.to(dev)]
        step = torch.tensor([si], dtype=torch.int32, device=dev)
        a = L.NarSampleArgs(logits_c=keep[0].data_ptr(), logits_u=keep[1].data_ptr(), ld_row=(Q - 1) * Kp, ld_q=Kp, S=S, n_q=Q, K=K,
                            row_offset=off, x=xd.data_ptr(), x_known=keep[2].data_ptr(), m=keep[3].data_ptr(), u1=keep[4].data_ptr(),
                            u2=keep[5].data_ptr(), consts=consts.data_ptr(), step=step.data_ptr(), guidance_w=3.0, temperature=0.7,
                            log_eps=log_eps(), div_mode=0, q0_override_steps=20)
        ops.nar_sample(a)
        torch.cuda.synchronize()
        from parity_util import ungated_mismatches
        n_mis, bad = ungated_mismatches(xd.cpu(), ref, s_unk, s_kn, m)
        total_bad += n_mis
        assert not bad, f"t={t}: ids differ from the oracle away from any tie: {bad[:5]}"
    print(f"nar_sample: {total_bad} tie-excused mismatching ids over {len(times) * S * Q}")


def test_nar_sample_known_rows_crafted_uniforms(dev):
    """The known-row branch (q_sample) evaluates the Gumbel score only for each lane's largest uniform and for the hit class
    (csrc/nar_sample.hip).  Crafted draws exercise what random ones never do: the largest uniform of a row duplicated at
    several indices (same lane: k and k + 64; other lanes; at the hit class), rows whose largest uniform is below the 0.9 guard
    (the plain all-classes loop), and the top float below 1.  Every id must equal a plain fp32 evaluation of all 1025 classes
    with the first index winning ties (diffuser.py:219-236)."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.tables import log_eps, nar_step_consts
    S, Q, K = 48, 8, 1025
    g = torch.Generator().manual_seed(23)
    times = [150, 1]
    consts = nar_step_consts(times, K).to(dev)
    x_known = torch.randint(0, K, (S, Q), generator=g)
    u = torch.rand(S, Q, K, generator=g)
    rows = u.view(S * Q, K)
    xk = x_known.view(-1)
    top = torch.nextafter(torch.tensor(1.0), torch.tensor(0.0)).item()
    for r in range(S * Q):
        mode = r % 8
        k0 = int(torch.randint(0, K - 200, (1,), generator=g))
        if mode == 0:
            rows[r] *= 0.85                                   # below the guard: the all-classes loop
        elif mode == 1:
            rows[r, k0] = rows[r, k0 + 64] = 0.99993          # the row maximum twice in ONE lane
        elif mode == 2:
            rows[r, k0 + 5] = rows[r, k0 + 70] = rows[r, k0 + 133] = 0.99991    # and across lanes
        elif mode == 3:
            rows[r, int(xk[r])] = 0.99995                     # the hit class holds the maximum
            rows[r, (int(xk[r]) + 64) % K] = 0.99995          # ... tied with a miss class of its lane
        elif mode == 4:
            rows[r, k0] = top                                 # the largest float below 1 (the clamp of -log u)
            rows[r, k0 + 1] = top
        elif mode == 5:
            rows[r, 0] = 0.0                                  # torch's 1.0 -> 0.0 reversal value at index 0
    m = torch.ones(S, Q, dtype=torch.uint8)
    lg = torch.zeros(1, Q - 1, 1028)
    for si, t in enumerate(times):
        c = consts[si].cpu()
        c4, c5 = float(c[4]), float(c[5])
        f32 = lambda v: torch.tensor(v, dtype=torch.float32)       # noqa: E731
        lae = lambda a, b: torch.maximum(a, b) + torch.log(torch.exp(a - torch.maximum(a, b)) + torch.exp(b - torch.maximum(a, b)))   # noqa: E731
        q_hit, q_miss = lae(f32(0.0) + f32(c4), f32(c5)), lae(f32(log_eps()) + f32(c4), f32(c5))
        gum = -torch.log(torch.clamp(-torch.log(torch.clamp(rows, min=1e-7)), min=1e-7))
        score = gum + q_miss
        score[torch.arange(S * Q), xk] = gum[torch.arange(S * Q), xk] + q_hit
        ref = score.argmax(dim=1)                                   # torch.argmax: first index of the maximum
        best = score.max(dim=1).values
        first = (score == best[:, None]).float().argmax(dim=1)
        assert torch.equal(ref, first)
        xd = torch.zeros(S, Q, dtype=torch.int64, device=dev)
        keep = [lg.to(dev), x_known.to(dev), m.to(dev), u.to(dev)]
        step = torch.tensor([si], dtype=torch.int32, device=dev)
        a = L.NarSampleArgs(logits_c=keep[0].data_ptr(), logits_u=keep[0].data_ptr(), ld_row=(Q - 1) * 1028, ld_q=1028, S=S, n_q=Q, K=K,
                            row_offset=S, x=xd.data_ptr(), x_known=keep[1].data_ptr(), m=keep[2].data_ptr(), u1=keep[3].data_ptr(),
                            u2=keep[3].data_ptr(), consts=consts.data_ptr(), step=step.data_ptr(), guidance_w=3.0, temperature=0.7,
                            log_eps=log_eps(), div_mode=0, q0_override_steps=1000)     # (no L0 override: every row reports its own draw)
        ops.nar_sample(a)
        torch.cuda.synchronize()
        got = xd.cpu().view(-1)
        diff = (got != ref).nonzero().view(-1)
        # the CPU libm and the device libm may round log differently: a differing id is accepted only between classes whose
        # reference scores are within 2 ulp of each other
        for r in diff.tolist():
            a_, b_ = float(score[r, got[r]]), float(score[r, ref[r]])
            assert abs(a_ - b_) <= 2 * abs(b_) * 2 ** -23, (t, r, r % 8, int(got[r]), int(ref[r]), a_, b_)
        print(f"known rows, t={t}: {len(diff)} ids differ within 2 ulp of a tie, of {S * Q}")


def test_nar_uniforms_equal_torch_rand(dev):
    """m5_nar_uniforms reproduces ``torch.rand`` on this device bit for bit (Philox4x32-10 with torch's launch geometry and
    rocrand's uint -> float map): whole draws at the NAR bench shape (11 M values: several grid-stride iterations) and at small
    sizes (one partial iteration), consecutive draws selected by the device step counter, and the merged form -- first draw on the
    rows sampled from the model, second draw on the known rows, one draw only at t = 0 -- that m5_nar_sample consumes; the
    generator advance per draw is the one the engine assumes."""
    from mars5_tts_amd import _lib as L, ops
    from mars5_tts_amd.nar_engine import _magic_div
    prop = torch.cuda.get_device_properties(dev)
    per_mp = prop.max_threads_per_multi_processor // 256
    for S, Q, K, seed, off0 in [(1349, 8, 1025, 1234, 0), (37, 8, 1025, 2 ** 63 + 11, 4096), (3, 2, 7, 5, 8), (450, 8, 1025, 99, 2 ** 33)]:
        n = S * Q * K
        G = 256 * min(prop.multi_processor_count * per_mp, (n + 255) // 256)
        inc = ((n - 1) // (4 * G) + 1) * 4
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        g.set_offset(off0)
        draws = []
        for _ in range(5):
            o = g.get_offset()
            draws.append(torch.rand((1, S, Q, K), generator=g, device=dev)[0])
            assert g.get_offset() - o == inc, (n, g.get_offset() - o, inc)
        wrap = lambda v: v - (1 << 64) if v >= (1 << 63) else v                 # noqa: E731
        rng = torch.tensor([wrap(seed), off0], dtype=torch.int64, device=dev)
        out = torch.full((S, Q, K), -1.0, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        km, ks = _magic_div(K, n)
        # plain form: the first draw of step i = draw 2 i of the generator
        for i in (0, 1, 2):
            step.fill_(i)
            a = L.NarUniformArgs(out=out.data_ptr(), n=n, K=K, k_magic=km, k_shift=ks, m=None, rng=rng.data_ptr(), inc=inc, grid_threads=G,
                                 step=step.data_ptr(), consts=None)
            ops.nar_uniforms(a)
            torch.cuda.synchronize()
            assert torch.equal(out, draws[2 * i]), f"n={n} step {i}: {int((out != draws[2 * i]).sum())} of {n} values differ from torch.rand"
        assert float(out.min()) >= 0.0 and float(out.max()) < 1.0
        # merged form
        gm = torch.Generator().manual_seed(seed % 1000)
        m = (torch.rand(S, Q, generator=gm) < 0.4).to(torch.uint8).to(dev)
        consts = torch.zeros(3, L.NAR_CONSTS, device=dev)
        consts[:, 6] = torch.tensor([199.0, 1.0, 0.0])
        for i in (0, 1, 2):
            step.fill_(i)
            out.fill_(-1.0)
            for magic in ((km, ks), (0, 0)):                                   # multiply-shift row index and the plain division
                a = L.NarUniformArgs(out=out.data_ptr(), n=n, K=K, k_magic=magic[0], k_shift=magic[1], m=m.data_ptr(), rng=rng.data_ptr(), inc=inc,
                                     grid_threads=G, step=step.data_ptr(), consts=consts.data_ptr())
                ops.nar_uniforms(a)
                torch.cuda.synchronize()
                want = torch.where(m[:, :, None].bool(), draws[2 * i + 1], draws[2 * i]) if i < 2 else draws[2 * i]
                assert torch.equal(out, want), f"merged form, n={n} step {i}: {int((out != want).sum())} values differ"
    # the exponential transform of the same draws (the AR sampler's Exp(1) noise: torch.multinomial's exponential_) -- incl. torch's
    # guard against log(1) -- over 4 M values and at the sampler's size (one value per Philox call: V <= launch width)
    for n_e, seed in [(4096, 321), (1 << 22, 7), (1025 * 4099, 2 ** 40 + 3)]:
        G = 256 * min(prop.multi_processor_count * per_mp, (n_e + 255) // 256)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        rows = [torch.empty(n_e, device=dev).exponential_(1, generator=g) for _ in range(3)]
        inc = ((n_e - 1) // (4 * G) + 1) * 4
        assert g.get_offset() == 3 * inc
        rng = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
        oute = torch.empty(n_e, device=dev)
        for i in range(3):
            rng[1] = i * inc
            a = L.NarUniformArgs(out=oute.data_ptr(), n=n_e, K=1, k_magic=0, k_shift=0, m=None, rng=rng.data_ptr(), inc=inc, grid_threads=G,
                                 step=None, consts=None, transform=1)
            ops.nar_uniforms(a)
            torch.cuda.synchronize()
            assert torch.equal(oute, rows[i]), f"exponential_, n={n_e} draw {i}: {int((oute != rows[i]).sum())} values differ (max rel {float(((oute - rows[i]).abs() / rows[i]).max()):.3g})"
        assert float(oute.min()) > 0.0
    # bad arguments answer with a status code
    a = L.NarUniformArgs(out=out.data_ptr(), n=n, K=K, k_magic=km // 2, k_shift=ks, m=m.data_ptr(), rng=rng.data_ptr(), inc=inc, grid_threads=G,
                         step=step.data_ptr(), consts=consts.data_ptr())
    assert L.lib.m5_nar_uniforms(a, None) == L.M5_ERR_ARG          # a multiply-shift that does not divide exactly is refused


def test_graph_capture_replay(dev):
    from mars5_tts_amd import ops
    st = torch.cuda.Stream(device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ops.Graph.begin(st.cuda_stream)
    ops.add_int(cnt, 1, stream=st.cuda_stream)
    ops.add_int(cnt, 2, stream=st.cuda_stream)
    gr = ops.Graph().end(st.cuda_stream)
    for _ in range(5):
        gr.launch(st.cuda_stream)
    st.synchronize()
    assert int(cnt.item()) == 15
    e0, e1 = ops.Event(), ops.Event()
    e0.record(st.cuda_stream)
    gr.launch(st.cuda_stream)
    e1.record(st.cuda_stream)
    assert e0.elapsed_ms(e1) >= 0.0


def test_expand_tokens_vs_host_table(dev, tiny_bundle):
    """m5_expand_tokens (AR -> NAR hand-off on device) against the host expansion of the same BPE tokenizer -- which
    tests/test_oracle_golden.py pins to the reference's ``decode_int`` (tokenizer_cases.npz): merged tokens, plain codes,
    special tokens (empty runs), text ids (clamped to code 0 like the reference's ``.clamp(min=0)``), more than one
    1024-token chunk, and the empty sequence."""
    import io
    from mars5_tts_amd import minbpe, ops
    st = minbpe.CodebookTokenizer()
    st.load(io.BytesIO(tiny_bundle.ar_ckpt["vocab"]["speechtok.model"].encode()))
    table = st.expansion_table()
    off, vals, mx = st.expansion_csr()
    assert mx >= 2, "the tiny speech tokenizer must contain merges"
    n_text = tiny_bundle.n_text
    g = torch.Generator().manual_seed(4)
    for n in (0, 1, 37, 1024, 2500):
        sp = torch.randint(0, len(table), (n,), generator=g)
        toks = sp + n_text
        if n > 5:
            toks[3] = 5                                   # a text id: clamps to speech id 0
            toks[4] = n_text + st.special_tokens["<|endofspeech|>"]
        want = [c for t in (toks - n_text).clamp(min=0).tolist() for c in table[t]]
        got = ops.expand_tokens(toks.to(dev), n_text, off.to(dev), vals.to(dev), mx)
        assert got.cpu().tolist() == want, f"n={n}"


def test_device_trim_matches_reference_fixture(dev, gold_dir):
    """m5_trim_bounds (the silence trim of ``tts()`` on the vocoder's device output) against what the REFERENCE's
    ``mars5/trim.py:110-178`` returned for the same deterministic waveforms (tests/golden/trim_cases.npz): the same
    [start, end] interval and samples for mono / stereo / all-zero / very short inputs at three thresholds."""
    import mars5_oracle as O
    from mars5_tts_amd.trim import trim_device
    fx = np.load(os.path.join(gold_dir, "trim_cases.npz"))
    waves = O.trim_test_waves()
    for i, top_db in fx["cases"].tolist():
        ref_idx = fx[f"idx_{i}_{top_db}"].tolist()
        w = waves[i].clone().to(torch.float32)
        if w.shape[-1] <= 1024:            # shorter than the reflect padding: the device path leaves it to the host trim
            continue
        y, idx = trim_device(w.to(dev), top_db=top_db)
        assert idx.tolist() == ref_idx, f"wave {i} top_db {top_db}: {idx.tolist()} vs reference {ref_idx}"
        assert torch.equal(y.cpu(), w[..., ref_idx[0]:ref_idx[1]])
