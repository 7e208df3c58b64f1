#!/usr/bin/env python
"""GPU microbench + correctness probe for m5_gemm on the NAR / AR-prefill shapes.
Each case: compare with a torch fp32 matmul of the same 16-bit operands, then time a hipGraph
of REP back-to-back launches with HIP events on the launch stream (host overhead amortised).
Run with M5_GEMM_V1=1 to time the first-generation kernel for an A/B."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
REP = int(os.environ.get("REP", "20"))

def ref_epi(a, w, bias, epi, c0):
    y = a.float() @ w.float().T
    if bias is not None:
        y = y + bias
    if epi == L.EPI_RESIDUAL:
        return c0 + y
    if epi == L.EPI_SWIGLU:
        yy = y.to(dt).float()
        return (torch.nn.functional.silu(yy[:, 0::2]).to(dt).float() * yy[:, 1::2])
    if epi == L.EPI_SILU_DT:
        return torch.nn.functional.silu(y)
    return y

def run_case(name, M, N, K, epi, bias=True, rpb=None, check=True, pad=0):
    g = torch.Generator(device="cpu").manual_seed(1)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev, dt)
    w = (torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)).to(dev, dt)
    if pad:
        ap = torch.zeros(M, K + pad, dtype=dt, device=dev); ap[:, :K] = a; a = ap[:, :K]
        wp = torch.zeros(N, K + pad, dtype=dt, device=dev); wp[:, :K] = w; w = wp[:, :K]
    b = torch.randn(N, generator=g).to(dev) if bias else None
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    sc = None
    if epi == L.EPI_QKV:
        H = 16 if K == 1024 else K // 64
        nsec = N // (H * 64)
        rpb = rpb or M
        B = (M + rpb - 1) // rpb
        Sp = (rpb + 63) // 64 * 64
        q = torch.zeros(B, H, rpb, 64, dtype=dt, device=dev)
        k = torch.zeros(B, H, rpb, 64, dtype=dt, device=dev)
        vt = torch.zeros(B, H, 64, Sp, dtype=dt, device=dev)
        sc = L.QkvScatter(q=q.data_ptr(), k=k.data_ptr() if nsec > 1 else None, vt=vt.data_ptr() if nsec > 2 else None,
                          rows_per_batch=rpb, n_heads=H, head_dim=64, q_bs=H * rpb * 64, q_hs=rpb * 64, q_rs=64,
                          k_bs=H * rpb * 64, k_hs=rpb * 64, k_rs=64, vt_bs=H * 64 * Sp, vt_hs=64 * Sp, vt_ds=Sp)
        out = None
    elif epi in (L.EPI_F32, L.EPI_RESIDUAL):
        out = torch.randn(M, N, generator=g).to(dev)
    elif epi == L.EPI_SWIGLU:
        out = torch.zeros(M, N // 2, dtype=dt, device=dev)
    else:
        out = torch.zeros(M, N, dtype=dt, device=dev)
    c0 = out.clone().float() if out is not None else None
    call = lambda: ops.gemm(a, w, out, epi, bias=b, scatter=sc, stream=st)          # noqa: E731
    if os.environ.get("DLN") == "1" and epi == L.EPI_RESIDUAL and N % 128 == 0 and N // 128 <= 8:
        # the deferred-LayerNorm producer form of the residual GEMM (also writes the centred 16-bit copy + row partials)
        xt = torch.zeros(M, N, dtype=dt, device=dev)
        part = torch.zeros(M, N // 128, 2, device=dev)
        cen, cen2 = torch.zeros(M, device=dev), torch.zeros(M, device=dev)
        dl = L.DeferredLN(mode=1, np=N // 128, xt=xt.data_ptr(), ld_xt=N, part=part.data_ptr(), cen_in=cen.data_ptr(), cen_out=cen2.data_ptr(),
                          delta=None, s=None, s_bs=0, eps=0.0, n_feat=N, rows_bs=M)
        call = lambda: ops.gemm_dln(a, w, out, epi, dl, bias=b, stream=st)         # noqa: E731
        name = name + " +dln-producer"
    with torch.cuda.stream(stream):
        call()
    stream.synchronize()
    err = None
    if check:
        if epi == L.EPI_QKV:
            y = ref_epi(a, w, b, epi, None)                      # (M, N)
            D = sc.n_heads * 64
            rows = torch.arange(M, device=dev)
            bb, ss = rows // rpb, rows % rpb
            errs = []
            qq = q[bb, :, ss, :].reshape(M, D).float()
            errs.append((qq - y[:, :D]).abs().max().item())
            if nsec > 1:
                kk = k[bb, :, ss, :].reshape(M, D).float()
                errs.append((kk - y[:, D:2 * D]).abs().max().item())
            if nsec > 2:
                vv = vt[bb, :, :, ss].reshape(M, D).float()
                errs.append((vv - y[:, 2 * D:]).abs().max().item())
            err = max(errs)
        else:
            y = ref_epi(a, w, b, epi, c0)
            err = (out.float() - y).abs().max().item()
    # timing
    with torch.cuda.stream(stream):
        stream.synchronize()
        ops.Graph.begin(st)
        for _ in range(REP):
            call()
        gr = ops.Graph().end(st)
        gr.launch(st)
        stream.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        for _ in range(5):
            gr.launch(st)
        e1.record(st)
        stream.synchronize()
    us = e0.elapsed_ms(e1) * 1e3 / (5 * REP)
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{name:40s} M={M:6d} N={N:5d} K={K:5d} epi={epi}  {us:8.2f} us  {tf:7.1f} TF  maxerr={err}", flush=True)
    return dict(name=name, M=M, N=N, K=K, epi=epi, us=us, tflops=tf, err=err)

CASES = [
    ("nar self qkv", 2816, 3072, 1024, L.EPI_QKV, True, 1408),
    ("nar l0 qkv", 1408, 3072, 1024, L.EPI_QKV, True, 1408),        # layer 0's shared self-attention block: one branch
    ("nar out_proj", 2816, 1024, 1024, L.EPI_RESIDUAL, True, None),
    ("nar cross q", 2816, 1024, 1024, L.EPI_QKV, True, 1408),
    ("nar p.b", 2816, 1024, 768, L.EPI_RESIDUAL, True, None),            # absorbed cross-attention: P . B with the residual epilogue
    ("nar swiglu", 2816, 6144, 1024, L.EPI_SWIGLU, False, None),
    ("nar linear2", 2816, 1024, 3072, L.EPI_RESIDUAL, True, None),
    ("nar head (1 of 7)", 1798, 1025, 1024, L.EPI_F32, True, None),
    ("nar heads folded", 1798, 7196, 1024, L.EPI_F32, True, None),     # round 4: the seven heads as ONE GEMM (rows padded 1025 -> 1028)
    ("enc qkv (hoisted)", 15600, 3072, 1024, L.EPI_QKV, True, 39),
    ("enc swiglu (hoisted)", 15600, 6144, 1024, L.EPI_SWIGLU, False, None),
    ("ar prefill qkv", 489, 4608, 1536, L.EPI_DT, False, None),
    ("ar prefill w13", 489, 7168, 1536, L.EPI_SWIGLU, False, None),
    ("ar prefill w2", 489, 1536, 3584, L.EPI_RESIDUAL, False, None),
    ("spk enc qkv", 451, 3072, 1024, L.EPI_QKV, True, 451),
    # batched NAR group (c3: 8 utterances x 2 branches x 2240 rows)
    ("big out_proj", 35840, 1024, 1024, L.EPI_RESIDUAL, True, None),
    ("big linear2", 35840, 1024, 3072, L.EPI_RESIDUAL, True, None),
    ("big swiglu", 35840, 6144, 1024, L.EPI_SWIGLU, False, None),
    ("big qkv", 35840, 3072, 1024, L.EPI_QKV, True, 2240),
    # one NAR group of 32 utterances (round 4: ~98 k real rows of 147 k padded ones are launched)
    ("huge out_proj", 98304, 1024, 1024, L.EPI_RESIDUAL, True, None),
    ("huge linear2", 98304, 1024, 3072, L.EPI_RESIDUAL, True, None),
    ("huge swiglu", 98304, 6144, 1024, L.EPI_SWIGLU, False, None),
    ("huge qkv", 98304, 3072, 1024, L.EPI_QKV, True, 2304),
]

def run_heads(name="nar heads x7"):
    """the 7-head logits GEMM of one reverse step exactly as nar_engine enqueues it (batch 7, N = 1025, strided C)"""
    nb_so, D, K1, Q1 = 1798, 1024, 1025, 7
    Kp = 1028
    g = torch.Generator(device="cpu").manual_seed(1)
    hn = (torch.randn(Q1, nb_so, D, generator=g) * 0.5).to(dev, dt)
    w = (torch.randn(Q1, K1, D, generator=g) / 32).to(dev, dt)
    b = torch.randn(Q1, K1, generator=g).to(dev)
    logits = torch.zeros(nb_so, Q1, Kp, device=dev)
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    call = lambda: ops.gemm(hn[0], w[0], logits, L.EPI_F32, bias=b, ldc=Q1 * Kp, batch=Q1, sA=nb_so * D, sW=K1 * D, sC=Kp, sBias=K1, stream=st)
    with torch.cuda.stream(stream):
        call()
        stream.synchronize()
        ref = torch.einsum("qmk,qnk->mqn", hn.float(), w.float()) + b[None]
        err = float((logits[..., :K1] - ref).abs().max())
        ops.Graph.begin(st)
        for _ in range(REP):
            call()
        gr = ops.Graph().end(st)
        gr.launch(st)
        stream.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        for _ in range(5):
            gr.launch(st)
        e1.record(st)
        stream.synchronize()
    us = e0.elapsed_ms(e1) * 1e3 / (5 * REP)
    print(f"{name:40s} M={nb_so * Q1:6d} N={K1:5d} K={D:5d} epi=0  {us:8.2f} us  {2.0 * nb_so * Q1 * K1 * D / us / 1e6:7.1f} TF  maxerr={err}", flush=True)


if __name__ == "__main__":
    if os.environ.get("HEADS") == "1":
        run_heads()
        sys.exit(0)
    res = []
    cfgs = [int(c) for c in os.environ.get("SWEEP", "").split(",") if c != ""]
    if os.environ.get("M5_GEMM_V1") == "1" or not cfgs:
        cfgs = [None]
    pads = [int(c) for c in os.environ.get("PADS", "0").split(",")]
    krots = os.environ.get("KROTS", "0").split(",")
    only = [s for s in os.environ.get("ONLY", "").split(",") if s]
    for (name, M, N, K, epi, bias, rpb) in CASES:
        if only and name not in only:
            continue
        for cfg in cfgs:
            for pad in pads:
                for kr in krots:
                    if cfg is not None:
                        os.environ["M5_GEMM_CFG"] = str(cfg)
                    os.environ["M5_GEMM_KROT"] = kr
                    r = run_case(f"{name} cfg={cfg} pad={pad} kr={kr}", M, N, K, epi, bias=bias, rpb=rpb, pad=pad)
                    r["cfg"] = cfg; r["pad"] = pad; r["krot"] = kr
                    res.append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "v1" if os.environ.get("M5_GEMM_V1") == "1" else "v2"
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"gemm_bench_{tag}.json"), "w"), indent=1)
