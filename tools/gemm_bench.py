#!/usr/bin/env python
"""GPU microbench + correctness probe for m5_gemm on the NAR / AR-prefill shapes.
Each case: compare with a torch fp32 matmul of the same 16-bit operands, then time a hipGraph
of REP back-to-back launches with HIP events on the launch stream (host overhead amortised).
Run with M5_GEMM_V1=1 to time the first-generation kernel for an A/B."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
REP = 20

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

def run_case(name, M, N, K, epi, bias=True, rpb=None, check=True):
    g = torch.Generator(device="cpu").manual_seed(1)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev, dt)
    w = (torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)).to(dev, dt)
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
    with torch.cuda.stream(stream):
        ops.gemm(a, w, out, epi, bias=b, scatter=sc, stream=st)
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
            ops.gemm(a, w, out, epi, bias=b, scatter=sc, stream=st)
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
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d} epi={epi}  {us:8.2f} us  {tf:7.1f} TF  maxerr={err}", flush=True)
    return dict(name=name, M=M, N=N, K=K, epi=epi, us=us, tflops=tf, err=err)

if __name__ == "__main__":
    res = []
    for Mrows, rpb in ((2698, 1349), (2816, 1408)):
        res.append(run_case("nar self qkv", Mrows, 3072, 1024, L.EPI_QKV, rpb=rpb))
        res.append(run_case("nar out_proj", Mrows, 1024, 1024, L.EPI_RESIDUAL))
        res.append(run_case("nar cross q", Mrows, 1024, 1024, L.EPI_QKV, rpb=rpb))
        res.append(run_case("nar swiglu", Mrows, 6144, 1024, L.EPI_SWIGLU, bias=False))
        res.append(run_case("nar linear2", Mrows, 1024, 3072, L.EPI_RESIDUAL))
    res.append(run_case("nar head (1 of 7)", 1798, 1025, 1024, L.EPI_F32))
    res.append(run_case("enc qkv (hoisted)", 15600, 3072, 1024, L.EPI_QKV, rpb=39))
    res.append(run_case("ar prefill qkv", 489, 4608, 1536, L.EPI_DT, bias=False))
    res.append(run_case("ar prefill w13", 489, 7168, 1536, L.EPI_SWIGLU, bias=False))
    res.append(run_case("ar prefill w2", 489, 1536, 3584, L.EPI_RESIDUAL, bias=False))
    res.append(run_case("timestep mlp silu", 200, 1024, 1024, L.EPI_SILU_DT))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "v1" if os.environ.get("M5_GEMM_V1") == "1" else "v2"
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"gemm_bench_{tag}.json"), "w"), indent=1)
