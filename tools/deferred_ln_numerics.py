#!/usr/bin/env python
"""CPU study for the next structural lever of the NAR step: taking the 48 LayerNorm launches of a step off the
chain by DEFERRING the normalisation into the consuming GEMM's epilogue,

    LN(x) W^T + b  =  r_m * ( x (W diag(g))^T  -  mu_m * s_n )  +  (W beta + b),      s_n = sum_k (W diag(g))[n, k],

with the GEMM running on bf16(x) (a second output of the producing residual GEMM) against the folded weights, and the row
statistics mu, r = 1 / sqrt(var + eps) coming from per-tile partial sums of the producer.  The rounding points move from
bf16(LN(x)) to bf16(x) and bf16(W g), and the subtraction mu * s_n can cancel.  This script compares, against the fp32 product,
  (b) today's 16-bit path:   bf16(LN_fp32(x)) @ bf16(W)^T, fp32 accumulation
  (c) the deferred form:     bf16(x) @ bf16(W g)^T, fp32 accumulation, fp32 epilogue as above
on residual streams of increasing hostility (per-channel scale spread, a few massive channels, a row mean of several sigma).
No GPU needed.  usage: python tools/deferred_ln_numerics.py [rows]"""
import sys
import torch

torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D, N, EPS = 1024, 3072, 1e-5


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def run(name, x, W, g, beta, b):
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    r = torch.rsqrt(var + EPS)
    ln = (x - mu) * r * g + beta
    ref = ln.double() @ W.double().T + b.double()
    # (b) today's path
    yb = (bf(ln) @ bf(W).T) + b
    # (c) deferred
    Wg = bf(W * g)
    s = Wg.sum(-1)
    acc = bf(x) @ Wg.T
    yc = r * (acc - mu * s) + (W @ beta + b)
    # (c') deferred with the mean removed from the 16-bit copy by the producer (x - mu is what gets rounded): no cancellation term
    acc2 = bf(x - mu) @ Wg.T
    yc2 = r * acc2 + (W @ beta + b)
    scale = ref.pow(2).mean().sqrt()
    out = []
    for y in (yb, yc, yc2):
        e = (y.double() - ref)
        out.append((float(e.pow(2).mean().sqrt() / scale), float(e.abs().max() / scale)))
    print(f"{name:58s} |mu|/sigma {float((mu.abs() / var.sqrt()).mean()):6.2f}   today rms {out[0][0]:.2e} max {out[0][1]:.2e}   "
          f"deferred rms {out[1][0]:.2e} max {out[1][1]:.2e}   deferred, centred copy rms {out[2][0]:.2e} max {out[2][1]:.2e}")


def main():
    W = torch.randn(N, D) / D ** 0.5
    g = 1.0 + 0.1 * torch.randn(D)
    beta = 0.1 * torch.randn(D)
    b = 0.1 * torch.randn(N)
    base = torch.randn(M, D)
    run("unit gaussian rows", base, W, g, beta, b)
    chan = torch.exp(0.7 * torch.randn(D))
    run("per-channel scale spread (log-normal, sigma 0.7)", base * chan, W, g, beta, b)
    x = base * chan
    x[:, [3, 400, 777]] *= 40.0
    run("+ three massive channels (x40)", x, W, g, beta, b)
    run("row mean of 3 sigma", base * chan + 3.0 * (base * chan).std(), W, g, beta, b)
    run("row mean of 10 sigma", base * chan + 10.0 * (base * chan).std(), W, g, beta, b)
    x = base * chan + 3.0
    x[:, [3, 400, 777]] *= 40.0
    run("mean 3 + massive channels", x, W, g, beta, b)
    g2 = g.clone()
    g2[[3, 400, 777]] = 0.02           # the usual companion of massive channels: a tiny LayerNorm gain on them
    run("mean 3 + massive channels with gain 0.02 on them", x, W, g2, beta, b)


if __name__ == "__main__":
    main()
