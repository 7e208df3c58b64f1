#!/usr/bin/env python
"""GPU microbench + correctness probe for m5_attention (16-bit operands).  M5_ATTN_V1=1 times the
first-generation kernel.  Reference: torch fp32 softmax(QK^T * scale + mask) V on the same operands."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
REP = int(os.environ.get("REP", "20"))

def case(name, B, H, Sq, Sk, causal=False, key_len=None):
    g = torch.Generator().manual_seed(5)
    Sr, Skr = (Sq + 63) // 64 * 64, (Sk + 63) // 64 * 64
    q = torch.randn(B, H, Sr, 64, generator=g).to(dev, dt)
    k = torch.randn(B, H, Skr, 64, generator=g).to(dev, dt)
    v = torch.randn(B, H, Skr, 64, generator=g).to(dev, dt)
    vt = v.transpose(2, 3).contiguous()                       # [B,H,64,Skr]
    D = H * 64
    o = torch.zeros(B, Sr, D, dtype=dt, device=dev)
    kl = torch.tensor(key_len, dtype=torch.int32, device=dev) if key_len is not None else None
    a = L.AttnArgs(q=q.data_ptr(), q_bs=H * Sr * 64, q_hs=Sr * 64, q_rs=64, k=k.data_ptr(), k_bs=H * Skr * 64, k_hs=Skr * 64, k_rs=64,
                   vt=vt.data_ptr(), vt_bs=H * 64 * Skr, vt_hs=64 * Skr, vt_ds=Skr, o=o.data_ptr(), o_bs=Sr * D, o_rs=D,
                   B=B, H=H, Sq=Sq, Sk=Sk, key_len=kl.data_ptr() if kl is not None else None, causal=1 if causal else 0,
                   scale=0.125, kv_index=None, kv_index_stride_k=0, kv_index_stride_v=0)
    torch.cuda.synchronize()          # q / k / vt / the zero fill of o were enqueued on the default stream: without this the fill can land AFTER the kernel's stores
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    with torch.cuda.stream(stream):
        ops.attention(dt, a, stream=st)
    stream.synchronize()
    s = torch.einsum("bhqd,bhkd->bhqk", q[:, :, :Sq].float(), k[:, :, :Sk].float()) * 0.125
    if causal:
        s = s.masked_fill(torch.ones(Sq, Sk, device=dev, dtype=torch.bool).triu(1), float("-inf"))
    if key_len is not None:
        for b in range(B):
            s[b, :, :, key_len[b]:] = float("-inf")
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v[:, :, :Sk].float())
    got = o[:, :Sq].view(B, Sq, H, 64).permute(0, 2, 1, 3).float()
    err = (got - ref).abs().max().item()
    csum = int(o[:, :Sq].contiguous().view(torch.int16).to(torch.int64).mul(torch.arange(1, B * Sq * D + 1, device=dev).view(B, Sq, D) % 8191 + 1).sum().item())
    if os.environ.get("STRESS"):          # the same launch N times: how many outputs differ from the first one's?
        bad = 0
        first = o.clone()
        with torch.cuda.stream(stream):
            for i in range(int(os.environ["STRESS"])):
                o.zero_()
                ops.attention(dt, a, stream=st)
                stream.synchronize()
                if not torch.equal(o, first):
                    bad += 1
        print(f"{name:28s} stress: {bad} of {os.environ['STRESS']} launches differ from the first (first maxerr {err:.4g})", flush=True)
        return
    with torch.cuda.stream(stream):
        ops.Graph.begin(st)
        for _ in range(REP):
            ops.attention(dt, a, stream=st)
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
    fl = 4.0 * B * H * Sq * Sk * 64 * (0.5 if causal else 1.0)
    print(f"{name:28s} B={B} H={H} Sq={Sq} Sk={Sk}  {us:8.2f} us  {fl / us / 1e6:7.1f} TF  maxerr={err:.4g} csum={csum}", flush=True)

if __name__ == "__main__":
    if os.environ.get("CASES"):            # CASES="B,H,Sq,Sk;B,H,Sq,Sk;..." : grid-shape experiments
        for c in os.environ["CASES"].split(";"):
            B, H, Sq, Sk = (int(v) for v in c.split(","))
            case(f"case {c}", B, H, Sq, Sk)
        sys.exit(0)
    case("nar self", 2, 16, 1349, 1349)
    case("nar self keylen", 2, 16, 1349, 1349, key_len=[1349, 1000])
    case("nar cross", 2, 16, 1349, 39)
    case("ar prefill causal", 1, 24, 489, 489, causal=True)
    case("spk enc", 1, 16, 451, 451)
    case("small odd", 3, 2, 77, 130, key_len=[130, 1, 65])
    case("causal odd", 1, 3, 200, 200, causal=True)
