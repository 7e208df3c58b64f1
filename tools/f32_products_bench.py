#!/usr/bin/env python
"""fp32 engines, exact fp32-MFMA products vs split-f16 products (csrc/gemm.hip X3, ops.set_f32_products): (1) the NAR GEMM
shapes back to back (graph-timed, error against a float64 product of the same operands), (2) the full-size fp32 NAR reverse
step in both modes, same process, alternating (ms per step and the largest logit difference between the modes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools library: M5_X3_BM forces the tile height of the split-f16 kernel
import torch
import bench
from mars5_tts_amd import synth, ops, _lib as L
from mars5_tts_amd.nar_engine import NARConfig, NARSession

dev = torch.device("cuda:0")
REP = 10


def gemm_case(name, M, N, K, epi):
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ w.double().T + b.double()
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    for mode, bm in (("exact", None), ("f16x3", "128"), ("f16x3", "64"), ("f16x3", None)):
        os.environ.pop("M5_X3_BM", None)
        if bm:
            os.environ["M5_X3_BM"] = bm
        prev = ops.set_f32_products(mode)
        out = torch.zeros(M, N, device=dev)
        with torch.cuda.stream(stream):
            ops.gemm(a, w, out, L.EPI_F32, bias=b, stream=st)
            stream.synchronize()
            err = float((out.double() - ref).abs().max() / ref.abs().max())
            ops.Graph.begin(st)
            for _ in range(REP):
                ops.gemm(a, w, out, epi, bias=b, stream=st)
            gr = ops.Graph().end(st)
            gr.launch(st)
            stream.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(st)
            for _ in range(3):
                gr.launch(st)
            e1.record(st)
            stream.synchronize()
        ops.set_f32_products(prev)
        us = e0.elapsed_ms(e1) * 1e3 / (3 * REP)
        print(f"{name:16s} M={M} N={N} K={K} {mode:6s} bm={bm or 'auto':4s} {us:9.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF  rel err vs float64 {err:.2e}", flush=True)


def step_bench():
    m, bundle = bench.build_model("f32", dev)
    eng = m.codecnar.engine()
    ref_codes = synth.make_ref_codes(450, seed=7).to(dev)
    S, off, Le = 1349, 899, 39
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 1025, (S, 8), generator=g)
    c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
    z = torch.zeros(S, 8, dtype=torch.long)
    mm = torch.zeros(S, 8, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1
    logits = {}
    for rnd in range(2):
        for mode in ("exact", "f16x3"):
            prev = ops.set_f32_products(mode)
            sess = NARSession(eng, NARConfig(T=200))
            sess.prepare(c_text, ref_codes[0].T.contiguous(), x, z, mm, off, list(range(199, 189, -1)))
            st = sess.stream.cuda_stream
            sess.enqueue_forward(st)
            sess.stream.synchronize()
            logits[mode] = sess.logits.float().cpu().clone()
            gen = torch.Generator(device=dev).manual_seed(1)
            from mars5_tts_amd.diffuser import _generator_uniform
            uni = _generator_uniform(dev, gen)
            sess.run(uni, True, n_steps=2)
            e0, e1 = ops.Event(), ops.Event()
            e0.record(st)
            sess.run(uni, True, n_steps=4)
            e1.record(st)
            sess.stream.synchronize()
            ops.set_f32_products(prev)
            print(f"round {rnd} fp32 NAR step, products {mode:6s} {e0.elapsed_ms(e1) / 4:8.3f} ms/step", flush=True)
            del sess
    d = (logits["exact"] - logits["f16x3"]).abs().max()
    print(f"first-step logits: max |exact - f16x3| = {float(d):.3e}  (max |logit| {float(logits['exact'].abs().max()):.2f})", flush=True)


if __name__ == "__main__":
    for c in [("qkv", 2816, 3072, 1024), ("out_proj", 2816, 1024, 1024), ("swiglu-shape", 2816, 6144, 1024), ("linear2", 2816, 1024, 3072),
              ("heads", 1798, 1025, 1024)]:
        gemm_case(*c, L.EPI_F32)
    os.environ.pop("M5_X3_BM", None)
    if os.environ.get("STEP", "1") == "1":
        step_bench()
