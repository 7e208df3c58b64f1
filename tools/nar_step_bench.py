#!/usr/bin/env python
"""A/B the NAR reverse-step graph in one process: env knobs are applied per session.
usage: python tools/nar_step_bench.py "M5_NAR_DUAL=0" "M5_NAR_DUAL=1" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.nar_engine import NARConfig, NARSession

def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codecnar.engine()
    ref_codes = synth.make_ref_codes(450, seed=7).to(dev)
    S, off, Le = 1349, 899, 39          # bench shapes: 450 + 449 prompt frames, 450 generated
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 1025, (S, 8), generator=g)
    c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
    z = torch.zeros(S, 8, dtype=torch.long)
    mm = torch.zeros(S, 8, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1
    variants = sys.argv[1:] or ["M5_NAR_DUAL=0", "M5_NAR_DUAL=1"]
    for rnd in range(2):
        for v in variants:
            kv = dict(s.split("=") for s in v.split(",") if s)
            for k, val in kv.items():
                os.environ[k] = val
            # M5_NAR_SAMEW=1 (experiment, WRONG results): every decoder layer uses layer 0's weights, so the step's ~0.5 GB weight
            # stream collapses to 30 MB that stay cache-resident -- how much of the step is cold-weight latency?
            if not hasattr(eng, "_dec_orig"):
                eng._dec_orig = list(eng.dec)
            eng.dec = [eng._dec_orig[0]] * len(eng._dec_orig) if os.environ.get("M5_NAR_SAMEW") == "1" else list(eng._dec_orig)
            sess = NARSession(eng, NARConfig(T=200))
            sess.prepare(c_text, ref_codes[0].T.contiguous(), x, z, mm, off, list(range(199, 159, -1)))
            gen = torch.Generator(device=dev).manual_seed(1)
            from mars5_tts_amd.diffuser import _generator_uniform
            uni = _generator_uniform(dev, gen)          # the product's draw (generator-backed: the uniforms are generated inside the step graph; M5_NAR_PHILOX=0: two torch.rand launches)
            sess.run(uni, True, n_steps=5)
            st = sess.stream.cuda_stream
            e0, e1 = ops.Event(), ops.Event()
            e0.record(st)
            sess.run(uni, True, n_steps=30)
            e1.record(st)
            sess.stream.synchronize()
            print(f"round {rnd} {v:40s} {e0.elapsed_ms(e1) / 30:7.3f} ms/step", flush=True)
            for k in kv:
                os.environ.pop(k, None)
            del sess

if __name__ == "__main__":
    main()
