#!/usr/bin/env python
"""A/B the AR decode-step graph in one process (env knobs applied per session).
usage: python tools/ar_step_bench.py "M5_AR_PREFETCH=0" "M5_AR_PREFETCH=1" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession

def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codeclm.engine()
    ref_codes = synth.make_ref_codes(450, seed=7)
    P, N = 488, 450
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(bundle.n_text, eng.shape.n_vocab - 1, (P,), generator=g)
    V = eng.shape.n_vocab
    variants = sys.argv[1:] or ["M5_AR_PREFETCH=0", "M5_AR_PREFETCH=1"]
    noise = torch.ones(N, V, device=dev)
    for rnd in range(2):
        for v in variants:
            kv = dict(s.split("=") for s in v.split(",") if s)
            for k, val in kv.items():
                os.environ[k] = val
            sess = ARSession(eng, P + N)
            cfg = ARSamplingConfig(temperature=0.7, topk=100, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=100,
                                   eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=5000)
            sess.configure_sampler(cfg, bundle.n_text, V - 1, noise)
            sess.prefill(prompt, ref_codes[0].T.contiguous())
            out = sess.decode(use_graph=True)
            from mars5_tts_amd import ar_engine
            s = ar_engine.LAST_STATS
            print(f"round {rnd} {v:40s} {1e3 * s['decode_ms'] / max(s['n_generated'] - 1, 1):8.1f} us/token  ({s['n_generated']} tokens, checksum {int(out.sum())}, persistent {int(sess.mega)}, err {int(sess.mega_err[0])})", flush=True)
            for k in kv:
                os.environ.pop(k, None)
            del sess

if __name__ == "__main__":
    main()
