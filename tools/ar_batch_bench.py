#!/usr/bin/env python
"""AR decode-step cost against the number of sequences decoded together (BASELINE config 3).
usage: python tools/ar_batch_bench.py [B ...]   (default 1 2 4 8 16 32; B = 1 also times the batch-1 GEMV graph)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.ar_engine import ARBatchSession, ARSamplingConfig, ARSession


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codeclm.engine()
    V = eng.shape.n_vocab
    P, N = 488, 96
    g = torch.Generator().manual_seed(3)
    cfg = ARSamplingConfig(temperature=0.7, topk=100, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=100,
                           eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=2000)
    eos = V - 1
    wbytes = eng.weight_bytes_per_token()
    Bs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
    for B in Bs:
        prompts = [torch.randint(bundle.n_text, V - 1, (P - 7 * (b % 5),), generator=g) for b in range(B)]
        refs = [synth.make_ref_codes(450, seed=7 + b)[0].T.contiguous() for b in range(B)]
        if B == 1:
            s1 = ARSession(eng, P + N)
            s1.configure_sampler(cfg, bundle.n_text, eos, torch.ones(N, V, device=dev))
            s1.prefill(prompts[0], refs[0])
            s1.decode()
            from mars5_tts_amd.ar_engine import LAST_STATS
            print(f"B=  1 (GEMV graph)   {LAST_STATS['decode_ms'] * 1e3 / LAST_STATS['decode_steps_launched']:8.1f} us/step", flush=True)
        bs = ARBatchSession(eng, [p.shape[0] + N for p in prompts])
        bs.configure_sampler(cfg, bundle.n_text, eos, torch.ones(B, N, V, device=dev))
        bs.prefill(prompts, refs)
        bs.decode()
        from mars5_tts_amd.ar_engine import LAST_STATS
        us = LAST_STATS['decode_ms'] * 1e3 / LAST_STATS['decode_steps_launched']
        print(f"B={B:3d} (batched graph) {us:8.1f} us/step  {us / B:7.1f} us/token  weights at {wbytes / us / 1e3:6.0f} GB/s  "
              f"generated {LAST_STATS['n_generated'][:4]}", flush=True)
        del bs


if __name__ == "__main__":
    main()
