#!/usr/bin/env python
"""N independent lone NAR sessions (one utterance each, the batch-1 step graph) in flight on N streams: ms per utterance-step
against N.  The batch-1 launches are one workgroup round at one or two waves per SIMD; a second and third chain fill the
CUs' idle issue slots without any kernel being aware of it.  usage: python tools/nar_lanes_probe.py [N ...]  (default 1 2 3 4 6)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.diffuser import _generator_uniform
from mars5_tts_amd.nar_engine import NARConfig, NARSession

STEPS = int(os.environ.get("STEPS", "40"))


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codecnar.engine()
    S, off, Le = 1349, 450, 39            # the driver bench's utterance: 450 prompt frames, 899 sampled rows
    lanes = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        ref = synth.make_ref_codes(450, seed=seed).to(dev)
        x = torch.randint(0, 1025, (S, 8), generator=g)
        c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
        z = torch.zeros(S, 8, dtype=torch.long)
        mm = torch.zeros(S, 8, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1
        sess = NARSession(eng, NARConfig(T=200), stream=torch.cuda.Stream())
        sess.prepare(c_text, ref[0].T.contiguous(), x, z, mm, off, list(range(199, 199 - 2 * STEPS - 8, -1)))
        uni = _generator_uniform(dev, torch.Generator(device=dev).manual_seed(seed))
        sess.run(uni, True, n_steps=4)            # captures the step graph
        return sess, uni

    for rnd in range(2):
        for n in lanes:
            pairs = [make(100 + i) for i in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s, u in pairs:
                s.run(u, True, n_steps=STEPS, wait=False)
            for s, _ in pairs:
                s.finish()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / STEPS
            print(f"round {rnd} lanes={n}   {ms:8.3f} ms per step of all lanes   {ms / n:7.3f} ms per utterance-step", flush=True)
            del pairs


if __name__ == "__main__":
    main()
