#!/usr/bin/env python
"""Would two independent launch chains fill each other's bubbles?  The NAR reverse step is one dependent chain of ~230 launches
whose fixed cost (boundary, first cold operand, epilogue drain) is 30-40 % of its time.  This probe runs
  (a) ONE session, both guidance branches in every launch (the product form, M = 2 x 1408 rows per launch),
  (b) ONE session with one branch only (guidance_w = 1: M = 1408 rows per launch) -- the half-size chain alone,
  (c) TWO such one-branch sessions on two streams, steps enqueued alternately -- the two chains side by side,
  (d) TWO full sessions on two streams (two utterances in flight),
and prints ms per step of each.  (c) < (a) would say: capture the two branches as parallel graph branches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.nar_engine import NARConfig, NARSession

STEPS = int(os.environ.get("STEPS", "30"))


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codecnar.engine()
    ref_codes = synth.make_ref_codes(450, seed=7).to(dev)
    S, off, Le = 1349, 899, 39
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 1025, (S, 8), generator=g)
    c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
    z = torch.zeros(S, 8, dtype=torch.long)
    mm = torch.zeros(S, 8, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1

    def make(gw):
        sess = NARSession(eng, NARConfig(T=200, guidance_w=gw), stream=torch.cuda.Stream())
        sess.prepare(c_text, ref_codes[0].T.contiguous(), x, z, mm, off, list(range(199, 199 - 3 * STEPS - 6, -1)))
        gen = torch.Generator(device=dev).manual_seed(1)
        uni = lambda shp: torch.rand(shp, generator=gen, device=dev)
        sess.run(uni, True, n_steps=5)            # captures the step graph
        sess.stream.synchronize()
        return sess, uni

    def timed(pairs):
        for s, _ in pairs:
            s.stream.synchronize()
        ev = [(ops.Event(), ops.Event()) for _ in pairs]
        for (s, _), (e0, _) in zip(pairs, ev):
            e0.record(s.stream.cuda_stream)
        for _ in range(STEPS):
            for s, u in pairs:
                s.step(u, True)
        for (s, _), (_, e1) in zip(pairs, ev):
            e1.record(s.stream.cuda_stream)
        for s, _ in pairs:
            s.stream.synchronize()
        return [e0.elapsed_ms(e1) / STEPS for e0, e1 in ev]

    for rnd in range(2):
        full = make(3.0)
        print(f"round {rnd} (a) one session, two branches per launch      {timed([full])[0]:7.3f} ms/step", flush=True)
        del full
        one = make(1.0)
        print(f"round {rnd} (b) one session, one branch                   {timed([one])[0]:7.3f} ms/step", flush=True)
        two = make(1.0)
        t = timed([one, two])
        print(f"round {rnd} (c) two one-branch sessions, two streams      {max(t):7.3f} ms per step pair  ({t[0]:.3f} / {t[1]:.3f})", flush=True)
        del one, two
        f1, f2 = make(3.0), make(3.0)
        t = timed([f1, f2])
        print(f"round {rnd} (d) two full sessions, two streams            {max(t):7.3f} ms per step pair  ({t[0]:.3f} / {t[1]:.3f})", flush=True)
        del f1, f2


if __name__ == "__main__":
    main()
