#!/usr/bin/env python
"""NAR reverse-step cost against the number of utterances refined together (BASELINE config 3).
usage: python tools/nar_batch_bench.py [U ...]   (default 1 2 4 8; all utterances S = 1349 unless MIXED=1)
GRAPH=0: launches go out one by one (rocprofv3 cannot trace the 310-node batched graph).
CONC=1: for every U also TWO sessions of U utterances each on two streams, steps enqueued alternately (two groups in flight)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")      # tools run on libmars5_hip_tools.so (knobs, probes; csrc/common.h)
import torch
import bench
from mars5_tts_amd import synth, ops
from mars5_tts_amd.nar_engine import NARBatchSession, NARConfig, NARSession


def item(eng, dev, S, off, Le, seed):
    g = torch.Generator().manual_seed(seed)
    ref = synth.make_ref_codes(off, seed=seed).to(dev)[0].T.contiguous()
    x = torch.randint(0, 1025, (S, 8), generator=g)
    c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
    z = torch.zeros(S, 8, dtype=torch.long)
    mm = torch.zeros(S, 8, dtype=torch.uint8); mm[:, 0] = 1; mm[:off] = 1
    return dict(c_text=c_text, c_codes=ref, x=x, x_known=z, m_mask=mm, row_offset=off)


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    eng = m.codecnar.engine()
    Us = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    mixed = os.environ.get("MIXED", "0") == "1"
    times = list(range(199, 179, -1))
    use_graph = os.environ.get("GRAPH", "1") != "0"
    for U in Us:
        g = torch.Generator().manual_seed(11)
        items = []
        for u in range(U):
            if mixed:      # reference 2-12 s (150-900 frames), 450 generated frames
                off = int(torch.randint(150, 901, (1,), generator=g))
                Le = int(torch.randint(20, 80, (1,), generator=g))
            else:
                off, Le = 450, 39
            items.append(item(eng, dev, off + 449 + 450 if not mixed else 2 * off + 450, off, Le, 100 + u))
        sess = NARBatchSession(eng, NARConfig(T=200))
        sess.prepare(items, times)
        gens = [torch.Generator(device=dev).manual_seed(u) for u in range(U)]
        unis = [(lambda shp, g=g_: torch.rand(shp, generator=g, device=dev)) for g_ in gens]
        sess.run(unis, use_graph, n_steps=3)
        st = sess.stream.cuda_stream
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        n = 10
        sess.run(unis, use_graph, n_steps=n)
        e1.record(st)
        sess.stream.synchronize()
        ms = e0.elapsed_ms(e1) / n
        fl = sum(eng.flops_per_step(s.S, s.mems[0].Le, s.s_out) for s in sess.subs)
        print(f"U={U:3d} rows={sess.ws.M:6d} (real {sum(2 * s.S for s in sess.subs)})  {ms:8.3f} ms/step  {ms / U:7.3f} ms/step/utt  "
              f"{fl / ms / 1e9:7.1f} TF", flush=True)
        if os.environ.get("CONC") == "1":
            pair = []
            for k in range(2):
                s2 = NARBatchSession(eng, NARConfig(T=200), stream=torch.cuda.Stream())      # (sessions share one stream by default)
                s2.prepare(items, times)
                s2.run(unis, True, n_steps=3)
                pair.append(s2)
            ev = [(ops.Event(), ops.Event()) for _ in pair]
            for s2, (a0, _) in zip(pair, ev):
                s2.stream.synchronize()
            for s2, (a0, _) in zip(pair, ev):
                a0.record(s2.stream.cuda_stream)
            for _ in range(n):
                for s2 in pair:
                    s2.step(unis, True)
            for s2, (_, a1) in zip(pair, ev):
                a1.record(s2.stream.cuda_stream)
            for s2 in pair:
                s2.stream.synchronize()
            ms2 = max(a0.elapsed_ms(a1) for a0, a1 in ev) / n
            print(f"U={U:3d} x 2 sessions on two streams: {ms2:8.3f} ms per step pair  {ms2 / (2 * U):7.3f} ms/step/utt  "
                  f"({ms / U / (ms2 / (2 * U)):.3f}x the single-session rate)", flush=True)
            del pair
        del sess, items


if __name__ == "__main__":
    main()
