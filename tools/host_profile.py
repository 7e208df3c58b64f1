#!/usr/bin/env python
"""Where does the host time of one bench utterance go?  cProfile over the third tts_from_codes call (graphs of the first two
are gone: every utterance builds its own sessions), sorted by cumulative time; plus wall-clock marks around the stages."""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from mars5_tts_amd import synth


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    m, bundle = bench.build_model("bf16", dev)
    ref_codes = synth.make_ref_codes(450, seed=7).to(dev)
    p_len, n_text_tok = bench.prompt_len(m, ref_codes)
    cfg = bench.make_cfg(n_text_tok, p_len, 450)
    for i in range(2):
        bench.run_utterance(m, ref_codes, cfg, 500 + i)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    bench.run_utterance(m, ref_codes, cfg, 777)
    pr.disable()
    print(f"utterance wall {time.perf_counter() - t0:.4f} s; bench log {bench.STEP_LOG[-1]}")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
