#!/usr/bin/env python
"""Where a layer of the persistent AR decode step (csrc/ar_mega.hip) spends its time: workgroup 0's waves stamp the 100 MHz
wall clock at every phase boundary (tools build); printed as us since the layer's start, mean over layers 2..25 of a few
eager steps (wave 0 = rows in every phase, wave 4 = gatherer with rows in P4 / P5, wave 7 = gatherer only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")
os.environ["M5_AR_MEGA"] = "1"
import torch
import bench
from mars5_tts_amd import synth
from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
m, bundle = bench.build_model("bf16", dev)
eng = m.codeclm.engine()
ref_codes = synth.make_ref_codes(450, seed=7)
P, N = 488, 64
g = torch.Generator().manual_seed(3)
V = eng.shape.n_vocab
prompt = torch.randint(bundle.n_text, V - 1, (P,), generator=g)
noise = torch.ones(N, V, device=dev)
sess = ARSession(eng, P + N)
cfg = ARSamplingConfig(temperature=0.7, topk=100, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=100,
                       eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=5000)
sess.configure_sampler(cfg, bundle.n_text, V - 1, noise)
sess.prefill(prompt, ref_codes[0].T.contiguous())
sess.mega_dbg = torch.zeros(4, 32, 16, 8, dtype=torch.int64, device=dev)
st = sess.stream.cuda_stream
sess.enqueue_head_and_sample(st)
acc = None
names = ["layer start", "P1 gathered", "P1 done", "P2 gathered (q k v)", "P2 done", "P3 gathered (O)", "P3 done",
         "P4 gathered (x)", "P4 done", "P5 gathered (h)", "P5 done", "  P1 normed", "  P1 weights landed", "  P1 products done",
         "  P4 normed", "  P4 weights landed"]
n = 0
for step in range(12):
    sess.enqueue_layers(st)
    sess.enqueue_head_and_sample(st)
    sess.stream.synchronize()
    if step >= 2:
        d = sess.mega_dbg.cpu().double()[:, 2:26, :16, :]       # (workgroup, layers, stamps, waves)
        rel = (d - d[0:1, :, 0:1, 0:1]) / 100.0                  # us since workgroup 0 / wave 0's layer start
        acc = rel.mean(1) if acc is None else acc + rel.mean(1)
        n += 1
acc /= n
print(f"persistent {int(sess.mega)} err {int(sess.mega_err[0])}; us since workgroup 0's layer start (mean of 24 layers x {n} steps)")
for wi, wn in enumerate(["workgroup 0 (cache scan)", "workgroup 192 (head merge)", "workgroup 255 (no P2 work)", "workgroup 100 (cache scan)"]):
    print(f"-- {wn}")
    print(f"{'':24s} " + " ".join(f"wave{w:1d}  " for w in range(8)))
    for k, nm in enumerate(names):
        print(f"{nm:24s} " + " ".join(f"{float(acc[wi, k, w]):6.2f} " for w in range(8)))
