"""Same-host CPU timing of the UNMODIFIED reference functions (``/root/reference``: ``ar_generate``,
``perform_simple_inference``) against this repo's oracle port, on the BASELINE configs[1] workload (full-size seeded
weights, 450-frame reference, deep clone).  TEST INFRASTRUCTURE; build container only (the reference does not travel).

bench.py's ``cpu_baseline`` is ``kind: "port"`` (the oracle timed on the GPU box's host cores, because the reference
cannot be shipped there); this script measures, once, how the port's cost relates to the reference's on one host, so the
port number carries a measured port/reference ratio.  Output: profiles/r2_ref_vs_port_cpu.json.

    python oracle/time_ref_vs_port.py [--tokens 16] [--nar-steps 2]
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))

import warnings  # noqa: E402
warnings.filterwarnings("ignore")

from mars5.ar_generate import ar_generate                                          # noqa: E402  (reference)
from mars5.diffuser import DSH, MultinomialDiffusion, perform_simple_inference      # noqa: E402  (reference)

import gen_golden as G                                                              # noqa: E402
import mars5_oracle as O                                                            # noqa: E402
from mars5_tts_amd import synth                                                     # noqa: E402

TEXT = "The quick brown rat jumped over the lazy dogs twice."
TRANSCRIPT = "We actually haven't managed to meet demand this year."


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=16)
    ap.add_argument("--nar-steps", type=int, default=2)
    args = ap.parse_args()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    b = synth.make_bundle("full", seed=0)
    tt, st = G.ref_tokenizers(b.ar_ckpt["vocab"])
    lm, nar = G.ref_models(b)
    ref_codes = synth.make_ref_codes(450, seed=7)
    text_full = tt.encode("<|startoftext|>" + TRANSCRIPT + " " + TEXT.strip() + "<|endoftext|>", allowed_special="all")
    sp = st.encode(" ".join(str(t) for t in ref_codes[0, 0].tolist()))
    prompt = torch.tensor(text_full + [p + len(tt.vocab) for p in sp], dtype=torch.long)
    P, N = int(prompt.shape[0]), args.tokens
    spk_ref = ref_codes[0].T.contiguous()
    kw = dict(temperature=0.7, topk=100, top_p=0.2, typical_p=1.0, alpha_frequency=3, alpha_presence=0.4, penalty_window=100,
              eos_penalty_decay=0.5, eos_penalty_factor=50.0)
    res = {"host": {"cpu": cpu_model(), "cores": cores, "torch": torch.__version__}, "workload": f"BASELINE configs[1]: P={P}, deep clone, full-size seeded weights"}

    # ---- AR: prefill + N tokens, reference then port (greedy-equivalent settings do not matter for cost)
    with torch.inference_mode():
        torch.manual_seed(1)
        t0 = time.perf_counter()
        out_ref = ar_generate(tt, st, lm, prompt, spk_ref, len(text_full) + 1, max_len=P + N, fp16=False, vocode=False, use_kv_cache=True,
                              n_phones_gen=100 * len(TEXT), **kw)
        t_ref_ar = time.perf_counter() - t0
        p = O.ARSamplingParams(temperature=0.7, top_k=100, top_p=0.2, alpha_frequency=3, alpha_presence=0.4, penalty_window=100,
                               eos_penalty_decay=0.5, eos_penalty_factor=50.0, n_phones_gen=100 * len(TEXT))
        g = torch.Generator().manual_seed(1)
        t0 = time.perf_counter()
        out_port = O.ar_generate_oracle(b.ar_ckpt["model"], b.ar_shape.nhead, b.n_text, b.n_speech, st.special_tokens["<|endofspeech|>"],
                                        prompt, spk_ref, P + N, p, generator=g, recompute_spk=True)
        t_port_ar = time.perf_counter() - t0
    res["ar"] = {"tokens": N, "reference_s": round(t_ref_ar, 3), "port_s": round(t_port_ar, 3), "port_over_reference": round(t_port_ar / t_ref_ar, 3),
                 "tokens_equal": bool(torch.equal(out_ref, out_port))}
    print(json.dumps(res["ar"]), flush=True)

    # ---- NAR: `nar-steps` reverse steps at the bench shape (S = 1349), reference then port
    T = args.nar_steps
    c_text = torch.tensor(text_full)[None]
    c_codes = ref_codes.permute(0, 2, 1).contiguous()
    x_l0 = torch.randint(0, 1024, (899,), generator=torch.Generator().manual_seed(2))      # 449 prompt frames + 450 generated: what the AR hands over
    _x = x_l0[None, :, None].repeat(1, 1, 8)
    batch = (c_text, c_codes, torch.tensor([c_text.shape[1]]), torch.tensor([c_codes.shape[1]]), _x, torch.zeros(1, _x.shape[1], dtype=torch.bool))
    with torch.inference_mode():
        torch.manual_seed(3)
        t0 = time.perf_counter()
        o_ref = perform_simple_inference(nar, batch, MultinomialDiffusion(1025, timesteps=200, device="cpu"), T, torch.float16,
                                         dsh=DSH(last_greedy=True, x_0_temp=0.7, guidance_w=3, deep_clone=True, jump_len=1, jump_n_sample=1,
                                                 q0_override_steps=20, enable_kevin_scaled_inference=True, progress=False), retain_quant0=True)
        t_ref_nar = time.perf_counter() - t0
        g = torch.Generator().manual_seed(3)
        t0 = time.perf_counter()
        o_port = O.perform_simple_inference_oracle(b.nar_ckpt["model"], b.nar_shape.nhead, c_text[0], c_codes[0], x_l0,
                                                   O.NARParams(T=T, deep_clone=True), generator=g, hoist=False)
        t_port_nar = time.perf_counter() - t0
    res["nar"] = {"reverse_steps": T, "S": int(c_codes.shape[1] + x_l0.shape[0]), "reference_s": round(t_ref_nar, 3), "port_s": round(t_port_nar, 3),
                  "port_over_reference": round(t_port_nar / t_ref_nar, 3), "ids_equal": bool(torch.equal(o_ref[0], o_port))}
    print(json.dumps(res["nar"]), flush=True)
    ar_tok = (t_ref_ar) / N
    res["reference_extrapolated_s_per_utterance"] = round(450 * t_ref_ar / N + 200 * t_ref_nar / T, 1)
    res["port_extrapolated_s_per_utterance"] = round(450 * t_port_ar / N + 200 * t_port_nar / T, 1)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r2_ref_vs_port_cpu.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
