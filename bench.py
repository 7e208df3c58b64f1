#!/usr/bin/env python
"""bench.py -- MARS5 hot-path throughput on MI355X.

python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)

A "step" is one utterance through the hot path ``Mars5TTS.tts_from_codes`` (= ``tts()`` between
the Encodec encoder and the Vocos decoder): prompt tokens + reference codes in -> AR prefill +
decode loop -> BPE expansion -> 200-step NAR diffusion -> final (S_out, 8) codec codes out.
Workload = BASELINE.json configs[1]: single utterance, deep clone, temperature 0.7, top_k 100,
bf16 operands, synthetic 6 s / 24 kHz reference (450 frames), ~20-token text + ~20-token
transcript, 450 generated frames, seeded random weights of the real geometry (n_vocab 4096).
Each rank processes its own utterances (replicas; no data-path collective), weak scaling.
One JSON line on rank 0; ``value`` = generated audio seconds per wall second over all ranks.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

TEXT = "The quick brown rat jumped over the lazy dogs twice."
TRANSCRIPT = "We actually haven't managed to meet demand this year."
PEAK_MFMA_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # MI355X_MICROARCH.md dense peaks
PEAK_HBM_GBS = 8000.0


PREFLIGHT = None     # filled by preflight(): what the pre-flight child saw (also carried in the JSON line)

_PREFLIGHT_CHILD = r"""
import sys, time, torch
t0 = time.time()
i = int(sys.argv[1])
torch.cuda.set_device(i)
d = torch.device("cuda", i)
x = torch.ones(1 << 20, device=d) * 2
assert float(x.sum().item()) == float(2 << 20), "arithmetic"
h = torch.arange(1 << 24, dtype=torch.int32).pin_memory()
g = h.to(d, non_blocking=True)
big = torch.empty(1 << 28, dtype=torch.uint8, device=d).fill_(7)
cp = big.clone()
torch.cuda.synchronize()
assert int(g[-1].item()) == (1 << 24) - 1 and int(g.to(torch.int64).sum().item()) == ((1 << 24) - 1) * (1 << 23), "host -> device copy"
assert bool(cp.view(torch.int64).eq(0x0707070707070707).all().item()), "fill / device -> device copy"
sys.path.insert(0, sys.argv[2])
from mars5_tts_amd import ops                      # the product library loads and one of its kernels runs
xs = torch.randn(64, 1024, device=d)
o = torch.empty(64, 1024, device=d, dtype=torch.bfloat16)
ops.layernorm(xs, torch.ones(1024, device=d), torch.zeros(1024, device=d), 1e-5, o)
torch.cuda.synchronize()
ref = torch.nn.functional.layer_norm(xs, (1024,))
assert float((o.float() - ref).abs().max()) < 0.05, "libmars5_hip layernorm"
print("PREFLIGHT_OK %.1f" % (time.time() - t0))
"""


def preflight(local: int, timeout_s: float = 240.0) -> dict:
    """Two seconds of GPU work in a CHILD process before anything else touches the device: fill, host->device and
    device->device copies, a reduction, one launch of the product library.  A sick box (round 3: `Memory access fault by GPU`
    before the first utterance had finished) kills the child, not the bench: rank 0 then prints a line that says so and the
    process exits with code 3 -- distinct from a product failure (1 / 134) and from a refused launch (2)."""
    import subprocess
    busy = None
    try:
        import glob
        vals = [int(open(f).read()) for f in sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))]
        busy = vals
    except Exception:       # noqa: BLE001
        pass
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", _PREFLIGHT_CHILD, str(local), ROOT], capture_output=True, text=True, timeout=timeout_s)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, so, se = -9, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "pre-flight child timed out"
    res = {"ok": rc == 0 and "PREFLIGHT_OK" in so, "rc": rc, "seconds": round(time.perf_counter() - t0, 1), "gpu_busy_percent_before": busy}
    if not res["ok"]:
        res["stderr_tail"] = se[-600:]
    return res


def run_leg(out: dict, name: str, fn, sync=None) -> bool:
    """One optional leg of the line: `fn()` fills `out`; an exception is recorded as out["<name>_error"] (and printed to stderr)
    instead of propagating, the enriched line is re-printed either way.  Returns whether the leg completed."""
    t_leg = time.perf_counter()
    done = False
    try:
        fn()
        out.setdefault("legs_done", []).append(name)
        done = True
    except Exception as e:          # noqa: BLE001 -- a diagnostic leg must never cost the headline
        import traceback
        out[f"{name}_error"] = f"{type(e).__name__}: {e}"[:600]
        traceback.print_exc(file=sys.stderr)
        if sync is not None:
            try:
                sync()
            except Exception as e2:     # noqa: BLE001
                out[f"{name}_error"] += f" | device unusable afterwards: {e2}"[:200]
    out.setdefault("leg_seconds", {})[name] = round(time.perf_counter() - t_leg, 1)
    emit(out)
    return done


def emit(out: dict) -> None:
    """One JSON line on stdout, flushed.  Called after the timed loop and again after every leg: the LAST line is the
    complete record, every earlier one is a valid (smaller) record of the same run."""
    print(json.dumps(out), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default=os.environ.get("MARS5_DTYPE", "bf16"), choices=["bf16", "f16", "f32"])
    ap.add_argument("--ref-frames", type=int, default=450)
    ap.add_argument("--n-gen", type=int, default=450)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cpu_baseline: torch threads of the CPU sample (0 = min(logical CPUs, 32), the measured "
                                                               "optimum: profiles/r5f_cpu_baseline_threads.txt)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-batch-leg", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the configs[2] (32 mixed-length requests) and configs[4] (60 s long-form) "
                    "one-step legs that the default N = 1 run reports beside the headline")
    ap.add_argument("--pipeline", action="store_true", help="also time the same utterances through tts_stream_from_codes (AR of request i+1 "
                    "enqueued beside the NAR steps of request i); measured in round 2: no overlap on this stack (5.90 vs 5.86 audio-s/s)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-preflight", action="store_true", help="skip the child-process GPU sanity check in front of the run")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default): BASELINE configs[1], one utterance per step.  c3: configs[2], one step = a batch of "
                         "--batch mixed-length requests through tts_batch_from_codes.  c4: configs[3], one step = --batch x N "
                         "mixed-length requests built on rank 0, scattered over the N ranks (RCCL), run, gathered back and "
                         "spot-checked (32 x 8 = 256 on a full node).  c5: configs[4], one long-form utterance "
                         "per step (60 s = 4500 generated frames: the AR context passes the 3000-slot rotating KV window)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ar-batch", type=int, default=32)
    ap.add_argument("--nar-batch", type=int, default=32, help="c3: utterances refined together per NAR group (32 since round 4: the row-tile "
                    "lists make padding a group to its longest member free -- 9.40 / 9.55 / 9.79 audio-s/s at 8 / 16 / 32 on one box; 8 before)")
    ap.add_argument("--nar-in-flight", type=int, default=2, help="c3: NAR groups refined at once, each on its own stream")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + rank census only, then exit (no model, no GPU work): with --backend gloo this is how the "
                         "CPU tests check that `--gpus N` really starts N ranks")
    ap.add_argument("--c4-per-request", action="store_true", help="c4: every rank runs its shard request by request (round 1-3 behaviour) instead of "
                    "refining it in NAR groups (tts_batch_from_ids; identical results)")
    ap.add_argument("--check-bundle", action="store_true", help="with --launch-check: also build the (tiny) synthetic checkpoint on rank 0 and "
                    "map it on the other ranks, as the N-rank runs do with the full one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL on the GPU box)")
    ap.add_argument("--same-device", action="store_true", help="--launch-check with more ranks than visible GPUs: rank r uses device r %% device_count "
                    "(two RCCL ranks on one GPU, if the library accepts that; tests only)")
    ap.add_argument("--collective-at-1", action="store_true",
                    help="c4 under a launcher with WORLD_SIZE=1: still initialise the process group and send the requests / results through the "
                         "scatter / gather / census collectives (RCCL init, device-tensor scatter and gather, /dev/shm bundle path on ONE GPU: what "
                         "a 1-GPU box can prove about the multi-GPU path; tests/test_gpu_e2e.py)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves, one process per GPU,
    through torch.distributed.run on 127.0.0.1 (what the driver's command line does), and return its exit code.  Fails
    loudly when fewer than N devices are visible (a silent N = 1 run would be recorded as an N-GPU number)."""
    import socket
    import subprocess
    if args.backend == "nccl":
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus and not (args.same_device and args.launch_check and n_dev >= 1):
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible on this node; refusing to run", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # RCCL needs dmabuf IPC on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(args, world: int, rank: int) -> None:
    """--launch-check: init the process group, one all-gather census, rank 0 prints what the collective saw."""
    import torch.distributed as dist
    from mars5_tts_amd import sharding as sh
    if args.backend == "nccl":
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr % max(torch.cuda.device_count(), 1) if args.same_device else lr)
    if world > 1:
        dist.init_process_group(backend=args.backend)
        census = sh.rank_census()
        seen = sh.LAST_STATS["ranks_seen"]
        shared = None
        if args.check_bundle:
            # the synthetic checkpoint as the N-rank runs get it: built on rank 0, mapped by the others; every rank reports a checksum
            b = shared_bundle(world, rank, size="tiny")
            cs = float(sum(float(v.double().sum()) for v in b.ar_ckpt["model"].values()) + sum(float(v.double().sum()) for v in b.nar_ckpt["model"].values()))
            allcs = [None] * world
            dist.all_gather_object(allcs, cs)
            shared = {"checksums_equal": len(set(allcs)) == 1, "ranks": world, "leftover_file": os.path.exists(f"/dev/shm/m5_bench_bundle_{os.environ.get('MASTER_PORT', '0')}.pt")}
        c4 = None
        if args.workload == "c4":
            # BASELINE configs[3] end to end THROUGH THIS FILE with CPU stand-ins for the three device stages (oracle/fakes.py:
            # test infrastructure; the product's host logic -- prompt assembly from wire ids, hand-off, grouping, prompt skipping --
            # is the real one): requests built on rank 0 -> scatter -> every rank refines its shard in groups -> gather -> rank 0
            # re-computes requests that ANOTHER rank ran and requires identical codes.
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            from fakes import cpu_standin_tts
            from mars5_tts_amd import synth
            m, inf = cpu_standin_tts(synth.make_vocab(30, 63))
            cfg = inf.InferenceConfig(deep_clone=True, temperature=0.7, top_k=100)
            n_total = 3 * world
            reqs = []
            for i in range(n_total):
                text, tr = f"Request number {i} says hello.", "A transcript " * (1 + i % 3)
                ids = m.texttok.encode("<|startoftext|>" + tr + ' ' + text.strip() + "<|endoftext|>", allowed_special='all')
                ref = synth.make_ref_codes(20 + 7 * i, seed=50 + i, merge_friendly=True)
                reqs.append(sh.Request(i, torch.tensor(ids, dtype=torch.long), ref[0].T.contiguous(), seed=500 + i, n_gen_est=12,
                                       n_phones_gen=len(text), max_len=len(ids) + ref.shape[-1] + 12))
            outs = sh.run_sharded(reqs if rank == 0 else None, n_total, None, src=0, batch_worker=request_batch_worker(m, cfg, 2, 2))
            parts = sh.lpt_partition([sh.estimate_cost(r) for r in reqs], world)
            if rank == 0:
                checked = verify_remote(m, cfg, reqs, set(parts[0]), outs, k=2)
                c4 = {"requests": n_total, "requests_per_rank": [len(p) for p in parts], "scatter_bytes": sh.LAST_STATS.get("scatter_bytes"),
                      "gather_bytes": sh.LAST_STATS.get("gather_bytes"), "remote_results_rechecked_equal": checked,
                      "frames": [int(o.shape[0]) for o in outs], "stages": "CPU stand-ins (oracle/fakes.cpu_standin_tts)"}
        dist.barrier()
        dist.destroy_process_group()
    else:
        census, seen, shared, c4 = [dict(rank=0)], 1, None, None
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": seen, "requested": args.gpus, "world_size_env": world, "shared_bundle": shared, "c4": c4,
                          "collective": {"backend": args.backend if world > 1 else None, "ranks_seen": seen, "census": census}}), flush=True)


WORDS = ("the quick brown rat jumped over lazy dogs twice while seven silver foxes watched from behind a quiet river bank and "
         "nobody in town could say why this year demand was never met").split()


def c3_requests(m, n: int, n_gen: int, seed: int = 11):
    """`n` requests, reference length uniform in [2 s, 12 s] (150-900 frames), text and
    transcript of 10-60 tokens together, output length forced to n_gen frames per request."""
    from mars5_tts_amd import synth
    g = torch.Generator().manual_seed(seed)
    texts, trs, refs, max_lens = [], [], [], []
    for i in range(n):
        frames = int(torch.randint(150, 901, (1,), generator=g))
        nw = int(torch.randint(3, 16, (1,), generator=g))
        w0 = int(torch.randint(0, len(WORDS), (1,), generator=g))
        words = [WORDS[(w0 + j) % len(WORDS)] for j in range(2 * nw)]
        texts.append(" ".join(words[:nw]).capitalize() + ".")
        trs.append(" ".join(words[nw:]).capitalize() + ".")
        ref = synth.make_ref_codes(frames, seed=100 + i)
        refs.append(ref)
        tt = m.texttok.encode("<|startoftext|>" + trs[-1] + ' ' + texts[-1].strip() + "<|endoftext|>", allowed_special='all')
        st = m.speechtok.encode(' '.join(str(t) for t in ref[0, 0].tolist()))
        max_lens.append(len(tt) + len(st) + n_gen)
    return texts, trs, refs, max_lens


class NarGroupSpy:
    """Re-check of a batched NAR refinement against LONE calls (the configs[2] legs ran a path -- row-tile lists +
    deferred LayerNorms at the real geometry -- whose equality with the lone path only tests/ asserted).  Wrapped around
    ``inference.perform_batch_inference`` for ONE batch step it remembers, per group, the staged inputs, the (seed, offset) of
    every request's private device generator right before the group consumed it, and the group's results; ``recheck`` then
    refines up to k requests of a group ALONE (``perform_simple_inference``, a generator put back to that seed and offset)
    and requires identical codes.  Outside every timed region."""

    def __init__(self):
        import inference as inf
        self.inf, self.orig, self.groups = inf, inf.perform_batch_inference, []

    def __enter__(self):
        def wrapped(model, batches, diff, T, dsh=None, generators=None, **kw):
            rec = dict(model=model, batches=list(batches), diff=diff, T=T, dsh=dsh, out=None,
                       snap=[(int(g.initial_seed()), int(g.get_offset())) for g in generators])
            self.groups.append(rec)
            r = self.orig(model, batches, diff, T, dsh=dsh, generators=generators, **kw)
            if not callable(r):
                rec["out"] = r
                return r

            def land():
                rec["out"] = r()
                return rec["out"]
            return land
        self.inf.perform_batch_inference = wrapped
        return self

    def __exit__(self, *exc):
        self.inf.perform_batch_inference = self.orig
        return False

    def recheck(self, dev, k: int = 2) -> dict:
        from mars5_tts_amd.diffuser import perform_simple_inference
        checked, rows = 0, []
        grp = max(self.groups, key=lambda g: len(g["batches"]))          # the largest group: the one the row-tile lists matter for
        n = len(grp["batches"])
        lens = [int(b[4].shape[1]) + int(b[1].shape[1]) for b in grp["batches"]]
        order = sorted(range(n), key=lambda i: lens[i])
        pick = [order[0], order[n // 2]][:k] if n > 1 else [0]          # the shortest member (most padding around it) and a middle one
        for i in dict.fromkeys(pick):
            g = torch.Generator(device=dev)
            g.manual_seed(grp["snap"][i][0])
            g.set_offset(grp["snap"][i][1])
            lone = perform_simple_inference(grp["model"], grp["batches"][i], grp["diff"], grp["T"], dsh=grp["dsh"], generator=g)
            same = bool(torch.equal(lone.cpu(), grp["out"][i].cpu()))
            assert same, f"request {i} of a group of {n}: the batched NAR refinement differs from the lone call in {int((lone.cpu() != grp['out'][i].cpu()).sum())} codes"
            checked += 1
            rows.append(lens[i])
        return {"group_results_rechecked_against_lone_calls": True, "requests_rechecked": checked, "group_size": n, "rechecked_total_rows": rows,
                "groups_in_step": len(self.groups)}


def main_c3(args, m, dev, world, rank, barrier):
    """One step = `--batch` mixed-length requests: AR decoded --ar-batch at a time (weights read once per step for
    all of them), NAR refined --nar-batch at a time.  value = generated audio seconds / wall second."""
    from inference import InferenceConfig
    from mars5_tts_amd import ar_engine, nar_engine
    texts, trs, refs, max_lens = c3_requests(m, args.batch, args.n_gen, seed=11 + rank)
    refs = [r.to(dev) for r in refs]
    cfg = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                          eos_estimated_gen_length_factor=100.0, eos_penalty_factor=50.0, eos_penalty_decay=0.5)

    import inference as inf
    split = {"ar_total_s": 0.0, "ar_decode_s": 0.0, "nar_total_s": 0.0, "nar_loop_s": 0.0}
    _ar, _nar = inf.ar_generate_batch, inf.perform_batch_inference

    def ar_timed(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = _ar(*a, **k)
        torch.cuda.synchronize()
        split["ar_total_s"] += time.perf_counter() - t
        split["ar_decode_s"] += ar_engine.LAST_STATS["decode_ms"] / 1e3
        return r

    span = {"t0": None, "t1": None}

    def nar_timed(*a, **k):
        # groups are enqueued without waiting (two in flight on two streams): NAR wall time = first enqueue -> last landing,
        # nar_loop_s = sum of the groups' own step-loop spans (they overlap, so the sum can exceed the wall time)
        if span["t0"] is None:
            torch.cuda.synchronize()
            span["t0"] = time.perf_counter()
        r = _nar(*a, **k)
        if not callable(r):
            span["t1"] = time.perf_counter()
            split["nar_loop_s"] += nar_engine.LAST_STATS["loop_ms"] / 1e3
            return r

        def land():
            out = r()
            span["t1"] = time.perf_counter()
            split["nar_loop_s"] += nar_engine.LAST_STATS["loop_ms"] / 1e3
            return out
        return land

    inf.ar_generate_batch, inf.perform_batch_inference = ar_timed, nar_timed

    def step(i):
        t0 = time.perf_counter()
        out = m.tts_batch_from_codes(texts, refs, trs, cfg, seeds=[1000 + rank * 10007 + i * args.batch + j for j in range(args.batch)],
                                     nar_batch=args.nar_batch, ar_batch=args.ar_batch, max_lens=max_lens, nar_in_flight=args.nar_in_flight)
        torch.cuda.synchronize()
        if span["t0"] is not None:
            split["nar_total_s"] += span["t1"] - span["t0"]
            span["t0"] = span["t1"] = None
        return time.perf_counter() - t0, sum(int(f.shape[0]) for _, f in out)

    for i in range(args.warmup):
        step(-1 - i)
    for k in split:
        split[k] = 0.0
    barrier()
    t0 = time.perf_counter()
    lat, frames = [], 0
    for i in range(args.steps):
        dt, nf = step(i)
        lat.append(dt)
        frames += nf
    barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = world
    if world > 1:
        from mars5_tts_amd import sharding as sh
        elapsed, frames = sh.reduce_timing(elapsed, float(frames))
        sh.rank_census()
        ranks_seen = sh.LAST_STATS["ranks_seen"]
    if rank != 0:
        return
    # one more (untimed) step under the spy: two of its group results against lone seeded calls
    inf.ar_generate_batch, inf.perform_batch_inference = _ar, _nar
    with NarGroupSpy() as spy:
        m.tts_batch_from_codes(texts, refs, trs, cfg, seeds=[77000 + j for j in range(args.batch)], nar_batch=args.nar_batch, ar_batch=args.ar_batch,
                               max_lens=max_lens, nar_in_flight=args.nar_in_flight)
        torch.cuda.synchronize()
    recheck = spy.recheck(dev, k=2)
    ref_frames = [int(r.shape[-1]) for r in refs]
    out = {
        "metric": "generated audio seconds/sec (RTF), deep-clone", "value": round(frames / 75.0 / elapsed, 4), "unit": "audio_s/s",
        "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "p50_batch_latency_s": round(statistics.median(lat), 3),
        "config": {"workload": f"BASELINE configs[2]: batch of {args.batch} mixed-length requests per GPU, deep-clone, temperature=0.7 "
                               f"top_k=100, reference 150-900 frames (2-12 s), text+transcript 10-60 tokens, {args.n_gen} generated frames "
                               "each, 200 DDPM steps x CFG, seeded random weights of the real geometry",
                   "requests_per_step": args.batch, "ar_batch": args.ar_batch, "nar_batch": args.nar_batch, "nar_groups_in_flight": args.nar_in_flight,
                   "reference_frames_min_mean_max": [min(ref_frames), round(sum(ref_frames) / len(ref_frames), 1), max(ref_frames)],
                   "parallelism": f"replicas x{world} (requests sharded by rank, no data-path collective)"},
        "time_split_s_per_step": {k: round(v / args.steps, 3) for k, v in split.items()},
        "last_ar_batch": {k: ar_engine.LAST_STATS.get(k) for k in ("decode_ms", "decode_steps_launched", "batch")},
        "last_nar_batch": {k: nar_engine.LAST_STATS.get(k) for k in ("loop_ms", "steps", "batch", "rows")},
    }
    out.update(recheck)
    print(json.dumps(out), flush=True)


def request_worker(m, base_cfg):
    """The per-request hot path as the sharded runs call it: a wire-format request in, final (G, 8) codes out."""
    import dataclasses

    def work(r):
        cfg = dataclasses.replace(base_cfg, generate_max_len_override=int(r.max_len))
        torch.manual_seed(int(r.seed))
        _, final = m.tts_from_ids(r.text_ids, r.ref_codes.T.contiguous()[None].to(m.device), int(r.n_phones_gen), cfg)
        return final.cpu()
    return work


def request_batch_worker(m, base_cfg, nar_batch: int, nar_in_flight: int):
    """A rank's whole shard through ``Mars5TTS.tts_batch_from_ids``: AR request by request (bit-exact batch-1 decode), NAR in
    groups -- result i is bit-identical to ``request_worker`` on request i (bench.verify_remote checks exactly that)."""
    def work(shard):
        if not shard:
            return []
        outs = m.tts_batch_from_ids([r.text_ids for r in shard], [r.ref_codes.T.contiguous()[None].to(m.device) for r in shard],
                                    [int(r.n_phones_gen) for r in shard], base_cfg, seeds=[int(r.seed) for r in shard], nar_batch=nar_batch,
                                    ar_batch=1, max_lens=[int(r.max_len) for r in shard], nar_in_flight=nar_in_flight)
        return [final.cpu() for _, final in outs]
    return work


def c4_requests(m, n: int, n_gen: int, factor: float, seed: int = 11):
    """SURVEY 8(d) config 4 = config 3's request generator, in the wire format of ``sharding.Request``."""
    from mars5_tts_amd.sharding import Request
    texts, trs, refs, max_lens = c3_requests(m, n, n_gen, seed=seed)
    reqs = []
    for i in range(n):
        ids = m.texttok.encode("<|startoftext|>" + trs[i] + ' ' + texts[i].strip() + "<|endoftext|>", allowed_special='all')
        reqs.append(Request(idx=i, text_ids=torch.tensor(ids, dtype=torch.long), ref_codes=refs[i][0].T.contiguous().cpu(), seed=1000 + i,
                            n_gen_est=n_gen, n_phones_gen=round(factor * len(texts[i])), max_len=max_lens[i]))
    return reqs


def verify_remote(m, base_cfg, reqs, mine_idx, gathered, k=2):
    """rank 0 after a gather: recompute up to `k` requests that ran on OTHER ranks (same per-request seed) and require the
    gathered codes to be identical -- results do not depend on which GPU produced them."""
    work = request_worker(m, base_cfg)
    checked = 0
    for r in reqs:
        if r.idx in mine_idx:
            continue
        ref = work(r)
        got = gathered[r.idx]
        assert got.shape == ref.shape and torch.equal(got, ref), f"request {r.idx}: gathered codes differ from a local seed-matched run"
        checked += 1
        if checked >= k:
            break
    return checked


def main_c4(args, m, dev, world, rank, barrier):
    """One step = args.batch * world requests on rank 0 -> scatter (LPT by estimated cost) -> every rank runs its shard
    through the batch-1 hot path -> gather on rank 0.  The timed region spans scatter + compute + gather."""
    import torch.distributed as dist
    from mars5_tts_amd import sharding as sh
    from inference import InferenceConfig
    cfg = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                          eos_estimated_gen_length_factor=100.0, eos_penalty_factor=50.0, eos_penalty_decay=0.5)
    n_total = args.batch * world
    reqs = c4_requests(m, n_total, args.n_gen, cfg.eos_estimated_gen_length_factor, seed=11)      # every rank can rebuild them (verification)
    work = request_worker(m, cfg)
    bwork = None if args.c4_per_request else request_batch_worker(m, cfg, args.nar_batch, args.nar_in_flight)

    coll = world > 1 or bool(getattr(args, "coll", False))       # --collective-at-1: one rank, the collectives still run

    def step():
        if coll:
            return sh.run_sharded(reqs if rank == 0 else None, n_total, work, src=0, batch_worker=bwork)
        if bwork is not None:
            return bwork(reqs)
        return [work(r) for r in reqs]

    if args.warmup:
        work(reqs[0])
    barrier()
    t0 = time.perf_counter()
    outs = None
    for _ in range(args.steps):
        outs = step()
    barrier()
    elapsed = time.perf_counter() - t0
    frames = float(sum(int(o.shape[0]) for o in outs)) * args.steps if rank == 0 else 0.0
    collective = None
    checked_local = None
    if world == 1 and bwork is not None and not coll:          # one rank: the group results against lone seeded calls of two requests
        checked_local = verify_remote(m, cfg, reqs, set(), outs, k=2)
    if coll:
        elapsed, frames = sh.reduce_timing(elapsed, frames)
        census = sh.rank_census()
        parts = sh.lpt_partition([sh.estimate_cost(r) for r in reqs], world)
        if rank == 0:
            # (one rank: nothing ran elsewhere -- re-run two of its own requests alone instead)
            checked = verify_remote(m, cfg, reqs, set(parts[0]) if world > 1 else set(), outs, k=2)
            collective = dict(backend=sh.LAST_STATS.get("backend"), ranks_seen=sh.LAST_STATS.get("ranks_seen"), census=census,
                              scatter_bytes=sh.LAST_STATS.get("scatter_bytes"), gather_bytes=sh.LAST_STATS.get("gather_bytes"),
                              remote_results_rechecked_equal=checked, requests_per_rank=[len(p) for p in parts])
    if rank != 0:
        return
    ref_frames = [int(r.ref_codes.shape[0]) for r in reqs]
    out = {
        "metric": "generated audio seconds/sec (RTF), deep-clone", "value": round(frames / 75.0 / elapsed, 4), "unit": "audio_s/s",
        "n_gpus": (collective["ranks_seen"] if collective else world), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {n_total} mixed-length deep-clone requests ({args.batch} per GPU) built on rank 0, "
                               f"scattered over {world} rank(s) by estimated cost, "
                               + ("batch-1 hot path per request" if args.c4_per_request else f"each rank: batch-1 AR decode per request, NAR in groups of {args.nar_batch} (tts_batch_from_ids)")
                               + ", results gathered on rank 0; "
                               f"reference 150-900 frames, {args.n_gen} generated frames each, temperature=0.7 top_k=100, 200 DDPM steps x CFG",
                   "requests_per_step": n_total, "reference_frames_min_mean_max": [min(ref_frames), round(sum(ref_frames) / len(ref_frames), 1), max(ref_frames)],
                   "parallelism": f"replicas x{world}, request scatter + result gather over {('RCCL' if args.backend == 'nccl' else args.backend) if coll else 'nothing (single rank)'}"},
        "collective": collective, "group_results_rechecked_against_lone_calls": checked_local,
    }
    print(json.dumps(out), flush=True)


def shared_bundle(world: int, rank: int, size: str = "full"):
    """The seeded synthetic checkpoint of the real geometry (1.2 B parameters, ~11 s of host time to synthesise).  With N ranks
    on one node rank 0 builds it ONCE and the others map it from /dev/shm (eight concurrent builds would fight over the host
    cores inside the driver's timeout); the file is keyed by MASTER_PORT (one per job) and removed by rank 0 afterwards."""
    from mars5_tts_amd import synth
    if world <= 1:
        return synth.make_bundle(size, seed=0)
    import torch.distributed as dist
    path = f"/dev/shm/m5_bench_bundle_{os.environ.get('MASTER_PORT', '0')}.pt"
    ok = [True]
    if rank == 0:
        b = synth.make_bundle(size, seed=0)
        try:
            torch.save(b, path + ".tmp")
            os.replace(path + ".tmp", path)
        except Exception as e:          # noqa: BLE001 -- e.g. a container with a 64-MB /dev/shm: every rank then builds its own copy
            print(f"bench.py: could not share the synthetic checkpoint through {path} ({type(e).__name__}: {e}); every rank builds its own", file=sys.stderr, flush=True)
            ok[0] = False
    dist.broadcast_object_list(ok, src=0)
    if rank != 0:
        b = torch.load(path, map_location="cpu", weights_only=False, mmap=True) if ok[0] else synth.make_bundle(size, seed=0)
    dist.barrier()
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return b


def build_model(dtype_name: str, dev, world: int = 1, rank: int = 0):
    from inference import Mars5TTS
    from mars5_tts_amd.ops import DT_NAME
    bundle = shared_bundle(world, rank)
    m = Mars5TTS(bundle.ar_ckpt, bundle.nar_ckpt, device=str(dev), codec=False, vocos=False)
    m.codeclm.set_engine_dtype(DT_NAME[dtype_name])
    m.codecnar.set_engine_dtype(DT_NAME[dtype_name])
    m.codeclm.engine()
    m.codecnar.engine()
    return m, bundle


def make_cfg(n_text_tokens_hint: int, p_len: int, n_gen: int):
    from inference import InferenceConfig
    # README sampling settings (reference README.md:122-123); output length is forced through
    # generate_max_len_override and an EOS penalty that stays on for the whole utterance
    # (random weights would otherwise emit <|endofspeech|> at random).
    return InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                           generate_max_len_override=p_len + n_gen, eos_estimated_gen_length_factor=100.0,
                           eos_penalty_factor=50.0, eos_penalty_decay=0.5)


def prompt_len(m, ref_codes, deep=True):
    tt = m.texttok.encode("<|startoftext|>" + TRANSCRIPT + ' ' + TEXT.strip() + "<|endoftext|>", allowed_special='all')
    st = m.speechtok.encode(' '.join(str(t) for t in ref_codes[0, 0].tolist()))
    return len(tt) + len(st), len(tt)


STEP_LOG = []       # per timed utterance: wall, the AR decode / NAR loop spans inside it, persistent-step recoveries


def run_utterance(m, ref_codes, cfg, seed):
    from mars5_tts_amd import ar_engine, nar_engine
    torch.manual_seed(seed)
    t0 = time.perf_counter()
    gen, final = m.tts_from_codes(TEXT, ref_codes, TRANSCRIPT, cfg)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    STEP_LOG.append((round(dt, 4), round(ar_engine.LAST_STATS.get("decode_ms", 0.0), 1), round(nar_engine.LAST_STATS.get("loop_ms", 0.0), 1),
                     int(ar_engine.LAST_STATS.get("persistent_recoveries", 0) or 0)))
    return dt, int(final.shape[0]), int(gen.shape[0])


# ------------------------------------------------------------------------------------ roofline
def roofline_leg(m, ref_codes, cfg, dtype_name):
    """Per-kernel durations of the NAR reverse step at the bench shapes, measured INSIDE a replayed hipGraph of the step's
    forward (clock-stamp launches between the launches: see below); launches are grouped the way rocprofv3 --stats groups
    them (by kernel name: every shape of one GEMM epilogue is one kernel) and the kernel with the largest total is reported
    against its roofline, per shape under `shapes` (profiles/ holds the rocprofv3 summary of the same command).  Plus the
    event-timed AR decode graph replays of the last utterance."""
    from mars5_tts_amd import ar_engine, nar_engine, ops
    from mars5_tts_amd import _lib as L
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    eng = m.codecnar.engine()
    ns = dict(nar_engine.LAST_STATS)
    ars = dict(ar_engine.LAST_STATS)
    S, s_out, Le = ns["S"], ns["s_out"], ns["Le"]
    off = S - s_out
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 1025, (S, 8), generator=g)
    c_text = torch.randint(0, eng.shape.n_text_vocab, (Le - 1,), generator=g)
    sess = NARSession(eng, NARConfig(T=200))
    z = torch.zeros(S, 8, dtype=torch.long)
    mm = torch.zeros(S, 8, dtype=torch.uint8)
    mm[:, 0] = 1
    mm[:off] = 1
    sess.prepare(c_text, ref_codes[0].T.contiguous(), x, z, mm, off, list(range(199, 199 - 24, -1)))
    st = sess.stream.cuda_stream
    es = 2 if dtype_name != "f32" else 4
    names_epi = {L.EPI_F32: "F32", L.EPI_RESIDUAL: "RESIDUAL", L.EPI_SWIGLU: "SWIGLU", L.EPI_QKV: "QKV", L.EPI_DT: "DT", L.EPI_SILU_DT: "SILU"}
    orig = {k: getattr(ops, k) for k in ("gemm", "gemm_dln", "attention", "layernorm", "layernorm_mean", "layernorm_twice", "xattn_scores", "xattn_scores_dln",
                                         "xattn_absorb", "chunked_embed")}

    slots = torch.zeros(2048, dtype=torch.int64, device=m.device)
    labels = []

    def timed(fn, label, flops, hbm_bytes):
        def w(*a, **kw):
            ops.clock_stamp(slots, len(labels), stream=st)
            labels.append((label(*a, **kw), flops(*a, **kw), hbm_bytes(*a, **kw)))
            fn(*a, **kw)
        return w

    def gemm_label(a, w_, out, epi, **kw):
        Mv = (kw.get("M") or a.shape[-2]) * kw.get("batch", 1)
        return f"gemm16_kernel<EPI_{names_epi.get(epi, epi)}> M={Mv} N={w_.shape[-2]} K={w_.shape[-1]}"

    def gemm_flops(a, w_, out, epi, **kw):
        return 2.0 * (kw.get("M") or a.shape[-2]) * kw.get("batch", 1) * w_.shape[-2] * w_.shape[-1]

    def gemm_bytes(a, w_, out, epi, **kw):
        Mv = (kw.get("M") or a.shape[-2]) * kw.get("batch", 1)
        N, K = w_.shape[-2], w_.shape[-1]
        c = Mv * N * (8 if epi == L.EPI_RESIDUAL else (4 if epi == L.EPI_F32 else es)) // (2 if epi == L.EPI_SWIGLU else 1)
        return float(Mv * K * es + N * K * es * kw.get("batch", 1) + c)

    orig["mark"] = ops.mark
    ops.gemm = timed(orig["gemm"], gemm_label, gemm_flops, gemm_bytes)
    # deferred-LayerNorm forms (same kernel names as rocprofv3 prints them: the DLN template argument is part of the instantiation, the
    # epilogue class is what groups them here)
    ops.gemm_dln = timed(orig["gemm_dln"], lambda a, w_, out, epi, dl, **kw: gemm_label(a, w_, out, epi, **kw) + ((" +dln-producer" if dl.mode == 1 else " +dln-consumer") if dl is not None else " +row-tiles"),
                         lambda a, w_, out, epi, dl, **kw: gemm_flops(a, w_, out, epi, **kw),
                         lambda a, w_, out, epi, dl, **kw: gemm_bytes(a, w_, out, epi, **kw) + (float((kw.get("M") or a.shape[-2]) * kw.get("batch", 1) * w_.shape[-2] * es) if (dl is not None and dl.mode == 1) else 0.0))
    ops.layernorm_mean = timed(orig["layernorm_mean"], lambda x_, g_, b_, eps, out, mo, **kw: f"layernorm_vec_kernel D={x_.shape[-1]} affine=1 +mean",
                               lambda *a, **kw: 0.0,
                               lambda x_, g_, b_, eps, out, mo, **kw: float((kw.get("M") or x_.shape[0]) * x_.shape[-1] * (4 + out.element_size())))
    ops.layernorm_twice = timed(orig["layernorm_twice"], lambda x_, g_, b_, eps, eps2, out, rps, **kw: f"layernorm_twice_vec_kernel D={x_.shape[-1]}",
                                lambda *a, **kw: 0.0,
                                lambda x_, g_, b_, eps, eps2, out, rps, **kw: float(rps * kw.get("n_seq", 1) * x_.shape[-1] * (4 + out.element_size())))
    ops.xattn_scores_dln = timed(orig["xattn_scores_dln"], lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, dl, **kw: f"gemm16_kernel<EPI_SOFTMAX_HEADS> M={M * batch} N={H * Lp} K={x_.shape[-1]}" + (" +dln-consumer" if dl is not None else " +row-tiles"),
                                 lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, dl, **kw: 2.0 * M * batch * H * Lp * x_.shape[-1],
                                 lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, dl, **kw: float(M * batch * (x_.shape[-1] + H * Lp) * es + batch * H * Lp * x_.shape[-1] * es))
    ops.attention = timed(orig["attention"], lambda dt, a_, **kw: f"attn16_kernel Sq={a_.Sq} Sk={a_.Sk}",
                          lambda dt, a_, **kw: 4.0 * a_.B * a_.H * a_.Sq * a_.Sk * 64,
                          lambda dt, a_, **kw: float(a_.B * a_.H * (2 * a_.Sq + 2 * a_.Sk) * 64 * es))
    ops.layernorm = timed(orig["layernorm"], lambda x_, g_, b_, eps, out, **kw: f"layernorm_vec_kernel D={x_.shape[-1]} affine={kw.get('n_affine', 1)}",
                          lambda *a, **kw: 0.0,
                          lambda x_, g_, b_, eps, out, **kw: float((kw.get("M") or x_.shape[0]) * x_.shape[-1] * (4 + out.element_size() * kw.get("n_affine", 1))))
    ops.xattn_scores = timed(orig["xattn_scores"], lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, **kw: f"gemm16_kernel<EPI_SOFTMAX_HEADS> M={M * batch} N={H * Lp} K={x_.shape[-1]}",
                             lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, **kw: 2.0 * M * batch * H * Lp * x_.shape[-1],
                             lambda x_, sX, a_tab, c_tab, p_out, sP, M, H, Lp, batch, **kw: float(M * batch * (x_.shape[-1] + H * Lp) * es + batch * H * Lp * x_.shape[-1] * es))
    ops.xattn_absorb = timed(orig["xattn_absorb"], lambda dt, ts, tl, nl, nsq, H, D, Lp, step, scale, **kw: f"absorb_kernel layers={nl} seq={nsq} Lp={Lp}",
                             lambda dt, ts, tl, nl, nsq, H, D, Lp, step, scale, **kw: 2.0 * 2 * nl * nsq * H * Lp * D * 64,
                             lambda dt, ts, tl, nl, nsq, H, D, Lp, step, scale, **kw: float(nl * (2 * D * D * es + nsq * 2 * H * Lp * D * es)))
    ops.chunked_embed = timed(orig["chunked_embed"], lambda out, *a, **kw: "chunked_embed_kernel", lambda *a, **kw: 0.0,
                              lambda out, *a, **kw: float(out.numel() * 4 * 2))

    def mark(label, stream=None):
        ops.clock_stamp(slots, len(labels), stream=st)
        labels.append((label, 0.0, 0.0))
    ops.mark = mark
    # In-graph timing.  The forward of one reverse step is captured with a one-lane clock-stamp launch (100 MHz wall clock ->
    # device memory, m5_clock_stamp) in front of every launch: in the replay, stamp i+1 runs when launch i has drained, so
    # t[i+1] - t[i] - (the same difference with nothing in between) is what launch i occupies in the replay, boundary
    # included.  (HIP events cannot be read back when recorded during a capture on this stack, and an eager event pair times
    # the launch on an otherwise idle GPU at higher clocks: 12 % rosier than rocprofv3 of the graph, round 2.)
    n_rep = 8
    capturing = False
    try:
        ops.Graph.begin(st)
        capturing = True
        sess.enqueue_forward(st)
        ops.clock_stamp(slots, len(labels), stream=st)
        capturing = False
        g_fwd = ops.Graph().end(st)
    finally:
        for k, v in orig.items():
            setattr(ops, k, v)
        if capturing:                        # an enqueue raised mid-capture: close the capture, or the stream stays unusable for
            try:                             # every leg after this one
                ops.Graph().end(st)
            except Exception:                # noqa: BLE001
                pass
    n_l = len(labels)
    cal_slots = torch.zeros(64, dtype=torch.int64, device=m.device)
    ops.Graph.begin(st)
    for i in range(64):
        ops.clock_stamp(cal_slots, i, stream=st)
    g_cal = ops.Graph().end(st)
    ops.Graph.begin(st)                      # the same forward without stamps: what the stamps add, and the whole-step time
    sess.enqueue_forward(st)
    g_plain = ops.Graph().end(st)
    g_cal.launch(st)
    g_cal.launch(st)
    sess.stream.synchronize()
    cd = (cal_slots[1:] - cal_slots[:-1]).cpu().tolist()
    stamp_us = sorted(cd)[len(cd) // 2] * 0.01
    acc_us = [0.0] * n_l
    n_clamped = 0
    for r in range(n_rep + 2):
        g_fwd.launch(st)
        ops.add_int(sess.step_ptr, 1, stream=st)
        sess.stream.synchronize()
        if r >= 2:
            t = slots[: n_l + 1].cpu().tolist()
            for i in range(n_l):
                d_us = (t[i + 1] - t[i]) * 0.01 - stamp_us
                n_clamped += d_us < 0.01
                acc_us[i] += max(d_us, 0.01)
    e0, e1 = ops.Event(), ops.Event()
    g_plain.launch(st)
    e0.record(st)
    for r in range(n_rep):
        g_plain.launch(st)
    e1.record(st)
    sess.stream.synchronize()
    fwd_plain_us = 1e3 * e0.elapsed_ms(e1) / n_rep
    per = [(lab, fl, by, 1e-3 * acc_us[i] / n_rep) for i, (lab, fl, by) in enumerate(labels)]
    timing = (f"in-graph: clock-stamp launches between the launches of the captured step (one serial chain, like the product graph "
              f"forward_graph_replay_us), {n_rep} replays; stamp-to-stamp overhead {stamp_us:.2f} us subtracted per interval")
    agg, cls = {}, {}
    for lab, fl, by, ms in per:
        for d, key in ((agg, lab), (cls, lab.split(" M=")[0].split(" Sq=")[0].split(" D=")[0].split(" layers=")[0])):
            a = d.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
            a["ms"] += ms
            a["flops"] += fl
            a["bytes"] += by
            a["n"] += 1
    kernels = {k: dict(launches_per_step=v["n"], avg_us=round(1e3 * v["ms"] / v["n"], 2), tflops=round(v["flops"] / v["ms"] / 1e9, 1),
                       alg_gbs=round(v["bytes"] / v["ms"] / 1e6, 1), step_share_us=round(1e3 * v["ms"], 1)) for k, v in agg.items()}
    peak = PEAK_MFMA_TFLOPS[dtype_name]
    # the dominant kernel = the kernel NAME with the largest total, as rocprofv3 --stats groups them (all shapes of one epilogue)
    dom = max(cls.items(), key=lambda kv: kv[1]["ms"])
    mfma_bound = dom[1]["flops"] > 0
    traffic, traffic_source = None, None
    try:        # HBM bytes per launch from the committed PMC passes (profiles/traffic.json), same shapes
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for k, v in tj.items():
            if isinstance(v, dict) and v.get("class") == dom[0] and dtype_name == "bf16":
                traffic = v["bytes_per_launch"]
                # counters cannot be read inside this run (rocprofv3 --pmc is its own process): the value is the committed PMC
                # pass over this kernel class, and the line says so
                traffic_source = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.sh): " + str(v.get("measured", ""))[:160]
    except Exception:
        traffic, traffic_source = None, None
    # Which roof bounds the class: its arithmetic intensity (algorithmic flops / algorithmic bytes, summed over the class's launches)
    # against the machine balance peak_flops / peak_bytes.  The residual-epilogue class reads and rewrites an fp32 C tile set (8 bytes
    # per output element, + 2 for the deferred LayerNorm's centred copy): K = 768 / 1024 launches sit at 125-160 flop/B, under the
    # balance of 312 flop/B -- HBM-bound by the roofline model.  Both fractions are reported.
    ai = dom[1]["flops"] / max(dom[1]["bytes"], 1.0)
    balance = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    tf_ach = dom[1]["flops"] / dom[1]["ms"] / 1e9
    gb_ach = dom[1]["bytes"] / dom[1]["ms"] / 1e6
    if mfma_bound and ai >= balance:
        roof = dict(bound="mfma", kernel=dom[0], achieved=round(tf_ach, 1), peak=peak, unit="TFLOP/s", frac=round(tf_ach / peak, 4))
    else:
        roof = dict(bound="hbm", kernel=dom[0], achieved=round(gb_ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gb_ach / PEAK_HBM_GBS, 4))
    roof.update(arithmetic_intensity_flop_per_byte=round(ai, 1), machine_balance_flop_per_byte=round(balance, 1),
                frac_of_mfma_peak=round(tf_ach / peak, 4), frac_of_hbm_peak=round(gb_ach / PEAK_HBM_GBS, 4),
                alg_bytes_per_launch=dom[1]["bytes"] / dom[1]["n"], alg_flops_per_launch=dom[1]["flops"] / dom[1]["n"])
    mfma_bound = roof["bound"] == "mfma"
    sum_us = 1e3 * sum(ms for _, _, _, ms in per)
    fwd_flops = sum(fl for _, fl, _, _ in per)
    roof.update(traffic=traffic, traffic_source=traffic_source, avg_launch_us=round(1e3 * dom[1]["ms"] / dom[1]["n"], 2), launches_per_step=dom[1]["n"], timing=timing,
                alg_per_launch=(dom[1]["flops"] if mfma_bound else dom[1]["bytes"]) / dom[1]["n"],
                shapes={k: v["avg_us"] for k, v in kernels.items() if k.startswith(dom[0])},
                forward_sum_of_intervals_us=round(sum_us, 1), forward_graph_replay_us=round(fwd_plain_us, 1),
                whole_step_frac=round(fwd_flops / fwd_plain_us / 1e6 / peak, 4), intervals_clamped_to_10ns=int(n_clamped))
    # NAR loop as a whole (graph replay + RNG + sample kernel), from the last timed utterance
    step_ms = ns["loop_ms"] / ns["steps"]
    nar = dict(ms_per_step=round(step_ms, 3), tflops=round(eng.flops_per_step(S, Le, s_out) / step_ms / 1e9, 1), S=S, Le=Le,
               frac_of_mfma_peak=round(eng.flops_per_step(S, Le, s_out) / step_ms / 1e9 / peak, 4))
    # AR decode: algorithmic bytes per token (weights once + KV read/write) / event-timed step
    ae = m.codeclm.engine()
    kv_per_pos = ae.shape.n_layers * ae.shape.nhead * 64 * 2 * es
    W = ae.shape.sliding_window                 # cached positions actually read per step: min(length, window)
    lens = range(int(ars["prefill_len"]), int(ars["final_len"]) + 1)
    avg_len = sum(min(t, W) for t in lens) / max(len(lens), 1)
    bytes_tok = ae.weight_bytes_per_token() + kv_per_pos * (avg_len + 1)
    tok_ms = ars["decode_ms"] / max(ars["n_generated"] - 1, 1)
    gbs = bytes_tok / tok_ms / 1e6
    form = ("3 launches: ar_mega_kernel (26 layers, 256 persistent workgroups, tagged-granule edges, LDS-DMA weight prefetch) + head + sampler, hipGraph"
            if ars.get("persistent") else
            "132 launches: 26 x {gemv qkv+rope, attn_decode, gemv wo, gemv w13+swiglu, gemv w2} + head + sampler, hipGraph")
    ar_traffic = None
    try:        # HBM-side bytes of one persistent step from the committed PMC passes (profiles/traffic.json; bf16, prompt ~490)
        if ars.get("persistent") and dtype_name == "bf16":
            ar_traffic = next(v["bytes_per_launch"] for v in tj.values() if isinstance(v, dict) and v.get("label") == "ar_mega_kernel")
    except Exception:
        ar_traffic = None
    ar = dict(bound="hbm", kernel=f"AR decode step ({form})", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", traffic=ar_traffic,
              frac=round(gbs / PEAK_HBM_GBS, 4), bytes_per_token=int(bytes_tok), us_per_token=round(1e3 * tok_ms, 1))
    return roof, ar, nar, kernels


# ------------------------------------------------------------------------------------ parity of the timed configuration
def parity_leg(m, bundle, ref_codes, dtype_name, n_ar=48):
    """The oracle as CHECKER of the configuration that was just timed (never inside the timed region): the engine's
    greedy AR tokens for the bench prompt, teacher-forced through the oracle on the GPU (reference autocast rounding in
    the bench dtype), and one NAR decoder pass + reverse step at the bench shape against the fp32 oracle.  The full
    450-step / both-dtype version with asserted tolerances is tests/test_gpu_parity16.py."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mars5_oracle as O
    from mars5_tts_amd import _lib as L
    from mars5_tts_amd.ar_engine import ARSamplingConfig, ARSession
    from mars5_tts_amd.nar_engine import NARConfig, NARSession
    from mars5_tts_amd.ops import DT_NAME
    dt = DT_NAME[dtype_name]
    odt = None if dtype_name == "f32" else dt
    dev = m.device
    eng = m.codeclm.engine()
    a, n = bundle.ar_shape, bundle.nar_shape
    tt = m.texttok.encode("<|startoftext|>" + TRANSCRIPT + ' ' + TEXT.strip() + "<|endoftext|>", allowed_special='all')
    sp = m.speechtok.encode(' '.join(str(t) for t in ref_codes[0, 0].tolist()))
    n_text = len(m.texttok.vocab)
    prompt = torch.tensor(tt + [s_ + n_text for s_ in sp], dtype=torch.long)
    ref = ref_codes[0].T.contiguous()
    P, V = int(prompt.shape[0]), a.n_vocab
    eos_sp = m.speechtok.special_tokens["<|endofspeech|>"]
    kw = dict(temperature=0.7, topk=1, top_p=0.2, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100, eos_penalty_factor=50.0,
              eos_penalty_decay=0.5, n_phones_gen=100 * len(TEXT))
    se = ARSession(eng, P + n_ar)
    se.configure_sampler(ARSamplingConfig(**kw), n_text, n_text + eos_sp, torch.ones(n_ar, V, device=dev))
    se.prefill(prompt, ref)
    sv = se.stream.cuda_stream
    logits = []
    for i in range(n_ar):
        if i:
            se.enqueue_layers(sv)
        se.enqueue_head_and_sample(sv)
        se.stream.synchronize()
        logits.append(se.logits.clone())
    toks = se.tokens[: int(se.state.cpu()[L.ST_NTOK])].clone()
    n_gen = int(toks.shape[0]) - P
    out = {"dtype": dtype_name, "oracle_device": str(dev)}
    with torch.device(dev), torch.inference_mode():
        sd = {k: v.to(dev) for k, v in bundle.ar_ckpt["model"].items()}
        if odt is not None:
            sd = O.round_linear_weights(sd, odt)
        p = O.ARSamplingParams(temperature=0.7, top_k=1, top_p=0.2, alpha_frequency=3.0, alpha_presence=0.4, penalty_window=100,
                               eos_penalty_factor=50.0, eos_penalty_decay=0.5, n_phones_gen=100 * len(TEXT))
        _, lo, choices = O.ar_generate_oracle(sd, a.nhead, n_text, bundle.n_speech, eos_sp, prompt.to(dev), ref.to(dev), P + n_ar, p,
                                              noise=torch.ones(n_ar, V), forced=toks, dt=odt)
        err = [float((logits[i] - lo[i]).abs().max()) for i in range(n_gen)]
        flips = [i for i in range(n_gen) if choices[i] != int(toks[P + i])]
        out.update(ar_steps=n_gen, ar_max_abs_dlogit=round(max(err), 5), ar_max_abs_logit=round(max(float(l.abs().max()) for l in lo), 2),
                   ar_greedy_agreement=round((n_gen - len(flips)) / n_gen, 4), ar_first_divergence=flips[0] if flips else None)
        # ---- the batched decode step (configs[2]): 4 sequences with prompts of 488 / 300 / 150 / 64 tokens advance 12 steps together;
        # every sequence's logits against the oracle teacher-forced on that sequence's own tokens
        from mars5_tts_amd.ar_engine import ARBatchSession
        Ps = [P, 300, 150, 64]
        prompts = [prompt[:q].clone() for q in Ps]
        nb = 12
        bs = ARBatchSession(eng, [q + nb for q in Ps])
        bs.configure_sampler(ARSamplingConfig(**kw), n_text, n_text + eos_sp, torch.ones(len(Ps), nb, V, device=dev))
        bs.prefill(prompts, [ref] * len(Ps))
        bl = []
        for i in range(nb):
            if i:
                bs.enqueue_layers(bs.stream.cuda_stream)
            bs.enqueue_head_and_sample(bs.stream.cuda_stream)
            bs.stream.synchronize()
            bl.append(bs.logits.clone())
        bstate = bs.state.cpu()
        worst_b = 0.0
        for q in range(len(Ps)):
            tq = bs.tokens[q, : int(bstate[q, L.ST_NTOK])].clone()
            _, lob, _ = O.ar_generate_oracle(sd, a.nhead, n_text, bundle.n_speech, eos_sp, prompts[q].to(dev), ref.to(dev), Ps[q] + nb, p,
                                             noise=torch.ones(nb, V), forced=tq, dt=odt)
            for i in range(min(int(tq.shape[0]) - Ps[q], len(lob), nb)):
                worst_b = max(worst_b, float((bl[i][q] - lob[i]).abs().max()))
        out.update(ar_batch_sequences=len(Ps), ar_batch_steps=nb, ar_batch_max_abs_dlogit=round(worst_b, 5))
        del sd, lo, bs
        # ---- NAR: one decoder pass and one reverse step at the bench shape
        nengine = m.codecnar.engine()
        g = torch.Generator().manual_seed(3)
        S, off, t = 2 * ref.shape[0] - 1 + 450, 2 * ref.shape[0] - 1, 100
        K = n.n_quant
        c_text = torch.tensor(tt, dtype=torch.long)
        x = torch.randint(0, 1024, (S, 8), generator=g, device="cpu")
        x_known = torch.zeros(S, 8, dtype=torch.long, device="cpu")
        mk = torch.zeros(S, 8, dtype=torch.uint8, device="cpu")
        mk[:, 0] = 1
        mk[:off] = 1
        x_known[:off] = x[:off]
        x_known[:, 0] = x[:, 0]
        gg = torch.Generator(device=dev).manual_seed(11)
        u1 = torch.rand((1, S, 8, K), generator=gg, device=dev)
        u2 = torch.rand((1, S, 8, K), generator=gg, device=dev)
        draws = iter([u1, u2])
        sess = NARSession(nengine, NARConfig(T=200, x_0_temp=0.7, guidance_w=3.0, deep_clone=True, q0_override_steps=20))
        sess.prepare(c_text, ref, x, x_known, mk, off, [t])
        sess.step(lambda shp: next(draws), use_graph=True)
        sess.stream.synchronize()
        so = S - off
        lg = sess.logits[:, :, :K]
        sdn = {k: v.to(dev) for k, v in bundle.nar_ckpt["model"].items()}
        if odt is not None:
            sdn = O.round_linear_weights(sdn, odt)
        lc = O.nar_forward(sdn, n.nhead, c_text.to(dev), ref.to(dev), x.to(dev), t, False)
        lu = O.nar_forward(sdn, n.nhead, c_text.to(dev), ref.to(dev), x.to(dev), t, True)
        zmax = float(torch.maximum(lc.abs().max(), lu.abs().max()))
        e_rel = max(float((lg[:so] - lc[off:, 1:]).abs().max()), float((lg[so:] - lu[off:, 1:]).abs().max())) / zmax
        refx = O.reverse_step(O.diffusion_tables(K, 200), lc, lu, x.to(dev), x_known.to(dev), mk.to(dev).bool(), t, u1[0], u2[0], 3.0, 0.7)
        refx[:, 0] = x_known[:, 0].to(dev)
        out.update(nar_S=S, nar_max_rel_dlogit=round(e_rel, 5), nar_max_abs_logit=round(zmax, 2),
                   nar_id_agreement=round(float((sess.x[off:, 1:] == refx[off:, 1:]).float().mean()), 4),
                   nar_known_ids_equal=bool((sess.x[:off] == refx[:off]).all()))
    return out


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline_leg(m, bundle, ref_codes, cfg, n_gen, budget_s=24.0, threads=0):
    """The oracle (a torch-CPU restatement of the reference path, incl. the reference's
    per-token speaker-encoder recompute and per-forward NAR speaker encoder so the COST is the
    reference's) on a bounded sample of the same workload, extrapolated linearly:
    AR prefill + a few decode tokens, and one NAR forward (x2 for CFG) at the bench shapes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mars5_oracle as O
    # more than 32 threads slow torch-CPU down on these shapes (measured on the GPU box's host with --cpu-threads 64 / 128:
    # profiles/r5f_cpu_baseline_threads.txt)
    cores = threads if threads > 0 else min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    tt = m.texttok.encode("<|startoftext|>" + TRANSCRIPT + ' ' + TEXT.strip() + "<|endoftext|>", allowed_special='all')
    sp = m.speechtok.encode(' '.join(str(t) for t in ref_codes[0, 0].tolist()))
    n_text = len(m.texttok.vocab)
    prompt = torch.tensor(tt + [s + n_text for s in sp], dtype=torch.long)
    ref = ref_codes[0].T.contiguous().cpu()
    sd_ar, sd_nar = bundle.ar_ckpt["model"], bundle.nar_ckpt["model"]
    nh = bundle.ar_shape.nhead
    t_start = time.perf_counter()
    with torch.inference_mode():
        st = O.ARState()
        t0 = time.perf_counter()
        O.codeclm_step(sd_ar, prompt, ref, st, 1, nh, recompute_spk=True)
        t_prefill = time.perf_counter() - t0
        toks, n_tok, t_acc = prompt, 0, 0.0
        while n_tok < 32 and (time.perf_counter() - t_start) < budget_s * 0.6:
            toks = torch.cat([toks, torch.tensor([n_text + 5 + n_tok])])
            t0 = time.perf_counter()
            O.codeclm_step(sd_ar, toks, ref, st, 2 + n_tok, nh, recompute_spk=True)
            t_acc += time.perf_counter() - t0
            n_tok += 1
        t_tok = t_acc / max(n_tok, 1)
        S = ref.shape[0] + ref.shape[0] - 1 + n_gen
        x = torch.randint(0, 1025, (S, 8), generator=torch.Generator().manual_seed(0))
        n_fwd, t_fwd = 0, 0.0
        while n_fwd < 3 and (n_fwd == 0 or (time.perf_counter() - t_start) < budget_s):
            t0 = time.perf_counter()
            O.nar_forward(sd_nar, bundle.nar_shape.nhead, torch.tensor(tt), ref, x, 100 + n_fwd, n_fwd % 2 == 1)     # cond / uncond alternate
            t_fwd += time.perf_counter() - t0
            n_fwd += 1
        t_step = 2.0 * t_fwd / n_fwd
    total = t_prefill + n_gen * t_tok + 200 * t_step
    audio_s = (n_gen - 1) / 75.0
    cpu_name = "unknown"
    try:
        cpu_name = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        pass
    ratio = None
    try:        # one-off same-host timing of the UNMODIFIED reference functions vs this port (oracle/time_ref_vs_port.py, build container)
        rj = json.load(open(os.path.join(ROOT, "profiles", "r2_ref_vs_port_cpu.json")))
        ratio = round(rj["port_extrapolated_s_per_utterance"] / rj["reference_extrapolated_s_per_utterance"], 3)
    except Exception:
        pass
    blas = "mkl" if torch.backends.mkl.is_available() else ("openblas/other" if torch.backends.openmp.is_available() else "unknown")
    return dict(value=round(audio_s / total, 5), unit="audio_s/s", cores=cores, kind="port", host_cpu=cpu_name, host_logical_cpus=os.cpu_count(),
                torch=torch.__version__, blas=blas, mkldnn=bool(torch.backends.mkldnn.is_available()), sample_wall_s=round(time.perf_counter() - t_start, 1),
                port_over_reference_time=ratio,
                sample=(f"oracle/mars5_oracle.py (torch-CPU fp32 port of the reference path with the reference's cost model) on this host, "
                        f"{cores} threads: AR prefill P={prompt.shape[0]} {t_prefill:.2f}s + {n_tok} decode tokens at {t_tok:.3f}s/token, "
                        f"{n_fwd} NAR forwards at S={S} -> x2 (CFG) = {t_step:.2f}s/step; extrapolated to {n_gen} tokens + 200 steps = {total:.0f}s/utterance; "
                        f"port/reference time ratio measured once on the build host: {ratio} (profiles/r2_ref_vs_port_cpu.json)"))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # N ranks, one per GPU; this process only waits for them
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.launch_check:
        launch_check(args, world, rank)
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    global PREFLIGHT
    if not args.no_preflight:
        PREFLIGHT = preflight(local)
        if not PREFLIGHT["ok"]:
            msg = {"preflight_failed": True, "rank": rank, "local_rank": local, "preflight": PREFLIGHT,
                   "meaning": "the GPU of this box failed a fill / copy / one-kernel sanity check in a child process BEFORE bench.py "
                              "touched it: a box problem, not a product result; nothing was measured"}
            print(json.dumps(msg), flush=True)
            print("bench.py: GPU pre-flight failed (exit code 3): " + json.dumps(PREFLIGHT), file=sys.stderr, flush=True)
            sys.exit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll = world > 1 or (args.collective_at_1 and args.workload == "c4" and "WORLD_SIZE" in os.environ)
    args.coll = coll
    if coll:
        import torch.distributed as dist
        dist.init_process_group(backend=args.backend, device_id=dev)
    from mars5_tts_amd import synth
    global TEXT
    workload_name = ("BASELINE configs[1]: single utterance deep-clone, temperature=0.7 top_k=100, 6 s / 450-frame synthetic "
                     "reference, ~20-token text + transcript, 450 generated frames, 200 DDPM steps x CFG, seeded random weights "
                     "(AR 1536d x 26L n_vocab 4096, NAR 1024d 8+16L)")
    if args.workload == "c5":
        args.n_gen = 4500
        TEXT = " ".join(WORDS[(7 * i) % len(WORDS)] for i in range(150)).capitalize() + "."
        workload_name = ("BASELINE configs[4]: long-form single utterance deep-clone, 60 s target (4500 generated frames, ~150-word "
                         "text), 6 s / 450-frame synthetic reference, temperature=0.7 top_k=100, hipGraph AR decode step over the "
                         "3000-slot rotating KV window, 200 DDPM steps x CFG at S = 5399, seeded random weights of the real geometry")
    m, bundle = build_model(args.dtype, dev, world, rank)
    ref_codes = synth.make_ref_codes(args.ref_frames, seed=7).to(dev)
    p_len, n_text_tok = prompt_len(m, ref_codes)
    cfg = make_cfg(n_text_tok, p_len, args.n_gen)
    if args.no_graph:
        import mars5_tts_amd.ar_engine as ae
        import mars5_tts_amd.nar_engine as ne
        _d, _r = ae.ARSession.decode, ne.NARSession.run
        ae.ARSession.decode = lambda self, use_graph=True, poll=32: _d(self, False, poll)
        ne.NARSession.run = lambda self, uniform, use_graph=True, n_steps=None: _r(self, uniform, False, n_steps)

    def barrier():
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload in ("c3", "c4"):
        (main_c3 if args.workload == "c3" else main_c4)(args, m, dev, world, rank, barrier)
        if coll:
            dist.destroy_process_group()
        return

    for i in range(args.warmup):
        run_utterance(m, ref_codes, cfg, 500 + i)

    # N > 1: the requests of ALL ranks are built on rank 0 and reach their rank through the request scatter (RCCL), the
    # results go back through the gather -- both OUTSIDE the timed region, which is pure replica compute as for N = 1 --
    # and rank 0 re-runs a request that another GPU ran to check that placement does not change a single code.
    shard, all_reqs, work = None, None, None
    if world > 1:
        from mars5_tts_amd import sharding as sh
        tt_ids = m.texttok.encode("<|startoftext|>" + TRANSCRIPT + ' ' + TEXT.strip() + "<|endoftext|>", allowed_special='all')
        all_reqs = [sh.Request(idx=r * args.steps + i, text_ids=torch.tensor(tt_ids, dtype=torch.long), ref_codes=ref_codes[0].T.contiguous().cpu(),
                               seed=1000 + r * 10007 + i, n_gen_est=args.n_gen,
                               n_phones_gen=round(cfg.eos_estimated_gen_length_factor * len(TEXT)), max_len=p_len + args.n_gen)
                    for r in range(world) for i in range(args.steps)]
        shard = sh.scatter_requests(all_reqs if rank == 0 else None, src=0)
        work = request_worker(m, cfg)

    barrier()
    t0 = time.perf_counter()
    lat, frames, results = [], 0, []
    for i in range(args.steps if shard is None else len(shard)):
        if shard is None:
            dt, n_out, _ = run_utterance(m, ref_codes, cfg, 1000 + rank * 10007 + i)
        else:
            t1 = time.perf_counter()
            final = work(shard[i])
            torch.cuda.synchronize()
            dt, n_out = time.perf_counter() - t1, int(final.shape[0])
            results.append((shard[i].idx, final))
        lat.append(dt)
        frames += n_out
    barrier()
    elapsed = time.perf_counter() - t0
    collective = None
    if world > 1:
        from mars5_tts_amd.sharding import reduce_timing
        elapsed, frames = reduce_timing(elapsed, float(frames))
        lats = [None] * world
        dist.all_gather_object(lats, lat)
        lat = [x for l in lats for x in l]
        gathered = sh.gather_results(results, len(all_reqs), dst=0)
        census = sh.rank_census()
        if rank == 0:
            checked = verify_remote(m, cfg, all_reqs, {r.idx for r in shard}, gathered, k=1)
            collective = dict(backend=sh.LAST_STATS.get("backend"), ranks_seen=sh.LAST_STATS.get("ranks_seen"), census=census,
                              scatter_bytes=sh.LAST_STATS.get("scatter_bytes"), gather_bytes=sh.LAST_STATS.get("gather_bytes"),
                              remote_results_rechecked_equal=checked, in_timed_region=False)
    audio_s = frames / 75.0
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    out = {
        "metric": "generated audio seconds/sec (RTF), deep-clone", "value": round(audio_s / elapsed, 4), "unit": "audio_s/s",
        "n_gpus": (collective["ranks_seen"] if collective else world), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "p50_latency_s": round(statistics.median(lat), 4),
        "per_step": {"note": "timed utterances in order: wall s, AR decode ms, NAR loop ms, persistent-decode recoveries",
                     "steps": [list(x) for x in STEP_LOG[-args.steps:]]} if world == 1 else None,
        "config": {"workload": workload_name,
                   "ar_prompt_tokens": p_len, "generated_frames_per_utterance": frames / (args.steps * world),
                   "parallelism": f"replicas x{world} (one utterance stream per GPU; requests scattered / results gathered over "
                                  f"{'RCCL' if world > 1 else 'nothing at N = 1'} outside the timed region)",
                   "hipgraph": not args.no_graph},
        "collective": collective,
        "preflight": PREFLIGHT,
        "legs_done": [],
    }
    # The headline goes out NOW, before any optional leg: every leg below re-prints the enriched line when it finishes (the LAST
    # line is the complete one), and a leg that raises is recorded as "<leg>_error" instead of taking the measurement with it
    # (round 3 lost its driver record to an exception in a leg that ran two minutes after the timed loop had finished).
    emit(out)

    def leg(name, fn):
        run_leg(out, name, fn, torch.cuda.synchronize)

    single = world == 1
    if not args.no_roofline:
        def _roofline():
            from mars5_tts_amd import ar_engine, nar_engine
            ar_stats, nar_stats = dict(ar_engine.LAST_STATS), dict(nar_engine.LAST_STATS)
            try:
                roof, ar_roof, nar, kernels = roofline_leg(m, ref_codes, cfg, args.dtype)
            finally:
                ar_engine.LAST_STATS.update(ar_stats)
                nar_engine.LAST_STATS.update(nar_stats)
            out["roofline"] = roof
            out["roofline_ar_decode"] = ar_roof
            out["nar_loop"] = nar
            out["kernels"] = kernels
            out["time_split_ms"] = {"ar_decode": round(ar_engine.LAST_STATS["decode_ms"], 1), "nar_loop": round(nar["ms_per_step"] * 200, 1)}
        leg("roofline", _roofline)
    if single and not args.no_cpu_baseline:
        leg("cpu_baseline", lambda: out.__setitem__("cpu_baseline", cpu_baseline_leg(m, bundle, ref_codes, cfg, args.n_gen, threads=args.cpu_threads)))
    if single and not args.no_parity and args.workload == "c2":
        leg("parity", lambda: out.__setitem__("parity", parity_leg(m, bundle, ref_codes, args.dtype)))
    if single and args.workload == "c2" and not args.no_batch_leg:
        # throughput mode beside the headline (BASELINE configs[2] in small): 8 mixed-length requests through the batched
        # AR decode + batched NAR refinement; the SECOND pass is timed (the first one captures the step graphs of these shapes)
        from inference import InferenceConfig
        bcfg = InferenceConfig(deep_clone=True, temperature=0.7, top_k=100, freq_penalty=3, rep_penalty_window=100,
                               eos_estimated_gen_length_factor=100.0, eos_penalty_factor=50.0, eos_penalty_decay=0.5)

        def batch_leg(n_req, ar_b, nar_b, passes=2, recheck=0):
            texts, trs, refs, max_lens = c3_requests(m, n_req, args.n_gen, seed=11)
            refs = [r.to(dev) for r in refs]
            dtb, res, chk = None, None, {}
            for ps in range(passes):
                # the warm-up pass runs under the spy (it only reads generator states and keeps references): its group
                # results are re-checked against lone calls after the timed pass
                with (NarGroupSpy() if (recheck and ps == 0) else contextlib.nullcontext()) as spy:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    res = m.tts_batch_from_codes(texts, refs, trs, bcfg, seeds=[1000 + j for j in range(n_req)], nar_batch=nar_b, ar_batch=ar_b,
                                                 max_lens=max_lens)
                    torch.cuda.synchronize()
                    dtb = time.perf_counter() - t0
                if spy is not None:
                    keep = spy
            if recheck:
                chk = keep.recheck(dev, k=recheck)
            r = {"value": round(sum(int(f.shape[0]) for _, f in res) / 75.0 / dtb, 4), "unit": "audio_s/s", "requests": n_req,
                 "s_per_batch": round(dtb, 3), "reference_frames": [int(r.shape[-1]) for r in refs],
                 "note": f"tts_batch_from_codes(ar_batch={ar_b}, nar_batch={nar_b}), pass {passes} of {passes} timed (pass 1 captures the graphs)"}
            r.update(chk)
            return r

        leg("batch8_mixed_lengths", lambda: out.__setitem__("batch8_mixed_lengths", batch_leg(8, 8, 8)))
        if not args.no_extra_legs:
            # BASELINE configs[2] in full (32 mixed-length requests per step) and configs[4] (60 s long-form utterance): one
            # timed step each after one warm-up step, so that the driver's record carries them (`--workload c3 / c5` are the
            # stand-alone forms with their own time splits)
            leg("c3_batch32", lambda: out.__setitem__("c3_batch32", batch_leg(32, args.ar_batch, args.nar_batch, recheck=2)))

            def _c5():
                global TEXT
                text_c2 = TEXT
                try:
                    TEXT = " ".join(WORDS[(7 * i) % len(WORDS)] for i in range(150)).capitalize() + "."
                    p5, nt5 = prompt_len(m, ref_codes)
                    cfg5 = make_cfg(nt5, p5, 4500)
                    run_utterance(m, ref_codes, cfg5, 700)
                    dt5, n5, _ = run_utterance(m, ref_codes, cfg5, 701)
                finally:
                    TEXT = text_c2
                from mars5_tts_amd import ar_engine as _ae, nar_engine as _ne
                out["c5_longform"] = {"value": round(n5 / 75.0 / dt5, 4), "unit": "audio_s/s", "s_per_utterance": round(dt5, 3), "generated_frames": n5,
                                      "ar_us_per_token": round(1e3 * _ae.LAST_STATS["decode_ms"] / max(_ae.LAST_STATS["n_generated"] - 1, 1), 1),
                                      "nar_ms_per_step": round(_ne.LAST_STATS["loop_ms"] / _ne.LAST_STATS["steps"], 3), "nar_S": _ne.LAST_STATS["S"],
                                      "note": "BASELINE configs[4]: 60 s target, AR context past the 3000-slot rotating KV window; second utterance timed"}
            leg("c5_longform", _c5)

            def _refprec():
                # The headline runs in BASELINE's dtype (bf16 operands in both stages).  The reference itself computes the AR stage
                # under fp16 autocast (ar_generate.py:59,67) and calls the NAR model in fp32 (diffuser.py:358-364, no autocast): the
                # same utterance at THAT arithmetic -- f16 AR engine + the exact-fp32 NAR engine (the parity instrument of
                # tests/test_gpu_e2e.py::test_full_size_goldens_f32: v_mfma_f32_16x16x4_f32 kernels, correct, not tuned) -- so the
                # line states what the reference's own precision costs here.  One utterance, engines built before the clock starts.
                from mars5_tts_amd import ar_engine as _ae, nar_engine as _ne, ops as _ops
                from mars5_tts_amd.ops import DT_NAME

                def one(products):
                    prev = _ops.set_f32_products(products)
                    try:
                        m.codecnar.set_engine_dtype(torch.float32)
                        m.codecnar.engine()
                        torch.cuda.synchronize()
                        torch.manual_seed(901)
                        t0 = time.perf_counter()
                        gen, final = m.tts_from_codes(TEXT, ref_codes, TRANSCRIPT, cfg)
                        torch.cuda.synchronize()
                        dtr = time.perf_counter() - t0
                    finally:
                        _ops.set_f32_products(prev)
                    nr = int(final.shape[0])
                    return final.cpu(), {
                        "value": round(nr / 75.0 / dtr, 4), "unit": "audio_s/s", "s_per_utterance": round(dtr, 3), "generated_frames": nr,
                        "ar_dtype": "f16", "nar_dtype": "f32", "nar_products": products,
                        "ar_us_per_token": round(1e3 * _ae.LAST_STATS["decode_ms"] / max(_ae.LAST_STATS["n_generated"] - 1, 1), 1),
                        "nar_ms_per_step": round(_ne.LAST_STATS["loop_ms"] / _ne.LAST_STATS["steps"], 3)}
                try:
                    m.codeclm.set_engine_dtype(torch.float16)
                    m.codeclm.engine()
                    ids_exact, rec = one("exact")
                    rec["note"] = ("the same configs[1] utterance at the reference's own arithmetic (fp16-autocast AR, fp32 NAR); one utterance, first of its "
                                   "engines (step-graph captures inside); exact fp32 MFMA products (an fmaf chain): the parity instrument, not a tuned path")
                    out["reference_precision"] = rec
                    ids_x3, rec3 = one("f16x3")
                    same = float((ids_x3 == ids_exact).float().mean()) if ids_x3.shape == ids_exact.shape else None
                    rec3["codec_ids_equal_to_exact_fp32_run"] = None if same is None else round(same, 6)
                    rec3["note"] = ("the same utterance and seed with the fp32 NAR's GEMM products on the f16 matrix pipe as three split-operand terms "
                                    "(csrc/gemm.hip X3: operand error 2^-22, fp32 accumulation; attention, norms and the posterior stay exact fp32): the "
                                    "fast parity-grade mode; `codec_ids_equal_to_exact_fp32_run` = fraction of the final (frames, 8) ids equal to the exact run's "
                                    "(the DDPM trajectory is chaotic: one near-tie flip cascades)")
                    out["reference_precision_f16x3"] = rec3
                finally:
                    m.codeclm.set_engine_dtype(DT_NAME[args.dtype])
                    m.codecnar.set_engine_dtype(DT_NAME[args.dtype])
            leg("reference_precision", _refprec)
    if single and args.workload == "c2" and args.pipeline:
        # serving mode, reported beside (never as) the headline: the same utterances as a pipelined stream -- request i+1's AR
        # decode overlaps request i's NAR steps (Mars5TTS.tts_stream_from_codes); throughput up, per-request latency not
        def _pipe():
            n_p = max(args.steps, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fr = 0
            for _, fin in m.tts_stream_from_codes([TEXT] * n_p, [ref_codes] * n_p, [TRANSCRIPT] * n_p, cfg, seeds=[3000 + i for i in range(n_p)]):
                fr += int(fin.shape[0])
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
            out["pipelined_stream"] = {"value": round(fr / 75.0 / dtp, 4), "unit": "audio_s/s", "requests": n_p, "s_per_request": round(dtp / n_p, 4),
                                       "note": "AR decode of request i+1 overlapped with the NAR steps of request i on two HIP streams; "
                                               "results identical to sequential seeded calls (tests/test_gpu_e2e.py)"}
        leg("pipelined_stream", _pipe)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
