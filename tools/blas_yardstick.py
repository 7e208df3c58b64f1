#!/usr/bin/env python
"""Library yardstick for the NAR decoder GEMM shapes: what does the vendor GEMM (hipBLASLt / rocBLAS behind torch.mm) reach
on the seven shapes of one reverse step, on the same box, same random operands, next to this repo's gemm16 kernels?
Tools only -- the product never calls a BLAS library.  Both sides are timed the same way: REP back-to-back launches captured
in a graph, replayed 5 times, HIP events around the replays (rounds interleaved).  Run it under
`rocprofv3 --kernel-trace --stats` to also get the library's kernel names (tile shapes) and per-kernel durations."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("M5_HIP_TOOLS", "1")
import torch
import mars5_tts_amd as pkg            # noqa
from mars5_tts_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
REP = int(os.environ.get("REP", "20"))

SHAPES = [   # (name, M, N, K, our epilogue)
    ("self qkv", 2816, 3072, 1024, L.EPI_DT),
    ("out_proj (residual)", 2816, 1024, 1024, L.EPI_RESIDUAL),
    ("xattn scores", 2816, 768, 1024, L.EPI_DT),
    ("xattn P.B (residual)", 2816, 1024, 768, L.EPI_RESIDUAL),
    ("swiglu", 2816, 6144, 1024, L.EPI_SWIGLU),
    ("linear2 (residual)", 2816, 1024, 3072, L.EPI_RESIDUAL),
    ("heads (7 x 1798 rows)", 12586, 1025, 1024, L.EPI_F32),
]


# MROWS=35840 (a batched NAR group of 16 utterances) / 98304 (a group of 32): the same seven shapes at throughput-mode row counts
# (how far is the large-M gemm16 configuration from what the vendor library reaches there?)
if os.environ.get("MROWS"):
    _m = int(os.environ["MROWS"])
    SHAPES = [(n, (_m if M == 2816 else _m * 12586 // 2816 // 64 * 64), N, K, e) for n, M, N, K, e in SHAPES]
    REP = max(2, REP // 4)


def time_graph(fn, stream):
    st = stream.cuda_stream
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(REP):
                fn()
        g.replay()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            g.replay()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


def main():
    res = []
    stream = torch.cuda.Stream()
    gen = torch.Generator(device="cpu").manual_seed(1)
    for name, M, N, K, epi in SHAPES:
        a = (torch.randn(M, K, generator=gen) * 0.5).to(dev, dt)
        w = (torch.randn(N, K, generator=gen) * (1.0 / K ** 0.5)).to(dev, dt)
        wt = w.t()                                            # torch.mm(a, w^T): the "TN" GEMM the library prefers for K-contiguous operands
        out_lib = torch.empty(M, N, dtype=dt, device=dev)
        if epi in (L.EPI_F32, L.EPI_RESIDUAL):
            out = torch.zeros(M, N, dtype=torch.float32, device=dev)
        elif epi == L.EPI_SWIGLU:
            out = torch.zeros(M, N // 2, dtype=dt, device=dev)
        else:
            out = torch.zeros(M, N, dtype=dt, device=dev)
        st = stream.cuda_stream
        lib = lambda: torch.mm(a, wt, out=out_lib)
        ours = lambda: ops.gemm(a, w, out, epi, stream=st)
        # correctness of ours on this shape (plain / fp32 epilogues only; the fused ones are covered by tests/)
        if epi in (L.EPI_DT, L.EPI_F32):
            with torch.cuda.stream(stream):
                ours()
                lib()
            stream.synchronize()
            err = float((out.float() - out_lib.float()).abs().max())
        else:
            err = None
        t_lib, t_our = [], []
        for _ in range(3):
            t_lib.append(time_graph(lib, stream))
            t_our.append(time_graph(ours, stream))
        fl = 2.0 * M * N * K
        r = dict(name=name, M=M, N=N, K=K, lib_us=round(min(t_lib), 2), ours_us=round(min(t_our), 2),
                 lib_tflops=round(fl / min(t_lib) / 1e6, 1), ours_tflops=round(fl / min(t_our) / 1e6, 1), max_abs_diff=err)
        print(json.dumps(r), flush=True)
        res.append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"blas_yardstick{os.environ.get('MROWS', '')}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
