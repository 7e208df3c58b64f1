"""Utterance-level sharding across the GPUs of one node.

Utterances are independent (the reference's ``tts()`` is per-utterance, ``inference.py:201-307``; the
only shared state is the RNG), so the multi-GPU form of the hot path is REPLICAS: one process per
GPU, full weights on each, requests partitioned by estimated cost.  The only collectives are the
request scatter and the result gather (~30 KB per utterance each way) -- ``torch.distributed`` over
RCCL/xGMI on the GPU box (backend ``nccl``), ``gloo`` in the CPU tests.  Per-utterance seeds make the
result of an utterance independent of which rank ran it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class Request:
    idx: int                    # position in the caller's batch
    text_ids: torch.Tensor      # (Lt,) int64: tokenised "<|startoftext|> transcript text <|endoftext|>"
    ref_codes: torch.Tensor     # (Lc, 8) int64 Encodec codes of the reference audio
    seed: int                   # per-utterance RNG seed (placement-independent results)
    n_gen_est: int = 450        # expected generated frames (cost model only)
    n_phones_gen: int = 0       # EOS-penalty length estimate: round(factor * len(text)) (inference.py:268)
    max_len: int = -1           # InferenceConfig.generate_max_len_override for this request (-1: the default cap)


def estimate_cost(req: Request) -> float:
    """Relative cost of one deep-clone utterance: AR decode steps (~ generated frames) stream the
    weights once each; every NAR step is quadratic-ish in S = 2 Lc + G (attention) on top of a
    linear GEMM term."""
    lc, g = int(req.ref_codes.shape[0]), int(req.n_gen_est)
    s = 2 * lc + g
    return 0.8 * g + 200 * (3.0e-3 * s + 4.0e-7 * s * s)


def lpt_partition(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first: items by decreasing cost onto the least-loaded rank.
    Deterministic (ties by index), so every rank computes the same partition."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return parts


LAST_STATS: dict = {}        # bytes moved by the last scatter_requests / gather_results of this process, rank census


def _dev(group=None) -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def reduce_timing(elapsed_s: float, units: float, group=None) -> Tuple[float, float]:
    """The bench contract for N > 1 replicas: (MAX over ranks of the timed region, SUM over ranks of the units
    processed).  Every rank gets the pair.  Two tiny all-reduces; the data path itself has no collective."""
    dev = _dev(group)
    t = torch.tensor([elapsed_s, units], dtype=torch.float64, device=dev)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(tmax[0]), float(t[1])


def _pack(reqs: List[Request]) -> torch.Tensor:
    """[n, then per request: idx, seed, n_gen_est, n_phones_gen, max_len, Lt, Lc, text_ids..., ref_codes (row-major)...]"""
    parts = [torch.tensor([len(reqs)], dtype=torch.int64)]
    for r in reqs:
        parts.append(torch.tensor([r.idx, r.seed, r.n_gen_est, r.n_phones_gen, r.max_len, r.text_ids.numel(), r.ref_codes.shape[0]],
                                  dtype=torch.int64))
        parts.append(r.text_ids.reshape(-1).to(torch.int64).cpu())
        parts.append(r.ref_codes.reshape(-1).to(torch.int64).cpu())
    return torch.cat(parts)


def _unpack(buf: torch.Tensor) -> List[Request]:
    buf = buf.cpu()
    n, p, out = int(buf[0]), 1, []
    for _ in range(n):
        idx, seed, n_gen, n_ph, max_len, lt, lc = (int(v) for v in buf[p:p + 7])
        p += 7
        text = buf[p:p + lt].clone()
        p += lt
        codes = buf[p:p + lc * 8].reshape(lc, 8).clone()
        p += lc * 8
        out.append(Request(idx, text, codes, seed, n_gen, n_ph, max_len))
    return out


def scatter_requests(requests: Optional[List[Request]], src: int = 0, group=None) -> List[Request]:
    """Rank `src` holds the whole batch; every rank returns its own shard (LPT by estimated cost).
    One broadcast of the per-rank payload sizes + one ``dist.scatter`` of padded int64 payloads."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    payloads: List[torch.Tensor] = []
    if rank == src:
        assert requests is not None
        parts = lpt_partition([estimate_cost(r) for r in requests], world)
        payloads = [_pack([requests[i] for i in part]) for part in parts]
        sizes = torch.tensor([p.numel() for p in payloads], dtype=torch.int64, device=dev)
    dist.broadcast(sizes, src=src, group=group)
    width = int(sizes.max())
    mine = torch.zeros(width, dtype=torch.int64, device=dev)
    chunks = None
    if rank == src:
        chunks = [torch.cat([p, torch.zeros(width - p.numel(), dtype=torch.int64)]).to(dev) for p in payloads]
    dist.scatter(mine, chunks, src=src, group=group)
    LAST_STATS["scatter_bytes"] = int(sizes.sum()) * 8           # payload int64 words over all ranks (padding excluded)
    LAST_STATS["scatter_bytes_this_rank"] = int(sizes[rank]) * 8
    return _unpack(mine[: int(sizes[rank])])


def gather_results(results: List[Tuple[int, torch.Tensor]], n_total: int, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """results: (request idx, codes (G, 8) int64) of this rank.  Rank `dst` returns the codes of all
    `n_total` requests in batch order; other ranks return None.  One all-reduce (max payload width)
    + one ``dist.gather``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    parts = [torch.tensor([len(results)], dtype=torch.int64)]
    for idx, codes in results:
        parts.append(torch.tensor([idx, codes.shape[0]], dtype=torch.int64))
        parts.append(codes.reshape(-1).to(torch.int64).cpu())
    buf = torch.cat(parts)
    width = torch.tensor([buf.numel()], dtype=torch.int64, device=dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX, group=group)
    w = int(width)
    mine = torch.cat([buf, torch.zeros(w - buf.numel(), dtype=torch.int64)]).to(dev)
    bins = [torch.zeros(w, dtype=torch.int64, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bins, dst=dst, group=group)
    LAST_STATS["gather_bytes_this_rank"] = int(buf.numel()) * 8
    if rank != dst:
        return None
    LAST_STATS["gather_bytes"] = sum(8 * (1 + sum(2 + g * 8 for g in _gather_lengths(b))) for b in bins)   # payload words, padding excluded
    out: List[Optional[torch.Tensor]] = [None] * n_total
    for b in bins:
        b = b.cpu()
        n, p = int(b[0]), 1
        for _ in range(n):
            idx, g = int(b[p]), int(b[p + 1])
            p += 2
            out[idx] = b[p:p + g * 8].reshape(g, 8).clone()
            p += g * 8
    assert all(o is not None for o in out), "gather_results: missing utterances"
    return out  # type: ignore[return-value]


def _gather_lengths(b: torch.Tensor) -> List[int]:
    """frame counts of the results packed in one rank's gather payload"""
    b = b.cpu()
    n, p, out = int(b[0]), 1, []
    for _ in range(n):
        g = int(b[p + 1])
        out.append(g)
        p += 2 + g * 8
    return out


def rank_census(group=None) -> List[dict]:
    """Every rank contributes (rank, local device index, device count it sees, cuda-or-not) through ONE all_gather over
    the job's backend (RCCL on the GPU box): the returned list, identical on every rank, is the evidence that the
    collective saw `world` distinct ranks / devices."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    on_gpu = dev.type == "cuda"
    mine = torch.tensor([rank, torch.cuda.current_device() if on_gpu else -1, torch.cuda.device_count() if on_gpu else 0, int(on_gpu)],
                        dtype=torch.int64, device=dev)
    allv = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    census = [dict(rank=int(v[0]), device=int(v[1]), devices_visible=int(v[2]), gpu=bool(int(v[3]))) for v in allv]
    LAST_STATS["ranks_seen"] = len({c["rank"] for c in census})
    LAST_STATS["backend"] = dist.get_backend(group)
    return census


def run_sharded(requests: Optional[List[Request]], n_total: int, worker: Optional[Callable[[Request], torch.Tensor]],
                src: int = 0, group=None, batch_worker: Optional[Callable[[List[Request]], List[torch.Tensor]]] = None) -> Optional[List[torch.Tensor]]:
    """scatter -> each rank runs `worker(request) -> (G, 8) codes` on its shard (seeded per utterance)
    -> gather on `src`.  `batch_worker(shard) -> [codes]` instead hands a rank its whole shard at once (it seeds per request
    itself: ``Mars5TTS.tts_batch_from_ids``), so that the rank can refine its requests in groups."""
    shard = scatter_requests(requests, src=src, group=group)
    if batch_worker is not None:
        outs = batch_worker(shard) if shard else []
        assert len(outs) == len(shard)
        return gather_results([(r.idx, o) for r, o in zip(shard, outs)], n_total, dst=src, group=group)
    done = []
    for r in shard:
        torch.manual_seed(r.seed)
        done.append((r.idx, worker(r)))
    return gather_results(done, n_total, dst=src, group=group)
