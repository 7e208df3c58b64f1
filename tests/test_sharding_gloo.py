"""Multi-process CPU tests of the N > 1 path (world_size 2, gloo): the request scatter / result
gather of ``mars5_tts_amd.sharding`` and its placement-independence contract."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mars5_tts_amd import sharding as sh


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_requests(n=7):
    g = torch.Generator().manual_seed(11)
    reqs = []
    for i in range(n):
        lc = int(torch.randint(150, 900, (1,), generator=g))
        lt = int(torch.randint(10, 60, (1,), generator=g))
        reqs.append(sh.Request(i, torch.randint(0, 2048, (lt,), generator=g), torch.randint(0, 1024, (lc, 8), generator=g),
                               seed=1000 + i, n_gen_est=200 + 37 * i))
    return reqs


def _fake_tts(r: sh.Request) -> torch.Tensor:
    """Deterministic stand-in for the GPU hot path: depends on the request AND on the RNG that
    run_sharded seeds per utterance, so it detects placement-dependent seeding."""
    g = 5 + r.idx
    noise = torch.randint(0, 1024, (g, 8))                       # global generator, seeded by run_sharded
    return (r.ref_codes[:g] + r.text_ids[:1] + noise) % 1024


def _worker(rank: int, world: int, port: int, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reqs = _make_requests() if rank == 0 else None
        n = 7
        shard = sh.scatter_requests(reqs, src=0)
        full = _make_requests()
        parts = sh.lpt_partition([sh.estimate_cost(r) for r in full], world)
        assert sorted(r.idx for r in shard) == sorted(parts[rank])
        for r in shard:                                          # payload survives the wire bit-exactly
            assert torch.equal(r.text_ids, full[r.idx].text_ids) and torch.equal(r.ref_codes, full[r.idx].ref_codes)
            assert r.seed == full[r.idx].seed and r.n_gen_est == full[r.idx].n_gen_est
        # bench.py's N > 1 reduction: max of the timed regions, sum of the units, on every rank
        tmax, total = sh.reduce_timing(1.5 + rank, 10.0 * (rank + 1))
        assert tmax == 1.5 + (world - 1) and total == 10.0 * world * (world + 1) / 2
        out = sh.run_sharded(reqs, n, _fake_tts, src=0)
        if rank == 0:
            assert out is not None and len(out) == n
            for r in full:                                       # same result as a single-process run
                torch.manual_seed(r.seed)
                assert torch.equal(out[r.idx], _fake_tts(r)), r.idx
            ok.value = 1
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_scatter_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    ok = ctx.Value("i", 0)
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ok)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ok.value == 1


# ---------------------------------------------------------------- the real host path behind the scatter / gather
def _cpu_tts(tiny_vocab):
    from fakes import cpu_standin_tts           # oracle/fakes.py (test infrastructure)
    return cpu_standin_tts(tiny_vocab)


def _host_requests(m, n=6):
    from mars5_tts_amd import synth
    reqs = []
    for i in range(n):
        text, tr = f"Request number {i} says hello.", "A transcript " * (1 + i % 3)
        ids = m.texttok.encode("<|startoftext|>" + tr + ' ' + text.strip() + "<|endoftext|>", allowed_special='all')
        ref = synth.make_ref_codes(20 + 7 * i, seed=50 + i, merge_friendly=True)
        reqs.append(sh.Request(i, torch.tensor(ids, dtype=torch.long), ref[0].T.contiguous(), seed=500 + i, n_gen_est=12,
                               n_phones_gen=len(text), max_len=len(ids) + ref.shape[-1] + 12))
    return reqs


def _host_worker(rank: int, world: int, port: int, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from mars5_tts_amd import synth
        m, inf = _cpu_tts(synth.make_vocab(30, 63))
        cfg = inf.InferenceConfig(deep_clone=True, temperature=0.7, top_k=100)
        reqs = _host_requests(m)
        work = bench.request_worker(m, cfg)                       # the worker bench.py --workload c4 / --gpus N uses
        out = sh.run_sharded(reqs if rank == 0 else None, len(reqs), work, src=0)
        census = sh.rank_census()
        assert [c["rank"] for c in census] == list(range(world)) and sh.LAST_STATS["ranks_seen"] == world
        if rank == 0:
            assert sh.LAST_STATS["scatter_bytes"] == sum(sh._pack([r]).numel() - 1 for r in reqs) * 8 + 8 * world
            parts = sh.lpt_partition([sh.estimate_cost(r) for r in reqs], world)
            assert bench.verify_remote(m, cfg, reqs, set(parts[0]), out, k=2) == 2
            for r in reqs:                                        # and the whole batch equals a single-process run
                torch.manual_seed(r.seed)
                assert torch.equal(out[r.idx], work(r)), r.idx
                assert out[r.idx].shape[1] == 8 and out[r.idx].shape[0] >= 1
            assert sh.LAST_STATS["gather_bytes"] == sum(8 * (2 + o.numel()) for o in out) + 8 * world
        # the same shard handed to the rank AT ONCE (bench.py --workload c4 since round 4: Mars5TTS.tts_batch_from_ids refines a
        # rank's requests in NAR groups): identical codes, request by request
        bwork = bench.request_batch_worker(m, cfg, nar_batch=4, nar_in_flight=2)
        out_b = sh.run_sharded(reqs if rank == 0 else None, len(reqs), None, src=0, batch_worker=bwork)
        if rank == 0:
            assert all(torch.equal(a, b) for a, b in zip(out, out_b))
            ok.value = 1
    finally:
        dist.destroy_process_group()


def test_sharded_run_drives_the_real_host_path_world2_gloo():
    """``bench.request_worker`` (= ``Mars5TTS.tts_from_ids``: prompt assembly from the wire-format ids, AR stage, BPE
    hand-off, NAR stage, prompt skipping) behind ``run_sharded`` on two gloo ranks: results identical to a single-process
    run, byte counters consistent with the payloads, census sees both ranks.  The two device stages are CPU stand-ins
    (no GPU here); on the GPU box the same code runs with the HIP engines over RCCL (``bench.py --workload c4``)."""
    ctx = mp.get_context("spawn")
    ok = ctx.Value("i", 0)
    port = _free_port()
    procs = [ctx.Process(target=_host_worker, args=(r, 2, port, ok)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ok.value == 1


def test_lpt_partition_properties():
    costs = [5.0, 1.0, 4.0, 4.0, 2.0, 9.0, 0.5]
    for world in (1, 2, 3, 8):
        parts = sh.lpt_partition(costs, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(costs)))                   # a partition
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) <= sum(costs) / world + max(costs)     # LPT bound
        assert parts == sh.lpt_partition(costs, world)           # deterministic
    assert sh.lpt_partition([], 4) == [[], [], [], []]


def test_cost_monotone():
    a = sh.Request(0, torch.zeros(20, dtype=torch.long), torch.zeros(150, 8, dtype=torch.long), 0, 200)
    b = sh.Request(1, torch.zeros(20, dtype=torch.long), torch.zeros(900, 8, dtype=torch.long), 0, 200)
    c = sh.Request(2, torch.zeros(20, dtype=torch.long), torch.zeros(150, 8, dtype=torch.long), 0, 900)
    assert sh.estimate_cost(b) > sh.estimate_cost(a) and sh.estimate_cost(c) > sh.estimate_cost(a)
