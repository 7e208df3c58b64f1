"""CPU-side checks of the drop-in boundary: the C-ABI library loads (no GPU needed for that) and
exports exactly the entry points ``include/mars5_hip.h`` declares; argument validation answers with
status codes, never exceptions or crashes; the host-side containers keep the reference's surface."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mars5_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    tools = "".join(re.findall(r"#ifdef M5_TOOLS(.*?)#endif", src, flags=re.S))
    prod = re.sub(r"#ifdef M5_TOOLS.*?#endif", "", src, flags=re.S)
    fn = lambda t: sorted(set(re.findall(r"\b(m5_[a-z0-9_]+)\s*\(", t)))     # noqa: E731
    return fn(prod), fn(tools)


def test_library_exports_every_header_symbol():
    import ctypes
    from mars5_tts_amd import _lib
    names, tool_names = _header_functions()
    assert len(names) >= 20 and len(tool_names) >= 5
    for n in names:
        assert hasattr(_lib.lib, n), f"libmars5_hip.so does not export {n}"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototype table and header disagree"
    assert sorted(_lib.TOOLS_PROTOTYPES) == tool_names
    assert _lib.lib.m5_version() == 1
    assert b"gfx950" in _lib.lib.m5_build_info()
    # the product library carries no diagnostics export and never reads the environment; the tools library has both
    for n in tool_names:
        assert not hasattr(_lib.lib, n), f"product library exports the tools-only symbol {n}"
    import subprocess
    undef = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undef, "libmars5_hip.so imports getenv: an environment variable could change what it computes"
    tools_lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libmars5_hip_tools.so"))
    for n in names + tool_names:
        assert hasattr(tools_lib, n), f"libmars5_hip_tools.so does not export {n}"


def test_argument_validation_returns_status_codes():
    from mars5_tts_amd import _lib
    lib = _lib.lib
    # null pointers / bad sizes are rejected before any launch (safe to call without a GPU)
    assert lib.m5_gemm(_lib.BF16, None, 64, None, 64, None, None, 64, 8, 8, 64, _lib.EPI_F32, None, 1, 0, 0, 0, 0, None) == _lib.M5_ERR_ARG
    assert lib.m5_layernorm(_lib.BF16, None, 0, None, None, 1e-5, None, 0, 1, 64, 1, 0, 0, None) == _lib.M5_ERR_ARG
    assert lib.m5_attention(_lib.BF16, None, None) == _lib.M5_ERR_ARG
    assert lib.m5_ar_sample(None, None) == _lib.M5_ERR_ARG
    assert lib.m5_nar_sample(None, None) == _lib.M5_ERR_ARG
    assert lib.m5_nar_uniforms(None, None) == _lib.M5_ERR_ARG
    bad = _lib.NarUniformArgs(out=None, n=16, K=4, k_magic=0, k_shift=0, m=None, rng=None, inc=4, grid_threads=256, step=None, consts=None)
    assert lib.m5_nar_uniforms(bad, None) == _lib.M5_ERR_ARG          # no output / no generator state: refused before any launch
    assert lib.m5_graph_begin(None) == _lib.M5_ERR_ARG
    with pytest.raises(_lib.Mars5HipError):
        _lib.check(_lib.M5_ERR_UNSUPPORTED, "unit test")


def test_every_compute_entry_point_rejects_null_arguments():
    """All-null / all-zero arguments are an argument error at EVERY compute entry point of the product library, before anything is
    launched (so the sweep runs without a GPU): a binding that passes a missing buffer gets a status, never a fault."""
    from mars5_tts_amd import _lib as L
    handles = {"m5_version", "m5_build_info", "m5_event_create", "m5_event_record", "m5_event_elapsed_ms", "m5_event_destroy",
               "m5_graph_end", "m5_graph_launch", "m5_graph_destroy"}          # no buffers to validate / need a live handle
    swept = 0
    for name, (_, args) in L.PROTOTYPES.items():
        if name in handles:
            continue
        vals = [0 if a in (C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64) else (0.0 if a in (C.c_float, C.c_double) else None)
                for a in args]
        assert getattr(L.lib, name)(*vals) == L.M5_ERR_ARG, name
        swept += 1
    assert swept >= 27


def test_stage_plans_are_validated_before_anything_is_launched():
    """m5_nar_step / m5_ar_decode_step / m5_stage_run (csrc/stage_plan.hip): a plan is checked op by op -- entry point code,
    argument count against the real prototype, stage kind, arena offsets of the argument structures -- and the first bad op
    returns M5_ERR_ARG with its index in *failed_op.  The ops used here fail their OWN argument validation (null buffers), so
    nothing reaches a device: the whole test runs without a GPU.  Also: the Python recorder (ops.StagePlan.recording) turns the
    library calls of a thread into exactly such a plan, fn codes in the header's enum order."""
    from mars5_tts_amd import _lib as L, ops
    hdr = open(os.path.join(ROOT, "include", "mars5_hip.h")).read()
    enum = re.search(r"enum \{\s*(M5_FN_GEMM = 1.*?)\};", hdr, flags=re.S).group(1)
    names = [n.strip().split(" ")[0] for n in enum.replace("\n", " ").split(",") if n.strip()]
    assert ["m5_" + n[len("M5_FN_"):].lower() for n in names] == sorted(L.PLAN_FN, key=L.PLAN_FN.get), "M5_FN_* enum and _lib.PLAN_FN disagree"
    assert int(re.search(r"#define M5_PLAN_MAX_ARGS (\d+)", hdr).group(1)) == L.PLAN_MAX_ARGS
    for name in L.PLAN_FN:
        assert len(L.PROTOTYPES[name][1]) - 1 <= L.PLAN_MAX_ARGS, name
    lib = L.lib
    failed = L.i32(123)

    def plan(ops_, arena=b""):
        arr = (L.PlanOp * max(len(ops_), 1))()
        for i, (fn, slots) in enumerate(ops_):
            arr[i].fn, arr[i].n_args = fn, len(slots)
            for j, v in enumerate(slots):
                arr[i].a[j] = v
        ar = (C.c_ubyte * max(len(arena), 1)).from_buffer_copy(arena or b"\0")
        return L.StagePlanC(ops=arr, n_ops=len(ops_), arena=C.cast(ar, C.c_void_p), arena_bytes=len(arena), failed_op=C.pointer(failed)), (arr, ar)

    for entry in (lib.m5_nar_step, lib.m5_ar_decode_step, lib.m5_stage_run):
        assert entry(None, None) == L.M5_ERR_ARG
        p, keep = plan([])
        assert entry(C.byref(p), None) == L.M5_OK and failed.value == -1          # an empty stage
    p, keep = plan([(999, [])])
    assert lib.m5_stage_run(C.byref(p), None) == L.M5_ERR_ARG and failed.value == 0     # unknown entry point
    add_int = L.PLAN_FN["m5_add_int"]
    p, keep = plan([(add_int, [0, 1, 2])])
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG                           # m5_add_int takes 2 arguments + stream
    p, keep = plan([(add_int, [0, 1])])
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG and failed.value == 0     # right shape, null pointer: the op's own check
    p, keep = plan([(L.PLAN_FN["m5_ar_sample"], [0])])
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG                           # an AR launch is not part of a NAR step
    assert lib.m5_ar_decode_step(C.byref(p), None) == L.M5_ERR_ARG                     # (as an AR step: null argument structure)
    p, keep = plan([(L.PLAN_FN["m5_nar_sample"], [0])])
    assert lib.m5_ar_decode_step(C.byref(p), None) == L.M5_ERR_ARG
    att = L.PLAN_FN["m5_attention"]
    p, keep = plan([(att, [L.BF16, 1 + 64])], arena=bytes(64))
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG                           # structure offset past the arena
    p, keep = plan([(att, [L.BF16, 1 + 4])], arena=bytes(C.sizeof(L.AttnArgs) + 16))
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG                           # misaligned structure offset
    p, keep = plan([(att, [L.BF16, 1 + 8])], arena=bytes(C.sizeof(L.AttnArgs) + 16))
    assert lib.m5_nar_step(C.byref(p), None) == L.M5_ERR_ARG and failed.value == 0     # in range: m5_attention refuses the zeroed args
    # the recorder
    sp = ops.StagePlan("nar_step")
    a = L.AttnArgs(B=1, H=2, Sq=3, Sk=4, scale=0.5)
    with sp.recording():
        assert lib.m5_add_int(None, 7, None) == L.M5_OK                                # recorded, not run (run directly: M5_ERR_ARG)
        assert lib.m5_layernorm(L.BF16, 4096, 1024, 8192, 12288, 1e-5, 16384, 1024, 10, 1024, 1, 0, 0, None) == L.M5_OK
        assert lib.m5_attention(L.BF16, C.byref(a), None) == L.M5_OK
        with pytest.raises(L.Mars5HipError):
            pass_through = lib.m5_version()                                           # (not a launch: executes)
            assert pass_through == 1
            raise L.Mars5HipError("sentinel")
    assert sp.n_ops == 3 and [o.fn for o in sp._arr[:3]] == [add_int, L.PLAN_FN["m5_layernorm"], att]
    assert list(sp._arr[0].a[:2]) == [0, 7] and sp._arr[1].n_args == 13
    assert sp._arr[1].a[5] == int.from_bytes(C.c_float(1e-5), "little")               # floats travel as their bits
    off = sp._arr[2].a[1] - 1
    got = L.AttnArgs.from_buffer_copy(bytes(sp._arena)[off:off + C.sizeof(L.AttnArgs)])
    assert (got.B, got.H, got.Sq, got.Sk, got.scale) == (1, 2, 3, 4, 0.5)             # the structure was copied into the arena
    assert sp.run(0, raise_on_error=False) == L.M5_ERR_ARG and sp._failed.value == 0         # (stream 0) op 0's null pointer is refused when the plan RUNS


def test_persistent_decode_step_refuses_what_it_cannot_run():
    """m5_ar_layers_persistent validates before it touches the device: missing pointers are an argument error, any geometry
    but the CodecLM one (dim 1536, hidden 3584, 24 heads), fp32 operands or more than 31 layers are 'unsupported' -- the
    status on which ARSession falls back to the per-launch step -- and nothing is launched in either case."""
    from mars5_tts_amd import _lib as L
    a = L.ArMegaArgs()
    assert L.lib.m5_ar_layers_persistent(L.BF16, C.byref(a), None) == L.M5_ERR_ARG
    buf = (C.c_uint64 * 4)()
    ptr = C.addressof(buf)
    for f, _ in L.ArMegaArgs._fields_:
        if f not in ("eps", "dim", "hidden", "n_heads", "layer0", "layer1", "w_alloc", "window", "scale", "dbg"):
            setattr(a, f, ptr)
    a.dim, a.hidden, a.n_heads, a.layer0, a.layer1, a.w_alloc, a.window = 1024, 3584, 24, 0, 26, 64, 64
    assert L.lib.m5_ar_layers_persistent(L.BF16, C.byref(a), None) == L.M5_ERR_UNSUPPORTED      # wrong width
    a.dim = 1536
    assert L.lib.m5_ar_layers_persistent(L.F32, C.byref(a), None) == L.M5_ERR_UNSUPPORTED       # parity mode: per-launch form
    a.layer1 = 40
    assert L.lib.m5_ar_layers_persistent(L.BF16, C.byref(a), None) == L.M5_ERR_UNSUPPORTED      # the tag has 5 layer bits
    assert L.AR_MEGA_GRANULES == int(re.search(r"#define M5_AR_MEGA_GRANULES (\d+)", open(os.path.join(ROOT, "include", "mars5_hip.h")).read()).group(1))


def test_inference_config_surface_matches_reference():
    """21 fields, names and defaults of reference inference.py:24-77."""
    from inference import InferenceConfig
    cfg = InferenceConfig()
    want = dict(temperature=0.7, top_k=200, top_p=0.2, typical_p=1.0, freq_penalty=3, presence_penalty=0.4, rep_penalty_window=80,
                eos_penalty_decay=0.5, eos_penalty_factor=1, eos_estimated_gen_length_factor=1.0, timesteps=200, x_0_temp=0.7,
                q0_override_steps=20, nar_guidance_w=3, max_prompt_dur=12, generate_max_len_override=-1, deep_clone=True,
                use_kv_cache=True, trim_db=27, beam_width=1, ref_audio_pad=0)
    assert {k: getattr(cfg, k) for k in want} == want
    assert len(cfg.__dataclass_fields__) == 21


def test_containers_take_reference_state_dict_names(tiny_bundle):
    from mars5_tts_amd import model
    a, n = tiny_bundle.ar_shape, tiny_bundle.nar_shape
    lm = model.CodecLM(a.n_vocab, dim=a.dim, nhead=a.nhead, n_layers=a.n_layers, n_spk_layers=a.n_spk_layers,
                       dim_ff_scale=a.hidden_dim / a.dim + 1e-9)
    lm.load_state_dict(tiny_bundle.ar_ckpt["model"])
    sd = lm.state_dict()
    for k in ("embed.weight", "ar.layers.0.attention.wq.weight", "ar.layers.0.feed_forward.w1.weight", "ar.norm.weight",
              "ar.output.weight", "spk_identity_emb.weight", "ref_chunked_emb.embs.0.weight"):
        assert k in sd, k
    nar = model.ResidualTransformer(n.n_text_vocab, n_quant=n.n_quant, dim=n.dim, nhead=n.nhead, enc_layers=n.enc_layers,
                                    dec_layers=n.dec_layers, n_spk_layers=n.n_spk_layers, t_emb_dim=n.t_emb_dim, p_cond_drop=0)
    nar.load_state_dict(tiny_bundle.nar_ckpt["model"])
    for k in ("tfm.decoder.layers.0.multihead_attn.in_proj_weight", "residual_decoder.7.1.weight", "text_embed.weight"):
        assert k in nar.state_dict(), k
    with pytest.raises(Exception):
        bad = dict(tiny_bundle.ar_ckpt["model"])
        bad.pop("ar.norm.weight")
        lm.load_state_dict(bad)


def test_cross_memory_table_layout():
    """Host-side table for the fused / batched cross-attention: one row per workspace sequence, in order, with the
    per-branch base addresses and per-step strides of that utterance's pre-projected memory."""
    import torch
    from mars5_tts_amd.blocks import CrossMemory, cross_memory_table
    H, T = 4, 3
    mems = []
    for le, lep in ((39, 64), (70, 128)):
        k = torch.zeros(T * 2, H, le, 64, dtype=torch.bfloat16)
        vt = torch.zeros(T * 2, H, 64, lep, dtype=torch.bfloat16)
        mems.append(CrossMemory(k, vt, le, lep, 2))
    tab, max_le = cross_memory_table(mems, "cpu")
    assert tab.shape == (4, 6) and tab.dtype == torch.int64 and max_le == 70
    for u, mem in enumerate(mems):
        for b in range(2):
            row = tab[2 * u + b].tolist()
            assert row[0] == mem.k[b].data_ptr() and row[1] == mem.vt[b].data_ptr()          # branch b of step 0
            assert row[2:4] == [mem.Le, mem.Lep]
            assert row[0] + row[4] * 2 == mem.k[2 + b].data_ptr() and row[1] + row[5] * 2 == mem.vt[2 + b].data_ptr()   # step 1


def test_bench_c3_requests_follow_the_survey_spec():
    """bench.py --workload c3 inputs: reference 150-900 frames, deterministic per seed."""
    import bench

    class Tok:
        def encode(self, s, allowed_special=None):
            return list(range(len(s.split())))

    class M:
        texttok, speechtok = Tok(), Tok()

    a = bench.c3_requests(M(), 16, 450, seed=11)
    b = bench.c3_requests(M(), 16, 450, seed=11)
    assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3] and all((x == y).all() for x, y in zip(a[2], b[2]))
    for ref, ml in zip(a[2], a[3]):
        assert ref.shape[0] == 1 and ref.shape[1] == 8 and 150 <= ref.shape[2] <= 900
        assert ml > 450
    assert len(set(r.shape[2] for r in a[2])) > 8          # genuinely mixed lengths


def test_hub_entry_points_assemble_the_reference_checkpoint_dicts(tiny_bundle, tmp_path, monkeypatch):
    """``hubconf.mars5_english`` (safetensors cache / explicit .pt paths, reference hubconf.py:17-75) and
    ``Mars5TTS.from_pretrained`` (reference inference.py:123-158) hand ``Mars5TTS(ar_ckpt, nar_ckpt, device)`` the
    {'vocab': {texttok.model, speechtok.model}, 'model': state_dict} dicts the reference builds -- no network: the hub
    cache and hf_hub_download are pointed at local files."""
    import torch
    from safetensors.torch import save_file
    import hubconf
    import inference
    b = tiny_bundle
    seen = []

    class Recorder:
        def __init__(self, ar_ckpt, nar_ckpt, device=None, **kw):
            seen.append((ar_ckpt, nar_ckpt, device))

    def same(ck, ref):
        assert ck["vocab"] == ref["vocab"] and set(ck["model"]) == set(ref["model"])
        assert all(torch.equal(ck["model"][k], ref["model"][k]) for k in ref["model"])

    # 1. torch.hub route, safetensors already in the hub cache (file names = the reference's release assets)
    hub = tmp_path / "hub"
    (hub / "checkpoints").mkdir(parents=True)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(hub))
    for url, ck in ((hubconf.ar_sf_url, b.ar_ckpt), (hubconf.nar_sf_url, b.nar_ckpt)):
        save_file({k: v.contiguous() for k, v in ck["model"].items()}, str(hub / "checkpoints" / url.rsplit("/", 1)[1]), metadata=ck["vocab"])
    monkeypatch.setattr(hubconf, "Mars5TTS", Recorder)
    model, cfg_cls = hubconf.mars5_english(device="cpu", ckpt_format="safetensors")
    assert isinstance(model, Recorder) and cfg_cls is inference.InferenceConfig and seen[-1][2] == "cpu"
    same(seen[-1][0], b.ar_ckpt)
    same(seen[-1][1], b.nar_ckpt)
    # 2. explicit .pt paths
    torch.save(b.ar_ckpt, tmp_path / "ar.pt")
    torch.save(b.nar_ckpt, tmp_path / "nar.pt")
    hubconf.mars5_english(device="cpu", ckpt_format="pt", ar_path=tmp_path / "ar.pt", nar_path=tmp_path / "nar.pt")
    same(seen[-1][0], b.ar_ckpt)
    same(seen[-1][1], b.nar_ckpt)
    with pytest.raises(AssertionError):
        hubconf.mars5_english(pretrained=False)
    with pytest.raises(AssertionError):
        hubconf.mars5_english(ckpt_format="onnx")
    # 3. huggingface route
    import huggingface_hub
    files = {"mars5_ar.safetensors": str(hub / "checkpoints" / hubconf.ar_sf_url.rsplit("/", 1)[1]),
             "mars5_nar.safetensors": str(hub / "checkpoints" / hubconf.nar_sf_url.rsplit("/", 1)[1])}
    monkeypatch.setattr(huggingface_hub, "hf_hub_download", lambda repo_id, filename, **kw: files[filename])
    got = []
    monkeypatch.setattr(inference.Mars5TTS, "__init__", lambda self, ar_ckpt, nar_ckpt, device=None, **kw: got.append((ar_ckpt, nar_ckpt, device)))
    inference.Mars5TTS.from_pretrained("CAMB-AI/MARS5-TTS", device="cpu")
    same(got[-1][0], b.ar_ckpt)
    same(got[-1][1], b.nar_ckpt)


def test_batch_entry_point_groups_by_length_and_restores_order(monkeypatch):
    """Host logic of ``tts_batch_from_codes`` (BASELINE config 3) without a GPU: requests are refined in groups of at
    most `nar_batch`, grouped by similar total NAR length, every request keeps ITS generator (seeded with its seed),
    at most `nar_in_flight` groups are open at once (each enqueued without waiting, landed oldest first), and results come
    back in request order."""
    import torch
    import inference
    from inference import InferenceConfig, Mars5TTS
    m = Mars5TTS.__new__(Mars5TTS)
    m.device = torch.device("cpu")
    m.default_T, m.diffusion_n_classes = 5, 1025
    m.codecnar = object()
    lens = [30, 5, 17, 9, 26, 12, 3]
    calls, seeds_seen = [], {}
    open_groups, peak = [], [0]

    def fake_ar_stage(pr, cfg, ar_noise, generator):
        i = pr["i"]
        seeds_seen[i] = generator.initial_seed()
        x = torch.full((1, lens[i], 8), i, dtype=torch.long)
        return torch.arange(lens[i]), (None, None, None, None, x, None), 2      # frames, batch tuple (x at [4]), skip_front

    def fake_batch(model, batches, diff, T, dsh=None, generators=None, wait=True, stream=None, **kw):
        ids = [int(b[4][0, 0, 0]) for b in batches]
        calls.append(ids)
        assert [g.initial_seed() for g in generators] == [1000 + i for i in ids]
        assert wait is False                                   # groups are enqueued, not waited for
        open_groups.append(ids)
        peak[0] = max(peak[0], len(open_groups))

        def land():
            assert open_groups[0] == ids                       # oldest first
            open_groups.pop(0)
            return [torch.full((1, lens[i] + 2, 8), i, dtype=torch.long) for i in ids]
        return land

    monkeypatch.setattr(Mars5TTS, "_prompt", lambda self, text, prompt_codec, ref_transcript, cfg, ref_handle=None: {"i": int(text)})
    monkeypatch.setattr(Mars5TTS, "_ar_stage_pr", lambda self, *a: fake_ar_stage(*a))
    monkeypatch.setattr(inference, "perform_batch_inference", fake_batch)
    n = len(lens)
    out = m.tts_batch_from_codes([str(i) for i in range(n)], [None] * n, [""] * n, InferenceConfig(), seeds=[1000 + i for i in range(n)], nar_batch=3)
    assert seeds_seen == {i: 1000 + i for i in range(n)}
    assert [len(c) for c in calls] == [3, 3, 1]
    assert peak[0] == 2 and not open_groups                    # nar_in_flight defaults to 2; everything landed
    flat = [i for c in calls for i in c]
    assert sorted(flat) == list(range(n)) and [lens[i] for i in flat] == sorted(lens)      # similar lengths share a pass
    for i, (frames, final) in enumerate(out):
        assert frames.shape[0] == lens[i] and final.shape == (lens[i], 8) and int(final[0, 0]) == i   # skip_front rows dropped, order kept


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every argument struct of include/mars5_hip.h as gcc lays it out (size and every field offset) against its
    ctypes mirror in mars5_tts_amd/_lib.py: a field added on one side only would silently shift the kernel arguments."""
    import ctypes
    import re
    import subprocess
    from mars5_tts_amd import _lib as L
    hdr = os.path.join(ROOT, "include", "mars5_hip.h")
    pairs = {"M5QkvScatter": L.QkvScatter, "M5AttnArgs": L.AttnArgs, "M5Prefetch": L.Prefetch, "M5GemvArgs": L.GemvArgs,
             "M5AttnDecodeArgs": L.AttnDecodeArgs, "M5SampleArgs": L.SampleArgs, "M5NarSampleArgs": L.NarSampleArgs,
             "M5ArMegaArgs": L.ArMegaArgs, "M5DeferredLN": L.DeferredLN, "M5RowTiles": L.RowTiles, "M5NarUniformArgs": L.NarUniformArgs,
             "M5PlanOp": L.PlanOp, "M5StagePlan": L.StagePlanC}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hdr}"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for ln in out.splitlines():
        cname, fname, val = ln.split()
        cls = pairs[cname]
        if fname == "size":
            assert ctypes.sizeof(cls) == int(val), (cname, ctypes.sizeof(cls), val)
        else:
            assert getattr(cls, fname).offset == int(val), (cname, fname, getattr(cls, fname).offset, val)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in pairs.values())
    # and no header struct is left without a mirror
    names = set(re.findall(r"^\} (M5\w+);", open(hdr).read(), flags=re.M))
    assert names == set(pairs), names ^ set(pairs)
