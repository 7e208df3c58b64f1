"""TEST INFRASTRUCTURE: deterministic stand-ins for the two third-party models that bracket the hot path in ``tts()``
(Encodec analysis, Vocos synthesis -- both absent here and out of scope).  The same objects are handed to the
unmodified reference ``inference.py`` (golden generation, ``oracle/gen_golden.py --only tts``) and to this repo's
``Mars5TTS`` (GPU test), so everything BETWEEN them -- prompt construction, AR decode, BPE hand-off, NAR refinement,
prompt skipping, silence trim -- is compared end to end through the public ``tts()`` entry point."""
from __future__ import annotations

import math

import torch
from torch import nn


class FakeCodec(nn.Module):
    """``codec.encode(x)[0][0]`` -> (1, 8, n_frames) int64 codes, n_frames = samples // 320 (24 kHz / 75 Hz), values a
    fixed function of the frame index and the waveform's length (so different references give different codes)."""

    def set_target_bandwidth(self, bw: float) -> None:
        self.bandwidth = bw

    def encode(self, x: torch.Tensor):
        n = int(x.shape[-1]) // 320
        g = torch.Generator().manual_seed(1000 + n)
        codes = torch.randint(0, 1024, (1, 8, n), generator=g, dtype=torch.long)
        return [(codes.to(x.device), None)]


class FakeVocos(nn.Module):
    """``decode(codes_to_features(tokens))``: one 320-sample frame per code frame, a 200 Hz tone whose amplitude is a
    function of the frame's 8 codes, with the first three and last two frames silent (so the trim has work to do).
    ``last_tokens`` keeps the (n_q, T) codes it was given -- the final NAR output, which ``tts()`` does not return."""

    def __init__(self):
        super().__init__()
        self.last_tokens = None

    def codes_to_features(self, tokens: torch.Tensor) -> torch.Tensor:
        self.last_tokens = tokens.detach().cpu().clone()
        return tokens.to(torch.float32)

    def decode(self, features: torch.Tensor, bandwidth_id=None) -> torch.Tensor:
        amp = ((features.sum(0) % 11) - 5.0) / 10.0                      # (T,)
        if amp.numel() > 5:
            amp[:3] = 0.0
            amp[-2:] = 0.0
        t = torch.arange(amp.numel() * 320, device=features.device, dtype=torch.float32) / 24000.0
        return (amp.repeat_interleave(320) * torch.sin(2 * math.pi * 200.0 * t))[None]


class CpuStreamHooks:
    """Replays the reference's use of the GLOBAL CPU generator inside ``tts()`` for an engine that runs on a GPU:
    one Exp(1) vector of (V,) per executed AR loop iteration (torch.multinomial, ar_generate.py:115), then the NAR
    stage's randint / rand draws (diffuser.py:411,364-393), all from ONE seeded CPU generator in program order."""

    def __init__(self, seed: int, device):
        self.g = torch.Generator().manual_seed(seed)
        self.dev = device
        self._state0 = None
        self._V = 0

    def ar_noise(self, n_steps: int, V: int) -> torch.Tensor:
        self._state0, self._V = self.g.get_state(), V
        return torch.stack([torch.empty(V).exponential_(1, generator=self.g) for _ in range(n_steps)])

    def after_ar(self, n_iterations: int) -> None:
        self.g.set_state(self._state0)                                    # rewind, then consume what the reference consumed
        for _ in range(n_iterations):
            torch.empty(self._V).exponential_(1, generator=self.g)

    def randint(self, shape):
        return torch.randint(0, 1025, shape, dtype=torch.long, generator=self.g)

    def uniform(self, shape):
        return torch.rand(shape, generator=self.g).to(self.dev)


def cpu_standin_tts(tiny_vocab):
    """(Moved here from tests/test_sharding_gloo.py in round 4 so that ``bench.py --launch-check --workload c4`` can use it.)
    A Mars5TTS whose host logic is the product's (prompt assembly from wire-format ids, BPE hand-off, prompt skipping)
    and whose device stages (AR decode, token expansion kernel, NAR refinement) are deterministic CPU stand-ins driven by
    the global RNG (what run_sharded seeds)."""
    import io
    import inference as inf
    from mars5_tts_amd import minbpe

    def fake_begin(model, c_text, c_codes, T, dsh=None, div_mode=0, diff=None, **kw):
        return None

    def fake_ar(texttok, speechtok, codeclm, xx, ss_gen, first_codex_idx, max_len=1500, generator=None, **kw):
        n = min(int(max_len) - int(xx.shape[0]), 9 + int(ss_gen.shape[0]) % 5)
        n_text = len(texttok.vocab)
        new = torch.randint(n_text, n_text + 1024, (max(n, 0),), generator=generator)   # global generator (seeded per request by run_sharded) or the request's own
        return torch.cat([xx.cpu(), new])

    def fake_nar(model, batch, diff, T, dtype=None, retain_quant0=True, dsh=None, generator=None, session=None, **kw):
        c_codes, x = batch[1], batch[4]
        out = x.clone()
        out[..., 1:] = torch.randint(0, 1024, out[..., 1:].shape, generator=generator)
        if dsh.deep_clone:
            out = torch.cat([c_codes.to(out.dtype), out], dim=1)
        return out

    def fake_nar_batch(model, batches, diff, T, dsh=None, generators=None, wait=True, stream=None, **kw):
        outs = [fake_nar(model, b, diff, T, dsh=dsh, generator=g) for b, g in zip(batches, generators)]     # request i draws from ITS generator
        return outs if wait else (lambda: outs)

    def fake_expand(tokens, n_text, off, vals, max_run, stream=None):       # CPU stand-in of m5_expand_tokens (same CSR table)
        out = []
        for t in (tokens - n_text).clamp(min=0).tolist():
            out.extend(vals[int(off[t]):int(off[t + 1])].tolist())
        return torch.tensor(out, dtype=torch.long)

    inf.begin_inference, inf.ar_generate, inf.perform_simple_inference = fake_begin, fake_ar, fake_nar
    inf.perform_batch_inference = fake_nar_batch
    inf.ops.expand_tokens = fake_expand
    m = inf.Mars5TTS.__new__(inf.Mars5TTS)
    m.device = torch.device("cpu")
    m.codec = m.vocos = False
    m.texttok = minbpe.RegexTokenizer()
    m.texttok.load(io.BytesIO(tiny_vocab["texttok.model"].encode()))
    m.speechtok = minbpe.CodebookTokenizer()
    m.speechtok.load(io.BytesIO(tiny_vocab["speechtok.model"].encode()))
    m.n_vocab = len(m.texttok.vocab) + len(m.speechtok.vocab)
    m.n_text_vocab = len(m.texttok.vocab) + 1
    m.diffusion_n_classes = 1025
    m.codeclm = m.codecnar = None
    m.default_T, m.sr, m.latent_sr = 4, 24000, 75
    m._expansion = m.speechtok.expansion_table()
    return m, inf
