"""TEST INFRASTRUCTURE (like everything under oracle/): shared checker for integer (sampled id) comparisons: equality, except where the ORACLE's own top-2 scores are closer
than the float noise of the formula.  The sampled id is an arg-max over 1025 perturbed scores
``log p + Gumbel(u)`` (reference diffuser.py:219-228); kernel and oracle evaluate the same fp32 formula with differently
ordered reductions (log-sum-exp over 1025 terms), so scores agree to a few ulp and an id can only differ where two
classes are within that distance -- a mismatch anywhere else is a bug.  Used by tests/ and __graft_entry__.smoke() only."""
import torch

SCORE_EPS = 5e-5        # scores reach |log 1e-7| = 16.1 (fp32 ulp there 1.9e-6); a handful of ulp through log-sum-exp + log


def ungated_mismatches(out: torch.Tensor, ref: torch.Tensor, s_unk: torch.Tensor, s_kn, m: torch.Tensor, eps: float = SCORE_EPS):
    """out / ref / m: (S, Q) ids and the known-mask; s_unk / s_kn: (S, Q, K) oracle scores of the two branches (s_kn None
    at t = 0, where the known branch is a copy).  Returns (n_mismatch, [(row, q, oracle_gap), ...] of the mismatches the
    oracle's own margin does NOT excuse)."""
    out, ref = out.to(ref.device), ref
    diff = (out != ref).nonzero().tolist()
    bad = []
    for r, q in diff:
        sc = s_kn if bool(m[r, q]) else s_unk
        if sc is None:
            bad.append((r, q, float("inf")))
            continue
        gap = float(sc[r, q, int(ref[r, q])] - sc[r, q, int(out[r, q])])
        if not gap <= eps:
            bad.append((r, q, gap))
    return len(diff), bad


def gate_trajectory(O, sd, nhead, c_text, c_codes, x_known, m, traj, guidance_w, temperature, q0_override_steps, eps, device):
    """Replay an ENGINE trajectory step by step against the oracle: `traj` = the dicts ``NARSession.run(on_step=...)``
    yields.  For each step the oracle (on `device`) maps the engine's own x_t and uniforms to its x_{t-1} and scores; an
    engine id that differs is excused only by an oracle tie (`eps` on the scores).  If nothing is unexcused, the engine's
    trajectory is one the reference arithmetic could have produced.  Returns (n_differing_ids, unexcused list)."""
    n_diff, bad = 0, []
    with torch.device(device), torch.inference_mode():
        sd = {k: v.to(device) for k, v in sd.items()}
        c_text, c_codes = c_text.to(device), c_codes.to(device)
        x_known, mb = x_known.to(device), m.to(device).bool()
        tb = O.diffusion_tables(1025, 200)
        spk = [O.nar_spk_vector(sd, c_codes, nhead, False), O.nar_spk_vector(sd, c_codes, nhead, True)]
        for rec in traj:
            t, x_t = rec["t"], rec["x_t"].to(device)
            lc = O.nar_forward(sd, nhead, c_text, c_codes, x_t, t, False, spk[0])
            lu = O.nar_forward(sd, nhead, c_text, c_codes, x_t, t, True, spk[1])
            u2 = rec["u2"].to(device) if rec["u2"] is not None else None
            ref, s_unk, s_kn = O.reverse_step(tb, lc, lu, x_t, x_known, mb, t, rec["u1"].to(device), u2, guidance_w, temperature,
                                              return_scores=True)
            if q0_override_steps < t:
                ref[:, 0] = x_known[:, 0]
            n, b = ungated_mismatches(rec["x_tm1"], ref, s_unk, s_kn, mb, eps)
            n_diff += n
            bad += [(rec["i"], t) + e for e in b]
    return n_diff, bad
