"""Test-only adapter: evaluate libpfgrad problem dicts with the CPU oracle.

Used by the CPU (`-m "not gpu"`) tests to exercise the *host* logic of the samplers (window
sampling, RNG order, prior gradients, updates, projections) against the reference's golden
trajectories without a GPU, by monkeypatching `sgmcmc_ssm_amd.particle_filters.run_windows`.
The product never imports this."""
from oracle import pf_oracle as po


def run_windows_oracle(problems, ctx=None, want_final=False):
    outs = []
    for q in problems:
        if q["rng"] != "replay":
            raise ValueError("the oracle can only replay host streams")
        pf = {"filter": "filter", "poyiadjis_n2": "poyiadjis_N2"}.get(q["smoother"], "nemeth")
        if q["stat"] == "predictive":
            pz = q.get("pred_z")
            r = po.pf_window(q["model"], q["theta"], q["y"], q["N"], q["z0"], q["u"], q["z"],
                             kernel=q["kernel"], pf="filter", stat="predictive", t1=q["t1"], tL=q["tL"],
                             weights=q["weights"], prior_mean=q["prior_mean"], prior_var=q["prior_var"],
                             num_steps_ahead=q["num_steps_ahead"],
                             pred_normals=(lambda t, k: None) if pz is None else (lambda t, k: pz[t, k]))
            outs.append(r)
            continue
        r = po.pf_window(q["model"], q["theta"], q["y"], q["N"], q["z0"], q["u"], q["z"],
                         kernel=q["kernel"], pf=pf, lambduh=q["lambduh"], stat=q["stat"],
                         t1=q["t1"], tL=q["tL"], weights=q["weights"],
                         prior_mean=q["prior_mean"], prior_var=q["prior_var"])
        outs.append(r)
    return outs
