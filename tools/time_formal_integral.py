"""Time the device formal integral at the tardis_example shape (20 shells, 3e4 lines, 10 000 frequencies x 1000 impact
parameters = the reference's default `integrated` spectrum) and the CPU oracle on a slice of the same problem."""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import formal  # noqa: E402
from tardis_amd import synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n_nu = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
S, Ln = 20, 30_000
prob = synthetic.make_problem(seed=4, n_packets=1, n_shells=S, n_lines=Ln, log_tau_mean=-2.0)
rng = np.random.default_rng(5)
nu_l = prob.opacity_state.line_list_nu
bb = 2 * 6.62606957e-27 * 3.33564e-11**2 * nu_l**3 / np.expm1(6.62606957e-27 * nu_l / (1.3806488e-16 * 1e4))
w = 0.5 * (prob.geometry.r_inner[0] / prob.geometry.r_outer) ** 2
jblue = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (S, Ln))).ravel()
jred = (bb[None, :] * w[:, None] * rng.uniform(0.5, 1.5, (S, Ln))).ravel()
att = (bb[None, :] * w[:, None] * rng.uniform(0.2, 1.2, (S, Ln)) * (1 - np.exp(-prob.opacity_state.tau_sobolev.T))).ravel()
freqs = np.linspace(2.99792458e10 / 20000e-8, 2.99792458e10 / 500e-8, n_nu)   # the spectrum grid of tardis_example
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
out = {"n_frequencies": n_nu, "n_impact_parameters": N, "n_shells": S, "n_lines": Ln}
for rep in range(3):
    t0 = time.perf_counter()
    L, _ = eng.formal_integral(1.0e4, freqs, att, jred, jblue, N)
    out["device_call_ms"] = 1e3 * (time.perf_counter() - t0)
    out["device_kernels_ms"] = eng.last_propagate_ms()
out["line_steps"] = eng.last_counters()["line_visits"]
out["GB_per_s_at_40B_per_step"] = 40.0 * out["line_steps"] / (out["device_kernels_ms"] * 1e-3) / 1e9
sub = slice(0, n_nu, max(n_nu // 100, 1))
t0 = time.perf_counter()
Lo, _ = formal.formal_integral(prob.geometry.r_inner, prob.geometry.r_outer, prob.time_explosion, nu_l, prob.opacity_state.tau_sobolev,
                               prob.opacity_state.electron_density, 1.0e4, freqs[sub], att, jred, jblue, N)
dt = time.perf_counter() - t0
out["cpu_oracle_1thread_ms_full_problem_extrapolated"] = 1e3 * dt * n_nu / len(freqs[sub])
out["max_rel_diff_vs_oracle_sample"] = float(np.max(np.abs(L[sub] - Lo) / np.abs(Lo)))
print(json.dumps(out))
