"""Time the SURVEY 8(f) rows that run on the device next to the hot path: packet source, spectrum reduction and
radiation-field update, at BASELINE config-2 sizes (and the host numpy equivalents beside them)."""
import json
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import spectrum, synthetic  # noqa: E402
from tardis_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
prob = synthetic.make_problem(seed=1, **{**synthetic.BASELINE_CONFIGS[2], "n_packets": 1000})
eng = Engine(0)
eng.set_option("track_last_interaction", 0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
out = {"n_packets": n}
radius = prob.geometry.r_inner[0]
for rep in range(3):
    t0 = time.perf_counter()
    eng.create_blackbody_packets(n, radius, 1.0e4)
    out["device_source_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
pc = synthetic.black_body_packets(n, radius, 1.0e4)
out["host_source_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
eng.set_packets(pc)
eng.synchronize()
out["host_to_device_upload_ms"] = 1e3 * (time.perf_counter() - t0)
eng.create_blackbody_packets(n, radius, 1.0e4)
eng.reset_estimators(); eng.propagate(); eng.synchronize()
t_sim = pc.time_of_simulation
for rep in range(3):
    t0 = time.perf_counter()
    sp = eng.packet_spectrum(t_sim)
    out["device_spectrum_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
res = eng.get_results(track_last_interaction=False, want_line_estimators=True)
out["download_results_ms"] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter()
he = spectrum.emitted_luminosity_histogram(res.output_nus, res.output_energies, t_sim, prob.spectrum_frequency_grid)
out["host_spectrum_ms"] = 1e3 * (time.perf_counter() - t0)
out["spectrum_rel_l2_device_vs_numpy"] = spectrum.relative_l2(sp["montecarlo_emitted_luminosity"], he)
g = prob.geometry
volume = 4.0 / 3.0 * np.pi * (g.r_outer**3 - g.r_inner**3)
for rep in range(3):
    t0 = time.perf_counter()
    rf = eng.radiation_field(t_sim, volume)
    out["device_radfield_ms"] = 1e3 * (time.perf_counter() - t0)
out["t_rad_range"] = [float(rf["t_radiative"].min()), float(rf["t_radiative"].max())]
print(json.dumps(out))
