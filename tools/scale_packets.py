import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine
for n in (2_500_000, 5_000_000, 10_000_000, 20_000_000):
    kw = dict(synthetic.BASELINE_CONFIGS[2]); kw["n_packets"] = n
    prob = synthetic.make_problem(seed=1, **kw)
    eng = Engine(0)
    eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
    eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
    best = 1e9
    for i in range(3):
        eng.reset_estimators(); eng.propagate(); eng.synchronize(); best = min(best, eng.last_propagate_ms())
    print(n, best, eng.last_kernel_times(), n / best / 1e3, flush=True)
    eng.close()
