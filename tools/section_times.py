"""Wall-time shares of the sections of the wave kernel's pass (needs a -DTMC_SECTION_TIMERS build):
   python tools/section_times.py n_packets key=value[,...] ..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine

EXTRA = int(os.environ.get("SEC_EXTRA", "0"))
NAMES = ["cold+refill+log", "epilogue", "macro walk", "finish", "fetch(+volleys)", "prologue", "sweep"]
kw = dict(synthetic.BASELINE_CONFIGS[int(os.environ.get("SEC_CONFIG", "2"))])
kw["n_packets"] = int(sys.argv[1])
prob = synthetic.make_problem(seed=1, **kw)
for spec in sys.argv[2:]:
    tot = []
    for sec in range(7):
        eng = Engine(0)
        for kv in spec.split(","):
            k, v = kv.split("=")
            eng.set_option(k, int(v))
        eng.set_option("debug_flags", 64 | (sec << 8) | EXTRA)
        eng.set_geometry(prob.geometry, prob.time_explosion); eng.set_opacity(prob.opacity_state)
        eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid); eng.set_packets(prob.packet_collection)
        for i in range(2):
            eng.reset_estimators(); eng.propagate(); eng.synchronize()
        tot.append(list(eng.last_counters().values())[7])
        ms = eng.last_kernel_times()["propagate_ms"]
        eng.close()
    s = sum(tot)
    print(f"{spec}: propagate {ms:.2f} ms")
    for n, t in zip(NAMES, tot):
        print(f"   {n:18s} {100.0 * t / s:5.1f} %   ({t:.3e} ticks)", flush=True)
