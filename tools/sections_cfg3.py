"""Wall-time shares of the sections of the wave kernel's pass on the configs[2] table shape (needs the -DTMC_SECTION_TIMERS
build: TARDIS_MC_LIB=tardis_amd/libtardis_mc_hip_timers.so), plus passes and sweep rounds per launch.
   python tools/sections_cfg3.py n_packets key=value[,...] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd import synthetic
from tardis_amd.engine import Engine

NAMES = ["cold+refill+log", "epilogue", "macro walk", "finish", "fetch(+volleys)", "prologue", "sweep"]
P = int(float(sys.argv[1]))
shape = dict(n_shells=20, n_lines=500_000, line_interaction_type="macroatom")
if os.environ.get("EXP_SHAPE") == "config2":
    shape = dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch")
if os.environ.get("EXP_SHAPE") == "config2v":
    shape = dict(n_shells=20, n_lines=30_000, line_interaction_type="downbranch", n_vpackets=10)
if os.environ.get("EXP_SHAPE") == "config5":
    shape = dict(n_shells=100, n_lines=500_000, line_interaction_type="macroatom", n_vpackets=10)
prob = synthetic.make_problem(seed=1, n_packets=1, **shape)
eng = Engine(0)
eng.set_geometry(prob.geometry, prob.time_explosion)
eng.set_opacity(prob.opacity_state)
eng.set_config(prob.montecarlo_configuration, prob.spectrum_frequency_grid)
eng.create_blackbody_packets(P, float(prob.geometry.r_inner[0]), 1.0e4)
for spec in sys.argv[2:]:
    base = 0
    for kv in spec.split(","):
        k, v = kv.split("=")
        if k == "debug_flags":
            base = int(v)
        else:
            eng.set_option(k, int(v))
    tot = []
    for sec in range(7):
        eng.set_option("debug_flags", base | 64 | (sec << 8))
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        tot.append(eng.last_counters()["reserved"])
    ms = eng.last_kernel_times()["propagate_ms"]
    extra = {}
    for flag, name in ((16, "rounds"), (32, "passes")):
        eng.set_option("debug_flags", base | flag)
        eng.reset_estimators(); eng.propagate(); eng.synchronize()
        extra[name] = eng.last_counters()["reserved"]
    c = eng.last_counters()
    s = sum(tot)
    print(f"{spec}: propagate {ms:.2f} ms ({P / ms / 1e3:.2f} Mpkt/s); passes {extra['passes']:.3e} (events/pass {c['events'] / extra['passes']:.1f}), "
          f"sweep rounds {extra['rounds']:.3e} ({extra['rounds'] / extra['passes']:.1f} per pass); ticks/pass {s / extra['passes']:.0f}")
    for n, t in zip(NAMES, tot):
        print(f"   {n:18s} {100.0 * t / s:5.1f} %   ({t / extra['passes']:.0f} ticks per pass)", flush=True)
eng.close()
