"""Run one golden case with engine options and report what differs from the golden:  python tools/dbg_case.py case [key=value,...] ..."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import _golden
from tardis_amd import state as st, transport
from tardis_amd.engine import Engine

name = sys.argv[1]
for spec in sys.argv[2:] or [""]:
    prob, g = _golden.load_case(name)
    eng = Engine(0)
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("="); eng.set_option(k, int(v))
    pc = prob.packet_collection
    trk = st.LastInteractionTrackers(pc.number_of_packets)
    try:
        transport.montecarlo_transport_with_vpackets(pc, prob.geometry, prob.time_explosion, prob.opacity_state, prob.montecarlo_configuration,
            prob.spectrum_frequency_grid, trk, prob.montecarlo_configuration.NUMBER_OF_VPACKETS, False, None, engine=eng)
    except Exception as e:
        print(spec, "EXC", e); continue
    keys = [k for k in g.files if k.startswith("ref_")] if hasattr(g, "files") else list(g.keys())
    out = []
    for f in st.LastInteractionTrackers.I64_FIELDS:
        for cand in (f"ref_trk_{f}", f"trk_{f}", f"ref_{f}", f):
            if cand in g:
                same = np.array_equal(getattr(trk, f), g[cand]); out.append(f"{f}:{'ok' if same else 'DIFF'}"); break
    print(f"[{spec}] nu_equal={np.array_equal(pc.output_nus, g['ref_output_nus']) if 'ref_output_nus' in g else '?'}", " ".join(out), flush=True)
    if not out: print("   golden keys:", list(g.keys())[:40])
    eng.close()
