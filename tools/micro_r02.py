"""Memory-system ceilings for the access patterns of the macro-atom walk and the sweeps (design input; GPU box).
Every lane reads random naturally aligned blocks of 8 ... 128 bytes out of tables the size of the walk's tables."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd.engine import Engine  # noqa: E402

eng = Engine(0)
names = {4: "random 8 B per lane", 5: "16 lanes x 8 B coalesced (128 B per group)", 6: "random 16 B per lane", 7: "random 32 B per lane",
         8: "random 64 B per lane", 9: "random 128 B per lane", 10: "dependent chain of random 8 B loads"}
iters = 128
for mb in (12, 64, 160, 1600):
    n = mb * 1_000_000 // 8
    for blocks in (1024, 4096):
        for which in (4, 5, 6, 7, 8, 9, 10):
            ms = eng.debug_microbench(which, n, iters, blocks)
            ops = blocks * 256 * iters
            nbytes = {4: 8, 5: 8, 6: 16, 7: 32, 8: 64, 9: 128, 10: 8}[which]
            print(f"table {mb:5d} MB  blocks {blocks:5d} ({blocks * 4 / 256:.0f} waves/CU)  {names[which]:44s} {ms:9.3f} ms  "
                  f"{ops / ms / 1e6:8.2f} G blocks/s  {ops * nbytes / ms / 1e9:7.3f} TB/s"
                  + (f"  {ms * 1e6 / iters:8.1f} ns per dependent load" if which == 10 else ""), flush=True)
eng.close()
