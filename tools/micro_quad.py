"""Does the address unit charge per lane or per distinct line?  Modes 11 / 12 of the micro-benchmark (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd.engine import Engine
eng = Engine(0)
iters = 256
for mb in (1, 2, 8, 160):
    n = mb * 1_000_000 // 8
    for blocks in (1024, 4096):
        for which, name in ((11, "lane-private 64 B blocks (4 x 16 B per lane)"), (12, "quad-shared 64 B blocks (lane j reads piece j)")):
            ms = eng.debug_microbench(which, n, iters, blocks)
            ops = blocks * 256 * iters
            print(f"table {mb:4d} MB blocks {blocks:5d}  {name:48s} {ms:8.3f} ms  {ops / ms / 1e6:8.2f} G blocks/s  {ops * 64 / ms / 1e9:7.3f} TB/s", flush=True)
eng.close()
