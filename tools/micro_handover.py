"""What would re-binning packets between the lanes of the grid cost?  (GPU box)

Two numbers on the same chip, same lanes in flight (4096 waves of 64, 16 per CU -- the propagation kernel's grid):
  * the rate at which lanes read random 64-byte sectors (the currency of the propagation kernel: it is bound by the number of
    sector requests that leave the L2);
  * the rate at which lanes hand a packet over through per-(shell, tile) queues: append a 128- or 64-byte record to a random
    one of 4900 bins (one returning atomic on the bin's cursor) and take one over from another bin.
Their ratio prices a hand-over in sector requests; tools/locality_stats.py says how many sector requests per trace a re-binned
wave could share at most.
    python tools/micro_handover.py > gpurun_out/micro_handover.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tardis_amd.engine import Engine  # noqa: E402

eng = Engine(0)
blocks, iters = 1024, 2000  # 1024 x 256 threads = 4096 waves
lanes = blocks * 256
n = 1 << 27  # 1 GiB table
rows = []
for which, name, sectors in ((8, "random 64-byte reads, 16-byte loads (one sector each)", 1), (9, "random 128-byte reads", 2),
                             (13, "hand-over, 128-byte records: atomic + 2 sectors written + 2 read", 4),
                             (14, "hand-over, 64-byte records: atomic + 1 sector written + 1 read", 2)):
    ms = min(eng.debug_microbench(which, n, iters, blocks) for _ in range(2))
    rate = lanes * iters / (ms * 1e-3)
    rows.append((name, ms, rate))
    print(f"{name:75s} {ms:9.2f} ms   {rate / 1e9:7.2f} G operations/s")
read64 = rows[0][2]
print()
for name, ms, rate in rows[2:]:
    print(f"{name:75s} = {read64 / rate:5.2f} random 64-byte sector reads each")
eng.close()
