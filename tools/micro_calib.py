"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE on the access patterns of the propagation kernel
(MI355X_MICROARCH.md, HBM: the counter is only calibrated for wide coalesced streaming reads).  Every lane reads random,
naturally aligned blocks of 16 / 32 / 64 / 128 bytes with 16-byte loads out of a 1.6 GB table (no cache reuse to speak of):
bytes requested per launch = blocks x 256 x iters x block size.  Run under `rocprofv3 --pmc FETCH_SIZE`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tardis_amd.engine import Engine  # noqa: E402

eng = Engine(0)
blocks, iters, n = 4096, 64, 1_600_000_000 // 8
for which, size in ((6, 16), (7, 32), (8, 64), (9, 128)):
    ms = eng.debug_microbench(which, n, iters, blocks)  # (two launches: warm-up + timed)
    print(f"microbench mode {which}: {blocks * 256 * iters} blocks of {size} B per launch = {blocks * 256 * iters * size / 1e9:.3f} GB; {ms:.3f} ms", flush=True)
eng.close()
