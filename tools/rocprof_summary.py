"""Summarise rocprofv3 rocpd (.db) outputs into text: per-kernel time statistics and PMC counter sums.
    python tools/rocprof_summary.py <dir with trace/ and pmc_*/ sub-directories> > profiles/<name>.txt
"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"{'Name':70s} {'Calls':>6s} {'TotalDurationNs':>16s} {'AverageNs':>14s} {'Percentage':>10s} {'MinNs':>12s} {'MaxNs':>12s}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{name[:70]:70s} {n:6d} {tot:16d} {avg:14.1f} {100.0 * tot / total:10.2f} {mn:12d} {mx:12d}")
    q = ("select s.kernel_name, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, s.group_segment_size, s.private_segment_size "
         "from rocpd_info_kernel_symbol s where s.kernel_name like '%propagate_wave%' or s.kernel_name like '%seed%' or s.kernel_name like '%accumulate%' or s.kernel_name like '%bin_%' or s.kernel_name like '%fi_%'")
    for r in c.execute(q):
        print("resources:", r)


def pmc_sums(db):
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, p.name, count(distinct d.dispatch_id), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, p.name")
    for name, counter, n, total in c.execute(q):
        if any(k in name for k in ("propagate", "seed", "accumulate", "bin_", "packet_source", "spectrum", "radfield", "fi_")):
            print(f"{name[:60]:60s} {counter:24s} dispatches={n:3d} sum_over_dispatches={total:.6g} per_dispatch={total / n:.6g}")


def traffic_json(root, out_path):
    """HBM bytes per launch of the propagation kernel from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md (HBM) prescribes: counters are in KiB and FETCH_SIZE under-reports coalesced reads by 2x."""
    import json
    vals = {}
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
        c = sqlite3.connect(db)
        q = ("select s.kernel_name, p.name, count(distinct d.dispatch_id), sum(e.value) from rocpd_pmc_event e "
             "join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id "
             "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, p.name")
        for name, counter, n, total in c.execute(q):
            if ("propagate_group" in name or "propagate_wave" in name) and counter in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[counter] = total / n
    if len(vals) == 2:
        hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        with open(out_path, "w") as f:
            json.dump({"hbm_bytes_per_launch": hbm, "FETCH_SIZE_KiB_per_launch": vals["FETCH_SIZE"],
                       "WRITE_SIZE_KiB_per_launch": vals["WRITE_SIZE"],
                       "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024; propagation kernel (propagate_wave_kernel / propagate_group_kernel), default bench workload"}, f)


def main():
    root = sys.argv[1]
    if len(sys.argv) > 2:
        traffic_json(root, sys.argv[2])
    for db in sorted(glob.glob(os.path.join(root, "trace*", "*.db"))):
        print(f"== kernel trace: {os.path.relpath(db, root)}")
        kernel_stats(db)
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "*.db"))):
        print(f"== counters: {os.path.relpath(db, root)}  (value summed over all hardware instances of the counter)")
        pmc_sums(db)


if __name__ == "__main__":
    main()
