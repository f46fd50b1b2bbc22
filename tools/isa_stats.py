#!/usr/bin/env python3
"""ISA statistics per kernel of a `hipcc --cuda-device-only -S` listing: flat vs global memory instructions, waitcnt shapes,
spills and scratch (from the .amdhsa metadata).  Usage: isa_stats.py file.s [substring-of-demangled-or-mangled-name ...]"""
import re
import subprocess
import sys


def main():
    path = sys.argv[1]
    pats = sys.argv[2:]
    text = open(path).read()
    # kernel bodies: "name:" ... "s_endpgm"; metadata in .amdgpu_metadata
    kernels = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z[\w$.]+):\s*(;.*)?$", line)
        if m and not line.startswith("."):
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None:
            kernels[cur].append(line)
            if line.strip().startswith(".end_amdhsa_kernel"):
                cur = None
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", blk)}
    names = list(kernels)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        if pats and not any(p in d or p in n for p in pats):
            continue
        body = kernels[n]
        c = lambda rx: sum(1 for l in body if re.search(rx, l.split(";")[0]))
        st = dict(
            insts=sum(1 for l in body if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")),
            flat_load=c(r"\bflat_load"), flat_store=c(r"\bflat_store"), flat_atomic=c(r"\bflat_atomic"),
            global_load=c(r"\bglobal_load"), global_store=c(r"\bglobal_store"), global_atomic=c(r"\bglobal_atomic"),
            scratch=c(r"\bscratch_"), buffer=c(r"\bbuffer_"), ds=c(r"\bds_"),
            wait_vm0_lgkm0=c(r"s_waitcnt vmcnt\(0\) lgkmcnt\(0\)"), wait_lgkm0=c(r"s_waitcnt lgkmcnt\(0\)"),
            wait_vmN=c(r"s_waitcnt vmcnt\([1-9]"), wait_vm0=c(r"s_waitcnt vmcnt\(0\)\s*$"),
            readlane=c(r"v_readlane|v_writelane"),
        )
        print(d[:200])
        print("   ", " ".join("%s=%d" % kv for kv in st.items()))
        print("   ", meta.get(n, {}))


if __name__ == "__main__":
    main()
