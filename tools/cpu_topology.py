#!/usr/bin/env python3
"""Which CPUs this process may run on and which of them are hyper-thread siblings (the GPU box grants a 16-CPU share)."""
import os
allowed = sorted(os.sched_getaffinity(0))
print("allowed", allowed)
seen = set()
for c in allowed:
    try:
        sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
    except OSError as e:
        sib = f"? ({e})"
    if sib not in seen:
        seen.add(sib)
        print(f"cpu{c}: siblings {sib}")
print("cores (distinct sibling sets among the allowed CPUs):", len(seen))
