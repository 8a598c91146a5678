#!/usr/bin/env python3
"""Phase breakdown of the correspondence kernel (lom_debug_match_stamps): shader-clock stamps taken by
the first lane of every workgroup around the phases of its FIRST query, plus the workgroup's end.
The stamped instantiation drains outstanding memory operations at every stamp, so the phases add up
to more than the product kernel's time; the shape (which phase dominates) is what this is for."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lidar_odometry_demo_amd as lom  # noqa: E402
from lidar_odometry_demo_amd import synth  # noqa: E402

PHASES = ["load+transform", "probe 27 slots", "prefix->LDS", "scan candidates", "group min", "normal+store",
          "remaining queries"]

boxes = synth.make_boxes()
L = lom.capi.lib()
for name, (nb, naz, nmap) in {"C2": (16, 1800, 500_000), "C3": (64, 2048, 2_000_000)}.items():
    scan, _, _, _ = synth.make_scan(nb, naz, boxes=boxes)
    mp, mn = synth.make_map_points(nmap, boxes=boxes)
    g = lom.VoxelGrid(0.5, 20)
    g.addCloud(mp, mn)
    d = torch.from_numpy(scan).to("cuda:0")
    torch.cuda.synchronize()
    cap = 1 << 16
    st = np.zeros((cap, 8), dtype=np.uint64)
    nblk = C.c_uint32(0)
    for rep in range(int(os.environ.get("LOM_STAMP_REPS", "3"))):   # the last of a few launches: caches and TLBs warm, as in an align
        rc = L.lom_debug_match_stamps(g.handle, C.c_void_p(d.data_ptr()), d.shape[0], 12, lom.capi.f3((0, 0, 0)),
                                      lom.capi.f4((1, 0, 0, 0)), C.c_float(0.3),
                                      st.ctypes.data_as(C.POINTER(C.c_ulonglong)), cap, C.byref(nblk))
        assert rc == 0, rc
    st = st[: nblk.value].astype(np.int64)
    t0 = st[:, 0].min()
    # s_memtime ticks at the shader clock (MI355X_MICROARCH.md); 2.4 GHz nominal
    tick_us = 1.0 / 2400.0
    print(f"{name}: {nblk.value} workgroups; first start -> last end {(st[:, 7].max() - t0) * tick_us:.2f} us; "
          f"start spread {(st[:, 0].max() - t0) * tick_us:.2f} us", flush=True)
    d_ph = np.diff(st, axis=1) * tick_us
    for i, ph in enumerate(PHASES):
        col = d_ph[:, i]
        print(f"   {ph:18s} median {np.median(col):6.2f} us   p90 {np.percentile(col, 90):6.2f}   max {col.max():6.2f}")
    print(f"   {'workgroup total':18s} median {np.median((st[:, 7] - st[:, 0]) * tick_us):6.2f} us", flush=True)
