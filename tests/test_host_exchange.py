"""The shared-memory exchange between the ranks of one node (lom_host_comm_*): plain host code in
the product library, so the N>1 data path is exercised here with real processes and no GPU.

* stress: 3 ranks, thousands of back-to-back all-reduces with random stalls -- every rank gets the
  same bits every time (double buffering holds), and they equal the rank-ordered sum;
* sharded align: the product's host driver over range-sharded source points with this exchange as
  the `allreduce` hook and the oracle's evaluators standing in for the HIP kernels, against the
  oracle's single-process align (pose bar 1e-4 m / 1e-4 rad)."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _stress_worker(rank, world, ident, n_iter, q):
    sys.path.insert(0, ROOT)
    import lidar_odometry_demo_amd as lom

    L = lom.capi.lib()
    h = C.c_void_p()
    assert L.lom_host_comm_create(rank, world, ident, C.byref(h)) == 0
    rng = np.random.default_rng(100 + rank)
    all_rng = [np.random.default_rng(100 + r) for r in range(world)]
    ok = True
    digest = 0.0
    for it in range(n_iter):
        vals = [g.standard_normal(32) for g in all_rng]          # every rank can predict every contribution
        mine = rng.standard_normal(32)
        assert np.array_equal(mine, vals[rank])
        buf = (C.c_double * 32)(*mine)
        if it % 97 == rank:                                      # uneven arrival
            for _ in range(20000):
                pass
        assert L.lom_host_comm_allreduce(h, buf, 32) == 0
        want = np.zeros(32)
        for r in range(world):
            want = want + vals[r]                                # rank order, like the library
        got = np.array(buf[:])
        ok = ok and np.array_equal(got, want)
        digest += float(got.sum())
    L.lom_host_comm_destroy(h)
    q.put((rank, ok, digest))


def _gather_worker(rank, world, ident, n_iter, q):
    sys.path.insert(0, ROOT)
    import lidar_odometry_demo_amd as lom

    L = lom.capi.lib()
    h = C.c_void_p()
    assert L.lom_host_comm_create(rank, world, ident, C.byref(h)) == 0
    ok = True
    for it in range(n_iter):
        nbytes = 1 + (it * 37) % 256                               # every size up to a full slot
        mine = bytes((rank * 31 + it + k) & 0xFF for k in range(nbytes))
        out = C.create_string_buffer(world * nbytes)
        assert L.lom_host_comm_allgather(h, mine, nbytes, out) == 0
        for r in range(world):
            want = bytes((r * 31 + it + k) & 0xFF for k in range(nbytes))
            ok = ok and out.raw[r * nbytes:(r + 1) * nbytes] == want
        if it % 5 == 0:                                            # interleaved with all-reduces: one sequence
            buf = (C.c_double * 1)(float(rank))
            assert L.lom_host_comm_allreduce(h, buf, 1) == 0
            ok = ok and buf[0] == float(sum(range(world)))
    L.lom_host_comm_destroy(h)
    q.put((rank, ok))


def _align_worker(rank, world, ident, q):
    sys.path.insert(0, ROOT)
    import lidar_odometry_demo_amd as lom
    from oracle import oracle as O
    from tests import scenes

    L = lom.capi.lib()
    h = C.c_void_p()
    assert L.lom_host_comm_create(rank, world, ident, C.byref(h)) == 0
    sm = scenes.small_synth_case()
    scan = sm["scan"]
    lo, hi = len(scan) * rank // world, len(scan) * (rank + 1) // world
    grid = O.VoxelGrid(0.5, 20)
    grid.addCloud(sm["map_xyz"], sm["map_nrm"])
    shard = O.Shard(grid, np.ascontiguousarray(scan[lo:hi]))
    OL = O.lib()
    me = lom.capi.MATCH_EVAL_FN(C.cast(OL.orc_shard_match_eval, C.c_void_p).value)
    ef = lom.capi.EVAL_FIXED_FN(C.cast(OL.orc_shard_eval_fixed, C.c_void_p).value)
    ar = lom.capi.ALLREDUCE_FN(lambda user, buf, count: L.lom_host_comm_allreduce(h, buf, count))
    hooks = lom.capi.AlignHooks(shard.handle, me, ef, ar)
    ot, oq = (C.c_float * 3)(), (C.c_float * 4)()
    st = lom.capi.AlignStats()
    rc = L.lom_align_with_hooks(C.byref(hooks), lom.capi.f3((0.02, -0.01, 0.0)),
                                lom.capi.f4(scenes.angle_axis_q(0.004, (0, 0, 1))), ot, oq, C.byref(st))
    L.lom_host_comm_destroy(h)
    q.put((rank, rc, list(ot), list(oq), st.outer_iterations, st.queries))


def _run(target, world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ident = os.urandom(16) + bytes(112)
    procs = [ctx.Process(target=target, args=(r, world, ident) + extra + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_host_exchange_stress_three_ranks():
    res = _run(_stress_worker, 3, 3000)
    assert all(ok for _, ok, _ in res)
    assert len({d for _, _, d in res}) == 1          # bitwise the same history on every rank


def test_host_allgather_bytes_three_ranks():
    """Raw-byte all-gather (carries the IPC handles of the device-to-device exchange), mixed with
    all-reduces on the same object."""
    res = _run(_gather_worker, 3, 400)
    assert all(ok for _, ok in res)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_align_over_host_exchange(world, oracle):
    from tests import scenes

    res = _run(_align_worker, world)
    assert all(rc == 0 for _, rc, *_ in res)
    poses = {tuple(t) + tuple(qq) for _, _, t, qq, _, _ in res}
    assert len(poses) == 1                            # every rank ends with the same pose, bit for bit
    _, _, t, qq, outer, queries = res[0]
    sm = scenes.small_synth_case()
    g = oracle.VoxelGrid(0.5, 20)
    g.addCloud(sm["map_xyz"], sm["map_nrm"])
    m = oracle.CloudMatcher()
    ref = m.align(g, sm["scan"], oracle.Pose3D((0.02, -0.01, 0.0), scenes.angle_axis_q(0.004, (0, 0, 1))))
    dt, dr = scenes.pose_delta(t, qq, ref.translation, ref.rotation)
    assert dt < 1e-4 and dr < 1e-4, (dt, dr)
    assert outer == m.stats["outer_iterations"] and queries == m.stats["queries"]


# ---- an exchange nobody completes is ABANDONED, and every rank sees that -----------------------------------
# Round 2's three-rank failure: ranks 0 and 1 gave up on an exchange after their deadline
# but left their slots published; rank 2 arrived later, found all three slots at its sequence number, "succeeded"
# and carried on alone.  Now a rank that walks away marks its slots, and whoever pairs with them fails as well.

def _abandon_worker(rank, world, ident, mode, q):
    sys.path.insert(0, ROOT)
    import time

    import lidar_odometry_demo_amd as lom

    L = lom.capi.lib()
    h = C.c_void_p()
    assert L.lom_host_comm_create(rank, world, ident, C.byref(h)) == 0
    buf = (C.c_double * 2)(1.0, float(rank))
    assert L.lom_host_comm_allreduce(h, buf, 2) == 0            # exchange 1: everybody is there
    assert buf[0] == float(world)
    out = {}
    if mode == "late":
        # exchange 2: the last rank arrives after the others' deadline (0.4 s) has passed
        L.lom_host_comm_set_timeout(h, C.c_double(0.4 if rank != world - 1 else 30.0))
        if rank == world - 1:
            time.sleep(1.5)
        t0 = time.time()
        buf = (C.c_double * 1)(1.0)
        out["rc2"] = L.lom_host_comm_allreduce(h, buf, 1)
        out["t2"] = time.time() - t0
    else:
        # exchange 2: rank 0 cannot continue and says so; the others are already waiting for it
        L.lom_host_comm_set_timeout(h, C.c_double(30.0))
        if rank == 0:
            time.sleep(0.5)
            assert L.lom_host_comm_abort(h) == 0
            out["rc2"], out["t2"] = -6, 0.0
        else:
            t0 = time.time()
            buf = (C.c_double * 1)(1.0)
            out["rc2"] = L.lom_host_comm_allreduce(h, buf, 1)
            out["t2"] = time.time() - t0
    out["err"] = L.lom_host_comm_last_error(h).decode()
    t0 = time.time()
    buf = (C.c_double * 1)(1.0)
    out["rc3"] = L.lom_host_comm_allreduce(h, buf, 1)           # and the object stays broken, at once
    out["t3"] = time.time() - t0
    big = C.create_string_buffer(world * 8)
    out["rc4"] = L.lom_host_comm_allgather(h, b"12345678", 8, big)
    L.lom_host_comm_destroy(h)
    q.put((rank, out))


@pytest.mark.parametrize("mode", ["late", "abort"])
def test_abandoned_exchange_fails_on_every_rank(mode):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ident = os.urandom(16) + bytes(112)
    procs = [ctx.Process(target=_abandon_worker, args=(r, world, ident, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert o["rc2"] == -6 and o["rc3"] == -6 and o["rc4"] == -6, (r, o)     # LOM_ERR_COMM, nobody carries on alone
        assert o["t3"] < 0.2, (r, o)
    if mode == "late":
        assert "timed out waiting for rank 2" in res[0]["err"] and "timed out waiting for rank 2" in res[1]["err"]
        assert "abandoned exchange" in res[2]["err"], res[2]                    # the late rank does not pair with stale slots
        assert res[2]["t2"] < 0.5                                               # ... and learns it at once, not after 30 s
    else:
        assert all("abandoned exchange" in res[r]["err"] for r in (1, 2)), res
        assert all(res[r]["t2"] < 5.0 for r in (1, 2)), res                     # at once, not after the 30 s deadline
