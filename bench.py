#!/usr/bin/env python3
"""Headline benchmark: ICP scan-match throughput (BASELINE.json metric) on synthetic
VLP16-shaped clouds, on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # starts its own N rank processes (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one CloudMatcher::align of one scan against the resident keyframe map
(<=35 outer iterations of {27-neighbour correspondence search, <=4 LM iterations}).
N=1 workload = BASELINE.json configs[1] ("C2"): 16 beams x 1800 azimuth steps vs a
500k-point map, 0.5 m voxels, cap 20.  N>1 = weak scaling: a (16*N)-beam scan in
firing (azimuth-major) order split into N contiguous index ranges, map replicated,
one RCCL all-gather of 32 f64 per residual evaluation.

The unit of `value` is one correspondence query (one source point searched against
its 27 neighbour voxels and reduced into the normal equations), counted over all
outer iterations and all ranks.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# same guide, "Indexed rows": uniformly random rows of a 38 MB table are served from the Infinity Cache at 8.6 TB/s
# chip-wide -- the level the C3 / C4 maps (48 MB payload + table) are read from
CACHE_PEAK_GBS = 8600.0
EVENT_PERIOD = 32      # HIP events around the k_match / k_lm launches of every 32nd step (see event_period)
EXTRA_BLOCKS = 5       # further blocks of K steps after the official one: median and spread of ms_per_step


_REAL_STDOUT = None


def claim_stdout():
    """Libraries underneath (RCCL's init banner, HIP warnings) printf to stdout; the contract is ONE
    JSON line there.  Everything else is sent to stderr; emit() writes the line to the real stdout."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


_T_START = time.perf_counter()


def progress(what):
    """LOM_BENCH_PROGRESS=1: where the run is, on stderr (the line itself stays the only thing on stdout)"""
    if os.environ.get("LOM_BENCH_PROGRESS"):
        print(f"[bench {time.perf_counter() - _T_START:7.1f} s] {what}", file=sys.stderr, flush=True)


def emit(line):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(line) + "\n").encode())


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher around it: this process -- which has not imported torch and has
    made no HIP call, and never will -- starts N fresh rank processes of this same file (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set as torch.distributed.run would set them), relays rank 0's ONE JSON line to its own stdout, and exits
    non-zero if any rank did.  Nothing is re-executed in a process that touched the GPU.  When one rank dies the
    others are ended by their exact pids (they would wait for it in a collective otherwise)."""
    import signal
    import socket
    import subprocess

    with socket.socket() as s:      # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # rank 0's stdout is the line; the other ranks print nothing there (their stdout joins stderr)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))

    def end_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)    # each rank is the leader of a session of its own
                except OSError:
                    pass
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass

    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("LOM_BENCH_LAUNCH_TIMEOUT_S", "1500"))
    t0 = time.time()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print(f"[bench] rank(s) {bad} failed; ending the others", file=sys.stderr)
                rc = 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > limit:
                print(f"[bench] ranks still running after {limit:.0f} s; ending them", file=sys.stderr)
                rc = 1
                break
            time.sleep(0.05)
    finally:
        end_all()
    reader.join(timeout=5)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        print("[bench] rank 0 printed no line", file=sys.stderr)
        rc = 1
    return rc


class Watchdog:
    """A side block (the other transport, the teardown) must not cost the line: if it is not through within `seconds`,
    `on_fire` runs (rank 0 emits the line it already holds) and the process leaves with os._exit -- a rank stuck in a
    collective some other rank never entered cannot be brought back any other way."""

    def __init__(self, seconds, on_fire):
        import threading

        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._on_fire = on_fire
        self._t.start()

    def _fire(self):
        try:
            self._on_fire()
        finally:
            sys.stderr.flush()
            os._exit(0)

    def cancel(self):
        self._t.cancel()


def build_workload(n_gpus, rank, config="C2"):
    from lidar_odometry_demo_amd import synth

    boxes = synth.make_boxes()
    n_beams = 16 * n_gpus
    if config == "C3":
        # BASELINE.json configs[2]: 64 beams x 2048 azimuth steps vs a 2M-point map (parity-tested in
        # tests/test_gpu_parity.py::test_c3_full_size_properties); not the default bench line
        scan, ring, az, _ = synth.make_scan(64, 2048, boxes=boxes)
        map_xyz, map_nrm = synth.make_map_points(2_000_000, boxes=boxes)
        return dict(scan=scan, shard=scan, map_xyz=map_xyz, map_nrm=map_nrm, config="C3",
                    name="C3 (BASELINE configs[2]): 64x2048 scan vs 2M-pt map, voxel 0.5 m, cap 20")
    if config == "C4":
        # BASELINE.json configs[3]: 128 beams x 2048 azimuth steps (<= 262,144 returns) vs the 2M-point map
        # replicated on every GPU; the scan is split into `n_gpus` contiguous index ranges (SURVEY.md 8d:
        # 8 x 32,768 on 8 GPUs; "also run with 1/2/4 ranks"), one exchange of the 32 reduced sums per
        # residual evaluation.  Total work is fixed: strong scaling.
        scan, ring, az, _ = synth.make_scan(128, 2048, boxes=boxes)
        map_xyz, map_nrm = synth.make_map_points(2_000_000, boxes=boxes)
        lo, hi = len(scan) * rank // n_gpus, len(scan) * (rank + 1) // n_gpus
        shard = np.ascontiguousarray(scan[lo:hi])
        return dict(scan=scan, shard=shard, map_xyz=map_xyz, map_nrm=map_nrm,
                    name=f"C4: 128x2048 scan ({len(scan)} returns) in {n_gpus} contiguous index range(s) "
                         f"vs replicated 2M-pt map, voxel 0.5 m, cap 20")
    if n_gpus == 1:
        scan, ring, az, _ = synth.make_scan(16, 1800, boxes=boxes)
        shard = scan
        name = "C2 (BASELINE configs[1]): VLP16 16x1800 scan vs 500k-pt map, voxel 0.5 m, cap 20"
    else:
        el = synth.beam_elevations(16)
        # 16*N beams spanning the VLP16 elevation range, firing (azimuth-major) order
        saved = synth.beam_elevations
        synth.beam_elevations = lambda n: np.linspace(el[0], el[-1], n)
        try:
            scan, ring, az, _ = synth.make_scan(n_beams, 1800, boxes=boxes)
        finally:
            synth.beam_elevations = saved
        # sector-major order: the scan is stored as N azimuth sectors of 360/N degrees, beam-major inside
        # a sector, and rank r takes the r-th N-th of that array (a contiguous index range).  Consecutive
        # queries are then neighbours along a beam, as in the N = 1 workload, and a rank touches one
        # sector of the map.
        sector = az.astype(np.int64) * n_gpus // 1800
        order = np.lexsort((az, ring, sector))
        scan = scan[order]
        lo, hi = len(scan) * rank // n_gpus, len(scan) * (rank + 1) // n_gpus
        shard = np.ascontiguousarray(scan[lo:hi])
        name = (f"C2 (BASELINE configs[1]) weak-scaled: {n_beams}x1800 scan, sector-major, in {n_gpus} contiguous index ranges "
                f"vs replicated 500k-pt map, voxel 0.5 m, cap 20")
    map_xyz, map_nrm = synth.make_map_points(500_000, boxes=boxes)
    return dict(scan=scan, shard=shard, map_xyz=map_xyz, map_nrm=map_nrm, name=name)


def cpu_baseline(work, budget_s=10.0):
    """The CPU restatement (oracle, kind "port") on the same workload, on this box's host
    cores: search over all cores, solve single-threaded (mirrors std::execution::par +
    Ceres num_threads=1).  Bounded sample: whole frames until ~budget_s."""
    from oracle import oracle as O

    # the GPU box grants a 16-CPU share per GPU whatever the affinity mask says
    cores = min(len(os.sched_getaffinity(0)), 16)
    g = O.VoxelGrid(0.5, 20)
    g.addCloud(work["map_xyz"], work["map_nrm"])
    m = O.CloudMatcher(nthreads=cores)
    frames, queries = 0, 0
    t0 = time.perf_counter()
    while True:
        m.align(g, work["scan"], O.Pose3D())
        frames += 1
        queries += m.stats["queries"]
        el = time.perf_counter() - t0
        if el > budget_s or frames >= 2000:
            break
    # also a 1-thread total (a few frames)
    m1 = O.CloudMatcher(nthreads=1)
    f1, q1 = 0, 0
    t1 = time.perf_counter()
    while True:
        m1.align(g, work["scan"], O.Pose3D())
        f1 += 1
        q1 += m1.stats["queries"]
        el1 = time.perf_counter() - t1
        if el1 > budget_s / 4 or f1 >= 500:
            break
    return {"value": queries / el / 1e6, "unit": "Mcorr/s", "cores": cores, "kind": "port",
            "cores_note": f"{cores} search threads: the GPU box grants a 16-CPU share per GPU, whatever the affinity mask "
                          f"({len(os.sched_getaffinity(0))} CPUs) or hardware_concurrency ({os.cpu_count()}) report; more "
                          "threads than that share only time-slice.  The solve is single-threaded (Ceres num_threads = 1)",
            "frames_per_s": frames / el,
            "single_thread_value": q1 / el1 / 1e6, "single_thread_frames_per_s": f1 / el1,
            "sample": f"{frames} full frames of the same workload ({queries} queries, {el:.1f} s) with the search on "
                      f"{cores} threads and the solve on one, then {f1} frames on one thread ({el1:.1f} s); "
                      "CPU restatement of the reference algorithm (oracle/), not the reference binary"}


def event_period(steps):
    """A step whose launches carry HIP events takes ~40 % longer (an event is a packet of its own on the stream): of a long
    block every 32nd step carries them (200 steps: 7 steps, 35 launches of each kernel), of a short one -- the driver's
    20 steps -- the first step alone.  Two of twenty had cost the short block 3.7 % (0.1323 against 0.1285 ms per step)."""
    return EVENT_PERIOD if steps >= 3 * EVENT_PERIOD else max(int(steps), 1)


def streaming(args, lom, steps=None, warmup=None, cpu_frames=40):
    """BASELINE.json configs[4]: 10 Hz VLP16 sequence through the full per-frame pipeline
    (time-normalise, deskew, classify, range filter, two down-samplers, align, cleanup, keyframe
    insert; defaults of LidarOdometry::Params).  A step = one processCloud; frames are generated
    on the host before the timed region.  Returns the line."""
    from lidar_odometry_demo_amd import synth

    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    boxes = synth.make_boxes()
    n_frames = warmup + steps
    frames = [synth.make_sequence_frame(k, boxes=boxes) for k in range(n_frames)]
    spin = lom.LidarOdometry()          # bring an idle GPU up to steady clocks (untimed, separate state)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        spin.processCloud(frames[0])
        spin.processCloud(frames[1])
    del spin
    odo = lom.LidarOdometry()
    for k in range(warmup):
        odo.processCloud(frames[k])
    q0 = odo.stats["queries_total"]      # (waits for the warm-up's last keyframe update)
    t0 = time.perf_counter()
    # the frames handed over back to back from compiled code (the reference's caller is the C++ node; bench.py's aligns
    # are issued the same way): the keyframe update of frame k runs beside frame k+1's host stages
    odo.processSequence(frames[warmup:n_frames])
    queries = odo.stats["queries_total"] - q0   # waits for the last keyframe update: inside the timed region
    elapsed = time.perf_counter() - t0
    pose = odo.getCurrentPose()
    odo_stats = dict(odo.stats)
    odo_counters = {"redone": odo.debugCounter(), "staged": odo.debugCounter(lom.capi.COUNTER_FRAMES_SENT_AHEAD),
                    "behind": odo.debugCounter(lom.capi.COUNTER_CLEANUPS_BEHIND_ALIGN)}
    del odo   # (its streams go with it: a process has four hardware queues, handles beyond that share them)
    # beside it (never `value`): the same frames to a caller that does NOT hold the next frame while one is processed -- a
    # live sensor --, i.e. without lom_odometry_hint_next: each frame is copied to pinned memory at the start of its own call
    os.environ["LOM_NO_SEND_AHEAD"] = "1"   # (read once, at create)
    try:
        live = lom.LidarOdometry()
    finally:
        del os.environ["LOM_NO_SEND_AHEAD"]
    for k in range(warmup):
        live.processCloud(frames[k])
    _ = live.stats["queries_total"]      # (waits for the warm-up's last keyframe update)
    t_live = time.perf_counter()
    live.processSequence(frames[warmup:n_frames])
    _ = live.stats["queries_total"]      # (... and for the last one: inside the timed region, as above)
    live_ms = (time.perf_counter() - t_live) / steps * 1e3
    same_pose = (live.getCurrentPose().translation.tobytes() == pose.translation.tobytes()
                 and live.getCurrentPose().rotation.tobytes() == pose.rotation.tobytes())
    del live
    gt_t, gt_q = synth.sequence_pose(n_frames * synth.FRAME_PERIOD)
    dq = abs(float(np.dot(pose.rotation.astype(np.float64), gt_q)))
    line = {
        "metric": "icp_correspondences_per_sec", "value": queries / elapsed / 1e6, "unit": "Mcorr/s", "n_gpus": 1,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "frames_per_s": steps / elapsed,
        "config": {"workload": "C5: streaming 10 Hz VLP16 sequence, full per-frame pipeline, default params",
                   "points_per_frame": int(np.mean([len(f) for f in frames])),
                   "keyframe_voxels": odo_stats["keyframe_voxels"],
                   "frames_redone_on_host_or_scans_redone": odo_counters["redone"],
                   "frames_handed_over_by": "lom_odometry_process_sequence: the caller's frame loop in compiled code; it announces "
                                            "frame i + 1 before frame i (lom_odometry_hint_next), which is then copied to pinned "
                                            "memory while frame i's align runs",
                   "frames_staged_during_the_previous_align": odo_counters["staged"],
                   "ms_per_frame_without_announcing_the_next_frame": live_ms,
                   "final_pose_bits_equal_without_announcing": bool(same_pose),
                   "cleanup_scans_behind_the_align": odo_counters["behind"],
                   "matching_points_last": odo_stats["matching_points"],
                   "drift_translation_m": float(np.linalg.norm(pose.translation.astype(np.float64) - gt_t)),
                   "drift_rotation_rad": 2.0 * float(np.arccos(min(1.0, dq))),
                   "drift_note": "x is weakly observable in the street-canyon scene and the reference's "
                                 "translation prior holds it back; yaw, y and z track"},
    }
    if not args.no_cpu_baseline and cpu_frames > 0:
        from oracle import oracle as O

        cores = min(len(os.sched_getaffinity(0)), 16)
        ref = O.LidarOdometry(nthreads=cores)
        m = min(n_frames, cpu_frames)
        t1 = time.perf_counter()
        q = 0
        for k in range(m):
            ref.processCloud(frames[k])
            q += ref.stats["queries"]
        el = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": q / el / 1e6, "unit": "Mcorr/s", "cores": cores, "kind": "port",
                                "frames_per_s": m / el,
                                "sample": f"first {m} frames of the same sequence ({el:.1f} s), CPU restatement"}
    return line


def concurrent_contexts(lom, torch, grid, d_scan, guess, steps, counts=(2, 3, 4, 8)):
    """Side figure (never `value`): `k` host threads, one scan context each (lom_scan_create), all aligning the same
    device-resident scan against the ONE keyframe at the same time -- what `const VoxelGrid&` allows the reference's
    callers (voxel_grid.h:206, cloud_matcher.h:15).  A solve keeps ~53 of the 256 CUs busy on a VLP16-sized scan and every
    launch of an align waits for the one before it, so concurrent callers are how one GPU is filled.  Two forms: contexts
    that share the whole GPU (each search grid fills every SIMD, so the callers' kernels queue behind each other), and
    contexts on k disjoint slices of the compute units (lom_scan_create_on_partition: a CU mask per stream), where the
    callers' chains run side by side.  The C calls release the GIL.
    (slices of 5, 6 or 7 hold a solve of 32 workgroups at once, like 8: the library counts what every (XCD, shader engine)
    pair of a slice is sure to have -- tests/test_gpu_parity.py::test_uneven_slices_hold_their_solve; until that was fixed a
    six-slice context waited out its patience on every align, which is why six was taken off this list.)"""
    import threading

    out = {}
    for k in counts:
        for form in ("shared", "partitioned"):
            if form == "shared" and k > 4:
                continue
            ctxs = [lom.ScanContext(grid, partition=((i, k) if form == "partitioned" else None)) for i in range(k)]
            for c in ctxs:
                lom.align_repeat(c, d_scan.data_ptr(), d_scan.shape[0], guess, 20)
            torch.cuda.synchronize()
            start = threading.Barrier(k + 1)
            res = [None] * k

            def work(i):
                start.wait()
                res[i] = lom.align_repeat(ctxs[i], d_scan.data_ptr(), d_scan.shape[0], guess, steps)[1]

            th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            start.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            out[f"{k}_{form}"] = {"contexts": k, "form": form, "frames_per_s": k * steps / el,
                                  "value": sum(r["queries"] for r in res) / el / 1e6, "unit": "Mcorr/s",
                                  "ms_per_align_per_context": el / steps * 1e3,
                                  "host_fallbacks": sum(r["host_fallback"] for r in res)}
            for c in ctxs:
                c.close()
    return out


def traffic_record(config):
    """HBM-side bytes per k_match launch as last collected with rocprofv3 --pmc (tools/collect_traffic.py; FETCH_SIZE
    doubled + WRITE_SIZE, MI355X_MICROARCH.md): a constant from profiles/, NOT a measurement of this run."""
    for tag in ("r04", "r03", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", f"traffic_{tag}.json" if config == "C2" else f"traffic_{tag}_{config.lower()}.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                rec = json.load(f)
            if rec.get("config", "C2") != config:
                continue
            return rec.get("hbm_bytes_per_launch"), (f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                                       f"separate passes, collected at commit {rec.get('commit', '?')}); "
                                                       "a constant, not measured in this run")
    return None, None


def match_roofline(n_queries, alg_per_launch, req_per_launch, avg_us, traffic, traffic_source, extra=None, big_map=False):
    """`roofline` of the dominant kernel on the SURVEY.md 8(d) definition for EVERY configuration: algorithmic bytes per
    launch / the kernel's average launch duration against the 8 TB/s HBM peak.  Where the map is served from L2 /
    Infinity Cache that fraction can exceed 1 (the formula also counts candidates the exact pruning never reads): the
    cache-level yardstick sits beside it under its own key, never in `frac`."""
    achieved = alg_per_launch / (avg_us * 1e-6) / 1e9
    roof = {
        "kernel": "k_match (27-neighbour correspondence search)",
        # what the counters say bounds the kernel (profiles/): the VLP16-sized launch is one resident round of dependent
        # round trips into an L2 / Infinity-Cache resident map; on the 2M-point map the kernel sits at the rate the caches
        # serve random rows at.  `peak` / `frac` stay priced against the HBM roofline (`priced_against`), as SURVEY.md 8(d)
        # defines them.
        "bound": ("cache bandwidth (L2 / Infinity-Cache served random rows)" if big_map
                  else "latency (cache-resident gather: a chain of dependent round trips per query)"),
        "priced_against": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": alg_per_launch,
        "avg_launch_us": avg_us,
        "requested_bytes_per_launch": req_per_launch,
        "requested_gbs": (req_per_launch / (avg_us * 1e-6) / 1e9 if req_per_launch else None),
        "requested_note": "bytes the kernel itself asks for: neighbour voxels that provably cannot hold a point "
                          "within max_dist are not scanned (exact pruning), + 52 B output/query",
        "hbm_measured_gbs": (traffic / (avg_us * 1e-6) / 1e9 if traffic else None),
        "hbm_measured_frac": (traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS if traffic else None),
        "kernel_only_mcorr_s": n_queries / (avg_us * 1e-6) / 1e6,
        "limited_by": "the length of the dependent chain per query, not HBM: the map (payload + table) is L2 / Infinity-Cache "
                      "resident, a query is source point -> 27 slots -> chunks of candidate rows -> winner's point and normal; "
                      "counters in profiles/ (waves 60-70 % in s_waitcnt; on the 2M-point map ~100 VALU instructions and ~70 "
                      "vector-L1 tag accesses per query, each cut by 11-17 % in round 3 with the time unmoved: DESIGN.md section 5)",
    }
    if req_per_launch:
        roof["cache_level"] = {
            "achieved": roof["requested_gbs"], "peak": CACHE_PEAK_GBS, "frac": roof["requested_gbs"] / CACHE_PEAK_GBS,
            "unit": "GB/s",
            "basis": "REQUESTED bytes / time against the Infinity-Cache-served rate of random rows (8.6 TB/s, "
                     "MI355X_MICROARCH.md 'Indexed rows'); a side figure, `frac` above stays on the 8(d) definition"}
    if extra:
        roof.update(extra)
    return roof


def counted_replay(lom, grid, d_scan, guess):
    """SURVEY.md 8(d)'s algorithmic bytes need cand(q), the stored points of all occupied voxels among a query's 27
    neighbours -- a count of the reference ALGORITHM that the product's searches do not produce (a neighbour the bound
    prunes is not even looked up).  One untimed replay of the same align with LOM_OPT_COUNT_CANDIDATES on gives them
    exactly: the align is a pure function of its inputs, same poses, same iterations."""
    grid.setOption(lom.capi.OPT_COUNT_CANDIDATES, 1)
    try:
        _, tot = lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, 1)
    finally:
        grid.setOption(lom.capi.OPT_COUNT_CANDIDATES, 0)
    return {"algorithmic_bytes_per_launch": tot["algorithmic_bytes"] / max(tot["match_launches"], 1),
            "cand_total": tot["cand_total"], "occ_total": tot["occ_total"], "queries": tot["queries"],
            "outer_iterations": tot["outer_iterations"]}


def align_block(lom, torch, work, dev, steps, warmup_aligns=200):
    """A short block of aligns of another configuration on this GPU (the default run's extra_configs): map build,
    warm-up, `steps` aligns issued back to back from compiled code, k_match / k_lm durations from HIP events on
    some of the aligns (event_period).  Same definitions as the main line."""
    grid = lom.VoxelGrid(0.5, 20, device=dev.index or 0)
    d_map_xyz = torch.from_numpy(work["map_xyz"]).to(dev)
    d_map_nrm = torch.from_numpy(work["map_nrm"]).to(dev)
    torch.cuda.synchronize()
    insert_us = grid.profileInsert(d_map_xyz.data_ptr(), d_map_nrm.data_ptr(), d_map_xyz.shape[0])
    del d_map_xyz, d_map_nrm
    d_scan = torch.from_numpy(work["shard"]).to(dev)
    torch.cuda.synchronize()
    guess = lom.Pose3D()
    lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, warmup_aligns)
    grid.setProfiling(event_period(steps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pose, tot = lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    grid.setProfiling(0)
    train_us, _, train_requested, pair_us = grid.profileMatch(d_scan.data_ptr(), d_scan.shape[0], pose, 0.3, reps=200)
    overhead = max(0.0, pair_us - train_us)
    prof = max(tot["profiled_launches"], 1)
    match_us = max(tot["match_kernel_ms"] * 1e3 / prof - overhead, 1e-3)
    lm_us = max(tot["lm_kernel_ms"] * 1e3 / max(tot["lm_profiled_launches"], 1) - overhead, 1e-3)
    alg = counted_replay(lom, grid, d_scan, guess)["algorithmic_bytes_per_launch"]
    n = int(d_scan.shape[0])
    return {
        "workload": work["name"], "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "value": tot["queries"] / elapsed / 1e6, "unit": "Mcorr/s", "frames_per_s": steps / elapsed,
        "outer_iterations_per_frame": tot["outer_iterations"] / steps, "evaluations_per_frame": tot["evaluations"] / steps,
        "scan_points": n, "map_points_stored": grid.pointCount(), "map_voxels": grid.size(),
        "roofline": match_roofline(n, alg, train_requested, match_us, *traffic_record(work.get("config", "")),
                                   {"train_avg_launch_us": train_us, "event_pair_overhead_us": overhead,
                                    "in_loop_launches_measured": tot["profiled_launches"]}, big_map=True),
        "k_lm_avg_us": lm_us, "k_match_avg_us": match_us, "insert_chain_us": insert_us,
        "pose": {"t": [float(v) for v in pose.translation], "q_wxyz": [float(v) for v in pose.rotation]},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run only: skip the short C3 / C4 blocks and the 200-frame C5 block attached as extra_configs")
    ap.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C2",
                    help="C2 = BASELINE.json configs[1] (the bench line; weak-scaled with --gpus N); C3 = configs[2], "
                         "single GPU only; C4 = configs[3], 128x2048 scan range-sharded over --gpus N ranks (any N "
                         "incl. 1, strong scaling); C5 = configs[4], streaming LidarOdometry::processCloud "
                         "(--steps = frames)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before torch is imported or any HIP call is made in this process)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    claim_stdout()
    if os.environ.get("LOM_BENCH_WATCHDOG_S"):
        # debugging aid: the Python stacks of all threads on stderr after that many seconds, then exit
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["LOM_BENCH_WATCHDOG_S"]), exit=True)

    import torch
    import torch.distributed as dist

    import lidar_odometry_demo_amd as lom

    n = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 and world != n:
        raise SystemExit(f"--gpus {n} needs WORLD_SIZE={n} (launch with torch.distributed.run); got {world}")
    # LOM_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, id broadcast, RCCL
    # communicator, exchange per evaluation) with a single rank: a rehearsal on a one-GPU box
    use_dist = n > 1 or bool(os.environ.get("LOM_BENCH_FORCE_DIST"))
    if use_dist and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # LOM_BENCH_ONE_DEVICE=1: all ranks on GPU 0 with a gloo process group -- a rehearsal of the
    # N > 1 flow on a one-GPU box (the numbers mean nothing: the ranks share the GPU)
    one_device = bool(os.environ.get("LOM_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the align is a chain of host<->device round trips: run on the socket the GPU hangs off
    pinned = lom.capi.pin_to_device_numa_node(local_rank) if not os.environ.get("LOM_NO_PIN") else None
    if use_dist:
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.config not in ("C2", "C4") and n != 1:
        raise SystemExit(f"--config {args.config} is a single-GPU configuration")
    if args.config == "C5":
        emit(streaming(args, lom))
        return
    progress("building the workload")
    work = build_workload(n, rank, args.config)
    progress("workload built")
    grid = lom.VoxelGrid(0.5, 20, device=local_rank)
    # the keyframe map: one bulk insert of device-resident points, bracketed by HIP events on the library's
    # stream (roofline entry of the insert chain; outside the timed region)
    d_map_xyz = torch.from_numpy(work["map_xyz"]).to(dev)
    d_map_nrm = torch.from_numpy(work["map_nrm"]).to(dev)
    torch.cuda.synchronize()
    insert_us = grid.profileInsert(d_map_xyz.data_ptr(), d_map_nrm.data_ptr(), d_map_xyz.shape[0])
    del d_map_xyz, d_map_nrm
    # live HIP-event measurement of k_match inside the timed region: an event pair costs the stream
    # ~5 us per launch (0.218 ms per step with every launch bracketed against 0.170 ms with none), so
    # the launches of every 32nd step carry the events and the others run as a caller would run them (event_period)
    grid.setProfiling(event_period(args.steps))
    d_scan = torch.from_numpy(work["shard"]).to(dev)
    torch.cuda.synchronize()

    # Exchange of the 32 reduced sums per residual evaluation (the path's one real exchange step).
    # "p2p" (default): the device-resident solve on every rank, the GPUs store their totals into each
    # other's HBM (IPC mappings, xGMI); attach runs a self-test on all ranks and the bench falls back
    # to "host" if it fails anywhere.  "host": host-driven solve, the ranks' hosts exchange through
    # shared memory.  "rccl": host-driven solve, all-gather over xGMI.
    exchange = os.environ.get("LOM_EXCHANGE", "p2p")
    host_comm = None
    selftest = {"result": None, "what": "no device-to-device exchange attached"}
    L = lom.capi.lib()

    def broadcast_id(make):
        import ctypes as C

        ident = torch.zeros(lom.capi.COMM_ID_BYTES, dtype=torch.uint8, device="cpu" if one_device else dev)
        if rank == 0:
            buf = C.create_string_buffer(lom.capi.COMM_ID_BYTES)
            lom.capi.check(make(buf))
            ident.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
        dist.broadcast(ident, 0)
        return bytes(ident.cpu().numpy().tobytes())

    def attach(kind):
        nonlocal host_comm, exchange
        import ctypes as C

        if kind == "rccl":
            lom.capi.check(L.lom_comm_init(grid.handle, rank, world, broadcast_id(L.lom_comm_unique_id)), grid.handle)
        else:
            h = C.c_void_p()
            lom.capi.check(L.lom_host_comm_create(rank, world, broadcast_id(L.lom_comm_host_id), C.byref(h)))
            host_comm = h
            if kind == "p2p":
                if L.lom_comm_attach_p2p(grid.handle, h) == 0:
                    selftest.update(result="passed", what="lom_comm_attach_p2p: 200 exchanges of known values through the "
                                                         "IPC mappings, verdict agreed by all ranks")
                    return
                selftest.update(result="failed", what=L.lom_last_error(grid.handle).decode())
                # the verdict is collective: every rank lands here together
                print(f"[bench] rank {rank}: device-to-device exchange not available "
                      f"({L.lom_last_error(grid.handle).decode()}); using the host exchange", file=sys.stderr)
                exchange = "host"
            lom.capi.check(L.lom_comm_attach_host(grid.handle, h), grid.handle)

    def detach():
        nonlocal host_comm
        L.lom_comm_finalize(grid.handle)
        if host_comm is not None:
            L.lom_host_comm_destroy(host_comm)
            host_comm = None

    if use_dist:
        attach(exchange)

    matcher = lom.CloudMatcher()
    guess = lom.Pose3D()

    def step():
        pose = matcher.alignDevice(grid, d_scan.data_ptr(), d_scan.shape[0], guess)
        return pose, matcher.stats

    # clocks and power state: a step is ~0.3 ms, so W of them do not bring an idle GPU up to its
    # steady state; run the same workload (untimed) for about a quarter of a second first
    # (a fixed count, not a time limit: with several ranks every step is a sequence of exchanges
    # and all ranks must take the same number of them)
    progress("map built; warm-up")
    lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, 600)
    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    grid.setProfiling(event_period(args.steps))   # (counts from here: the block's first step carries the events)
    fence()
    t0 = time.perf_counter()
    # exactly K steps, issued back to back from compiled code (the reference's callers are C++)
    pose, tot = lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, args.steps)
    queries, launches = tot["queries"], tot["match_launches"]
    match_ms = tot["match_kernel_ms"]
    profiled = tot["profiled_launches"]
    outer, evals = tot["outer_iterations"], tot["evaluations"]
    launch_ms, wait_ms = tot["host_launch_ms"], tot["host_wait_ms"]
    fence()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lm_ms, lm_profiled = tot["lm_kernel_ms"], tot["lm_profiled_launches"]
    valid_last = tot["valid_last"]
    # the reference-algorithm counts of this align (all ranks take part: an align is a sequence of exchanges), untimed
    progress("timed block done")
    replay = counted_replay(lom, grid, d_scan, guess)
    # ... and what producing them costs the product: one block of K steps with the counts on (never `value`)
    grid.setOption(lom.capi.OPT_COUNT_CANDIDATES, 1)
    fence()
    tc = time.perf_counter()
    lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, args.steps)
    fence()
    counted_ms = (time.perf_counter() - tc) / args.steps * 1e3
    grid.setOption(lom.capi.OPT_COUNT_CANDIDATES, 0)

    # spread: EXTRA_BLOCKS further blocks of the same K steps, each fenced like the official one (never `value`)
    block_ms = [elapsed / args.steps * 1e3]
    for _ in range(EXTRA_BLOCKS):
        fence()
        tb = time.perf_counter()
        _, tot_b = lom.align_repeat(grid, d_scan.data_ptr(), d_scan.shape[0], guess, args.steps)
        fence()
        el = time.perf_counter() - tb
        # the kernels' event samples of these blocks count as well (a short block carries the events of ONE step: five
        # launches of each kernel are a thin average)
        match_ms += tot_b["match_kernel_ms"]
        profiled += tot_b["profiled_launches"]
        lm_ms += tot_b["lm_kernel_ms"]
        lm_profiled += tot_b["lm_profiled_launches"]
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if one_device else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        block_ms.append(el / args.steps * 1e3)

    # roofline probe for the dominant kernel, outside the timed region: a back-to-back train of
    # k_match launches at the converged pose under one HIP event pair on the library's stream
    train_us, train_bytes, train_requested, pair_us = grid.profileMatch(d_scan.data_ptr(), d_scan.shape[0], pose, 0.3,
                                                                        reps=200)
    # PCIe-inclusive variant (scan handed over as a host buffer every step); never `value`
    t1 = time.perf_counter()
    for _ in range(max(3, args.steps // 4)):
        matcher.align(grid, work["shard"], guess)
    pcie_ms = (time.perf_counter() - t1) / max(3, args.steps // 4) * 1e3

    # what a scaling run needs to be believed: who took part (rank ids gathered through the exchange object the
    # transport itself uses), on which devices, and every rank's own step time
    my_ms = my_elapsed / args.steps * 1e3
    props = torch.cuda.get_device_properties(local_rank)
    my_dev = f"{getattr(props, 'name', '?')}|pci {getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', -1):02x}:" \
             f"{getattr(props, 'pci_device_id', -1):02x}|uuid {getattr(props, 'uuid', '?')}"
    if use_dist:
        objs = [None] * world
        dist.all_gather_object(objs, (rank, my_dev, my_ms))
        rank_ms = [o[2] for o in objs]
        devices_seen = sorted({o[1] for o in objs})
        if host_comm is not None:
            import ctypes as C

            mine = (C.c_int32 * 1)(rank)
            got = (C.c_int32 * world)()
            rc_g = L.lom_host_comm_allgather(host_comm, mine, 4, got)
            ranks_seen = {"through": f"lom_host_comm_allgather (the exchange object of the '{exchange}' transport)",
                          "ranks": [int(v) for v in got] if rc_g == 0 else None, "rc": int(rc_g)}
        else:
            ranks_seen = {"through": "torch.distributed all_gather_object (RCCL transport: the library's communicator "
                                     "carries sums only)", "ranks": [o[0] for o in objs], "rc": 0}
    else:
        rank_ms, devices_seen = [my_ms], [my_dev]
        ranks_seen = {"through": "single rank", "ranks": [0], "rc": 0}

    if rank == 0:
        # stats are global (summed over ranks) after the in-library all-gather
        value = queries / elapsed / 1e6
        # -- the dominant kernel, k_match: duration measured live inside the timed region (HIP events on the
        # library's stream around the launches of every 32nd step, of a short block's first step).  An event pair around ONE short kernel carries
        # packet overhead; it is measured right here (same launches at the final pose once as a back-to-back
        # train under one pair, once with a pair each) and subtracted: the result is what
        # `rocprofv3 --kernel-trace --stats` reports as the kernel's average duration (profiles/).
        event_overhead_us = max(0.0, pair_us - train_us)
        in_loop_raw_us = match_ms * 1e3 / max(profiled, 1)
        in_loop_us = max(in_loop_raw_us - event_overhead_us, 1e-3) if profiled else train_us
        alg_per_launch = replay["algorithmic_bytes_per_launch"] / n      # counters are totals over ranks after the exchange
        req_per_launch = train_requested if n == 1 else None   # counted at the final pose (the train)
        achieved = alg_per_launch / (in_loop_us * 1e-6) / 1e9
        traffic, traffic_source = traffic_record(args.config) if n == 1 else (None, None)
        roof = match_roofline(int(d_scan.shape[0]), alg_per_launch, req_per_launch, in_loop_us, traffic, traffic_source, {
            "avg_launch_us_method": "HIP event pairs around the k_match launches of every 32nd step (short blocks: the first step) inside the timed "
                                    "region, minus the per-pair event overhead measured in this run",
            "in_loop_raw_us": in_loop_raw_us,
            "event_pair_overhead_us": event_overhead_us,
            "in_loop_launches_measured": profiled,
            "in_loop_launches_from": f"the timed block and the {EXTRA_BLOCKS} blocks of the same K steps behind it (ms_per_step_blocks)",
            "train_avg_launch_us": train_us,
            "train_note": "200 back-to-back launches at the final pose under one event pair (best case: warm caches, "
                          "converged pose)",
            "launches": launches,
            "algorithmic_bytes_from": "a counted replay of the same align outside the timed region (LOM_OPT_COUNT_CANDIDATES: "
                                      "every query looks up all 27 slots and tallies the reference algorithm's cand(q)); the "
                                      "timed searches do not look up a neighbour voxel that the distance bound prunes",
            "ms_per_step_with_counts": counted_ms,
        }, big_map=(args.config != "C2"))
        # -- the other kernels of the path
        lm_raw_us = lm_ms * 1e3 / max(lm_profiled, 1)
        lm_us = max(lm_raw_us - event_overhead_us, 1e-3) if lm_profiled else None
        evals_per_launch = evals / max(outer, 1)
        lm_bytes = 36.0 * (valid_last / n) * evals_per_launch   # 36 B per valid match and evaluation
        stored = grid.pointCount()
        n_map = int(len(work["map_xyz"]))
        ins_bytes = 40.0 * n_map + 24.0 * stored
        kernels = [
            {"kernel": "k_match", "algorithmic_bytes_per_launch": alg_per_launch, "avg_us": in_loop_us,
             "gbs": achieved, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS, "launches_per_step": launches / args.steps},
            {"kernel": "k_lm (evaluation + reduction + exchange + LM policy, one launch per outer iteration)",
             "algorithmic_bytes_per_launch": lm_bytes, "avg_us": lm_us,
             "gbs": (lm_bytes / (lm_us * 1e-6) / 1e9 if lm_us else None),
             "frac_of_hbm_peak": (lm_bytes / (lm_us * 1e-6) / 1e9 / HBM_PEAK_GBS if lm_us else None),
             "evaluations_per_launch": evals_per_launch, "launches_per_step": outer / args.steps,
             "workgroups": tot["lm_workgroups"], "cus_busy": tot["lm_workgroups"], "cus_total": 256,
             "cus_note": "one workgroup per CU (its workgroups wait for each other: the grid is held to what is resident); "
                         "the other CUs are free for concurrent callers (concurrent_contexts)",
             "bytes_note": "36 B per valid match and evaluation (source point, plane origin, plane normal)",
             "note": "a latency chain, not a stream: per evaluation one pass over <= 1 point per lane, a workgroup "
                     "reduction, one exchange between the workgroups through HBM and a serial 6x6 policy step on one "
                     "wave; phase stamps and SQ counters in profiles/"},
            {"kernel": "insert chain of the bulk map build (k_bi_claim, k_bi_colscan, k_bi_scatter, k_bi_group, "
                       "k_bi_flagscan, k_bi_place)",
             "algorithmic_bytes_per_launch": ins_bytes, "avg_us": insert_us,
             "gbs": ins_bytes / (insert_us * 1e-6) / 1e9, "frac_of_hbm_peak": ins_bytes / (insert_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
             "bytes_note": "per input point 24 B read (point, normal) + one 16-B slot; per stored point 24 B written",
             "points": n_map, "stored": stored,
             "note": "one call, all kernels under one HIP event pair; per-kernel split in profiles/ (kernel stats); "
                     "the partitioned bulk insert (one slot look per point, a compare-and-swap per new voxel, everything per "
                     "voxel in LDS, slab rows written side by side): bound by the scattered passes of the partition (16-byte "
                     "records) and of the place kernel's gather of the input rows, not by bytes; the four-kernel path of "
                     "batches up to 65,536 points takes 2.5-3x as long (tools/ab_insert.py, profiles/r04_i_ab_insert.txt)"},
        ]
        bsorted = sorted(block_ms)
        line = {
            "metric": "icp_correspondences_per_sec",
            "value": value,
            "unit": "Mcorr/s",
            "n_gpus": n,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (args.config == "C4" and n > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "f32 distances on f64-transformed queries (search); f64 residuals, Jacobians and solve",
            "data": "synthetic",
            "frames_per_s": args.steps / elapsed,
            "ms_per_step_blocks": {"blocks": block_ms, "median": bsorted[len(bsorted) // 2], "min": bsorted[0],
                                   "max": bsorted[-1],
                                   "note": f"the official block first, then {EXTRA_BLOCKS} more blocks of {args.steps} steps"},
            "config": {
                "workload": work["name"],
                "ranks_seen": ranks_seen,
                "devices": devices_seen,
                "exchange_selftest": selftest,
                "ms_per_step_ranks": {"min": min(rank_ms), "max": max(rank_ms)},
                "scan_points_per_gpu": int(d_scan.shape[0]),
                "scan_points_total": int(len(work["scan"])),
                "map_points_stored": stored,
                "map_voxels": grid.size(),
                "pcie_inclusive_ms_per_step": pcie_ms,
                "valid_match_rate": valid_last / max(1, int(len(work["scan"]))),
                "outer_iterations_per_frame": outer / args.steps,
                "evaluations_per_frame": evals / args.steps,
                "parallelism": f"source-range x{n}, map replicated" if n > 1 else "single GPU",
                "host_cpus": (f"{len(pinned)} CPUs of the GPU's NUMA node" if pinned else "not pinned"),
                "exchange": (exchange if use_dist else None),
                "exchange_note": ("32 f64 per rank per residual evaluation; p2p = device-resident solve, the GPUs "
                                  "store their totals into each other's HBM; host = host-driven solve, "
                                  "shared-memory exchange between the ranks' hosts; rccl = host-driven solve, "
                                  "all-gather over xGMI" if use_dist else None),
                "other_exchange": None,
            },
            "host_breakdown_ms_per_step": {"in_launch_calls": launch_ms / args.steps,
                                           "waiting_for_results": wait_ms / args.steps},
            "roofline": roof,
            "roofline_kernels": kernels,
        }
        progress("line assembled")
        if n == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(work)
            progress("cpu baseline done")
            line["speedup_vs_cpu_port"] = value / line["cpu_baseline"]["value"]
        line["pose"] = {"t": [float(v) for v in pose.translation], "q_wxyz": [float(v) for v in pose.rotation]}
        # (C2 only: on a slice of the GPU the 128 x 512-thread solve of the 64- / 128-beam scans is not co-resident, every
        # launch sits out its patience and the align is redone by the host-driven loop -- minutes per block)
        if n == 1 and args.config == "C2" and not use_dist and not args.no_extras:
            try:
                cc = concurrent_contexts(lom, torch, grid, d_scan, guess, min(args.steps, 100), counts=(2, 3, 4, 8))
                line["concurrent_contexts"] = dict(cc, note="k threads, one scan context each, the same scan against the one "
                                                            "keyframe at the same time (shared: every context on the whole GPU; "
                                                            "partitioned: context i on slice i of k of the compute units); a side "
                                                            "figure, never `value`")
                line["frames_per_s_4ctx"] = cc["4_partitioned"]["frames_per_s"]
                progress("concurrent contexts done")
            except Exception as e:  # noqa: BLE001
                line["concurrent_contexts"] = {"error": repr(e)[:300]}
        if n == 1 and args.config == "C2" and not args.no_extras and not use_dist:
            # the other single-GPU configurations of BASELINE.json, short blocks on the same box (never `value`)
            extras = {}
            t_x = time.perf_counter()
            try:
                w3 = build_workload(1, 0, "C3")
                extras["C3"] = align_block(lom, torch, w3, dev, steps=max(20, args.steps // 4))
                progress("extra C3 done")
            except Exception as e:  # noqa: BLE001  (an extra must not cost the line)
                w3 = None
                extras["C3"] = {"error": repr(e)[:300]}
            try:
                # C4 on this one GPU: the same 2M-point map, the 128-beam scan (the N > 1 run shards this scan)
                from lidar_odometry_demo_amd import synth
                scan4, _, _, _ = synth.make_scan(128, 2048, boxes=synth.make_boxes())
                w4 = (dict(w3, scan=scan4, shard=scan4, config="C4",
                           name=f"C4 (BASELINE configs[3]) on one GPU: 128x2048 scan ({len(scan4)} returns) vs 2M-pt map, "
                                f"voxel 0.5 m, cap 20") if w3 is not None else build_workload(1, 0, "C4"))
                extras["C4"] = align_block(lom, torch, w4, dev, steps=max(20, args.steps // 4), warmup_aligns=100)
                progress("extra C4 done")
                del w4
            except Exception as e:  # noqa: BLE001
                extras["C4"] = {"error": repr(e)[:300]}
            w3 = None
            try:
                c5 = streaming(args, lom, steps=200, warmup=10, cpu_frames=0)
                extras["C5"] = {"workload": c5["config"]["workload"], "frames": 200, "ms_per_frame": c5["ms_per_step"],
                                "frames_per_s": c5["frames_per_s"], "value": c5["value"], "unit": "Mcorr/s",
                                "drift_translation_m": c5["config"]["drift_translation_m"],
                                "drift_rotation_rad": c5["config"]["drift_rotation_rad"],
                                "points_per_frame": c5["config"]["points_per_frame"],
                                "keyframe_voxels": c5["config"]["keyframe_voxels"]}
            except Exception as e:  # noqa: BLE001
                extras["C5"] = {"error": repr(e)[:300]}
            extras["wall_s"] = time.perf_counter() - t_x
            line["extra_configs"] = extras
    else:
        line = None

    # The other transports, timed in the same run for comparison (never `value`): the north_star names an RCCL
    # all-reduce per iteration, the default is the device-to-device exchange -- both get a number.  The line is complete
    # before this block starts and a watchdog guards it: a transport that fails or hangs on some rank costs the side
    # figure, not the line (LOM_BENCH_NO_COMPARE=1 skips the block).
    if use_dist and not os.environ.get("LOM_BENCH_NO_COMPARE"):
        others = []

        def fire():
            if rank == 0:
                others.append({"error": "watchdog: the comparison block did not finish; the ranks were ended"})
                line["config"]["other_exchange"] = others
                emit(line)
            print(f"[bench] rank {rank}: comparison block timed out, leaving", file=sys.stderr)

        alts = [a for a in ("rccl", "host") if a != exchange]
        for alt in alts:
            if alt == "rccl" and one_device:
                others.append({"exchange": "rccl", "skipped": "LOM_BENCH_ONE_DEVICE rehearsal: RCCL refuses several ranks "
                                                              "on one device"})
                continue
            dog = Watchdog(float(os.environ.get("LOM_BENCH_COMPARE_TIMEOUT_S", "120")), fire)
            try:
                detach()
                attach(alt)
                for _ in range(min(3, args.warmup)):
                    step()
                fence()
                t_alt = time.perf_counter()
                q_alt = 0
                for _ in range(args.steps):
                    q_alt += step()[1]["queries"]
                fence()
                el = time.perf_counter() - t_alt
                t = torch.tensor([el], dtype=torch.float64, device="cpu" if one_device else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                others.append({"exchange": alt, "value": q_alt / float(t.item()) / 1e6, "unit": "Mcorr/s",
                               "ms_per_step": float(t.item()) / args.steps * 1e3,
                               "what": ("host-driven solve; per residual evaluation k_eval + k_sum_records, one RCCL all-gather "
                                        "of 32 f64 per rank, D2H copy, rank-ordered host sum" if alt == "rccl" else
                                        "host-driven solve; the ranks' hosts exchange the 32 sums through shared memory")})
            except Exception as e:  # noqa: BLE001
                others.append({"exchange": alt, "error": repr(e)[:300]})
            dog.cancel()
        if rank == 0:
            line["config"]["other_exchange"] = others

    if rank == 0:
        emit(line)

    if use_dist:
        # the line is out: a teardown that hangs (a transport left half attached above) must not turn into a failed run
        dog = Watchdog(60.0, lambda: print(f"[bench] rank {rank}: teardown timed out, leaving", file=sys.stderr))
        detach()
        dist.barrier()
        dist.destroy_process_group()
        dog.cancel()


if __name__ == "__main__":
    main()
