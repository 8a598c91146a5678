"""`python bench.py --gpus N` as the driver types it: without a launcher around it the script starts its own N rank
processes (bench.self_launch), relays rank 0's single JSON line and exits with the ranks' verdict."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                   # exactly the bare command: no launcher's environment
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def test_self_launch_reports_failing_ranks_and_leaves_nothing_behind():
    """No GPU in this container: every rank fails when it selects its device.  The launcher must come back non-zero,
    print no line, and not hang on the ranks that are still alive."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU flavour of this test is below)")
    rc, lines, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                          {"LOM_BENCH_LAUNCH_TIMEOUT_S": "200"}, timeout=400)
    assert rc != 0
    assert lines == []
    assert "failed; ending the others" in err or "still running" in err


def test_event_sampling_of_short_and_long_blocks():
    """bench.event_period: a step that carries HIP events takes ~40 % longer, so a short block (the driver's 20 steps) carries
    them on its first step only and a long one on every 32nd."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.event_period(20) == 20 and bench.event_period(5) == 5 and bench.event_period(1) == 1
    assert bench.event_period(95) == 95 and bench.event_period(96) == 32 and bench.event_period(200) == 32
    assert bench.event_period(0) == 1


@pytest.mark.gpu
def test_plain_command_with_two_ranks_prints_one_line():
    """`python bench.py --gpus 2 --steps 5`, nothing else: two fresh rank processes (here both on the one GPU of the
    box, LOM_BENCH_ONE_DEVICE=1: a rehearsal of the flow, the number means nothing), one JSON line, both ranks seen
    through the exchange object the transport itself uses, the other transport timed beside it."""
    rc, lines, err = _run(["--gpus", "2", "--steps", "5", "--warmup", "2"], {"LOM_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["metric"] == "icp_correspondences_per_sec"
    cfg = line["config"]
    assert cfg["ranks_seen"]["ranks"] == [0, 1], cfg["ranks_seen"]
    assert cfg["exchange"] in ("p2p", "host")
    assert cfg["exchange_selftest"]["result"] in ("passed", "failed")
    assert line["value"] > 0 and line["ms_per_step"] > 0
    others = cfg["other_exchange"]
    assert isinstance(others, list) and others, others
    assert any(o.get("exchange") == "host" and o.get("value", 0) > 0 for o in others) or cfg["exchange"] == "host", others
    assert any(o.get("exchange") == "rccl" for o in others)                 # skipped here, with the reason given


@pytest.mark.gpu
def test_one_rank_line_is_the_same_with_and_without_the_launcher_path():
    """--gpus 1 never self-launches; its value must not depend on the N > 1 plumbing being present: two short runs
    agree within 10 % (the second with LOM_BENCH_FORCE_DIST=1, the multi-rank code path with one rank)."""
    a = _run(["--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extras"], {}, timeout=600)
    b = _run(["--gpus", "1", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-extras"],
             {"LOM_BENCH_FORCE_DIST": "1", "LOM_BENCH_NO_COMPARE": "1"}, timeout=600)
    assert a[0] == 0 and b[0] == 0, (a[2][-2000:], b[2][-2000:])
    la, lb = json.loads(a[1][-1]), json.loads(b[1][-1])
    assert la["n_gpus"] == lb["n_gpus"] == 1
    assert abs(la["value"] - lb["value"]) / la["value"] < 0.10, (la["value"], lb["value"])
