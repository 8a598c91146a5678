"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the product's host align
driver over range-sharded source points (tests/_dist_worker.py)."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_align_gloo(tmp_path, world):
    out = tmp_path / "dist.json"
    env = dict(os.environ)
    env["LOM_DIST_OUT"] = str(out)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["OMP_NUM_THREADS"] = "1"
    port = 29510 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(out.read_text())
    assert res["world"] == world
    assert res["dt"] < 1e-4 and res["dr"] < 1e-4, res          # BASELINE.json pose bar
    assert res["outer"] == res["outer_ref"]
    assert res["queries"] == res["queries_ref"]                # shards cover every source point once
    assert res["cand"] == res["cand_ref"]
    assert res["allreduces"] == res["evaluations"]             # exactly one exchange per residual evaluation
