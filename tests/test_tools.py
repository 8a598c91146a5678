"""The measurement helpers under tools/ parse what LOM_DEBUG_TIMING and rocprofv3 write: small synthetic inputs keep them
from rotting unnoticed (no GPU)."""
import os
import subprocess
import sys

from tests.conftest import ROOT


def _run(script, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_stage_times_averages_the_last_frames(tmp_path):
    log = tmp_path / "log.txt"
    with open(log, "w") as f:
        for k in range(10):
            f.write(f"  front end enq.     {10.0 + k:8.1f} us\n  align              {100.0:8.1f} us\n")
            if k % 2:
                f.write(f"  upd status         {20.0:8.1f} us\n")   # (a lap of another thread: not in every block)
            f.write(f"processCloud total {110.0 + k:8.1f} us\n")
    out = _run("stage_times.py", str(log), "-4")
    rows = {ln[:18].strip(): float(ln[18:].split()[0]) for ln in out.splitlines() if ln.strip()}
    assert rows["front end enq."] == 17.5 and rows["align"] == 100.0 and rows["total"] == 117.5   # frames 6..9
    out = _run("stage_times.py", str(log), "8")
    rows = {ln[:18].strip(): float(ln[18:].split()[0]) for ln in out.splitlines() if ln.strip()}
    assert rows["front end enq."] == 18.5                                                           # frames 8, 9


def _trace(path, frames=12):
    names = ["lom::k_fe_stats(lom_point_xyzirt const*, unsigned int)", "lom::k_fe_deskew(int)",
             "void lom::k_match<16, 4, 7, false, false, false, false>(lom::MapView)", "void lom::k_lm<256, 64, 1, false>(int)",
             "void lom::k_cleanup_scan<1>(float const*)", "lom::k_gather_words(lom::WordPtrs)"]
    durs = [18000, 5000, 10000, 16000, 5000, 3000]
    with open(path, "w") as f:
        f.write('"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        t = 1_000_000
        for _ in range(frames):
            for n, (name, d) in enumerate(zip(names, durs)):
                q = 3 if n < 2 else 1
                f.write(f'"KERNEL_DISPATCH",1,{q},"{name}",{t},{t + d}\n')
                t += d + 1500
            t += 20000


def test_trace_tools_read_a_kernel_trace(tmp_path):
    d = tmp_path / "raw" / "host"
    d.mkdir(parents=True)
    _trace(d / "1_kernel_trace.csv")
    out = _run("trace_window.py", str(tmp_path / "raw"), "0.5", "12")
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 12 and "k_fe_stats" in out and "k_match<16, 4, 7, false, false, false, false>" in out
    assert lines[0].split()[0] == "0.0"
    out = _run("c5_timeline.py", str(tmp_path / "raw"))
    assert "most common kernel sequence" in out and "6 launches" in out
    assert "k_fe_stats launches that start while an align's kernels run: 0 of 12" in out
    busy = [ln for ln in out.splitlines() if ln.startswith("device busy per frame")][0]
    assert abs(float(busy.split("median")[1].split()[0]) - 57.0) < 0.01   # 18 + 5 + 10 + 16 + 5 + 3 us of kernels
