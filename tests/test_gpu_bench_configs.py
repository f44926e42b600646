"""bench.py end to end on the BASELINE.json configurations that are NOT its default line: the bench's own
parity gate (exit status 3 when the timed run's chi2 trace leaves the oracle's by more than the bar) is exercised on
Venice-1778 fp32 (configs[3]) and Final-13682 fp64 (configs[4]) too, not only on Ladybug-1723."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,steps,bar", [("venice-1778", 3, 1e-4), ("final-13682", 2, 1e-6), ("ladybug-49", 6, 1e-4)])
def test_bench_line_and_parity_gate(workload, steps, bar):
    # roofline.traffic: measured live by two rocprofv3 --pmc child passes on the small configuration, looked up on the two large
    # ones (their child passes would rebuild 5 M / 29 M-observation problems twice)
    live = workload == "ladybug-49"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "1",
           "--repeats", "1", "--no-also", "--parity-only", "--pmc-traffic", "auto" if live else "off"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.returncode, out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert workload in line["config"]["workload"] and line["steps_run"] == steps and line["value"] > 0
    assert line["parity_steps"] >= 2 and line["parity_rel"] is not None and line["parity_rel"] < bar
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["kind"] == "port"
    if live and line["roofline"]["bound"] == "hbm":
        assert line["roofline"]["traffic"] > 0 and line["roofline"]["traffic_source"].startswith("measured by this run")
