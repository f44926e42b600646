"""One bench.py line + its per-kernel table, as text: python tools/bench_kernels.py <tag> -- <bench.py arguments>
(writes gpurun_out/<tag>.json / <tag>_kernels.json)"""
import json
import os
import subprocess
import sys

tag = sys.argv[1]
args = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--" else sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
kj = os.path.join(root, "gpurun_out", tag + "_kernels.json")
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-also", "--no-cpu-baseline", "--pmc-traffic", "off", "--dump-kernels", kj, *args],
                     capture_output=True, text=True)
lines = [x for x in out.stdout.splitlines() if x.startswith("{")]
if not lines:
    print(out.stderr[-2000:])
    sys.exit(1)
open(os.path.join(root, "gpurun_out", tag + ".json"), "w").write(lines[-1] + "\n")
d = json.loads(lines[-1])
print(f"== {tag}: {d['value']:.1f} {d['unit']}  ({d['ms_per_step'] * 1e3:.1f} us per step, value range {d.get('value_min')} .. {d.get('value_max')}), parity_rel {d.get('parity_rel')}")
k = json.load(open(kj))["kernels"]
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["total_ms"]):
    a = max(1, v.get("active_launches", v["launches"]))
    print(f"   {n:22s} n={v['launches']:4d} active={a:4d} {v['total_ms'] * 1e3 / a:8.1f} us/launch  {v['total_ms'] * 1e3 / max(d['steps'], 1):8.1f} us/step  {v['bytes_per_launch'] / 1e6:8.1f} MB")
