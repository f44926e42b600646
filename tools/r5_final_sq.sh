cd $GRAFT_REPO_ROOT
python bench.py --workload final-13682 --no-cpu-baseline --no-also --steps 3 --warmup 1 --repeats 3 > gpurun_out/bench_final13682.json 2> gpurun_out/bench_final13682.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench_final13682.json").readline())
r = l["roofline"]
print("value", l["value"], l["value_min"], l["value_max"])
print({k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "traffic", "traffic_over_algorithmic", "valu_busy", "mem_wait", "waves_per_simd", "limiter")})
print("calibrated", r.get("traffic_calibrated")); print(r.get("sq", {}).get("raw")); print(r["kernels"])
PY
python bench.py --workload venice-1778 --no-cpu-baseline --no-also --repeats 3 > gpurun_out/bench_venice1778.json 2> gpurun_out/bench_venice1778.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench_venice1778.json").readline())
r = l["roofline"]
print("value", l["value"], l["value_min"], l["value_max"])
print({k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "traffic", "traffic_over_algorithmic", "valu_busy", "mem_wait", "waves_per_simd", "limiter")})
print("calibrated", r.get("traffic_calibrated")); print(r["kernels"])
PY
