cd $GRAFT_REPO_ROOT
(time python bench.py) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench_default.json").readline())
r = l["roofline"]
print("value", l["value"], l["value_min"], l["value_max"], "parity", l["parity_rel"])
print({k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "traffic", "traffic_over_algorithmic", "valu_busy", "mem_wait", "waves_per_simd", "limiter")})
print("calibrated", r.get("traffic_calibrated")); print("residency", r.get("cache_residency", {}).get("served_from"), r.get("cache_residency", {}).get("working_set_bytes"))
print("ref-equivalent", r.get("reference_equivalent_pcg_iteration"))
for e in l["also"]:
    print(e["workload"][:70], "|", e["value"], e.get("value_min"), e.get("value_max"), "parity", e.get("parity_rel"), "roof", (e.get("roofline") or {}).get("kernel"), (e.get("roofline") or {}).get("frac"))
PY
timeout 1200 python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_parity.py tests/test_gpu_bench_configs.py -x -q -m gpu 2>&1 | tail -5
