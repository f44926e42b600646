# tangent width of the dual-number Jacobians (GRAPHITE_ENGINE_TANGENT_WIDTH): k3 and pinhole factors on the Ladybug-1723 shape (GPU box)
cd $GRAFT_REPO_ROOT
python - <<'PY'
from graphite_amd import synth
synth.write_bal("/tmp/l1723.txt", synth.make_config("ladybug-1723"))
PY
# the two other widths are builds of the same client (nothing else in the tree builds them)
set -e
mkdir -p build
for tw in 6 12; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -Iinclude -DGRAPHITE_ENGINE_TANGENT_WIDTH=$tw tests/cpp/test_engine_model.hip -Lgraphite_amd -lgraphite_mi355x -Wl,-rpath,$PWD/graphite_amd -o build/test_engine_model_tw$tw
done
for b in test_engine_model test_engine_model_tw6 test_engine_model_tw12; do
  [ -x ./build/$b ] || { echo "missing ./build/$b (python __graft_entry__.py builds test_engine_model)" >&2; exit 1; }
  for mode in k3 pinhole; do
    echo "== $b $mode"; GRAPHITE_ENGINE=model GR_PROFILE_KERNELS=1 timeout 300 ./build/$b /tmp/l1723.txt pcg 20 $mode stored fp64 2>&1 | grep -E "KERNEL linearize |LM_LOOP" | cut -c1-120
  done
done
