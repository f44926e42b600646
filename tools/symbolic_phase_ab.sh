#!/bin/bash
# ON THE GPU BOX: wall time of the generic layer's Schur path (block-sparse H + S structure built, two LM iterations) on a
# Ladybug-1723-shape graph, device-side symbolic phase (build/test_generic_bal) next to the round-2 host hash-map phase
# (build/test_generic_bal_hostsym, built from the previous commit when present).  GRAPHITE_GENERIC_ONLY=1 keeps the graph
# off the gr_bal engine.
set -u
W=${1:-ladybug-1723}
export TMPDIR=/tmp GRAPHITE_GENERIC_ONLY=1
F=/tmp/symab_problem.txt
python3 - "$W" "$F" <<'PY'
import sys
from graphite_amd import synth
synth.write_bal(sys.argv[2], synth.make_config(sys.argv[1]))
PY
for B in build/test_generic_bal_hostsym build/test_generic_bal; do
  [ -x $B ] || continue
  for IT in 0 2; do
    S=$(date +%s%N); $B $F pcg-schur $IT stored > /tmp/symab.out 2>&1; E=$(date +%s%N)
    echo "$B iterations=$IT wall $(( (E - S) / 1000000 )) ms  $(grep FINAL_CHI2 /tmp/symab.out)"
  done
done
