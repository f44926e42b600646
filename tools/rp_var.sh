#!/bin/bash
# phase stamps of the resident launch under its timing variants (GR_RP_VAR bits: see RpParams::var); two fixed inner iterations
for v in ${RP_VARS:-128 129 132 144 160 192 255}; do
  echo "== var $v"
  GR_RP_VAR=$v timeout 120 python tools/rp_phases.py ${RP_CFG:-ladybug-1723} ${RP_DT:-f64} 2 2>&1 | grep "rp-debug" | tail -11 | awk '{printf "%s/%s ", $4, $5} END{print ""}'
done
