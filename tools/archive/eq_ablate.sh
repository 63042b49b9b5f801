#!/bin/bash
# Timing ablations of the EQ cascade kernel: builds variant libraries on the GPU box and probes each.
cd "$(dirname "$0")/.."
for abl in 0 1 2 4 8 15; do
  [ $abl = 0 ] && cp airwave_amd/libairwave_hip.so airwave_amd/libairwave_hip_eqabl0.so
  echo -n "ABL=$abl: "; AIRWAVE_HIP_LIBRARY=$PWD/airwave_amd/libairwave_hip_eqabl$abl.so python tools/eq_probe.py 512 960000 2>/dev/null | tail -1
done
