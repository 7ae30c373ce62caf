#!/bin/bash
# shader clock and package power while ONE variant of tools/micro/kmat_lab runs for a few seconds (rocm-smi, two samples each)
L=tools/micro/_bin/kmat_lab
for name in "V1 V0 without" "S0 stores" "V0 tile"; do
  echo "=== $name"
  LAB_ONLY="$name" timeout -k 5 60 $L 4096 2 10 16000 &
  pid=$!
  sleep 2.0; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power" | head -6
  sleep 0.7; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -4
  wait $pid
done
echo "=== idle"; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|power" | head -6
