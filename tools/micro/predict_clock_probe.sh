#!/bin/bash
# shader clock and package power while the headline loop (bench.py, BASELINE cfg 4) runs: rocm-smi once a second
python bench.py --steps 2500 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r4_clock_bench.json 2> gpurun_out/r4_clock_bench.err &
pid=$!
for i in $(seq 1 60); do
  if ! kill -0 $pid 2>/dev/null; then break; fi
  echo "t=$i $(rocm-smi --showclocks --showpower 2>&1 | grep -i 'sclk\|Package Power' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
done
wait $pid
cut -c1-200 gpurun_out/r4_clock_bench.json
