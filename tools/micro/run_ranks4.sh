#!/bin/bash
# run the 4-rank gloo rehearsal of bench.py up to 3 times; stop at the first run that does not finish in time
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  port=$((29600 + i))
  GPB_DIST_BACKEND=gloo GPB_BENCH_WATCHDOG=90 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 5 110 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 4 --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r4u_run$i.out 2> gpurun_out/r4u_run$i.err
  rc=$?
  echo "run $i rc=$rc $(date +%s)" | tee -a gpurun_out/r4u_summary.txt
  if [ $rc -ne 0 ]; then break; fi
done
