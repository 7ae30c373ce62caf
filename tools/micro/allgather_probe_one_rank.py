#!/usr/bin/env python3
"""WalkerSharding.time_allgather on the one-rank nccl group a one-GPU box can form: launch and protocol cost of the in-stream
ncclAllGather of the C ABI against torch.distributed's, without a wire (bench.py's extras.allgather_probe at N > 1)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
from gpbayestools_hic_amd.dist import WalkerSharding
from gpbayestools_hic_amd.workload import build_chain
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
chain, emu, info = build_chain(1)
sh = WalkerSharding()
out = {"torch_distributed_us": [round(sh.time_allgather(256), 2) for _ in range(3)]}
why = sh.try_direct(emu._engine_ready())
out["direct_path"] = why is None
out["c_abi_in_stream_us"] = [round(sh.time_allgather(256), 2) for _ in range(3)]
print(json.dumps(out))
dist.destroy_process_group()
