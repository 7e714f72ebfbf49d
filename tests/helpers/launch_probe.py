"""A rank of the launcher test: gloo process group, one all-reduce, rank 0 prints a JSON line the
way bench.py does (n_gpus = what the process group saw)."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
total = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64)
dist.all_reduce(total)
if dist.get_rank() == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(total.item()),
                      "argv": sys.argv[1:], "master": os.environ["MASTER_ADDR"]}))
dist.barrier()
dist.destroy_process_group()
if "--fail" in sys.argv and int(os.environ["RANK"]) == 1:
    sys.exit(3)
