"""Stand-in for a rank of bench.py in the CPU test of gficf_amd.launch.spawn_ranks: joins a gloo group from the rank
environment, sums the ranks with one all-reduce, rank 0 prints ONE JSON line.  `--fail-rank R`: rank R exits with
code 3 before the collective (the others would wait for it forever); `--hang`: every rank sleeps."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--fail-rank", type=int, default=-1)
ap.add_argument("--hang", action="store_true")
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == a.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
if rank == a.fail_rank:
    sys.exit(3)
if a.hang:
    time.sleep(600)
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t)
if rank != 0:
    print(f"rank {rank} says hello on stdout")          # must not reach the launcher's stdout
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": int(t.item())}))
dist.destroy_process_group()
