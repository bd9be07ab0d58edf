"""One rank of the world_size-2 CPU (gloo) test of the multi-GPU protocol.

Exactly what bench.py / a multi-GPU query does per rank, with the oracle standing in for
the device kernels (this is a test of the sharding + reduction protocol, not of kernels):
  shard = rv_shard_range(N, world, rank)          row ranges, 64-row aligned
  rows  = generator(first_row = shard.begin)      global row index => no data exchange
  filter/project locally -> gather in rank order == the unsharded result
  {SUM, COUNT} partials -> all_reduce(SUM) of 2 x int64 == the unsharded aggregate
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from rivulus_amd import capi  # noqa: E402
from rivulus_amd.capi import RV_INT64, Predicate, Term, synth_spec  # noqa: E402


def main():
    n_global = int(sys.argv[1])
    out_path = sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    begin, end = capi.shard_range(n_global, world, rank)
    x = pyoracle.generate(synth_spec(RV_INT64, seed=42, length=end - begin, first_row=begin))
    pred = Predicate([Term(0, ">", 899)])
    local = pyoracle.filter_project([x], pred, [0])[0]
    s, _, c = pyoracle.filter_agg([x], pred, 0)

    # aggregate: the same 2 x int64 payload rv_comm_allreduce_sum_count sends over RCCL
    t = torch.tensor([s, c], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)

    # filter/project: no collective on the data path; rank-order concat on the host
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.length], dtype=torch.int64))
    gathered = [None] * world
    dist.all_gather_object(gathered, local.logical_values().tolist())
    if rank == 0:
        json.dump({"sum": int(t[0]), "count": int(t[1]), "counts": [int(c[0]) for c in counts],
                   "rows": [v for part in gathered for v in part], "ranges": [capi.shard_range(n_global, world, r) for r in range(world)]},
                  open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
