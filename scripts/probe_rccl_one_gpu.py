"""Can RCCL ("nccl" backend) run two ranks on ONE device?  (Every box of this build has one
GPU; DESIGN.md section 5 states what could and could not be exercised.)  Prints the outcome."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world)
        t = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "all_reduce ->", t.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:                                   # noqa: BLE001
        print("rank", rank, "FAILED:", type(e).__name__, str(e).splitlines()[0][:300], flush=True)
        sys.exit(3)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
