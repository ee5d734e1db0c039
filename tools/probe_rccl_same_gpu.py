"""Does RCCL accept two ranks on ONE GPU on this box?  (decides whether the nccl path of the slab exchange can be tested here)"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda")
        peer = 1 - rank
        r = torch.empty_like(t)
        ops = [dist.P2POp(dist.isend, t, peer), dist.P2POp(dist.irecv, r, peer)]
        for q in dist.batch_isend_irecv(ops):
            q.wait()
        torch.cuda.synchronize()
        print("rank", rank, "received", float(r[0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print("rank", rank, "FAILED:", repr(e)[:300], flush=True)
        sys.exit(3)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
