"""Probe: can two RCCL ranks share one device on this box? (decides how the native neighbour path is tested)"""
import os, sys, torch, torch.distributed as dist
def main():
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    x = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    y = torch.zeros(4, device="cuda")
    ops = [dist.P2POp(dist.isend, x, 1 - rank), dist.P2POp(dist.irecv, y, 1 - rank)]
    for r in dist.batch_isend_irecv(ops): r.wait()
    torch.cuda.synchronize()
    print("rank", rank, "allreduce", x.tolist(), "p2p", y.tolist(), flush=True)
    dist.destroy_process_group()
main()
