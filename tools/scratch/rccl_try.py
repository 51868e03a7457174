import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        x = torch.full((4,), float(rank), device="cuda", dtype=torch.float64)
        out = torch.empty(4*world, device="cuda", dtype=torch.float64)
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print("world", world, "rank", rank, "ok", out.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("world", world, "rank", rank, "FAILED", repr(e)[:300], flush=True)
if __name__ == "__main__":
    world = int(sys.argv[1])
    mp.spawn(w, args=(world, 29511 + world), nprocs=world, join=True)
