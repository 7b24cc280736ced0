"""probe: which 2-process collectives work on ONE GPU box (for the world-2 GPU test): nccl on a shared device, gloo with CUDA tensors"""
import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, backend, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group(backend, rank=rank, world_size=2)
        torch.cuda.set_device(0)
        t = torch.full((1 << 20,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[{backend}] rank {rank}: all_reduce ok -> {t[0].item()}", flush=True)
        d = torch.tensor([1.0 + rank], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(d); print(f"[{backend}] rank {rank}: fp64 ok {d.item()}", flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(f"[{backend}] rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)

if __name__ == "__main__":
    for i, backend in enumerate(sys.argv[1:] or ["gloo", "nccl"]):
        try:
            mp.spawn(worker, args=(backend, 29610 + i), nprocs=2, join=True)
        except Exception as e:
            print(f"[{backend}] spawn failed: {str(e)[:300]}")
