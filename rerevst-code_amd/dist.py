"""Multi-GPU plumbing: one process per GPU, frames sharded, ONE broadcast of the 70 KB
shared-state blob per (video, style) — over RCCL/xGMI on GPUs (torch.distributed backend
"nccl" is RCCL on ROCm), over gloo in the CPU tests.  There is no per-frame collective:
after compute() every frame depends only on the weights and the blob."""
import os

import numpy as np

STATE_FLOATS = 17536


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).
    Returns (rank, world, local_rank); a no-op (0,1,0) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(local)
                kw["device_id"] = torch.device("cuda", local)
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def broadcast_state(blob, src=0, n_styles=1):
    """Broadcast the state blob(s) ([n_styles*17536] float32 numpy, None on non-source ranks)
    from `src`; returns the numpy blob on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blob
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    t = torch.empty(n_styles * STATE_FLOATS, dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32).reshape(-1)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
