"""Multi-GPU sharding of a frame: the reference's NumTaskRows-row tasks (Src/RayTracerProgram.cpp:282,294-301)
dealt round-robin over the ranks, and the one exchange the path has -- gathering every rank's rows to rank 0.
Backend-agnostic (`nccl` = RCCL over xGMI on the GPUs, `gloo` in the CPU tests); pixels keep their global
indices and random streams, so the gathered image does not depend on the number of ranks."""
import numpy as np


def task_rows_of_rank(height, task_rows, rank, world):
    """Row indices (ascending) of the tasks t = rank, rank + world, ... of a `height`-row frame."""
    n_tasks = (height + task_rows - 1) // task_rows
    rows = [np.arange(t * task_rows, min((t + 1) * task_rows, height)) for t in range(rank, n_tasks, world)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def gather_rows(buffers, height, task_rows, rank, world, dist, device=None):
    """buffers: list of 2-D tensors shaped (height, k_i) holding this rank's rows at their global positions.
    Gathers the rows every rank owns into rank 0's tensors (in place).  One gather per buffer."""
    import torch
    rows = [task_rows_of_rank(height, task_rows, r, world) for r in range(world)]
    mine = torch.from_numpy(rows[rank]).to(device if device is not None else buffers[0].device)
    most = max(len(r) for r in rows)          # a gather needs equal shapes: ranks with one task fewer pad their block
    for buf in buffers:
        local = torch.zeros((most, buf.shape[1]), dtype=buf.dtype, device=buf.device)
        local[:len(rows[rank])] = buf.index_select(0, mine)
        if rank == 0:
            parts = [torch.empty((most, buf.shape[1]), dtype=buf.dtype, device=buf.device) for _ in range(world)]
        else:
            parts = None
        dist.gather(local, parts, dst=0)
        if rank == 0:
            for r in range(1, world):
                buf.index_copy_(0, torch.from_numpy(rows[r]).to(buf.device), parts[r][:len(rows[r])])
    return buffers
