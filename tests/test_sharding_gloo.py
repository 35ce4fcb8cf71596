"""The N>1 path on CPU: world_size 2 and 3 over gloo.  Each rank renders its share of the reference's 10-row tasks
(here with the CPU oracle standing in for the device, as a checker of the sharding logic only), the rows are
gathered to rank 0 with the same code bench.py uses over RCCL, and the result must equal the one-rank frame bit
for bit -- the image must not depend on the number of ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, asset

W, H, DEPTH, NS, SEED, TASK_ROWS = 96, 57, 4, 4, 4242, 10     # 57 rows: the last task is a partial one


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from raytracerwin_amd import sharding
    s = O.Scene()
    sh = s.add_mesh_obj(asset("TorusKnot.obj"))
    s.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    fb = O.Framebuffer(W, H)
    rows = sharding.task_rows_of_rank(H, TASK_ROWS, rank, world)
    for p in range(2):                                     # two accumulated passes
        for r in rows:
            s.render_range(fb, int(r) * W, int(r) * W + W - 1, DEPTH, False, p, NS, SEED)
    accum, argb = fb.read()
    ta = torch.from_numpy(accum.reshape(H, W * 4).copy())
    tc = torch.from_numpy(argb.astype(np.int64).reshape(H, W).copy())
    sharding.gather_rows([ta, tc], H, TASK_ROWS, rank, world, dist)
    if rank == 0:
        np.savez(out_path, accum=ta.numpy(), argb=tc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_rows_partition_is_exact():
    from raytracerwin_amd import sharding
    for height, rows, world in [(1080, 10, 8), (57, 10, 3), (5, 10, 4), (2160, 10, 8), (100, 7, 2)]:
        got = np.sort(np.concatenate([sharding.task_rows_of_rank(height, rows, r, world) for r in range(world)]))
        assert (got == np.arange(height)).all()


@pytest.mark.parametrize("world", [2, 3])
def test_gathered_frame_equals_single_rank_frame(tmp_path, oracle_mod, world):
    O = oracle_mod
    s = O.Scene()
    sh = s.add_mesh_obj(asset("TorusKnot.obj"))
    s.set_material(sh, [(O.MAT_DIFFUSE, (1, 1, 1), 0, 0, 0)])
    fb = O.Framebuffer(W, H)
    for p in range(2):
        s.render_range(fb, 0, W * H - 1, DEPTH, False, p, NS, SEED)
    accum, argb = fb.read()
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g = np.load(out)
    assert (g["accum"].reshape(-1, 4).view(np.uint32) == accum.view(np.uint32)).all()
    assert (g["argb"].reshape(-1) == argb.astype(np.int64)).all()
