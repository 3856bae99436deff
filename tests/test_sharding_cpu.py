"""Host-side logic of the sharded path without a GPU: the partitioner, and the per-iteration pose exchange protocol
(pack own rows → all-gather of equal-sized blocks → unpack the other ranks' rows) run by two processes over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import harness as H

pkg = H.load_product()


def test_partition_by_axis_equal_slabs():
    rng = np.random.default_rng(3)
    n = 1001
    pos = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    inv = np.ones(n, np.float32); inv[0] = 0.0; inv[500] = 0.0
    for nranks in (1, 2, 4, 8):
        owner = pkg.partition_by_axis(pos, inv, nranks, axis=0)
        assert owner[0] == -1 and owner[500] == -1
        dyn = owner >= 0
        counts = np.bincount(owner[dyn], minlength=nranks)
        assert counts.sum() == n - 2 and counts.max() - counts.min() <= 1
        # slabs are ordered along the axis
        for r in range(nranks - 1):
            assert pos[owner == r, 0].max() <= pos[owner == r + 1, 0].min()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _exchange_worker(rank, world, port, n, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    inv = np.ones(n, np.float32); inv[0] = 0.0
    owner = pkg.partition_by_axis(pos, inv, world, axis=2)
    # the same grouping the library builds on the device: solved bodies rank by rank
    perm = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)])
    starts = np.concatenate([[0], np.cumsum([np.sum(owner == r) for r in range(world)])])
    max_count = int(np.diff(starts).max())
    # every rank starts from the same global table, then "solves" (perturbs) only its own bodies
    table = np.zeros((n, 8), np.float32); table[:, :3] = pos; table[:, 7] = 1.0
    for it in range(3):
        mine = perm[starts[rank]:starts[rank + 1]]
        table[mine, :3] += np.float32(0.01 * (it + 1) * (rank + 1))
        send = np.zeros((max_count, 8), np.float32); send[:len(mine)] = table[mine]
        recv = [torch.zeros(max_count, 8) for _ in range(world)]
        dist.all_gather(recv, torch.from_numpy(send))
        for r in range(world):
            if r == rank:
                continue
            theirs = perm[starts[r]:starts[r + 1]]
            table[theirs] = recv[r].numpy()[:len(theirs)]
    np.save(os.path.join(out_dir, f"table{rank}.npy"), table)
    dist.destroy_process_group()


def test_pose_exchange_protocol_two_ranks_gloo(tmp_path):
    world, n = 2, 257
    mp.spawn(_exchange_worker, args=(world, _free_port(), n, 11, str(tmp_path)), nprocs=world, join=True)
    t0, t1 = np.load(tmp_path / "table0.npy"), np.load(tmp_path / "table1.npy")
    assert H.same_bits(t0, t1), "after the gathers every rank must hold the same global pose table"
    # and it equals the serial application of both ranks' updates
    rng = np.random.default_rng(11)
    pos = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    inv = np.ones(n, np.float32); inv[0] = 0.0
    owner = pkg.partition_by_axis(pos, inv, world, axis=2)
    want = pos.copy()
    for it in range(3):
        for r in range(world):
            want[owner == r] += np.float32(0.01 * (it + 1) * (r + 1))
    assert H.same_bits(t0[:, :3], want)
