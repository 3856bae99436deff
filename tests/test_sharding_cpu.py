"""Host-side logic of the sharded path without a GPU: the partitioner and the exchange layout are the LIBRARY's own host
functions (avbd_partition_by_axis, avbd_exchange_layout — the code the device path sizes its buffers and rows with); the
per-iteration exchange protocol built on them (every rank writes its rows into its block of the gather buffer, in-place
all-gather of equal blocks, everybody reads the other ranks' rows) is run by two processes over gloo."""
import os
import socket

import numpy as np
import pytest
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


def test_exchange_layout_rows_are_a_bijection_onto_the_blocks():
    rng = np.random.default_rng(9)
    for nranks in (1, 2, 3, 8):
        n = 500
        owner = rng.integers(-1, nranks, n).astype(np.int32)
        starts, perm, row, most = pkg.exchange_layout(owner, nranks)
        counts = np.bincount(owner[owner >= 0], minlength=nranks)
        assert most == counts.max() and list(np.diff(starts)) == list(counts)
        assert (row[owner < 0] == -1).all()
        block = max(1, most)
        for r in range(nranks):
            mine = np.nonzero(owner == r)[0]
            assert list(perm[starts[r]:starts[r + 1]]) == list(mine)                 # creation order inside a rank
            assert list(row[mine]) == list(r * block + np.arange(len(mine)))
    with pytest.raises(Exception):
        pkg.exchange_layout(np.array([0, 5], np.int32), 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _exchange_worker(rank, world, port, n, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    inv = np.ones(n, np.float32); inv[0] = 0.0
    owner = pkg.partition_by_axis(pos, inv, world, axis=2)
    # the library's own layout: rows of the gather buffers, blocks of max_count rows per rank
    starts, perm, row, max_count = pkg.exchange_layout(owner, world)
    block = max(1, max_count)
    # every rank starts from the same global table, then "solves" (perturbs) only its own bodies
    table = np.zeros((n, 8), np.float32); table[:, :3] = pos; table[:, 7] = 1.0
    gather = [np.zeros((world * block, 8), np.float32) for _ in range(2)]           # two buffers, alternating like the device path
    for it in range(3):
        mine = perm[starts[rank]:starts[rank + 1]]
        table[mine, :3] += np.float32(0.01 * (it + 1) * (rank + 1))
        g = gather[it & 1]
        g[row[mine]] = table[mine]                                                  # publish: each body's row of this rank's block
        blocks = [torch.zeros(block, 8) for _ in range(world)]
        dist.all_gather(blocks, torch.from_numpy(g[rank * block:(rank + 1) * block].copy()))   # in place: block r comes from rank r
        for r in range(world):
            g[r * block:(r + 1) * block] = blocks[r].numpy()
        theirs = np.nonzero((owner >= 0) & (owner != rank))[0]
        table[theirs] = g[row[theirs]]                                              # what a boundary task reads
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
