"""CPU, world_size 2 over gloo: the sharding / all-gather plumbing of the multi-GPU path
(crypto_primitives_amd/parallel.py).  The per-rank compute is the CPU oracle here (a
stand-in for the per-GPU kernel call, allowed in tests); what is under test is that
row-sharding + all-gather reassembles exactly the single-process result, including ragged
row counts, and that elementwise sharding covers every element exactly once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, synth_elements, ints_to_array

BN254 = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, cols, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from crypto_primitives_amd.parallel import elementwise_sharded, matvec_rowsharded, shard_range
    from oracle.fp_oracle import Oracle
    o = Oracle(BN254, 4)
    A = ints_to_array(synth_elements(BN254, 4, rows * cols, 5), 4)
    x = ints_to_array(synth_elements(BN254, 4, cols, 6), 4)
    lo, hi = shard_range(rows, rank, world)

    def local_matvec(A_local, nrows, ncols, xv):
        y = o.matvec(A_local.numpy().view(np.uint64).reshape(-1, 4), nrows, ncols, xv.numpy().view(np.uint64))
        return torch.from_numpy(y.view(np.int64))

    A_local = torch.from_numpy(A.reshape(rows, cols, 4)[lo:hi].reshape(-1, 4).view(np.int64).copy())
    y = matvec_rowsharded(local_matvec, A_local, rows, cols, torch.from_numpy(x.view(np.int64)), 4)
    full = o.matvec(A, rows, cols, x)
    assert np.array_equal(y.numpy().view(np.uint64), full), "row-sharded mat-vec != single-process result"

    # elementwise: each rank its own slice, no collective; slices tile [0, n) exactly
    n = 1001
    a = torch.from_numpy(ints_to_array(synth_elements(BN254, 4, n, 7), 4).view(np.int64))
    b = torch.from_numpy(ints_to_array(synth_elements(BN254, 4, n, 8), 4).view(np.int64))
    elo, ehi, res = elementwise_sharded(
        lambda u, v: torch.from_numpy(o.mul(u.numpy().view(np.uint64), v.numpy().view(np.uint64)).view(np.int64)), n, a, b)
    exp = o.mul(a.numpy().view(np.uint64), b.numpy().view(np.uint64))
    assert np.array_equal(res.numpy().view(np.uint64), exp[elo:ehi])
    spans = [None] * world
    dist.all_gather_object(spans, (elo, ehi))
    assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("rows,cols", [(8, 33), (7, 40), (1, 5)])
def test_rowsharded_matvec_and_elementwise_world2(tmp_path, rows, cols):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, rows, cols, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_range_is_a_partition():
    from crypto_primitives_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1000, 2**28):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
