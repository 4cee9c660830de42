"""world_size-2 and -8 gloo tests of the multi-GPU path's host logic (block split, X1 scatter, X2 gatherv):
rank 0 owns the input, both ranks encode their blocks (here with the oracle standing in for the GPU
kernel — the data movement is what is under test), rank 0 re-assembles the payloads in block order
and compares with the single-process result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, total, bs, q):
    sys.path.insert(0, HERE)
    import oracle_lib
    import pkgload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = pkgload.load_submodule("sharding")
    orc = oracle_lib.load()
    dev = torch.device("cpu")
    src = torch.from_numpy(orc.gen_bytes(99, total)) if rank == 0 else None
    stats = {}
    local, first, sizes = sh.scatter_blocks(src, total, bs, dev, stats=stats)
    outs, offs, lens, pos = [], [], [], 0
    for s in sizes:
        blk = local[pos:pos + s].numpy()
        pos += s
        tab = orc.sf_build(orc.hist256(blk))
        rc, enc = orc.sf_encode(blk, tab)
        assert rc == 0
        offs.append(sum(len(o) for o in outs))
        lens.append(enc.size)
        outs.append(enc)
    packed = torch.from_numpy(np.concatenate(outs)) if outs else torch.empty(0, dtype=torch.uint8)
    n_blocks = (total + bs - 1) // bs
    got = sh.gather_payloads(packed, offs, lens, n_blocks, dev, stats=stats)
    if rank == 0:
        data = orc.gen_bytes(99, total)
        ok = len(got) == n_blocks
        # the root really talked to the others: one send per rank that holds blocks (X1) and one receive per remote block (X2)
        remote = sum(sh.block_range(n_blocks, world, r)[1] for r in range(1, world))
        holders = sum(1 for r in range(1, world) if sh.block_range(n_blocks, world, r)[1])
        ok = ok and stats.get("p2p_ops", 0) == holders + remote
        for b in range(n_blocks):
            blk = data[b * bs:(b + 1) * bs]
            rc, enc = orc.sf_encode(blk, orc.sf_build(orc.hist256(blk)))
            ok = ok and got[b].numpy().tobytes() == enc.tobytes()
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,bs", [(5 * 65536 + 777, 65536), (65536, 65536), (3 * 4096, 4096)])
def test_scatter_encode_gather_two_ranks(total, bs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("total,bs", [(13 * 4096 + 5, 4096), (3 * 4096, 4096)])
def test_scatter_encode_gather_eight_ranks(total, bs):
    """The world size of the node the scaling bench runs on: 14 blocks over 8 ranks (two ranks hold two blocks, the last
    block is ragged) and 3 blocks over 8 ranks (five ranks hold nothing and still take part in every collective)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 8, port, total, bs, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_block_range_is_a_partition(shafa):
    import pkgload
    sh = pkgload.load_submodule("sharding")
    for n in (0, 1, 7, 8, 9, 128, 1024):
        for w in (1, 2, 4, 8):
            seen = []
            for r in range(w):
                f, c = sh.block_range(n, w, r)
                seen += list(range(f, f + c))
            assert seen == list(range(n))
    assert sh.block_sizes(5 * 100 + 7, 100) == [100] * 5 + [7]
    assert sh.block_sizes(500, 100) == [100] * 5 and sh.block_sizes(0, 100) == []
