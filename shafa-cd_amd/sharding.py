"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed).

Blocks are independent: the data path needs no collective at all when every rank
already holds its shard (bench.py's default, "weak" scaling).  When a single reader/writer rank owns
the file — the reference's fread loop + ordered write callbacks (c.c:392-411, c.c:247) — the two
exchange steps are

    X1  scatter_blocks : root -> ranks, contiguous runs of whole blocks         (point-to-point)
    X2  gather_payloads: ranks -> root, per-block sizes (all_gather of int64) then the
                         variable-size payloads, re-assembled in block order    (gatherv by send/recv)

Both use torch.distributed point-to-point ops, so the same code runs over RCCL/xGMI ("nccl" backend,
CUDA tensors) and over gloo (CPU tensors; the world_size-2 tests).
"""
import torch
import torch.distributed as dist


def block_range(n_blocks, world, rank):
    """Contiguous, balanced split: ranks < n_blocks % world own one block more.  Returns (first, count)."""
    base, rem = divmod(n_blocks, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def block_sizes(total_bytes, block_size):
    """Sizes of the blocks of a file (reference utils/file.c:78-85: only the last one may be short)."""
    n = (total_bytes + block_size - 1) // block_size
    return [block_size] * (n - 1) + [total_bytes - (n - 1) * block_size] if n else []


def scatter_blocks(src, total_bytes, block_size, device, root=0, group=None, stats=None):
    """X1.  `src` (root only) holds the whole input; every rank gets a tensor with its own blocks,
    back to back.  Returns (local_tensor, first_block, sizes_of_local_blocks).  `stats` (a dict, optional) counts the
    point-to-point operations this rank posted ("p2p_ops"): a test that means to exercise them can tell."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = block_sizes(total_bytes, block_size)
    first, count = block_range(len(sizes), world, rank)
    my_bytes = sum(sizes[first:first + count])
    local = torch.empty(max(my_bytes, 1), dtype=torch.uint8, device=device)[:my_bytes]
    if rank == root:
        ops = []
        for r in range(world):
            f, c = block_range(len(sizes), world, r)
            lo, nb = f * block_size, sum(sizes[f:f + c])
            if r == root:
                local.copy_(src[lo:lo + nb])
            elif nb:
                ops.append(dist.P2POp(dist.isend, src[lo:lo + nb], r, group))
        if stats is not None:
            stats["p2p_ops"] = stats.get("p2p_ops", 0) + len(ops)
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
    elif my_bytes:
        if stats is not None:
            stats["p2p_ops"] = stats.get("p2p_ops", 0) + 1
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, local, root, group)]):
            w.wait()
    return local, first, sizes[first:first + count]


def gather_payloads(local_out, local_offsets, local_sizes, n_blocks, device, root=0, group=None, stats=None):
    """X2.  Block j of this rank is local_out[local_offsets[j] : local_offsets[j] + local_sizes[j]].
    Root returns the list of all payload tensors in global block order; other ranks return None.
    Every payload travels as its own view of local_out in ONE batch of point-to-point operations per rank (no packing
    copy on the sender; operations between two ranks match in the order they were posted)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    max_cnt = (n_blocks + world - 1) // world
    mine = torch.zeros(max_cnt, dtype=torch.int64, device=device)
    if local_sizes:
        mine[:len(local_sizes)] = torch.tensor([int(s) for s in local_sizes], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    views = [local_out[int(o):int(o) + int(s)] for o, s in zip(local_offsets, local_sizes)]
    if rank != root:
        ops = [dist.P2POp(dist.isend, v, root, group) for v in views if v.numel()]
        if stats is not None:
            stats["p2p_ops"] = stats.get("p2p_ops", 0) + len(ops)
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None
    out, ops = [None] * n_blocks, []
    for r in range(world):
        f, c = block_range(n_blocks, world, r)
        sz = [int(x) for x in every[r][:c].tolist()]
        if r == root:
            for j in range(c):
                out[f + j] = views[j]
            continue
        buf = torch.empty(max(sum(sz), 1), dtype=torch.uint8, device=device)
        pos = 0
        for j, s_ in enumerate(sz):
            out[f + j] = buf[pos:pos + s_]
            if s_:
                ops.append(dist.P2POp(dist.irecv, out[f + j], r, group))
            pos += s_
    if stats is not None:
        stats["p2p_ops"] = stats.get("p2p_ops", 0) + len(ops)
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return out
