"""Multi-GPU path on hardware (-m gpu): X1 scatter -> HIP SF encode -> X2 gatherv over RCCL (backend "nccl") at the
world sizes the box offers (1 GPU: world 1, so the collectives at least run through RCCL once; more GPUs: one rank
per GPU), compared with the oracle; and bench.py's self-launching --gpus N path (c.c:392-411, multithread.c:70-87)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, total, bs, q):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import oracle_lib
    import pkgload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    pkg = pkgload.load()
    pkg.lib().shafa_hip_init(rank)
    sh = pkgload.load_submodule("sharding")
    orc = oracle_lib.load()
    zt = pkg.zipf_table(1.2)
    data = orc.gen_bytes(424242, total, zt)
    src = torch.from_numpy(data).to(dev) if rank == 0 else None
    stats = {}
    local, first, sizes = sh.scatter_blocks(src, total, bs, dev, stats=stats)         # X1
    nb = len(sizes)
    st = torch.cuda.Stream(device=dev)
    bt = pkg.Batch(max(nb, 1), bs)
    off = [b * bs for b in range(nb)]
    d_freq = torch.zeros(max(nb, 1) * 256, dtype=torch.int64, device=dev)
    got = None
    torch.cuda.synchronize()
    if nb:
        bt.hist256(st, local, off, sizes, d_freq)
        bt.finish(st, nb)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(-1, 256)[:nb]
    tables = [pkg.sf_build_codes(freq[b]) for b in range(nb)]
    lens = [t.lens().astype(np.uint64) for t in tables]
    enc_n = [int(((freq[b] * lens[b]).sum() + 7) // 8) for b in range(nb)]
    cap = (max(enc_n + [0]) + 4096 + 255) // 256 * 256
    eoff = [b * cap for b in range(nb)]
    d_enc = torch.empty(max(nb, 1) * cap, dtype=torch.uint8, device=dev)
    d_enc_n = torch.zeros(max(nb, 1), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    if nb:
        bt.sf_encode(st, local, off, sizes, tables, d_enc, eoff, [cap] * nb, d_enc_n)
        bt.finish(st, nb)
        assert [int(x) for x in d_enc_n.cpu().numpy()[:nb]] == enc_n
    n_blocks = (total + bs - 1) // bs
    got = sh.gather_payloads(d_enc, eoff, enc_n, n_blocks, dev, stats=stats)           # X2
    ok = True
    if rank == 0:
        ok = len(got) == n_blocks and stats.get("p2p_ops", 0) > 0       # blocks really went over RCCL (world >= 2)
        for b in range(n_blocks):
            blk = data[b * bs:(b + 1) * bs]
            rc, enc = orc.sf_encode(blk, orc.sf_build(orc.hist256(blk)))
            ok = ok and rc == 0 and got[b].cpu().numpy().tobytes() == enc.tobytes()
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def _gpu_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("total,bs", [(5 * (1 << 20) + 4096 + 16, 1 << 20), (3 * (8 << 20), 8 << 20)])
def test_scatter_hip_encode_gather_over_rccl(total, bs):
    import torch.multiprocessing as mp
    world = max(1, _gpu_count())                   # one rank per visible GPU (8 on a full node)
    if world < 2:
        # at world 1 every point-to-point list is empty: nothing of X1 / X2 would be exercised, so say so instead of passing;
        # the movement itself is covered at world 2 and 8 over gloo (tests/test_sharding_gloo.py), the HIP kernels everywhere else
        pytest.skip("one GPU on this box: X1 scatter / X2 gatherv over RCCL need two ranks (no send or receive happens at world 1)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + total) % 1500
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, bs, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _bench(argv, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment starts its own two ranks.  With one GPU on the box
    the ranks share it (--oversubscribe, gloo for the barrier / max-reduce): the launch, rendezvous, per-rank shard
    generation, barrier and rank-0 JSON are what is under test, the number is marked invalid."""
    n = _gpu_count()
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "4", "--block-mib", "8", "--no-cpu"]
    if n < 2:
        argv.append("--oversubscribe")
    j = _bench(argv)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert ("invalid" in j) == (n < 2)
    if n >= 2:
        assert "scatter_gather" in j and j["scatter_gather"]["backend"] == "nccl"


def test_bench_at_world_size_8():
    """`bench.py --gpus 8` with tiny blocks: the rank plumbing of the node the scaling bench runs on (rendezvous of eight
    ranks, shard offsets, barrier, max-reduce, per-rank gather, rank-0 line) is not first exercised by the driver.  On a box
    with fewer than eight GPUs the ranks share devices (--oversubscribe: gloo for the collectives, line marked invalid)."""
    n = _gpu_count()
    argv = ["--gpus", "8", "--steps", "1", "--warmup", "1", "--blocks", "2", "--block-mib", "8", "--no-cpu", "--no-pipeline",
            "--no-host-path"]
    if n < 8:
        argv.append("--oversubscribe")
    j = _bench(argv, timeout=900)
    assert j["n_gpus"] == 8 and j["value"] > 0 and j["scaling"] == "weak"
    assert ("invalid" in j) == (n < 8)
    pr = j["per_rank"]
    assert len(pr["encode_ms"]) == 8 and len(pr["decode_ms"]) == 8 and all(x > 0 for x in pr["encode_ms"] + pr["decode_ms"])
    assert j["config"]["blocks_per_gpu"] == 2


def test_bench_on_every_visible_gpu_over_rccl():
    """`bench.py --gpus <all visible>` (not oversubscribed): one rank per GPU over RCCL, the X1/X2 leg present with
    backend nccl, and per-rank encode/decode times in the line so that a straggler is visible.  On a one-GPU box this is
    the N=1 line with --scatter-gather (the collectives still run through RCCL)."""
    n = max(1, _gpu_count())
    argv = ["--gpus", str(n), "--steps", "2", "--warmup", "1", "--blocks", "8", "--block-mib", "8", "--no-cpu"]
    if n == 1:
        argv.append("--scatter-gather")
    j = _bench(argv, timeout=600)
    assert j["n_gpus"] == n and "invalid" not in j
    assert "scatter_gather" in j and "error" not in j["scatter_gather"] and j["scatter_gather"]["backend"] == "nccl"
    pr = j["per_rank"]
    assert len(pr["encode_ms"]) == n and len(pr["decode_ms"]) == n
    assert pr["encode_ms_min"] <= pr["encode_ms_max"] and pr["decode_ms_min"] <= pr["decode_ms_max"]
    assert all(x > 0 for x in pr["encode_ms"] + pr["decode_ms"])


def test_bench_single_gpu_line_has_the_contract_fields():
    j = _bench(["--steps", "2", "--warmup", "1", "--blocks", "8", "--cpu-sample-blocks", "1", "--scatter-gather", "--pipeline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["vs_baseline"] is None and j["dtype"] == "u8"
    r = j["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert j["cpu_baseline"]["kind"] in ("reference", "port")
    assert j["scatter_gather"]["backend"] == "nccl"
    p = j["pipeline"]
    for fam in ("K1_hist256", "K2_rle_encode_hist", "K3_sf_encode", "K4_sf_decode", "K5_rle_decode"):
        assert p[fam]["frac"] > 0
    assert 0.79 < j["config"]["compressed_ratio"] < 0.83      # the surveyed cfg-4 stream: SF output 0.812 n
    hp = j["host_path"]                                        # layer 3 and the CLI on the box's host, never part of `value`
    assert "error" not in hp and hp["pipe_3_slots"]["round_trip_identical"] and hp["cli_tmpfs"]["round_trip_identical"]
    assert hp["pipe_3_slots"]["sf_encode_GiBs"] > 1 and hp["cli_tmpfs"]["GiBs"]["c"] > 0.1
    assert len(j["per_rank"]["encode_ms"]) == 1
