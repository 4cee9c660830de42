#!/usr/bin/env python3
"""bench.py — Shannon-Fano encode+decode throughput of the MI355X codec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config[3] weak-scaled to one GPU — Zipf(1.2) bytes in 64 MiB
blocks (-b M), 8 GiB (128 blocks) per GPU by default; every rank owns its own shard of the global
stream (rank r generates bytes [r*shard, (r+1)*shard) on-device), no data-path collective.
One step = Module C over every resident block (SF bit-pack encode, tables from the block histograms
via Module T, prepared before the timed region like the reference's .cod file) followed by Module D
over every block (SF decode back to the bytes).  value = uncompressed GiB of all ranks / step time.
Inputs are resident in HBM when the timed region starts; outputs stay in HBM.

The JSON line also carries `roofline` (dominant kernel, HIP-event timed on the launch stream) and,
at N=1, `cpu_baseline` (the reference binary oracle/_ref/shafa when present, else the oracle port,
timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (8.0 TB/s)
GIB = float(1 << 30)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=128, help="64 MiB blocks per GPU")
    ap.add_argument("--block-mib", type=int, default=64)
    ap.add_argument("--dist", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16)
    ap.add_argument("--encode-only", action="store_true")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="also time X1 (root scatters whole blocks) + encode + X2 (root gathers the payloads); extra JSON field")
    ap.add_argument("--zipf-s", type=float, default=1.2, help="Zipf exponent of the synthetic bytes (metric config: 1.2)")
    return ap.parse_args()


def cpu_baseline(args, pkg, zt):
    """Reference CPU path timed on the host cores: Module C then Module D (SF only) of the reference
    binary on a bounded sample of the same stream (its own wall clock incl. file I/O on tmpfs, as the
    reference's timer does, c.c:321,463).  Falls back to the oracle port (1 thread) without it."""
    import oracle_lib
    orc = oracle_lib.load()
    nblk = args.cpu_sample_blocks
    bs = args.block_mib << 20
    n = nblk * bs
    data = orc.gen_bytes(20260101, n, zt if args.dist == "zipf" else None)
    cores = os.cpu_count() or 1
    ref = oracle_lib.REF_BIN
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    flag = {64: "M", 8: "m"}.get(args.block_mib)
    if os.path.exists(ref) and flag:
        with tempfile.TemporaryDirectory(dir=shm) as tmp:
            p = os.path.join(tmp, "s")
            data.tofile(p)
            subprocess.run([ref, "s", "-m", "f", "-b", flag], cwd=tmp, capture_output=True, check=True)
            stem = "s.rle" if os.path.exists(p + ".rle") else "s"
            subprocess.run([ref, stem + ".freq", "-m", "t"], cwd=tmp, capture_output=True, check=True)
            t0 = time.perf_counter()
            subprocess.run([ref, stem, "-m", "c"], cwd=tmp, capture_output=True, check=True)
            t1 = time.perf_counter()
            if stem == "s":
                os.remove(p)
            subprocess.run([ref, stem + ".shaf", "-m", "d", "-d", "s"], cwd=tmp, capture_output=True, check=True)
            t2 = time.perf_counter()
            in_bytes = os.path.getsize(os.path.join(tmp, stem))
        return {"value": in_bytes / GIB / (t2 - t0), "unit": "GiB/s", "cores": min(nblk, cores),
                "kind": "reference",
                "sample": f"{nblk} x {args.block_mib} MiB {args.dist} blocks; reference -m c then -m d -d s, "
                          f"one thread per block ({nblk} threads on {cores} cores), wall incl. tmpfs I/O",
                "encode_GiBs": in_bytes / GIB / (t1 - t0), "decode_GiBs": in_bytes / GIB / (t2 - t1)}
    # oracle port, single thread, smaller sample
    n = min(n, 32 << 20)
    d = data[:n]
    tab = orc.sf_build(orc.hist256(d))
    t0 = time.perf_counter()
    rc, enc = orc.sf_encode(d, tab)
    t1 = time.perf_counter()
    rc, dec = orc.sf_decode(enc, tab, n)
    t2 = time.perf_counter()
    return {"value": n / GIB / (t2 - t0), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": f"{n >> 20} MiB {args.dist}; oracle sf_encode + sf_decode, 1 thread",
            "encode_GiBs": n / GIB / (t1 - t0), "decode_GiBs": n / GIB / (t2 - t1)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import pkgload
    pkg = pkgload.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pkg.lib().shafa_hip_init(local)

    bs = args.block_mib << 20
    nb = args.blocks
    shard = nb * bs
    zt = pkg.zipf_table(args.zipf_s)
    st = torch.cuda.Stream(device=dev)

    # ---- resident inputs: this rank's shard of the global synthetic stream -----------------------
    d_in = torch.empty(shard, dtype=torch.uint8, device=dev)
    d_map = torch.from_numpy(zt).to(dev) if args.dist == "zipf" else None
    with torch.cuda.stream(st):
        pkg.gen_bytes(st, 20260101, rank * shard, d_in, shard, d_map)
    bt = pkg.Batch(nb, bs)
    in_off = [b * bs for b in range(nb)]
    in_n = [bs] * nb
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt.hist256(st, d_in, in_off, in_n, d_freq)
    bt.finish(st, nb)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    tables = bt._tables([pkg.sf_build_codes(freq[b]) for b in range(nb)])     # Module T, host
    lens = np.stack([tables[b].lens() for b in range(nb)]).astype(np.uint64)
    enc_bits = (freq * lens).sum(axis=1)
    enc_bytes = (enc_bits + 7) // 8
    cap = int(((int(enc_bytes.max()) + 4096 + 255) // 256) * 256)
    out_off = [b * cap for b in range(nb)]
    out_cap = [cap] * nb
    d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
    d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_dec = torch.empty(shard, dtype=torch.uint8, device=dev)

    have_decode = not args.encode_only

    def encode():
        bt.sf_encode(st, d_in, in_off, in_n, tables, d_enc, out_off, out_cap, d_enc_n)

    def decode():
        bt.sf_decode(st, d_enc, out_off, enc_bytes, tables, in_n, d_dec, in_off)

    # ---- correctness before timing (bit-exact round trip; encoded sizes = sum(freq*len)) ----------
    encode()
    bt.finish(st, nb)
    got_n = d_enc_n.cpu().numpy().astype(np.uint64)
    dbg = bool(os.environ.get("SHAFA_ENC_DBG"))          # timing experiments: outputs are wrong on purpose
    if dbg:
        have_decode = False
        os.environ["SHAFA_BENCH_ORACLE_CHECK"] = "0"
    assert dbg or (got_n == enc_bytes).all(), "encoded sizes differ from sum(freq*len)/8"
    if have_decode:
        try:
            decode()
            bt.finish(st, nb)
            assert torch.equal(d_dec, d_in), "decode(encode(x)) != x"
        except pkg.ShafaError as e:
            if e.code != pkg.OUTSIDE_MODULE:
                raise
            have_decode = False
    if rank == 0 and os.environ.get("SHAFA_BENCH_ORACLE_CHECK", "1") == "1":
        import oracle_lib
        orc = oracle_lib.load()
        blk = d_in[:min(bs, 4 << 20)].cpu().numpy()           # bounded oracle spot check of block 0's head
        otab = orc.sf_build(orc.hist256(blk))
        rc, want = orc.sf_encode(blk, otab)
        t = pkg.CodeTable()
        import ctypes as C
        C.memmove(C.byref(t), C.byref(otab), C.sizeof(t))
        assert pkg.sf_encode(blk, t).tobytes() == want.tobytes(), "HIP encode differs from oracle"

    def step():
        encode()
        if have_decode:
            decode()

    for _ in range(args.warmup):
        step()
    bt.finish(st, nb)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[3 * i].record(st)
        encode()
        ev[3 * i + 1].record(st)
        if have_decode:
            decode()
        ev[3 * i + 2].record(st)
    st.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    rc, errs = bt.finish(st, nb)
    assert rc == 0
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    enc_ms = [ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(args.steps)]
    dec_ms = [ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(args.steps)]
    enc_t = float(np.mean(enc_ms)) * 1e-3
    dec_t = float(np.mean(dec_ms)) * 1e-3 if have_decode else 0.0
    total_in = float(shard)
    total_enc = float(enc_bytes.sum())
    # algorithmic bytes per launch: encode n + enc ; decode enc + n
    alg = total_in + total_enc
    enc_gbs = alg / enc_t / 1e9
    dec_gbs = alg / dec_t / 1e9 if have_decode else None
    # HBM traffic per launch: measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
    # (tools/gpu_pmc_traffic.sh, gfx950 FETCH correction applied) on this workload; scaled by the launch's bytes.
    traffic = {"sf_encode": None, "sf_decode": None}
    tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if args.dist == "zipf" and args.zipf_s == 1.2 and args.block_mib == 64 and os.path.exists(tpath):
        with open(tpath) as f:
            per_byte = json.load(f)["bytes_per_input_byte"]
        traffic = {k: per_byte[k] * total_in for k in traffic}
    dominant = "sf_decode" if have_decode and dec_t > enc_t else "sf_encode"
    roof = {"bound": "hbm", "kernel": dominant,
            "achieved": dec_gbs if dominant == "sf_decode" else enc_gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic[dominant],
            "traffic_source": "profiles/r1_traffic.json (rocprofv3 PMC, separate run)" if traffic[dominant] else None}
    roof["frac"] = roof["achieved"] / HBM_PEAK_GBS

    # ---- optional: root-scatter-included encode: never part of `value` -------------
    sg = None
    if args.scatter_gather:
        import pkgload as _pl
        shd = _pl.load_submodule("sharding")
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1)
        total_all = world * shard
        src = None
        if rank == 0:                                    # the reader rank holds the whole stream
            src = torch.empty(total_all, dtype=torch.uint8, device=dev)
            pkg.gen_bytes(None, 20260101, 0, src, total_all, d_map)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        local, first_blk, sizes = shd.scatter_blocks(src, total_all, bs, dev)              # X1
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        bt.sf_encode(st, local, in_off, in_n, tables, d_enc, out_off, out_cap, d_enc_n)     # same tables: same stream
        bt.finish(st, nb)
        t2 = time.perf_counter()
        got = shd.gather_payloads(d_enc, out_off, [int(x) for x in enc_bytes], world * nb, dev)   # X2
        torch.cuda.synchronize()
        dist.barrier()
        t3 = time.perf_counter()
        assert torch.equal(local, d_in), "scattered shard differs from the resident one"
        if rank == 0:
            assert len(got) == world * nb and all(g is not None for g in got)
            sg = {"x1_scatter_ms": (t1 - t0) * 1e3, "encode_ms": (t2 - t1) * 1e3, "x2_gather_ms": (t3 - t2) * 1e3,
                  "encode_GiBs_root_scatter_gather_included": total_all / GIB / (t3 - t0)}
    dist_name = args.dist if args.zipf_s == 1.2 or args.dist != 'zipf' else 'zipf(s=%g)' % args.zipf_s
    if rank == 0:
        out = {
            "metric": "GiB/s Shannon-Fano encode+decode, 64 MiB blocks, 1/2/4/8 GPUs; bit-exact",
            "value": world * total_in * args.steps / GIB / elapsed,
            "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"cfg4 shard: {nb} x {args.block_mib} MiB {dist_name} blocks per GPU "
                                   f"(-b {'M' if args.block_mib == 64 else args.block_mib}), Module C encode + "
                                   f"Module D decode" + ("" if have_decode else " [decode unavailable: encode only]"),
                       "blocks_per_gpu": nb, "block_bytes": bs, "parallelism": f"blocks sharded over {world} GPU(s)",
                       "compressed_ratio": total_enc / total_in},
            "encode_GiBs": world * total_in / GIB / enc_t,
            "decode_GiBs": world * total_in / GIB / dec_t if have_decode else None,
            "encode_ms": enc_t * 1e3, "decode_ms": dec_t * 1e3 if have_decode else None,
            "roofline": roof,
            "roofline_encode": {"bound": "hbm", "achieved": enc_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": enc_gbs / HBM_PEAK_GBS, "traffic": traffic["sf_encode"],
                                "algorithmic_bytes_per_launch": alg},
            "roofline_decode": ({"bound": "hbm", "achieved": dec_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": dec_gbs / HBM_PEAK_GBS, "traffic": traffic["sf_decode"],
                                 "algorithmic_bytes_per_launch": alg} if have_decode else None),
        }
        if sg:
            out["scatter_gather"] = sg
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, pkg, zt)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
