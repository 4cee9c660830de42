#!/usr/bin/env python3
"""bench.py — Shannon-Fano encode+decode throughput of the MI355X codec (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment the script starts the N ranks itself (torch.distributed.run, one
process per GPU, rendezvous on 127.0.0.1) before anything touches the GPU; under the driver's own
`python -m torch.distributed.run ... bench.py --gpus N` it is simply rank RANK of WORLD_SIZE.

Workload (config.workload): BASELINE config[3] weak-scaled to one GPU — the surveyed cfg-4 stream
"Zipf(1.2) bytes mod 256" (Shannon-Fano output 0.812 n) in 64 MiB blocks (-b M), 8 GiB
(128 blocks) per GPU by default; every rank owns its own shard of the global stream (rank r generates bytes
[r*shard, (r+1)*shard) on-device), no data-path collective.  One step = Module C over every resident block
(SF bit-pack encode, tables from the block histograms via Module T, prepared before the timed region like
the reference's .cod file) followed by Module D over every block (SF decode back to the bytes).
value = uncompressed GiB of all ranks / step time.  Inputs are resident in HBM when the timed region starts.

The JSON line also carries `roofline` (dominant kernel sequence, HIP-event timed on the launch stream),
`cpu_baseline` at N=1 (the reference binary oracle/_ref/shafa when present, else the oracle port, timed on
the host cores on a bounded sample), `scatter_gather` at N>1 (X1/X2 over RCCL) and, with
--pipeline, the F -> T -> C and D(SF+RLE) legs at -b M.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "shafa-cd_amd")


def load_pkg(sub=None):
    """The product package by path (its directory name has a hyphen); `sub`: one of its submodules."""
    import importlib
    import importlib.util
    if "shafa_cd_amd" not in sys.modules:
        spec = importlib.util.spec_from_file_location("shafa_cd_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                      submodule_search_locations=[PKG_DIR])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["shafa_cd_amd"] = mod
        spec.loader.exec_module(mod)
    return importlib.import_module("shafa_cd_amd." + sub) if sub else sys.modules["shafa_cd_amd"]


def load_oracle():
    """tests/oracle_lib.py: the CPU oracle, for the pre-timing spot check and the cpu_baseline leg ONLY."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    return oracle_lib

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (8.0 TB/s)
GIB = float(1 << 30)
METRIC = "GiB/s Shannon-Fano encode+decode, 64 MiB blocks, 1/2/4/8 GPUs; bit-exact"
SEED = 20260101


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=128, help="blocks per GPU")
    ap.add_argument("--block-mib", type=int, default=64)
    ap.add_argument("--dist", default="zipfmod", choices=["zipfmod", "zipf", "uniform"],
                    help="zipfmod = Zipf(s) over the integers mod 256 (surveyed cfg-4 stream); zipf = Zipf(s) truncated to 256 ranks")
    ap.add_argument("--zipf-s", type=float, default=1.2)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-sample-blocks", type=int, default=16)
    ap.add_argument("--encode-only", action="store_true")
    ap.add_argument("--tiles", action="store_true",
                    help="time the F-fed encoder in the headline instead: Module F's tile histograms (prepared before the timed "
                         "region) -> one-shot grid sf_encode6.  Not Module C as c.c:306-472 defines it (input + .cod), so not "
                         "the default: the default line is the chained one-pass encoder and reports this one as `encode_f_fed`")
    ap.add_argument("--no-tiles", action="store_true", help="(kept for old command lines: the chained encoder is the default)")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="time X1 (root scatters whole blocks) + encode + X2 (root gathers the payloads) also at N=1")
    ap.add_argument("--no-scatter-gather", action="store_true", help="skip the X1/X2 leg at N>1")
    ap.add_argument("--pipeline", action="store_true",
                    help="(kept for old command lines: the pipeline leg runs by default)")
    ap.add_argument("--pipeline-blocks", type=int, default=32,
                    help="blocks of the pipeline leg (`pipeline` object: per-family times and roofline fractions of K1, K2, K5 "
                         "on run-heavy data at this block size; bounded, a fraction of a second of GPU time)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the pipeline legs")
    ap.add_argument("--pipeline-only", action="store_true",
                    help="run one pipeline leg and nothing else (for profiles of the leg alone: tools/gpu_prof.sh <tag> --pipeline-only --pipeline-kind K)")
    ap.add_argument("--pipeline-kind", default=None, choices=["runs", "mixed"], help="the leg's data (default: both legs)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the `host_path` object (layer-3 pipe PCIe-inclusive rates and the CLI end to end on a tmpfs file)")
    ap.add_argument("--host-path-blocks", type=int, default=32)
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="shafa_hip_set_option(NAME, VALUE) before anything runs (A/B of kernel variants)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="test aid: allow more ranks than GPUs (ranks share devices, collectives over gloo); never a valid scaling number")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """Start the N ranks as children of this process (which has not touched the GPU and never will)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def csrc_hash():
    """SHA-256 over the kernel sources: ties profiles/*_traffic.json to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "shafa-cd_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp")):
            with open(os.path.join(d, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()


def measured_traffic(workload_key):
    """HBM bytes per input byte from the newest profiles/r*_traffic.json whose csrc hash and workload match
    this tree (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/gpu_traffic.sh); None otherwise, so a
    stale constant is never reported."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if fn.endswith("_traffic.json"):
            try:
                with open(os.path.join(pdir, fn)) as f:
                    j = json.load(f)
            except Exception:
                continue
            # profiles taken with the pipeline leg on mix its launches into the per-kernel means: headline traffic
            # comes from a profile of the headline alone (--no-pipeline)
            if j.get("csrc_sha256") == csrc_hash() and j.get("workload_key") == workload_key and not j.get("pipeline_key"):
                best = (fn, j)
    return best


def dist_table(pkg, args):
    synth = load_pkg("synth")
    if args.dist == "zipfmod":
        return synth.zipf_mod256_table(args.zipf_s)
    if args.dist == "zipf":
        return synth.zipf_table(args.zipf_s)
    return None


def cpu_baseline(args, pkg, zt):
    """Reference CPU path timed on the host cores: Module C then Module D (SF only) of the reference
    binary on a bounded sample of the same stream (its own wall clock incl. file I/O on tmpfs, as the
    reference's timer does, c.c:321,463).  Falls back to the oracle port (1 thread) without it."""
    oracle_lib = load_oracle()
    orc = oracle_lib.load()
    nblk = args.cpu_sample_blocks
    bs = args.block_mib << 20
    n = nblk * bs
    data = orc.gen_bytes(SEED, n, zt)
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
        # the same two modules with --no-multithread (multithread.c:130-137: process and write inline, one core) on a prefix
        # of the sample — SURVEY.md §8(d) / BASELINE.md §3 ask for both figures
        st_blocks = min(nblk, 2)
        with tempfile.TemporaryDirectory(dir=shm) as tmp:
            p = os.path.join(tmp, "s")
            data[:st_blocks * bs].tofile(p)
            subprocess.run([ref, "s", "-m", "f", "-b", flag], cwd=tmp, capture_output=True, check=True)
            stem = "s.rle" if os.path.exists(p + ".rle") else "s"
            subprocess.run([ref, stem + ".freq", "-m", "t"], cwd=tmp, capture_output=True, check=True)
            u0 = time.perf_counter()
            subprocess.run([ref, stem, "-m", "c", "--no-multithread"], cwd=tmp, capture_output=True, check=True)
            u1 = time.perf_counter()
            if stem == "s":
                os.remove(p)
            subprocess.run([ref, stem + ".shaf", "-m", "d", "-d", "s", "--no-multithread"], cwd=tmp, capture_output=True, check=True)
            u2 = time.perf_counter()
            st_bytes = os.path.getsize(os.path.join(tmp, stem))
        return {"value": in_bytes / GIB / (t2 - t0), "unit": "GiB/s", "cores": min(nblk, cores),
                "kind": "reference",
                "sample": f"{nblk} x {args.block_mib} MiB {args.dist} blocks; reference -m c then -m d -d s, "
                          f"one thread per block ({nblk} threads on {cores} cores), wall incl. tmpfs I/O",
                "encode_GiBs": in_bytes / GIB / (t1 - t0), "decode_GiBs": in_bytes / GIB / (t2 - t1),
                "no_multithread": {"value": st_bytes / GIB / (u2 - u0), "cores": 1,
                                   "sample": f"{st_blocks} x {args.block_mib} MiB blocks, --no-multithread",
                                   "encode_GiBs": st_bytes / GIB / (u1 - u0), "decode_GiBs": st_bytes / GIB / (u2 - u1)}}
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


class Comm:
    """Barrier + max-reduce around the timed region.  RCCL (backend "nccl") with one GPU per rank; gloo only in
    the --oversubscribe test mode where ranks share a device."""

    def __init__(self, args, torch, dist):
        self.dist, self.torch = dist, torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
        self.backend = "nccl"
        if self.world > ndev:                  # decided from WORLD_SIZE, i.e. the same on every rank
            if not args.oversubscribe:
                raise SystemExit(f"bench.py: {self.world} ranks but only {ndev} GPU(s) visible "
                                 "(--oversubscribe shares devices for plumbing tests only)")
            self.backend = "gloo"
        self.device_index = self.local % ndev
        self.oversubscribed = self.backend == "gloo"
        self.dev = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.dev)               # before the process group: RCCL binds to the current device
        if self.world > 1 or args.scatter_gather:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def host_path_leg(args, pkg, torch, dev, d_in):
    """What sits between a file and the kernels, on the GPU box's host (never part of `value`): (1) layer 3, pinned host ->
    pinned host through H2D, kernels and D2H (bin/pipe_rate); (2) the CLI end to end on a file in tmpfs, bit-exact round
    trip included (bin/shafa; the reference's own drivers do the same four steps, shafa.c:157-250).  Child processes."""
    import re
    import shutil
    out = {}
    bs = args.block_mib << 20
    nb = max(1, min(args.host_path_blocks, args.blocks))
    exe_dir = os.path.join(PKG_DIR, "bin")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([os.path.join(exe_dir, "pipe_rate"), str(nb), str(args.block_mib), "3"], capture_output=True, text=True,
                       env=env, timeout=300)
    m = {k: float(v) for k, v in re.findall(r"(sf_encode|sf_decode)\s.*?([0-9.]+) GiB/s", r.stdout)}
    out["pipe_3_slots"] = {"blocks": nb, "sf_encode_GiBs": m.get("sf_encode"), "sf_decode_GiBs": m.get("sf_decode"),
                           "what": "layer 3, pinned host -> pinned host: H2D + kernels + D2H, 3 blocks in flight",
                           "round_trip_identical": "round trip identical" in r.stdout}
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="shafa_bench_", dir=shm)
    try:
        p = os.path.join(tmp, "z")
        with open(p, "wb") as f:
            for b in range(nb):
                f.write(d_in[b * bs:(b + 1) * bs].cpu().numpy().tobytes())
        h0 = hashlib.sha256(open(p, "rb").read()).hexdigest()
        flag = {64: "M", 8: "m"}.get(args.block_mib)
        cli = os.path.join(exe_dir, "shafa")
        times = {}
        for name, argv in (("f", [p, "-m", "f", "-b", flag]), ("t", [p + ".freq", "-m", "t"]), ("c", [p, "-m", "c"])):
            t0 = time.perf_counter()
            rr = subprocess.run([cli] + argv, capture_output=True, text=True, env=env, timeout=600)
            times[name] = time.perf_counter() - t0
            if rr.returncode != 0:
                raise RuntimeError(f"shafa -m {name}: rc {rr.returncode} {rr.stderr[-200:]}")
        # the default run, `shafa z -b M` (f, t and c in one process, shafa.c:293-298): one upload per block (host/modules.c
        # shafa_ftc_compress, layer 3 SHAFA_OP_FTC) against the same three modules as separate passes in one process (SHAFA_FTC=0)
        shaf0 = hashlib.sha256(open(p + ".shaf", "rb").read()).hexdigest()
        for name, ftc in (("ftc_one_upload", "2"), ("ftc_three_passes", "0")):
            for q in (".shaf", ".cod", ".freq", ".rle", ".rle.freq", ".rle.cod", ".rle.shaf"):
                if os.path.exists(p + q):
                    os.remove(p + q)
            t0 = time.perf_counter()
            rr = subprocess.run([cli, p, "-b", flag], capture_output=True, text=True, env=dict(env, SHAFA_FTC=ftc), timeout=600)
            times[name] = time.perf_counter() - t0
            if rr.returncode != 0:
                raise RuntimeError(f"shafa z -b {flag} (SHAFA_FTC={ftc}): rc {rr.returncode} {rr.stderr[-200:]}")
            assert hashlib.sha256(open(p + ".shaf", "rb").read()).hexdigest() == shaf0, "default run: .shaf differs from -m f / t / c"
        os.remove(p)
        t0 = time.perf_counter()
        rr = subprocess.run([cli, p + ".shaf", "-m", "d"], capture_output=True, text=True, env=env, timeout=600)
        times["d"] = time.perf_counter() - t0
        ok = rr.returncode == 0 and hashlib.sha256(open(p, "rb").read()).hexdigest() == h0
        gib = nb * bs / GIB
        out["cli_tmpfs"] = {"file_GiB": gib, "seconds": times, "GiBs": {k: gib / v for k, v in times.items() if k != "t"},
                            "round_trip_identical": ok,
                            "what": "bin/shafa -m f / t / c / d on a file in tmpfs, wall clock of each process (start-up, HIP "
                                    "initialisation, first use of the pinned buffers and teardown included: 0.4-0.5 s of each); the rest is bound by the host's tmpfs writes, DESIGN 1.1"}
        assert ok, "CLI round trip differs"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def pipeline_key(args, nb, kind):
    return f"pipeline:{kind}:{args.block_mib}:{nb}"


def measured_pipeline_traffic(pkey):
    """per-kernel HBM bytes per launch of a pipeline leg from the newest profiles/*_traffic.json with this csrc hash
    and pipeline key (tools/gpu_prof.sh <tag> --pipeline-only --pipeline-kind <kind>); None otherwise."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if fn.endswith("_traffic.json"):
            try:
                with open(os.path.join(pdir, fn)) as f:
                    j = json.load(f)
            except Exception:
                continue
            if j.get("csrc_sha256") == csrc_hash() and j.get("pipeline_key") == pkey:
                best = (fn, j)
    return best


PIPELINE_KINDS = {
    "runs": "Zipf(1.2) symbols in geometric runs (p = 0.35): RLE has work on every block (cfg-0 shape at -b M)",
    "mixed": "cfg[2] stand-in, the statistics of tests/golden/full_mixed_M: block 0 a binary section (records, zero / 0xFF "
             "padding runs), the other blocks dictionary text; block 0 accepts RLE for the whole file (f.c:250-258), so the "
             "text blocks are RLE-coded although they gain nothing",
}


def pipeline_blocks(kind, synth, torch, dev, bs, nb):
    """the leg's input, resident: nb blocks of bs bytes"""
    import numpy as np
    if kind == "runs":
        blk = torch.from_numpy(synth.runs_stream(11, bs, synth.zipf_table(1.2))).to(dev)
        return blk.repeat(nb)
    # generated in 8 MiB units on the host (the generators are numpy, seconds per 8 MiB) and tiled to the block size: the codec
    # has no memory beyond a run of bytes, so a block's RLE and Shannon-Fano statistics are its unit's
    unit = min(bs, 8 << 20)
    binu = torch.from_numpy(synth.binary_stream(21, unit)).to(dev)
    txtu = torch.from_numpy(synth.text_stream(22, unit)).to(dev)
    out = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
    for b in range(nb):
        u = binu if b == 0 else torch.roll(txtu, 4099 * b)
        out[b * bs:(b + 1) * bs] = u.repeat(bs // unit)
    return out


def pipeline_leg(args, pkg, torch, dev, st, steps, nb, kind):
    """F (RLE + histogram of the RLE bytes + their tile histograms) -> T (host) -> C (SF encode of the RLE bytes), then D (SF
    decode + RLE decode); per-family HIP-event times and algorithmic-byte rooflines; per-family HBM
    traffic from the PMC passes of a profile of THIS code and THIS leg alone, else null."""
    import numpy as np
    synth = load_pkg("synth")
    bs = args.block_mib << 20
    d_in = pipeline_blocks(kind, synth, torch, dev, bs, nb)
    bt = pkg.Batch(nb, 2 * bs + 64)
    off, n = [b * bs for b in range(nb)], [bs] * nb
    rcap = 2 * bs + 64
    roff = [b * rcap for b in range(nb)]
    d_rle = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    d_rle_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    thb = pkg.tile_hist_bytes(rcap)
    thoff = [b * thb for b in range(nb)]
    d_th = torch.zeros(nb * thb, dtype=torch.uint8, device=dev)

    def timed(fn):
        torch.cuda.synchronize()
        fn()
        bt.finish(st, nb)
        best = None                                        # launches of a few blocks take 50-500 us a call: one hiccup of the
        for _ in range(3 if nb <= 8 else 1):               #   host in a loop of `steps` calls doubles the mean, so the best of
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      #   three loops is reported there
            e0.record(st)
            for _ in range(steps):
                fn()
            e1.record(st)
            bt.finish(st, nb)
            t = e0.elapsed_time(e1) / steps * 1e-3
            best = t if best is None or t < best else best
        return best

    out = {"workload": f"{nb} x {args.block_mib} MiB blocks: " + PIPELINE_KINDS[kind]}
    # K1: make_freq of the input with its tile histograms (what Module F leaves when RLE is declined, f.c:325)
    t_h = timed(lambda: bt.hist256_tiles(st, d_in, off, n, d_freq, d_th, [b * pkg.tile_hist_bytes(bs) for b in range(nb)]))
    t_f = timed(lambda: bt.rle_encode_tiles(st, d_in, off, n, d_rle, roff, [rcap] * nb, d_rle_n, d_freq, d_th, thoff))
    rle_n = [int(x) for x in d_rle_n.cpu().numpy()]
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    t0 = time.perf_counter()
    tables = pkg.sf_build_codes_batch(freq)            # Module T for the launch's blocks (host: one call, up to 8 threads)
    t_t = time.perf_counter() - t0
    # Module T on the device (csrc/sf_tables.hip): the histograms never leave the GPU, the launch's tables come back in ONE copy
    # (8.25 KB a block) because the encoder's launcher picks its kernel forms from the code lengths on the host.  This is the
    # leg's T: its time goes into F_T_C_GiBs; the host's is reported next to it.  Bit-identical to the host's tables.
    import ctypes as C
    tsz = C.sizeof(pkg.CodeTable)
    d_tabs = torch.empty(nb * tsz, dtype=torch.uint8, device=dev)
    h_tabs = torch.empty(nb * tsz, dtype=torch.uint8).pin_memory()

    def device_t():
        bt.sf_build_codes(st, nb, d_freq, d_tabs)
        with torch.cuda.stream(st):
            h_tabs.copy_(d_tabs, non_blocking=True)
    torch.cuda.synchronize()
    device_t()
    bt.finish(st, nb)
    t0 = time.perf_counter()
    for _ in range(steps):
        device_t()
        bt.finish(st, nb)                              # the tables are needed on the host: every round ends in a synchronisation
    t_td = (time.perf_counter() - t0) / steps
    dev_tables = (pkg.CodeTable * nb).from_buffer_copy(h_tabs.numpy().tobytes())
    for b in range(nb):
        assert bytes(dev_tables[b]) == bytes(tables[b]), f"Module T on the device differs from the host's in block {b}"
    tables = [dev_tables[b] for b in range(nb)]
    lens = np.stack([tables[b].lens() for b in range(nb)]).astype(np.uint64)
    enc_bytes = [int(x) for x in ((freq * lens).sum(axis=1) + 7) // 8]
    cap = ((max(enc_bytes) + 4096 + 255) // 256) * 256
    eoff = [b * cap for b in range(nb)]
    d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
    d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    t_c = timed(lambda: bt.sf_encode_tiles(st, d_rle, roff, rle_n, tables, d_th, thoff, d_enc, eoff, [cap] * nb, d_enc_n))
    assert [int(x) for x in d_enc_n.cpu().numpy()] == enc_bytes
    d_sym = torch.empty(nb * rcap, dtype=torch.uint8, device=dev)
    t_ds = timed(lambda: bt.sf_decode(st, d_enc, eoff, enc_bytes, tables, rle_n, d_sym, roff))
    dcap = bs + 2048
    doff = [b * dcap for b in range(nb)]
    d_dec = torch.empty(nb * dcap, dtype=torch.uint8, device=dev)
    d_dec_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    t_dr = timed(lambda: bt.rle_decode(st, d_sym, roff, rle_n, d_dec, doff, [bs + 1024] * nb, d_dec_n))
    if os.environ.get("SHAFA_BENCH_ABLATION") != "2":             # (timing builds with wrong output, tools/dbg: the line says so)
        assert [int(x) for x in d_dec_n.cpu().numpy()] == n, "pipeline round trip: sizes differ"
        for b in range(nb):
            assert torch.equal(d_dec[doff[b]:doff[b] + bs], d_in[off[b]:off[b] + bs]), f"pipeline round trip differs in block {b}"
    tot, rle_tot, enc_tot = float(nb * bs), float(sum(rle_n)), float(sum(enc_bytes))

    # HBM traffic per family: the PMC passes (FETCH_SIZE x 2, WRITE_SIZE; separate runs) of a profile of this tree taken with
    # `--pipeline-only --pipeline-kind <kind>`, every kernel of the family summed; null without such a profile
    fams = {"K1_hist256": lambda k: k.startswith("hist256") and "<false>" in k,
            "K2_rle_encode_hist": lambda k: k.startswith("rle3_") or (k.startswith("hist256") and "<true>" in k),
            "K3_sf_encode": lambda k: k.startswith("sfe"),
            "K4_sf_decode": lambda k: k.startswith("sfd_"),
            "K5_rle_decode": lambda k: k.startswith("rle_decode")}
    traffic = {k: None for k in fams}
    tsrc = None
    m = measured_pipeline_traffic(pipeline_key(args, nb, kind))
    if m:
        pk = m[1]["per_kernel_bytes"]
        for fam, pred in fams.items():
            v = sum(x["read"] + x["written"] for k, x in pk.items() if pred(k))
            traffic[fam] = v or None
        tsrc = f"profiles/{m[0]}"

    def fam(alg, t, name):
        return {"ms": t * 1e3, "algorithmic_bytes": alg, "achieved_GBs": alg / t / 1e9, "frac": alg / t / 1e9 / HBM_PEAK_GBS,
                "GiBs_of_original": tot / GIB / t, "traffic": traffic[name]}

    out.update({"rle_ratio": rle_tot / tot, "sf_ratio_of_rle": enc_tot / rle_tot, "traffic_source": tsrc,
                "K1_hist256": fam(tot, t_h, "K1_hist256"), "K2_rle_encode_hist": fam(tot + rle_tot, t_f, "K2_rle_encode_hist"),
                "T_device_ms": t_td * 1e3, "T_host_ms": t_t * 1e3, "K3_sf_encode": fam(rle_tot + enc_tot, t_c, "K3_sf_encode"),
                "K4_sf_decode": fam(enc_tot + rle_tot, t_ds, "K4_sf_decode"), "K5_rle_decode": fam(rle_tot + tot, t_dr, "K5_rle_decode"),
                "F_T_C_GiBs": tot / GIB / (t_f + t_td + t_c), "D_GiBs": tot / GIB / (t_ds + t_dr),
                "tile_histograms": "K1 and K2 also write the 256 x u16 histogram of every 32 KiB tile (1.6 % of the bytes "
                                   "they count), which lets K3 run as a one-shot grid (sf_encode6)"})
    bt.close()
    return out


def small_launch_rows(args, pkg, torch, dev, st):
    """The `runs` pipeline leg at the launch sizes a file at -b M really has (1, 2 and 8 blocks): the fixed sequence of a launch
    weighs most there.  Compact rows: ms and fraction of the HBM peak per family (24 calls back to back,
    the best of three such loops)."""
    rows = {}
    for nb in (1, 2, 8):                               # (2: what the CLI launches at -b M — its groups close at 128 MiB)
        leg = pipeline_leg(args, pkg, torch, dev, st, 24, nb, "runs")
        rows[f"{nb}_blocks"] = {k: {"ms": round(v["ms"], 4), "frac": round(v["frac"], 4)} for k, v in leg.items()
                                if isinstance(v, dict) and "frac" in v}
        rows[f"{nb}_blocks"].update({"F_T_C_GiBs": leg["F_T_C_GiBs"], "D_GiBs": leg["D_GiBs"]})
    return rows


def uniform_leg(pkg, torch, dev, st):
    """cfg[1]: 1 GiB of uniform bytes at -b m (128 x 8 MiB), Module C encode and Module D decode.  Codes of 8 / 9 bits never
    re-synchronise, so the decoder takes its EXACT kernels (sfd_sync16 / sfd_countfsm / sfd_wstage), not the speculative ones:
    the path any incompressible section of a real file takes."""
    import numpy as np
    nb, bs = 128, 8 << 20
    d_in = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(st):
        pkg.gen_bytes(st, SEED + 1, 0, d_in, nb * bs, None)
    bt = pkg.Batch(nb, bs)
    off, n = [b * bs for b in range(nb)], [bs] * nb
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    bt.hist256(st, d_in, off, n, d_freq)
    bt.finish(st, nb)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    tables = pkg.sf_build_codes_batch(freq)
    lens = np.stack([tables[b].lens() for b in range(nb)]).astype(np.uint64)
    enc_bytes = [int(x) for x in ((freq * lens).sum(axis=1) + 7) // 8]
    cap = ((max(enc_bytes) + 4096 + 255) // 256) * 256
    eoff = [b * cap for b in range(nb)]
    d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
    d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_dec = torch.empty(nb * bs, dtype=torch.uint8, device=dev)

    def timed(fn, steps=8):
        torch.cuda.synchronize()
        fn()
        bt.finish(st, nb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps):
            fn()
        e1.record(st)
        bt.finish(st, nb)
        return e0.elapsed_time(e1) / steps * 1e-3

    t_c = timed(lambda: bt.sf_encode(st, d_in, off, n, tables, d_enc, eoff, [cap] * nb, d_enc_n))
    assert [int(x) for x in d_enc_n.cpu().numpy()] == enc_bytes
    t_d = timed(lambda: bt.sf_decode(st, d_enc, eoff, enc_bytes, tables, n, d_dec, off))
    assert torch.equal(d_dec, d_in), "uniform leg: decode differs from the input"
    alg = float(nb * bs + sum(enc_bytes))
    bt.close()
    return {"workload": "cfg[1]: 128 x 8 MiB uniform bytes (-b m)", "compressed_ratio": sum(enc_bytes) / float(nb * bs),
            "encode_ms": t_c * 1e3, "encode_frac": alg / t_c / 1e9 / HBM_PEAK_GBS,
            "decode_ms": t_d * 1e3, "decode_frac": alg / t_d / 1e9 / HBM_PEAK_GBS,
            "decode_path": "exact (no speculation: 8/9-bit codes do not re-synchronise)"}


def mover_leg():
    """What a kernel that only moves the encoder's bytes reaches on THIS device right now (tools/ubench/mover, built by
    __graft_entry__.build(): 8 GiB read, 6.5 GiB written, nothing computed): the ceiling the encoder's fraction is read against.
    A child process (never under a profiler); None when the binary is not there."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "mover")
    if not os.path.exists(exe):
        return None
    r = subprocess.run([exe, "headline"], capture_output=True, text=True, timeout=120)
    fr = [float(l.rsplit(":", 1)[1].strip(" )")) for l in r.stdout.splitlines() if "of 8 TB/s" in l]
    if len(fr) < 2:
        return None
    return {"one_shot": fr[0], "persistent": fr[1]}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))          # before importing torch / touching the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    pkg = load_pkg()

    comm = Comm(args, torch, dist)
    world, rank, dev = comm.world, comm.rank, comm.dev
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(dev)
    pkg.lib().shafa_hip_init(comm.device_index)
    for o in args.opt:
        k, v = o.split("=", 1)
        if pkg.lib().shafa_hip_set_option(k.encode(), int(v)) != 0:
            raise SystemExit(f"bench.py: unknown option {o}")

    bs = args.block_mib << 20
    nb = args.blocks
    shard = nb * bs
    zt = dist_table(pkg, args)
    st = torch.cuda.Stream(device=dev)

    if args.pipeline_only:                     # a profile of one leg alone (its kernels are not mixed with the headline's)
        pnb = max(1, min(args.blocks, args.pipeline_blocks))
        kind = args.pipeline_kind or "runs"
        leg = pipeline_leg(args, pkg, torch, dev, st, args.steps, pnb, kind)
        print(json.dumps({"metric": METRIC, "invalid": "--pipeline-only: no headline measurement", "pipeline_kind": kind,
                          "config": {"blocks_per_gpu": pnb, "block_bytes": bs}, "pipeline": leg}), flush=True)
        comm.close()
        return

    # ---- resident inputs: this rank's shard of the global synthetic stream -----------------------
    d_in = torch.empty(shard, dtype=torch.uint8, device=dev)
    d_map = torch.from_numpy(zt).to(dev) if zt is not None else None
    with torch.cuda.stream(st):
        pkg.gen_bytes(st, SEED, rank * shard, d_in, shard, d_map)
    bt = pkg.Batch(nb, bs)
    in_off = np.arange(nb, dtype=np.uint64) * np.uint64(bs)       # host arrays once, not per call: at -b m the launcher's
    in_n = np.full(nb, bs, dtype=np.uint64)                        # host time is of the order of the kernels' time
    d_freq = torch.zeros(nb * 256, dtype=torch.int64, device=dev)
    # Module F's products, prepared before the timed region like the .freq / .cod files: the block histograms — and, for the
    # F-fed encoder only (`encode_f_fed`, or the headline with --tiles), the histogram of every 32 KiB tile (include/shafa_hip.h
    # "Tile histograms").  The headline times Module C as the reference defines it: the input and the .cod tables, nothing else.
    use_tiles = args.tiles and not args.no_tiles
    side_tiles = not use_tiles and nb <= 256          # (a second copy of the encoded blocks: not at cfg[3]'s full 1024)
    thb = pkg.tile_hist_bytes(bs)
    th_off = np.arange(nb, dtype=np.uint64) * np.uint64(thb)
    d_th = torch.zeros(nb * thb if (use_tiles or side_tiles) else 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()              # torch's fills run on its own stream: finish them before ours starts
    if use_tiles or side_tiles:
        bt.hist256_tiles(st, d_in, in_off, in_n, d_freq, d_th, th_off)
    else:
        bt.hist256(st, d_in, in_off, in_n, d_freq)
    bt.finish(st, nb)
    freq = d_freq.cpu().numpy().astype(np.uint64).reshape(nb, 256)
    tables = pkg.sf_build_codes_batch(freq)                                     # Module T, host
    lens = np.stack([tables[b].lens() for b in range(nb)]).astype(np.uint64)
    enc_bits = (freq * lens).sum(axis=1)
    enc_bytes = (enc_bits + 7) // 8
    cap = int(((int(enc_bytes.max()) + 4096 + 255) // 256) * 256)
    out_off = np.arange(nb, dtype=np.uint64) * np.uint64(cap)
    out_cap = np.full(nb, cap, dtype=np.uint64)
    enc_bytes = enc_bytes.astype(np.uint64)
    d_enc = torch.empty(nb * cap, dtype=torch.uint8, device=dev)
    d_enc_n = torch.zeros(nb, dtype=torch.int64, device=dev)
    d_dec = torch.empty(shard, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    have_decode = not args.encode_only

    def encode_chained():
        bt.sf_encode(st, d_in, in_off, in_n, tables, d_enc, out_off, out_cap, d_enc_n)

    def encode_tiles():
        bt.sf_encode_tiles(st, d_in, in_off, in_n, tables, d_th, th_off, d_enc, out_off, out_cap, d_enc_n)

    encode = encode_tiles if use_tiles else encode_chained

    def decode():
        bt.sf_decode(st, d_enc, out_off, enc_bytes, tables, in_n, d_dec, in_off)

    # ---- correctness before timing (bit-exact round trip; encoded sizes = sum(freq*len)) ----------
    abl = os.environ.get("SHAFA_BENCH_ABLATION")                  # timing builds with wrong output (tools/dbg): no checks,
    ablation = abl in ("1", "2")                                  #   and the line says so.  "1": ablated encoder (no decode),
    if abl == "1":                                                #   "2": ablated decoder (decoded bytes are not compared)
        have_decode = False
    encode()
    bt.finish(st, nb)
    got_n = d_enc_n.cpu().numpy().astype(np.uint64)
    assert ablation or (got_n == enc_bytes).all(), "encoded sizes differ from sum(freq*len)/8"
    if have_decode:
        decode()
        bt.finish(st, nb)
        assert ablation or torch.equal(d_dec, d_in), "decode(encode(x)) != x"
    if rank == 0 and os.environ.get("SHAFA_BENCH_ORACLE_CHECK", "1") == "1" and not ablation:
        import ctypes as C
        orc = load_oracle().load()
        blk = d_in[:min(bs, 4 << 20)].cpu().numpy()           # bounded oracle spot check of block 0's head
        otab = orc.sf_build(orc.hist256(blk))
        rc, want = orc.sf_encode(blk, otab)
        t = pkg.CodeTable()
        C.memmove(C.byref(t), C.byref(otab), C.sizeof(t))
        assert pkg.sf_encode(blk, t).tobytes() == want.tobytes(), "HIP encode differs from oracle"

    # the other encoder on the same blocks: same bytes (and its time, for the line)
    other_t = None
    other = encode_chained if use_tiles else encode_tiles
    if (use_tiles or side_tiles) and not ablation and nb <= 256:
        ref_enc = d_enc.clone()
        d_enc.zero_()
        other()
        bt.finish(st, nb)
        for b in range(nb):
            o, m = int(out_off[b]), int(enc_bytes[b])
            assert torch.equal(d_enc[o:o + m], ref_enc[o:o + m]), f"tile path and chained encoder differ in block {b}"
        del ref_enc
        for _ in range(3):                                   # (the checks above left the GPU idle: clocks, first touches)
            other()
        bt.finish(st, nb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            other()
        e1.record(st)
        bt.finish(st, nb)
        other_t = e0.elapsed_time(e1) / 5 * 1e-3
        encode()
        bt.finish(st, nb)

    def step():
        encode()
        if have_decode:
            decode()

    for _ in range(args.warmup):
        step()
    bt.finish(st, nb)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    comm.barrier()
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
    comm.barrier()
    t1 = time.perf_counter()
    rc, errs = bt.finish(st, nb)
    assert rc == 0
    elapsed = comm.max(t1 - t0)

    enc_ms = [ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(args.steps)]
    dec_ms = [ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(args.steps)]
    enc_t = float(np.mean(enc_ms)) * 1e-3
    dec_t = float(np.mean(dec_ms)) * 1e-3 if have_decode else 0.0
    # every rank's own kernel-sequence times (HIP events on its stream): a straggler shows here
    per_rank = None
    if world > 1:
        t_ = torch.tensor([enc_t * 1e3, dec_t * 1e3], dtype=torch.float64, device=dev if comm.backend == "nccl" else "cpu")
        all_ = [torch.zeros_like(t_) for _ in range(world)]
        dist.all_gather(all_, t_)
        e_ = [float(x[0]) for x in all_]
        d_ = [float(x[1]) for x in all_]
    else:
        e_, d_ = [enc_t * 1e3], [dec_t * 1e3]
    per_rank = {"encode_ms": e_, "decode_ms": d_, "encode_ms_min": min(e_), "encode_ms_max": max(e_),
                "decode_ms_min": min(d_), "decode_ms_max": max(d_)}
    total_in = float(shard)
    total_enc = float(enc_bytes.sum())
    # algorithmic bytes per launch: encode n + enc ; decode enc + n
    alg = total_in + total_enc
    enc_gbs = alg / enc_t / 1e9
    dec_gbs = alg / dec_t / 1e9 if have_decode else None
    # HBM traffic per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_traffic.sh,
    # gfx950 FETCH correction applied) on this workload AND this csrc hash; null when no matching profile exists.
    wkey = f"{args.dist}:{args.zipf_s:g}:{args.block_mib}:{args.blocks}" + (":tiles" if use_tiles else ":chained")
    traffic = {"sf_encode": None, "sf_decode": None}
    tsrc = None
    m = measured_traffic(wkey)
    if m:
        per_byte = m[1]["bytes_per_input_byte"]
        traffic = {k: (per_byte[k] * total_in if per_byte.get(k) is not None else None) for k in traffic}
        tsrc = f"profiles/{m[0]} (rocprofv3 PMC, separate passes, csrc {m[1]['csrc_sha256'][:12]})"
    dominant = "sf_decode" if have_decode and dec_t > enc_t else "sf_encode"
    roof = {"bound": "hbm", "kernel": dominant,
            "achieved": dec_gbs if dominant == "sf_decode" else enc_gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic[dominant], "traffic_source": tsrc}
    roof["frac"] = roof["achieved"] / HBM_PEAK_GBS

    # ---- root-scatter-included encode: always at N>1, never part of `value` --------
    sg = None
    if ((world > 1 and not args.no_scatter_gather) or args.scatter_gather) and not comm.oversubscribed:
        def scatter_gather_leg():
            shd = load_pkg("sharding")
            sg_nb = min(nb, 16)                               # bounded: the root holds world * sg_nb blocks
            sg_shard = sg_nb * bs
            total_all = world * sg_shard
            src = None
            if rank == 0:                                     # the reader rank holds the whole stream
                src = torch.empty(total_all, dtype=torch.uint8, device=dev)
                for r in range(world):                        # rank r's first sg_nb blocks, as resident on rank r
                    pkg.gen_bytes(None, SEED, r * shard, src[r * sg_shard:(r + 1) * sg_shard], sg_shard, d_map)
            torch.cuda.synchronize()
            comm.barrier()
            t0 = time.perf_counter()
            local, first_blk, sizes = shd.scatter_blocks(src, total_all, bs, dev)              # X1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bt.sf_encode(st, local, in_off[:sg_nb], in_n[:sg_nb], tables[:sg_nb], d_enc, out_off[:sg_nb], out_cap[:sg_nb],
                         d_enc_n)                                            # same tables: same stream
            bt.finish(st, sg_nb)
            t2 = time.perf_counter()
            got = shd.gather_payloads(d_enc, [int(x) for x in out_off[:sg_nb]], [int(x) for x in enc_bytes[:sg_nb]], world * sg_nb, dev)   # X2
            torch.cuda.synchronize()
            comm.barrier()
            t3 = time.perf_counter()
            assert torch.equal(local, d_in[:sg_shard]), "scattered shard differs from the resident one"
            if rank != 0:
                return None
            assert len(got) == world * sg_nb and all(g is not None for g in got)
            return {"blocks_per_gpu": sg_nb, "backend": comm.backend,
                    "x1_scatter_ms": (t1 - t0) * 1e3, "encode_ms": (t2 - t1) * 1e3, "x2_gather_ms": (t3 - t2) * 1e3,
                    "encode_GiBs_resident_sharded": total_all / GIB / (t2 - t1),
                    "encode_GiBs_root_scatter_gather_included": total_all / GIB / (t3 - t0)}
        try:
            sg = scatter_gather_leg()
        except Exception as e:                             # the headline line must not depend on the X1/X2 leg
            sg = {"error": f"{type(e).__name__}: {e}"[:300]} if rank == 0 else None
        encode()                                           # restore d_enc for anything that follows
        bt.finish(st, nb)

    pipe, pipe_mixed = None, None
    if not args.no_pipeline and rank == 0:
        pnb = max(1, min(args.blocks, args.pipeline_blocks))
        legs = {}
        for kind in ([args.pipeline_kind] if args.pipeline_kind else ["runs", "mixed"]):
            try:
                legs[kind] = pipeline_leg(args, pkg, torch, dev, st, 8, pnb, kind)
            except AssertionError:
                raise                                  # a parity failure is never swallowed
            except Exception as e:                     # (memory on a small device: the headline line must still come out)
                legs[kind] = {"error": f"{type(e).__name__}: {e}"[:300]}
        pipe, pipe_mixed = legs.get("runs"), legs.get("mixed")
    small_rows, uni = None, None
    if not args.no_pipeline and rank == 0 and args.block_mib == 64:
        for name, fn in (("small", lambda: small_launch_rows(args, pkg, torch, dev, st)), ("uniform", lambda: uniform_leg(pkg, torch, dev, st))):
            try:
                r = fn()
            except AssertionError:
                raise
            except Exception as e:
                r = {"error": f"{type(e).__name__}: {e}"[:300]}
            if name == "small":
                small_rows = r
            else:
                uni = r

    host_path = None
    profiled = any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (world == 1 and rank == 0 and not args.no_host_path and not profiled      # no child processes under a profiler
            and args.dist != "uniform" and args.block_mib in (8, 64)):
        try:
            host_path = host_path_leg(args, pkg, torch, dev, d_in)
        except AssertionError:
            raise
        except Exception as e:
            host_path = {"error": f"{type(e).__name__}: {e}"[:300]}

    dist_name = {"zipfmod": f"Zipf({args.zipf_s:g}) mod 256", "zipf": f"Zipf({args.zipf_s:g}) truncated to 256 ranks",
                 "uniform": "uniform"}[args.dist]
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": world * total_in * args.steps / GIB / elapsed,
            "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"cfg4 shard: {nb} x {args.block_mib} MiB {dist_name} blocks per GPU "
                                   f"(-b {'M' if args.block_mib == 64 else 'm' if args.block_mib == 8 else args.block_mib}), Module C encode + "
                                   f"Module D decode" + ("" if have_decode else " [encode only]"),
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
            "encode_path": ("F-fed (--tiles): tile histograms from Module F (prepared before the timed region) -> tile offsets (dot "
                            "+ scan, inside the timed encode) -> one-shot grid sf_encode6" if use_tiles
                            else "Module C as c.c:306-472 defines it: the block and its .cod table -> chained one-pass encoder "
                                 "sf_encode4 (shafa_hipd_sf_encode)"),
        }
        if other_t and use_tiles:
            out["encode_chained"] = {"ms": other_t * 1e3, "frac": alg / other_t / 1e9 / HBM_PEAK_GBS,
                                     "what": "the same blocks through shafa_hipd_sf_encode (no tile histograms): sf_encode4's chained scan"}
        elif other_t:
            out["encode_f_fed"] = {"ms": other_t * 1e3, "frac": alg / other_t / 1e9 / HBM_PEAK_GBS,
                                   "what": "the same blocks through shafa_hipd_sf_encode_tiles, fed Module F's 32 KiB tile histograms "
                                           "(computed outside this time: hist256_tiles / rle_encode_tiles leave them, one more pass "
                                           "over the input when they have to be made for the encoder alone): the C stage of a "
                                           "device-resident F -> T -> C, not Module C on its own"}
            out["roofline_encode_f_fed"] = out["encode_f_fed"]["frac"]
        out["per_rank"] = per_rank
        if comm.oversubscribed:
            out["invalid"] = "oversubscribed: ranks share GPUs (plumbing test only)"
        if ablation:
            out["invalid"] = "SHAFA_BENCH_ABLATION=1: timing build, results unchecked"
        if sg:
            out["scatter_gather"] = sg
        if pipe:
            out["pipeline"] = pipe
        if pipe_mixed:
            out["pipeline_mixed"] = pipe_mixed
        if small_rows:
            out["pipeline_small_launches"] = small_rows
        if uni:
            out["uniform_stream"] = uni
        if world == 1 and not profiled and args.block_mib == 64 and args.dist == "zipfmod" and not args.no_host_path:
            try:
                mv = mover_leg()                       # a child process: after everything that is timed in this one
            except Exception:
                mv = None
            if mv:
                out["roofline_encode"]["mover_frac"] = mv["one_shot"]
                out["roofline_encode"]["mover_persistent_frac"] = mv["persistent"]
                out["roofline_encode"]["mover"] = ("tools/ubench/mover on this device in this run: a kernel that only moves the encoder's "
                                                    "bytes (8 GiB read, 6.5 GiB written, nothing computed) as a one-shot grid of 32 KiB "
                                                    "workgroups / as a persistent grid; the chained encoder is a persistent kernel")
        if host_path:
            out["host_path"] = host_path
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, pkg, zt)
        print(json.dumps(out), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
