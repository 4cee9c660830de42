#!/usr/bin/env python3
"""Summarise one tools/gpu_prof.sh run: per-kernel average duration (rocprofv3 --stats), SQ counters per dispatch
(upper-half mean: warm-up and small helper dispatches are ignored), and HBM traffic per kernel with the gfx950
FETCH_SIZE correction (x2 for wide coalesced reads, MI355X_MICROARCH.md §HBM).  Also writes
<dir>/traffic.json (bytes per input byte for sf_encode / sf_decode, tied to the csrc hash) for bench.py."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
args = sys.argv[3:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


def csrc_hash():
    h = hashlib.sha256()
    c = os.path.join(ROOT, "shafa-cd_amd", "csrc")
    for fn in sorted(os.listdir(c)):
        if fn.endswith((".hip", ".hpp")):
            with open(os.path.join(c, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()


print(f"# profile {tag}: bench.py {' '.join(args)}   csrc {csrc_hash()[:12]}")
bench = {}
try:
    with open(os.path.join(d, "bench.json")) as f:
        for line in f:
            if line.startswith("{"):
                bench = json.loads(line)
    print(f"# bench (under the profiler): encode {bench.get('encode_ms')} ms, decode {bench.get('decode_ms')} ms, "
          f"ratio {bench.get('config', {}).get('compressed_ratio')}")
except Exception as e:
    print("# no bench line:", e)

print("\n## kernel durations (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(d + "/kstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) < 0.3:
            continue
        print(f"{short(r['Name'])[:56]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:10.1f}  max_us {float(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):5.1f}%")


def counters(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {}
        for c, v in cs.items():
            big = sorted(v)[len(v) // 2:]
            out[k][c] = sum(big) / len(big)
    return out


sq = counters("pmc1")
for k, v in counters("pmc2").items():
    sq.setdefault(k, {}).update(v)
print("\n## SQ counters per dispatch (upper-half mean)")
for k in sorted(sq):
    if not any(x in k for x in ("sf", "rle", "hist")):
        continue
    c = sq[k]
    if c.get("SQ_INSTS_VALU", 0) < 1e5:
        continue
    print(k)
    print("   " + "  ".join(f"{n[3:]}={c[n]:.4g}" for n in sorted(c)))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS conflict share {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f}; "
              f"VALU per wave {c['SQ_INSTS_VALU'] / max(c.get('SQ_WAVES', 1), 1):.0f}")

fe, wr = counters("pmcF"), counters("pmcW")
print("\n## HBM traffic per dispatch, MB (FETCH_SIZE x2 = gfx950 correction for 16 B/lane streaming reads)")
per = {}
for k in sorted(set(fe) | set(wr)):
    if not any(x in k for x in ("sf", "rle", "hist")):
        continue
    r = fe.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2      # counter unit: KB
    w = wr.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    per[k] = (r, w)
    if r + w > 1e6:
        print(f"{k[:56]:58s} read {r / 1e6:10.1f}  written {w / 1e6:10.1f}")

total_in = bench.get("config", {}).get("blocks_per_gpu", 0) * bench.get("config", {}).get("block_bytes", 0)
if total_in:
    # the headline times ONE encoder (the chained sfe5 by default, the F-fed sfe6 family with --tiles); the other one in the
    # trace is bench.py's comparison leg (`encode_f_fed` / `encode_chained`), not part of a launch
    tiles = "--tiles" in args
    enc = sum(r + w for k, (r, w) in per.items()
              if k.startswith("sfe") and (k.startswith("sfe6") == tiles or not any(q.startswith("sfe6_kernel") for q in per)))
    dec = sum(r + w for k, (r, w) in per.items() if k.startswith("sfd"))
    ratio = bench["config"].get("compressed_ratio") or 0.0
    print(f"\n## bytes per input byte: sf_encode {enc / total_in:.3f}  sf_decode {dec / total_in:.3f}  (algorithmic {1 + ratio:.3f})")
    a = {"--dist": "zipfmod", "--zipf-s": "1.2", "--block-mib": "64", "--blocks": "128"}
    for i, x in enumerate(args):
        if x in a and i + 1 < len(args):
            a[x] = args[i + 1]
    j = {"what": "HBM traffic from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, tools/gpu_prof.sh); "
                 "FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes",
         "workload_key": f"{a['--dist']}:{float(a['--zipf-s']):g}:{a['--block-mib']}:{a['--blocks']}" + (":tiles" if "--tiles" in args else ":chained"),
         # a profile of one pipeline leg alone (--pipeline-only --pipeline-kind K): what bench.py's `pipeline` objects take
         # their per-family traffic from; a headline profile carries null
         "pipeline_key": ("pipeline:%s:%s:%s" % (next((args[i + 1] for i, x in enumerate(args) if x == "--pipeline-kind"), "runs"), a["--block-mib"],
                                                 min(int(a["--blocks"]), int(next((args[i + 1] for i, x in enumerate(args) if x == "--pipeline-blocks"), 32))))
                          if "--pipeline-only" in args else None),
         "bench_args": args, "csrc_sha256": csrc_hash(),
         "bytes_per_input_byte": {"sf_encode": enc / total_in if enc else None, "sf_decode": dec / total_in if dec else None},
         "algorithmic_bytes_per_input_byte": 1 + ratio,
         "per_kernel_bytes": {k: {"read": r, "written": w} for k, (r, w) in per.items() if r + w > 1e6}}
    with open(os.path.join(d, "traffic.json"), "w") as f:
        json.dump(j, f, indent=1)
