#!/bin/bash
# timeline of layer 3 (bin/pipe_rate): kernels and memory copies of the last blocks of the encode and decode passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
export TMPDIR=/tmp LD_LIBRARY_PATH=$R/shafa-cd_amd
out=$R/gpurun_out/pipe_trace
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -o r -- "$R/shafa-cd_amd/bin/pipe_rate" ${1:-8} 64 ${2:-3} > "$out/run.log" 2>&1
cat "$out/run.log" | tail -3
python3 - "$out" <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/r_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:36], r.get("Stream_Id", "")))
for f in glob.glob(d + "/**/r_memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    if rows: print(list(rows[0].keys()))
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s" % (r.get("Direction", r.get("Name", "")), r.get("Bytes", r.get("Size", ""))), r.get("Stream_Id", "")))
ev.sort()
big = [e for e in ev if e[1] - e[0] > 50000]      # > 50 us
t0 = big[0][0]
# the 45 big events in front of the first decode kernel: the steady state of the timed encode pass
first_dec = next(i for i, e in enumerate(big) if "sfd_" in e[2])
for s, e, n, st in big[max(0, first_dec - 45):first_dec]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  stream {st:>3s}  {n}")
PY
