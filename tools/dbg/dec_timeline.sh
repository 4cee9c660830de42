#!/bin/bash
# Kernel timeline of one sf_decode call (rocprofv3 --kernel-trace): start, duration and the gap in front of every kernel.
# usage (through gpurun): tools/dbg/dec_timeline.sh [zipf|runs|uniform] [blocks] [block MiB]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=/tmp/tl_$$; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $R/tools/dbg/dec_host_time.py ${1:-zipf} ${2:-32} ${3:-64} > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/t_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0] for r in rows]
# the last call with speculation: from the last sfd_tables to the sfd_wstage behind it
idx = [i for i, n in enumerate(names) if n.startswith('sfd_tables')]
i0 = idx[-1]
prev_end = None
t0 = int(rows[i0]['Start_Timestamp'])
for i in range(i0 - 2, len(rows)):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{names[i][:34]:36s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}")
    prev_end = e
    if names[i].startswith('sfd_wstage') or names[i].startswith('sfd_write'):
        break
PY
