#!/bin/bash
# CPU-side C under AddressSanitizer + UBSan (SURVEY.md §4/§5 "sanitizers"): the oracle restatement, the host's formats,
# Module T and the module drivers' host logic.  GPU code is out of reach of the sanitizers on this pool; nothing here
# needs a GPU.  Writes the report to profiles/r6_sanitizers.txt (SAN_OUT: another name).
#   1. make -C oracle SAN=1, make -C shafa-cd_amd/host SAN=1  (into _san/, next to the normal builds)
#   2. the parser / Module T corpus (tools/san/san_corpus.c)
#   3. malformed .freq FILES through the sanitized CLI's Module T (host-only: runs without a GPU)
#   4. the whole CPU test suite with the sanitized oracle, host library and CLI (runtimes preloaded into python)
# The sanitized CLI also runs on the GPU box (the module drivers' group / FIFO / pipe logic needs a device): the recipe and its
# record are section 5 of the report (tests/test_cli.py -m gpu and tools/soak_cli.py with SHAFA_CLI=shafa-cd_amd/host/_san/shafa).
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R" || exit 1
OUT=$R/profiles/${SAN_OUT:-r6_sanitizers.txt}
make -C oracle SAN=1 --no-print-directory > /dev/null || exit 1
make -C shafa-cd_amd/host SAN=1 --no-print-directory > /dev/null || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:exitcode=99 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=99
{
echo "# tools/san/run_san.sh — gcc $(gcc -dumpversion), -fsanitize=address,undefined -fno-sanitize-recover=undefined, $(date -u +%F)"
echo "## 2. parser / Module T corpus"
shafa-cd_amd/host/_san/san_corpus; echo "exit $?"
echo "## 3. malformed .freq files through the sanitized CLI (Module T alone)"
T=$(mktemp -d); bad=0; n=0
python3 - "$T" <<'PY'
import os, sys
d = sys.argv[1]
good = b"@N@2@1000@" + b";".join(str(i % 7).encode() for i in range(256)) + b"@24@" + b"1;" * 255 + b"1" + b"@0"
cases = {"good": good, "empty": b"", "at": b"@", "hdr_only": b"@N@2", "no_blocks": b"@N@0@0", "bad_mode": b"@X@1@5@1@0",
         "huge_n": b"@N@99999999999999999999@5@1@0", "neg": b"@N@-1@5@1@0", "short_block": b"@N@1@1000@1;2;3@0",
         "long_field": b"@N@1@10@" + b"9" * 400 + b";" * 255 + b"@0", "many_fields": b"@N@1@10@" + b"1;" * 400 + b"@0",
         "nul": b"@N@1@10@1;\x00;2" + b";" * 253 + b"@0", "no_trailer": good[:-2], "trunc": good[: len(good) // 2],
         "letters": b"@N@1@10@a;b;c" + b";" * 253 + b"@0", "rle_mode": good.replace(b"@N@", b"@R@")}
for k in range(1, 40):
    cases[f"cut{k}"] = good[: len(good) * k // 40]
for name, data in cases.items():
    with open(os.path.join(d, name + ".freq"), "wb") as f:
        f.write(data)
PY
for f in "$T"/*.freq; do
  LD_LIBRARY_PATH=$R/shafa-cd_amd/host/_san:$R/shafa-cd_amd shafa-cd_amd/host/_san/shafa "$f" -m t > /dev/null 2> "$T/err"; rc=$?
  n=$((n+1))
  if [ $rc -eq 99 ] || grep -q "ERROR: AddressSanitizer\|runtime error" "$T/err"; then bad=$((bad+1)); echo "REPORT for $(basename $f):"; head -20 "$T/err"; fi
done
echo "$n files, $bad with a sanitizer report"
rm -rf "$T"
echo "## 4. CPU test suite with the sanitized oracle, host library and CLI"
LD_PRELOAD="$ASAN:$UBSAN" LD_LIBRARY_PATH=$R/shafa-cd_amd/host/_san:$R/shafa-cd_amd SHAFA_ORACLE_LIB=$R/oracle/_san/libshafa_oracle.so \
  SHAFA_HOST_LIB=$R/shafa-cd_amd/host/_san/libshafa_host.so SHAFA_CLI=$R/shafa-cd_amd/host/_san/shafa \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} 2>&1 | tee "$OUT"
