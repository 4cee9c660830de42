"""Soak of the host CLI (shafa-cd_amd/bin/shafa) against the reference binary built from its own sources (oracle/_ref/shafa) for a
given number of seconds (default 300): random files (1 KiB .. 3 MiB; skewed, runs, uniform, zero-heavy, long runs, text-like),
random block sizes (-b K / m / M or the 64 KiB default), RLE forced or not (-c r), --no-multithread or not; both binaries run
the same session (F -> T -> C in one command or module by module, then D: both decoders, or -d s and then the RLE decoder) in
their own directories, and after every command the exit code, the stderr text, the stdout summary (runtime and the authors'
banner masked, as tests/test_cli.py does) and every file in the directory must be the same.  Sessions the reference itself
cannot finish (it crashes on a last block of one symbol) are counted and skipped.
usage (through gpurun): python tools/soak_cli.py [seconds] [seed]"""
import filecmp
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pkgload
from golden.make_golden import mask_stdout

synth = pkgload.load_submodule("synth")
OURS = os.environ.get("SHAFA_CLI") or os.path.join(ROOT, "shafa-cd_amd", "bin", "shafa")      # (SHAFA_CLI: the sanitized build, tools/san)
REF = os.path.join(ROOT, "oracle", "_ref", "shafa")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed0)
ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "shafa-cd_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))


def make(n):
    kind = int(rng.integers(0, 7))
    s = int(rng.integers(0, 1 << 30))
    r = np.random.default_rng(s)
    if kind == 0:
        return "zipf", synth.gen_bytes(s, n, synth.zipf_table(float(rng.uniform(0.8, 2.4))))
    if kind == 1:
        return "runs", synth.runs_stream(s, n, synth.zipf_table(float(rng.uniform(0.9, 2.0))))
    if kind == 2:
        return "uniform", r.integers(0, 256, size=n, dtype=np.uint8)
    if kind == 3:
        a = r.integers(0, 256, size=n, dtype=np.uint8)
        a[r.random(n) < float(rng.uniform(0.2, 0.9))] = 0
        return "zeros", a
    if kind == 4:
        out = np.empty(n, dtype=np.uint8)
        i = 0
        while i < n:
            ln = int(r.choice([1, 2, 3, 4, 5, 254, 255, 256, 257, 511, 765, int(r.integers(1, 3000))]))
            out[i:i + ln] = int(r.choice([0, 0, 1, 255, int(r.integers(0, 256))]))
            i += ln
        return "longruns", out
    if kind == 5:
        return "text", (32 + (r.zipf(1.3, size=n) % 95)).astype(np.uint8)
    a = synth.runs_stream(s, n, synth.zipf_table(1.2))
    lo = int(r.integers(0, n))
    hi = min(n, lo + int(r.integers(1, 200000)))
    a[lo:hi] = r.integers(0, 256, size=hi - lo, dtype=np.uint8)
    return "mixed", a


def run(binary, argv, cwd):
    try:
        r = subprocess.run([binary] + argv, cwd=cwd, capture_output=True, timeout=300, env=ENV)
    except subprocess.TimeoutExpired:
        return -999, "timeout", ""
    return r.returncode, r.stderr.decode("utf-8", "replace"), mask_stdout(r.stdout.decode("utf-8", "replace"))


def same_dirs(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    if fa != fb:
        return f"files differ: ours {fa}, reference {fb}"
    for f in fa:
        if not filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False):
            return f"{f} differs"
    return None


t0, sessions, cmds, skipped, nbytes = time.time(), 0, 0, 0, 0
base = tempfile.mkdtemp(prefix="shafa_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    while time.time() - t0 < budget:
        k = int(rng.integers(0, 10))
        n = int(rng.integers(1024, 70000)) if k < 3 else int(rng.integers(70000, 800000)) if k < 8 else int(rng.integers(800000, 3 << 20))
        if k == 9:                                  # block-split edges: whole blocks, or a short tail (file.c:78-85)
            n = 65536 * int(rng.integers(1, 6)) + int(rng.choice([0, 1, 7, 15, 1023, 1024, 1025]))
        kind, data = make(n)
        opts = []
        b = str(rng.choice(["", "", "K", "m", "M"]))
        if b:
            opts += ["-b", b]
        if int(rng.integers(0, 3)) == 0:
            opts += ["-c", "r"]
        nomt = ["--no-multithread"] if int(rng.integers(0, 4)) == 0 else []
        da, db = os.path.join(base, "ours"), os.path.join(base, "ref")
        for d in (da, db):
            shutil.rmtree(d, ignore_errors=True)
            os.makedirs(d)
            with open(os.path.join(d, "x"), "wb") as f:
                f.write(np.ascontiguousarray(data, dtype=np.uint8).tobytes())
        by_module = int(rng.integers(0, 2)) == 1
        tag = f"seed0={seed0} session={sessions} {kind} n={n} opts={opts + nomt} by_module={by_module}"
        session, crashed, stop = [], False, False

        def both(argv):
            global cmds
            ro = run(OURS, argv, da)
            rr = run(REF, argv, db)
            cmds += 1
            return ro, rr

        def step(argv):
            """-> False when the session cannot go on (an error both agree on, or the reference crashed)"""
            global crashed
            ro, rr = both(argv)
            if rr[0] < 0 or rr[0] > 1:              # the reference crashed (signal) — nothing to compare with
                crashed = True
                return False
            assert ro == rr, f"{tag}: {argv}\n ours      {ro}\n reference {rr}"
            d = same_dirs(da, db)
            assert d is None, f"{tag}: after {argv}: {d}"
            return ro[0] == 0

        try:
            ok = True
            if by_module:
                ok = step(["x", "-m", "f"] + opts + nomt)
                src = "x.rle" if os.path.exists(os.path.join(db, "x.rle.freq")) else "x"
                ok = ok and step([src + ".freq", "-m", "t"] + nomt)
                ok = ok and step([src, "-m", "c"] + nomt)
            else:
                ok = step(["x"] + opts + nomt)
                src = "x.rle" if os.path.exists(os.path.join(db, "x.rle.shaf")) else "x"
            if ok:
                for d in (da, db):
                    os.rename(os.path.join(d, "x"), os.path.join(d, "orig__x"))
                if src == "x.rle" and int(rng.integers(0, 2)) == 0:
                    for d in (da, db):
                        os.rename(os.path.join(d, "x.rle"), os.path.join(d, "kept__x.rle"))
                    ok = step(["x.rle.shaf", "-m", "d", "-d", "s"] + nomt) and step(["x.rle", "-m", "d"] + nomt)
                else:
                    ok = step([src + ".shaf"] + (["-m", "d"] if int(rng.integers(0, 2)) else []) + nomt)
                if ok:
                    assert filecmp.cmp(os.path.join(da, "x"), os.path.join(da, "orig__x"), shallow=False), f"{tag}: round trip"
        except AssertionError:
            print("FAILED:", tag, flush=True)
            keep = os.path.join(ROOT, "gpurun_out", "soak_cli_failure")
            shutil.rmtree(keep, ignore_errors=True)
            if n <= (1 << 20):
                shutil.copytree(base, keep)
            raise
        sessions += 1
        skipped += 1 if crashed else 0
        nbytes += n
finally:
    shutil.rmtree(base, ignore_errors=True)
print(f"soak_cli: seed {seed0}, {sessions} sessions ({cmds} commands per binary, {nbytes / 2**20:.0f} MiB of input), {time.time() - t0:.0f} s: "
      f"exit codes, messages, summaries and every file equal to the reference binary's; {skipped} sessions cut short where the "
      f"reference crashed")
