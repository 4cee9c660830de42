#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE binary.

Run in the dev container only (needs oracle/_ref/shafa, built by `make -C oracle` from the
sources under /root/reference).  The fixtures are DATA: deterministic inputs plus the files the
reference wrote for them (.rle / .freq / .cod / .shaf, decoded round trips) and, for large
outputs, SHA-256 + size only.  No reference source text is stored.

    python tests/golden/make_golden.py

Every case directory gets a manifest.json:
    {"cmds": [[argv...], ...], "files": {name: {"size": n, "sha256": h, "stored": bool}}, ...}
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "shafa")
STORE_LIMIT = 400 * 1024  # files above this are recorded by hash only


# ---------------------------------------------------------------- deterministic inputs
# the generators live in the product package (shafa-cd_amd/synth.py: bench.py uses them too); tests may depend on the
# product, not the other way round
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload  # noqa: E402

_synth = pkgload.load_submodule("synth")
splitmix64, gen_bytes, zipf_table = _synth.splitmix64, _synth.gen_bytes, _synth.zipf_table
zipf_mod256_table, runs_stream = _synth.zipf_mod256_table, _synth.runs_stream
mixed_file_stream = _synth.mixed_file_stream


def edge_stream():
    """Crafted run lengths around every threshold of f.c:38-52, incl. a run over a block edge."""
    parts = []
    rng_fill = gen_bytes(77, 70000)
    rng_fill = np.where(rng_fill < 2, 7, rng_fill).astype(np.uint8)  # filler without zeros/ones
    k = 0
    for sym in (0x41, 0x00, 0xFF):
        for L in list(range(1, 6)) + list(range(254, 261)) + list(range(509, 516)) + [765, 769]:
            parts.append(np.full(L, sym, dtype=np.uint8))
            parts.append(np.array([rng_fill[k] | 2, (rng_fill[k + 1] | 2) ^ 1], dtype=np.uint8))
            k += 2
    body = np.concatenate(parts)
    # a 300-byte run straddling the 65536 boundary
    pad = 65536 - 150 - body.size
    assert pad > 0
    filler = (gen_bytes(78, pad) | 1).astype(np.uint8)
    filler[-1] = 0x11
    stream = np.concatenate([body, filler, np.full(300, 0x42, dtype=np.uint8)])
    tail = (gen_bytes(79, 70000 - stream.size) | 1).astype(np.uint8)
    tail[0] = 0x13
    out = np.concatenate([stream, tail])
    assert out.size == 70000
    return out


def textlike_stream(seed, n):
    """Geometric-ish symbol distribution: a few hot symbols and a long tail of rare ones,
    so that Shannon-Fano code lengths spread from 1-2 bits to 16+ bits."""
    w = 0.62 ** np.arange(256, dtype=np.float64)
    cdf = np.cumsum(w) / np.sum(w)
    edges = np.minimum(np.floor(cdf * 65536.0 + 0.5).astype(np.int64), 65536)
    edges[-1] = 65536
    table = np.zeros(65536, dtype=np.uint8)
    lo = 0
    perm = (np.arange(256) * 37 + 11) % 256  # scatter symbol ids
    for k in range(256):
        table[lo:edges[k]] = perm[k]
        lo = max(lo, edges[k])
    out = gen_bytes(seed, n, table)
    # make sure every symbol appears at least once in block 0 (rare => long codes)
    out[1000:1256] = np.arange(256, dtype=np.uint8)
    return out


# ---------------------------------------------------------------- running the reference
BANNER_RE = re.compile(r"^[^\n]*,\s*a9\d{4},\s*MIEI/CD,[^\n]*\n", re.M)       # the authors' name lines (f.c:134-135 etc.)
RUNTIME_RE = re.compile(r"^(Module runtime \((?:in )?milliseconds\): )[0-9.]+$", re.M)


def mask_stdout(text):
    """What is compared of a module's stdout summary (f.c:132-177, t.c:219-243, c.c:282-303, d.c:44-65):
    everything except the two author-name lines each module prints first (dropped: our host does not
    print them) and the measured runtime (replaced by <ms>)."""
    return RUNTIME_RE.sub(r"\1<ms>", BANNER_RE.sub("", text))


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def make_input(gen):
    """Rebuild an input from its manifest entry {"kind": ..., "seed": ..., "n": ...} (used by the GPU tests
    for inputs that are too large to store)."""
    kind, n, seed = gen["kind"], gen["n"], gen.get("seed", 0)
    if kind == "uniform":
        return gen_bytes(seed, n)
    if kind == "zipf":
        return gen_bytes(seed, n, zipf_table(gen.get("s", 1.2)))
    if kind == "zipfmod":
        return gen_bytes(seed, n, zipf_mod256_table(gen.get("s", 1.2)))
    if kind == "runs":
        return runs_stream(seed, n, zipf_table(1.2))
    if kind == "textlike":
        return textlike_stream(seed, n)
    if kind == "mixed":
        return mixed_file_stream(seed, n, gen.get("binary_first", True))
    if kind == "const":
        return np.full(n, gen["byte"], dtype=np.uint8)
    if kind == "alt01":
        return (np.arange(n, dtype=np.uint32) & 1).astype(np.uint8)
    if kind == "zeros_then_uniform":
        z = gen["zeros"]
        return np.concatenate([np.zeros(z, dtype=np.uint8), gen_bytes(seed, n - z)])
    raise ValueError(kind)


def corrupt_cod_block(path, k):
    """Replace the first '0' / '1' of block k's codes in a .cod file by '2' (c.c:127-131: not a code character)."""
    with open(path, "rb") as f:
        fields = f.read().split(b"@")          # ['', 'R', '<n>', '<size>', '<codes>', '<size>', '<codes>', ..., '0']
    i = 4 + 2 * k
    codes = bytearray(fields[i])
    j = next(q for q, ch in enumerate(codes) if ch in b"01")
    codes[j] = ord("2")
    fields[i] = bytes(codes)
    with open(path, "wb") as f:
        f.write(b"@".join(fields))


def run_case(name, files_in, cmds, note="", store_inputs=True, expect_rc=None, generators=None):
    """files_in: {fname: bytes}; cmds: list of argv lists (without the binary), run in order in a
    scratch dir; files whose name starts with 'decoded__' are produced by copying after -m d."""
    if ONLY and name not in ONLY:
        return
    out_dir = os.path.join(HERE, name)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.makedirs(out_dir)
    man = {"note": note, "cmds": [], "files": {}, "inputs": sorted(files_in)}
    if generators:
        man["generators"] = generators
    with tempfile.TemporaryDirectory() as tmp:
        for fn, data in files_in.items():
            with open(os.path.join(tmp, fn), "wb") as f:
                f.write(data)
        for step in cmds:
            if step[0] == "__copy__":   # ["__copy__", src, dst]
                shutil.copyfile(os.path.join(tmp, step[1]), os.path.join(tmp, step[2]))
                man["cmds"].append(step)
                continue
            if step[0] == "__rm__":
                os.remove(os.path.join(tmp, step[1]))
                man["cmds"].append(step)
                continue
            if step[0] == "__corrupt_cod__":   # ["__corrupt_cod__", file, block]
                corrupt_cod_block(os.path.join(tmp, step[1]), step[2])
                man["cmds"].append(step)
                continue
            r = subprocess.run([REF] + step, cwd=tmp, capture_output=True, timeout=1800)
            man["cmds"].append({"argv": step, "rc": r.returncode,
                                "stderr": r.stderr.decode("utf-8", "replace"),
                                "stdout": mask_stdout(r.stdout.decode("utf-8", "replace"))})
        for fn in sorted(os.listdir(tmp)):
            p = os.path.join(tmp, fn)
            size = os.path.getsize(p)
            is_input = fn in files_in
            stored = size <= STORE_LIMIT and (store_inputs or not is_input)
            if fn.startswith(("decoded__", "orig__", "keep__")):
                stored = False          # copies of other files: the hash is enough
            man["files"][fn] = {"size": size, "sha256": sha(p), "stored": stored}
            if stored:
                shutil.copyfile(p, os.path.join(out_dir, fn))
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    total = sum(v["size"] for v in man["files"].values() if v["stored"])
    print(f"{name}: {len(man['files'])} files, {total} bytes stored")


ONLY = set(sys.argv[1:])          # `python make_golden.py full_mixed_M` regenerates just that case


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/shafa missing: run `make -C oracle` in the dev container first")
    zt = zipf_table(1.2)

    # 1. Zipf-with-runs, default 64 KiB blocks, 4 blocks incl. a partial one; full F->T->C then
    #    all three decode modes.
    d = runs_stream(1234, 212345, zt).tobytes()
    run_case("runs_default", {"x": d}, [
        ["x"],                                   # f+t+c  -> x.rle x.rle.freq x.rle.cod x.rle.shaf
        ["__copy__", "x", "orig__x"], ["__rm__", "x"],
        ["x.rle.shaf"],                          # d: SF + RLE -> x
        ["__copy__", "x", "decoded__sf_rle"], ["__rm__", "x"],
        ["__copy__", "x.rle", "keep__x.rle"],
        ["x.rle.shaf", "-m", "d", "-d", "s"],    # SF only -> x.rle
        ["__copy__", "x.rle", "decoded__sf_only"],
        ["x.rle", "-m", "d"],                    # RLE only (needs x.rle.freq) -> x
        ["__copy__", "x", "decoded__rle_only"],
    ], note="212345 B Zipf(1.2) symbols with geometric runs; default -b (64 KiB)")

    # 2. crafted run-length edges, RLE forced
    run_case("edges_forced_rle", {"e": edge_stream().tobytes()}, [
        ["e", "-m", "f", "-c", "r"], ["e.rle.freq", "-m", "t"], ["e.rle", "-m", "c"],
        ["__copy__", "e", "orig__e"], ["__rm__", "e"],
        ["e.rle.shaf"], ["__copy__", "e", "decoded__sf_rle"],
    ], note="runs of 1-5, 254-260, 509-515, 765, 769 of 0x41/0x00/0xFF; 300-run over the 64 KiB edge")

    # 3. uniform: RLE rejected by block 0 -> @N@ path
    run_case("uniform_no_rle", {"u": gen_bytes(42, 150000).tobytes()}, [
        ["u"], ["__copy__", "u", "orig__u"], ["__rm__", "u"],
        ["u.shaf"], ["__copy__", "u", "decoded__sf"],
    ], note="150000 uniform bytes; RLE rejected (<5 %), .freq/.cod/.shaf of the raw blocks")

    # 4. uniform with RLE forced (RLE expands) and -c f (both .freq files)
    run_case("uniform_forced_both", {"v": gen_bytes(43, 100000).tobytes()}, [
        ["v", "-m", "f", "-c", "r"],
        ["__copy__", "v.rle", "keep__v.rle"], ["__copy__", "v.rle.freq", "keep__v.rle.freq"],
    ], note="forced RLE on incompressible data")
    run_case("runs_force_freq", {"w": runs_stream(99, 90000, zt).tobytes()}, [
        ["w", "-m", "f", "-c", "f"],
    ], note="-c f: x.freq (@N@) next to x.rle/x.rle.freq (@R@)")

    # 5. config[0]: one 640 KiB block, -b K, Module F; runs variant (accepted) and uniform (rejected)
    run_case("cfg0_K_runs", {"k": runs_stream(7, 655360, zt).tobytes()}, [
        ["k", "-m", "f", "-b", "K"],
    ], note="BASELINE config[0]: single 640 KiB block, -b K, Module F; input = runs_stream(7, 655360)",
        store_inputs=False)
    run_case("cfg0_K_uniform", {"k": gen_bytes(8, 655360).tobytes()}, [
        ["k", "-m", "f", "-b", "K"],
    ], note="BASELINE config[0] uniform variant; input = gen_bytes(8, 655360)", store_inputs=False)

    # 6. text-like: code lengths from 1 to 16+ bits, 8 MiB-class block size flag but small file
    run_case("textlike_m", {"t": textlike_stream(5, 300000).tobytes()}, [
        ["t", "-b", "m"], ["__copy__", "t", "orig__t"], ["__rm__", "t"],
        ["t.shaf"], ["__copy__", "t", "decoded__sf"],
    ], note="geometric symbol distribution; long Shannon-Fano codes; one block at -b m")

    # 7. size limits (f.c:220,366): 1024 B is accepted, 1023 B is _FILE_TOO_SMALL
    run_case("tiny_1024", {"a": runs_stream(3, 1024, zt).tobytes()}, [["a"]])
    run_case("tiny_1023", {"a": runs_stream(3, 1023, zt).tobytes()}, [["a"]])

    # 8. Module T alone on hand-made .freq files: single symbol, ties, Fibonacci (deep codes)
    def freq_text(blocks, mode="N"):
        s = f"@{mode}@{len(blocks)}"
        for size, fr in blocks:
            s += f"@{size}@"
            prev = None
            for i, v in enumerate(fr):
                if prev is None or v != prev:
                    s += str(v)
                prev = v
                if i != 255:
                    s += ";"
        return (s + "@0").encode()

    single = [0] * 256
    single[65] = 5000
    ties = [100] * 256
    fib = [0] * 256
    a, b = 1, 1
    for i in range(40):
        fib[(i * 7) % 256] = a
        a, b = b, a + b
    two = [0] * 256
    two[3] = 10
    two[200] = 10
    geo = [max(1, int(60000 * 0.5 ** i)) for i in range(256)]
    run_case("t_handmade", {"h.freq": freq_text([(5000, single), (25600, ties), (sum(fib), fib),
                                                 (20, two), (sum(geo), geo)])}, [
        ["h.freq", "-m", "t"],
    ], note="Module T on hand-made histograms: single symbol (all codes empty), all ties, "
            "Fibonacci (deep tree), two symbols, geometric with a long tail of ones")

    # 10. full-size blocks (hash-only; inputs are rebuilt on the GPU box from "generators"): SURVEY.md §4
    MiB = 1 << 20

    def big_case(name, fn, gen, cmds, note):
        if ONLY and name not in ONLY:
            return
        run_case(name, {fn: make_input(gen).tobytes()}, cmds, note=note, store_inputs=False, generators={fn: gen})

    # cfg-1 shape: uniform bytes at -b m (8 MiB blocks), 2 blocks; F (RLE rejected) -> T -> C, then D
    big_case("full_uniform_m", "u", {"kind": "uniform", "seed": 4201, "n": 16 * MiB}, [
        ["u", "-b", "m"], ["__copy__", "u", "orig__u"], ["__rm__", "u"],
        ["u.shaf"], ["__copy__", "u", "decoded__sf"],
    ], "2 x 8 MiB uniform blocks, -b m: 8/9-bit codes")
    # cfg-2/3 shape: 2 x 64 MiB Zipf blocks at -b M, full F -> T -> C -> D
    big_case("full_zipf_M", "z", {"kind": "zipf", "seed": 4202, "n": 128 * MiB}, [
        ["z", "-b", "M"], ["__copy__", "z", "orig__z"], ["__rm__", "z"],
        ["z.shaf"], ["__copy__", "z", "decoded__sf"],
    ], "2 x 64 MiB Zipf(1.2, 256 ranks) blocks, -b M")
    # the surveyed cfg-4 stream (Zipf(1.2) mod 256), one 64 MiB block + a ragged 5 MiB + 77 B tail block, RLE forced
    big_case("full_zipfmod_M_forced_rle", "y", {"kind": "zipfmod", "seed": 4203, "n": 69 * MiB + 77}, [
        ["y", "-m", "f", "-c", "r", "-b", "M"], ["y.rle.freq", "-m", "t"], ["y.rle", "-m", "c"],
        ["__copy__", "y", "orig__y"], ["__rm__", "y"],
        ["y.rle.shaf"], ["__copy__", "y", "decoded__sf_rle"],
    ], "64 MiB + ragged block of Zipf(1.2) mod 256, -c r forced RLE (expands), F -> T -> C -> D(SF+RLE)")
    # one 64 MiB single-byte run: 263172 capped triples + remainder (f.c:38-52)
    big_case("full_single_run_M", "r", {"kind": "const", "byte": 0x41, "n": 64 * MiB}, [
        ["r", "-b", "M"], ["__copy__", "r", "orig__r"], ["__rm__", "r"],
        ["r.rle.shaf"], ["__copy__", "r", "decoded__sf_rle"],
    ], "64 MiB of 0x41: RLE gives 263172 {0,0x41,255} triples + {0,0x41,4}")
    # 0/1 alternation: every zero becomes a triple, RLE output = 2n (the f.c:244 worst-case class), forced
    big_case("full_alt01_M", "a", {"kind": "alt01", "n": 64 * MiB}, [
        ["a", "-m", "f", "-c", "r", "-b", "M"], ["a.rle.freq", "-m", "t"], ["a.rle", "-m", "c"],
        ["__copy__", "a", "orig__a"], ["__rm__", "a"],
        ["a.rle.shaf"], ["__copy__", "a", "decoded__sf_rle"],
    ], "64 MiB of 0,1,0,1,...: forced RLE doubles the block (128 MiB .rle)")
    # long-tail text-like block: code lengths beyond 16 bits
    big_case("full_longtail_M", "t", {"kind": "textlike", "seed": 4206, "n": 64 * MiB}, [
        ["t", "-b", "M"], ["__copy__", "t", "orig__t"], ["__rm__", "t"],
        ["t.shaf"], ["__copy__", "t", "decoded__sf"],
    ], "64 MiB geometric symbol distribution with every byte present: Lmax 17-32")

    # structured stand-in for BASELINE config[2] (Silesia is not available offline): a binary section (records, zero and
    # 0xFF padding runs) in block 0, dictionary text in blocks 1 and 2.  Block 0 alone decides RLE for the whole file
    # (f.c:250-258): it accepts, so the text blocks are RLE-coded although they would have declined.
    big_case("full_mixed_M", "s", {"kind": "mixed", "seed": 4207, "n": 192 * MiB, "binary_first": True}, [
        ["s", "-b", "M"], ["__copy__", "s", "orig__s"], ["__rm__", "s"],
        ["s.rle.shaf"], ["__copy__", "s", "decoded__sf_rle"],
    ], "3 x 64 MiB at -b M: binary section then dictionary text; RLE verdict of block 0 applied to the text blocks")

    # a first block far smaller (as .rle) than the ones behind it: the drivers size their groups from the first block
    # (host/modules.c run_groups: a group also closes on a byte budget, a large block goes alone)
    big_case("full_skewed_blocks_m", "k", {"kind": "zeros_then_uniform", "seed": 4208, "zeros": 8 * MiB, "n": 24 * MiB + 333}, [
        ["k", "-b", "m"], ["__copy__", "k", "orig__k"], ["__rm__", "k"],
        ["k.rle.shaf"], ["__copy__", "k", "decoded__sf_rle"],
    ], "8 MiB of zeros (98 KB of .rle), then 16 MiB + 333 B of noise, -b m: block 0 accepts RLE for the file")

    # 11. block-split edges through the drivers (file.c:78-85 last-block size, f.c:231-236 block loop)
    big_case("edge_exact_K", "q", {"kind": "runs", "seed": 21, "n": 2 * 655360}, [
        ["q", "-b", "K"], ["__copy__", "q", "orig__q"], ["__rm__", "q"],
        ["q.rle.shaf"], ["__copy__", "q", "decoded__sf_rle"],
    ], "a file of exactly two 640 KiB blocks at -b K: no short last block")
    for tail in (1, 7, 15):
        d = runs_stream(30 + tail, 3 * 65536 + tail, zt).tobytes()
        run_case(f"edge_tail_{tail}", {"p": d}, [
            ["p"], ["__copy__", "p", "orig__p"], ["__rm__", "p"],
            ["p.rle.shaf"], ["__copy__", "p", "decoded__sf_rle"],
            ["__copy__", "p.rle", "keep__p.rle"], ["__rm__", "p"],
            ["p.rle", "-m", "d"], ["__copy__", "p", "decoded__rle_only"],
        ], note=f"three 64 KiB blocks and a last block of {tail} byte(s), default block size")
    # more blocks than the drivers keep in flight (three slots per device), a malformed .cod block in the middle: the blocks
    # in front of it are on disk when the error surfaces (ordered write chain, c.c:254-267, multithread.c:75-86)
    big_case("edge_bad_cod_mid", "g", {"kind": "runs", "seed": 41, "n": 12 * 65536}, [
        ["g", "-m", "f"], ["g.rle.freq", "-m", "t"],
        ["__corrupt_cod__", "g.rle.cod", 6],
        ["g.rle", "-m", "c"],
    ], "12 blocks of 64 KiB, block 6 of the .cod file malformed: Module C fails, what it wrote before stays")

    # 12. hundreds of default-size blocks: the drivers take blocks of less than 2 MiB through the pipe in GROUPS (layer 3,
    #     shafa_pipe_submit_group) — several groups per slot, a ragged last block, every module and every decode mode
    big_case("many_default_rle", "m", {"kind": "runs", "seed": 51, "n": 700 * 65536 + 12345}, [
        ["m", "-c", "f"], ["__copy__", "m", "orig__m"], ["__rm__", "m"],
        ["m.rle.shaf"], ["__copy__", "m", "decoded__sf_rle"], ["__rm__", "m"],
        ["m.rle.shaf", "-m", "d", "-d", "s"], ["__copy__", "m.rle", "decoded__sf"],
        ["m.rle", "-m", "d", "-d", "r"], ["__copy__", "m", "decoded__rle_only"],
    ], "701 blocks of 64 KiB (runs: RLE accepted), -c f: F (RLE + both histograms), T, C, then D fused, D SF only, D RLE only")
    big_case("many_default_plain", "n", {"kind": "zipf", "seed": 52, "n": 530 * 65536 + 7}, [
        ["n"], ["__copy__", "n", "orig__n"], ["__rm__", "n"],
        ["n.shaf"], ["__copy__", "n", "decoded__sf"],
    ], "531 blocks of 64 KiB (Zipf: RLE declined by block 0), F (histograms only), T, C, D")
    big_case("many_default_single_run", "k", {"kind": "const", "byte": 0x41, "n": 40 * 65536 + 300}, [
        ["k"], ["__copy__", "k", "orig__k"], ["__rm__", "k"],
        ["k.rle.shaf"], ["__copy__", "k", "decoded__sf_rle"],
        ["__copy__", "k.rle", "keep__k.rle"], ["__rm__", "k"],
        ["k.rle", "-m", "d"], ["__copy__", "k", "decoded__rle_only"],
    ], "41 blocks of 64 KiB of one byte: every block's RLE bytes (774) expand 85-fold — the pipe's groups are sized for eightfold and decode again")
    big_case("many_default_bad_cod", "h", {"kind": "runs", "seed": 53, "n": 600 * 65536}, [
        ["h", "-m", "f"], ["h.rle.freq", "-m", "t"],
        ["__corrupt_cod__", "h.rle.cod", 437],
        ["h.rle", "-m", "c"],
    ], "600 blocks of 64 KiB, block 437 of the .cod file malformed (inside the second group of a slot): Module C fails there")

    # 9. CLI behaviour samples (exit codes + stderr text)
    run_case("cli_errors", {"z": runs_stream(11, 5000, zt).tobytes()}, [
        ["z", "-m", "f", "-m", "c"],          # c after f without t: error after F ran (shafa.c:193)
        ["z", "-m", "x"],                     # bad option value
        ["nonexistent"],                      # file can't be accessed
        ["z", "-m", "d", "-d", "s"],          # wrong extension
        ["z", "extra"],                       # two files
    ], note="argv errors: return codes and stderr texts")


if __name__ == "__main__":
    main()
