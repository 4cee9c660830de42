"""Soak of every kernel family against the oracle for a given number of seconds (default 300): random inputs (skewed, uniform,
zero-heavy, long runs, small alphabets, text-like; sizes biased to the kernels' tile edges), random alternative kernel paths
(shafa_hip_set_option), and for each input the whole chain of the hot path, every stage compared with the oracle's:
    hist256 | rle_encode (+ histogram of its output) | Module T on the device's histogram | sf_encode | sf_decode | rle_decode
plus rle_decode and sf_decode of arbitrary bytes (return code and bytes).  The oracle is the checker here, as in tests/.
usage (through gpurun): python tools/soak_all.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import pkgload
import oracle_lib

shafa = pkgload.load()
oracle = oracle_lib.load()
synth = pkgload.load_submodule("synth")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed0)

DEFAULTS = {"sf_encode_one_pass_min_blocks": 0, "sf_encode_lanes": 0, "sf_encode_window_bits": 0, "sf_decode_speculate": 1,
            "sf_decode_path": 0, "rle_encode_general": 0, "rle_encode_one_pass": 0}
ALTS = [{}, {}, {}, {"sf_encode_one_pass_min_blocks": 1 << 30}, {"sf_encode_one_pass_min_blocks": 1},
        {"sf_encode_one_pass_min_blocks": 1, "sf_encode_lanes": 256}, {"sf_encode_one_pass_min_blocks": 1, "sf_encode_lanes": 512},
        {"sf_encode_one_pass_min_blocks": 1, "sf_encode_window_bits": 4}, {"sf_decode_speculate": 0}, {"sf_decode_speculate": 2},
        {"sf_decode_speculate": 2}, {"sf_decode_path": 1}, {"sf_decode_path": 2}, {"rle_encode_general": 1},
        {"rle_encode_one_pass": 1}, {"rle_encode_one_pass": 1, "rle_encode_general": 1}]
EDGES = [4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20]


def size():
    k = int(rng.integers(0, 10))
    if k == 0:
        return int(rng.integers(0, 70))
    if k <= 3:
        return max(0, int(rng.choice(EDGES)) * int(rng.integers(1, 4)) + int(rng.integers(-4, 5)))
    if k <= 6:
        return int(rng.integers(70, 40000))
    if k <= 8:
        return int(rng.integers(40000, 600000))
    return int(rng.integers(600000, 5 << 20))


def table_of(otab):
    t = shafa.CodeTable()
    C.memmove(C.byref(t), C.byref(otab), C.sizeof(t))
    return t


def make(n):
    kind = int(rng.integers(0, 9))
    s = int(rng.integers(0, 1 << 30))
    r = np.random.default_rng(s)
    if n == 0:
        return "empty", np.zeros(0, dtype=np.uint8)
    if kind == 0:
        return "zipf", oracle.gen_bytes(s, n, shafa.zipf_table(float(rng.uniform(0.7, 2.6))))
    if kind == 1:
        return "runs", synth.runs_stream(s, n, shafa.zipf_table(float(rng.uniform(0.9, 2.0))))
    if kind == 2:
        return "uniform", r.integers(0, 256, size=n, dtype=np.uint8)
    if kind == 3:                                  # zero-heavy: 0 is an escape, a symbol and a count in the RLE stream
        a = r.integers(0, 256, size=n, dtype=np.uint8)
        a[r.random(n) < float(rng.uniform(0.2, 0.95))] = 0
        return "zeros", a
    if kind == 4:                                  # long runs (past 255: segmentation), of zeros too
        out = np.empty(n, dtype=np.uint8)
        i = 0
        while i < n:
            ln = int(r.choice([1, 2, 3, 4, 5, 254, 255, 256, 257, 510, 511, 765, int(r.integers(1, 5000))]))
            out[i:i + ln] = int(r.choice([0, 0, 1, 255, int(r.integers(0, 256))]))
            i += ln
        return "longruns", out
    if kind == 5:                                  # small alphabet, skewed: short codes, many symbols per stream byte
        k = int(rng.integers(1, 7))
        p = r.random(k) ** 3 + 1e-3
        return f"alpha{k}", r.choice(r.integers(0, 256, size=k), size=n, p=p / p.sum()).astype(np.uint8)
    if kind == 6:
        return "text", (32 + (r.zipf(1.3, size=n) % 95)).astype(np.uint8)
    if kind == 7:                                  # one dominant symbol and a long tail: Lmax past 16
        a = (r.zipf(1.08, size=n) % 256).astype(np.uint8)
        a[r.random(n) < 0.5] = 7
        return "tail", a
    a = oracle.gen_bytes(s, n, shafa.zipf_table(1.2))           # zipf with a stretch of one byte and a stretch of noise
    lo = int(r.integers(0, n))
    a[lo:lo + int(r.integers(1, 100000))] = int(r.integers(0, 256))
    lo = int(r.integers(0, n))
    seg = a[lo:lo + int(r.integers(1, 50000))]
    seg[:] = r.integers(0, 256, size=seg.size, dtype=np.uint8)
    return "patch", a


def diff(a, b):
    if a.size != b.size:
        return f"sizes {a.size} / {b.size}"
    w = np.flatnonzero(a != b)
    return f"first difference at {int(w[0])} of {a.size}" if w.size else "equal"


def chain(tag, data):
    f = shafa.hist256(data)
    assert (f == oracle.hist256(data)).all(), f"{tag}: hist256"
    want_rle = oracle.rle_encode(data)
    got_rle, fr = shafa.rle_encode(data, want_freq=True)
    assert got_rle.tobytes() == want_rle.tobytes(), f"{tag}: rle_encode {diff(got_rle, want_rle)}"
    assert (fr == oracle.hist256(want_rle)).all(), f"{tag}: rle_encode histogram"
    if data.size:
        back = shafa.rle_decode(want_rle, cap=data.size + 16)
        assert back.tobytes() == data.tobytes(), f"{tag}: rle_decode {diff(back, data)}"
    for name, src, freq in (("plain", data, f), ("rle", want_rle, fr)):
        if src.size == 0:
            continue
        otab = oracle.sf_build(freq)
        t = shafa.sf_build_codes(freq)
        assert bytes(t) == bytes(table_of(otab)), f"{tag}/{name}: Module T"
        rc, enc = oracle.sf_encode(src, otab)
        assert rc == 0
        got = shafa.sf_encode(src, t)
        assert got.tobytes() == enc.tobytes(), f"{tag}/{name}: sf_encode {diff(got, enc)}"
        wrc, want = oracle.sf_decode(enc, otab, src.size)        # (a one-symbol block has a table without codes: rc 4, d.c:514-551)
        grc, dec = shafa.sf_decode(enc, t, src.size, raw_rc=True)
        assert grc == wrc, f"{tag}/{name}: sf_decode rc {grc}, oracle {wrc}"
        if wrc == 0:
            assert want.tobytes() == src.tobytes()
            assert dec.tobytes() == src.tobytes(), f"{tag}/{name}: sf_decode {diff(dec, src)}"
        if enc.size > 8 and int(rng.integers(0, 3)) == 0:      # the same stream damaged: the oracle's return code and bytes
            bad = enc.copy()
            i = int(rng.integers(0, bad.size))
            bad[i:i + int(rng.integers(1, 40))] ^= np.uint8(int(rng.integers(1, 256)))
            wrc, want = oracle.sf_decode(bad, otab, src.size)
            grc, gotd = shafa.sf_decode(bad, t, src.size, raw_rc=True)
            assert grc == wrc, f"{tag}/{name}: damaged stream rc {grc}, oracle {wrc}"
            if wrc == 0:
                assert gotd.tobytes() == want.tobytes(), f"{tag}/{name}: damaged stream {diff(gotd, want)}"
    if data.size and data.size < 400000:                       # any byte string is an RLE token stream (d.c:116-197)
        cap = int(rng.choice([data.size, 4 * data.size, 64 * data.size]))
        wrc, want = oracle.rle_decode(data, cap=cap)
        grc, gotd = shafa.rle_decode(data, cap=cap, raw_rc=True)
        assert grc == wrc, f"{tag}: rle_decode of raw bytes rc {grc}, oracle {wrc}"
        if wrc == 0:
            assert gotd.tobytes() == want.tobytes(), f"{tag}: rle_decode of raw bytes {diff(gotd, want)}"


def group_chain(tag, pipe, datas):
    """the same chain for several blocks per launch (shafa_pipe_submit_group): F, C, then D with both decoders fused"""
    k = len(datas)
    pipe.submit_group(0, shafa.OP_RLE_ENCODE, datas, flags=shafa.PIPE_INPUT_HIST)
    rc, brc, outs, res = pipe.wait_group(0, k)
    assert rc == 0 and not any(brc), f"{tag}: group rle_encode rc {rc} {list(brc)}"
    rle = []
    for j, d in enumerate(datas):
        want = oracle.rle_encode(d)
        assert outs[j] == want.tobytes(), f"{tag}: group rle_encode block {j}"
        assert list(res[j].freq) == list(oracle.hist256(want)) and list(res[j].freq_in) == list(oracle.hist256(d)), \
            f"{tag}: group histograms block {j}"
        rle.append(want)
    keep = [j for j in range(k) if np.count_nonzero(np.bincount(rle[j], minlength=256)) >= 2]     # a table with codes
    if not keep:
        return
    otabs = [oracle.sf_build(oracle.hist256(rle[j])) for j in keep]
    tabs = [table_of(t) for t in otabs]
    caps = [(rle[j].size * max(1, int(max(t.lens()))) + 7) // 8 + 16 for j, t in zip(keep, tabs)]
    pipe.submit_group(1, shafa.OP_SF_ENCODE, [rle[j] for j in keep], tables=tabs, out_caps=caps)
    rc, brc, outs, res = pipe.wait_group(1, len(keep))
    assert rc == 0 and not any(brc), f"{tag}: group sf_encode rc {rc} {list(brc)}"
    encs = []
    for x, j in enumerate(keep):
        orc, want = oracle.sf_encode(rle[j], otabs[x])
        assert orc == 0 and outs[x] == want.tobytes(), f"{tag}: group sf_encode block {j}"
        encs.append(want)
    for op, wants in ((shafa.OP_SF_DECODE, [rle[j] for j in keep]), (shafa.OP_SF_RLE_DECODE, [datas[j] for j in keep])):
        pipe.submit_group(0, op, encs, tables=tabs, n_symbols=[rle[j].size for j in keep])
        rc, brc, outs, res = pipe.wait_group(0, len(keep))
        assert rc == 0 and not any(brc), f"{tag}: group decode op {op} rc {rc} {list(brc)}"
        for x, j in enumerate(keep):
            assert outs[x] == wants[x].tobytes(), f"{tag}: group decode op {op} block {j}"


shafa.lib().shafa_hip_init(0)
gpipe = shafa.Pipe(2)
t0, rounds, nbytes, kinds = time.time(), 0, 0, {}
while time.time() - t0 < budget:
    if rounds % 8 == 7:                            # every eighth round: a launch of several blocks
        opts = ALTS[int(rng.integers(0, len(ALTS)))]
        datas = []
        for _ in range(int(rng.integers(2, 20))):
            n = min(max(1, size()), 1 << 20)
            datas.append(np.ascontiguousarray(make(n)[1], dtype=np.uint8))
        tag = f"seed0={seed0} round={rounds} group of {len(datas)} sizes {[d.size for d in datas]} {opts}"
        for k, v in opts.items():
            shafa.set_option(k, v)
        try:
            group_chain(tag, gpipe, datas)
        except Exception:
            print("FAILED:", tag, flush=True)
            raise
        finally:
            for k, v in DEFAULTS.items():
                shafa.set_option(k, v)
        rounds += 1
        nbytes += sum(d.size for d in datas)
        kinds["group"] = kinds.get("group", 0) + 1
        continue
    opts = ALTS[int(rng.integers(0, len(ALTS)))]
    n = size()
    kind, data = make(n)
    tag = f"seed0={seed0} round={rounds} {kind} n={n} {opts}"
    for k, v in opts.items():
        shafa.set_option(k, v)
    try:
        chain(tag, np.ascontiguousarray(data, dtype=np.uint8))
    except Exception:
        print("FAILED:", tag, flush=True)
        raise
    finally:
        for k, v in DEFAULTS.items():
            shafa.set_option(k, v)
    rounds += 1
    nbytes += n
    kinds[kind.rstrip("0123456789")] = kinds.get(kind.rstrip("0123456789"), 0) + 1
print(f"soak_all: seed {seed0}, {rounds} inputs, {nbytes / 2**20:.0f} MiB, {time.time() - t0:.0f} s: every stage equal to the oracle's;"
      f" inputs by kind {dict(sorted(kinds.items()))}")
