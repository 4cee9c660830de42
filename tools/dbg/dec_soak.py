"""Soak of the Shannon-Fano decoder against the oracle: random block sizes, skews, alphabets and output alignments for a given
number of seconds (default 120), speculation modes 0 / 1 / 2.   usage (through gpurun): python tools/dbg/dec_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import pkgload
import oracle_lib
import test_gpu_decode_spec as T

shafa = pkgload.load()
oracle = oracle_lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
t0, rounds, nblocks = time.time(), 0, 0
while time.time() - t0 < budget:
    blocks = []
    for _ in range(int(rng.integers(1, 7))):
        n = int(rng.choice([int(rng.integers(1, 9000)), int(rng.integers(9000, 300000)), int(rng.integers(300000, 3 << 20))]))
        kind = int(rng.integers(0, 4))
        seed = int(rng.integers(0, 1 << 30))
        if kind == 0:
            b = T.skewed(seed, n, float(rng.uniform(0.3, 0.995)), nsym=int(rng.integers(2, 40)))
        elif kind == 1:
            b = T.zipfmod(oracle, seed, n)
        elif kind == 2:
            b = oracle.gen_bytes(seed, n, shafa.zipf_table(float(rng.uniform(0.8, 2.5))))
        else:
            b = T.zipfmod(oracle, seed, n)
            lo = int(rng.integers(0, n)); hi = min(n, lo + int(rng.integers(1, 200000)))
            b[lo:hi] = np.bincount(b, minlength=256).argmax()
        blocks.append(b)
    T.check(shafa, oracle, blocks, [0, 1, 2], out_shift=int(rng.integers(0, 4)) * 16)
    rounds += 1
    nblocks += len(blocks)
print(f"dec_soak: {rounds} launches x 3 modes, {nblocks} blocks, {time.time() - t0:.0f} s: all bit-exact")
