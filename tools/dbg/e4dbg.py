import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pkgload, oracle_lib
import test_gpu_encode_onepass as T
pkg = pkgload.load(); orc = oracle_lib.load()
pkg.lib().shafa_hip_init(0)
pkg.set_option("sf_encode_one_pass_min_blocks", 1)
zt = pkg.zipf_table(1.2)
def trial(name, blocks, tables):
    try:
        T.run_batch(pkg, orc, blocks, tables)
        print(name, "OK")
    except AssertionError as e:
        print(name, "FAIL", str(e)[:300])
for nb in (1, 2, 4, 8, 40):
    for kind in ("zipf", "uniform"):
        blocks = [orc.gen_bytes(900 + i, 70000 + 100 * i, zt if kind == "zipf" else None) for i in range(nb)]
        tables = [orc.sf_build(orc.hist256(b)) for b in blocks]
        trial(f"{kind} nb={nb} lmax={max(t.lens().max() for t in tables)}", blocks, tables)
# the failing case: tables from another histogram
blocks = [orc.gen_bytes(900 + i, n) for i, n in enumerate([70000, 8192, 500000, 33])]
tables = [orc.sf_build(orc.hist256(orc.gen_bytes(900 + i, 1 << 20))) for i in range(4)]
trial("uniform foreign tables", blocks, tables)
tables = [orc.sf_build(orc.hist256(b)) for b in blocks]
trial("uniform own tables", blocks, tables)
print("---- narrowing")
def ft(i): return orc.sf_build(orc.hist256(orc.gen_bytes(900 + i, 1 << 20)))
for sizes in ([70000], [33], [8192], [70000, 8192], [70000, 8192, 500000], [70000, 8192, 500000, 33], [33, 33, 33, 33], [70000]*4):
    blocks = [orc.gen_bytes(900 + i, n) for i, n in enumerate(sizes)]
    trial(f"foreign {sizes}", blocks, [ft(i) for i in range(len(sizes))])
    trial(f"foreign-same-table {sizes}", blocks, [ft(0) for i in range(len(sizes))])
