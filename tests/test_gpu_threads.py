"""Layer 1 under the reference's threading model (-m gpu): the reference calls compress_to_buffer /
process_shafa_decomp from one pthread per block at the same time (utils/multithread.c:70-87 with c.c:411, d.c:735), so the
functions that replace them must be callable from several host threads at once; and no entry point may leave the calling
thread's current device changed (a torch caller, or layer 1 next to a multi-device pipe)."""
import threading

import numpy as np
import pytest

from test_gpu_parity import first_diff, to_shafa_table

pytestmark = pytest.mark.gpu


def test_layer1_from_four_host_threads_at_once(oracle, shafa):
    shafa.lib().shafa_hip_init(0)
    zt = shafa.zipf_table(1.2)
    import golden.make_golden as mg
    errors = []
    barrier = threading.Barrier(4)

    def work(tidx):
        try:
            rng_sizes = [70001 + 977 * tidx, 300000 + 13 * tidx, 1 << 20]
            barrier.wait()
            for rep in range(3):
                for n in rng_sizes:
                    data = oracle.gen_bytes(1000 * tidx + n + rep, n, zt)
                    assert (shafa.hist256(data) == oracle.hist256(data)).all(), "hist256"
                    otab = oracle.sf_build(oracle.hist256(data))
                    t = to_shafa_table(shafa, otab)
                    rc, want = oracle.sf_encode(data, otab)
                    enc = shafa.sf_encode(data, t)
                    assert enc.tobytes() == want.tobytes(), f"sf_encode thread {tidx} n={n}: {first_diff(enc, want)}"
                    back = shafa.sf_decode(want, t, n)
                    assert back.tobytes() == data.tobytes(), f"sf_decode thread {tidx} n={n}"
                    runs = mg.runs_stream(77 * tidx + rep, n // 4, zt)
                    rle = shafa.rle_encode(runs)
                    assert rle.tobytes() == oracle.rle_encode(runs).tobytes(), f"rle_encode thread {tidx}"
                    assert shafa.rle_decode(rle).tobytes() == runs.tobytes(), f"rle_decode thread {tidx}"
        except BaseException as e:          # noqa: BLE001 - reported by the main thread
            errors.append(f"thread {tidx}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(600)
        assert not th.is_alive(), "a layer-1 call did not return"
    assert not errors, "\n".join(errors[:4])


def test_no_entry_point_changes_the_callers_current_device(oracle, shafa):
    """Interleaves pipe submit / wait on every selected device with layer-1 calls and torch work; the calling thread's
    current device (torch's and HIP's are the same thing) must be what the caller set.  With one GPU the device list
    names it twice, which still takes the per-slot guard path."""
    import torch
    n = torch.cuda.device_count()
    devs = list(range(n)) if n > 1 else [0, 0]
    cur = n - 1                                     # the caller works on the LAST device; layer 1 lives on devs[0]
    torch.cuda.set_device(cur)
    assert shafa.init_devices(devs) == len(devs)
    try:
        pipe = shafa.Pipe(2 * len(devs))
        zt = shafa.zipf_table(1.2)
        blocks = [oracle.gen_bytes(31 + i, 150000 + 11 * i, zt) for i in range(pipe.n_slots)]
        x = torch.arange(1000, device=f"cuda:{cur}")
        for i, b in enumerate(blocks):
            pipe.submit(i, shafa.OP_RLE_ENCODE, b)
            assert torch.cuda.current_device() == cur, "pipe.submit changed the current device"
            enc = shafa.rle_encode(b[:5000])                       # layer 1 (device devs[0]) in between
            assert enc.tobytes() == oracle.rle_encode(b[:5000]).tobytes()
            assert torch.cuda.current_device() == cur, "layer 1 changed the current device"
            assert int((x + i).sum().item()) == 499500 + 1000 * i  # torch still works where it was
        for i, b in enumerate(blocks):
            rc, out, r = pipe.wait(i)
            assert out == oracle.rle_encode(b).tobytes()
            assert torch.cuda.current_device() == cur, "pipe.wait changed the current device"
        pipe.close()
        assert torch.cuda.current_device() == cur, "pipe destroy changed the current device"
    finally:
        shafa.init_devices([0])
        torch.cuda.set_device(0)


def test_two_processes_sharing_the_gpu_finish_without_device_errors():
    """Chained scans wait for predecessor tiles that belong to other workgroups; with a second process on the same GPU
    (or a profiler) those may be late by milliseconds.  The wait is bounded by wall-clock seconds, not by a poll count,
    so contention shows up as a slower run, never as SHAFA_DEVICE_ERROR."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SHAFA_BENCH_ORACLE_CHECK"] = "0"
    argv = [sys.executable, os.path.join(root, "bench.py"), "--blocks", "24", "--steps", "6", "--warmup", "1", "--no-cpu",
            "--pipeline-blocks", "8"]
    procs = [subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for _ in range(3)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        line = [ln for ln in out.splitlines() if ln.startswith("{")]
        assert len(line) == 1 and json.loads(line[0])["value"] > 0
