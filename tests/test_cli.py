"""The C host CLI (shafa-cd_amd/bin/shafa) against the reference CLI's recorded behaviour
(tests/golden/*/manifest.json: argv, exit code, stderr, every produced file's size + SHA-256).

CPU part: argv errors and the loud failure without a GPU.  GPU part (-m gpu): every golden case is
replayed with our binary and every file it writes must be byte-identical to the reference's."""
import hashlib
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.environ.get("SHAFA_CLI") or os.path.join(ROOT, "shafa-cd_amd", "bin", "shafa")    # (SHAFA_CLI: tools/san/run_san.sh)


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def run(argv, cwd):
    r = subprocess.run([CLI] + argv, cwd=cwd, capture_output=True, timeout=600)
    return r.returncode, r.stderr.decode("utf-8", "replace"), r.stdout.decode("utf-8", "replace")


def manifest(case):
    with open(os.path.join(GOLD, case, "manifest.json")) as f:
        return json.load(f)


def test_cli_binary_is_built():
    assert os.path.exists(CLI), "run __graft_entry__.build()"


def test_cli_argv_errors_match_reference(tmp_path):
    man = manifest("cli_errors")
    shutil.copyfile(os.path.join(GOLD, "cli_errors", "z"), tmp_path / "z")
    checked = 0
    for cmd in man["cmds"]:
        argv = cmd["argv"]
        if argv[:3] == ["z", "-m", "f"]:
            continue                       # runs Module F: GPU part below
        rc, err, _ = run(argv, tmp_path)
        assert (rc, err) == (cmd["rc"], cmd["stderr"]), argv
        checked += 1
    assert checked == 4
    rc, err, _ = run([], tmp_path)
    assert rc == 1 and err == "No file input\n"
    rc, err, _ = run(["-m", "f"], tmp_path)
    assert rc == 1 and err == "No file input\n"
    rc, err, _ = run(["z", "-b", "X"], tmp_path)
    assert rc == 1 and err == "Wrong Options' syntax\n"
    rc, err, _ = run(["z.freq", "-m", "t"], tmp_path)          # .freq missing: file can't be accessed
    assert rc == 1 and err.startswith("Module t: Something went wrong...\nFile can't be accessed")
    rc, err, _ = run(["z", "-m", "t"], tmp_path)
    assert rc == 1 and err == "Module t: Wrong extension... Should end in .freq\n"


def test_cli_fails_loudly_without_gpu(tmp_path, shafa):
    if shafa.lib().shafa_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    shutil.copyfile(os.path.join(GOLD, "cli_errors", "z"), tmp_path / "z")
    rc, err, _ = run(["z"], tmp_path)
    assert rc == 1 and "GPU device error" in err
    assert not os.path.exists(tmp_path / "z.rle") and not os.path.exists(tmp_path / "z.freq")


def test_cli_module_t_alone_on_cpu(tmp_path):
    """Module T is host-only (256 symbols per block): it runs without a GPU and must reproduce h.cod."""
    shutil.copyfile(os.path.join(GOLD, "t_handmade", "h.freq"), tmp_path / "h.freq")
    rc, err, out = run(["h.freq", "-m", "t"], tmp_path)
    assert rc == 0, err
    assert sha(tmp_path / "h.cod") == manifest("t_handmade")["files"]["h.cod"]["sha256"]
    assert "Module:T (Calculation of symbol codes)" in out and "Number of blocks: 5" in out


GPU_CASES = ["runs_default", "edges_forced_rle", "uniform_no_rle", "uniform_forced_both", "runs_force_freq",
             "textlike_m", "tiny_1024", "tiny_1023", "cli_errors", "cfg0_K_runs", "cfg0_K_uniform",
             # block-split edges (file.c:78-85, f.c:231-236) and an error in the middle of the ordered write chain (c.c:254-267)
             "edge_exact_K", "edge_tail_1", "edge_tail_7", "edge_tail_15", "edge_bad_cod_mid"]
# full-size blocks (8 MiB / 64 MiB; inputs rebuilt from the manifest's generators, outputs pinned by SHA-256)
FULL_CASES = ["full_uniform_m", "full_zipf_M", "full_zipfmod_M_forced_rle", "full_single_run_M", "full_alt01_M",
              "full_longtail_M", "full_mixed_M", "full_skewed_blocks_m",
              # hundreds of default-size blocks: the drivers' group mode (several groups per slot)
              "many_default_rle", "many_default_plain", "many_default_bad_cod", "many_default_single_run"]


def scratch_dir(tmp_path, case):
    """Full-size cases write up to ~0.7 GiB: keep them in tmpfs when there is one."""
    if case.startswith("full_") and os.path.isdir("/dev/shm"):
        import tempfile
        return tempfile.mkdtemp(prefix="shafa_" + case + "_", dir="/dev/shm")
    return str(tmp_path)


# Where our host deliberately does NOT do what the reference does: the files on disk are
# still compared with the reference's byte for byte.
DIVERGENCES = {
    # a last block of ONE byte is a single-symbol block: all codes empty, `@0@` in the .shaf, and the reference's decoder
    # dereferences NULL (d.c:533, rc -11 here).  Ours: _FILE_UNRECOGNIZABLE for that block, the blocks before it written.
    ("edge_tail_1", ("p.rle.shaf",)): {"rc": 1, "stderr_has": "Module d"},
    # the reference drops the thread chain's error in Module C (c.c:421 ignores multithread_wait()'s result; with
    # --no-multithread the same file fails with rc 1): it prints a summary and exits 0 with a truncated .shaf.  Ours
    # reports the malformed block; what was written before it is the same.
    ("edge_bad_cod_mid", ("g.rle", "-m", "c")): {"rc": 1, "stderr_has": "Module c"},
    ("many_default_bad_cod", ("h.rle", "-m", "c")): {"rc": 1, "stderr_has": "Module c"},       # the same, deep inside a group
}


def replay(case, work):
    """Replay the recorded reference session with our CLI: rc, stderr and the masked stdout summary of every
    command, then every produced file (size + SHA-256)."""
    import golden.make_golden as mg
    man = manifest(case)
    for fn in man["inputs"]:
        src = os.path.join(GOLD, case, fn)
        dst = os.path.join(work, fn)
        if os.path.exists(src):
            shutil.copyfile(src, dst)
        else:                                           # big inputs are regenerated from their generator
            if "generators" in man:
                data = mg.make_input(man["generators"][fn])
            else:
                zt = mg.zipf_table(1.2)
                data = mg.runs_stream(7, 655360, zt) if case == "cfg0_K_runs" else mg.gen_bytes(8, 655360)
            data.tofile(dst)
            assert sha(dst) == man["files"][fn]["sha256"], f"{case}: regenerated input {fn} differs"
    for cmd in man["cmds"]:
        if isinstance(cmd, list):
            if cmd[0] == "__copy__":
                shutil.copyfile(os.path.join(work, cmd[1]), os.path.join(work, cmd[2]))
            elif cmd[0] == "__rm__":
                os.remove(os.path.join(work, cmd[1]))
            elif cmd[0] == "__corrupt_cod__":
                mg.corrupt_cod_block(os.path.join(work, cmd[1]), cmd[2])
            continue
        rc, err, out = run(cmd["argv"], work)
        div = DIVERGENCES.get((case, tuple(cmd["argv"])))
        if div:
            assert rc == div["rc"] and div["stderr_has"] in err, f"{case} {cmd['argv']}: rc {rc} stderr {err!r}"
            continue
        assert rc == cmd["rc"], f"{case} {cmd['argv']}: rc {rc} stderr {err!r}"
        assert err == cmd["stderr"], f"{case} {cmd['argv']}"
        # stdout summaries (f.c:132-177, t.c:219-243, c.c:282-303, d.c:44-65): identical except the measured
        # runtime; the reference's two author-name lines per module are not printed by our host
        assert not mg.BANNER_RE.search(out), f"{case} {cmd['argv']}: unexpected author banner"
        assert mg.mask_stdout(out) == cmd["stdout"], f"{case} {cmd['argv']}: stdout summary differs"
    return man


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES + FULL_CASES)
def test_cli_replays_reference_session(case, tmp_path):
    work = scratch_dir(tmp_path, case)
    try:
        _replay_and_check(case, work)
    finally:
        if work != str(tmp_path):
            shutil.rmtree(work, ignore_errors=True)


def _replay_and_check(case, tmp_path):
    from pathlib import Path
    tmp_path = Path(tmp_path)
    man = replay(case, str(tmp_path))
    produced = sorted(os.listdir(tmp_path))
    assert produced == sorted(man["files"]), f"{case}: file set differs"
    for fn, meta in man["files"].items():
        p = tmp_path / fn
        assert os.path.getsize(p) == meta["size"], f"{case}/{fn}: size {os.path.getsize(p)} != {meta['size']}"
        assert sha(p) == meta["sha256"], f"{case}/{fn}: content differs from the reference's file"


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["many_default_rle", "full_skewed_blocks_m", "full_zipfmod_M_forced_rle"])
def test_cli_with_a_full_nodes_slot_count(case, tmp_path, monkeypatch):
    """SHAFA_DEVICES=0,0,0,0,0,0,0,0: the C host spreads its pipe slots over the listed devices, three per device — 24 slots,
    eight H2D streams' worth of bookkeeping, the ordered retire across all of them — as on a full 8-GPU node, here on the one
    GPU there is.  Same files, byte for byte, as the reference (groups of small blocks, a ragged -b m file, 64 MiB blocks)."""
    monkeypatch.setenv("SHAFA_DEVICES", "0,0,0,0,0,0,0,0")
    work = scratch_dir(tmp_path, case)
    try:
        _replay_and_check(case, work)
    finally:
        if work != str(tmp_path):
            shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["2", "0"])
@pytest.mark.parametrize("case", ["full_zipf_M", "full_skewed_blocks_m", "full_uniform_m", "full_single_run_M"])
def test_cli_default_run_one_upload_per_block(case, mode, tmp_path, monkeypatch):
    """`shafa file -b m|M` runs f, t and c in one process (shafa.c:293-298): host/modules.c shafa_ftc_compress does them with ONE
    upload per block (layer 3 SHAFA_OP_FTC: F on the device, T on the host, C from the bytes still on the device).
    SHAFA_FTC=2: that driver must have done the run; SHAFA_FTC=0: the three separate passes.  Either way every file and
    summary is the reference's (RLE accepted and rejected by block 0, a ragged last block, a single-run block)."""
    monkeypatch.setenv("SHAFA_FTC", mode)
    work = scratch_dir(tmp_path, case)
    try:
        _replay_and_check(case, work)
    finally:
        if work != str(tmp_path):
            shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case,stem", [("runs_default", "x.rle"), ("full_uniform_m", "u")])
def test_cli_module_c_reads_a_fifo(case, stem, tmp_path):
    """The reference's Module C reads its input with fread, front to back (c.c:392): a FIFO works.  Ours reads blocks with
    pread; an input that cannot seek is read in order instead.  Same .shaf as from the regular file (small blocks in groups,
    and 8 MiB blocks one per slot)."""
    import threading
    import golden.make_golden as mg
    man = manifest(case)
    work = scratch_dir(tmp_path, case)
    try:
        src = os.path.join(work, "regular")
        if os.path.exists(os.path.join(GOLD, case, stem)):
            shutil.copyfile(os.path.join(GOLD, case, stem), src)
        else:                                           # (a -b m case: Module F's file is the input itself, RLE declined)
            mg.make_input(man["generators"][stem]).tofile(src)
        assert sha(src) == man["files"][stem]["sha256"]
        shutil.copyfile(os.path.join(GOLD, case, stem + ".cod"), os.path.join(work, stem + ".cod"))
        fifo = os.path.join(work, stem)
        os.mkfifo(fifo)

        def feed():
            with open(src, "rb") as f, open(fifo, "wb") as w:
                shutil.copyfileobj(f, w, 1 << 20)
        th = threading.Thread(target=feed)
        th.start()
        rc, err, _ = run([stem, "-m", "c"], work)
        th.join()
        assert rc == 0, err
        assert sha(os.path.join(work, stem + ".shaf")) == man["files"][stem + ".shaf"]["sha256"]
    finally:
        if work != str(tmp_path):
            shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("fifo_name,side,argv,decoded", [
    ("x.rle.shaf", ["x.rle.cod"], ["x.rle.shaf"], "decoded__sf_rle"),                 # d.c:673,697-706: fscanf / fread of the .shaf
    ("x.rle", ["x.rle.freq"], ["x.rle", "-m", "d"], "decoded__rle_only"),             # d.c:75-95 load_rle
])
def test_cli_module_d_reads_a_fifo(fifo_name, side, argv, decoded, tmp_path):
    """Module D's inputs through a FIFO (the reference reads them front to back with fscanf / fread): the block headers of
    the .shaf are then parsed byte by byte, the payloads read in order.  Same decoded file as from the regular file."""
    import threading
    man = manifest("runs_default")
    work = str(tmp_path)
    src = os.path.join(work, "regular")
    shutil.copyfile(os.path.join(GOLD, "runs_default", fifo_name), src)
    for f in side:
        shutil.copyfile(os.path.join(GOLD, "runs_default", f), os.path.join(work, f))
    fifo = os.path.join(work, fifo_name)
    os.mkfifo(fifo)

    def feed():
        with open(src, "rb") as f, open(fifo, "wb") as w:
            shutil.copyfileobj(f, w, 1 << 16)
    th = threading.Thread(target=feed)
    th.start()
    rc, err, _ = run(argv, work)
    th.join()
    assert rc == 0, err
    assert sha(os.path.join(work, "x")) == man["files"][decoded]["sha256"]


@pytest.mark.gpu
def test_cli_module_d_fifo_with_groups_closing_on_the_budget(tmp_path):
    """701 RLE blocks of 64 KiB through a FIFO: Module D works in groups of up to 256 blocks, has to stage the payloads it
    reads past, and a group closes on the pipe's byte budget with a staged block in hand (run_groups' `held`), which then
    opens the next group: the payload must be owned by exactly one record (it was freed twice)."""
    import threading
    case = "many_default_rle"
    work = scratch_dir(tmp_path, case + "_fifo")
    try:
        man = replay(case, work)
        src = os.path.join(work, "regular")
        os.rename(os.path.join(work, "m.rle.shaf"), src)
        for f in ("m", "m.rle"):
            if os.path.exists(os.path.join(work, f)):
                os.remove(os.path.join(work, f))
        fifo = os.path.join(work, "m.rle.shaf")
        os.mkfifo(fifo)

        def feed():
            with open(src, "rb") as f, open(fifo, "wb") as w:
                shutil.copyfileobj(f, w, 1 << 16)
        th = threading.Thread(target=feed)
        th.start()
        rc, err, _ = run(["m.rle.shaf"], work)
        th.join()
        assert rc == 0, err
        assert sha(os.path.join(work, "m")) == man["files"]["decoded__sf_rle"]["sha256"]
    finally:
        if work != str(tmp_path):
            shutil.rmtree(work, ignore_errors=True)


@pytest.mark.gpu
def test_cli_device_list_errors_are_reported(tmp_path):
    """SHAFA_DEVICES naming a GPU the node does not have (or garbage) is an error message and exit 1 before any module
    runs — not a silent fall-back to device 0; a valid list works; Module T alone ignores the variable (it touches no GPU)."""
    shutil.copyfile(os.path.join(GOLD, "cli_errors", "z"), tmp_path / "z")
    for bad in ("99", "0,99", "x", "-1", "0;1"):
        env = dict(os.environ, SHAFA_DEVICES=bad)
        r = subprocess.run([CLI, "z", "-m", "f"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 1 and "SHAFA_DEVICES" in r.stderr, (bad, r.returncode, r.stderr)
        assert not os.path.exists(tmp_path / "z.rle") and not os.path.exists(tmp_path / "z.freq"), bad
    env = dict(os.environ, SHAFA_DEVICES="0")
    r = subprocess.run([CLI, "z", "-m", "f"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    stem = "z.rle" if os.path.exists(tmp_path / "z.rle") else "z"
    env = dict(os.environ, SHAFA_DEVICES="99")
    r = subprocess.run([CLI, stem + ".freq", "-m", "t"], cwd=tmp_path, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and os.path.exists(tmp_path / (stem + ".cod")), r.stderr
