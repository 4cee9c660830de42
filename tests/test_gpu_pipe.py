"""GPU parity tests, part 3: the layer-3 block pipeline (shafa_pipe_*).  Every op
of the pipeline is checked bit-exactly against the oracle while several blocks are in flight, and
results must come back in submission order whatever the per-block sizes."""
import numpy as np
import pytest

from test_gpu_parity import streams, to_shafa_table

pytestmark = pytest.mark.gpu


def blocks_of(oracle, shafa, count, base):
    import golden.make_golden as mg
    zt = shafa.zipf_table(1.2)
    s = streams(oracle, shafa)
    out = []
    for i in range(count):
        n = base + 7919 * i * (i % 3)                       # different sizes so slots finish out of step
        out.append(mg.runs_stream(100 + i, n, zt) if i % 2 else s["zipf"](n))
    return out


def test_pipe_every_op_matches_oracle_with_blocks_in_flight(oracle, shafa):
    blocks = blocks_of(oracle, shafa, 7, 300000)
    pipe = shafa.Pipe(3)
    assert pipe.n_slots == 3
    # encode side: RLE (+ both histograms), then SF encode of the RLE bytes
    rle, tabs, enc = [], [], []

    def run(n_items, submit, retire):
        sub = ret = 0
        while ret < n_items:
            if sub < n_items and sub - ret < pipe.n_slots:
                submit(sub, sub % pipe.n_slots)
                sub += 1
            else:
                retire(ret, ret % pipe.n_slots)
                ret += 1

    def ret_rle(i, slot):
        rc, out, r = pipe.wait(slot)
        want = oracle.rle_encode(blocks[i])
        assert out == want.tobytes(), f"block {i}: rle bytes differ"
        assert list(r.freq) == list(oracle.hist256(want)), f"block {i}: rle histogram"
        assert list(r.freq_in) == list(oracle.hist256(blocks[i])), f"block {i}: input histogram"
        rle.append(np.frombuffer(out, dtype=np.uint8))

    run(len(blocks), lambda i, slot: pipe.submit(slot, shafa.OP_RLE_ENCODE, blocks[i], flags=shafa.PIPE_INPUT_HIST), ret_rle)

    def ret_hist(i, slot):
        rc, out, r = pipe.wait(slot)
        assert list(r.freq) == list(oracle.hist256(blocks[i])), f"block {i}: hist"

    run(len(blocks), lambda i, slot: pipe.submit(slot, shafa.OP_HIST, blocks[i]), ret_hist)

    for b in rle:
        tabs.append(oracle.sf_build(oracle.hist256(b)))

    def sub_enc(i, slot):
        lmax = int(max(to_shafa_table(shafa, tabs[i]).lens()))
        pipe.submit(slot, shafa.OP_SF_ENCODE, rle[i], table=to_shafa_table(shafa, tabs[i]), out_cap=(len(rle[i]) * lmax + 7) // 8 + 16)

    def ret_enc(i, slot):
        rc, out, r = pipe.wait(slot)
        orc, want = oracle.sf_encode(rle[i], tabs[i])
        assert orc == 0 and out == want.tobytes(), f"block {i}: sf payload differs"
        enc.append(np.frombuffer(out, dtype=np.uint8))

    run(len(blocks), sub_enc, ret_enc)

    # decode side: SF only, RLE only, and the fused SF+RLE op
    def ret_sfdec(i, slot):
        rc, out, r = pipe.wait(slot)
        assert out == rle[i].tobytes(), f"block {i}: sf decode"

    run(len(blocks), lambda i, slot: pipe.submit(slot, shafa.OP_SF_DECODE, enc[i], table=to_shafa_table(shafa, tabs[i]),
                                                 n_symbols=len(rle[i])), ret_sfdec)

    def ret_rledec(i, slot):
        rc, out, r = pipe.wait(slot)
        assert out == blocks[i].tobytes(), f"block {i}: rle decode"

    run(len(blocks), lambda i, slot: pipe.submit(slot, shafa.OP_RLE_DECODE, rle[i]), ret_rledec)

    def ret_fused(i, slot):
        rc, out, r = pipe.wait(slot)
        assert r.mid_n == len(rle[i])
        assert out == blocks[i].tobytes(), f"block {i}: fused sf+rle decode"

    run(len(blocks), lambda i, slot: pipe.submit(slot, shafa.OP_SF_RLE_DECODE, enc[i], table=to_shafa_table(shafa, tabs[i]),
                                                 n_symbols=len(rle[i])), ret_fused)
    pipe.close()


def test_pipe_ftc_one_residency_matches_the_three_modules(oracle, shafa):
    """SHAFA_OP_FTC: Module F on the device (RLE bytes + both histograms, or the plain histogram), Module T on the host from
    the histogram that came back, Module C from the bytes still on the device (shafa_pipe_ftc_encode) — three blocks in
    flight, each stage of every block against the oracle's block_compression / make_freq / code table / binary_coding."""
    blocks = blocks_of(oracle, shafa, 6, 2500000)
    pipe = shafa.Pipe(3)
    modes = [shafa.PIPE_FTC_RLE | shafa.PIPE_FTC_PLAIN | shafa.PIPE_INPUT_HIST, shafa.PIPE_FTC_RLE, shafa.PIPE_FTC_PLAIN,
             shafa.PIPE_FTC_RLE | shafa.PIPE_INPUT_HIST, shafa.PIPE_FTC_PLAIN, shafa.PIPE_FTC_RLE | shafa.PIPE_FTC_PLAIN]
    use_rle = [True, True, False, True, False, False]
    for i, b in enumerate(blocks):
        slot = i % pipe.n_slots
        pipe.submit(slot, shafa.OP_FTC, b, flags=modes[i])
        if i >= 2:                                          # keep three slots busy: retire the block two behind, both stages
            _ftc_finish(oracle, shafa, pipe, blocks, modes, use_rle, i - 2)
    for i in (len(blocks) - 2, len(blocks) - 1):
        _ftc_finish(oracle, shafa, pipe, blocks, modes, use_rle, i)
    pipe.close()


def _ftc_finish(oracle, shafa, pipe, blocks, modes, use_rle, i):
    slot = i % pipe.n_slots
    b = blocks[i]
    rc, out, r = pipe.wait(slot)
    want_rle = oracle.rle_encode(b)
    if modes[i] & shafa.PIPE_FTC_RLE:
        assert out == want_rle.tobytes(), f"block {i}: RLE bytes differ"
        assert list(r.freq) == list(oracle.hist256(want_rle)), f"block {i}: histogram of the RLE bytes"
        if modes[i] & (shafa.PIPE_FTC_PLAIN | shafa.PIPE_INPUT_HIST):
            assert list(r.freq_in) == list(oracle.hist256(b)), f"block {i}: histogram of the input"
    else:
        assert out == b"" and list(r.freq) == list(oracle.hist256(b)), f"block {i}: plain histogram"
    src = want_rle if use_rle[i] else b
    tab = oracle.sf_build(oracle.hist256(src))
    stab = to_shafa_table(shafa, tab)
    lmax = int(max(stab.lens()))
    pipe.ftc_encode(slot, use_rle[i], stab, (src.size * lmax + 7) // 8 + 16)
    rc, enc, r2 = pipe.wait(slot)
    assert enc == oracle.sf_encode(src, tab)[1].tobytes(), f"block {i}: .shaf payload differs (use_rle {use_rle[i]})"


def test_pipe_errors_surface_at_wait_in_block_order(oracle, shafa):
    s = streams(oracle, shafa)
    good = s["zipf"](50000)
    tab = oracle.sf_build(oracle.hist256(good))
    _, enc = oracle.sf_encode(good, tab)
    pipe = shafa.Pipe(2)
    # slot 0: fine; slot 1: truncated stream -> FILE_UNRECOGNIZABLE at wait(1), not at submit
    pipe.submit(0, shafa.OP_SF_DECODE, enc, table=to_shafa_table(shafa, tab), n_symbols=len(good))
    pipe.submit(1, shafa.OP_SF_DECODE, enc[: len(enc) // 2], table=to_shafa_table(shafa, tab), n_symbols=len(good))
    rc0, out0, _ = pipe.wait(0, raw_rc=True)
    rc1, out1, _ = pipe.wait(1, raw_rc=True)
    assert rc0 == 0 and out0 == good.tobytes()
    assert rc1 == shafa.FILE_UNRECOGNIZABLE
    # a busy slot refuses a second submit; an idle slot refuses wait
    pipe.submit(0, shafa.OP_HIST, good)
    with pytest.raises(shafa.ShafaError):
        pipe.submit(0, shafa.OP_HIST, good)
    pipe.wait(0)
    rc, _, _ = pipe.wait(0, raw_rc=True)
    assert rc == shafa.OUTSIDE_MODULE
    pipe.close()


def test_pipe_slots_spread_over_the_selected_devices(oracle, shafa):
    """shafa_hip_init_devices: slot i of a pipe lives on devices[i % n] (SURVEY.md §8(b) "init(device list)").  On a
    one-GPU box the list names device 0 twice, which still takes the per-slot device path; results stay bit-exact and
    in submission order."""
    import torch
    n = torch.cuda.device_count()
    devs = list(range(n)) if n > 1 else [0, 0]
    assert shafa.init_devices(devs) == len(devs)
    try:
        pipe = shafa.Pipe(3 * len(devs))
        assert pipe.devices == [devs[i % len(devs)] for i in range(pipe.n_slots)]
        blocks = blocks_of(oracle, shafa, 2 * pipe.n_slots + 1, 200000)
        sub = ret = 0
        while ret < len(blocks):
            if sub < len(blocks) and sub - ret < pipe.n_slots:
                pipe.submit(sub % pipe.n_slots, shafa.OP_RLE_ENCODE, blocks[sub])
                sub += 1
            else:
                rc, out, r = pipe.wait(ret % pipe.n_slots)
                want = oracle.rle_encode(blocks[ret])
                assert out == want.tobytes() and list(r.freq) == list(oracle.hist256(want)), f"block {ret}"
                ret += 1
        pipe.close()
    finally:
        shafa.init_devices([0])
    assert shafa.lib().shafa_hip_init_devices((__import__("ctypes").c_int * 1)(99), 1) == shafa.OUTSIDE_MODULE


def test_pipe_groups_match_oracle_for_every_op(oracle, shafa):
    """shafa_pipe_submit_group / shafa_pipe_wait_group: many small blocks per slot, groups in several slots at once, every op
    against the oracle block by block — including a block whose table is malformed and a block whose stream is cut, which
    must fail alone, in place, with the blocks around them intact (the drivers stop at the first failed block in order)."""
    import golden.make_golden as mg
    zt = shafa.zipf_table(1.2)
    rng = np.random.default_rng(11)
    n_groups, per = 5, 37
    sizes = [int(x) for x in rng.integers(3000, 90000, n_groups * per)]
    sizes[3], sizes[40] = 5, 17                                   # tiny blocks inside a group
    blocks = [mg.runs_stream(500 + i % 7, 90000, zt)[:sizes[i]] if i % 3 else
              np.frombuffer(oracle.gen_bytes(900 + i, sizes[i], zt).tobytes(), dtype=np.uint8) for i in range(len(sizes))]
    pipe = shafa.Pipe(3)
    groups = [list(range(g * per, (g + 1) * per)) for g in range(n_groups)]

    def run(submit, retire):
        sub = ret = 0
        while ret < n_groups:
            if sub < n_groups and sub - ret < pipe.n_slots:
                submit(groups[sub], sub % pipe.n_slots)
                sub += 1
            else:
                retire(groups[ret], ret % pipe.n_slots)
                ret += 1

    # F: RLE bytes + both histograms; plain histogram
    rle = {}

    def ret_rle(ids, slot):
        rc, brc, outs, res = pipe.wait_group(slot, len(ids))
        assert rc == 0 and not any(brc), (rc, brc)
        for j, i in enumerate(ids):
            want = oracle.rle_encode(blocks[i])
            assert outs[j] == want.tobytes(), f"block {i}: rle bytes differ"
            assert list(res[j].freq) == list(oracle.hist256(want)), f"block {i}: rle histogram"
            assert list(res[j].freq_in) == list(oracle.hist256(blocks[i])), f"block {i}: input histogram"
            rle[i] = np.frombuffer(outs[j], dtype=np.uint8)

    run(lambda ids, slot: pipe.submit_group(slot, shafa.OP_RLE_ENCODE, [blocks[i] for i in ids], flags=shafa.PIPE_INPUT_HIST), ret_rle)

    def ret_hist(ids, slot):
        rc, brc, outs, res = pipe.wait_group(slot, len(ids))
        assert rc == 0 and not any(brc)
        for j, i in enumerate(ids):
            assert list(res[j].freq) == list(oracle.hist256(blocks[i])), f"block {i}: hist"

    run(lambda ids, slot: pipe.submit_group(slot, shafa.OP_HIST, [blocks[i] for i in ids]), ret_hist)

    # C: SF encode of the RLE bytes; one block of the third group gets a table without a code for a byte it contains
    tabs = {i: to_shafa_table(shafa, oracle.sf_build(oracle.hist256(rle[i]))) for i in rle}
    bad_i = groups[2][5]
    bad_tab = to_shafa_table(shafa, oracle.sf_build(oracle.hist256(rle[bad_i])))
    sym = int(rle[bad_i][len(rle[bad_i]) // 2])
    bad_tab.len[sym] = 0
    enc = {}

    def sub_enc(ids, slot):
        ts = [bad_tab if i == bad_i else tabs[i] for i in ids]
        caps = [(rle[i].size * max(1, int(max(t.lens()))) + 7) // 8 + 16 for i, t in zip(ids, ts)]
        pipe.submit_group(slot, shafa.OP_SF_ENCODE, [rle[i] for i in ids], tables=ts, out_caps=caps)

    def ret_enc(ids, slot):
        rc, brc, outs, res = pipe.wait_group(slot, len(ids))
        assert rc == 0
        for j, i in enumerate(ids):
            if i == bad_i:
                assert brc[j] == shafa.FILE_UNRECOGNIZABLE, f"block {i}: rc {brc[j]}"
                continue
            assert brc[j] == 0, f"block {i}: rc {brc[j]}"
            orc, want = oracle.sf_encode(rle[i], oracle.sf_build(oracle.hist256(rle[i])))
            assert orc == 0 and outs[j] == want.tobytes(), f"block {i}: SF bytes differ"
            enc[i] = np.frombuffer(outs[j], dtype=np.uint8)

    run(sub_enc, ret_enc)

    # D: SF decode, RLE decode, and both fused; one stream of the second group is cut short
    enc[bad_i] = enc[groups[2][4]]                               # something decodable in the failed block's place
    tabs_d = dict(tabs)
    tabs_d[bad_i] = tabs[groups[2][4]]
    nsym = {i: rle[i].size for i in rle}
    nsym[bad_i] = rle[groups[2][4]].size
    cut_i = groups[1][9]
    huge_i = groups[3][5]                                        # a .cod that announces 2^56 symbols for a few KB of stream

    def sub_dec(op):
        def f(ids, slot):
            data = [enc[i][:max(1, enc[i].size // 2)] if i == cut_i else enc[i] for i in ids]
            pipe.submit_group(slot, op, data, tables=[tabs_d[i] for i in ids],
                              n_symbols=[(1 << 56) if i == huge_i else nsym[i] for i in ids])
        return f

    def ret_dec(fused):
        def f(ids, slot):
            rc, brc, outs, res = pipe.wait_group(slot, len(ids))
            assert rc == 0
            for j, i in enumerate(ids):
                if i == cut_i or i == huge_i:                     # (the row sizes of the group must not wrap on 2^56: pipe.hip)
                    assert brc[j] == shafa.FILE_UNRECOGNIZABLE, f"block {i}: rc {brc[j]}"
                    continue
                assert brc[j] == 0, f"block {i}: rc {brc[j]}"
                src = groups[2][4] if i == bad_i else i
                want = blocks[src] if fused else rle[src]
                assert outs[j] == want.tobytes(), f"block {i}: decoded bytes differ (fused {fused})"
                if fused:
                    assert res[j].mid_n == rle[src].size
        return f

    run(sub_dec(shafa.OP_SF_DECODE), ret_dec(False))
    run(sub_dec(shafa.OP_SF_RLE_DECODE), ret_dec(True))

    def ret_rled(ids, slot):
        rc, brc, outs, res = pipe.wait_group(slot, len(ids))
        assert rc == 0 and not any(brc), (rc, brc)
        for j, i in enumerate(ids):
            assert outs[j] == blocks[i].tobytes(), f"block {i}: RLE decode differs"

    run(lambda ids, slot: pipe.submit_group(slot, shafa.OP_RLE_DECODE, [rle[i] for i in ids]), ret_rled)
    # a slot holds a group or a single block, alternately
    pipe.submit(0, shafa.OP_HIST, blocks[0])
    rc, out, r = pipe.wait(0)
    assert rc == 0 and list(r.freq) == list(oracle.hist256(blocks[0]))
    pipe.close()
