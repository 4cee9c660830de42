#!/usr/bin/env python3
"""CPU model of sfe6_kernel's data movement (shafa-cd_amd/csrc/sf_encode6.hip): the one-shot Shannon-Fano encoder whose
tiles know their output bit offset before the launch (tile histograms x code lengths, scanned).

A workgroup owns one tile = SUB sub-tiles of (lanes x 32) symbols.  Sub-tile k is emitted with sfe5's plain-store algebra
(tools/model/sfe5_model.py) into LDS buffer k & 1, laid out in the OUTPUT's 16-byte alignment: buffer dword i of sub-tile
k is tile-local dword 4 * (S_k >> 7) + i, where S_k is the sub-tile's first bit counted from the 16-byte piece that holds
the tile's first bit.  What this model checks against a direct concatenation of the codes:
  * lane / wave / sub-tile carries (the partial dword in front of a string), the lead bits re-encoded from the 32 symbols
    in front of the tile;
  * the straddling piece between two sub-tiles: its complete dwords are copied from buffer k-1 to the head of buffer k;
  * ownership: every output byte of the block is written exactly once — pieces by 16-byte stores, the tile's head and
    tail dwords one by one, the block's last bytes by the last tile.
Not part of the product; run by hand:  python tools/model/sfe6_model.py
"""
import random

M32 = 0xFFFFFFFF
SUB = 4


def alignbit(hi, lo, sh):
    return (((hi << 32) | lo) >> (sh & 31)) & M32


def make_oct(codes, lens, syms):
    v, L = 0, 0
    for s in syms:
        v = (v << lens[s]) | codes[s]
        L += lens[s]
    return [(v >> (32 * i)) & M32 for i in range(5)], L


def bswap(x):
    return int.from_bytes(x.to_bytes(4, "little"), "big")


class Out:
    def __init__(self, nbytes):
        self.b = [None] * nbytes

    def store(self, addr, data, who):
        for i, x in enumerate(data):
            assert 0 <= addr + i < len(self.b), ("store out of bounds", addr + i, len(self.b), who)
            assert self.b[addr + i] is None, ("byte written twice", addr + i, who, self.b[addr + i])
            self.b[addr + i] = (x, who)

    def store_dword(self, addr, w, who):        # w = big-endian stream dword
        self.store(addr, list(w.to_bytes(4, "big")), who)


def emit_subtile(codes, lens, data, NL, NWV, NW, base, prev_tail, buf):
    """sfe5's emission of one sub-tile of NL*NWV lanes x 32 symbols into `buf` (dict dword -> value), the sub-tile's
    first bit at buffer bit `base` (< 128), the partial dword in front of it = the last bits of prev_tail.
    returns (T, last 32 bits of the string)"""
    NT = NL * NWV
    octs = [[make_oct(codes, lens, data[32 * g + 8 * k: 32 * g + 8 * k + 8]) for k in range(4)] for g in range(NT)]
    tot = [sum(o[1] for o in octs[g]) for g in range(NT)]
    tail = []
    for g in range(NT):
        v = 0
        for k in range(4):
            r, L = octs[g][k]
            v = (v << L) | sum(r[i] << (32 * i) for i in range(5))
        tail.append(v & M32)
    E, acc = [], base
    for g in range(NT):
        acc += tot[g]
        E.append(acc)

    def store(addr, val, who):
        assert addr not in buf, ("double LDS store", addr, who, buf.get(addr))
        buf[addr] = val & M32

    for g in range(NT):
        lane = g % NL
        s0 = E[g] - tot[g]
        if g == 0:
            c = alignbit(prev_tail, 0, s0)
        elif lane == 0:
            c = alignbit(tail[g - 1], 0, s0)               # previous wave's tail through LDS
        else:
            c = alignbit(tail[g - 1], 0, E[g - 1])         # fin of lane - 1 (DPP)
        e = s0
        for k in range(4):
            r, L = octs[g][k]
            e += L
            p, ps = e >> 5, (e - L) >> 5
            j = p - ps
            assert j <= NW - 1
            x = [alignbit(r[0], 0, e)] + [alignbit(r[i], r[i - 1], e) for i in range(1, 5)]
            v = x[j] | c
            if j >= 1:
                store(ps, v, (g, k, 's'))
            for i in range(1, NW - 1):
                if j >= i + 1:
                    store(p - i, x[i], (g, k, i))
            c = v if j == 0 else x[0]
        if g == NT - 1:
            store(e >> 5, c, 'final')
    return E[-1] - base, tail[-1]


def run_tile(codes, lens, blk, t, TILE, NL, NWV, NW, toff, out, last_tile):
    B, Eend = toff[t], toff[t + 1]
    P = B >> 7
    S = [B & 127]
    data = blk[t * TILE:(t + 1) * TILE]
    # lead word: the last 32 symbols in front of the tile, right-aligned (bit i = stream bit B - 1 - i)
    lead = 0
    if t > 0:
        D = 0
        for i in range(32):
            s = blk[t * TILE - 1 - i]
            lead |= (codes[s] << D) & M32 if D < 32 else 0
            D += lens[s]
    SUBN = TILE // SUB
    bufs = [None, None]
    ltail = lead
    np_prev = 0

    def store_sub(k):
        npk = (S[k + 1] >> 7) - (S[k] >> 7)
        d0 = S[0] >> 5
        p0 = 1 if (k == 0 and d0) else 0
        buf = bufs[k & 1]
        for p in range(p0, npk):
            dw = [buf[4 * p + i] for i in range(4)]
            out.store(16 * (P + (S[k] >> 7) + p), sum((list(w.to_bytes(4, "big")) for w in dw), []), ('piece', t, k, p))
        if k == 0 and d0:
            for i in range(d0, 4):
                out.store_dword(16 * P + 4 * i, buf[i], ('head', t, i))
        return npk

    for k in range(SUB):
        buf = {}
        Tk, lt = emit_subtile(codes, lens, data[k * SUBN:(k + 1) * SUBN], NL, NWV, NW, S[k] & 127, ltail, buf)
        S.append(S[k] + Tk)
        bufs[k & 1] = buf
        if k > 0:                      # straddle copy: complete dwords of the piece the sub-tile starts in
            prev = bufs[(k - 1) & 1]
            for i in range((S[k] >> 5) & 3):
                assert i not in buf
                buf[i] = prev[4 * np_prev + i]
            store_sub_prev = store_sub(k - 1)
            assert store_sub_prev == np_prev
        np_prev = (S[k + 1] >> 7) - (S[k] >> 7)
        ltail = lt
    store_sub(SUB - 1)
    assert S[SUB] - S[0] == Eend - B, "tile total differs from the offsets"
    buf = bufs[(SUB - 1) & 1]
    cnt = (S[SUB] >> 5) & 3
    for i in range(cnt):
        out.store_dword(16 * (P + (S[SUB] >> 7)) + 4 * i, buf[4 * np_prev + i], ('tail', t, i))
    if last_tile and (Eend & 31):
        w = buf[4 * np_prev + cnt]
        nb = ((Eend & 31) + 7) >> 3
        out.store(16 * (P + (S[SUB] >> 7)) + 4 * cnt, list(w.to_bytes(4, "big"))[:nb], ('last', t))


def trial(seed, NL, NWV, lmax, NW, ntiles):
    rnd = random.Random(seed)
    lens = [rnd.randint(1, lmax) for _ in range(256)]
    if seed % 3 == 0:
        lens = [rnd.choice([1, 1, 1, 2, lmax]) for _ in range(256)]
    if seed % 5 == 0:
        lens = [1] * 256
    if seed % 7 == 0:
        lens = [lmax] * 256
    codes = [rnd.getrandbits(l) for l in lens]
    TILE = SUB * NL * NWV * 32
    n = ntiles * TILE
    blk = [rnd.randrange(256) if seed % 2 else min(255, int(rnd.expovariate(0.08))) for _ in range(n)]
    toff = [0]
    for t in range(ntiles):
        toff.append(toff[-1] + sum(lens[s] for s in blk[t * TILE:(t + 1) * TILE]))
    total = toff[-1]
    nbytes = (total + 7) >> 3
    out = Out(nbytes)
    order = list(range(ntiles))
    rnd.shuffle(order)                       # tiles are independent workgroups
    for t in order:
        run_tile(codes, lens, blk, t, TILE, NL, NWV, NW, toff, out, t == ntiles - 1)
    v = 0
    for s in blk:
        v = (v << lens[s]) | codes[s]
    v <<= 8 * nbytes - total
    want = v.to_bytes(nbytes, "big")
    for i in range(nbytes):
        assert out.b[i] is not None, ("byte never written", i, nbytes)
        assert out.b[i][0] == want[i], ("byte differs", i, out.b[i], want[i])


if __name__ == "__main__":
    cnt = 0
    for seed in range(1, 40):
        for (NL, NWV, lmax, NW) in ((4, 4, 8, 3), (4, 4, 12, 4), (2, 4, 16, 5), (8, 2, 10, 4)):
            trial(seed, NL, NWV, lmax, NW, 1 + seed % 4)
            cnt += 1
    print("sfe6 model: %d trials ok" % cnt)
