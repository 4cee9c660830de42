#!/usr/bin/env python3
"""CPU model of sfe5's window emission (shafa-cd_amd/csrc/sf_encode5.hip): every lane owns 32 consecutive symbols
(four octs), octs are right-aligned register strings, and the tile-local LDS window is filled with PLAIN dword stores:
an oct stores every dword it has bits in except the one it ends in, which travels to the next oct as a carry
(next lane: DPP wave_shr:1; next wave: the previous wave's last 32 bits via LDS).  Checks the algebra against a direct
concatenation of the codes.  Not part of the product; run by hand:  python tools/model/sfe5_model.py
"""
import random
import sys

M32 = 0xFFFFFFFF


def alignbit(hi, lo, sh):
    return (((hi << 32) | lo) >> (sh & 31)) & M32


def make_oct(codes, lens, syms):
    v, L = 0, 0
    for s in syms:
        v = (v << lens[s]) | codes[s]
        L += lens[s]
    return [(v >> (32 * i)) & M32 for i in range(5)], L


def run_tile(codes, lens, data, NT, NW):
    """returns (window dwords dict, T)"""
    NWV = NT // 64
    octs = [[make_oct(codes, lens, data[32 * g + 8 * k: 32 * g + 8 * k + 8]) for k in range(4)] for g in range(NT)]
    tot = [sum(o[1] for o in octs[g]) for g in range(NT)]
    # last 32 bits of every lane's string
    tail = []
    for g in range(NT):
        v = 0
        for k in range(4):
            r, L = octs[g][k]
            v = ((v << L) | (r[0] | (r[1] << 32) | (r[2] << 64) | (r[3] << 96) | (r[4] << 128)))
        tail.append(v & M32)
    E = []
    acc = 0
    for g in range(NT):
        acc += tot[g]
        E.append(acc)
    T = acc
    win = {}

    def store(addr, val, who):
        assert addr not in win, ("double store", addr, who, win.get(addr))
        win[addr] = (val & M32, who)

    for g in range(NT):
        lane, wv = g & 63, g >> 6
        s0 = E[g] - tot[g]
        if lane == 0:
            c = alignbit(tail[g - 1], 0, s0) if wv > 0 else 0
        else:
            c = alignbit(tail[g - 1], 0, E[g - 1])          # fin of lane - 1, by DPP
        e = s0
        for k in range(4):
            r, L = octs[g][k]
            e += L
            s = e - L
            p, ps = e >> 5, s >> 5
            j = p - ps
            assert j <= NW - 1, (j, L)
            x = [alignbit(r[0], 0, e), alignbit(r[1], r[0], e), alignbit(r[2], r[1], e), alignbit(r[3], r[2], e),
                 alignbit(r[4], r[3], e)]
            v = x[j] | c
            if j >= 1:
                store(ps, v, (g, k, 's'))
            for i in range(1, NW - 1):
                if j >= i + 1:
                    store(p - i, x[i], (g, k, i))
            c = v if j == 0 else x[0]
        if g == NT - 1:
            store(e >> 5, c, 'final')
            store((e >> 5) + 1, 0, 'pad')
    return win, T


def direct(codes, lens, data):
    v, L = 0, 0
    for s in data:
        v = (v << lens[s]) | codes[s]
        L += lens[s]
    return v, L


def trial(seed, NT, lmax, NW):
    rnd = random.Random(seed)
    # random prefix-free-ish lengths do not matter for the placement algebra: any (code, len >= 1)
    lens = [rnd.randint(1, lmax) for _ in range(256)]
    if seed % 3 == 0:
        lens = [rnd.choice([1, 1, 1, 2, lmax]) for _ in range(256)]
    if seed % 5 == 0:
        lens = [1] * 256
    if seed % 7 == 0:
        lens = [lmax] * 256
    codes = [rnd.getrandbits(l) for l in lens]
    data = [rnd.randrange(256) if seed % 2 else min(255, int(rnd.expovariate(0.08))) for _ in range(32 * NT)]
    win, T = run_tile(codes, lens, data, NT, NW)
    v, L = direct(codes, lens, data)
    assert L == T
    nd = (T >> 5) + 1
    v <<= (32 * nd - T)                          # left-align in nd dwords, zero padded
    for d in range(nd):
        want = (v >> (32 * (nd - 1 - d))) & M32
        assert d in win, ("dword never stored", d, T)
        assert win[d][0] == want, ("dword differs", d, hex(win[d][0]), hex(want), win[d][1])
    assert set(win) == set(range(nd + 1)), "stores outside [0, T/32 + 1]"


if __name__ == "__main__":
    n = 0
    for seed in range(1, 60):
        for (NT, lmax, NW) in ((256, 8, 3), (256, 12, 4), (256, 16, 5), (128, 15, 5)):
            trial(seed, NT, lmax, NW)
            n += 1
    print("sfe5 model: %d trials ok" % n)
