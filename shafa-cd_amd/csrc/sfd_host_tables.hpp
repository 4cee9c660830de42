// sfd_host_tables.hpp — host side: a block's code table (shafa_code_table) into the look-up tables and tries the kernels read
// Part of sf_decode.hip's translation unit (included there; not compiled on its own).
#pragma once

namespace {
struct HostTab {
    std::vector<u32> trie;     // pairs
    std::vector<u16> lut, lut2, lut13;
    std::vector<u8> lenlut;
    std::vector<u16> longtab;  // LONG_BYTES / 2 entries when 12 < Lmax <= 16 and the code is complete
    std::vector<u16> long32;   // LONG32_BYTES / 2 entries when 12 < Lmax <= 32 and the code is complete
    std::vector<u8> lenlut32;  // lenlut with escape ids 128 + k for the 13-bit prefixes of longer codes (with long32)
    u32 K, K1, lmax;
    bool ok, empty, complete, complete16, complete32;
};

void build_host_tab(const shafa_code_table &t, HostTab &h)
{
    h.trie.assign(2, 0xFFFFFFFFu);
    h.ok = true;
    h.complete = false;
    h.complete16 = false;
    h.complete32 = false;
    h.lmax = 0;
    for (int s = 0; s < 256; ++s) h.lmax = t.len[s] > h.lmax ? t.len[s] : h.lmax;
    h.empty = h.lmax == 0;
    h.K = h.lmax < (u32)LUT_MAXK ? (h.lmax ? h.lmax : 1) : (u32)LUT_MAXK;
    h.lut.assign((size_t)1 << h.K, 0);
    h.K1 = h.lmax < (u32)LEN_MAXK ? (h.lmax ? h.lmax : 1) : (u32)LEN_MAXK;
    h.lenlut.assign(((size_t)1 << h.K1) + 4, 0);
    h.lut13.assign(((size_t)1 << h.K1) + 2, 0);        // codes of <= K1 bits; 0 = longer
    auto code_of = [&](int s) {
        u64 code = 0;       // only the first 32 bits are ever needed here
        const u32 L = t.len[s] < 32 ? t.len[s] : 32;
        for (u32 q = 0; q < L; ++q) code = (code << 1) | ((t.bits[s][q >> 3] >> (7 - (q & 7))) & 1u);
        return (u32)code;   // first min(len,32) bits, right-aligned
    };
    for (int s = 0; s < 256 && h.ok; ++s) {
        const u32 L = t.len[s];
        if (!L) continue;
        u32 node = 0;
        for (u32 q = 0; q < L; ++q) {
            const u32 bit = (t.bits[s][q >> 3] >> (7 - (q & 7))) & 1u;
            u32 &slot = h.trie[2 * node + bit];
            if (q == L - 1) {
                if (slot != 0xFFFFFFFFu) { h.ok = false; break; }          // duplicate / prefix of another
                slot = 0x80000000u | (u32)s;
            } else {
                if (slot == 0xFFFFFFFFu) {
                    slot = (u32)(h.trie.size() / 2);
                    h.trie.push_back(0xFFFFFFFFu);
                    h.trie.push_back(0xFFFFFFFFu);
                } else if (slot & 0x80000000u) { h.ok = false; break; }     // passes through a leaf
                node = h.trie[2 * node + bit];
            }
        }
        if (h.ok && L <= h.K) {
            const u32 code = code_of(s);
            const u32 lo = code << (h.K - L), cnt = 1u << (h.K - L);
            for (u32 i = 0; i < cnt; ++i) h.lut[lo + i] = (u16)(s | (L << 8));
        }
        if (h.ok && L <= h.K1) {
            const u32 code = code_of(s);
            const u32 lo = code << (h.K1 - L), cnt = 1u << (h.K1 - L);
            memset(h.lenlut.data() + lo, (int)L, cnt);
            if (!h.lut13.empty())
                for (u32 i = 0; i < cnt; ++i) h.lut13[lo + i] = (u16)(s | (L << 8));
        }
    }
    if (!h.ok) return;
    h.complete = h.lmax <= (u32)LEN_MAXK;               // every K1-bit window starts a code: pair table usable
    for (size_t i = 0; h.complete && i < ((size_t)1 << h.K1); ++i) h.complete = h.lenlut[i] != 0;
    if (h.lmax <= 16) {                                 // prefix-free (checked above) + Kraft sum 1 = complete tree
        u64 kraft = 0;
        for (int s = 0; s < 256; ++s) if (t.len[s]) kraft += 1ull << (16 - t.len[s]);
        h.complete16 = kraft == 65536;
    }
    if (h.lmax > (u32)SYM3_MAXK && h.lmax <= 32) {      // complete tree with codes of 13..32 bits
        u64 kraft = 0;
        for (int s = 0; s < 256; ++s) if (t.len[s]) kraft += 1ull << (32 - t.len[s]);
        h.complete32 = kraft == (1ull << 32);
    }
    if (h.complete32) {                                 // 12-bit prefixes of the codes of 13..32 bits + their sub-tries
        h.long32.assign(LONG32_BYTES / 2, 0);
        struct Item { u32 key, node; };
        std::vector<Item> roots;
        // walk the trie to depth SYM3_MAXK
        std::vector<Item> frontier{{0u, 0u}};
        for (int depth = 0; depth < SYM3_MAXK; ++depth) {
            std::vector<Item> next;
            for (const Item &it : frontier)
                for (u32 b = 0; b < 2; ++b) {
                    const u32 c = h.trie[2 * it.node + b];
                    if (c != 0xFFFFFFFFu && !(c & 0x80000000u)) next.push_back({(it.key << 1) | b, c});
                }
            frontier.swap(next);
        }
        roots = frontier;                               // internal nodes at depth 12, keys ascending by construction? sort anyway
        std::sort(roots.begin(), roots.end(), [](const Item &a, const Item &b) { return a.key < b.key; });
        std::vector<u32> order;                         // sub-trie nodes, renumbered in BFS order
        std::vector<int> newid(h.trie.size() / 2, -1);
        if (roots.size() > (size_t)LONG_PFX) h.complete32 = false;
        for (size_t g = 0; g < roots.size() && h.complete32; ++g) {
            std::vector<u32> q{roots[g].node};
            for (size_t qi = 0; qi < q.size(); ++qi) {
                const u32 nd = q[qi];
                newid[nd] = (int)order.size();
                order.push_back(nd);
                for (u32 b = 0; b < 2; ++b) {
                    const u32 c = h.trie[2 * nd + b];
                    if (!(c & 0x80000000u)) q.push_back(c);
                }
            }
        }
        // BFS numbering above interleaves ids before children are numbered: assign child ids in a second pass
        if (order.size() > 256) h.complete32 = false;
        if (h.complete32) {
            h.long32[0] = (u16)roots.size();
            h.long32[1] = (u16)order.size();               // sub-trie nodes in use (the passes that walk them keep only those in LDS)
            for (size_t g = 0; g < roots.size(); ++g) {
                h.long32[8 + g] = (u16)roots[g].key;
                h.long32[8 + LONG_PFX + g] = (u16)newid[roots[g].node];
            }
            for (size_t i = 0; i < order.size(); ++i)
                for (u32 b = 0; b < 2; ++b) {
                    const u32 c = h.trie[2 * order[i] + b];
                    h.long32[8 + 2 * LONG_PFX + 2 * i + b] = (c & 0x80000000u) ? (u16)(0x8000u | (c & 0xFFu)) : (u16)newid[c];
                }
            // 13-bit prefixes of the codes longer than 13 bits: lenlut32 names their trie node
            if (h.lmax > (u32)LEN_MAXK) {
                h.lenlut32.assign(h.lenlut.begin(), h.lenlut.end());
                std::vector<Item> f13{{0u, 0u}};
                for (int depth = 0; depth < LEN_MAXK; ++depth) {
                    std::vector<Item> next;
                    for (const Item &it : f13)
                        for (u32 b = 0; b < 2; ++b) {
                            const u32 c = h.trie[2 * it.node + b];
                            if (c != 0xFFFFFFFFu && !(c & 0x80000000u)) next.push_back({(it.key << 1) | b, c});
                        }
                    f13.swap(next);
                }
                if (f13.size() > (size_t)LONG_PFX) { h.complete32 = false; h.long32.clear(); h.lenlut32.clear(); }
                else
                    for (size_t k = 0; k < f13.size(); ++k) {
                        h.long32[8 + 2 * LONG_PFX + 512 + k] = (u16)newid[f13[k].node];
                        h.lenlut32[f13[k].key] = (u8)(128 + k);
                    }
            }
        } else h.long32.clear();
    }
    if (h.complete16 && h.lmax > (u32)SYM3_MAXK) {      // codes of 13..16 bits, grouped by their first 12 bits
        h.longtab.assign(LONG_BYTES / 2, 0);
        std::vector<u32> keys;
        for (int s = 0; s < 256; ++s) if (t.len[s] > (u32)SYM3_MAXK) keys.push_back(code_of(s) >> (t.len[s] - SYM3_MAXK));
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        if (keys.size() > (size_t)LONG_PFX) { h.complete16 = false; h.longtab.clear(); }
        else {
            h.longtab[0] = (u16)keys.size();
            for (size_t g = 0; g < keys.size(); ++g) h.longtab[8 + g] = (u16)keys[g];
            for (int s = 0; s < 256; ++s) {
                const u32 L = t.len[s];
                if (L <= (u32)SYM3_MAXK) continue;
                const u32 code = code_of(s), key = code >> (L - SYM3_MAXK);
                const size_t g = std::lower_bound(keys.begin(), keys.end(), key) - keys.begin();
                const u32 suf = (code & ((1u << (L - SYM3_MAXK)) - 1)) << (16 - L), cnt = 1u << (16 - L);
                for (u32 i = 0; i < cnt; ++i) h.longtab[8 + LONG_PFX + g * 16 + suf + i] = (u16)(s | (L << 8));
            }
        }
    }
    // level 2: group the codes of K+1..K+8 bits by their first K bits
    for (int s = 0; s < 256; ++s) {
        const u32 L = t.len[s];
        if (L <= h.K || L > h.K + 8) continue;
        const u32 pre = code_of(s) >> (L > 32 ? 32 - h.K : L - h.K);      // L <= K+8 <= 19 here
        if (h.lut[pre]) continue;                                           // group already built
        u32 maxl = 0;
        for (int s2 = 0; s2 < 256; ++s2) {
            const u32 L2 = t.len[s2];
            if (L2 > h.K && L2 <= h.K + 8 && (code_of(s2) >> (L2 - h.K)) == pre && L2 > maxl) maxl = L2;
        }
        const u32 nb = maxl - h.K;
        const u32 base = (u32)h.lut2.size();
        if (base + (1u << nb) > (u32)LUT2_MAX) continue;                    // no room: these codes use the trie
        h.lut2.resize(base + (1u << nb), 0);
        for (int s2 = 0; s2 < 256; ++s2) {
            const u32 L2 = t.len[s2];
            if (L2 > h.K && L2 <= h.K + nb && (code_of(s2) >> (L2 - h.K)) == pre) {
                const u32 sub = code_of(s2) & ((1u << (L2 - h.K)) - 1);
                const u32 lo = sub << (h.K + nb - L2), cnt = 1u << (h.K + nb - L2);
                for (u32 i = 0; i < cnt; ++i) h.lut2[base + lo + i] = (u16)(s2 | (L2 << 8));
            }
        }
        h.lut[pre] = (u16)(0x8000u | ((nb - 1) << 12) | base);
    }
    if (h.lut2.size() & 1) h.lut2.push_back(0);
}
}  // namespace
