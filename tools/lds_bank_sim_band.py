"""Diagnostic: LDS bank conflicts of fwd_band's channel sums (thread t = pixel * G + group, one ds_read_b128 per
(channel quad, direction)) and of its NCHW transposing commit, as a function of the slab's row stride Ppb.
profiles/r03_b_fwd_band_pmc_baseline.csv: 52 % of the LDS-active cycles of the [4096,512,7,7] forward were conflicts
(G = 8, Ppb = 53); this model reproduces that and picks the stride per G.   usage: python tools/lds_bank_sim_band.py"""
from lds_bank_sim import cycles_read_b128, cycles_write_b128, swz


def read_cycles(G, Ps, W, Ppb, T):
    tot = n = 0
    for w in range((T + 63) // 64):
        for d in (0, 1, W - 1, W, W + 1):
            addrs = []
            for l in range(64):
                t = w * 64 + l
                gl, lp = t % G, t // G
                if lp >= Ps:
                    addrs.append(None)
                    continue
                q = lp + d
                q = q if q < Ps else lp
                addrs.append((gl * Ppb + swz(q)) * 16)
            tot += cycles_read_b128(addrs)
            n += 1
    return tot / n


def write_cycles(Ps, Ppb, T, ncq):
    NQb = (Ps + 3) // 4
    tot = n = 0
    for w in range((T + 63) // 64):
        for j in range(4):
            addrs = []
            for l in range(64):
                i = w * 64 + l
                cq, pq = divmod(i, NQb)
                if cq >= ncq:
                    addrs.append(None)
                    continue
                ps = min(4 * pq, Ps - 4)
                addrs.append((cq * Ppb + swz(ps + j)) * 16)
            tot += cycles_write_b128(addrs)
            n += 1
    return tot / n


if __name__ == "__main__":
    for (G, Ps, W, T, label) in ((8, 49, 7, 448, "[4096,512,7,7]: one band, G = 8"), (32, 21, 7, 704, "[64,512,7,7]: 4 bands, G = 32"),
                                  (16, 49, 7, 832, "[256,512,7,7]: G = 16"), (4, 98, 14, 448, "14x14, 2 bands, G = 4"),
                                  (2, 196, 14, 448, "14x14 one band G = 2"), (8, 98, 14, 832, "14x14 2 bands G = 8")):
        base = ((Ps + 3) & ~3)
        print(label)
        for Ppb in range(base, base + 18):
            r, wcy = read_cycles(G, Ps, W, Ppb, T), write_cycles(Ps, Ppb, T, 64)
            print(f"   Ppb={Ppb:4d} (mod 16 = {Ppb % 16:2d})  read {r:5.2f} (ideal 4)   write {wcy:5.2f} (ideal 8)")
