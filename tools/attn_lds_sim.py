#!/usr/bin/env python
"""Dev check (CPU): the LDS images of attn_dma_kernel - what each 16-byte slot holds after the DMA pieces land, what each lane's
ds_read_b128 fetches, and the bank-slot conflicts per ds_read_b128 lane group (MI355X_MICROARCH.md, LDS table)."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def span(KC):
    return 1 if KC % 2 else (2 if KC % 4 == 2 else (4 if KC % 8 == 4 else (8 if KC % 16 == 8 else 16)))


def kswz(KC, rho):
    S = span(KC)
    return {1: 0, 2: (rho >> 3) & 1, 4: ((rho >> 2) & 1) | (((rho >> 4) & 1) << 1), 8: (rho >> 1) & 7, 16: rho & 15}[S]


def worst(addrs_by_lane):
    w = 1
    for g in GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addrs_by_lane[l] // 16) % 16, set()).add(addrs_by_lane[l])
        w = max(w, max(len(v) for v in slots.values()))
    return w


for KC in (5, 6, 8, 10, 16, 20):
    Dh, KS, S = KC * 8, (KC + 1) // 2, span(KC)
    img = {}
    for i in range(64 * KC):  # DMA: LDS chunk i <- (key, channel chunk)
        rho, cpos = divmod(i, KC)
        c = cpos ^ kswz(KC, rho)
        key = (rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1)
        assert 0 <= c < KC
        img[i * 16] = (key, c)
    assert len(set(img.values())) == 64 * KC
    wk = 1
    for kb in range(2):
        for s in range(KS):
            addrs = {}
            for lane in range(64):
                r, hh = lane & 31, lane >> 5
                kaddr = r * KC * 16 + ((hh ^ kswz(KC, r)) << 4)
                base = r * KC * 16 if (KC & 1 and s == KS - 1) else kaddr
                lo, hi = ((2 * s) & (S - 1)) << 4, ((2 * s) & ~(S - 1)) << 4
                a = (base ^ lo) + hi + kb * 32 * KC * 16
                addrs[lane] = a
                key, c = img[a]
                rho = kb * 32 + r
                assert key == ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1)), (KC, kb, s, lane)
                assert c == min(2 * s + hh, KC - 1), (KC, kb, s, lane, c)
            wk = max(wk, worst(addrs))
    # V^T image: rows Dh data + ones + zero, 8 chunks each
    NV = (Dh + 31) // 32
    vimg = {}
    for i in range(Dh * 8):
        d, cpos = divmod(i, 8)
        vimg[i * 16] = (d, cpos ^ ((d >> 1) & 7))
    wv = 1
    for dv in range(NV):
        for kb in range(2):
            for s2 in range(2):
                addrs = {}
                for lane in range(64):
                    r, hh = lane & 31, lane >> 5
                    R = dv * 32 + r
                    lrow = R if R < Dh + 1 else Dh + 1
                    va = lrow * 128 + ((((lrow >> 1) & 7) ^ hh) << 4)
                    a = va ^ ((kb * 4 + 2 * s2) << 4)
                    addrs[lane] = a
                    if lrow < Dh:
                        assert vimg[a] == (lrow, kb * 4 + 2 * s2 + hh), (KC, dv, kb, s2, lane)
                    else:
                        assert lrow * 128 <= a < lrow * 128 + 128
                wv = max(wv, worst(addrs))
    print(f"KC {KC:2d} (Dh {Dh:3d}): K image ok, worst K conflict {wk}-way; V^T image ok, worst V^T conflict {wv}-way")
