#!/usr/bin/env python3
"""LDS bank-conflict model of the split-slab Lloyd pass with level 1 kept compact (kmeans_pass_mfma_kernel<1, 3, 5, 4, true, L0T>,
csrc/kmeans.hip, `CL1`), access by access, for the 4x6 bank (36 level-0 planes, 36 level-1 planes, the count row, 7 rows of padding).

Rules: MI355X_MICROARCH.md, section LDS (see tools/design/lds_bank_model.py). Prints the extra LDS cycles per TILE (four waves) that
SQ_LDS_BANK_CONFLICT counts, for the layout that ships and, with flags, for its predecessors of round 6:
    lds_bank_model_narrow.py                 ships: rows swizzled, 16-byte update reads for the first L0T = 2 plane tiles
    lds_bank_model_narrow.py --b64           swizzled rows, every update operand as two 8-byte reads
    lds_bank_model_narrow.py --b64 --noswz   the first compact build
Measured (profiles/r6_pmc.txt and its predecessors, cycles per launch / 38 848 tiles): 13.1 M = 338, 8.7 M = 223, 6.4 M = 165 per tile.
The addresses are restated from the kernel (row_addr, row_swz, sdst, a_tr, a_up); keep the two in step."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
NOSWZ, B64 = "--noswz" in sys.argv, "--b64" in sys.argv
L0T = 0 if B64 else 2
KP_TP = 256
KP_PITCH = KP_TP * 2 + 64
KP_P1 = 128 + 48
DL0, DL1, D = 36, 36, 72
ROWS = 80

_a = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
_b = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
G128 = [_a, _b, [l + 32 for l in _a], [l + 32 for l in _b]]
G2X32 = [list(range(32)), list(range(32, 64))]
G8X8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def extra_cycles(addrs, groups, nbytes, nbanks):
    extra = 0
    for g in groups:
        per_bank = {}
        for l in g:
            if addrs[l] is None:
                continue
            for d in range(max(1, nbytes // 4)):
                dw = addrs[l] // 4 + d
                per_bank.setdefault(dw % nbanks, set()).add(dw)
        if per_bank:
            extra += max(len(s) for s in per_bank.values()) - 1
    return extra


rd128 = lambda a: extra_cycles(a, G128, 16, 64)
rd64 = lambda a: extra_cycles(a, G2X32, 8, 64)
wr128 = lambda a: extra_cycles(a, G8X8, 16, 32)
row_addr = lambda r: DL0 * KP_PITCH + (r - DL0) * KP_P1 if r >= DL0 else r * KP_PITCH
row_swz = lambda r: 0 if NOSWZ or r >= DL0 else ((r >> 3) & 1) * 32

tot = {}
add = lambda k, v: tot.__setitem__(k, tot.get(k, 0) + v)
for wave in range(4):
    # assign: ds_read_b64_tr_b16, K-step kk, read rd, sub-tile sub (a_tr + sub * 64)
    for sub in range(2):
        for kk in range(5):
            for rd in range(2):
                ad = []
                for lane in range(64):
                    i16, pxblk, hh = lane & 15, (lane >> 4) & 1, lane >> 5
                    r = 16 * kk + 8 * hh + (i16 >> 2) + 4 * rd
                    off = ((wave * 64 + 16 * pxblk + 4 * (i16 & 3)) * 2) ^ row_swz(r) if r < DL0 else pxblk * 32 + wave * 8
                    ad.append(row_addr(r) + off + sub * 64)
                add("assign  ds_read_b64_tr_b16", rd64(ad))
    # update: plane tile pt, half hf (a_up + hf * 64): one ds_read_b128 below L0T, else two ds_read_b64
    for hf in range(2):
        for pt in range(5):
            a0, a1 = [], []
            for lane in range(64):
                un, ug = lane & 15, lane >> 4
                r = 16 * pt + un
                full = r < DL0
                off = ((wave * 64 + 8 * ug) * 2) ^ row_swz(r) if full else (ug >> 1) * 32 + wave * 8
                a0.append(row_addr(r) + off + hf * 64)
                a1.append(a0[-1] + (8 if full else 0))
            if pt < L0T:
                add("update  ds_read_b128 (plane tiles of level 0)", rd128(a0))
            else:
                add("update  2 x ds_read_b64", rd64(a0) + rd64(a1))
        add("update  labels ds_read_b64", rd64([wave * 64 + hf * 32 + 8 * (l >> 4) for l in range(64)]))
# staging: items of 16 slots, three rounds of 256 threads; level 0: two ds_write_b128 128 bytes apart, level 1: two, 16 bytes apart
n0, nitems = 16 * DL0, 16 * DL0 + 4 * DL1
for i in range(3):
    for w in range(4):
        a0, a1 = [], []
        for lane in range(64):
            ci = min(w * 64 + lane + 256 * i, nitems - 1)
            c1 = ci - n0
            if c1 >= 0:
                d = row_addr(DL0 + (c1 >> 2)) + (c1 & 3) * 32
                a0.append(d), a1.append(d + 16)
            else:
                d = (ci >> 4) * KP_PITCH + (((ci & 1) * 256 + ((ci & 15) >> 1) * 16) ^ row_swz(ci >> 4))
                a0.append(d), a1.append(d + 128)
        add("staging  ds_write_b128", wr128(a0) + wr128(a1))
total = 0
for k in sorted(tot):
    print(f"{k:48s} {tot[k]:5d} extra LDS cycles per tile")
    total += tot[k]
print(f"{'total':48s} {total:5d}   (the writes' conflicts cost no time: a wide store's cycles are set by its register transfer)")
