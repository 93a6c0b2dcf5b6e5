#!/usr/bin/env python3
"""Exhaustive bank-conflict check of the halo kernel's LDS image (lic_halo_bf16.h): for every tap (plane, column
shift), halo row, K step and ds_read_b128 lane group the 16 lanes must touch 16 distinct 16-byte bank slots.
CPU only: python tools/halo_bank_check.py"""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
WP, HR = [34, 33], 19
PB = [0, HR * 34 * 64]
worst = 0
for pl in (0, 1):
    for sh in ((0, 1, 2) if pl == 0 else (0, 1)):
        for hr in range(HR):
            for ks in (0, 1):
                for g in GROUPS:
                    pos = set()
                    for lane in g:
                        li, lh = lane & 31, lane >> 5
                        hc = li + sh
                        addr = PB[pl] + (hr * WP[pl] + hc) * 64 + 16 * ((2 * ks + lh) ^ ((hc >> 2) & 3))
                        pos.add((addr // 16) % 16)
                    worst = max(worst, 16 - len(pos))
print("bank slots lost to conflicts in the worst lane group:", worst)
assert worst == 0
