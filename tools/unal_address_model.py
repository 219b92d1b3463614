#!/usr/bin/env python3
"""CPU model of the addresses the aligned-down LDS-DMA rows (UNAL variants of k_sobel_dma, k_fused_dma, k_conv3x3_dma) fetch.
For every (kernel, row bytes, strip, base offset, row) of a sweep it checks that
  * every 16-byte block fetched contains at least one byte of the readable frame (so no access can leave the frame's pages),
  * the bytes the row step needs -- the strip and its left / right neighbours -- are all inside fetched blocks,
  * the piece lands inside its LDS slot at a dword-aligned address.
Developer tool; run it after touching the a' / tail rules:  python3 tools/unal_address_model.py"""
import itertools, sys

STRIP = 1024


def rules(kind, ch, a, xw_units, has_right, nl, nr, strip=STRIP):
    S = strip * ch
    if kind == "sobel":
        ap = 16 if (a == 0 and xw_units > 0) else a
        t0 = S - 16 if (ap == 0 and not has_right) else S
        t1 = S + 16 if (ap == 16 and has_right) else t0
    elif kind == "fused":
        ap = a + 16 if (a < ch and xw_units > 0) else a
        t0 = S - 16 if (ap == 0 and not has_right) else S
        t1 = S + 16 if (ap > 16 or (has_right and ap + ch > 16)) else t0
    else:  # conv (units are bytes)
        ap = a + 16 if a < nl else a
        t0 = S - 16 if (ap == 0 and not has_right) else S
        t1 = S + 16 if (ap > 16 or (nr > 0 and ap + nr > 16)) else t0
    return ap, t0, t1


def check(kind, ch, width_units, rows, base, halo_rows, strip=STRIP):
    """width_units: pixels (sobel, fused) or bytes (conv: ch = 1 and the halo is 3 units); strip = 2048: k_sobel_dma2's strip pairs."""
    if width_units < strip:
        return 0
    P = width_units * ch
    F0 = base - (P if halo_rows else 0)
    F1 = base + (rows + (1 if halo_rows else 0)) * P
    n = 0
    nstrips = (width_units + strip - 1) // strip
    for sidx in range(nstrips):
        xw = sidx * strip
        if xw + strip > width_units:
            xw = width_units - strip
        has_right = xw + strip < width_units
        hal = 3 if kind == "conv" else 1
        nl = min(hal, xw) * ch
        nr = min(hal, width_units - xw - strip) * ch
        ylo, yhi = (-1, rows) if halo_rows else (0, rows - 1)
        for yy in sorted({ylo, ylo + 1, 1, rows // 2, yhi - 1, yhi} & set(range(ylo, yhi + 1))):
            A = base + yy * P + xw * ch
            a = A & 15
            ap, t0, t1 = rules(kind, ch, a, xw, has_right, nl if kind == "conv" else None, nr if kind == "conv" else None, strip)
            G = A - ap
            assert G % 16 == 0
            S = strip * ch
            blocks = [G + 16 * k for k in range(S // 16)] + [G + t0, G + t1]
            for b in blocks:
                assert b < F1 and b + 16 > F0, (kind, ch, width_units, rows, base & 15, sidx, yy, "block outside the frame", b - F0, F1 - b)
            covered = set(blocks)
            for byte in range(A - nl, A + S + nr):
                assert (byte & ~15) in covered, (kind, ch, width_units, base & 15, sidx, yy, "needed byte not fetched", byte - A)
            land = 16 + (a & 3) - ap          # LDS offset of the piece inside the slot
            assert land >= 0 and land % 4 == 0 and land + S + 32 <= 16 + S + 32 + 3, (kind, a, ap, land)
            n += 1
    return n


def sweep(widths, offsets=range(16)):
    widths = list(widths)
    total = 0
    for kind, ch, strip in (("sobel", 1, STRIP), ("sobel", 1, 2 * STRIP), ("fused", 3, STRIP), ("fused", 4, STRIP), ("conv", 1, STRIP)):
        for w in widths:
            for so in offsets:
                for rows, halo in ((1, False), (2, False), (5, False), (7, True), (1, True)):
                    if kind == "conv" and halo:
                        continue                 # the convolution has a zero border: rows outside the frame are never read
                    total += check(kind, ch, w, rows, (1 << 20) + so, halo, strip)
    return total


if __name__ == "__main__":
    print("ok:", sweep(list(range(1024, 1060)) + list(range(2040, 2070)) + [3073, 4099, 5000, 16370, 16380]), "rows checked")
