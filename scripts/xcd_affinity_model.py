#!/usr/bin/env python
"""A MODEL, not a measurement: for the kernels of one UNITER-base forward layer at the benchmark shape (32 examples x 96 tokens),
which fraction of the bytes a consumer workgroup reads from the previous kernel's output was WRITTEN by a workgroup on the same XCD
— i.e. could be served by that XCD's L2 if the producer stored with the default policy — under the tile -> XCD maps the library
uses today (csrc/gemm_args.cuh: tile_of_block / xcd_remap, csrc/gemm.hip: pick_xr; hardware block b runs on XCD b % 8) and under
one row-block -> XCD affinity for the whole chain.
usage: python scripts/xcd_affinity_model.py"""
import numpy as np

B, L, H, I, HEADS = 32, 96, 768, 3072, 12
T = B * L


def pick_xr(tiles_m, tiles_n, bm, bn):
    best, best_cost = 0, -1
    xr = 1
    while xr <= 8:
        xc = 8 // xr
        if tiles_m % xr == 0 and tiles_n % xc == 0:
            cost = (tiles_m // xr) * bm + (tiles_n // xc) * bn
            if best_cost < 0 or cost < best_cost:
                best_cost, best = cost, xr
        xr *= 2
    return best


def xcd_remap(bid, nblk):
    q, r = nblk >> 3, nblk & 7
    xcd, loc = bid & 7, bid >> 3
    start = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return start + loc


def gemm_owner(M, N, bm, bn, xr=None):
    """[M, N] array: XCD that writes each output element."""
    tiles_m, tiles_n = (M + bm - 1) // bm, N // bn
    if xr is None:
        xr = pick_xr(tiles_m, tiles_n, bm, bn)
    owner = np.zeros((M, N), dtype=np.int8)
    for bx in range(tiles_m * tiles_n):
        xcd = bx & 7
        if xr > 0:
            loc, xc = bx >> 3, 8 // xr
            sub_m, sub_n = tiles_m // xr, tiles_n // xc
            tm = (xcd // xc) * sub_m + loc // sub_n
            tn = (xcd % xc) * sub_n + loc % sub_n
        else:
            tile = xcd_remap(bx, tiles_m * tiles_n)
            tm, tn = tile // tiles_n, tile % tiles_n
        owner[tm * bm:(tm + 1) * bm, tn * bn:(tn + 1) * bn] = xcd
    return owner, xr


def gemm_reader(M, K_cols, bm, N, bn, xr=None):
    """For a GEMM whose M-side operand is [M, K_cols]: per XCD, the set of operand rows its tiles read (all K columns)."""
    tiles_m, tiles_n = (M + bm - 1) // bm, N // bn
    out_owner, xr = gemm_owner(M, N, bm, bn, xr)
    rows = [set() for _ in range(8)]
    weight = np.zeros((8, tiles_m), dtype=np.int32)         # how many tiles of XCD x read row block tm
    for tm in range(tiles_m):
        for tn in range(tiles_n):
            weight[out_owner[tm * bm, tn * bn], tm] += 1
    return weight, xr


def local_fraction_gemm(prod_owner, M, K_cols, bm, N, bn, xr=None):
    """Fraction of M-side operand bytes (counted once per reading tile) written by the reading tile's XCD."""
    weight, xr = gemm_reader(M, K_cols, bm, N, bn, xr)
    tiles_m = weight.shape[1]
    same = total = 0
    for tm in range(tiles_m):
        block = prod_owner[tm * bm:(tm + 1) * bm, :]
        for x in range(8):
            if weight[x, tm]:
                same += weight[x, tm] * int((block == x).sum())
                total += weight[x, tm] * block.size
    return same / total, xr


def attention_owner(affinity):
    """ctx [T, H]: XCD of the workgroup that writes each element.  Workgroup w handles units bh = 2w, 2w + 1 (bh = b * heads + h)."""
    owner = np.zeros((T, H), dtype=np.int8)
    for w in range(B * HEADS // 2):
        for slot in range(2):
            bh = 2 * w + slot
            b, h = bh // HEADS, bh % HEADS
            xcd = (b // (B // 8)) if affinity else (w % 8)
            owner[b * L:(b + 1) * L, h * 64:(h + 1) * 64] = xcd
    return owner


def attention_local_fraction(qkv_owner, affinity):
    same = total = 0
    for w in range(B * HEADS // 2):
        for slot in range(2):
            bh = 2 * w + slot
            b, h = bh // HEADS, bh % HEADS
            xcd = (b // (B // 8)) if affinity else (w % 8)
            for part in range(3):
                block = qkv_owner[b * L:(b + 1) * L, part * H + h * 64: part * H + (h + 1) * 64]
                same += int((block == xcd).sum())
                total += block.size
    return same / total


def rows_owner(affinity, rows_per_block=4):
    """LayerNorm: block k normalises rows 4k .. 4k+3; XCD = k % 8 today, the row block's XCD under affinity."""
    owner = np.zeros((T, H), dtype=np.int8)
    for k in range(T // rows_per_block):
        r0 = k * rows_per_block
        owner[r0:r0 + rows_per_block, :] = ((r0 // L) // (B // 8)) if affinity else (k % 8)
    return owner


def rows_local_fraction(prod_owner, affinity, rows_per_block=4):
    mine = rows_owner(affinity, rows_per_block)
    return float((prod_owner == mine).mean())


def main():
    print("# scripts/xcd_affinity_model.py — a MODEL of where a consumer's input bytes were written (same XCD = could be an L2 hit), no measurement")
    print("# UNITER-base forward layer, 32 x 96 tokens; GEMM tiles as in uniter_amd/tuned/gfx950.json; affinity: example b -> XCD b // 4 everywhere (what 8 XCD rows of 96-row tiles give)")
    for affinity in (False, True):
        xr8 = 8 if affinity else None
        print("## %s" % ("one row-block -> XCD affinity for the whole chain (GEMM tiles of 96 rows with 8 XCD rows: row block tm -> XCD tm // 4; attention units and LayerNorm rows by example)" if affinity
                         else "today's maps (each kernel picks its own)"))
        qkv_owner, xr = gemm_owner(T, 3 * H, 128 if not affinity else 96, 128, xr8)
        f = attention_local_fraction(qkv_owner, affinity)
        print("  QKV GEMM (xr=%d) -> attention: %5.1f %% of the q/k/v bytes a unit reads were written on its XCD" % (xr, 100 * f))
        ctx_owner = attention_owner(affinity)
        f, xr = local_fraction_gemm(ctx_owner, T, H, 96, H, 96, xr8)
        print("  attention -> out-proj GEMM (xr=%d): %5.1f %%" % (xr, 100 * f))
        z1_owner, _ = gemm_owner(T, H, 96, 96, xr8)
        print("  out-proj -> LayerNorm: %5.1f %%" % (100 * rows_local_fraction(z1_owner, affinity)))
        a_owner = rows_owner(affinity)
        f, xr = local_fraction_gemm(a_owner, T, H, 96, I, 192, xr8)
        print("  LayerNorm -> FFN1 GEMM (xr=%d): %5.1f %%" % (xr, 100 * f))
        g_owner, _ = gemm_owner(T, I, 96, 192, xr8)
        f, xr = local_fraction_gemm(g_owner, T, I, 96, H, 96, xr8)
        print("  FFN1 -> FFN2 GEMM (xr=%d): %5.1f %%" % (xr, 100 * f))
        z2_owner, _ = gemm_owner(T, H, 96, 96, xr8)
        print("  FFN2 -> LayerNorm: %5.1f %%" % (100 * rows_local_fraction(z2_owner, affinity)))
        y_owner = rows_owner(affinity)
        f, xr = local_fraction_gemm(y_owner, T, H, 128 if not affinity else 96, 3 * H, 128, xr8)
        print("  LayerNorm -> next QKV GEMM (xr=%d): %5.1f %%" % (xr, 100 * f))


if __name__ == '__main__':
    main()
