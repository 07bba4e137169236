"""CPU model of k_atb_bx3 (csrc/l2o_atb.h): the weight-gradient contraction A^T Bm on the bf16 matrix pipe.

Two things that can be checked without a GPU:
  * the ARITHMETIC: both operands split into three bf16 levels (RNE), the six products x1y1, x1y2, x2y1, x2y2, x1y3,
    x3y1 per 8-row K-group with the pipe's measured rounding (every product chopped toward zero at 2^-24 of the group's
    largest, scripts/microbench/mfma_round_probe.hip), fp32 accumulation over groups, blocks and split-K partials --
    against a float64 product, in units of sum |a||b|;
  * the LDS IMAGE: [level][column][4 octets of 16 bytes] with the octet index XORed with (column >> 1) & 3 is free of
    bank conflicts for the lane groups ds_read_b128 / ds_write_b128 are really served in (MI355X_MICROARCH.md, LDS),
    and the reader of (tile, m, kq) finds what the writer of (column, wave) stored.
The GPU tests (test_hip_kernels.py: wgrad / atb) run the kernel itself against float64."""
import numpy as np

from mfma_emulator import split3


def chop_group(P):
    """P [..., 8] float64 products of one K-group: toward zero at 2^-24 of the group's largest."""
    m = np.abs(P).max(-1, keepdims=True)
    e = np.floor(np.log2(np.maximum(m, 1e-300)))
    q = np.ldexp(1.0, (e - 24).astype(int))
    return np.trunc(P / q) * q


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)


def atb_bx3_model(A, B, groups=4):
    """G = A^T B the way k_atb_bx3 forms it: 32-row blocks dealt round-robin to `groups` persistent workgroups, per
    block and product one MFMA = four K-groups of 8 rows added to the fp32 accumulator one after the other."""
    R, KA = A.shape
    KB = B.shape[1]
    a_lv, b_lv = split3(A), split3(B)
    a_lv = [x.astype(np.float64) for x in a_lv]
    b_lv = [x.astype(np.float64) for x in b_lv]
    order = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]            # small products first (the kernel's order)
    part = np.zeros((groups, KA, KB))
    nblk = (R + 31) // 32
    for blk in range(nblk):
        g = blk % groups
        acc = part[g]
        for la, lb in order:
            for kq in range(4):
                r0 = blk * 32 + 8 * kq
                rows = slice(r0, min(r0 + 8, R))
                if rows.start >= R:
                    continue
                a, b = a_lv[la][rows], b_lv[lb][rows]                    # [k, KA], [k, KB]
                P = a.T[:, None, :] * b.T[None, :, :]                    # [KA, KB, k]
                if P.shape[-1] < 8:
                    P = np.concatenate([P, np.zeros(P.shape[:2] + (8 - P.shape[-1],))], -1)
                acc = f32(acc + chop_group(P).sum(-1))
        part[g] = acc
    out = np.zeros((KA, KB))
    for g in range(groups):                                              # (the real reduction's order differs; fp32 adds)
        out = f32(out + part[g])
    return out


def test_bx3_contraction_model_matches_float64():
    rng = np.random.default_rng(0)
    R, KA, KB = 2048 + 13, 12, 20
    A = np.tanh(rng.standard_normal((R, KA))).astype(np.float32)
    A[:, -1] = 1.0                                                       # the bias column: products of one sign
    B = (1e-3 * np.exp(2.0 * rng.standard_normal((R, 1))) * rng.standard_normal((R, KB))).astype(np.float32)
    got = atb_bx3_model(A, B)
    want = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    # an exact-product fp32 accumulation of the same blocks, for scale
    ref32 = np.zeros((KA, KB))
    for r0 in range(0, R, 8):
        ref32 = f32(ref32 + A[r0:r0 + 8].astype(np.float64).T @ B[r0:r0 + 8].astype(np.float64))
    e_bx3 = float((np.abs(got - want) / mag).max())
    e_f32 = float((np.abs(ref32 - want) / mag).max())
    print("bf16x3 model: err / sum|a||b| %.3g; exact products, fp32 sums: %.3g" % (e_bx3, e_f32))
    assert e_bx3 < 5e-7
    assert e_bx3 < 4 * e_f32 + 2e-8                                      # the split costs no more than the accumulation does


def test_three_level_split_is_exact_to_24_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(100000) * np.exp(4 * rng.standard_normal(100000))).astype(np.float32)
    x1, x2, x3 = split3(x)
    back = x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64)
    assert float(np.max(np.abs(back - x) / np.abs(x))) < 2.0 ** -23


# ---- the LDS image -----------------------------------------------------------------------------------------------------
READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]               # ds_read_b128: 4 x 16 lanes, banks (a / 4) mod 64
WRITE_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]        # ds_write_b128: 8 x 8 lanes, banks (a / 4) mod 32


def image_offset16(level, col, octet, ncols):
    """index (in 16-byte units) of rows 8 octet .. 8 octet + 7 of `col` at split level `level` (csrc/l2o_atb.h)."""
    return (level * ncols + col) * 4 + (octet ^ ((col >> 1) & 3))


def banks(off16, nbanks):
    return {(4 * off16 + d) % nbanks for d in range(4)}


def test_lds_image_is_conflict_free_and_consistent():
    for ncols in (96, 176, 112, 192):
        # writer: wave wv (= octet), chunk u, lane = column 64 u + lane
        for wv in range(4):
            for u in range((ncols + 63) // 64):
                for grp in WRITE_GROUPS:
                    seen = set()
                    for lane in grp:
                        col = 64 * u + lane
                        if col >= ncols:
                            continue
                        b = banks(image_offset16(0, col, wv, ncols), 32)
                        assert not (seen & b), ("write conflict", ncols, wv, u, grp)
                        seen |= b
        # reader: tile t, lane (m, kq) reads column 16 t + m, octet kq, through base + immediate offsets
        for t in range(ncols // 16):
            for level in range(3):
                for grp in READ_GROUPS:
                    seen = set()
                    for lane in grp:
                        m, kq = lane & 15, lane >> 4
                        lane_base = m * 4 + (kq ^ ((m >> 1) & 3))        # what the kernel keeps per lane
                        off = lane_base + (level * ncols + 16 * t) * 4   # + the tile's compile-time offset
                        assert off == image_offset16(level, 16 * t + m, kq, ncols)
                        b = banks(off, 64)
                        assert not (seen & b), ("read conflict", ncols, t, grp)
                        seen |= b
    # the layout that looks natural, (column >> 2) & 3, is a 2-way conflict on the reads (what the first version had)
    bad = 0
    for grp in READ_GROUPS:
        seen = []
        for lane in grp:
            m, kq = lane & 15, lane >> 4
            seen.append(frozenset(banks(m * 4 + (kq ^ ((m >> 2) & 3)), 64)))
        bad += len(seen) - len(set(seen))
    assert bad > 0
