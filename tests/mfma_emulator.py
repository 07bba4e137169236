"""NumPy emulation of what the HIP LSTM tile code does with a ``wpack`` buffer.

It restates, lane by lane, the data flow of ``lstm_tile_step`` in
open_l2o_amd/csrc/l2o_common.h under the documented v_mfma_f32_16x16x4_f32 operand
layout (A: lane l holds A[i=l&15][k=l>>4]; B: lane l holds B[k=l>>4][j=l&15];
D: lane l register r holds D[row=4*(l>>4)+r][col=l&15]).  It lets the CPU test-suite
validate the host weight packer, the gate-row permutation and the packed state
layout without a GPU.
"""
import numpy as np

KNT, KH = 5, 20


def wp_rows(pre):
    ks1 = 10 if pre == 2 else 6
    a1 = 0
    b1 = ks1 * KNT
    a2 = b1 + 4 * KNT
    b2 = a2 + 10 * KNT
    wl = b2 + 4 * KNT
    bl = wl + KNT
    fc = bl + 1
    return dict(ks1=ks1, a1=a1, b1=b1, a2=a2, b2=b2, wl=wl, bl=bl, fc=fc, total=fc + 3 * KNT)


def mfma16(a, b, c):
    """a, b: [64] per-lane operands; c: [64, 4] accumulators."""
    lanes = np.arange(64)
    A = np.zeros((16, 4), np.float64)
    Bm = np.zeros((4, 16), np.float64)
    A[lanes & 15, lanes >> 4] = a
    Bm[lanes >> 4, lanes & 15] = b
    Dm = A @ Bm
    out = c.astype(np.float64).copy()
    for r in range(4):
        out[:, r] += Dm[4 * (lanes >> 4) + r, lanes & 15]
    return out


def sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def gates(acc, c):
    cn = np.empty_like(c)
    hn = np.empty_like(c)
    for t in range(KNT):
        gi, gj, gf, go = acc[t][:, 0], acc[t][:, 1], acc[t][:, 2], acc[t][:, 3]
        cn[:, t] = sig(gf + 1.0) * c[:, t] + sig(gi) * np.tanh(gj)
        hn[:, t] = np.tanh(cn[:, t]) * sig(go)
    return cn, hn


def tile_step(wpack, pre, h1, c1, h2, c2, in0, in1):
    """All per-lane arrays are [64, 5] (state) / [64] (inputs). Returns d[64] + new state."""
    R = wp_rows(pre)
    W = np.asarray(wpack, np.float64).reshape(-1)[: R["total"] * 64].reshape(-1, 64)
    q = np.arange(64) >> 4
    acc2 = [np.stack([W[R["b2"] + t * 4 + r] for r in range(4)], 1) for t in range(KNT)]
    for kk in range(KNT):
        for t in range(KNT):
            acc2[t] = mfma16(W[R["a2"] + (5 + kk) * KNT + t], h2[:, kk], acc2[t])
    if pre == 2:
        fc = np.stack([W[R["fc"] + KNT + t] * in1 + (W[R["fc"] + t] * in0 + W[R["fc"] + 2 * KNT + t])
                       for t in range(KNT)], 1)
        fc = np.where(fc > 0, fc, np.expm1(np.minimum(fc, 0)))
        acc1 = [np.stack([W[R["b1"] + t * 4 + r] for r in range(4)], 1) for t in range(KNT)]
        for kk in range(KNT):
            for t in range(KNT):
                acc1[t] = mfma16(W[R["a1"] + (5 + kk) * KNT + t], h1[:, kk], acc1[t])
        for kk in range(KNT):
            for t in range(KNT):
                acc1[t] = mfma16(W[R["a1"] + kk * KNT + t], fc[:, kk], acc1[t])
    else:
        acc1 = [np.zeros((64, 4)) for _ in range(KNT)]
        for kk in range(KNT):
            for t in range(KNT):
                acc1[t] = mfma16(W[R["a1"] + kk * KNT + t], h1[:, kk], acc1[t])
        bv = np.where(q == 0, in0, np.where(q == 1, in1, np.where(q == 2, 1.0, 0.0)))
        for t in range(KNT):
            acc1[t] = mfma16(W[R["a1"] + 5 * KNT + t], bv, acc1[t])
    c1n, h1n = gates(acc1, c1)
    for kk in range(KNT):
        for t in range(KNT):
            acc2[t] = mfma16(W[R["a2"] + kk * KNT + t], h1n[:, kk], acc2[t])
    c2n, h2n = gates(acc2, c2)
    d = np.zeros(64)
    for t in range(KNT):
        d += h2n[:, t] * W[R["wl"] + t]
    lanes = np.arange(64)
    d = d + d[lanes ^ 16]
    d = d + d[lanes ^ 32]
    return d + W[R["bl"]], h1n, c1n, h2n, c2n


def ref_to_lanes(arr_ref, tile_coords):
    """[N,20] reference-layout state rows of the 16 coordinates of a tile -> [64,5]."""
    out = np.zeros((64, KNT))
    for l in range(64):
        c, q = l & 15, l >> 4
        if tile_coords[c] is None:
            continue
        for t in range(KNT):
            out[l, t] = arr_ref[tile_coords[c], 4 * t + q]
    return out


def lanes_to_ref(lane_arr, arr_ref, tile_coords):
    for l in range(64):
        c, q = l & 15, l >> 4
        if tile_coords[c] is None:
            continue
        for t in range(KNT):
            arr_ref[tile_coords[c], 4 * t + q] = lane_arr[l, t]


def pack_state_numpy(h1, c1, h2, c2, B, D):
    """Documented packed layout: tile-major, per tile [5][64][4] floats, element
    e = 4*j + w of a lane is array e // 5 (h1,c1,h2,c2), slice e % 5 (unit 4*slice + q)."""
    tpp = (D + 15) // 16
    st = np.zeros((B * tpp, 5, 64, 4), np.float32)
    refs = [h1, c1, h2, c2]
    for b in range(B):
        for tw in range(tpp):
            for l in range(64):
                c, q = l & 15, l >> 4
                j = tw * 16 + c
                if j >= D:
                    continue
                for e in range(20):
                    a, t = e // 5, e % 5
                    st[b * tpp + tw, e // 4, l, e % 4] = refs[a][b * D + j, 4 * t + q]
    return st.reshape(-1)


# ---------------------------------------------------------------------------
# bf16x3 form (open_l2o_amd/csrc/l2o_lstm_bx3.h): v_mfma_f32_16x16x32_bf16 with
# A: lane l holds A[m = l&15][k = 8*(l>>4) + i], B: lane l holds B[k = 8*(l>>4) + i][n = l&15],
# i = 0..7 packed two bf16 per dword (slot 2j in the low half of dword j).
# ---------------------------------------------------------------------------
CH_L1H, CH_L2A, CH_L2B, CH_L1X = 0, 1, 2, 3
L2E = 1.4426950408889634


def bx_nchunks(pre):
    return 4 if pre == 2 else 3


def bx_packed_default(pre):
    return pre != 2


def bx_packed_words(pre):
    return bx_nchunks(pre) * KNT * 4 * 256       # (round 4: RNNProp's wpack carries the packed section too, for k_unroll_lds)


def bx_level_words(pre):
    return bx_nchunks(pre) * KNT * 3 * 256


def bx_win_words(pre):
    return 0 if pre == 2 else 2 * KNT * 256


BX_BIAS_WORDS = 2 * KNT * 4 * 4      # fp32 accumulator inits [layer][M-tile][lane group][gate], l2o::bx::bias_off


def bx_words(pre):
    return bx_packed_words(pre) + bx_level_words(pre) + bx_win_words(pre) + BX_BIAS_WORDS


def slot_desc(j, r, h):
    """(unit 0..4 | 5 = bias, activation split level, weight split level) of K-slot h of register r of packed
    MFMA j -- the table of l2o::bx::slot_desc (csrc/l2o_lstm_bx3.h), restated."""
    pair = lambda first, x, w: (first + h, x, w)
    table = {
        0: pair(0, 0, 0), 1: pair(2, 0, 0), 2: (4, 0, h), 3: (5, 0, h),
        4: pair(0, 0, 1), 5: pair(2, 0, 1), 6: pair(0, 0, 2), 7: pair(2, 0, 2),
        8: pair(0, 1, 0), 9: pair(2, 1, 0), 10: (4, h, 0 if h else 2), 11: pair(0, 1, 1),
        12: pair(2, 1, 1), 13: pair(0, 2, 0), 14: pair(2, 2, 0), 15: (4, 1 + h, 0 if h else 1),
    }
    return table[j * 4 + r]


def bias_level(kq, h):
    return h if kq == 0 else (2 if (kq == 1 and h == 0) else -1)


def bf16_rne(x):
    """float32 array -> nearest-even bf16, returned as float32 values."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    x1 = bf16_rne(x)
    r = (x - x1).astype(np.float32)
    x2 = bf16_rne(r)
    r2 = (r - x2).astype(np.float32)
    return x1, x2, bf16_rne(r2)


def _unpack_frag(words):
    """[64, 4] uint32 -> [64, 8] float values of the bf16 slots."""
    out = np.zeros((64, 8), np.float64)
    for j in range(4):
        lo = ((words[:, j] & 0xFFFF).astype(np.uint32) << 16).view(np.float32)
        hi = (words[:, j] & 0xFFFF0000).astype(np.uint32).view(np.float32)
        out[:, 2 * j], out[:, 2 * j + 1] = lo, hi
    return out


def mfma_bf16(a_slots, b_slots, c):
    """a_slots, b_slots: [64, 8] per-lane operand values; c: [64, 4]."""
    lanes = np.arange(64)
    A = np.zeros((16, 32))
    Bm = np.zeros((32, 16))
    for i in range(8):
        A[lanes & 15, 8 * (lanes >> 4) + i] = a_slots[:, i]
        Bm[8 * (lanes >> 4) + i, lanes & 15] = b_slots[:, i]
    Dm = A @ Bm
    out = c.astype(np.float64).copy()
    for r in range(4):
        out[:, r] += Dm[4 * (lanes >> 4) + r, lanes & 15]
    return out


def _bop(v5, with_one):
    """the B operand slots of a 5-value vector per lane ([64,5]) for the three split levels."""
    q = np.arange(64) >> 4
    levels = split3(v5)
    out = []
    for li, lv in enumerate(levels):
        s = np.zeros((64, 8))
        s[:, :5] = lv
        if li == 0 and with_one:
            s[:, 7] = np.where(q == 0, 1.0, 0.0)
        out.append(s)
    return out


def _bop_packed(v5, with_one):
    """the B operand slots of the four packed MFMAs of a chunk."""
    q = np.arange(64) >> 4
    levels = split3(v5)
    out = []
    for j in range(4):
        s = np.zeros((64, 8))
        for r in range(4):
            for h in range(2):
                unit, xl, _ = slot_desc(j, r, h)
                if unit < 5:
                    s[:, 2 * r + h] = levels[xl][:, unit]
                elif with_one:
                    s[:, 2 * r + h] = [1.0 if bias_level(int(k), h) >= 0 else 0.0 for k in q]
        out.append(s)
    return out


PRODUCTS = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (2, 0)]     # (x level, w level)


def gates_scaled(acc, c):
    """acc = [-log2e*i, 2log2e*j, -log2e*(f+1), -log2e*o] -> (c', h')."""
    cn = np.empty_like(c)
    hn = np.empty_like(c)
    for t in range(KNT):
        e_i, e_f, e_o = np.exp2(acc[t][:, 0]), np.exp2(acc[t][:, 2]), np.exp2(acc[t][:, 3])
        tj = np.tanh(acc[t][:, 1] / (2 * L2E))
        cn[:, t] = c[:, t] / (1 + e_f) + tj / (1 + e_i)
        hn[:, t] = np.tanh(cn[:, t]) / (1 + e_o)
    return cn, hn


def tile_step_bx3(wpack, pre, h1, c1, h2, c2, in0, in1, packed=None):
    """The data flow of l2o::bx::tile_step on the packed weights (float64 accumulation); packed: the 4-MFMA form
    (default for the DM nets) or the one-MFMA-per-product form."""
    if packed is None:
        packed = bx_packed_default(pre)
    R = wp_rows(pre)
    wp32 = np.ascontiguousarray(wpack, np.float32)
    W = wp32.astype(np.float64).reshape(-1)[: R["total"] * 64].reshape(-1, 64)
    U = wp32.view(np.uint32)
    base = R["total"] * 64
    nf = 4 if packed else 3
    fbase = base + (0 if packed else bx_packed_words(pre))
    frag = lambda ch, t, s: _unpack_frag(U[fbase + ((ch * KNT + t) * nf + s) * 256:][:256].reshape(64, 4))
    win_off = base + bx_packed_words(pre) + bx_level_words(pre)

    def chunk(ch, vec, with_one, acc):
        if packed:
            b = _bop_packed(vec, with_one)
            for j in range(4):
                for t in range(KNT):
                    acc[t] = mfma_bf16(frag(ch, t, j), b[j], acc[t])
            return acc
        b = _bop(vec, with_one)
        for (xl, wl) in PRODUCTS:
            for t in range(KNT):
                acc[t] = mfma_bf16(frag(ch, t, wl), b[xl], acc[t])
        return acc

    # the accumulators start from the layer's pre-scaled bias (fp32 table in wpack, read per lane group), the fragments'
    # bias slots are zero (round 3: the bias left the truncating in-group sums of the bf16 matrix pipe)
    bias_off = win_off + bx_win_words(pre)
    q = np.arange(64) >> 4
    btab = wp32[bias_off:bias_off + BX_BIAS_WORDS].astype(np.float64).reshape(2, KNT, 4, 4)
    binit = lambda layer: [btab[layer, t][q] for t in range(KNT)]
    acc2 = chunk(CH_L2B, h2, True, binit(1))
    acc1 = chunk(CH_L1H, h1, True, binit(0))
    if pre == 2:
        fc = np.stack([W[R["fc"] + KNT + t] * in1 + (W[R["fc"] + t] * in0 + W[R["fc"] + 2 * KNT + t])
                       for t in range(KNT)], 1)
        fc = np.where(fc > 0, fc, np.expm1(np.minimum(fc, 0)))
        acc1 = chunk(CH_L1X, fc, False, acc1)
    else:
        for t in range(KNT):
            w0 = wp32[win_off + t * 256:][:256].reshape(64, 4).astype(np.float64)
            acc1[t] = acc1[t] + w0 * np.asarray(in0)[:, None]
            if pre == 1:
                w1 = wp32[win_off + (KNT + t) * 256:][:256].reshape(64, 4).astype(np.float64)
                acc1[t] = acc1[t] + w1 * np.asarray(in1)[:, None]
    c1n, h1n = gates_scaled(acc1, c1)
    acc2 = chunk(CH_L2A, h1n, True, acc2)
    c2n, h2n = gates_scaled(acc2, c2)
    d = np.zeros(64)
    for t in range(KNT):
        d += h2n[:, t] * W[R["wl"] + t]
    lanes = np.arange(64)
    d = d + d[lanes ^ 16]
    d = d + d[lanes ^ 32]
    return d + W[R["bl"]], h1n, c1n, h2n, c2n


# ---- the transposed products of the BPTT step (csrc/l2o_bwd_mfma.h) -------------------------
def bxb_tiles1(pre):
    return 3 if pre == 2 else 2


def bxb_words(pre):
    return (3 + bxb_tiles1(pre)) * 4 * 3 * 256


def tgemm_bx3(wpack, pre, layer, dz):
    """The data flow of l2o::bxb::tgemm on the packed weights.  dz: four [64, 5] lane arrays (gate
    types i, j, f, o; lane (c, q) holds units 4i + q).  Returns (first, second): [64, 5] lane arrays of
    d(first 20 input rows) and d(second 20 input rows) of the layer's W (first is None for the DM
    layer 1, whose gradient w.r.t. the 1-2 input features is not needed)."""
    R = wp_rows(pre)
    U = np.ascontiguousarray(wpack, np.float32).view(np.uint32)
    base = R["total"] * 64 + bx_words(pre)
    tile0, nt = (0, 3) if layer == 2 else (3, bxb_tiles1(pre))
    frag = lambda m, r, s: _unpack_frag(U[base + (((tile0 + m) * 4 + r) * 3 + s) * 256:][:256].reshape(64, 4))
    acc = [np.zeros((64, 4)) for _ in range(nt)]
    for r in range(4):
        b = _bop(dz[r], False)
        for (xl, wl) in PRODUCTS:
            for m in range(nt):
                acc[m] = mfma_bf16(frag(m, r, wl), b[xl], acc[m])
    if nt == 3:
        first = np.concatenate([acc[0], acc[2][:, 0:1]], 1)
        second = np.concatenate([acc[1], acc[2][:, 1:2]], 1)
        return first, second
    return None, np.concatenate([acc[0], acc[1][:, 0:1]], 1)
