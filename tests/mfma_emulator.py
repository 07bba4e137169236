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
    W = np.asarray(wpack, np.float64).reshape(-1, 64)
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
