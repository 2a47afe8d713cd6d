"""Lane-level NumPy model of csrc/st_tail.hip (test infrastructure, CPU only).

The fused transformer-tail kernel keeps 32 tokens per wave in registers and multiplies a stream of 1-KB weight fragments
(v_mfma_f32_32x32x16 A operands) into them.  Every index map of that kernel -- which weight element sits in which lane of which
fragment, which register of an accumulator tile is which channel, how an accumulator becomes the next product's B operand, where
the 40-wide heads sit in 48-slot groups -- is restated here 1:1 (same function names as the device code) and executed on one
wave with exact fp32/fp64 arithmetic, so that tests/test_st_tail_layout_cpu.py can check the whole dataflow against the plain
formulas of BasicTransformerBlock._forward (attention.py:271-275) + proj_out (attention.py:338-340) without a GPU.
"""
import numpy as np

C = 320
NT = 10            # 32-channel tiles
KS = 20            # k16 steps over C
HEADS = 8
DH = 40
HID = 1280
CHUNK = 32         # hidden units per feed-forward chunk
NCHUNK = HID // CHUNK


def sigma(i):
    """MFMA A-row (= D-row) index i of a 32x32 tile -> channel inside the tile: lane half hh, register r hold channel 16 hh + r."""
    return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3)


def drow(r, hh):
    """D register r of lane half hh -> MFMA D-row index."""
    return (r & 3) + 8 * (r >> 2) + 4 * hh


def kmap(ks, hk, j):
    """k16 step ks, lane half hk, element j of a B fragment taken from accumulators -> channel."""
    return 32 * (ks >> 1) + 16 * hk + 8 * (ks & 1) + j


def mfma(afrag, bfrag, acc):
    """v_mfma_f32_32x32x16: afrag/bfrag [64 lanes][8], acc [64 lanes][16] (in place)."""
    A = np.zeros((32, 16), np.float64)
    B = np.zeros((16, 32), np.float64)
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = afrag[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = bfrag[l]
    D = A @ B
    for l in range(64):
        for r in range(16):
            acc[l, r] += D[drow(r, l >> 5), l & 31]


def acc_to_bfrags(acc_tile):
    """accumulator tile [64][16] -> its two k16-step B fragments [2][64][8]: step s takes registers 8s..8s+7."""
    return np.stack([acc_tile[:, 0:8], acc_tile[:, 8:16]])


# ------------------------------------------------------------------------------------------------ weight fragment streams
def frag_generic(W, orow_of_i, icol_of):
    """one A fragment: lane (i, hk), element j = W[orow(i)][icol(hk, j)] (None -> 0)."""
    f = np.zeros((64, 8), np.float64)
    for l in range(64):
        i, hk = l & 31, l >> 5
        o = orow_of_i(i)
        if o is None:
            continue
        for j in range(8):
            c = icol_of(hk, j)
            if c is not None:
                f[l, j] = W[o, c]
    return f


def q_slot(tq, i):
    """output slot (tile tq, D-row i) of the padded q projection -> (head, d) or None: head h owns k16 steps 3h..3h+2."""
    sl = sigma(i)
    hh, r = sl >> 4, sl & 15
    ksq = 2 * tq + (r >> 3)
    head, d = ksq // 3, 16 * (ksq % 3) + 8 * hh + (r & 7)
    return (head, d) if d < DH else None


def head_in(hd, ksl, hk, j):
    """k16 step ksl (0..2) of head hd, lane half hk, element j of an O fragment -> input channel of attn2.to_out or None."""
    d = 32 * (ksl >> 1) + 16 * hk + 8 * (ksl & 1) + j
    return hd * DH + d if d < DH else None


def pack_weights(Wo1, Wq, g2, Wo2, W1, g3, W2, Wp):
    """The weight stream in consumption order ([nfrag][64][8]); Wq / W1 carry the folded LayerNorm gammas."""
    fr = []
    for tn in range(NT):                                   # A: attn1.to_out
        for ks in range(KS):
            fr.append(frag_generic(Wo1, lambda i: 32 * tn + sigma(i), lambda hk, j: kmap(ks, hk, j)))
    Wq_f = Wq * g2[None, :]
    for pp in range(4):                                    # B: per head pair [q 60][to_out 60]
        for tl in range(3):
            tq = 3 * pp + tl

            def orow(i, tq=tq):
                s = q_slot(tq, i)
                return None if s is None else s[0] * DH + s[1]
            for ks in range(KS):
                fr.append(frag_generic(Wq_f, orow, lambda hk, j: kmap(ks, hk, j)))
        for hl in range(2):
            hd = 2 * pp + hl
            for tn in range(NT):
                for ksl in range(3):
                    fr.append(frag_generic(Wo2, lambda i: 32 * tn + sigma(i), lambda hk, j: head_in(hd, ksl, hk, j)))
    W1_f = W1 * g3[None, :]

    def w1_frags(cc):                                      # [x tile | gate tile] of chunk cc: 40 fragments
        out = []
        for gate in range(2):
            def orow(i, gate=gate):
                u = CHUNK * cc + sigma(i)
                return HID + u if gate else u
            for ks in range(KS):
                out.append(frag_generic(W1_f, orow, lambda hk, j: kmap(ks, hk, j)))
        return out

    def w2_frags(cc):                                      # 20 fragments: 2 k16 steps per output tile
        return [frag_generic(W2, lambda i: 32 * tn + sigma(i), lambda hk, j: CHUNK * cc + 16 * hk + 8 * ksl + j)
                for tn in range(NT) for ksl in range(2)]
    # C: feed-forward, 32 hidden units per chunk, software-pipelined: ff.net.0 of chunk c+1 streams in front of ff.net.2 of chunk c
    fr += w1_frags(0)
    for cc in range(NCHUNK - 1):
        fr += w1_frags(cc + 1)
        fr += w2_frags(cc)
    fr += w2_frags(NCHUNK - 1)
    for tn in range(NT):                                   # D: proj_out
        for ks in range(KS):
            fr.append(frag_generic(Wp, lambda i: 32 * tn + sigma(i), lambda hk, j: kmap(ks, hk, j)))
    return np.stack(fr)


def pack_kv(K2, V2, Nk):
    """context K / V of one sample ([Nk][C] each) -> per head pair 60 fragments: [K h0: 9][V h0: 12][K h1: 9][V h1: 12][pad 18]."""
    out = np.zeros((4, 60, 64, 8), np.float64)
    for pp in range(4):
        for hl in range(2):
            hd = 2 * pp + hl
            for kt in range(3):
                for ks in range(3):
                    f = out[pp, hl * 21 + kt * 3 + ks]
                    for l in range(64):
                        i, hk = l & 31, l >> 5
                        key = 32 * kt + sigma(i)
                        for j in range(8):
                            d = 16 * ks + 8 * hk + j
                            if key < Nk and d < DH:
                                f[l, j] = K2[key, hd * DH + d]
            for dt in range(2):
                for kk in range(6):
                    f = out[pp, hl * 21 + 9 + dt * 6 + kk]
                    kt, s = kk >> 1, kk & 1
                    for l in range(64):
                        i, hk = l & 31, l >> 5
                        d = 32 * dt + sigma(i)
                        for j in range(8):
                            key = 32 * kt + 16 * hk + 8 * s + j
                            if key < Nk and d < DH:
                                f[l, j] = V2[key, hd * DH + d]
    return out


def pack_vectors(bo1, Wq, b2n, bo2, b1, W1, b3n, bff2, bp):
    """fp32 vectors in the order the kernel indexes them: bo1[320] | bq[384 slots] | bo2[320] | b1'[NCHUNK*2*32] | b2[320] | bp[320]."""
    bq_full = Wq @ b2n                                      # beta of norm2 through attn2.to_q
    bq = np.zeros(384)
    for tq in range(12):
        for i in range(32):
            s = q_slot(tq, i)
            if s is not None:
                bq[32 * tq + sigma(i)] = bq_full[s[0] * DH + s[1]]
    b1_full = b1 + W1 @ b3n
    b1p = np.zeros(NCHUNK * 2 * 32)
    for cc in range(NCHUNK):
        for gate in range(2):
            for c in range(32):
                u = CHUNK * cc + c
                b1p[(cc * 2 + gate) * 32 + c] = b1_full[HID + u if gate else u]
    return dict(bo1=bo1, bq=bq, bo2=bo2, b1=b1p, b2=bff2, bp=bp)


# ------------------------------------------------------------------------------------------------ one wave of the kernel
def lane_rows_to_acc(X):
    """[32 rows][C] -> accumulator tiles [NT][64][16]: lane (row, hh), register r of tile t = X[row][32 t + 16 hh + r]."""
    acc = np.zeros((NT, 64, 16))
    for l in range(64):
        for t in range(NT):
            acc[t, l] = X[l & 31, 32 * t + 16 * (l >> 5):32 * t + 16 * (l >> 5) + 16]
    return acc


def acc_to_rows(acc):
    X = np.zeros((32, C))
    for l in range(64):
        for t in range(NT):
            X[l & 31, 32 * t + 16 * (l >> 5):32 * t + 16 * (l >> 5) + 16] = acc[t, l]
    return X


def rows_to_bfrags(X):
    """[32 rows][C] in memory order -> 20 B fragments: lane (row, hk) of step ks holds X[row][kmap(ks, hk, 0..7)]."""
    fr = np.zeros((KS, 64, 8))
    for ks in range(KS):
        for l in range(64):
            c0 = kmap(ks, l >> 5, 0)
            fr[ks, l] = X[l & 31, c0:c0 + 8]
    return fr


def layernorm_frags(acc, eps=1e-5):
    """two-pass LayerNorm statistics of each lane pair's row from the 10 accumulator tiles -> 20 B fragments of (x - mean) * rstd."""
    s = acc.sum(axis=(0, 2))                                # per lane
    s = s + np.roll(s, 32)                                  # lane ^ 32
    mean = s / C
    q = ((acc - mean[None, :, None]) ** 2).sum(axis=(0, 2))
    q = q + np.roll(q, 32)
    rstd = 1.0 / np.sqrt(q / C + eps)
    y = (acc - mean[None, :, None]) * rstd[None, :, None]
    return np.concatenate([acc_to_bfrags(y[t]) for t in range(NT)])   # [20][64][8], step 2t+s


def gelu(x):
    from math import erf
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


def run_wave(att, h, x_in, wfr, kvfr, vec, Nk, scale):
    """att/h/x_in: [32][C] rows of one wave.  Returns out [32][C].  Mirrors st_tail_kernel's order of operations."""
    pos = 0

    def take(n):
        nonlocal pos
        r = wfr[pos:pos + n]
        pos += n
        return r

    lane_hh = np.arange(64) >> 5
    # 1. h1 = att . Wo1^T + bo1 + h
    acc = lane_rows_to_acc(h + vec["bo1"][None, :])
    yf = rows_to_bfrags(att)
    for tn in range(NT):
        w = take(KS)
        for ks in range(KS):
            mfma(w[ks], yf[ks], acc[tn])
    # 2. norm2 (gamma/beta folded into the q weights / bias)
    y2 = layernorm_frags(acc)
    for pp in range(4):
        # 3. q of two heads: three 32-slot tiles
        qacc = np.zeros((3, 64, 16))
        for tl in range(3):
            tq = 3 * pp + tl
            for l in range(64):
                qacc[tl, l] = vec["bq"][32 * tq + 16 * lane_hh[l]:32 * tq + 16 * lane_hh[l] + 16]
            w = take(KS)
            for ks in range(KS):
                mfma(w[ks], y2[ks], qacc[tl])
        qf = np.concatenate([acc_to_bfrags(qacc[tl]) for tl in range(3)])     # 6 steps: head hl owns 3hl..3hl+2
        kv = kvfr[pp]
        oacc_frags = []
        for hl in range(2):
            # 4. S^T = K . q^T (keys on D rows), softmax over keys, O^T = V^T . P^T
            S = np.zeros((3, 64, 16))
            for kt in range(3):
                for ks in range(3):
                    mfma(kv[hl * 21 + kt * 3 + ks], qf[3 * hl + ks], S[kt])
            key = (32 * np.arange(3)[:, None, None] + 16 * lane_hh[None, :, None] + np.arange(16)[None, None, :])
            Sm = np.where(key < Nk, S * scale, -np.inf)
            m = Sm.max(axis=(0, 2))
            m = np.maximum(m, np.roll(m, 32))
            Pm = np.exp(Sm - m[None, :, None])
            lsum = Pm.sum(axis=(0, 2))
            lsum = lsum + np.roll(lsum, 32)
            pf = np.concatenate([acc_to_bfrags(Pm[kt]) for kt in range(3)])    # 6 key steps
            O = np.zeros((2, 64, 16))
            for dt in range(2):
                for kk in range(6):
                    mfma(kv[hl * 21 + 9 + dt * 6 + kk], pf[kk], O[dt])
            O = O / lsum[None, :, None]
            of = np.concatenate([acc_to_bfrags(O[0]), acc_to_bfrags(O[1])[:1]])  # 3 steps
            oacc_frags.append(of)
        # 5. h2 += o . Wo2^T for both heads
        for hl in range(2):
            w = take(30)
            for tn in range(NT):
                for ksl in range(3):
                    mfma(w[tn * 3 + ksl], oacc_frags[hl][ksl], acc[tn])
    for l in range(64):
        for t in range(NT):
            acc[t, l] += vec["bo2"][32 * t + 16 * lane_hh[l]:32 * t + 16 * lane_hh[l] + 16]
    # 6. norm3 -> feed-forward in chunks of 64 hidden units
    y3 = layernorm_frags(acc)
    for t in range(NT):
        for l in range(64):
            acc[t, l] += vec["b2"][32 * t + 16 * lane_hh[l]:32 * t + 16 * lane_hh[l] + 16]
    def gemm1(cc):
        a1 = np.zeros((2, 64, 16))
        for ti in range(2):
            for l in range(64):
                a1[ti, l] = vec["b1"][(cc * 2 + ti) * 32 + 16 * lane_hh[l]:(cc * 2 + ti) * 32 + 16 * lane_hh[l] + 16]
            w = take(KS)
            for ks in range(KS):
                mfma(w[ks], y3[ks], a1[ti])
        return a1

    def gemm2(a1):
        gf = acc_to_bfrags(a1[0] * gelu(a1[1]))                                  # 2 steps
        w = take(20)
        for tn in range(NT):
            for ksl in range(2):
                mfma(w[tn * 2 + ksl], gf[ksl], acc[tn])
    a1 = gemm1(0)
    for cc in range(NCHUNK - 1):
        a1n = gemm1(cc + 1)
        gemm2(a1)
        a1 = a1n
    gemm2(a1)
    # 9. out = h3 . Wp^T + bp + x_in
    hf = np.concatenate([acc_to_bfrags(acc[t]) for t in range(NT)])
    out = lane_rows_to_acc(x_in + vec["bp"][None, :])
    for tn in range(NT):
        w = take(KS)
        for ks in range(KS):
            mfma(w[ks], hf[ks], out[tn])
    assert pos == len(wfr)
    return acc_to_rows(out)
