"""Index maps of the fused transformer-tail kernel (csrc/st_tail.hip), checked on the CPU: the lane-level model of one wave
(tests/st_tail_emul.py: fragment packing, accumulator -> operand conversions, head slots, key masks) must reproduce
BasicTransformerBlock._forward's tail + proj_out (attention.py:271-275, :338-340) computed with plain matrix formulas."""
import numpy as np

from tests import st_tail_emul as EM


def _ln(x, g, b, eps=1e-5):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + eps) * g + b


def test_wave_model_matches_plain_formulas():
    rng = np.random.default_rng(7)
    C, HID, Nk = EM.C, EM.HID, 77
    r = lambda *s: rng.standard_normal(s)
    att, h, x_in = r(32, C), r(32, C), r(32, C)
    Wo1, bo1 = r(C, C) / 18, r(C) * 0.1
    g2, b2n, g3, b3n = 1 + 0.1 * r(C), 0.1 * r(C), 1 + 0.1 * r(C), 0.1 * r(C)
    Wq, Wo2, bo2 = r(C, C) / 18, r(C, C) / 18, r(C) * 0.1
    K2, V2 = r(Nk, C), r(Nk, C)
    W1, b1 = r(2 * HID, C) / 18, r(2 * HID) * 0.1
    W2, bff2 = r(C, HID) / 36, r(C) * 0.1
    Wp, bp = r(C, C) / 18, r(C) * 0.1
    scale = EM.DH ** -0.5
    # plain formulas
    h1 = att @ Wo1.T + bo1 + h
    q = _ln(h1, g2, b2n) @ Wq.T
    o = np.zeros_like(q)
    for hd in range(EM.HEADS):
        sl = slice(hd * EM.DH, (hd + 1) * EM.DH)
        s = q[:, sl] @ K2[:, sl].T * scale
        p = np.exp(s - s.max(-1, keepdims=True))
        o[:, sl] = (p / p.sum(-1, keepdims=True)) @ V2[:, sl]
    h2 = o @ Wo2.T + bo2 + h1
    u = _ln(h2, g3, b3n) @ W1.T + b1
    h3 = (u[:, :HID] * EM.gelu(u[:, HID:])) @ W2.T + bff2 + h2
    ref = h3 @ Wp.T + bp + x_in
    # the kernel's dataflow
    wfr = EM.pack_weights(Wo1, Wq, g2, Wo2, W1, g3, W2, Wp)
    assert wfr.shape == (3280, 64, 8)
    kvfr = EM.pack_kv(K2, V2, Nk)
    vec = EM.pack_vectors(bo1, Wq, b2n, bo2, b1, W1, b3n, bff2, bp)
    out = EM.run_wave(att, h, x_in, wfr, kvfr, vec, Nk, scale)
    assert np.abs(out - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
