"""CPU emulation of the wave-level algorithm of csrc/sa_mlp.hip.

No GPU is available when the kernels are written, so the register-chaining scheme (which k-step
consumes which accumulator register, how the weight stream is ordered, the flipped last layer and
the in-lane max) is restated here in numpy on top of an emulated ``v_mfma_f32_32x32x2_f32`` with
the operand layouts documented for gfx950:

    A[i][k] is supplied by lane i + 32k;   B[k][j] by lane j + 32k;
    D[row][col]: lane = col + 32*((row>>2)&1), register = (row&3) + 4*(row>>3).

If this test passes and the HIP kernel disagrees with the oracle on the GPU, the bug is in the
kernel's transcription, not in the scheme.
"""
import numpy as np
import pytest

PAD = -1


def mfma_32x32x2(a, b, c):
    """a[64], b[64], c[64,16] -> d[64,16]  (float64 emulation of the f32 instruction)."""
    A = np.stack([a[:32], a[32:]], axis=1)  # [i][k]
    Bm = np.stack([b[:32], b[32:]], axis=0)  # [k][j]
    D = A @ Bm  # [32 rows][32 cols]
    d = c.copy()
    for lane in range(64):
        col, h = lane & 31, lane >> 5
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * h
            d[lane, r] += D[row, col]
    return d


class Cfg:
    def __init__(self, CF, C1, C2, C3):
        self.CF, self.C1, self.C2, self.C3 = CF, C1, C2, C3
        self.CIN = 3 + CF
        self.KS0 = 2 if CF == 1 else 2 + CF // 2
        self.KS1, self.KS2 = C1 // 2, C2 // 2
        self.OT1, self.OT2, self.OT3 = C1 // 32, C2 // 32, C3 // 32

    def chan0(self, t, h):
        if self.CF == 1:
            return 2 * t + h
        if t == 0:
            return h
        if t == 1:
            return PAD if h else 2
        return 3 + h * (self.CF // 2) + (t - 2)

    @staticmethod
    def chan_tile(t, h):
        it, r = t >> 4, t & 15
        return it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h


def weight_operand(W, ot, in_ch_of_half):
    """one float per lane for one MFMA step: W[ot*32 + (lane&31)][channel of this lane's half]"""
    v = np.zeros(64)
    for lane in range(64):
        ch = in_ch_of_half(lane >> 5)
        v[lane] = 0.0 if ch == PAD else W[ot * 32 + (lane & 31), ch]
    return v


def bias_tile(bias, ot):
    c = np.zeros((64, 16))
    for lane in range(64):
        for r in range(16):
            c[lane, r] = bias[ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)]
    return c


def wave_sa_mlp(cfg, X, W1, b1, W2, b2, W3, b3):
    """X [nsample, CIN] grouped inputs of ONE query -> pooled [C3], following the kernel."""
    ns = X.shape[0]
    omax = np.full((cfg.OT3, 64), -np.inf)
    for rt in range(0, ns, 32):
        pts = X[rt:rt + 32]
        # layer-1 B operands: lane (col, half) holds its point's channel chan0(t, half)
        x0 = np.zeros((cfg.KS0, 64))
        for t in range(cfg.KS0):
            for lane in range(64):
                ch = cfg.chan0(t, lane >> 5)
                x0[t, lane] = 0.0 if ch == PAD else pts[lane & 31, ch]
        a1 = [bias_tile(b1, ot) for ot in range(cfg.OT1)]
        for s in range(cfg.KS0 * cfg.OT1):
            t, ot = divmod(s, cfg.OT1)
            a1[ot] = mfma_32x32x2(weight_operand(W1, ot, lambda h: cfg.chan0(t, h)), x0[t], a1[ot])
        a1 = [np.maximum(a, 0) for a in a1]
        a2 = [bias_tile(b2, ot) for ot in range(cfg.OT2)]
        for s in range(cfg.KS1 * cfg.OT2):
            t, ot = divmod(s, cfg.OT2)
            a2[ot] = mfma_32x32x2(weight_operand(W2, ot, lambda h: cfg.chan_tile(t, h)), a1[t >> 4][:, t & 15], a2[ot])
        a2 = [np.maximum(a, 0) for a in a2]
        for ot in range(cfg.OT3):
            a3 = np.zeros((64, 16))
            for t in range(cfg.KS2):
                a3 = mfma_32x32x2(a2[t >> 4][:, t & 15], weight_operand(W3, ot, lambda h: cfg.chan_tile(t, h)), a3)
            omax[ot] = np.maximum(omax[ot], a3.max(axis=1))
    out = np.zeros(cfg.C3)
    for ot in range(cfg.OT3):
        v = np.maximum(omax[ot][:32], omax[ot][32:])
        out[ot * 32:(ot + 1) * 32] = np.maximum(v + b3[ot * 32:(ot + 1) * 32], 0)
    return out


@pytest.mark.parametrize("shape", [(1, 64, 64, 64), (64, 128, 128, 256)])
def test_register_chain_equals_plain_mlp(shape):
    cfg = Cfg(*shape)
    rng = np.random.default_rng(sum(shape))
    ns = 64
    X = rng.normal(size=(ns, cfg.CIN))
    W1, b1 = rng.normal(size=(cfg.C1, cfg.CIN)) * 0.3, rng.normal(size=cfg.C1) * 0.1
    W2, b2 = rng.normal(size=(cfg.C2, cfg.C1)) * 0.1, rng.normal(size=cfg.C2) * 0.1
    W3, b3 = rng.normal(size=(cfg.C3, cfg.C2)) * 0.1, rng.normal(size=cfg.C3) * 0.1
    got = wave_sa_mlp(cfg, X, W1, b1, W2, b2, W3, b3)
    h = np.maximum(X @ W1.T + b1, 0)
    h = np.maximum(h @ W2.T + b2, 0)
    h = np.maximum(h @ W3.T + b3, 0)
    np.testing.assert_allclose(got, h.max(axis=0), rtol=1e-10, atol=1e-10)


def test_gemm_k_permutation():
    """csrc/dense.hip: within a 16-wide K slab lane-half h consumes k = 8h + 4v + u at step (v,u)."""
    rng = np.random.default_rng(3)
    X, W = rng.normal(size=(32, 16)), rng.normal(size=(32, 16))
    c = np.zeros((64, 16))
    for v in range(2):
        for u in range(4):
            a = np.array([X[l & 31, 8 * (l >> 5) + 4 * v + u] for l in range(64)])
            b = np.array([W[l & 31, 8 * (l >> 5) + 4 * v + u] for l in range(64)])
            c = mfma_32x32x2(a, b, c)
    Y = X @ W.T
    for lane in range(64):
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            assert abs(c[lane, r] - Y[row, lane & 31]) < 1e-12
