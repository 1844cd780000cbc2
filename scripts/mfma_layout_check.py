"""Numpy emulation of v_mfma_f32_32x32x2_f32 lane semantics (cdna_hip_programming.md 3:
A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]) used to
validate the register-chained MLP layout of lp_renderer_mfma.hip before it ever runs on a GPU.
Run: python scripts/mfma_layout_check.py"""
import numpy as np

L = np.arange(64)
H = L >> 5
R = L & 31


def feat(q, h):
    return (q & 3) + 8 * (q >> 2) + 4 * h


def mfma(a, b, acc):
    """a, b: [64] per-lane operands; acc: [16, 64] per-lane accumulators -> new acc."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[R, H] = a; B[H, R] = b
    Dm = A @ B  # [32, 32]
    out = acc.copy()
    for q in range(16):
        out[q] += Dm[feat(q, H), R]
    return out


rng = np.random.default_rng(0)
C, Hd, NR = 16, 32, 32
X = rng.standard_normal((NR, C)); W1 = rng.standard_normal((C, Hd)); b1 = rng.standard_normal(Hd)
W2 = rng.standard_normal((Hd, Hd)); b2 = rng.standard_normal(Hd)

# forward: lane (ray r, half h) holds x0[kk] = X[r][feat(kk,h)], kk < C/2
x0 = np.stack([X[R, feat(kk, H)] for kk in range(C // 2)])
acc = np.stack([b1[feat(q, H)] for q in range(16)])
for kk in range(C // 2):
    acc = mfma(W1[feat(kk, H), R], x0[kk], acc)      # A = W1[feat(kk,h)][i=l&31], B = activation
h1 = np.maximum(acc, 0)
ref_h1 = np.maximum(X @ W1 + b1, 0)
assert np.allclose(h1, np.stack([ref_h1[R, feat(q, H)] for q in range(16)])), "layer 1"
acc = np.stack([b2[feat(q, H)] for q in range(16)])
for kk in range(16):
    acc = mfma(W2[feat(kk, H), R], h1[kk], acc)
ref_y2 = ref_h1 @ W2 + b2
assert np.allclose(acc, np.stack([ref_y2[R, feat(q, H)] for q in range(16)])), "layer 2"

# backward dX^T = W . dY^T : A = W[i=l&31][feat(kk,h)], B = dY[kk]
dY = rng.standard_normal((NR, Hd))
dy = np.stack([dY[R, feat(q, H)] for q in range(16)])
acc = np.zeros((16, 64))
for kk in range(16):
    acc = mfma(W2[R, feat(kk, H)], dy[kk], acc)
ref_dx = dY @ W2.T
assert np.allclose(acc, np.stack([ref_dx[R, feat(q, H)] for q in range(16)])), "dX"
# first layer (16 inputs padded to 32 rows): rows >= C are zero weights
acc = np.zeros((16, 64))
W1p = np.zeros((32, Hd)); W1p[:C] = W1
for kk in range(16):
    acc = mfma(W1p[R, feat(kk, H)], dy[kk], acc)
ref_dx0 = dY @ W1.T
assert np.allclose(acc[:8], np.stack([ref_dx0[R, feat(q, H)] for q in range(8)])), "dX0"

# dW = X^T dY through LDS tiles T[ray][33]: A = TX[2kk+h][l&31], B = TY[2kk+h][l&31]
TX = np.zeros((32, 33)); TY = np.zeros((32, 33))
for q in range(16):
    TX[R, feat(q, H)] = h1[q]; TY[R, feat(q, H)] = dy[q]
acc = np.zeros((16, 64))
for kk in range(16):
    acc = mfma(TX[2 * kk + H, R], TY[2 * kk + H, R], acc)
ref_dW = ref_h1.T @ dY
assert np.allclose(acc, np.stack([ref_dW[feat(q, H), R] for q in range(16)])), "dW"   # lane: i=feat(q,h), j=l&31
print("MFMA layout algebra OK")
