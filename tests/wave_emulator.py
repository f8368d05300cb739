"""Lane-level numpy emulation of one wave of csrc/lstm.hip.

It executes the kernel's algorithm — record stream walk, v_mfma_f32_16x16x4_f32
operand/result lane maps, D-fragment reuse as next step's B operand — on the
stream produced by the real host packer (fnssl_lstm_pack, which needs no GPU).
Used by the CPU test-suite to check the packing / K-permutation logic against
the oracle before anything runs on a device.  Test infrastructure only.

MFMA 16x16x4 (f32) lane maps (guide: cdna_hip_programming.md §3):
    A[i][k]: lane l = i + 16*k          B[k][j]: lane l = j + 16*k
    D[row][col]: lane l = col + 16*g, register r, row = 4*g + r
"""
import numpy as np

F32 = np.float32
LANE = np.arange(64)
N_OF = LANE & 15
G_OF = LANE >> 4


def mfma_16x16x4(a_lane, b_lane, acc):
    """acc [64, 4] += A*B with a_lane/b_lane the per-lane operand registers [64]."""
    A = a_lane.reshape(4, 16).T          # A[i][k] = a_lane[i + 16k]
    B = b_lane.reshape(4, 16)            # B[k][j] = b_lane[j + 16k]
    D = acc.copy()
    for k in range(4):                   # k-ordered fma chain like the hardware
        prod = np.outer(A[:, k], B[k, :]).astype(F32)       # [row, col]
        # gather to lanes: lane (col=n, g), reg r -> row 4g + r
        rows = (4 * G_OF[:, None] + np.arange(4)[None, :])  # [64, 4]
        D = (D + prod[rows, N_OF[:, None]]).astype(F32)
    return D


def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def run_wave(stream, x_sum, x_cat, H, nsteps, reverse=False):
    """Emulate one wave (16 sequences).

    stream : packed floats from fnssl_lstm_pack for (c0, c2, H)
    x_sum  : [16, nsteps, c0] summed input (or None), x_cat: [16, nsteps, c2] (or None)
    returns h [16, nsteps, H]
    """
    c0 = 0 if x_sum is None else x_sum.shape[2]
    c2 = 0 if x_cat is None else x_cat.shape[2]
    NS = H // 16
    recs = stream.reshape(-1, 64, 4)     # record -> [lane][component]
    nv0, ns0 = c0 >> 4, (c0 & 15) >> 2
    nv2, ns2 = c2 >> 4, (c2 & 15) >> 2
    qps = 1 + nv0 + ns0 + nv2 + ns2 + NS
    assert recs.shape[0] == NS * qps * 4
    hold = np.zeros((NS, 64, 4), F32)    # h_{t-1} in D layout: [slice][lane][reg]
    cst = np.zeros((NS, 64, 4), F32)
    out = np.zeros((16, nsteps, H), F32)
    for step in range(nsteps):
        tt = nsteps - 1 - step if reverse else step
        new_hold = np.zeros_like(hold)
        for s in range(NS):
            quad = s * qps
            acc = [recs[quad * 4 + q].copy() for q in range(4)]   # bias records
            quad += 1

            def kstep(rec, b_lane):
                for q in range(4):
                    acc[q] = mfma_16x16x4(rec[:, q], b_lane, acc[q])

            for v in range(nv0):
                xv = x_sum[N_OF[:, None], tt, 16 * v + 4 * G_OF[:, None] + np.arange(4)[None, :]]   # [64, 4]
                for j in range(4):
                    kstep(recs[quad * 4 + j], xv[:, j])
                quad += 1
            for u in range(ns0):
                kstep(recs[quad * 4], x_sum[N_OF, tt, 16 * nv0 + 4 * u + G_OF])
                quad += 1
            for v in range(nv2):
                xv = x_cat[N_OF[:, None], tt, 16 * v + 4 * G_OF[:, None] + np.arange(4)[None, :]]
                for j in range(4):
                    kstep(recs[quad * 4 + j], xv[:, j])
                quad += 1
            for u in range(ns2):
                kstep(recs[quad * 4], x_cat[N_OF, tt, 16 * nv2 + 4 * u + G_OF])
                quad += 1
            for sp in range(NS):
                for j in range(4):
                    kstep(recs[quad * 4 + j], hold[sp][:, j])
                quad += 1
            assert quad == (s + 1) * qps
            ig, fg, og = sigmoid(acc[0]), sigmoid(acc[1]), sigmoid(acc[3])
            gg = np.tanh(acc[2], dtype=F32)
            cn = (fg * cst[s] + ig * gg).astype(F32)
            hn = (og * np.tanh(cn, dtype=F32)).astype(F32)
            cst[s] = cn
            new_hold[s] = hn
            # store: lane (n, g) reg r -> out[n, tt, 16 s + 4 g + r]
            for r in range(4):
                out[N_OF, tt, 16 * s + 4 * G_OF + r] = hn[:, r]
        hold = new_hold
    return out
