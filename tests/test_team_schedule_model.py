"""CPU model of the multi-CU coordinate-descent team's hand-off schedule (csrc/cd_team.hip, section "multi-CU team").

The kernels split one serial recurrence over workgroups that see each other's results several blocks late.  What makes
that exact is pure bookkeeping: every entry of H = Q w must receive the SAME fused multiply-adds in the SAME order as in the
sequential recurrence (sklearn's enet_coordinate_descent_gram, _cd_fast.pyx:644-682) whoever applies them --
  remote keepers   blocks < v - LAG         (their image v - LAG is what the extractor posts for block v),
  gatherer wave    blocks v - LAG .. v - 2   (couplings Q[ii_a(v - l), ii_j(v)] staged by the stager wave),
  chain wave       block v - 1 and the steps of block v before the coordinate's own.
This test replays that schedule in plain Python (libm's fma through ctypes) on a small problem and checks that w and the final
H are bit-identical to the sequential recurrence, for several lags including the shipped one and for the first blocks
(v < LAG), where the base is the fit's initial image.  It does not touch the GPU library."""
import ctypes
import ctypes.util

import numpy as np
import pytest

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3
fma = _libm.fma
B = 8   # coordinate steps per block, as in the kernels


def _problem(c, seed=5):
    rs = np.random.RandomState(seed)
    Z = rs.randn(6 * c, c) * (0.3 + rs.rand(c))
    y = Z @ np.where(rs.rand(c) < 0.4, rs.randn(c), 0.0) + 0.1 * rs.randn(6 * c)
    return np.ascontiguousarray(Z.T @ Z), Z.T @ y


def _soft(q_i, Qd, alpha, w_old, H_i):
    Hp = fma(-w_old, Qd, H_i)          # H without the coordinate's own contribution
    tmp = q_i - Hp
    return float(np.copysign(max(abs(tmp) - alpha, 0.0) / Qd, tmp))


def _sequential(Q, q, alpha, order):
    c = len(q)
    w, H = np.zeros(c), np.zeros(c)
    for i in order:
        wo = w[i]
        wn = _soft(q[i], Q[i, i], alpha, wo, H[i])
        for j in range(c):               # two fma per entry, in sklearn's order: remove the old, add the new
            H[j] = fma(wn, Q[i, j], fma(-wo, Q[i, j], H[j]))
        w[i] = wn
    return w, H


def _scheduled(Q, q, alpha, order, lag, slice_cols):
    """The same recurrence with the H entries of every `slice_cols` columns kept by a 'remote' that is `lag` blocks behind."""
    c = len(q)
    n_blocks = len(order) // B
    w = np.zeros(c)
    images = [np.zeros(c)]               # images[t] = H after the first t blocks (what the keepers hold / publish)
    published = []                       # per block: [(i, w_old, w_new)] in step order (the chain wave's publication)

    def apply_block(H, blk):
        for (i, wo, wn) in published[blk]:
            for j in range(c):
                H[j] = fma(wn, Q[i, j], fma(-wo, Q[i, j], H[j]))

    for v in range(n_blocks):
        idx = order[B * v:B * v + B]
        # extractor: the value of each coordinate of block v out of its owner's image v - lag (image 0 for the first blocks);
        # the owner only has the blocks < v - lag applied, which is all `images` may be asked for here
        base_image = images[max(v - lag, 0)]
        this_block = []
        for a, i in enumerate(idx):
            owner = i // slice_cols      # (every remote posts a record; the gatherer takes each value from its owner)
            assert owner * slice_cols <= i < (owner + 1) * slice_cols
            Hv = base_image[i]
            # gatherer wave: blocks v - lag .. v - 2; chain wave: block v - 1
            for u in range(max(v - lag, 0), v):
                for (i2, wo, wn) in published[u]:
                    Hv = fma(wn, Q[i2, i], fma(-wo, Q[i2, i], Hv))
            # chain wave: the steps of this block before the coordinate's own (couplings qc)
            for (i2, wo, wn) in this_block:
                Hv = fma(wn, Q[i2, i], fma(-wo, Q[i2, i], Hv))
            wo = w[i]
            wn = _soft(q[i], Q[i, i], alpha, wo, Hv)
            w[i] = wn
            this_block.append((i, wo, wn))
        published.append(this_block)
        # keepers (any number of blocks later): image v + 1 = image v with block v applied
        nxt = images[v].copy()
        apply_block(nxt, v)
        images.append(nxt)
    return w, images[-1]


@pytest.mark.parametrize("lag", [1, 3, 5])
def test_lagged_schedule_is_bit_identical_to_the_sequential_recurrence(lag):
    c = 48
    Q, q = _problem(c)
    alpha = 0.05 * np.abs(q).max()
    rs = np.random.RandomState(99)
    # 5 epochs' worth of coordinates, drawn with replacement as rand_int does (repeats inside and across blocks included)
    order = [int(i) for i in rs.randint(0, c, size=5 * c - (5 * c) % B)]
    w_ref, H_ref = _sequential(Q, q, alpha, order)
    w, H = _scheduled(Q, q, alpha, order, lag, slice_cols=16)
    assert np.array_equal(w, w_ref) and np.array_equal(H, H_ref)
    assert 0 < int(np.sum(w_ref != 0)) < c      # a sparse solution: both no-op and moving steps occurred
