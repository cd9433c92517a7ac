// The linear task order of the persistent blocked Cholesky (chol_step.hip: k_chol_chain), in plain C++ so that a host test can
// walk it (tests/host/test_chain_order.cpp): every task must come after everything it waits for -- that is what makes the
// one-counter hand-out free of deadlock for any number of resident workgroups.
#pragma once
#if defined(__HIPCC__) || defined(__CUDACC__)
#define CP_HD __host__ __device__
#else
#define CP_HD
#endif

constexpr int CHAIN_L_MAX = 4;
constexpr int CHAIN_NTR_MAX = 32;    // right-hand-side tile columns the control block has words for (n_pad <= 4096)

struct ChainShape {
    int nblk, ntr, L;
    CP_HD int width(int i) const { return nblk - i + ntr; }
    CP_HD int pre(int s) const { return s >= 2 ? width(s) : 0; }
    CP_HD int rest(int s) const {
        if (s <= L || (s - 1) % L != 0 || s + 1 >= nblk) return 0;
        const int n = nblk - s - 1;                      // rows s + 1 .. nblk - 1
        return n * (n + 1) / 2 + n * ntr;
    }
    CP_HD int segment(int s) const { return pre(s) + width(s) + rest(s); }
    CP_HD int total() const {
        int t = 0;
        for (int s = 0; s < nblk; ++s) t += segment(s);
        return t;
    }
};

enum { TASK_PRE = 0, TASK_CHAIN = 1, TASK_REST = 2 };
struct ChainTask {
    int kind, s, i, xi;        // xi: position in row i (factor tiles first)
    int r0, kcnt;              // block rows [r0, r0 + kcnt) to apply
};

CP_HD inline ChainTask chain_decode(const ChainShape &sh, int t) {
    ChainTask k;
    int s = 0;
    for (;; ++s) {
        const int seg = sh.segment(s);
        if (t < seg) break;
        t -= seg;
    }
    k.s = s;
    if (t < sh.pre(s)) {
        k.kind = TASK_PRE;
        k.i = s;
        k.xi = t;
        k.r0 = sh.L * ((s - 2) / sh.L);
        k.kcnt = s - 1 - k.r0;
        return k;
    }
    t -= sh.pre(s);
    if (t < sh.width(s)) {
        k.kind = TASK_CHAIN;
        k.i = s;
        k.xi = t;
        k.r0 = s == 0 ? 0 : s - 1;
        k.kcnt = s == 0 ? 0 : 1;
        return k;
    }
    t -= sh.width(s);
    k.kind = TASK_REST;
    int i = s + 1;
    while (t >= sh.width(i)) {
        t -= sh.width(i);
        ++i;
    }
    k.i = i;
    k.xi = t;
    k.r0 = s - 1 - sh.L;
    k.kcnt = sh.L;
    return k;
}
