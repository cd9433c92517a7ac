// The task order of the persistent blocked Cholesky (channel-pruning_amd/csrc/chain_order.h) walked on the host for every
// shape: the hand-out of tasks off ONE counter is free of deadlock only if every task comes after everything it waits for.
// Simulates the version words exactly as k_chol_chain polls and raises them:
//   * a task on tile (i, x) applying block rows [r0, r0 + kcnt) finds ver[i][x] == r0 (its tile has exactly the rows before),
//     and every finished tile it multiplies with -- (r, i) and (r, x) for r in the range -- has ver == r + 1 ALREADY;
//   * every tile receives the rows 0 .. i - 1 once each, in order, the last one in its chain task, and ends final (i + 1);
//   * the task count equals ChainShape::total(), every (tile, row) pair is covered, nothing is decoded twice.
// Build: g++ -O2 -std=c++17 -I channel-pruning_amd/csrc tests/host/test_chain_order.cpp
#include <cstdio>
#include <vector>

#include "chain_order.h"

static int check(int nblk, int ntr, int L) {
    const ChainShape sh{nblk, ntr, L};
    const int XW = nblk + ntr, total = sh.total();
    std::vector<int> ver(size_t(nblk) * XW, 0);
    long long applied = 0;
    int chain_tasks = 0;
    for (int t = 0; t < total; ++t) {
        const ChainTask k = chain_decode(sh, t);
        const int i = k.i;
        if (i < 0 || i >= nblk || k.xi < 0 || k.xi >= sh.width(i)) return printf("task %d: tile out of range\n", t), 1;
        const bool rhs = k.xi >= nblk - i;
        const int x = rhs ? nblk + (k.xi - (nblk - i)) : i + k.xi;
        if (k.kcnt < 0 || k.kcnt > (L > 1 ? L : 1) || k.r0 < 0 || k.r0 + k.kcnt > i) return printf("task %d: bad row range\n", t), 1;
        if (k.kind == TASK_CHAIN ? (k.r0 + k.kcnt != i || k.kcnt != (i ? 1 : 0) || k.s != i) : k.kcnt < 1)
            return printf("task %d: chain task must apply exactly row i - 1\n", t), 1;
        if (ver[size_t(i) * XW + x] != k.r0) return printf("task %d on tile (%d, %d): ver %d, wants %d\n", t, i, x, ver[size_t(i) * XW + x], k.r0), 1;
        for (int r = k.r0; r < k.r0 + k.kcnt; ++r) {
            if (ver[size_t(r) * XW + i] != r + 1) return printf("task %d: U[%d, %d] not final yet\n", t, r, i), 1;
            if (ver[size_t(r) * XW + x] != r + 1) return printf("task %d: tile (%d, %d) not final yet\n", t, r, x), 1;
        }
        applied += k.kcnt;
        if (k.kind == TASK_CHAIN) {
            ver[size_t(i) * XW + x] = i + 1;
            ++chain_tasks;
            // the diagonal role of a step comes before its panel roles (they wait for its operator inside the task)
            if (k.xi > 0 && ver[size_t(i) * XW + i] != i + 1) return printf("task %d: panel before its diagonal tile\n", t), 1;
        } else {
            ver[size_t(i) * XW + x] = k.r0 + k.kcnt;
        }
    }
    long long want_applied = 0;
    int tiles = 0;
    for (int i = 0; i < nblk; ++i)
        for (int xi = 0; xi < sh.width(i); ++xi) {
            const bool rhs = xi >= nblk - i;
            const int x = rhs ? nblk + (xi - (nblk - i)) : i + xi;
            if (ver[size_t(i) * XW + x] != i + 1) return printf("tile (%d, %d) not final at the end\n", i, x), 1;
            want_applied += i;
            ++tiles;
        }
    if (applied != want_applied || chain_tasks != tiles) return printf("coverage: %lld of %lld rows, %d of %d tiles\n", applied, want_applied, chain_tasks, tiles), 1;
    return 0;
}

int main() {
    int shapes = 0;
    for (int L = 1; L <= CHAIN_L_MAX; ++L)
        for (int nblk = 1; nblk <= 48; ++nblk)
            for (int ntr : {0, 1, 2, 4, 16, CHAIN_NTR_MAX}) {
                if (check(nblk, ntr, L)) return printf("FAILED: nblk %d ntr %d L %d\n", nblk, ntr, L), 1;
                ++shapes;
            }
    if (check(144, 16, 4) || check(144, 1, 2)) return 1;
    printf("chain order ok: %d shapes\n", shapes + 2);
    return 0;
}
