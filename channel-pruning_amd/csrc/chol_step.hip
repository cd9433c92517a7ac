// Blocked Cholesky G = U^T U (upper factor, 128-blocks) as ONE launch per block step, the trailing update applied lazily.
//
// Why one launch per step.  On a chip that is busy with other layers' products every kernel of a dependent chain waits
// for its dispatch whatever its size: a side-car probe next to the vgg16 job (tools/ubench/sidecar.hip,
// profiles/r04_sidecar_next_to_vgg16_job.md) times launch -> completion of a ONE-workgroup kernel at 70-100 us median and
// 350-570 us at the 90th percentile (18 us on an idle chip) -- the same for 64 threads without LDS and for 512 threads with
// 158 KB of it.  The chain's cost there is its launch count, so the two launches per block step of rounds 1-3 (diagonal
// block + panel, then the trailing update as a GEMM) become one, and that launch overlaps the one serial piece of a step
// (the 128 x 128 factorisation in one workgroup) with the chip-filling part of the previous step:
//
//   launch s, one workgroup per upper tile (i, j), s <= i <= j < nblk, row s first:
//     every tile   S = G[i,j] - sum_r U[r,i]^T U[r,j]        the updates not applied yet (MFMA).  Block row s: r = s - 1.  The
//                                                           tiles below it are touched at every SECOND launch (even s) with
//                                                           r = s - 2 and s - 1 (K = 256); at odd s only block row s + 1
//                                                           rides along and takes r = s - 1 early, so that the serial piece
//                                                           of every step stays at K = 128
//     (s, s)       S = U_ss^T U_ss in LDS with look-ahead over its 16-column panels, T_q = (16 x 16 diagonal blocks)^-1, each block
//                  column published (flag word = columns out) as it becomes final; then U_ss^-1  -> U[s,s], operator, TI_s, TIT_s
//     (s, j > s)   U[s,j] = U_ss^-T S by block forward substitution, block column by block column BEHIND the publication
//                                                                                                -> U[s,j], Lt[j,s]
//     (i > s, j)   G[i,j] = S
//   The right-hand sides R of the solve that follows ride along as extra tile columns of every block row (B operand: block
//   row s - 1 of Y instead of U): when the last step ends, R holds Y = U^-T R -- the forward substitution costs no launch.
//
// A workgroup of row s waits only for workgroup 0 of its own launch, which the dispatcher starts first; the tiles below row s
// wait for nothing, so the factorisation of block s runs while the rest of the chip applies update s - 1.
//
// "The operator of block s" (36 blocks of 16 x 16: the upper triangle of the 8 x 8 block grid): the strictly upper blocks
// of U_ss and, in the diagonal slots, T_q = U_qq^-1.  It is what a panel workgroup needs for the substitution (no 128 x 128
// inverse on the chain); the diagonal role leaves it in the tile G[s,s] it came from.  TI_s = U_ss^-1 / TIT_s, which the
// substitution kernels of refit.hip multiply with, are computed by the same workgroup AFTER it has raised the flag.
//
// LDS: 75,776 B per workgroup (operand chunks of the update / the packed upper triangle of the tile), 512 threads, <= 128
// VGPRs: two workgroups per CU, or one next to a 72 KB / 128-VGPR GEMM workgroup of another layer -- a workgroup that needs
// a CU to itself waits for one while GEMM launches with workgroups still to place keep refilling every half CU that frees up
// (rocprof timeline of the job: the first steps of a factorisation took 1.1-1.4 ms next to other layers' Gram GEMMs).
#include "cp_common.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

typedef double v4f64s __attribute__((ext_vector_type(4)));
typedef double v2f64s __attribute__((ext_vector_type(2)));

namespace {

constexpr int NB = 128, PNB = 16, NPAN = NB / PNB, PT = 512;
constexpr int KCH_BASE = 16;                 // k-rows of an operand chunk staged through LDS (two double2 per thread and operand)
constexpr int KCH = KCH_BASE;
constexpr int SLD = NB + 16;                 // padded chunk row: the two k-rows a 32-lane half reads fall on disjoint bank halves
constexpr int PACK = 36 * PNB * PNB;         // packed upper triangle of a 128 x 128 tile in 16 x 16 blocks
constexpr int LDS_DOUBLES = PACK + 2 * NB;   // + dinv[128] + dref[128]
static_assert(2 * KCH * SLD <= PACK, "the operand chunks and the packed tile share the same LDS");

// slot of block (bi, bj), bi <= bj, in the packed upper triangle (row-major)
__device__ __forceinline__ constexpr int pk(int bi, int bj) { return bi * NPAN - bi * (bi - 1) / 2 + (bj - bi); }
// element (r, c) of the tile, block row <= block column
#define CP_PK(r, c) sm[pk((r) >> 4, (c) >> 4) * 256 + ((r) & 15) * 16 + ((c) & 15)]

// The roles below are functions of their own (not inlined), so their pointer parameters arrive as GENERIC pointers and every
// access through them would be a flat_ instruction.  A flat access counts on lgkmcnt as well as vmcnt: the s_waitcnt
// lgkmcnt(0) in front of the first LDS read of a chunk then also waits for the global loads of the NEXT chunk that were
// just issued -- the prefetch is gone and every chunk pays a full memory latency (745 flat_ against 4 global_ instructions in
// this file before; a 256-deep tile update ran 95 us alone on its CU for 28 us of matrix work).  Every global operand is
// therefore cast to the global address space at the head of its role.
#define CP_GLOBAL __attribute__((address_space(1)))
typedef CP_GLOBAL double *gdp;
typedef const CP_GLOBAL double *gcdp;
typedef CP_GLOBAL v2f64s *gv2p;
typedef const CP_GLOBAL v2f64s *gcv2p;

__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

__device__ __forceinline__ double read_lane(double v, int lane) {   // lane: wave-uniform (constant after unrolling) -> SGPR pair
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// spin_limit: 1 << 26 polls of ~100 ns (seconds); 0 in the test of the time-out path (cp_debug_chol_fail_flag_wait)
__device__ __forceinline__ void flag_wait(const int *flag, int *info, int spin_limit, int want = 1, int *stop = nullptr) {
    // RELAXED agent-scope load (global_load ... sc1: never served from this XCD's non-coherent L2 lines).  An ACQUIRE here is a
    // `buffer_inv sc1` after EVERY poll -- an invalidation of the whole L2 of the XCD, issued by ~35 panel workgroups for
    // ~45 us per step, on top of the one all their waves issued after the wait: during a job's factorisation phase every
    // XCD lost its L2 contents every microsecond or two, for every kernel running next to this one.  The operator the flag
    // announces is read with sc1 loads as well (role_panel), so no fence is needed at all.
    for (int spin = 0; spin < spin_limit; ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
        if (stop && (spin & 15) == 15 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        __builtin_amdgcn_s_sleep(2);
    }
    if (stop) __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Not observed outside the test: the host reads info[0] != 0 as "this factorisation is not to be trusted" and the refit
    // goes on to its rank-revealing path (refit.hip: refit_solve_tail -> refit_robust), which factors again -- a slower
    // route to the same result and never a hang (that path can still end in CP_ERR_NUMERIC on a matrix it cannot factor
    // either; a time-out alone does not make a layer fail).
    atomicCAS(info, 0, 0x7fffffff);
}

// acc[t] (wave w: rows 16 t + fk + 4 r, column 16 w + fi of the tile) -= A^T B over K = 128 kcnt, A = U[r0 .., i-block], B = U[r0 .., j-block]
// or, for a right-hand-side tile, Y[r0 .., j-block] (k-major blocks of kcnt consecutive block rows, leading dimensions ld / ldb).  Both operands go through LDS in chunks of KCH k-rows (wide coalesced
// loads, the next chunk in flight while the current one is multiplied); SAME: A and B are the same block (diagonal tile),
// and only the upper blocks t <= w are wanted.
// Block column of the DIAGONAL tile a wave owns.  Column w has w + 1 upper blocks, and waves w and w + 4 share a SIMD: with
// column = wave the four SIMDs carry 6 / 8 / 10 / 12 blocks of the K = 128 update (the serial piece of a step waits for the
// last); with the columns handed out as 7, 6, 5, 4 | 0, 1, 2, 3 every SIMD carries 9.
__device__ __forceinline__ int diag_col(int wave) { return wave < 4 ? 7 - wave : wave - 4; }

// BSC1 (the persistent form, right-hand-side tiles): the B operand -- block rows of Y that their owners finished IN PLACE, in
// R -- is read with agent-scope (sc1) loads: those addresses were read before, as the tiles they used to be, by workgroups
// of any XCD, and a plain load could be served from such a stale L2 line.  (U is written once, into a buffer nobody read
// before: plain loads, shared through the L2 by the workgroups of an XCD.)
__device__ __forceinline__ v2f64s load_v2_sc1(gcv2p ptr) {
    typedef const CP_GLOBAL unsigned long long *gcup_;
    const gcup_ q = (gcup_)ptr;
    v2f64s v;
    v[0] = __longlong_as_double((long long)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    v[1] = __longlong_as_double((long long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return v;
}

template <bool SAME, bool BSC1 = false>
__device__ __forceinline__ void tile_update(v4f64s (&acc)[NPAN], gcdp Ab, int ld, gcdp Bb, int ldb, int kcnt, double *sm) {
    const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
    const int wave = SAME ? diag_col(__builtin_amdgcn_readfirstlane(tid >> 6)) : __builtin_amdgcn_readfirstlane(tid >> 6);
    // the diagonal tile stages ONE operand, so its chunks can be twice as deep in the same LDS: 4 instead of 8 chunks (two
    // barriers and an exposed LDS / global latency each) on the serial piece of the step
    constexpr int KCH = SAME ? 2 * KCH_BASE : KCH_BASE;
    double *As = sm, *Bs = SAME ? sm : sm + KCH * SLD;
    constexpr int PER = KCH * NB / 2 / PT;   // double2 loads per thread, operand and chunk (2; 4 for the diagonal tile)
    v2f64s ar[PER], br[PER];
    // uniform base + 32-bit lane offset: the loads take the scalar-base form instead of a 64-bit address pair per load
    const int goff = (tid >> 6) * ld + (tid & 63) * 2;   // thread's (row, column pair) inside a chunk; +8 rows per i
    const int goffb = (tid >> 6) * ldb + (tid & 63) * 2;
    auto gload = [&](int ch) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            gcdp ab = Ab + size_t(ch * KCH + (PT >> 6) * i) * ld, bb = Bb + size_t(ch * KCH + (PT >> 6) * i) * ldb;
            ar[i] = *(gcv2p)(ab + goff);
            if constexpr (!SAME) {
                if constexpr (BSC1) br[i] = load_v2_sc1((gcv2p)(bb + goffb));
                else br[i] = *(gcv2p)(bb + goffb);
            }
        }
    };
    // kcnt (1, 2; up to CHAIN_L in the persistent form) consecutive block rows of the operands: K = 128 kcnt, the rows of a
    // block row are contiguous in k
    const int nch = kcnt * (NB / KCH);
    gload(0);
    for (int ch = 0; ch < nch; ++ch) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + PT * i, r = e >> 6, c = (e & 63) * 2;
            *reinterpret_cast<v2f64s *>(&As[r * SLD + c]) = ar[i];
            if constexpr (!SAME) *reinterpret_cast<v2f64s *>(&Bs[r * SLD + c]) = br[i];
        }
        __syncthreads();
        if (ch + 1 < nch) gload(ch + 1);
#pragma unroll
        for (int kk = 0; kk < KCH / 4; ++kk) {
            const double b = Bs[(kk * 4 + fk) * SLD + 16 * wave + fi];
#pragma unroll
            for (int t = 0; t < NPAN; ++t) {
                if (SAME && t > wave) continue;   // wave-uniform
                const double a = -As[(kk * 4 + fk) * SLD + 16 * t + fi];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
}

// The diagonal role in phases, the long ones functions of their own (not inlined: the register allocator then sees one phase
// at a time -- as one function the 16 x 16 in-register factorisation spilled on the chain at 128 VGPRs):
//   diag_to_lds            the tile (upper blocks t <= column of `acc`) into the packed LDS tile, pivot references
//   diag_factor_lookahead  U_ss^T U_ss = S in LDS with look-ahead over the 16-column panels; U[s,s] and the operator of block s
//                          leave for global memory as they become final, the flag word counts the published block columns
//   diag_inverse           TI_s / TIT_s, after the last publication
__device__ __forceinline__ void diag_to_lds(v4f64s (&acc)[NPAN], double *sm, gcdp dg0_blk) {
    const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
    const int wave = diag_col(__builtin_amdgcn_readfirstlane(tid >> 6));
    double *dref = sm + PACK + NB;
#pragma unroll
    for (int t = 0; t < NPAN; ++t) {
        if (t > wave) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[pk(t, wave) * 256 + (fk + 4 * r) * 16 + fi] = acc[t][r];
    }
    if (tid < NB) dref[tid] = dg0_blk[tid];
    __syncthreads();
}

#ifdef CP_CHOL_STAMPS   // diagnostic build (tools/ubench/chol_bulk.hip): where the look-ahead factorisation spends its cycles
__device__ unsigned long long cp_chol_diag_stamps[128];
#define CP_DSTAMP(slot, who)                                                                                  \
    do {                                                                                                      \
        if (threadIdx.x == (who)) cp_chol_diag_stamps[(slot)] = __builtin_readcyclecounter();                 \
    } while (0)
#else
#define CP_DSTAMP(slot, who) do { } while (0)
#endif

// ---- the diagonal role: look-ahead over the 16-column panels -------------------------------------------------------------
// What a plain right-looking sweep over the panels (rounds 1-4, HISTORY.md) leaves on the chain per panel: the 16 x 16
// factorisation by ONE wave while seven wait at a barrier, the U12 substitution as one thread per column (16 dependent steps of
// LDS broadcasts), the trailing update, three barriers; then, after the last panel, the inversion of the eight diagonal blocks,
// U[s,s] and the operator to global memory, the flag.  Here:
//   * the in-register factorisation of a diagonal block carries the identity along (the same row operations): it ends with
//     U_pp AND U_pp^-T = T_p^T in registers, so T_p goes straight to its slot of the operator and no inversion pass is left;
//   * U_pj = T_p^T A_pj is four MFMAs per 16 x 16 block (wave per block column) instead of a thread per column;
//   * wave 0 applies panel p to block (p + 1, p + 1) FIRST and factors it while waves 1 .. 7 apply panel p to the other 27
//     (... 0) blocks: the factorisation of the next diagonal block hides the trailing update (or the other way round);
//   * U[s,s] and the operator leave for global memory block by block as they become final (fire-and-forget stores issued by
//     the wave that holds the block in registers); after the last panel only the flag is left.
// Two barriers per panel.  LDS as before: the packed upper triangle; the diagonal slots end up holding T_p.
__device__ __forceinline__ void store_block_global(gdp Ub, CP_GLOBAL unsigned long long *Go, int ld, int bi, int bj, const v4f64s &y,
                                                   int fk, int fi, bool upper_only) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * bi + fk + 4 * r, col = 16 * bj + fi;
        const double v = (upper_only && fi < fk + 4 * r) ? 0.0 : y[r];
        Ub[row * ld + col] = v;                                            // U[s,s]
        if (Go) __hip_atomic_store(Go + row * ld + col, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);              // operator: strictly upper blocks of U_ss
    }
}

// wave 0: v = the (fully updated) diagonal block p in the D lay-out -> U_pp (global U[s,s]), T_p (its LDS slot; wave 7 copies it
// into the operator).
// A lone wave issues an instruction every ~7 cycles whatever it is, so a pivot costs what its instruction count costs (the
// first version of this function: ~60 instructions, 550 cycles per pivot, 8.8k per block).  This one keeps the block UNSCALED
// while it eliminates -- a[i][j] -= a[k][i] a[k][j] / a[k][k] -- so that
//   * the cross-lane fetches of row k (ds_bpermute with lane addresses computed once per call) do not wait for 1 / sqrt: they are
//     in flight while the reciprocal of the pivot is refined;
//   * only the rows that still change are fetched / updated (compile-time: r > k / 4; lane-dependent only for r = k / 4);
//   * the pivot test is one vector compare after the loop (a failed pivot poisons the block with NaN / Inf, which the caller
//     discards anyway once info[0] is set);
// and scales the rows by 1 / sqrt(pivot) once at the end.  ~33 instructions per pivot.
__device__ __forceinline__ double bperm(int addr, double x) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void factor_block_inreg(v4f64s v, double *sm, int p, double piv_tol, int *info, int blk, gdp Ub,
                                                   int ld, int lane, int fk, int fi) {
    const double *dref = sm + PACK + NB;
    const int k0 = p * PNB;
    v4f64s w, sc;                                // w: the identity carried along (ends as U_pp^-T); sc[r]: 1 / sqrt(pivot of row fk + 4 r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w[r] = (fk + 4 * r == fi) ? 1.0 : 0.0;
        sc[r] = 1.0;
    }
    const double thr = piv_tol * dref[k0 + fi];   // the pivot of column fi has to stay above this
    bool bad = false;
    int a_row[4], a_col[4][4];                    // byte addresses of the source lanes: (kq, fi) and (kq, fk + 4 r)
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        a_row[kq] = (kq * 16 + fi) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) a_col[kq][r] = (kq * 16 + fk + 4 * r) * 4;
    }
#pragma unroll
    for (int k = 0; k < PNB; ++k) {
        const int kr = k >> 2, kq = k & 3;
        const double piv = read_lane(v[kr], kq * 16 + k);          // a[k][k], uniform
        // row k (unscaled) to every lane that needs it: column fi of it, and the columns that are this lane's rows
        const double urow = bperm(a_row[kq], v[kr]);
        const double xrow = bperm(a_row[kq], w[kr]);
        double ucol[4];
#pragma unroll
        for (int r = kr; r < 4; ++r) ucol[r] = bperm(a_col[kq][r], v[kr]);
        // 1 / piv and 1 / sqrt(piv) meanwhile
        const double inv = rsqrt_nr(piv);
        const double rinv = inv * inv;
        if (fi == k && fk == kq) bad = bad || !(piv > thr);          // lane (kq, k): thr belongs to column k
        const double ut = urow * rinv, xt = xrow * rinv;
        if (fk <= kq) ucol[kr] = 0.0;                                // rows fk + 4 kr <= k of this register do not change
#pragma unroll
        for (int r = kr; r < 4; ++r) {
            v[r] = fma(-ucol[r], ut, v[r]);
            w[r] = fma(-ucol[r], xt, w[r]);
        }
        if (fk == kq) sc[kr] = inv;
    }
    const unsigned long long any_bad = __ballot(bad);
    if (any_bad != 0 && lane == 0) {
        // the lane that tests pivot k is (k & 3) * 16 + k: lane order is not pivot order (pivot 1 sits in lane 17, pivot 4 in
        // lane 4), and after the first failing pivot the block is NaN-poisoned so that later pivots are flagged too.  Report
        // the smallest column among the flagged lanes -- the first pivot that really failed
        int first = 16;
        for (unsigned long long m = any_bad; m; m &= m - 1) first = min(first, (__ffsll((long long)m) - 1) & 15);
        atomicCAS(info, 0, blk * NB + k0 + first + 1);
    }
    double *Dp = sm + pk(p, p) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = fk + 4 * r;
        const double u = v[r] * sc[r], x = w[r] * sc[r];
        Dp[fi * 16 + row] = x;                                              // T_p[fi][row] = U^-T[row][fi]
        Ub[(k0 + row) * ld + k0 + fi] = fi >= row ? u : 0.0;                // U_pp, zeros below the diagonal
    }
}

// The operator is PUBLISHED block column by block column -- the flag word counts the columns q whose blocks (k, q), k < q, and
// T_q are in global memory -- so that the panel workgroups substitute behind the factorisation instead of starting when it
// ends (role_panel).  Who stores what: the strictly upper blocks by the waves 1 .. 7 that compute them (phase A; each waits for
// its stores before the barrier that ends phase B), T_q by wave 7, after the barrier that opens phase B of panel q: it copies
// T_q out of LDS, waits for the copy and raises the count.  Wave 0 -- the chain -- issues no operator store and never waits for
// memory.
__device__ __noinline__ void diag_factor_lookahead(double *sm, double piv_tol, int *info, int blk, double *Ub_, double *Gss_, int ld) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fk = lane >> 4, fi = lane & 15;
    const gdp Ub = (gdp)Ub_;
    CP_GLOBAL unsigned long long *const Go = (CP_GLOBAL unsigned long long *)Gss_;
    // the blocks below the diagonal of U[s,s] are zero
    for (int e = tid; e < 28 * 256; e += PT) {
        const int r = (e >> 4) & 15, c = e & 15;
        int a = 0, q = e >> 8;                  // the q-th of the 28 strictly upper block pairs (a, a + 1 + q') ...
        while (q >= 7 - a) {
            q -= 7 - a;
            ++a;
        }
        Ub[(16 * (a + 1 + q) + r) * ld + 16 * a + c] = 0.0;   // ... mirrored: block row a + 1 + q, block column a
    }
    if (wave == 0) {
        v4f64s v;
        const double *D0 = sm + pk(0, 0) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = D0[(fk + 4 * r) * 16 + fi];
        CP_DSTAMP(0, 0);
        factor_block_inreg(v, sm, 0, piv_tol, info, blk, Ub, ld, lane, fk, fi);
        CP_DSTAMP(1, 0);
    }
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NPAN; ++p) {
        CP_DSTAMP(8 + 8 * p + 0, 0);
        // phase A: U_pj = T_p^T A_pj, waves 1 .. 7 -> block columns p + 1 .. p + 7 (wave 0 issues no global store)
        const int j = wave == 0 ? NPAN : p + wave;
        if (j < NPAN) {
            const double *Tp = sm + pk(p, p) * 256;
            double *Cp = sm + pk(p, j) * 256;
            v4f64s y = {0., 0., 0., 0.};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                y = __builtin_amdgcn_mfma_f64_16x16x4f64(Tp[(4 * r + fk) * 16 + fi], Cp[(4 * r + fk) * 16 + fi], y, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Cp[(fk + 4 * r) * 16 + fi] = y[r];
            store_block_global(Ub, Go, ld, p, j, y, fk, fi, false);
        }
        CP_DSTAMP(8 + 8 * p + 1, 0);
        __syncthreads();
        CP_DSTAMP(8 + 8 * p + 2, 0);
        // publisher (wave 7, AFTER the barrier so that the chain does not wait for its store round trip): T_p out of its LDS
        // slot, then the count -- block column p is complete
        if (wave == NPAN - 1) {
            const double *Tp = sm + pk(p, p) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __hip_atomic_store(Go + (16 * p + fk + 4 * r) * ld + 16 * p + fi,
                                   (unsigned long long)__double_as_longlong(Tp[(fk + 4 * r) * 16 + fi]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CP_HANDOFF_RELEASE();          // nothing unless built with -DCP_HANDOFF_FENCES=1 (cp_common.h)
            if (lane == 0) __hip_atomic_store(info + 1 + blk, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p + 1 >= NPAN) break;
        // phase B: wave 0 takes block (p + 1, p + 1) -- update, then factor -- the others the rest of the trailing blocks
        const int rt = NPAN - p - 1;                     // trailing grid: rt x rt blocks, upper triangle
        if (wave == 0) {
            const double *Rq = sm + pk(p, p + 1) * 256;
            const double *Cq = sm + pk(p + 1, p + 1) * 256;
            v4f64s c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = Cq[(fk + 4 * r) * 16 + fi];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double rv = Rq[(kk * 4 + fk) * 16 + fi];
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(-rv, rv, c, 0, 0, 0);
            }
            CP_DSTAMP(8 + 8 * p + 3, 0);
            factor_block_inreg(c, sm, p + 1, piv_tol, info, blk, Ub, ld, lane, fk, fi);
            CP_DSTAMP(8 + 8 * p + 4, 0);
        } else {
            const int ntile = rt * (rt + 1) / 2;
            for (int e = wave; e < ntile; e += PT / 64 - 1) {        // e = 0 is wave 0's block
                int a = int((sqrtf(8.f * float(e) + 1.f) - 1.f) * 0.5f);
                while ((a + 1) * (a + 2) / 2 <= e) ++a;
                while (a * (a + 1) / 2 > e) --a;
                const int b = e - a * (a + 1) / 2;  // b <= a
                const int bi = p + 1 + b, bj = p + 1 + a;
                double *Cij = sm + pk(bi, bj) * 256;
                const double *Ri = sm + pk(p, bi) * 256, *Rj = sm + pk(p, bj) * 256;
                v4f64s c;
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = Cij[(fk + 4 * r) * 16 + fi];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ri[(kk * 4 + fk) * 16 + fi], Rj[(kk * 4 + fk) * 16 + fi], c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Cij[(fk + 4 * r) * 16 + fi] = c[r];
            }
            CP_DSTAMP(8 + 8 * p + 5, 64);      // wave 1 done with its share of the trailing blocks
            // this wave's phase-A stores (the blocks (p, j) of the operator) are out before block column p + 1 is published
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        CP_DSTAMP(8 + 8 * p + 6, 0);
    }
    // (the last column was published by wave 7 after the barrier of the last panel)
}

// TI_b = U_bb^-1 (upper) and TIT_b = its transpose from the operator in LDS: what the substitution kernels of refit.hip
// multiply with.  The diagonal role computes them AFTER it has raised the flag, i.e. off the chain, while the panel
// workgroups substitute.  V = U^-1 by block back-substitution, V_ij = -T_i sum_{k=i+1..j} U_ik V_kj: wave jb owns block
// column jb and keeps its V blocks in registers (the D lay-out of a block is the B operand of the next product).
template <int JB>
__device__ __forceinline__ void inverse_column(const double *sm, gdp TIb, gdp TITb) {
    const int lane = threadIdx.x & 63, fk = lane >> 4, fi = lane & 15;
    v4f64s V[JB + 1];
    {
        const double *Tj = sm + pk(JB, JB) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) V[JB][r] = Tj[(fk + 4 * r) * 16 + fi];
    }
#pragma unroll
    for (int ib = JB - 1; ib >= 0; --ib) {
        v4f64s S = {0., 0., 0., 0.};
#pragma unroll
        for (int kb = ib + 1; kb <= JB; ++kb) {
            const double *Uik = sm + pk(ib, kb) * 256;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                S = __builtin_amdgcn_mfma_f64_16x16x4f64(Uik[fi * 16 + kk * 4 + fk], V[kb][kk], S, 0, 0, 0);
        }
        const double *Ti = sm + pk(ib, ib) * 256;
        v4f64s W = {0., 0., 0., 0.};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) W = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ti[fi * 16 + kk * 4 + fk], S[kk], W, 0, 0, 0);
        V[ib] = W;
    }
    // rows of block column JB: blocks ib <= JB hold V, the blocks below are zero
#pragma unroll
    for (int ib = 0; ib < NPAN; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ib + fk + 4 * r, col = 16 * JB + fi;
            double v = 0.0;
            if (ib <= JB) v = V[ib < JB + 1 ? ib : JB][r];
            TIb[row * NB + col] = v;
            TITb[col * NB + row] = v;
        }
}

__device__ __noinline__ void diag_inverse(const double *sm, double *TIb_, double *TITb_) {
    const gdp TIb = (gdp)TIb_, TITb = (gdp)TITb_;
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
        case 0: inverse_column<0>(sm, TIb, TITb); break;
        case 1: inverse_column<1>(sm, TIb, TITb); break;
        case 2: inverse_column<2>(sm, TIb, TITb); break;
        case 3: inverse_column<3>(sm, TIb, TITb); break;
        case 4: inverse_column<4>(sm, TIb, TITb); break;
        case 5: inverse_column<5>(sm, TIb, TITb); break;
        case 6: inverse_column<6>(sm, TIb, TITb); break;
        default: inverse_column<7>(sm, TIb, TITb); break;
    }
}

#ifdef CP_CHOL_STAMPS   // diagnostic build (tools/ubench/chol_bulk.hip): shader-clock stamps of the phases of a bulk workgroup
__device__ unsigned long long cp_chol_stamps[4096 * 8];
#define CP_STAMP(slot)                                                                                       \
    do {                                                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096) cp_chol_stamps[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define CP_STAMP(slot) do { } while (0)
#endif

// A tile of launch s: where it lives, and the B operand of its update (block row s - 1 of U, or of Y for a right-hand side)
struct Tile {
    double *T;          // the 128 x 128 tile (G[i,j] or R[i,jr])
    int ldt;
    const double *B;    // U[r0, j-block] or Y[r0, jr-block]: the first of the kcnt block rows to apply
    int ldb;
    int kcnt;           // block rows to apply: 0 (nothing yet), 1 or 2
};

// the tile itself crosses workgroups (and XCDs) in the persistent form -- the next task on it may run anywhere -- so it is read
// and written with agent-scope (sc1) accesses there; between launches the kernel boundary does that job
template <bool SC1>
__device__ __forceinline__ double tile_ld(gcdp ptr) {
    if constexpr (SC1) {
        typedef const CP_GLOBAL unsigned long long *gcup_;
        return __longlong_as_double((long long)__hip_atomic_load((gcup_)ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
        return *ptr;
    }
}
template <bool SC1>
__device__ __forceinline__ void tile_st(gdp ptr, double v) {
    if constexpr (SC1) {
        typedef CP_GLOBAL unsigned long long *gup_;
        __hip_atomic_store((gup_)ptr, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *ptr = v;
    }
}

// acc <- the tile, then the pending updates (see the head of the file)
template <bool DIAG, bool PERSIST = false>
__device__ __forceinline__ void tile_load_update(v4f64s (&acc)[NPAN], const Tile &t_, const double *Ai_, int ld, double *sm,
                                                 bool rhs = false) {
    const gcdp T = (gcdp)t_.T, Bop = (gcdp)t_.B, Ai = (gcdp)Ai_;
    const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
    const int wave = DIAG ? diag_col(__builtin_amdgcn_readfirstlane(tid >> 6)) : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int toff = fk * t_.ldt + 16 * wave + fi;   // lane's offset inside a (16 t + 4 r)-row band of the tile: scalar base + 32-bit offset
#pragma unroll
    for (int t = 0; t < NPAN; ++t) {
        if (DIAG && t > wave) {
            acc[t] = v4f64s{0., 0., 0., 0.};
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = tile_ld<PERSIST>((T + size_t(16 * t + 4 * r) * t_.ldt) + toff);
    }
    CP_STAMP(1);
    if (t_.kcnt > 0) {
        if (PERSIST && !DIAG && rhs) tile_update<DIAG, PERSIST && !DIAG>(acc, Ai, ld, Bop, t_.ldb, t_.kcnt, sm);
        else tile_update<DIAG>(acc, Ai, ld, Bop, t_.ldb, t_.kcnt, sm);
    }
    CP_STAMP(2);
}

// the three roles of a workgroup of launch s; each is a function of its own (not inlined) so that the register allocation of
// one role does not see the live ranges of the others (inlined, the kernel spilled ~100 registers even at 256), and each ends
// the program itself
template <bool PERSIST>
__device__ __noinline__ void role_bulk(Tile t_, const double *__restrict__ Ai, int ld, int s, double *sm, bool rhs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fk = lane >> 4, fi = lane & 15;
    v4f64s acc[NPAN];
    CP_STAMP(0);
    tile_load_update<false, PERSIST>(acc, t_, Ai, ld, sm, rhs);
    const int toff = fk * t_.ldt + 16 * wave + fi;
    const gdp T = (gdp)t_.T;
#pragma unroll
    for (int t = 0; t < NPAN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile_st<PERSIST>((T + size_t(16 * t + 4 * r) * t_.ldt) + toff, acc[t][r]);
    CP_STAMP(3);
    if constexpr (!PERSIST) __builtin_amdgcn_endpgm();
}

template <bool PERSIST>
__device__ __noinline__ void role_diag(Tile t_, const double *__restrict__ Ai, double *__restrict__ Uss, int ld, int s,
                                       const double *__restrict__ dg0, double piv_tol, double *__restrict__ TIb,
                                       double *__restrict__ TITb, int *info, double *sm) {
    CP_STAMP(0);
    {
        v4f64s acc[NPAN];
        tile_load_update<true, PERSIST>(acc, t_, Ai, ld, sm);
        diag_to_lds(acc, sm, (gcdp)dg0 + size_t(s) * NB);
    }
    CP_STAMP(3);
    diag_factor_lookahead(sm, piv_tol, info, s, Uss, t_.T, ld);   // U[s,s], the operator and the flag leave from inside
    CP_STAMP(4);
    CP_STAMP(5);
    diag_inverse(sm, TIb, TITb);
    CP_STAMP(6);
    if constexpr (!PERSIST) __builtin_amdgcn_endpgm();
}

// out / ldo: where U[s,j] (or Y[s,jr]) goes; Ltjs: the transposed copy of a factor tile, null for a right-hand side
// stop (persistent form): the factorisation's abort word -- a time-out anywhere ends every wait of the launch
template <bool PERSIST>
__device__ __noinline__ void role_panel(Tile t_, const double *__restrict__ Ai, int ld, int s, double *__restrict__ out, int ldo,
                                        double *__restrict__ Ltjs, const double *__restrict__ Gss, int *info, int spin_limit,
                                        double *sm, bool rhs, int *stop) {
    const int tid = threadIdx.x, lane = tid & 63, fk = lane >> 4, fi = lane & 15;
    v4f64s acc[NPAN];
    CP_STAMP(0);
    tile_load_update<false, PERSIST>(acc, t_, Ai, ld, sm, rhs);
    // Substitution BEHIND the factorisation: block column q of the operator (the blocks (k, q), k < q, and T_q) is fetched and
    // applied as soon as the diagonal workgroup has published it (the flag word counts the published columns), so that after
    // the last publication only the last of the eight steps is left -- not the operator load and all eight.
    typedef const CP_GLOBAL unsigned long long *gcup;
    const gcup Go = (gcup)Gss;
#pragma unroll
    for (int q = 0; q < NPAN; ++q) {
        if (tid == 0) flag_wait(info + 1 + s, info, spin_limit, q + 1, stop);  // bounded; running out is reported as a failed factorisation
        __syncthreads();
        CP_HANDOFF_ACQUIRE();
        if (q == NPAN - 1) CP_STAMP(3);
        // (q + 1) blocks of 256 words, a word per thread and round: sc1 loads (see flag_wait), all in flight together
        constexpr int ROUNDS_MAX = (NPAN * 256 + PT - 1) / PT;
        const int rounds = ((q + 1) * 256 + PT - 1) / PT;  // compile-time after unrolling
        unsigned long long wv[ROUNDS_MAX];
#pragma unroll
        for (int i = 0; i < ROUNDS_MAX; ++i) {
            const int e = tid + PT * i;                    // word e of the column: block k = e / 256, element (r, c)
            if (i < rounds && e < (q + 1) * 256)
                wv[i] = __hip_atomic_load(Go + (16 * (e >> 8) + ((e >> 4) & 15)) * ld + 16 * q + (e & 15), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < ROUNDS_MAX; ++i) {
            const int e = tid + PT * i;
            if (i < rounds && e < (q + 1) * 256) sm[pk(e >> 8, q) * 256 + (e & 255)] = __longlong_as_double((long long)wv[i]);
        }
        __syncthreads();
        if (q == NPAN - 1) CP_STAMP(4);
        // block row q of the tile: acc[q] <- T_q^T (acc[q] - sum_{k < q} U_kq^T acc[k]); a finished block in the D lay-out is
        // the B operand of the next product as it is
#pragma unroll
        for (int k = 0; k < q; ++k) {
            const double *Ukq = sm + pk(k, q) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ukq[(4 * r + fk) * 16 + fi], acc[k][r], acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const double *Tq = sm + pk(q, q) * 256;
        v4f64s y = {0., 0., 0., 0.};
#pragma unroll
        for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Tq[(4 * r + fk) * 16 + fi], acc[q][r], y, 0, 0, 0);
        acc[q] = y;
        // rows 16 q .. 16 q + 15 of the tile are final: on their way (U[s,j] or Y[s,jr] row-major and, for a factor tile, the
        // transpose Lt[j,s]) while the next block column is waited for.  Persistent form: U[s,j] / Y[s,jr] are read by other
        // workgroups of this launch (sc1 stores, waited for before the tile is announced); Lt only by later kernels
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int uoff = fk * ldo + 16 * wave + fi, loff = (16 * wave + fi) * ld + fk;
        const gdp Uo = (gdp)out, Lo = (gdp)Ltjs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tile_st<PERSIST>((Uo + size_t(16 * q + 4 * r) * ldo) + uoff, y[r]);
            if (Ltjs) (Lo + (16 * q + 4 * r))[loff] = y[r];
        }
    }
    CP_STAMP(5);
    if constexpr (!PERSIST) __builtin_amdgcn_endpgm();
}

// R (p_pad x ntr 128-column tiles, leading dimension ldr; null: none): right-hand sides riding along -- block row i of the
// launch has ntr more tiles after its nblk - i factor tiles, and after the last step R holds Y = U^-T R (the forward
// substitution of the normal-equation solve, for free in the launches of the factorisation).
__global__ void __launch_bounds__(PT, 4)
k_chol_step(double *__restrict__ G, double *__restrict__ U, double *__restrict__ Lt, int ld, int nblk, int s,
            const double *__restrict__ dg0, double piv_tol, double *__restrict__ TI, double *__restrict__ TIT, int *info,
            double *__restrict__ R, int ldr, int ntr, int spin_limit) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int i = s, jt;
    {   // tile -> (i, jt): block row i holds nblk - i factor tiles, then ntr right-hand-side tiles; row s first.
        // (Row s AFTER the rows below it, so that its workgroups are dispatched when the diagonal tile's flag is almost up
        //  instead of holding CUs while they spin, was measured equal: vgg16 job 26.4 / 26.9 against 26.7 / 26.9 ms.)
        int t = blockIdx.x;
        while (t >= nblk - i + ntr) {
            t -= nblk - i + ntr;
            ++i;
        }
        jt = t;
    }
    const bool rhs = jt >= nblk - i;
    const int j = rhs ? jt - (nblk - i) : i + jt;     // right-hand-side tile column, or block column of the factor
    // Block rows still to apply to this tile.  The tiles below block row s + 1 are only touched at EVEN s >= 2, with the two
    // block rows s - 2 and s - 1 at once (K = 256: half the read-modify-write passes over G, 10.7 instead of 8 flop per byte).
    // The serial piece of a step stays at K = 128 all the same: at odd s the tiles of block row s + 1 ride along and take
    // block row s - 1 early, so the tiles of block row s -- the ones that are factored / substituted in this launch -- always
    // find exactly one block row, s - 1, left to apply.
    const int kcnt = s == 0 ? 0 : ((i > s && !(s & 1)) ? 2 : 1), r0 = s == 0 ? 0 : ((i > s && !(s & 1)) ? s - 2 : s - 1);
    const double *Urow = U + size_t(r0) * NB * ld;
    const double *Ai = Urow + size_t(i) * NB;
    Tile t_;
    t_.kcnt = kcnt;
    if (rhs) {
        t_.T = R + size_t(i) * NB * ldr + size_t(j) * NB;
        t_.ldt = ldr;
        t_.B = R + size_t(r0) * NB * ldr + size_t(j) * NB;
        t_.ldb = ldr;
    } else {
        t_.T = G + size_t(i) * NB * ld + size_t(j) * NB;
        t_.ldt = ld;
        t_.B = Urow + size_t(j) * NB;
        t_.ldb = ld;
    }
    const double *Gss = G + size_t(s) * NB * ld + size_t(s) * NB;   // where the diagonal role leaves the operator of block s
    // The workgroups of block row s are the serial piece of the step: their waves get the higher issue priority over the
    // bulk workgroups (of this or any other layer's launch) that share their SIMDs.
    if (i == s) __builtin_amdgcn_s_setprio(3);
    if (i > s)            // below the block row of this step: the updated tile goes back
        role_bulk<false>(t_, Ai, ld, s, sm, rhs);
    else if (!rhs && j == s)
        role_diag<false>(t_, Ai, U + size_t(s) * NB * ld + size_t(s) * NB, ld, s, dg0, piv_tol, TI + size_t(s) * NB * NB,
                         TIT + size_t(s) * NB * NB, info, sm);
    else if (!rhs)
        role_panel<false>(t_, Ai, ld, s, U + size_t(s) * NB * ld + size_t(j) * NB, ld, Lt + size_t(j) * NB * ld + size_t(s) * NB,
                          Gss, info, spin_limit, sm, false, nullptr);
    else
        role_panel<false>(t_, Ai, ld, s, t_.T, ldr, nullptr, Gss, info, spin_limit, sm, true, nullptr);
}

// =============================================================================================================================
// The persistent form: the whole factorisation as ONE task list, handed out inside one to four launches (cp_chol_factor_steps
// cuts the list at step boundaries: the trailing matrix shrinks, and each launch brings the workgroups ITS rows can use).
//
// In a job the launch-per-step chain above costs what its launch COUNT costs: next to other layers' products a step that takes
// 68-75 us alone took 265 us (profiles/r05_kernels_vgg16.md: 243 launches, 64.5 ms of stream time per job), because every launch
// is a barrier for the whole factorisation and then waits for workgroup slots on a full chip.  Here the same tile tasks -- the
// same arithmetic on every tile, in the same order: U comes out bit for bit as from k_chol_step -- are handed out INSIDE one
// launch: a grid of W workgroups that stay resident and take the next task off ONE counter, in a fixed linear order in which
// every task comes after everything it depends on.  A workgroup that has claimed a task waits (bounded) for that task's
// inputs; what it waits for was claimed earlier, hence by a workgroup that is resident and running -- no deadlock whatever
// number of the W workgroups the chip lets in, and none between the launches of different layers.
//
// Tiles and tasks.  Row i of the tile grid has width(i) = nblk - i + ntr tiles: the factor tiles (i, x), x = i .. nblk - 1, then
// the right-hand-side tiles.  Tile (i, x) receives the updates of block rows 0 .. i - 2 in chunks of at most L = CHAIN_L rows
// ([0, L), [L, 2L), ... -- the last one may be shorter) as bulk tasks, and block row i - 1 together with its final role
// (factor / substitute) as the chain task of step i, so that the serial piece of a step stays at K = 128.  Linear order,
// for s = 0 .. nblk - 1:
//     pre(s)    the LAST chunk of every tile of row s (rows .. s - 2; inputs: steps <= s - 2)           [s >= 2]
//     chain(s)  diagonal role, then the panel roles of row s (inputs: pre(s), step s - 1, the operator of s)
//     rest(s)   the full chunk [s - 1 - L, s - 1) of every tile of the rows below s                     [s - 1 a multiple of L, s > L]
// pre(s + 1) and chain(s + 1) do not depend on rest(s): the chain runs ahead of the chip-filling part of the previous
// steps as far as free workgroups let it, instead of waiting at a launch boundary.  With L = 4 a tile is read and written
// every fourth step (K = 512 per pass) instead of every second: about half the HBM traffic of the launch-per-step form.
//
// Hand-over between workgroups (no fences, as for the operator -- see flag_wait):
//   * a tile in progress (G / R in place) may be continued by any workgroup on any XCD: read and written with sc1 accesses;
//   * a finished tile of U is written ONCE into a buffer nobody has read in this launch: sc1 stores, plain loads (shared
//     through the XCD's L2); a finished right-hand-side tile is written in place: sc1 stores and sc1 loads;
//   * ver[i][x] (one word per tile) = block rows applied so far, i + 1 once the tile is final; a task polls the words of its
//     tile and of the finished tiles it multiplies with.  Every wave waits for its own stores (vmcnt(0)) before the barrier
//     after which thread 0 raises the word.
//   * ctl[1] is the launch's stop word: a wait that runs out sets it (and info[0]), every other wait of the launch returns at
//     once, the workgroups drain the counter without working, and the host takes its rank-revealing route as for a failed pivot.
#include "chain_order.h"

// lanes 0 .. cnt - 1 of wave 0 each poll one word until it reaches its target; -> false when the launch was stopped / timed out
__device__ __forceinline__ bool chain_wait(const int *word, int want, bool active, int *stop, int *info, int spin_limit) {
    for (int spin = 0;; ++spin) {
        const bool ok = !active || __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
        if (__all(ok)) return true;
        if ((spin & 15) == 15 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        if (spin >= spin_limit) {
            __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicCAS(info, 0, 0x7fffffff);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

#ifdef CP_CHAIN_STATS   // tools/ubench/chol_chain -DCP_CHAIN_STATS: [kind] cycles waited for inputs, [4 + kind] tasks, [8 + kind] cycles working
__device__ unsigned long long g_chain_stats[12];
#endif
__global__ void __launch_bounds__(PT, 4)
k_chol_chain(double *__restrict__ G, double *__restrict__ U, double *__restrict__ Lt, int ld, int nblk, int L, int t_begin, int total,
             int phase,
             const double *__restrict__ dg0, double piv_tol, double *__restrict__ TI, double *__restrict__ TIT, int *info, int *ctl,
             double *__restrict__ R, int ldr, int ntr, int spin_limit) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_task, s_go;
#ifdef CP_CHAIN_STATS
    __shared__ unsigned long long s_t0;
#endif
    const int tid = threadIdx.x;
    const ChainShape sh{nblk, ntr, L};
    const int XW = nblk + CHAIN_NTR_MAX;
    int *const stop = ctl + 1, *const ver = ctl + 8;
    for (;;) {
        if (tid == 0) s_task = t_begin + atomicAdd(ctl + 2 + phase, 1);    // the tasks [t_begin, total) of this launch
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(s_task);
        if (t >= total) return;
        const ChainTask k = chain_decode(sh, t);
        const int i = k.i, s = k.s;
        const bool rhs = k.xi >= nblk - i;
        const int j = rhs ? k.xi - (nblk - i) : i + k.xi;        // right-hand-side tile column, or block column of the factor
        const int x = rhs ? nblk + j : j;                         // column of the tile in the ver table
        // ---- wait for the inputs: this tile at r0 rows applied; for every block row r to apply, the finished tiles (r, i), (r, x)
        if (tid < 64) {
            const int lane = tid;
            const int *word = ver;
            int want = 0;
            const bool active = lane < 1 + 2 * k.kcnt;
            if (lane == 0) {
                word = ver + i * XW + x;
                want = k.r0;
            } else if (active) {
                const int r = k.r0 + ((lane - 1) >> 1);
                word = ver + r * XW + ((lane & 1) ? i : x);
                want = r + 1;
            }
#ifdef CP_CHAIN_STATS
            const unsigned long long w0 = __builtin_readcyclecounter();
#endif
            const bool go = chain_wait(word, want, active, stop, info, spin_limit);
#ifdef CP_CHAIN_STATS
            if (lane == 0) {
                atomicAdd(&g_chain_stats[k.kind], __builtin_readcyclecounter() - w0);          // cycles waited, by task kind
                atomicAdd(&g_chain_stats[4 + k.kind], 1ull);                                    // tasks
                s_t0 = __builtin_readcyclecounter();
            }
#endif
            if (lane == 0) s_go = go ? 1 : 0;
        }
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(s_go)) continue;      // stopped: drain the counter
        const double *Urow = U + size_t(k.r0) * NB * ld;
        const double *Ai = Urow + size_t(i) * NB;
        Tile t_;
        t_.kcnt = k.kcnt;
        if (rhs) {
            t_.T = R + size_t(i) * NB * ldr + size_t(j) * NB;
            t_.ldt = ldr;
            t_.B = R + size_t(k.r0) * NB * ldr + size_t(j) * NB;
            t_.ldb = ldr;
        } else {
            t_.T = G + size_t(i) * NB * ld + size_t(j) * NB;
            t_.ldt = ld;
            t_.B = Urow + size_t(j) * NB;
            t_.ldb = ld;
        }
        int done = k.r0 + k.kcnt;                                 // what ver[i][x] becomes
        if (k.kind != TASK_CHAIN) {
            role_bulk<true>(t_, Ai, ld, s, sm, rhs);
        } else {
            const double *Gss = G + size_t(s) * NB * ld + size_t(s) * NB;   // where the diagonal role leaves the operator of block s
            __builtin_amdgcn_s_setprio(3);
            if (!rhs && j == s)
                role_diag<true>(t_, Ai, U + size_t(s) * NB * ld + size_t(s) * NB, ld, s, dg0, piv_tol, TI + size_t(s) * NB * NB,
                                TIT + size_t(s) * NB * NB, info, sm);
            else if (!rhs)
                role_panel<true>(t_, Ai, ld, s, U + size_t(s) * NB * ld + size_t(j) * NB, ld,
                                 Lt + size_t(j) * NB * ld + size_t(s) * NB, Gss, info, spin_limit, sm, false, stop);
            else
                role_panel<true>(t_, Ai, ld, s, t_.T, ldr, nullptr, Gss, info, spin_limit, sm, true, stop);
            __builtin_amdgcn_s_setprio(0);
            done = s + 1;
        }
        // every wave's stores are acknowledged before the barrier; then the tile's word goes up
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CP_HANDOFF_RELEASE();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ver + i * XW + x, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CP_CHAIN_STATS
        if (tid == 0) atomicAdd(&g_chain_stats[8 + k.kind], __builtin_readcyclecounter() - s_t0);   // cycles working
#endif
        CP_HANDOFF_ACQUIRE();
    }
}

hipError_t lds_opt_in(int device) {   // > 64 KB of dynamic LDS needs an explicit opt-in, once per device
    static std::mutex mu;
    static bool done[64] = {};
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && device < 64 && done[device]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_step), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(LDS_DOUBLES * sizeof(double)));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(LDS_DOUBLES * sizeof(double)));
    if (e != hipSuccess) return e;
    if (device >= 0 && device < 64) done[device] = true;
    return e;
}

}  // namespace

namespace {
std::atomic<int> g_chain_form_override{-1};   // cp_debug_set_chol_form: -1 = what CP_CHOL_FORM says
int chain_form() {            // CP_CHOL_FORM=steps: the launch-per-step form (the reference form of the tests, A/B measurements)
    static const int v = [] {
        const char *e = getenv("CP_CHOL_FORM");
        return (e && !strcmp(e, "steps")) ? 0 : 1;
    }();
    const int o = g_chain_form_override.load(std::memory_order_relaxed);
    return o >= 0 ? o : v;
}
int chain_lazy() {
    static const int v = [] {
        const char *e = getenv("CP_CHOL_LAZY");
        const int x = e ? atoi(e) : CHAIN_L_MAX;
        return x < 1 ? 1 : (x > CHAIN_L_MAX ? CHAIN_L_MAX : x);
    }();
    return v;
}
int chain_wg_per_blk() {
    static const int v = [] {
        const char *e = getenv("CP_CHOL_WG_PER_BLK");
        const int x = e ? atoi(e) : 6;
        return x < 1 ? 1 : x;
    }();
    return v;
}
}  // namespace

// G (p_pad x p_pad, upper tiles valid, destroyed) = U^T U: U (upper, block rows), the off-diagonal blocks of Lt = U^T,
// TI_b = U_bb^-1, TIT_b = U_bb^-T per diagonal block.  info (zeroed by the caller's k_diag_prepare): [0] 1 + the first
// pivot <= piv_tol * its original diagonal dg0 (also NaN), [1 + b] block b factored.
// R (optional, p_pad x n_pad with n_pad % 128 == 0): overwritten with U^-T R, the forward substitution, in the same launches.
int cp_chol_factor_steps(cp_ctx *ctx, double *G, double *U, double *Lt, int ld, int nblk, const double *dg0,
                         double piv_tol, double *TI, double *TIT, int *info, double *R, int n_pad) {
    if (R && n_pad % NB) return cp_set_error(ctx, CP_ERR_ARG, "chol: right-hand sides need n_pad %% 128 == 0 (got %d)", n_pad);
    CP_HIP(ctx, lds_opt_in(ctx->device));
    const size_t lds = size_t(LDS_DOUBLES) * sizeof(double);
    const int ntr = R ? n_pad / NB : 0;
    int spin_limit = 1 << 26;
    if (ctx->chol_test_fail_flag_waits > 0) {   // test hook: the panel workgroups of THIS factorisation give up at once
        --ctx->chol_test_fail_flag_waits;
        spin_limit = 0;
    }
    const int form = chain_form(), lazy = chain_lazy();
    const int wg_per_blk = cp_knob(CP_KNOB_CHOL_WG) > 0 ? cp_knob(CP_KNOB_CHOL_WG) : chain_wg_per_blk();
    if (form == 1 && ntr <= CHAIN_NTR_MAX) {
        const ChainShape sh{nblk, ntr, lazy};
        // W: what the factorisation can keep busy on average, not what its widest step could use -- resident workgroups that
        // wait for the chain hold slots other layers' launches want (alone, a chain uses ~13 % of the matrix time of the slots
        // of its widest step).  The trailing matrix shrinks, so the task list is cut into `phases` launches at step boundaries,
        // each with the workgroups ITS rows can keep busy: the others' half CUs go back to the dispatcher in between.
        static const int phases_cfg = [] {
            const char *e = getenv("CP_CHOL_PHASES");
            const int v = e ? atoi(e) : 4;           // vgg16 job: 24.6 / 24.2 / 24.0 / 23.8 ms with 1 / 2 / 3 / 4 (and 6 / 4 / 5 / 6
            return v < 1 ? 1 : (v > 6 ? 6 : v);      // workgroups per block row), 24.1-24.2 ms with the launch-per-step form
        }();
        const int phases_want = cp_knob(CP_KNOB_CHOL_PHASES) > 0 ? std::min(cp_knob(CP_KNOB_CHOL_PHASES), 6) : phases_cfg;
        const int phases = std::max(1, std::min(phases_want, nblk / 4));       // at least four block rows per launch
        int t_begin = 0, s_begin = 0;
        for (int ph = 0; ph < phases; ++ph) {
            const int s_end = ph + 1 == phases ? nblk : (nblk * (ph + 1)) / phases;
            int t_end = t_begin;
            for (int s2 = s_begin; s2 < s_end; ++s2) t_end += sh.segment(s2);
            int W = wg_per_blk * (nblk - s_begin) + ntr;
            if (W > t_end - t_begin) W = t_end - t_begin;
            if (W > 512) W = 512;
            k_chol_chain<<<W, PT, lds, ctx->stream>>>(G, U, Lt, ld, nblk, lazy, t_begin, t_end, ph, dg0, piv_tol, TI, TIT, info,
                                                       info + cp_chol_ctl_offset(nblk), R, n_pad, ntr, spin_limit);
            CP_LAUNCH_CHECK(ctx);
            t_begin = t_end;
            s_begin = s_end;
        }
        return CP_OK;
    }
    for (int s = 0; s < nblk; ++s) {
        // block row s always; at even s >= 2 every tile below it (two block rows of updates at once), at odd s block row s + 1
        const int n = nblk - s;
        int tiles = n + ntr;
        if (s >= 2 && !(s & 1)) tiles = n * (n + 1) / 2 + n * ntr;
        else if ((s & 1) && n > 1) tiles += (n - 1) + ntr;
        k_chol_step<<<tiles, PT, lds, ctx->stream>>>(G, U, Lt, ld, nblk, s, dg0, piv_tol, TI, TIT, info, R, n_pad, ntr, spin_limit);
        CP_LAUNCH_CHECK(ctx);
    }
    return CP_OK;
}

// cp_debug_set_chol_form: which form the factorisations launched from now on take, process-wide -- 1 the persistent launch
// (k_chol_chain), 0 one launch per 128-column step (k_chol_step), -1 back to the default / CP_CHOL_FORM.  Results are bit for
// bit the same; bench.py's A/B leg and the tests use it.  Returns the form now in force.
extern "C" int cp_debug_set_chol_form(int form) {
    g_chain_form_override.store(form < 0 ? -1 : (form ? 1 : 0), std::memory_order_relaxed);
    return chain_form();
}

// Test hooks (not part of the public ABI).  cp_debug_chol_fail_flag_wait: the next `count` factorisations on this context run
// with a spin limit of 0 -- their panel workgroups do not wait for the diagonal workgroup's flag, report the time-out in
// info[0] and go on with whatever they find: the real time-out path of flag_wait.
extern "C" int cp_debug_chol_fail_flag_wait(cp_ctx *ctx, int count) {
    if (!ctx) return CP_ERR_ARG;
    ctx->chol_test_fail_flag_waits = count;
    return CP_OK;
}

namespace {
__global__ void __launch_bounds__(256) k_lds_hog(unsigned long long ticks_100mhz, int *sink) {
    extern __shared__ __attribute__((aligned(16))) double hog_sm[];
    hog_sm[threadIdx.x] = double(threadIdx.x);     // the allocation is what matters: nothing else fits on the CU meanwhile
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(32);
    if (hog_sm[(threadIdx.x + 1) & 255] < 0.0) sink[0] = 1;
}
}  // namespace

// cp_debug_lds_hog: n_wg workgroups that each hold lds_bytes of LDS (up to 160 KB: a whole CU) for `usec` microseconds on
// the context's stream -- a filler that keeps LDS-hungry workgroups of other launches (k_chol_step: 75 KB) off the CUs.
extern "C" int cp_debug_lds_hog(cp_ctx *ctx, int lds_bytes, int usec, int n_wg) {
    if (!ctx || lds_bytes < 2048 || lds_bytes > 160 * 1024 || usec < 0 || n_wg <= 0) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    static std::mutex mu;
    {
        std::lock_guard<std::mutex> lock(mu);
        CP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (cp_arena_reserve(ctx, 4096) != CP_OK) return CP_ERR_NOMEM;
    k_lds_hog<<<n_wg, 256, size_t(lds_bytes), ctx->stream>>>(static_cast<unsigned long long>(usec) * 100ull,
                                                              reinterpret_cast<int *>(ctx->arena));
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}
