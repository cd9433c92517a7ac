// Coordinate-descent LASSO on the Gram matrix: the device form of
// sklearn.linear_model._cd_fast.enet_coordinate_descent_gram (_cd_fast.pyx:564-737), which is
// what Lasso.fit(Z, reY) inside solve() (lib/decompose.py:453-466) reduces to once
// Q = Zc^T Zc, q = Zc^T yc are available (cp_lasso_gram).  The coordinate order is sklearn's
// our_rand_r xorshift32 (sklearn/utils/_random.pxd:20-35) so the visit sequence, the epoch
// count and therefore the zero pattern of w are those of the reference.
//
// Execution model.  The sweep is a strictly sequential dependency chain (every coordinate
// update needs H = Q w as left by the previous one), so it is latency- not bandwidth- or
// MFMA-bound.  ONE wavefront runs the whole fit (or the whole alpha search): no barriers, no
// inter-workgroup traffic.
//   * H lives in registers: lane l holds H[l + 64 r], r < R (R = ceil(c/64), template).
//   * The visit order does not depend on the data, so the RNG runs D steps ahead on the
//     scalar unit and row Q[ii,:] of every future step is already in flight (register ring of
//     D slots, L2-resident Q) when the chain reaches it.
//   * w, q and diag(Q) sit in LDS; the scalar operands of a step (w_ii, q_ii, Q_ii) are
//     uniform ds_reads issued one step early, H_ii comes out of the register file with one
//     v_readlane pair.
//   * Every multiply-add that sklearn's daxpy performs is an explicit fma, in the same order,
//     so w is bit-identical to oracle/cd_oracle.c::cpo_enet_cd_gram (the epoch-end dual-gap
//     reductions are tree-ordered and only feed the stopping test).
#include "cp_common.h"
#include "xorshift_jump.h"
#include "cd_shared.h"

// tuning switches (tools/cd_bench.py builds variants with -D...)
#ifndef CD_UNCOND_R
#define CD_UNCOND_R 4  // R <= this: the "!= 0" guards of the two daxpy are dropped
#endif
#ifndef CD_TOUCH
#define CD_TOUCH 1     // 1: one explicit wait per step for the slot's rows
#endif
#ifndef CD_BRANCH_SEL
#define CD_BRANCH_SEL 1  // 1: w_cur patch through a (rarely taken) scalar branch
#endif

namespace {

using namespace cdk;

template <int R>
struct Ring {
    // steps per register set; two sets ping-pong, so a row is requested D..2D steps before use
    static constexpr int D = R <= 2 ? 8 : (R <= 8 ? 4 : (R <= 16 ? 2 : 1));
};

// Everything step t needs except w_ii and H_ii, fetched >= D steps early.
template <int R>
struct Slot {
    int ii;
    double row[R];  // Q[ii, lane + 64 r]
    double qi, Qii, den;
};

// One fit.  w_lds holds the warm start on entry and the solution on exit.
// feat[4 j + {0,1,2}] = { q[j], Q[j,j], Q[j,j] + beta (or its reciprocal when RECIP) }.
// RECIP: multiply by 1/(Qii + beta) instead of dividing (CP_CD_RECIPROCAL).
// ALIGNED: c is a multiple of 2*D, so an epoch ends exactly on a group boundary and the
// end-of-epoch test is hoisted out of the per-step code.
template <int R, bool RECIP, bool ALIGNED, bool DELTA>
__device__ __forceinline__ FitOut cd_fit(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                         uint32_t seed, int max_iter, double tol_scaled, double d_w_tol,
                                         double y_norm2, double *w_lds, const double *feat) {
    constexpr int D = Ring<R>::D;
    const int lane = threadIdx.x;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    // lanes past the last column re-read column c-1: their H entries are never consumed
    uint32_t colb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = r * WAVE + lane;
        colb[r] = uint32_t(col < c ? col : c - 1) * 8u;
    }
    double H[R];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = 0.0;

    // H = Q w, accumulated row by row in index order (cpo_enet_cd_gram does the same).
    constexpr int U = R <= 4 ? 8 : (R <= 8 ? 4 : (R <= 16 ? 2 : 1));
    for (int j0 = 0; j0 < c; j0 += U) {
        double row[U][R];
        double wj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            wj[u] = j < c ? w_lds[j] : 0.0;
            const uint32_t roff = uint32_t(j < c ? j : c - 1) * row_stride_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) row[u][r] = load_q(rsrc, colb[r], roff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (wj[u] != 0.0) {
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(wj[u], row[u][r], H[r]);
            }
    }

    IdxStream rng;
    rng.init(seed, uint32_t(c), row_stride_bytes, lane);
    int pos = 0;  // first unconsumed value of the current 64-value index batch (multiple of D)

    Slot<R> A[D], B[D];
    auto fill = [&](Slot<R>(&S)[D]) {  // request the next D steps' operands
        if (pos == 64) {
            rng.next_batch();
            pos = 0;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            int ii;
            uint32_t roff;
            rng.take(pos + d, ii, roff);
            S[d].ii = ii;
#pragma unroll
            for (int r = 0; r < R; ++r) S[d].row[r] = load_q(rsrc, colb[r], roff);
            const double2 qQ = *reinterpret_cast<const double2 *>(feat + 4 * ii);
            S[d].qi = qQ.x;
            S[d].Qii = qQ.y;
            S[d].den = feat[4 * ii + 2];
        }
        pos += D;
    };
    // Make the compiler drain a set's loads here (they were issued >= D steps ago), so that no
    // load is in flight across the loop back-edge: a loop-carried pending load would be waited
    // for with vmcnt(0) at its first use, which would also wait for the set requested just
    // before it.
    auto settle = [&](Slot<R>(&S)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(S[d].row[r]));
        }
    };

    FitOut out;
    out.gap = tol_scaled + 1.0;
    out.n_iter = 0;
    int n_iter = 0, f = 0;
    double w_max = 0.0, d_w_max = 0.0;

    // end of epoch, _cd_fast.pyx:684-731; returns true when the fit is finished
    auto epoch_end = [&]() -> bool {
        bool done = false;
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            double s_qw = 0, s_wh = 0, s_ww = 0, s_l1 = 0, m_xta = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int col = r * WAVE + lane;
                if (col < c) {
                    const double wv = w_lds[col], qv = feat[4 * col];
                    const double xta = qv - H[r] - beta * wv;
                    s_qw += wv * qv;
                    s_wh += wv * H[r];
                    s_ww += wv * wv;
                    s_l1 += fabs(wv);
                    m_xta = fmax(m_xta, fabs(xta));
                }
            }
            const double q_dot_w = wave_sum(s_qw), wh = wave_sum(s_wh), w_norm2 = wave_sum(s_ww),
                         l1 = wave_sum(s_l1), dual_norm = wave_max(m_xta);
            const double R_norm2 = y_norm2 + wh - 2.0 * q_dot_w;
            double const_, gap;
            if (dual_norm > alpha) {
                const_ = alpha / dual_norm;
                const double A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += alpha * l1 - const_ * y_norm2 + const_ * q_dot_w + 0.5 * beta * (1.0 + const_ * const_) * w_norm2;
            out.gap = gap;
            if (gap < tol_scaled) done = true;
        }
        ++n_iter;
        w_max = 0.0;
        d_w_max = 0.0;
        f = 0;
        return done || n_iter == max_iter;
    };

    fill(A);
    double w_cur = w_lds[A[0].ii];  // w_ii of the step about to run

    // One coordinate update (_cd_fast.pyx:644-682); `nx` is the slot of the following step.
    // The two daxpy of the original are guarded by "!= 0" tests there; an fma with a zero
    // multiplier returns its addend unchanged, so for narrow problems (R <= 2) the guards are
    // dropped (same values, fewer instructions on the serial path); wider ones branch.
    auto step = [&](const Slot<R> &S, const Slot<R> &nx) {
        const int ii = S.ii;
        const double w_pre = w_lds[nx.ii];  // issued early; patched below if nx.ii == ii
        const double Qii = S.Qii;
#if CD_TOUCH
        double rowv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) rowv[r] = S.row[r];
        asm volatile("" : "+v"(rowv[R - 1]));
#else
        const double *rowv = S.row;
#endif
        if (Qii != 0.0) {  // _cd_fast.pyx:651
            const double w_ii = w_cur;
            double hsel = H[0];
            if (R > 1) {
                const int r_ii = ii >> 6;
#pragma unroll
                for (int r = 1; r < R; ++r) hsel = (r == r_ii) ? H[r] : hsel;
            }
            const double H_ii = fma(-w_ii, Qii, read_lane(hsel, ii & 63));
            const double tmp = S.qi - H_ii;
            // fsign(tmp) * fmax(|tmp| - alpha, 0)
            const double thr = copysign(fmax(fabs(tmp) - alpha, 0.0), tmp);
            const double w_new = RECIP ? thr * S.den : thr / S.den;
            if (DELTA) {  // CP_CD_DELTA: one axpy with the difference
                const double dlt = w_new - w_ii;
                if (R <= CD_UNCOND_R || dlt != 0.0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(dlt, rowv[r], H[r]);
                }
            } else if (R <= CD_UNCOND_R) {
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(w_new, rowv[r], fma(-w_ii, rowv[r], H[r]));
            } else {
                if (w_ii != 0.0) {  // H -= w_ii * Q[ii]
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(-w_ii, rowv[r], H[r]);
                }
                if (w_new != 0.0) {  // H += w[ii] * Q[ii]
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(w_new, rowv[r], H[r]);
                }
            }
            if (lane == 0) w_lds[ii] = w_new;
            d_w_max = fmax(d_w_max, fabs(w_new - w_ii));
            w_max = fmax(w_max, fabs(w_new));
#if CD_BRANCH_SEL
            w_cur = w_pre;
            if (__builtin_expect(nx.ii == ii, 0)) w_cur = w_new;
#else
            w_cur = (nx.ii == ii) ? w_new : w_pre;
#endif
        } else {
            w_cur = w_pre;
        }
    };

    for (;;) {
        fill(B);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            step(A[d], d + 1 < D ? A[d + 1] : B[0]);
            if (!ALIGNED)
                if (++f == c)
                    if (epoch_end()) goto fit_done;
        }
        fill(A);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            step(B[d], d + 1 < D ? B[d + 1] : A[0]);
            if (!ALIGNED)
                if (++f == c)
                    if (epoch_end()) goto fit_done;
        }
        settle(A);
        if (ALIGNED) {
            f += 2 * D;
            if (f == c)
                if (epoch_end()) goto fit_done;
        }
    }
fit_done:
    out.n_iter = n_iter;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = r * WAVE + lane;
        cnt += (col < c && w_lds[col] != 0.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
    out.nnz = cnt;
    return out;
}

// ---- LDS hand-off primitives shared by the two-wave forms (see cd_fit_duo / cd_fit_assist below) ----
struct DuoCtl {
    int seqA;   // blocks published by the chain wave
    int seqB;   // images published by the keeper (image k = H after the first k blocks; count = k + 1)
    int batB;   // index batches published by the keeper
    int stop;   // chain -> keeper: the fit is over
    int err;    // a bounded wait ran out (never in a correct run)
    int n_iter, nnz, pad;
    double gap;
};

// wait until *p >= need (returns false if `stop` was raised or the bound ran out)
__device__ __forceinline__ bool duo_wait(int *p, int need, DuoCtl *ctl, bool watch_stop) {
    for (int spin = 0;; ++spin) {
        if (duo_load(p) >= need) return true;
        if (watch_stop && duo_load(&ctl->stop)) return false;
        if (spin > (1 << 22)) {
            duo_store(&ctl->err, 1);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int R>
struct DuoLds {
    static constexpr int B = 8, IMG = R * WAVE;
    double *img;       // [2][IMG]
    double *pub;       // [4][2 * B]
    uint32_t *ii;      // [3][64]
    uint64_t *dup;     // [4] lanes whose coordinate repeats inside their block
    uint64_t *xdup;    // [4] lanes whose coordinate also occurs in the block before theirs
    DuoCtl *ctl;
    static __host__ __device__ constexpr int doubles() { return 2 * IMG + 4 * 2 * B + 3 * 32 + 8 + int(sizeof(DuoCtl) / 8) + 2; }
    __device__ void bind(double *base) {
        img = base;
        pub = img + 2 * IMG;
        ii = reinterpret_cast<uint32_t *>(pub + 4 * 2 * B);
        dup = reinterpret_cast<uint64_t *>(ii + 3 * 64);
        xdup = dup + 4;
        ctl = reinterpret_cast<DuoCtl *>(xdup + 4);
    }
};


// ---------------------------------------------------------------------------------------------
// Blocked variant (c % B == 0, R <= 8).  A single wave issues one instruction every ~5-6.5 cycles
// regardless of dependencies, so a step costs what its instruction count costs.  Here the
// per-coordinate scalars of B consecutive steps live on B lanes: at block start every lane
// fetches "its" H[ii] (through an LDS image of H), w[ii], and -- one block ahead -- the B
// entries Q[ii_a, ii_lane] that couple it to the block's steps; the B updates then run as
// straight vector code (lane a's result is latched at step a) with ONE readlane pair per step
// to broadcast the step's (w_old, w_new) -- or just their difference with CP_CD_DELTA -- to the
// full-width axpy on H.  No per-step register selects, LDS round trips or scalar bookkeeping.
// The fma sequence applied to every H entry is unchanged, so w stays bit-identical to the oracle.
// Blocks in which a coordinate repeats (detected per 64-value index batch) take a scalar path.
template <int R>
struct Blk {
    static constexpr int B = R <= 4 ? 8 : 4;
};

template <int R, int B>
struct BSet {
    double row[B][R];  // Q[ii_a, lane + 64 r]
    double qc[B];      // Q[ii_a, ii_lane]
};

// ASSIST: a second wave of the workgroup (cd_assist_wave) runs the index stream and the duplicate scan and hands
// every 64-value batch over through the LDS ring of `L`; this wave then spends ~30 instead of ~190 instructions
// per batch on it.
template <int R, bool RECIP, bool DELTA, bool ASSIST = false>
__device__ __forceinline__ FitOut cd_fit_blocked(const double *__restrict__ Q, int ldq, int c, double alpha,
                                                 double beta, uint32_t seed, int max_iter, double tol_scaled,
                                                 double d_w_tol, double y_norm2, double *w_lds, const double *feat,
                                                 double *h_lds, DuoLds<R> *L = nullptr) {
    constexpr int B = Blk<R>::B, NBLK = 64 / B;
    const int lane = threadIdx.x;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    // register r of lane l holds column colof(r): 64 r + l, or -- packed -- 128 (r/2) + 2 l + (r&1), so
    // that one 16-byte load fetches two of a lane's row elements
    constexpr bool PK = CP_CD_PACKED && (R % 2 == 0);
    auto colof = [&](int r) -> int { return PK ? (r >> 1) * 2 * WAVE + 2 * lane + (r & 1) : r * WAVE + lane; };
    uint32_t colb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = colof(r);
        colb[r] = uint32_t(col < c ? col : (PK ? c - 2 + (r & 1) : c - 1)) * 8u;
    }
    auto load_row = [&](double (&dst)[R], uint32_t roff) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < R; r += 2) load_q2(rsrc, colb[r], roff, dst[r], dst[r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = load_q(rsrc, colb[r], roff);
        }
    };
    double H[R];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = 0.0;
    constexpr int U = R <= 4 ? 8 : 4;
    for (int j0 = 0; j0 < c; j0 += U) {  // H = Q w in index order (as the oracle)
        double row[U][R];
        double wj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            wj[u] = j < c ? w_lds[j] : 0.0;
            const uint32_t roff = uint32_t(j < c ? j : c - 1) * row_stride_bytes;
            load_row(row[u], roff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (wj[u] != 0.0) {
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(wj[u], row[u][r], H[r]);
            }
    }

    IdxStream rng;
    int kb = 0;  // ASSIST: index of the batch `rng.idx / rng.off` describe
    auto take_batch = [&]() {  // ASSIST: batch kb from the ring
        duo_wait(&L->ctl->batB, kb + 1, L->ctl, false);
        rng.idx = L->ii[(kb % 3) * 64 + lane];
        rng.off = rng.idx * row_stride_bytes;
        duo_store(&L->ctl->seqA, kb + 1);  // slot kb % 3 may be reused
    };
    if (ASSIST)
        take_batch();
    else
        rng.init(seed, uint32_t(c), row_stride_bytes, lane);

    // Coupling loads of step a are made out of range (-> 0.0) for the lanes at or before position a of
    // their block: a lane's private H[ii] then stops changing once its own step is over, so the value
    // its update chain recomputes in the remaining steps IS its final coefficient -- no per-step latch.
    const uint32_t rel = uint32_t(lane) & uint32_t(B - 1);
    constexpr uint32_t OOB = 0x80000000u;  // beyond num_records with or without the row offset, no 32-bit wrap
    auto mask_offsets = [&](uint32_t voff, uint32_t (&vm)[B]) {
#pragma unroll
        for (int a = 0; a < B; ++a) vm[a] = rel > uint32_t(a) ? voff : OOB;
    };
    // per-lane data of the CURRENT 64-value batch (lane l <-> stream value 64*batch + l)
    uint32_t ii_v, voff_v;
    uint32_t vm_cur[B], vm_nxt[B];
    double q_v, Qd_v, den_v;
    uint64_t dupmask;
    auto adopt_batch = [&]() {
        ii_v = rng.idx;
        voff_v = ii_v * 8u;
        mask_offsets(voff_v, vm_cur);
        const double2 qQ = *reinterpret_cast<const double2 *>(feat + 4 * ii_v);
        q_v = qQ.x;
        Qd_v = qQ.y;
        den_v = feat[4 * ii_v + 2];
        if (ASSIST) {
            dupmask = L->dup[kb & 3];
        } else {
            bool dup = false;
#pragma unroll
            for (int sft = 1; sft < B; ++sft) {
                const int other = __shfl(int(ii_v), (lane & ~(B - 1)) | ((lane + sft) & (B - 1)), WAVE);
                dup |= (uint32_t(other) == ii_v);
            }
            dupmask = __ballot(dup);
        }
    };

    BSet<R, B> SA, SB;
    // request the operands of block (base .. base+B-1) of the batch whose offsets are in (off_vec, voff_vec)
    auto fill = [&](BSet<R, B> &S, uint32_t off_vec, const uint32_t (&vm)[B], int base) {
#pragma unroll
        for (int a = 0; a < B; ++a) {
            const uint32_t roff = uint32_t(__builtin_amdgcn_readlane(int(off_vec), base + a));
            load_row(S.row[a], roff);
            S.qc[a] = load_q(rsrc, vm[a], roff);
        }
    };
    auto settle = [&](BSet<R, B> &S) {
#pragma unroll
        for (int a = 0; a < B; ++a) {
            asm volatile("" : "+v"(S.qc[a]));
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(S.row[a][r]));
        }
    };

    FitOut out;
    out.gap = tol_scaled + 1.0;
    out.n_iter = 0;
    int n_iter = 0, f = 0;
    double wmax_v = 0.0, dmax_v = 0.0;  // per-lane running maxima, reduced at the end of an epoch

    auto epoch_end = [&]() -> bool {
        const double w_max = wave_max(wmax_v), d_w_max = wave_max(dmax_v);
        bool done = false;
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            double s_qw = 0, s_wh = 0, s_ww = 0, s_l1 = 0, m_xta = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int col = colof(r);
                if (col < c) {
                    const double wv = w_lds[col], qv = feat[4 * col];
                    const double xta = qv - H[r] - beta * wv;
                    s_qw += wv * qv;
                    s_wh += wv * H[r];
                    s_ww += wv * wv;
                    s_l1 += fabs(wv);
                    m_xta = fmax(m_xta, fabs(xta));
                }
            }
            const double q_dot_w = wave_sum(s_qw), wh = wave_sum(s_wh), w_norm2 = wave_sum(s_ww),
                         l1 = wave_sum(s_l1), dual_norm = wave_max(m_xta);
            const double R_norm2 = y_norm2 + wh - 2.0 * q_dot_w;
            double const_, gap;
            if (dual_norm > alpha) {
                const_ = alpha / dual_norm;
                const double A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += alpha * l1 - const_ * y_norm2 + const_ * q_dot_w + 0.5 * beta * (1.0 + const_ * const_) * w_norm2;
            out.gap = gap;
            if (gap < tol_scaled) done = true;
        }
        ++n_iter;
        wmax_v = 0.0;
        dmax_v = 0.0;
        f = 0;
        return done || n_iter == max_iter;
    };

    // B coordinate updates (_cd_fast.pyx:644-682) for the lanes base..base+B-1 of the current batch
    auto compute = [&](const BSet<R, B> &S, int base) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < R; r += 2)
                *reinterpret_cast<double2 *>(h_lds + r * WAVE + 2 * lane) = make_double2(H[r], H[r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) h_lds[r * WAVE + lane] = H[r];
        }
        double Hs_v = h_lds[ii_v];   // H[ii_lane] as of now
        double wo_v = w_lds[ii_v];   // w[ii_lane]
        double wn_keep = 0.0;        // lane a's new coefficient, latched at step a
        const uint64_t blockmask = ((uint64_t(1) << B) - 1) << base;
        uint64_t wmask = blockmask;  // lanes that write their coefficient back (later duplicate wins)
        const bool has_dup = (dupmask & blockmask) != 0;
        if (!has_dup) {
            double wn_v = 0.0;
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                // every lane evaluates "its" update against its private H; lane la's is the one that counts now,
                // lanes before it reproduce their final value (their couplings to this and later steps read 0)
                const double Hp = fma(-wo_v, Qd_v, Hs_v);
                const double tmp = q_v - Hp;
                const double thr = copysign(fmax(fabs(tmp) - alpha, 0.0), tmp);
                wn_v = RECIP ? thr * den_v : thr / den_v;
                if (DELTA) {
                    const double d_a = read_lane(wn_v - wo_v, la);
                    Hs_v = fma(d_a, S.qc[a], Hs_v);
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(d_a, S.row[a][r], H[r]);
                } else {
                    const double wo_a = read_lane(wo_v, la), wn_a = read_lane(wn_v, la);
                    Hs_v = fma(wn_a, S.qc[a], fma(-wo_a, S.qc[a], Hs_v));
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(wn_a, S.row[a][r], fma(-wo_a, S.row[a][r], H[r]));
                }
            }
            wn_keep = wn_v;
        } else {  // a coordinate repeats inside the block: later visits must see the earlier result
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                const double Hp = fma(-wo_v, Qd_v, Hs_v);
                const double tmp = q_v - Hp;
                const double thr = copysign(fmax(fabs(tmp) - alpha, 0.0), tmp);
                const double wn_v = RECIP ? thr * den_v : thr / den_v;
                wn_keep = lane == la ? wn_v : wn_keep;
                const double wo_a = read_lane(wo_v, la), wn_a = read_lane(wn_v, la);
                if (DELTA) {
                    const double d_a = read_lane(wn_v - wo_v, la);
                    Hs_v = fma(d_a, S.qc[a], Hs_v);
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(d_a, S.row[a][r], H[r]);
                } else {
                    Hs_v = fma(wn_a, S.qc[a], fma(-wo_a, S.qc[a], Hs_v));
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(wn_a, S.row[a][r], fma(-wo_a, S.row[a][r], H[r]));
                }
                const uint32_t ii_a = uint32_t(__builtin_amdgcn_readlane(int(ii_v), la));
                const bool later_same = ii_v == ii_a && lane > la && lane < base + B;
                wo_v = later_same ? wn_a : wo_v;
                if (__ballot(later_same) != 0) wmask &= ~(uint64_t(1) << la);
            }
        }
        if ((blockmask >> lane) & 1) {
            dmax_v = fmax(dmax_v, fabs(wn_keep - wo_v));
            wmax_v = fmax(wmax_v, fabs(wn_keep));
        }
        if ((wmask >> lane) & 1) w_lds[ii_v] = wn_keep;
    };

    adopt_batch();
    fill(SA, rng.off, vm_cur, 0);
    for (;;) {  // one 64-value batch per iteration
        for (int g = 0; g < NBLK; g += 2) {
            fill(SB, rng.off, vm_cur, (g + 1) * B);
            compute(SA, g * B);
            f += B;
            if (f == c)
                if (epoch_end()) goto fit_done;
            if (g + 2 < NBLK) {
                fill(SA, rng.off, vm_cur, (g + 2) * B);
                compute(SB, (g + 1) * B);
            } else {  // last block of the batch: request block 0 of the next batch first
                if (ASSIST) {  // only rng.idx / rng.off change; ii_v etc. still describe this batch
                    ++kb;
                    take_batch();
                } else {
                    rng.next_batch();
                }
                mask_offsets(rng.idx * 8u, vm_nxt);
                fill(SA, rng.off, vm_nxt, 0);
                compute(SB, (g + 1) * B);
            }
            f += B;
            if (f == c)
                if (epoch_end()) goto fit_done;
            settle(SA);
        }
        adopt_batch();
    }
fit_done:
    out.n_iter = n_iter;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = colof(r);
        cnt += (col < c && w_lds[col] != 0.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
    out.nnz = cnt;
    return out;
}

// run-time -> compile-time dispatch.  flags: CP_CD_RECIPROCAL | CP_CD_DELTA.
// debug/bench aid: shader-clock cycles spent inside the last cd_fit of the last launch and the
// number of coordinate steps it ran (read back by cp_debug_cd_cycles; not part of the public ABI)
__device__ unsigned long long g_cd_debug[8];

// ---------------------------------------------------------------------------------------------
// Two-wave form of the blocked variant (c % 8 == 0, R <= 8; 128-thread workgroup).  A single wave
// issues one instruction per ~5-6.5 cycles whatever the dependencies, so the step costs its
// instruction count; here the count is split between two waves on two SIMDs of the CU:
//   chain wave  (wave 0): the scalar recurrence only -- per step the 7-op soft-threshold chain, one
//                readlane pair and one fma on the lanes' private H[ii]; it never touches the full H.
//   keeper wave (wave 1): owns H in registers, fetches the rows Q[ii,:], applies the block's 8 axpys
//                with the differences the chain wave publishes, and exposes H as an LDS image after
//                every block; it also runs the index stream (xorshift jump-ahead, duplicate scan).
// The chain wave runs one block ahead of the keeper: the private H[ii] of block t+1 is taken from
// the image after block t-1 plus the 8 updates of block t through the couplings Q[ii_a(t), ii_l(t+1)]
// (same fma sequence as the keeper applies to that element, so w stays bit-identical).  Hand-offs are
// sequence counters in LDS (acquire/release at workgroup scope), polled with a bound.
template <int R, bool RECIP, bool DELTA>
__device__ __forceinline__ void duo_keeper(const double *__restrict__ Q, int ldq, int c, uint32_t seed, const double *w_lds,
                                           DuoLds<R> &L) {
    constexpr int B = 8;
    const int lane = threadIdx.x & 63;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    constexpr bool PK = CP_CD_PACKED && (R % 2 == 0);
    auto colof = [&](int r) -> int { return PK ? (r >> 1) * 2 * WAVE + 2 * lane + (r & 1) : r * WAVE + lane; };
    uint32_t colb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = colof(r);
        colb[r] = uint32_t(col < c ? col : (PK ? c - 2 + (r & 1) : c - 1)) * 8u;
    }
    auto load_row = [&](double (&dst)[R], uint32_t roff) {
        if (PK) {
#pragma unroll
            for (int r = 0; r < R; r += 2) load_q2(rsrc, colb[r], roff, dst[r], dst[r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = load_q(rsrc, colb[r], roff);
        }
    };
    double H[R];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = 0.0;
    constexpr int U = R <= 4 ? 8 : 4;
    for (int j0 = 0; j0 < c; j0 += U) {  // H = Q w in index order (as the oracle)
        double row[U][R];
        double wj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            wj[u] = j < c ? w_lds[j] : 0.0;
            load_row(row[u], uint32_t(j < c ? j : c - 1) * row_stride_bytes);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (wj[u] != 0.0) {
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(wj[u], row[u][r], H[r]);
            }
    }
    auto write_image = [&](int k) {
        double *im = L.img + (k & 1) * DuoLds<R>::IMG;
        if (PK) {
#pragma unroll
            for (int r = 0; r < R; r += 2) *reinterpret_cast<double2 *>(im + r * WAVE + 2 * lane) = make_double2(H[r], H[r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) im[r * WAVE + lane] = H[r];
        }
        duo_store(&L.ctl->seqB, k + 1);
    };
    write_image(0);

    IdxStream rng;
    rng.init(seed, uint32_t(c), row_stride_bytes, lane);
    uint32_t prev_idx = 0xffffffffu;  // coordinates of the batch published before (none yet)
    auto publish_batch = [&](int k) {
        L.ii[(k % 3) * 64 + lane] = rng.idx;
        bool dup = false, xd = false;
        const int bs = lane & ~(B - 1);
#pragma unroll
        for (int sft = 1; sft < B; ++sft) {
            const int other = __shfl(int(rng.idx), bs | ((lane + sft) & (B - 1)), WAVE);
            dup |= (uint32_t(other) == rng.idx);
        }
#pragma unroll
        for (int sft = 0; sft < B; ++sft) {  // the block before: lanes bs-8.. of this batch, or 56.. of the previous one
            const int src = ((bs - B) & 63) + sft;
            const int o_same = __shfl(int(rng.idx), src, WAVE), o_prev = __shfl(int(prev_idx), src, WAVE);
            xd |= uint32_t(bs == 0 ? o_prev : o_same) == rng.idx;
        }
        const uint64_t m = __ballot(dup), mx = __ballot(xd);
        if (lane == 0) {
            L.dup[k & 3] = m;
            L.xdup[k & 3] = mx;
        }
        prev_idx = rng.idx;
        duo_store(&L.ctl->batB, k + 1);
    };
    publish_batch(0);  // the ring runs two batches ahead of the one being applied
    uint32_t off_cur = rng.off;
    rng.next_batch();
    publish_batch(1);
    uint32_t off_nxt = rng.off;
    rng.next_batch();
    publish_batch(2);
    uint32_t off_n2 = rng.off;
    int batch = 0;

    double rowA[B][R], rowB[B][R];
    auto fill = [&](double (&S)[B][R], uint32_t off_vec, int base) {
#pragma unroll
        for (int a = 0; a < B; ++a) load_row(S[a], uint32_t(__builtin_amdgcn_readlane(int(off_vec), base + a)));
    };
    auto settle = [&](double (&S)[B][R]) {
#pragma unroll
        for (int a = 0; a < B; ++a)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(S[a][r]));
    };
    // apply block t with the rows in S; returns false when the fit is over
    unsigned long long waitB = 0, blocksB = 0;
    auto apply = [&](const double (&S)[B][R], int t) -> bool {
        const unsigned long long w0 = __builtin_readcyclecounter();
        const bool okw = duo_wait(&L.ctl->seqA, t + 1, L.ctl, true);
        waitB += __builtin_readcyclecounter() - w0;
        ++blocksB;
        if (!okw) {
            if (lane == 0) {
                g_cd_debug[4] = waitB;
                g_cd_debug[5] = blocksB;
            }
            return false;
        }
        const double *pb = L.pub + (t & 3) * 2 * B;
#pragma unroll
        for (int a = 0; a < B; ++a) {
            if (DELTA) {
                const double d_a = pb[a];  // same address in every lane: LDS broadcast
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(d_a, S[a][r], H[r]);
            } else {
                const double wo_a = pb[2 * a], wn_a = pb[2 * a + 1];
#pragma unroll
                for (int r = 0; r < R; ++r) H[r] = fma(wn_a, S[a][r], fma(-wo_a, S[a][r], H[r]));
            }
        }
        write_image(t + 1);
        return true;
    };
    fill(rowA, off_cur, 0);
    for (int t = 0;; t += 2) {  // two blocks per iteration (register sets A / B); 8 blocks per batch
        const int g = t & 7;
        fill(rowB, off_cur, (g + 1) * B);
        if (!apply(rowA, t)) break;
        if (g + 2 < 8) {
            fill(rowA, off_cur, (g + 2) * B);
        } else {
            fill(rowA, off_nxt, 0);
        }
        if (!apply(rowB, t + 1)) break;
        settle(rowA);
        if (g + 2 >= 8) {  // batch roll-over
            off_cur = off_nxt;
            off_nxt = off_n2;
            rng.next_batch();
            ++batch;
            publish_batch(batch + 2);
            off_n2 = rng.off;
        }
    }
}

template <int R, bool RECIP, bool DELTA>
__device__ __forceinline__ void duo_chain(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                          int max_iter, double tol_scaled, double d_w_tol, double y_norm2, double *w_lds,
                                          const double *feat, DuoLds<R> &L) {
    constexpr int B = 8;
    const int lane = threadIdx.x & 63;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    const uint32_t rel = uint32_t(lane) & uint32_t(B - 1);
    constexpr uint32_t OOB = 0x80000000u;
    DuoCtl *ctl = L.ctl;

    struct Batch {
        uint32_t ii, off, voff;
        uint32_t vm[B];  // coupling column offsets, out of range for lanes at or before position a of their block
        double q, Qd, den;
        uint64_t dupmask, xdupmask;
    };
    auto load_batch = [&](Batch &bt, int k) -> bool {
        if (!duo_wait(&ctl->batB, k + 1, ctl, false)) return false;
        bt.ii = L.ii[(k % 3) * 64 + lane];
        bt.dupmask = L.dup[k & 3];
        bt.xdupmask = L.xdup[k & 3];
        bt.off = bt.ii * row_stride_bytes;
        bt.voff = bt.ii * 8u;
#pragma unroll
        for (int a = 0; a < B; ++a) bt.vm[a] = rel > uint32_t(a) ? bt.voff : OOB;
        const double2 qQ = *reinterpret_cast<const double2 *>(feat + 4 * bt.ii);
        bt.q = qQ.x;
        bt.Qd = qQ.y;
        bt.den = feat[4 * bt.ii + 2];
        return true;
    };
    struct CSet {
        double qc[B];  // Q[ii_a, ii_lane] within the block (0 for finished lanes)
        double qx[B];  // Q[ii_a(previous block), ii_lane]
    };
    // couplings of block `base` of batch `bt`; the block before it is block `prev_base` of batch `pb`
    auto fill = [&](CSet &S, const Batch &bt, int base, const Batch &pb, int prev_base, bool has_prev) {
#pragma unroll
        for (int a = 0; a < B; ++a) {
            const uint32_t roff = uint32_t(__builtin_amdgcn_readlane(int(bt.off), base + a));
            S.qc[a] = load_q(rsrc, bt.vm[a], roff);
        }
        if (has_prev) {
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const uint32_t roff = uint32_t(__builtin_amdgcn_readlane(int(pb.off), prev_base + a));
                S.qx[a] = load_q(rsrc, bt.voff, roff);
            }
        }
    };
    auto settle = [&](CSet &S) {
#pragma unroll
        for (int a = 0; a < B; ++a) {
            asm volatile("" : "+v"(S.qc[a]));
            asm volatile("" : "+v"(S.qx[a]));
        }
    };

    Batch cur, nxt;
    if (!load_batch(cur, 0) || !load_batch(nxt, 1)) return;
    int batch = 0;
    int n_iter = 0, f = 0;
    double wmax_v = 0.0, dmax_v = 0.0;
    double gap_out = tol_scaled + 1.0;
    double dp0[B], dp1[B];  // what the previous block published (DELTA: dp0 = differences; else dp0 = w_old, dp1 = w_new)
#pragma unroll
    for (int a = 0; a < B; ++a) dp0[a] = dp1[a] = 0.0;

    // the image after t_done blocks is complete: dual gap from it (same arithmetic as the one-wave form)
    auto epoch_end = [&](int t_done) -> bool {
        const double w_max = wave_max(wmax_v), d_w_max = wave_max(dmax_v);
        bool done = false;
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            if (!duo_wait(&ctl->seqB, t_done + 1, ctl, false)) return true;
            const double *im = L.img + (t_done & 1) * DuoLds<R>::IMG;
            double s_qw = 0, s_wh = 0, s_ww = 0, s_l1 = 0, m_xta = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int col = r * WAVE + lane;
                if (col < c) {
                    const double wv = w_lds[col], qv = feat[4 * col], hv = im[col];
                    const double xta = qv - hv - beta * wv;
                    s_qw += wv * qv;
                    s_wh += wv * hv;
                    s_ww += wv * wv;
                    s_l1 += fabs(wv);
                    m_xta = fmax(m_xta, fabs(xta));
                }
            }
            const double q_dot_w = wave_sum(s_qw), wh = wave_sum(s_wh), w_norm2 = wave_sum(s_ww),
                         l1 = wave_sum(s_l1), dual_norm = wave_max(m_xta);
            const double R_norm2 = y_norm2 + wh - 2.0 * q_dot_w;
            double const_, gap;
            if (dual_norm > alpha) {
                const_ = alpha / dual_norm;
                const double A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += alpha * l1 - const_ * y_norm2 + const_ * q_dot_w + 0.5 * beta * (1.0 + const_ * const_) * w_norm2;
            gap_out = gap;
            if (gap < tol_scaled) done = true;
        }
        ++n_iter;
        wmax_v = 0.0;
        dmax_v = 0.0;
        f = 0;
        return done || n_iter == max_iter;
    };

    unsigned long long waitA = 0, repairs = 0;
    // per-lane inputs of a block, fetched while the block before it is still running
    struct Pre {
        double Hn;   // image part of H[ii] (the previous block's 8 updates are added through qx)
        double wo;   // w[ii]
        int seq;     // seqB as seen just before Hn was read
    };
    auto prefetch = [&](Pre &pr, const Batch &nb, int t_next) {  // for block t_next: image t_next - 1
        pr.seq = duo_load(&ctl->seqB);
        pr.Hn = (L.img + ((t_next - 1) & 1) * DuoLds<R>::IMG)[nb.ii];
        pr.wo = w_lds[nb.ii];
    };

    // one block: lanes base..base+7 of `bt`; after step PF the inputs of the next block (lanes nbase.. of `nb`)
    // are requested into `pn`
    constexpr int PF = 5;
    auto compute = [&](const CSet &S, const Batch &bt, int base, int t, const Pre &pc, bool has_prev, const Batch &nb,
                       int nbase, Pre &pn) {
        double Hs_v = pc.Hn;
        if (has_prev) {
#pragma unroll
            for (int a = 0; a < B; ++a) {
                if (DELTA)
                    Hs_v = fma(dp0[a], S.qx[a], Hs_v);
                else
                    Hs_v = fma(dp1[a], S.qx[a], fma(-dp0[a], S.qx[a], Hs_v));
            }
        }
        double wo_v = pc.wo;
        const uint64_t blockmask = ((uint64_t(1) << B) - 1) << base;
        uint64_t wmask = blockmask;
        const bool has_dup = (bt.dupmask & blockmask) != 0;
        double wn_keep = 0.0, p0_v = 0.0, p1_v = 0.0;  // what this lane publishes for its own step
        if (!has_dup) {
            double wn_v = 0.0;
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                const double Hp = fma(-wo_v, bt.Qd, Hs_v);
                const double tmp = bt.q - Hp;
                const double thr = copysign(fmax(fabs(tmp) - alpha, 0.0), tmp);
                wn_v = RECIP ? thr * bt.den : thr / bt.den;
                if (DELTA) {
                    const double d_a = read_lane(wn_v - wo_v, la);
                    dp0[a] = d_a;
                    Hs_v = fma(d_a, S.qc[a], Hs_v);
                } else {
                    const double wo_a = read_lane(wo_v, la), wn_a = read_lane(wn_v, la);
                    dp0[a] = wo_a;
                    dp1[a] = wn_a;
                    Hs_v = fma(wn_a, S.qc[a], fma(-wo_a, S.qc[a], Hs_v));
                }
                if (a == PF) prefetch(pn, nb, t + 1);
            }
            wn_keep = wn_v;
            p0_v = DELTA ? wn_v - wo_v : wo_v;
            p1_v = wn_v;
        } else {  // a coordinate repeats inside the block: later visits must see the earlier result
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                const double Hp = fma(-wo_v, bt.Qd, Hs_v);
                const double tmp = bt.q - Hp;
                const double thr = copysign(fmax(fabs(tmp) - alpha, 0.0), tmp);
                const double wn_v = RECIP ? thr * bt.den : thr / bt.den;
                const bool mine = lane == la;
                wn_keep = mine ? wn_v : wn_keep;
                const double wo_a = read_lane(wo_v, la), wn_a = read_lane(wn_v, la);
                if (DELTA) {
                    const double dv = wn_v - wo_v;
                    const double d_a = read_lane(dv, la);
                    p0_v = mine ? dv : p0_v;
                    dp0[a] = d_a;
                    Hs_v = fma(d_a, S.qc[a], Hs_v);
                } else {
                    p0_v = mine ? wo_v : p0_v;
                    p1_v = mine ? wn_v : p1_v;
                    dp0[a] = wo_a;
                    dp1[a] = wn_a;
                    Hs_v = fma(wn_a, S.qc[a], fma(-wo_a, S.qc[a], Hs_v));
                }
                const uint32_t ii_a = uint32_t(__builtin_amdgcn_readlane(int(bt.ii), la));
                const bool later_same = bt.ii == ii_a && lane > la && lane < base + B;
                wo_v = later_same ? wn_a : wo_v;
                if (__ballot(later_same) != 0) wmask &= ~(uint64_t(1) << la);
                if (a == PF) prefetch(pn, nb, t + 1);
            }
        }
        if ((blockmask >> lane) & 1) {
            double *pb = L.pub + (t & 3) * 2 * B;
            if (DELTA) {
                pb[rel] = p0_v;
            } else {
                pb[2 * rel] = p0_v;
                pb[2 * rel + 1] = p1_v;
            }
            dmax_v = fmax(dmax_v, fabs(wn_keep - wo_v));
            wmax_v = fmax(wmax_v, fabs(wn_keep));
        }
        if ((wmask >> lane) & 1) w_lds[bt.ii] = wn_keep;
        duo_store(&ctl->seqA, t + 1);
        // rare repairs of the prefetch: the keeper had not published image t yet, or the next block revisits
        // a coordinate this block just changed
        if (pn.seq < t + 1) {
            const unsigned long long w0 = __builtin_readcyclecounter();
            duo_wait(&ctl->seqB, t + 1, ctl, false);
            pn.Hn = (L.img + (t & 1) * DuoLds<R>::IMG)[nb.ii];
            waitA += __builtin_readcyclecounter() - w0;
            ++repairs;
        }
        if ((nb.xdupmask >> nbase) & ((uint64_t(1) << B) - 1)) pn.wo = w_lds[nb.ii];
    };

    CSet SA, SB;
    Pre pa, pb2;
    fill(SA, cur, 0, cur, 0, false);
    duo_wait(&ctl->seqB, 1, ctl, false);
    pa.Hn = L.img[cur.ii];
    pa.wo = w_lds[cur.ii];
    pa.seq = 1;
    for (int t = 0;; t += 2) {  // blocks t (set A) and t+1 (set B); 8 blocks per batch
        const int g = t & 7;
        // ---- block t ----
        fill(SB, cur, (g + 1) * B, cur, g * B, true);
        compute(SA, cur, g * B, t, pa, t > 0, cur, (g + 1) * B, pb2);
        f += B;
        if (f == c && epoch_end(t + 1)) break;
        // ---- block t+1 ----
        const bool roll = g + 2 >= 8;
        if (!roll) {
            fill(SA, cur, (g + 2) * B, cur, (g + 1) * B, true);
            compute(SB, cur, (g + 1) * B, t + 1, pb2, true, cur, (g + 2) * B, pa);
        } else {
            fill(SA, nxt, 0, cur, (g + 1) * B, true);
            compute(SB, cur, (g + 1) * B, t + 1, pb2, true, nxt, 0, pa);
        }
        f += B;
        if (f == c && epoch_end(t + 2)) break;
        settle(SA);
        if (roll) {
            cur = nxt;
            ++batch;
            if (!load_batch(nxt, batch + 1)) break;
        }
    }
    duo_store(&ctl->stop, 1);
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = r * WAVE + lane;
        cnt += (col < c && w_lds[col] != 0.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
    if (lane == 0) {
        g_cd_debug[2] = waitA;
        g_cd_debug[3] = repairs;
        ctl->gap = gap_out;
        ctl->n_iter = duo_load(&ctl->err) ? -1 : n_iter;
        ctl->nnz = cnt;
    }
}

// ---- one-wave kernel + assist wave (c <= 256, c % 8 == 0) -----------------------------------------
// The second wave only runs the index stream: xorshift jump-ahead, rand_int, in-block duplicate scan,
// published per 64-value batch through the ring (three batches ahead at most, paced by seqA = batches taken).
template <int R>
__device__ __forceinline__ void cd_assist_wave(int ldq, int c, uint32_t seed, DuoLds<R> &L) {
    constexpr int B = 8;
    const int lane = threadIdx.x & 63;
    IdxStream rng;
    rng.init(seed, uint32_t(c), uint32_t(ldq) * 8u, lane);
    for (int k = 0;; ++k) {
        if (k >= 3 && !duo_wait(&L.ctl->seqA, k - 2, L.ctl, true)) return;  // slot k % 3 free once batch k-3 is taken
        if (k > 0) rng.next_batch();
        L.ii[(k % 3) * 64 + lane] = rng.idx;
        bool dup = false;
#pragma unroll
        for (int sft = 1; sft < B; ++sft) {
            const int other = __shfl(int(rng.idx), (lane & ~(B - 1)) | ((lane + sft) & (B - 1)), WAVE);
            dup |= (uint32_t(other) == rng.idx);
        }
        const uint64_t m = __ballot(dup);
        if (lane == 0) L.dup[k & 3] = m;
        duo_store(&L.ctl->batB, k + 1);
        if (duo_load(&L.ctl->stop)) return;
    }
}

template <int R, bool RECIP, bool DELTA>
__device__ __forceinline__ FitOut cd_fit_assist(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                                uint32_t seed, int max_iter, double tol_scaled, double d_w_tol,
                                                double y_norm2, double *w_lds, const double *feat, double *duo_base) {
    DuoLds<R> L;
    L.bind(duo_base);
    if (threadIdx.x == 0) {
        L.ctl->seqA = 0;
        L.ctl->seqB = 0;
        L.ctl->batB = 0;
        L.ctl->stop = 0;
        L.ctl->err = 0;
    }
    __syncthreads();
    if ((threadIdx.x >> 6) == 1) {
        cd_assist_wave<R>(ldq, c, seed, L);
    } else {
        const FitOut o = cd_fit_blocked<R, RECIP, DELTA, true>(Q, ldq, c, alpha, beta, seed, max_iter, tol_scaled, d_w_tol,
                                                               y_norm2, w_lds, feat, L.img /* H image */, &L);
        duo_store(&L.ctl->stop, 1);
        if ((threadIdx.x & 63) == 0) {
            L.ctl->gap = o.gap;
            L.ctl->n_iter = o.n_iter;
            L.ctl->nnz = o.nnz;
        }
    }
    __syncthreads();
    FitOut out;
    out.gap = L.ctl->gap;
    out.n_iter = L.ctl->n_iter;
    out.nnz = L.ctl->nnz;
    __syncthreads();
    return out;
}

// both waves of the workgroup call this; returns the same FitOut in all threads
template <int R, bool RECIP, bool DELTA>
__device__ __forceinline__ FitOut cd_fit_duo(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                             uint32_t seed, int max_iter, double tol_scaled, double d_w_tol,
                                             double y_norm2, double *w_lds, const double *feat, double *duo_base) {
    DuoLds<R> L;
    L.bind(duo_base);
    if (threadIdx.x == 0) {
        L.ctl->seqA = 0;
        L.ctl->seqB = 0;
        L.ctl->batB = 0;
        L.ctl->stop = 0;
        L.ctl->err = 0;
    }
    __syncthreads();
    if ((threadIdx.x >> 6) == 1)
        duo_keeper<R, RECIP, DELTA>(Q, ldq, c, seed, w_lds, L);
    else
        duo_chain<R, RECIP, DELTA>(Q, ldq, c, alpha, beta, max_iter, tol_scaled, d_w_tol, y_norm2, w_lds, feat, L);
    __syncthreads();
    FitOut out;
    out.gap = L.ctl->gap;
    out.n_iter = L.ctl->n_iter;
    out.nnz = L.ctl->nnz;
    __syncthreads();
    return out;
}

template <int R>
__device__ __forceinline__ FitOut cd_fit_any(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                             uint32_t seed, int max_iter, double tol_scaled, double d_w_tol,
                                             double y_norm2, int flags, double *w_lds, const double *feat,
                                             double *h_lds) {
    const bool recip = flags & CP_CD_RECIPROCAL, delta = flags & CP_CD_DELTA;
#define CP_FIT_ARGS Q, ldq, c, alpha, beta, seed, max_iter, tol_scaled, d_w_tol, y_norm2, w_lds, feat
    if constexpr (R <= 8) {
        if (c % Blk<R>::B == 0) {
            if (recip) {
                if (delta) return cd_fit_blocked<R, true, true>(CP_FIT_ARGS, h_lds);
                return cd_fit_blocked<R, true, false>(CP_FIT_ARGS, h_lds);
            }
            if (delta) return cd_fit_blocked<R, false, true>(CP_FIT_ARGS, h_lds);
            return cd_fit_blocked<R, false, false>(CP_FIT_ARGS, h_lds);
        }
    }
    const bool aligned = (c % (2 * Ring<R>::D)) == 0;
    if (recip) {
        if (delta) return aligned ? cd_fit<R, true, true, true>(CP_FIT_ARGS) : cd_fit<R, true, false, true>(CP_FIT_ARGS);
        return aligned ? cd_fit<R, true, true, false>(CP_FIT_ARGS) : cd_fit<R, true, false, false>(CP_FIT_ARGS);
    }
    if (delta) return aligned ? cd_fit<R, false, true, true>(CP_FIT_ARGS) : cd_fit<R, false, false, true>(CP_FIT_ARGS);
    return aligned ? cd_fit<R, false, true, false>(CP_FIT_ARGS) : cd_fit<R, false, false, false>(CP_FIT_ARGS);
#undef CP_FIT_ARGS
}

// LDS image shared by both kernels: w[c] | feat[4 c]
__device__ __forceinline__ void load_features(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                              const double *__restrict__ w_in, int c, double l2, int flags,
                                              double *w_lds, double *feat) {
    for (int j = threadIdx.x; j < c; j += WAVE) {
        const double dj = Q[size_t(j) * ldq + j];
        w_lds[j] = w_in ? w_in[j] : 0.0;
        feat[4 * j + 0] = q[j];
        feat[4 * j + 1] = dj;
        // zero-diagonal features are skipped by sklearn (_cd_fast.pyx:651); the blocked path makes their
        // update a no-op through the denominator instead (their row of Q, q and H entry are all zero)
        feat[4 * j + 2] = dj == 0.0 ? ((flags & CP_CD_RECIPROCAL) ? 0.0 : 1.0)
                                    : ((flags & CP_CD_RECIPROCAL) ? 1.0 / (dj + l2) : dj + l2);
        feat[4 * j + 3] = 0.0;
    }
    __syncthreads();
}


template <int R>
__global__ void __launch_bounds__(WAVE) k_cd_fit(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                                 const double *__restrict__ stats, int c, double l1, double l2,
                                                 uint32_t seed, int max_iter, double tol, int flags,
                                                 double *__restrict__ w, DevResult *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *feat = smem, *w_lds = smem + 4 * c, *h_lds = smem + 5 * c;
    load_features(Q, ldq, q, w, c, l2, flags, w_lds, feat);
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    FitOut o = cd_fit_any<R>(Q, ldq, c, l1, l2, seed, max_iter, tol_scaled, tol, y_norm2, flags, w_lds, feat, h_lds);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        g_cd_debug[0] = t1 - t0;
        g_cd_debug[1] = (unsigned long long)o.n_iter * (unsigned long long)c;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += WAVE) w[j] = w_lds[j];
    if (threadIdx.x == 0) {
        res->gap = o.gap;
        res->tol_scaled = tol_scaled;
        res->n_iter = o.n_iter;
        res->nnz = o.nnz;
        res->edge_margin = -1.0;
        res->gap_margin = -1.0;
    }
}

// Whole alpha search of lib/decompose.py:490-525 on the device.
template <int R>
__device__ __forceinline__ void
cd_search_body(const double *__restrict__ Q, int ldq, const double *__restrict__ q, const double *__restrict__ stats,
            int c, double M, double right0, double rank, double lbound, double rbound,
            const uint32_t *__restrict__ seeds, int max_fits, int max_iter, double tol, int flags,
            double *__restrict__ w, double *__restrict__ w_host, DevResult *__restrict__ log,
            double *__restrict__ log_alpha, int *__restrict__ fits_used, double *__restrict__ alpha_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *feat = smem, *w_lds = smem + 4 * c, *h_lds = smem + 5 * c;
    load_features(Q, ldq, q, nullptr, c, 0.0, flags, w_lds, feat);
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    int fit = 0;
    double left = 0.0, right = right0, alpha = right0;
    bool bracketing = true, ok = false;
    while (fit < max_fits) {
        alpha = bracketing ? right : (left + right) / 2;
        FitOut o = cd_fit_any<R>(Q, ldq, c, alpha * M, 0.0, seeds[fit], max_iter, tol_scaled, tol, y_norm2, flags,
                                 w_lds, feat, h_lds);
        if (threadIdx.x == 0) {
            log[fit].gap = o.gap;
            log[fit].tol_scaled = tol_scaled;
            log[fit].n_iter = o.n_iter;
            log[fit].nnz = o.nnz;
            log[fit].edge_margin = -1.0;
            log[fit].gap_margin = -1.0;
            log_alpha[fit] = alpha;
        }
        ++fit;
        const double tmp = double(o.nnz);
        if (bracketing) {  // decompose.py:502-515
            if (tmp < rank)
                bracketing = false;
            else
                right *= 2;
        } else {  // decompose.py:516-525
            if (tmp > rbound)
                left = alpha;
            else if (tmp < lbound)
                right = alpha;
            else {
                ok = true;
                break;
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += WAVE) {
        w[j] = w_lds[j];
        w_host[j] = w_lds[j];
    }
    if (threadIdx.x == 0) {
        *fits_used = ok ? fit : -fit;  // negative: ran out of pre-drawn seeds
        *alpha_out = alpha;
    }
}


// ---- two-wave kernels (128 threads): same contracts as k_cd_fit / k_cd_search ---------------------
template <int R, typename... A>
__device__ __forceinline__ FitOut cd_fit_duo_any(int flags, A... args) {
    const bool recip = flags & CP_CD_RECIPROCAL, delta = flags & CP_CD_DELTA;
    if constexpr (R <= 4) {  // c <= 256: one-wave kernel, the second wave only prepares the index batches
        if (recip) {
            if (delta) return cd_fit_assist<R, true, true>(args...);
            return cd_fit_assist<R, true, false>(args...);
        }
        if (delta) return cd_fit_assist<R, false, true>(args...);
        return cd_fit_assist<R, false, false>(args...);
    } else {         // 256 < c <= 512: chain wave + keeper wave
        if (recip) {
            if (delta) return cd_fit_duo<R, true, true>(args...);
            return cd_fit_duo<R, true, false>(args...);
        }
        if (delta) return cd_fit_duo<R, false, true>(args...);
        return cd_fit_duo<R, false, false>(args...);
    }
}

__device__ __forceinline__ void load_features_wg(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                                 const double *__restrict__ w_in, int c, double l2, int flags,
                                                 double *w_lds, double *feat) {
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        const double dj = Q[size_t(j) * ldq + j];
        w_lds[j] = w_in ? w_in[j] : 0.0;
        feat[4 * j + 0] = q[j];
        feat[4 * j + 1] = dj;
        feat[4 * j + 2] = dj == 0.0 ? ((flags & CP_CD_RECIPROCAL) ? 0.0 : 1.0)
                                    : ((flags & CP_CD_RECIPROCAL) ? 1.0 / (dj + l2) : dj + l2);
        feat[4 * j + 3] = 0.0;
    }
    __syncthreads();
}

template <int R>
__global__ void __launch_bounds__(2 * WAVE) k_cd_fit_duo(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                                         const double *__restrict__ stats, int c, double l1, double l2,
                                                         uint32_t seed, int max_iter, double tol, int flags,
                                                         double *__restrict__ w, DevResult *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *feat = smem, *w_lds = smem + 4 * c, *duo = smem + 5 * c;
    load_features_wg(Q, ldq, q, w, c, l2, flags, w_lds, feat);
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    FitOut o = cd_fit_duo_any<R>(flags, Q, ldq, c, l1, l2, seed, max_iter, tol_scaled, tol, y_norm2, w_lds, feat, duo);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        g_cd_debug[0] = t1 - t0;
        g_cd_debug[1] = (unsigned long long)o.n_iter * (unsigned long long)c;
    }
    for (int j = threadIdx.x; j < c; j += blockDim.x) w[j] = w_lds[j];
    if (threadIdx.x == 0) {
        res->gap = o.gap;
        res->tol_scaled = tol_scaled;
        res->n_iter = o.n_iter;
        res->nnz = o.nnz;
        res->edge_margin = -1.0;
        res->gap_margin = -1.0;
    }
}

template <int R>
__device__ __forceinline__ void
cd_search_duo_body(const double *__restrict__ Q, int ldq, const double *__restrict__ q, const double *__restrict__ stats,
                int c, double M, double right0, double rank, double lbound, double rbound,
                const uint32_t *__restrict__ seeds, int max_fits, int max_iter, double tol, int flags,
                double *__restrict__ w, double *__restrict__ w_host, DevResult *__restrict__ log,
                double *__restrict__ log_alpha, int *__restrict__ fits_used, double *__restrict__ alpha_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *feat = smem, *w_lds = smem + 4 * c, *duo = smem + 5 * c;
    load_features_wg(Q, ldq, q, nullptr, c, 0.0, flags, w_lds, feat);
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    int fit = 0;
    double left = 0.0, right = right0, alpha = right0;
    bool bracketing = true, ok = false;
    while (fit < max_fits) {
        alpha = bracketing ? right : (left + right) / 2;
        FitOut o = cd_fit_duo_any<R>(flags, Q, ldq, c, alpha * M, 0.0, seeds[fit], max_iter, tol_scaled, tol, y_norm2,
                                     w_lds, feat, duo);
        if (threadIdx.x == 0) {
            log[fit].gap = o.gap;
            log[fit].tol_scaled = tol_scaled;
            log[fit].n_iter = o.n_iter;
            log[fit].nnz = o.nnz;
            log[fit].edge_margin = -1.0;
            log[fit].gap_margin = -1.0;
            log_alpha[fit] = alpha;
        }
        ++fit;
        const double tmp = double(o.nnz);
        if (bracketing) {  // decompose.py:502-515
            if (tmp < rank)
                bracketing = false;
            else
                right *= 2;
        } else {  // decompose.py:516-525
            if (tmp > rbound)
                left = alpha;
            else if (tmp < lbound)
                right = alpha;
            else {
                ok = true;
                break;
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        w[j] = w_lds[j];
        w_host[j] = w_lds[j];
    }
    if (threadIdx.x == 0) {
        *fits_used = ok ? fit : -fit;  // negative: ran out of pre-drawn seeds
        *alpha_out = alpha;
    }
}

// One alpha search per workgroup.  k_cd_search / k_cd_search_duo: one layer per launch; the _batch forms take up
// to CP_CD_MAX_BATCH layers of the same width (blockIdx.x picks the argument block): independent layers that share a
// stream then run their searches side by side instead of one after the other.
template <int R>
__global__ void __launch_bounds__(WAVE)
k_cd_search(const double *__restrict__ Q, int ldq, const double *__restrict__ q, const double *__restrict__ stats,
            int c, double M, double right0, double rank, double lbound, double rbound,
            const uint32_t *__restrict__ seeds, int max_fits, int max_iter, double tol, int flags,
            double *__restrict__ w, double *__restrict__ w_host, DevResult *__restrict__ log,
            double *__restrict__ log_alpha, int *__restrict__ fits_used, double *__restrict__ alpha_out) {
    cd_search_body<R>(Q, ldq, q, stats, c, M, right0, rank, lbound, rbound, seeds, max_fits, max_iter, tol, flags, w, w_host, log, log_alpha, fits_used, alpha_out);
}
template <int R>
__global__ void __launch_bounds__(2 * WAVE)
k_cd_search_duo(const double *__restrict__ Q, int ldq, const double *__restrict__ q, const double *__restrict__ stats,
            int c, double M, double right0, double rank, double lbound, double rbound,
            const uint32_t *__restrict__ seeds, int max_fits, int max_iter, double tol, int flags,
            double *__restrict__ w, double *__restrict__ w_host, DevResult *__restrict__ log,
            double *__restrict__ log_alpha, int *__restrict__ fits_used, double *__restrict__ alpha_out) {
    cd_search_duo_body<R>(Q, ldq, q, stats, c, M, right0, rank, lbound, rbound, seeds, max_fits, max_iter, tol, flags, w, w_host, log, log_alpha, fits_used, alpha_out);
}
template <int R>
__global__ void __launch_bounds__(WAVE) k_cd_search_batch(CdSearchBatch b) {
    const CdSearchArgs &a = b.a[blockIdx.x];
    cd_search_body<R>(a.Q, a.ldq, a.q, a.stats, a.c, a.M, a.right0, a.rank, a.lbound, a.rbound, a.seeds, a.max_fits, a.max_iter, a.tol, a.flags, a.w, a.w_host, a.log, a.log_alpha, a.fits_used, a.alpha_out);
}
template <int R>
__global__ void __launch_bounds__(2 * WAVE) k_cd_search_duo_batch(CdSearchBatch b) {
    const CdSearchArgs &a = b.a[blockIdx.x];
    cd_search_duo_body<R>(a.Q, a.ldq, a.q, a.stats, a.c, a.M, a.right0, a.rank, a.lbound, a.rbound, a.seeds, a.max_fits, a.max_iter, a.tol, a.flags, a.w, a.w_host, a.log, a.log_alpha, a.fits_used, a.alpha_out);
}

}  // namespace

// More than 64 KB of dynamic LDS (c > 1228: w, the per-feature scalars and the H image are 5c + 64R doubles) needs an
// explicit opt-in per kernel: per device, idempotent, a few microseconds -- done on every such launch.  A workgroup may
// take the CU's whole 160 KB (MI355X_MICROARCH.md, LDS), which covers c = 2048 (96 KB): ResNet-50's shortcut blobs.
constexpr size_t CD_LDS_DEFAULT = 64 * 1024, CD_LDS_MAX = 160 * 1024;
template <typename K>
static hipError_t cd_lds_optin(K kernel, size_t lds) {
    if (lds <= CD_LDS_DEFAULT) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
}

// team form (cd_team.hip): chain wave + K keeper waves; c % 8 == 0, c <= 2048, flags 0 or 3
bool cp_cd_team_wanted(int c, int flags);
int cp_cd_team_fit_launch(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats, int c, double l1_reg,
                          double l2_reg, uint32_t seed, int max_iter, double tol, int flags, double *w, void *dres,
                          bool allow_multi);
int cp_cd_team_search_launch(cp_ctx *ctx, const void *batch, int n_jobs, int c, bool allow_multi);
bool cp_cd_multi_wanted(int c);
extern "C" int cp_debug_cd_team_cycles(cp_ctx *ctx, unsigned long long *out8);
// Which kernel family a context's last launch ran lives in the context (cp_ctx::last_cd_was_team): resident layer sets drive
// these entry points from several threads at once.  The cycle counters behind cp_debug_cd_cycles are device globals shared by
// all launches: they are only meaningful for single-stream use (tools/cd_bench.py).

// a fit of the log reported a hand-off time-out between the waves / workgroups of a team (n_iter = -1)
static int first_timed_out_fit(const DevResult *lg, int fits_used, int max_fits) {
    const int nfit = std::min(max_fits, fits_used < 0 ? -fits_used : fits_used);
    for (int f = 0; f < nfit; ++f)
        if (lg[f].n_iter < 0) return f;
    return -1;
}

#define CP_CD_DISPATCH(KERNEL, R_, ...)                                      \
    switch (R_) {                                                            \
        case 1: KERNEL<1><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;   \
        case 2: KERNEL<2><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;   \
        case 4: KERNEL<4><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;   \
        case 8: KERNEL<8><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;   \
        case 16: CP_HIP(ctx, cd_lds_optin(KERNEL<16>, lds)); KERNEL<16><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break; \
        default: CP_HIP(ctx, cd_lds_optin(KERNEL<32>, lds)); KERNEL<32><<<1, WAVE, lds, ctx->stream>>>(__VA_ARGS__); break; \
    }

static int pick_R(int c) {
    int R = 1;
    while (R * WAVE < c) R *= 2;
    return R;
}

// 128-thread kernels: c % 8 == 0 and c <= 512 (CP_CD_DUO=0 keeps the 64-thread kernels)
static bool use_duo(int c) {
    static const bool on = !(getenv("CP_CD_DUO") && atoi(getenv("CP_CD_DUO")) == 0);
    return on && c % 8 == 0 && c <= 8 * WAVE;  // c <= 256: one-wave kernel + assist wave; above: chain + keeper waves
}
static size_t duo_lds_bytes(int c) {
    const int R = pick_R(c);
    const int extra = R == 1 ? DuoLds<1>::doubles() : R == 2 ? DuoLds<2>::doubles() : R == 4 ? DuoLds<4>::doubles() : DuoLds<8>::doubles();
    return (size_t(5) * c + size_t(extra)) * sizeof(double);
}
#define CP_CD_DISPATCH_DUO(KERNEL, R_, ...)                                              \
    switch (R_) {                                                                        \
        case 1: KERNEL<1><<<1, 2 * WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;         \
        case 2: KERNEL<2><<<1, 2 * WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;         \
        case 4: KERNEL<4><<<1, 2 * WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;         \
        default: KERNEL<8><<<1, 2 * WAVE, lds, ctx->stream>>>(__VA_ARGS__); break;        \
    }

extern "C" int cp_enet_cd_gram(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats, int c,
                               double l1_reg, double l2_reg, uint32_t seed, int max_iter, double tol, int flags,
                               double *w, cp_cd_result *result) {
    if (!ctx || !Q || !q || !stats || !w || !result || c <= 0 || ldq < c || max_iter <= 0) return CP_ERR_ARG;
    if (c > 32 * WAVE) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d > %d channels", c, 32 * WAVE);
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_TRY(cp_arena_reserve(ctx, 4096));
    DevResult *dres = reinterpret_cast<DevResult *>(cp_arena_take(ctx, sizeof(DevResult)));
    const bool duo = use_duo(c);
    const size_t lds = duo ? duo_lds_bytes(c) : (size_t(5) * c + size_t(WAVE) * pick_R(c)) * sizeof(double);
    if (lds > CD_LDS_MAX) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d needs %zu B of LDS (limit %zu)", c, lds, CD_LDS_MAX);
    const int R = pick_R(c);
    cp_stage_begin(ctx);
    ctx->last_cd_was_team = cp_cd_team_wanted(c, flags);
    static_assert(sizeof(DevResult) == sizeof(cp_cd_result), "layout");
    CP_TRY(cp_pinned_reserve(ctx, 4096));
    // The multi-CU team (c > 512) depends on its 1 + G workgroups being resident together and talks through global memory
    // with bounded waits.  Should a hand-off time out (n_iter = -1; the kernel then leaves w untouched), the fit is re-run
    // by the one-workgroup team: the same arithmetic in the same order (bit-identical w), only slower.
    const bool may_fall_back = ctx->last_cd_was_team && cp_cd_multi_wanted(c);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool multi = attempt == 0;
        if (ctx->last_cd_was_team) {
            CP_TRY(cp_cd_team_fit_launch(ctx, Q, ldq, q, stats, c, l1_reg, l2_reg, seed, max_iter, tol, flags, w, dres, multi));
        } else if (duo) {
            CP_CD_DISPATCH_DUO(k_cd_fit_duo, R, Q, ldq, q, stats, c, l1_reg, l2_reg, seed, max_iter, tol, flags, w, dres);
        } else {
            CP_CD_DISPATCH(k_cd_fit, R, Q, ldq, q, stats, c, l1_reg, l2_reg, seed, max_iter, tol, flags, w, dres);
        }
        CP_LAUNCH_CHECK(ctx);
        cp_stage_mark(ctx, "cd_fit");
        CP_HIP(ctx, hipMemcpyAsync(ctx->pinned, dres, sizeof(DevResult), hipMemcpyDeviceToHost, ctx->stream));
        CP_HIP(ctx, cp_stream_wait(ctx));
        memcpy(result, ctx->pinned, sizeof(cp_cd_result));
        if (result->n_iter >= 0) return CP_OK;
        if (!(may_fall_back && multi)) break;
        ++ctx->cd_fallbacks;
    }
    return cp_set_error(ctx, CP_ERR_NUMERIC, "cd: a hand-off between the waves of the coordinate-descent team timed out "
                        "(CP_CD_TEAM=0 runs the one- / two-wave kernels)");
}

bool cp_cd_multi_wanted(int c);
extern "C" int cp_cd_kernel_form(int c, int flags) {
    if (c <= 0 || c > 32 * WAVE) return -1;
    if (cp_cd_team_wanted(c, flags)) return cp_cd_multi_wanted(c) ? CP_CD_FORM_MULTI : CP_CD_FORM_TEAM;
    return use_duo(c) ? CP_CD_FORM_DUO : CP_CD_FORM_WAVE;
}

extern "C" int cp_debug_cd_cycles(cp_ctx *ctx, unsigned long long *out2) {
    if (!ctx || !out2) return CP_ERR_ARG;
    if (ctx->last_cd_was_team) return cp_debug_cd_team_cycles(ctx, out2);
    CP_HIP(ctx, cp_stream_wait(ctx));
    CP_HIP(ctx, hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_cd_debug), 8 * sizeof(unsigned long long)));
    return CP_OK;
}

extern "C" int cp_lasso_alpha_search(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats,
                                     int c, double M, double alpha_right0, double rank, double lbound, double rbound,
                                     const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                                     double *w, int *fits_used, double *alpha_out, cp_cd_result *fit_log,
                                     double *fit_alpha) {
    if (!ctx || !Q || !q || !stats || !w || !seeds || !fits_used || !alpha_out || c <= 0 || ldq < c ||
        max_fits <= 0 || max_fits > 4096 || max_iter <= 0)
        return CP_ERR_ARG;
    if (c > 32 * WAVE) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d > %d channels", c, 32 * WAVE);
    CP_HIP(ctx, hipSetDevice(ctx->device));
    // Everything small the kernel reads or reports lives in pinned host memory that the device
    // addresses directly: no copy packets before or after the launch -- with many layers in
    // flight every extra packet in the stream costs tens of microseconds of dispatch latency.
    const size_t log_bytes = size_t(max_fits) * sizeof(DevResult), al_bytes = size_t(max_fits) * sizeof(double),
                 seed_bytes = cp_align_up(size_t(max_fits) * sizeof(uint32_t), 64), w_bytes = size_t(c) * sizeof(double);
    const size_t off_log = 128, off_al = off_log + log_bytes, off_seed = off_al + al_bytes, off_w = off_seed + seed_bytes;
    CP_TRY(cp_pinned_reserve(ctx, off_w + w_bytes));
    char *h = ctx->pinned;
    int *hfits = reinterpret_cast<int *>(h);
    double *halpha = reinterpret_cast<double *>(h + 64);
    *hfits = 0;
    memcpy(h + off_seed, seeds, size_t(max_fits) * sizeof(uint32_t));
    const bool duo = use_duo(c);
    const size_t lds = duo ? duo_lds_bytes(c) : (size_t(5) * c + size_t(WAVE) * pick_R(c)) * sizeof(double);
    if (lds > CD_LDS_MAX) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d needs %zu B of LDS (limit %zu)", c, lds, CD_LDS_MAX);
    const int R = pick_R(c);
    cp_stage_begin(ctx);
    cp_stage_mark(ctx, "cd_search_begin");   // opens the bracket of the search (timing mode 2)
    ctx->last_cd_was_team = cp_cd_team_wanted(c, flags);
    // A search always starts from w = 0 inside the kernel, so a search whose multi-CU team reported a hand-off time-out
    // (n_iter = -1 in its log) is simply run again by the one-workgroup team: bit-identical fits, only slower.
    const bool may_fall_back = ctx->last_cd_was_team && cp_cd_multi_wanted(c);
    const DevResult *lg = reinterpret_cast<const DevResult *>(h + off_log);
    int timed_out_fit = -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        *hfits = 0;
        if (ctx->last_cd_was_team) {
            CdSearchBatch batch;
            memset(&batch, 0, sizeof(batch));
            CdSearchArgs &a = batch.a[0];
            a.Q = Q; a.ldq = ldq; a.q = q; a.stats = stats; a.c = c; a.M = M; a.right0 = alpha_right0; a.rank = rank;
            a.lbound = lbound; a.rbound = rbound; a.seeds = reinterpret_cast<const uint32_t *>(h + off_seed);
            a.max_fits = max_fits; a.max_iter = max_iter; a.tol = tol; a.flags = flags; a.w = w;
            a.w_host = reinterpret_cast<double *>(h + off_w); a.log = reinterpret_cast<DevResult *>(h + off_log);
            a.log_alpha = reinterpret_cast<double *>(h + off_al); a.fits_used = hfits; a.alpha_out = halpha;
            CP_TRY(cp_cd_team_search_launch(ctx, &batch, 1, c, attempt == 0));
        } else if (duo) {
            CP_CD_DISPATCH_DUO(k_cd_search_duo, R, Q, ldq, q, stats, c, M, alpha_right0, rank, lbound, rbound,
                               reinterpret_cast<const uint32_t *>(h + off_seed), max_fits, max_iter, tol, flags, w,
                               reinterpret_cast<double *>(h + off_w), reinterpret_cast<DevResult *>(h + off_log),
                               reinterpret_cast<double *>(h + off_al), hfits, halpha);
        } else {
            CP_CD_DISPATCH(k_cd_search, R, Q, ldq, q, stats, c, M, alpha_right0, rank, lbound, rbound,
                           reinterpret_cast<const uint32_t *>(h + off_seed), max_fits, max_iter, tol, flags, w,
                           reinterpret_cast<double *>(h + off_w), reinterpret_cast<DevResult *>(h + off_log),
                           reinterpret_cast<double *>(h + off_al), hfits, halpha);
        }
        CP_LAUNCH_CHECK(ctx);
        cp_stage_mark(ctx, "cd_alpha_search");
        CP_HIP(ctx, cp_stream_wait(ctx));
        timed_out_fit = first_timed_out_fit(lg, *hfits, max_fits);
        if (timed_out_fit < 0 || !(may_fall_back && attempt == 0)) break;
        ++ctx->cd_fallbacks;
    }
    ctx->pinned_w = reinterpret_cast<const double *>(h + off_w);
    memcpy(fits_used, hfits, sizeof(int));
    memcpy(alpha_out, halpha, sizeof(double));
    if (fit_log) memcpy(fit_log, h + off_log, log_bytes);
    if (fit_alpha) memcpy(fit_alpha, h + off_al, al_bytes);
    if (timed_out_fit >= 0) {   // whatever the sign of fits_used: a fit that ran on stale data voids the search
        if (*fits_used > 0) *fits_used = -*fits_used;
        return cp_set_error(ctx, CP_ERR_NUMERIC, "alpha search: a hand-off between the waves of the coordinate-descent team timed "
                            "out in fit %d (CP_CD_TEAM=0 runs the one- / two-wave kernels)", timed_out_fit);
    }
    if (*fits_used < 0) return cp_set_error(ctx, CP_ERR_NUMERIC, "alpha search did not terminate within %d fits", max_fits);
    return CP_OK;
}

// ---- batched alpha search: several layers of the same width, one launch, one workgroup each ---------------------
namespace {
struct SearchPinned {  // lay-out of the pinned block the search kernel reads its seeds from and reports into
    size_t off_log, off_al, off_seed, off_w, total;
    SearchPinned(int max_fits, int c) {
        const size_t log_bytes = size_t(max_fits) * sizeof(DevResult), al_bytes = size_t(max_fits) * sizeof(double),
                     seed_bytes = cp_align_up(size_t(max_fits) * sizeof(uint32_t), 64);
        off_log = 128;
        off_al = off_log + log_bytes;
        off_seed = off_al + al_bytes;
        off_w = off_seed + seed_bytes;
        total = off_w + size_t(c) * sizeof(double);
    }
};
}  // namespace

int cp_alpha_search_enqueue_batch(cp_ctx *const *ctxs, int n_jobs, const cp_search_job *jobs, bool allow_multi) {
    if (!ctxs || !jobs || n_jobs <= 0 || n_jobs > CP_CD_MAX_BATCH) return CP_ERR_ARG;
    cp_ctx *ctx = ctxs[0];
    const int c = jobs[0].c;
    if (c <= 0 || c > 32 * WAVE) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d", c);
    CdSearchBatch batch;
    memset(&batch, 0, sizeof(batch));
    for (int l = 0; l < n_jobs; ++l) {
        const cp_search_job &j = jobs[l];
        if (j.c != c || j.ldq < c || j.max_fits <= 0 || j.max_fits > 4096 || j.max_iter <= 0 || !j.Q || !j.q || !j.stats ||
            !j.w || !j.seeds)
            return cp_set_error(ctx, CP_ERR_ARG, "alpha search batch: job %d does not match the batch (c=%d vs %d)", l, j.c, c);
        cp_ctx *cl = ctxs[l];
        const SearchPinned lay(j.max_fits, c);
        CP_TRY(cp_pinned_reserve(cl, lay.total));
        char *h = cl->pinned;
        *reinterpret_cast<int *>(h) = 0;
        memcpy(h + lay.off_seed, j.seeds, size_t(j.max_fits) * sizeof(uint32_t));
        cl->pinned_w = reinterpret_cast<const double *>(h + lay.off_w);
        CdSearchArgs &a = batch.a[l];
        a.Q = j.Q; a.ldq = j.ldq; a.q = j.q; a.stats = j.stats; a.c = c; a.M = j.M; a.right0 = j.alpha_right0;
        a.rank = j.rank; a.lbound = j.lbound; a.rbound = j.rbound;
        a.seeds = reinterpret_cast<const uint32_t *>(h + lay.off_seed);
        a.max_fits = j.max_fits; a.max_iter = j.max_iter; a.tol = j.tol; a.flags = j.flags; a.w = j.w;
        a.w_host = reinterpret_cast<double *>(h + lay.off_w);
        a.log = reinterpret_cast<DevResult *>(h + lay.off_log);
        a.log_alpha = reinterpret_cast<double *>(h + lay.off_al);
        a.fits_used = reinterpret_cast<int *>(h);
        a.alpha_out = reinterpret_cast<double *>(h + 64);
    }
    const bool duo = use_duo(c);
    const size_t lds = duo ? duo_lds_bytes(c) : (size_t(5) * c + size_t(WAVE) * pick_R(c)) * sizeof(double);
    if (lds > CD_LDS_MAX) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "cd: c=%d needs %zu B of LDS (limit %zu)", c, lds, CD_LDS_MAX);
    const int R = pick_R(c);
    cp_stage_begin(ctx);
    cp_stage_mark(ctx, "cd_search_begin");   // opens the bracket of the search (timing mode 2)
    bool team = cp_cd_team_wanted(c, jobs[0].flags);
    for (int l = 1; l < n_jobs; ++l) team = team && jobs[l].flags == jobs[0].flags;
    for (int l = 0; l < n_jobs; ++l) ctxs[l]->last_cd_was_team = team;
    if (team) {
        CP_TRY(cp_cd_team_search_launch(ctx, &batch, n_jobs, c, allow_multi));
    } else if (duo) {
        switch (R) {
            case 1: k_cd_search_duo_batch<1><<<n_jobs, 2 * WAVE, lds, ctx->stream>>>(batch); break;
            case 2: k_cd_search_duo_batch<2><<<n_jobs, 2 * WAVE, lds, ctx->stream>>>(batch); break;
            case 4: k_cd_search_duo_batch<4><<<n_jobs, 2 * WAVE, lds, ctx->stream>>>(batch); break;
            default: k_cd_search_duo_batch<8><<<n_jobs, 2 * WAVE, lds, ctx->stream>>>(batch); break;
        }
    } else {
        switch (R) {
            case 1: k_cd_search_batch<1><<<n_jobs, WAVE, lds, ctx->stream>>>(batch); break;
            case 2: k_cd_search_batch<2><<<n_jobs, WAVE, lds, ctx->stream>>>(batch); break;
            case 4: k_cd_search_batch<4><<<n_jobs, WAVE, lds, ctx->stream>>>(batch); break;
            case 8: k_cd_search_batch<8><<<n_jobs, WAVE, lds, ctx->stream>>>(batch); break;
            case 16:
                CP_HIP(ctx, cd_lds_optin(k_cd_search_batch<16>, lds));
                k_cd_search_batch<16><<<n_jobs, WAVE, lds, ctx->stream>>>(batch);
                break;
            default:
                CP_HIP(ctx, cd_lds_optin(k_cd_search_batch<32>, lds));
                k_cd_search_batch<32><<<n_jobs, WAVE, lds, ctx->stream>>>(batch);
                break;
        }
    }
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "cd_alpha_search");
    return CP_OK;
}

int cp_alpha_search_collect(cp_ctx *ctx, int c, int max_fits, int *fits_used, double *alpha_out, cp_cd_result *fit_log,
                            double *fit_alpha, bool *timed_out) {
    const SearchPinned lay(max_fits, c);
    const char *h = ctx->pinned;
    memcpy(fits_used, h, sizeof(int));
    memcpy(alpha_out, h + 64, sizeof(double));
    if (fit_log) memcpy(fit_log, h + lay.off_log, size_t(max_fits) * sizeof(DevResult));
    if (fit_alpha) memcpy(fit_alpha, h + lay.off_al, size_t(max_fits) * sizeof(double));
    const int bad = first_timed_out_fit(reinterpret_cast<const DevResult *>(h + lay.off_log), *fits_used, max_fits);
    if (timed_out) *timed_out = bad >= 0;
    if (bad >= 0) {   // whatever the sign of fits_used: a fit that ran on stale data voids the search
        if (*fits_used > 0) *fits_used = -*fits_used;
        return cp_set_error(ctx, CP_ERR_NUMERIC, "alpha search: a hand-off between the waves / workgroups of the coordinate-descent "
                            "team timed out in fit %d", bad);
    }
    if (*fits_used < 0) return CP_ERR_NUMERIC;
    return CP_OK;
}

extern "C" int cp_debug_cd_fail_multi(cp_ctx *ctx, int on) {
    if (!ctx) return CP_ERR_ARG;
    ctx->cd_test_fail_multi = on != 0;
    return CP_OK;
}

extern "C" int cp_debug_cd_fallbacks(cp_ctx *ctx) { return ctx ? ctx->cd_fallbacks : -1; }
