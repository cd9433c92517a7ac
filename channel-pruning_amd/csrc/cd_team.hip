// Coordinate-descent LASSO on the Gram matrix, team form: ONE chain wave, K keeper waves and a stager wave.
//
// Same recurrence, same visit order, same fma sequence per H entry as cd_gram.hip (sklearn's
// enet_coordinate_descent_gram, _cd_fast.pyx:564-737; coordinate stream our_rand_r, _random.pxd:20-35), so w, n_iter and
// the zero pattern are bit-identical to oracle/cd_oracle.c::cpo_enet_cd_gram in sklearn's own operation order (flags 0).
//
// Two forms live here: the ONE-WORKGROUP team (c <= 512 by default: this header and the first half of the file) and the
// MULTI-CU team (512 < c <= 2048: its keepers in 2-4 further workgroups, one CU each, talking through global memory; see
// the section "multi-CU team" further down).  Both run the same chain wave (team_chain), the same keeper (team_keeper) and
// the same stager (team_stager).
//
// Why a team.  A single wavefront issues one instruction every ~5-6.5 cycles whatever the dependencies, so a coordinate
// step costs what its instruction count costs, and the step is a serial chain.  cd_gram.hip splits the count over two
// waves (chain + keeper) for 256 < c <= 512 and runs everything in one wave otherwise; its keeper applies 2 c / 64 fma per
// lane and step and is the bottleneck at c = 512 (331 cycles per step measured), and the one-wave forms for c > 512 are
// bound by the row traffic one wave can keep in flight (0.41 us per step at c = 1024, 0.69 at c = 2048).  Here
//   chain wave  (wave 0)      the scalar recurrence only: per step the soft-threshold chain, one readlane pair per published
//                              value and the fma pair on the lanes' private H[ii]; never touches the full H;
//   keeper k    (wave 1 + k)   owns H[:, 64 R k .. 64 R (k + 1)) in registers (R <= 4 doubles per lane), fetches its slice of
//                              the rows Q[ii, :] sixteen steps ahead, applies a block's 8 updates with what the chain wave
//                              published and exposes its slice of H as an LDS image after every block;
//   stager      (wave K + 1)   publishes the index stream's batches and stages the couplings Q[ii_a, ii_j] of every block
//                              in LDS, three blocks ahead of the chain wave.
// K = ceil(c / 256) keepers (R <= 4: 2 R <= 8 fma per wave and step whatever c is): four waves, one per SIMD, up to
// c = 512; with CP_CD_MULTI=0 also four keepers up to c = 1024 and six keepers of 384 columns up to 2048 (eight waves: two
// per SIMD, so every wave keeps 256 registers).  Hand-offs are sequence counters in LDS, one writer each (the LDS executes
// a wave's instructions in program order, so a counter written after its payload lands after it); no barrier inside a fit.
//
// The division of the soft-threshold step.  sklearn divides by Q_ii (+ beta); an IEEE f64 division is ~11 dependent
// instructions on the chain.  With r = RN(1 / d) computed once per feature by a true division (load time, off the chain):
//     q0 = RN(m r);  rem = RN(m - d q0) (exact: fma);  q1 = RN(q0 + rem r)          (Markstein 1990)
// q1 is the correctly rounded quotient m / d for every finite m >= 0, d > 0 away from over / underflow (three operations;
// checked against the hardware division on 8e8 random and adversarial operand pairs by tests/host/test_markstein.c, and
// by every bit-exactness test of the CD kernels).  Features or l1 weights outside [2^-400, 2^400] take the IEEE division.
#include "cp_common.h"
#include "xorshift_jump.h"
#include "cd_shared.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace {
using namespace cdk;

constexpr int B = 8;        // coordinate steps per block (= per hand-off)
constexpr int KMAX = 8;     // keeper waves at most

struct TeamCtl {
    int seqA;          // blocks published by the chain wave
    int batB;          // index batches published by keeper 0
    int stop;          // chain -> keepers: the fit is over
    int err;           // a bounded wait ran out (never in a correct run)
    int n_iter, nnz;
    int mk_ok;         // every feature's denominator is inside the range the three-operation division is exact on
    int cplSeq;        // blocks whose couplings keeper 0 (multi-CU team: the stager wave) has staged in LDS
    int hnSlot[4];     // multi-CU team: hnSlot[v & 3] = v + 1 once a gatherer wave has published block v's starting H values
    int seqB[KMAX];    // images published by keeper k (image t = H after the first t blocks; count = t + 1)
    double gap;
    double edge_margin, gap_margin;   // tie sentinels of the fit (cp_cd_result)
};

template <int R, int K, int NI = 2>
struct TeamLds {
    static constexpr int IMG = 64 * R * K;
    static constexpr int PUB_RING = NI > 2 ? 16 : 4, II_RING = 3, CPL_REC = 2 * B;
    double *img;       // [NI][IMG]
    double *edge;      // [IMG] | |tmp| - alpha | of every coordinate's LAST update in the fit (tie sentinel)
    double *pub;       // [4][2 * B]
    double *cpl;       // [4][B][2 * B] couplings of a block, staged by keeper 0: [j][a] = Q[ii_a, ii_j] (0 for j <= a),
                       //               [j][B + a] = Q[ii_a(previous block), ii_j]
    uint32_t *ii;      // [3][64]
    uint64_t *dup;     // [4] lanes whose coordinate repeats inside their block
    uint64_t *xdup;    // [4] lanes whose coordinate also occurs in the block before theirs
    TeamCtl *ctl;
    static __host__ __device__ constexpr int doubles() {
        return (NI + 1) * IMG + PUB_RING * 2 * B + 4 * B * 2 * B + 3 * 32 + 8 + int(sizeof(TeamCtl) / 8) + 2;
    }
    __device__ void bind(double *base) {
        img = base;
        edge = img + NI * IMG;
        pub = edge + IMG;
        cpl = pub + PUB_RING * 2 * B;
        ii = reinterpret_cast<uint32_t *>(cpl + 4 * B * 2 * B);
        dup = reinterpret_cast<uint64_t *>(ii + 3 * 64);
        xdup = dup + 4;
        ctl = reinterpret_cast<TeamCtl *>(xdup + 4);
    }
};

// No-op updates (measured, OFF by default).  sklearn skips the axpy of a coefficient that is zero (`if w_ii != 0`,
// _cd_fast.pyx:656 / 675), so a coordinate that is zero before AND after its step leaves H untouched -- here its fma pair adds
// (+-0) * Q: the same bits.  Skipping the pairs of such steps in every wave that applies published steps (CP_CD_SKIP_APPLY:
// keepers, gatherers, the chain wave's catch-up) and not re-evaluating the block's remaining lanes after one
// (CP_CD_SKIP_CHAIN) is exact and was expected to pay on sparse searches (35-40 % of the steps of the ResNet-50 searches move
// nothing, 70 % in tools/cd_bench.py's problem).  It does not: the compiler if-converts the pairs anyway and the branches
// around the re-evaluation cost more than the dependent chain they skip -- c = 512 226 -> 254 cycles per step with
// CP_CD_SKIP_CHAIN, 295 with CP_CD_SKIP_APPLY, c = 2048 253 -> 270 / 300; ResNet-50 job 35.3 -> 38.6 ms with both.
// The step is bound by its instruction count and the readlane round trips, not by the f64 chain alone.
#ifndef CP_CD_SKIP_CHAIN   // the chain wave's own steps (and their re-evaluation)
#define CP_CD_SKIP_CHAIN 0
#endif
#ifndef CP_CD_SKIP_APPLY   // keepers, gatherers and the chain wave's catch-up of the block before
#define CP_CD_SKIP_APPLY 0
#endif
template <bool ON>
__device__ __forceinline__ bool moves_t(double a) { return !ON || (uint64_t(__double_as_longlong(a)) << 1) != 0; }
template <bool ON>
__device__ __forceinline__ bool moves_t(double w_old, double w_new) {
    return !ON || ((uint64_t(__double_as_longlong(w_old)) | uint64_t(__double_as_longlong(w_new))) << 1) != 0;
}
__device__ __forceinline__ bool moves_h(double a) { return moves_t<CP_CD_SKIP_APPLY != 0>(a); }
__device__ __forceinline__ bool moves_h(double w_old, double w_new) { return moves_t<CP_CD_SKIP_APPLY != 0>(w_old, w_new); }
__device__ __forceinline__ bool moves_c(double a) { return moves_t<CP_CD_SKIP_CHAIN != 0>(a); }
__device__ __forceinline__ bool moves_c(double w_old, double w_new) { return moves_t<CP_CD_SKIP_CHAIN != 0>(w_old, w_new); }

// wait until *p >= need (false if `stop` was raised or the bound ran out)
__device__ __forceinline__ bool team_wait(int *p, int need, TeamCtl *ctl, bool watch_stop) {
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
// every keeper has published image `need - 1`
template <int K>
__device__ __forceinline__ bool images_ready(TeamCtl *ctl, int need, int lane) {
    return __ballot(duo_load(&ctl->seqB[lane % K]) >= need) == ~uint64_t(0);
}
template <int K>
__device__ __forceinline__ bool team_wait_images(TeamCtl *ctl, int need, int lane) {
    for (int spin = 0;; ++spin) {
        if (images_ready<K>(ctl, need, lane)) return true;
        if (spin > (1 << 22)) {
            duo_store(&ctl->err, 1);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ unsigned long long g_team_debug[8];
#ifdef CP_CD_MULTI_TRACE   // timing experiment: wall-clock stamps (10 ns units) of one block's trip through the multi-CU team
__device__ unsigned long long g_multi_trace[16];
#define CP_TRACE(slot, cond)                                                             \
    do {                                                                                 \
        if ((cond) && (threadIdx.x & 63) == 0) g_multi_trace[slot] = wall_clock64();     \
    } while (0)
constexpr int TRACE_BLOCK = 2000;
#else
#define CP_TRACE(slot, cond) do { } while (0)
#endif
#ifdef CP_CD_TEAM_TRACE   // timing experiment: shader-clock stamps of one block's trip chain wave -> keeper 0 -> chain wave
__device__ unsigned long long g_team_trace[16];
#define CP_TTRACE(slot, cond)                                                                           \
    do {                                                                                                \
        if ((cond) && (threadIdx.x & 63) == 0) g_team_trace[slot] = __builtin_readcyclecounter();       \
    } while (0)
constexpr int TTRACE_BLOCK = 1000;
#else
#define CP_TTRACE(slot, cond) do { } while (0)
#endif

// ---- cross-workgroup mailbox of the multi-CU team (see the section "multi-CU team" below) --------------------------------
// Every word is read and written with relaxed agent-scope atomics only (sc1 loads / stores: coherent across the XCDs' L2s)
// and carries its own validity: a slot holds SENT until its one writer stores the value, and its one reader puts SENT back.
#ifndef CP_CD_MULTI_LAG
#define CP_CD_MULTI_LAG 5
#endif
constexpr int XLAG = CP_CD_MULTI_LAG;      // blocks the remotes run behind the chain wave (2 <= XLAG <= 5)
constexpr unsigned long long SENT = 0x7FF8C0DEC0DEC0DEull;   // a quiet NaN no arithmetic produces
constexpr int RING = XLAG > 5 ? 16 : 8;                      // blocks of slots per ring (>= XLAG + 3)
struct MultiBox {
    unsigned long long *dl;     // [G][RING][2 B]  home -> remote g: what the chain wave published for a block
    unsigned long long *hv;     // [G][RING][B]    remote g -> home: H[ii_j] of a block's coordinates, LAG blocks stale
    unsigned long long *fitw;   // [c]             home -> remotes: w at the start of a fit
    unsigned long long *himg;   // [G * slice]     remotes -> home: H at the end of an epoch
    unsigned long long *ctl;    // [0,G) fit posted | [G,2G) fit stopped | [2G,3G) stop acknowledged | [3G,4G) epoch image | [4G] abort
                                // | (4G, 5G] remote workgroup g is resident
    int G, slice;
};
__device__ __forceinline__ unsigned long long xld(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xst(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double xld_d(const unsigned long long *p) { return __longlong_as_double((long long)xld(p)); }
__device__ __forceinline__ void xst_d(unsigned long long *p, double v) { xst(p, (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int XSPIN = 1 << 21;   // bound of a wait on another workgroup (~1 s; never reached in a correct run)
__device__ __forceinline__ bool multi_aborted(const MultiBox *box) { return xld(box->ctl + 4 * box->G) != 0; }
__device__ __forceinline__ void multi_abort(const MultiBox *box, TeamCtl *ctl) {
    xst(box->ctl + 4 * box->G, 1);
    duo_store(&ctl->err, 1);
    duo_store(&ctl->stop, 1);
}
// Poll one mailbox word per lane until no lane of `want` reads SENT any more (returns 1, the words in `out`), `over(word)`
// holds in some lane (returns 0) or the bound runs out (returns -1).  An sc1 load takes ~0.4 us here, so LOOKS of them are
// kept in flight, issued from inline assembly and collected with an explicit vmcnt: a new look every ~0.1 us instead of one
// per round trip.  The looks still in flight when the call returns keep writing into lk.r: the caller owns those registers
// and hands them back with poll_settle() once its urgent work is done (a destination the compiler had re-used for something
// else in the meantime would be overwritten under its feet).
template <int LOOKS>
struct Looks {
    unsigned long long r[LOOKS];
};
template <int LOOKS>
__device__ __forceinline__ void poll_settle(Looks<LOOKS> &lk) {
    vm_drain();
#pragma unroll
    for (int i = 0; i < LOOKS; ++i) asm volatile("" : "+v"(lk.r[i]));
}
template <int LOOKS, class Over>
__device__ __forceinline__ int poll_words(Looks<LOOKS> &lk, const unsigned long long *addr, bool want, unsigned long long &out,
                                          Over over) {
    unsigned long long(&r)[LOOKS] = lk.r;
#pragma unroll
    for (int i = 0; i < LOOKS; ++i) {
        r[i] = SENT;   // whatever the register allocator does with a look's destination before it lands, it reads "not yet"
        asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(r[i]) : "v"(addr) : "memory");
    }
    for (int spin = 0; spin < XSPIN; ++spin) {
#pragma unroll
        for (int i = 0; i < LOOKS; ++i) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[i]) : "n"(LOOKS - 1) : "memory");   // the oldest look has landed
            const unsigned long long v = r[i];
            if (__ballot(want && v == SENT) == 0) {
                out = v;
                return 1;
            }
            if (over(v)) return 0;
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(r[i]) : "v"(addr) : "memory");
        }
    }
    return -1;
}

// chain wave: every remote workgroup has posted its slice of H after `epoch` epochs of fit `fit`
__device__ __forceinline__ bool multi_wait_epoch_image(const MultiBox *box, int fit, int epoch, int lane, TeamCtl *ctl) {
    const unsigned long long need = ((unsigned long long)(fit + 1) << 24) | (unsigned long long)epoch;
    for (int spin = 0;; ++spin) {
        const unsigned long long v = lane < box->G ? xld(box->ctl + 3 * box->G + lane) : ~0ull;
        if (__ballot(v >= need) == ~uint64_t(0)) return true;
        if (spin > XSPIN || (spin % 64 == 63 && multi_aborted(box))) {
            multi_abort(box, ctl);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// keeper wave k: columns [64 R k, 64 R (k + 1)) of H
// ---------------------------------------------------------------------------------------------------------------------
// NI: images kept in LDS (the multi-CU team's extractor reads them NI - 2 blocks late); BATCHES: keeper 0 publishes the index
// batches in LDS (remote workgroups of the multi-CU team, for their extractor wave; elsewhere the stager wave does);
// slice0: first column of this workgroup's slice of H (images are indexed relative to it)
template <int R, int K, bool DELTA, int NI = 2, bool BATCHES = false>
__device__ __forceinline__ void team_keeper(const double *__restrict__ Q, int ldq, int c, uint32_t seed, const double *w_lds,
                                            TeamLds<R, K, NI> &L, int k, int slice0 = 0) {
    const int lane = threadIdx.x & 63;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    // register r of lane l holds column colof(r): packed (R even) 128 (r/2) + 2 l + (r&1), so that one 16-byte load
    // fetches two of a lane's row elements; columns past c re-read the last pair: their H entries are never consumed
    constexpr bool PK = CP_CD_PACKED && (R % 2 == 0);
    const int rel0 = k * 64 * R, col0 = slice0 + rel0;
    auto colof = [&](int r) -> int { return col0 + (PK ? (r >> 1) * 2 * WAVE + 2 * lane + (r & 1) : r * WAVE + lane); };
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
    constexpr int U = 8;
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
    int *my_seq = &L.ctl->seqB[k];
    auto write_image = [&](int t) {
        double *im = L.img + (t % NI) * TeamLds<R, K, NI>::IMG + rel0;
        if (PK) {
#pragma unroll
            for (int r = 0; r < R; r += 2) *reinterpret_cast<double2 *>(im + r * WAVE + 2 * lane) = make_double2(H[r], H[r + 1]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) im[r * WAVE + lane] = H[r];
        }
        duo_store(my_seq, t + 1);
        CP_TRACE(3, NI > 2 && k == 0 && slice0 == 0 && t == TRACE_BLOCK + 1);
    };
    write_image(0);

    // every keeper runs the index stream itself (no hand-off on the rows' critical path); keeper 0 also publishes the
    // batches, with their duplicate scans, for the chain wave
    IdxStream rng;
    rng.init(seed, uint32_t(c), row_stride_bytes, lane);
    uint32_t prev_idx = 0xffffffffu;  // coordinates of the batch published before (none yet)
    auto publish_batch = [&](int kb) {
        if (!BATCHES || k != 0) return;
        L.ii[(kb % 3) * 64 + lane] = rng.idx;
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
            L.dup[kb & 3] = m;
            L.xdup[kb & 3] = mx;
        }
        prev_idx = rng.idx;
        duo_store(&L.ctl->batB, kb + 1);
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

    // (Measured and not kept: skipping the row of a coordinate that sits at zero inside its dead zone -- 35-45 % of the steps
    //  once the support has settled, the rows requested out of range so that the load counts stay static -- and fetching it
    //  on the spot when the coordinate turns non-zero after all.  The vector-memory queue returns in order, so such a late
    //  row waits behind the 16-32 prefetches in flight: c = 1024 279 -> 523-605 cycles per step, c = 2048 473 -> 602-632.)
    double rowA[B][R], rowB[B][R];
    auto fill = [&](double (&S)[B][R], uint32_t off_vec, int base) {
#pragma unroll
        for (int a = 0; a < B; ++a) {
            uint32_t roff = uint32_t(__builtin_amdgcn_readlane(int(off_vec), base + a));
#ifdef CP_CD_DEBUG_ALIAS  // timing experiment only (results are garbage): every row request aliases onto the first few rows
            roff %= uint32_t(CP_CD_DEBUG_ALIAS) * row_stride_bytes;
#endif
            load_row(S[a], roff);
        }
    };
    auto settle = [&](double (&S)[B][R]) {
#pragma unroll
        for (int a = 0; a < B; ++a)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(S[a][r]));
    };
    unsigned long long waitB = 0, blocksB = 0;
    // apply block t with the rows in S; false when the fit is over
    auto apply = [&](const double (&S)[B][R], int t) -> bool {
        const unsigned long long w0 = __builtin_readcyclecounter();
        const bool okw = team_wait(&L.ctl->seqA, t + 1, L.ctl, true);
        waitB += __builtin_readcyclecounter() - w0;
        ++blocksB;
        if (!okw) {
            if (lane == 0 && k == 0) {
                g_team_debug[4] = waitB;
                g_team_debug[5] = blocksB;
            }
            return false;
        }
        const double *pb = L.pub + (t & (TeamLds<R, K, NI>::PUB_RING - 1)) * 2 * B;
#pragma unroll
        for (int a = 0; a < B; ++a) {
            if (DELTA) {
                const double d_a = pb[a];  // same address in every lane: LDS broadcast
                if (moves_h(d_a)) {
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(d_a, S[a][r], H[r]);
                }
            } else {
                const double wo_a = pb[2 * a], wn_a = pb[2 * a + 1];
                if (moves_h(wo_a, wn_a)) {
#pragma unroll
                    for (int r = 0; r < R; ++r) H[r] = fma(wn_a, S[a][r], fma(-wo_a, S[a][r], H[r]));
                }
            }
        }
        write_image(t + 1);
        return true;
    };
    if constexpr (PK) {
        // The same two-set ring, but with the row loads issued from inline assembly and waited for with an explicit vmcnt.
        // The compiler's own wait insertion drains loop-carried loads completely (settle() below exists to give that drain a
        // harmless place), which makes every second block wait for the rows requested just before it: an L2 round trip per
        // two blocks while Q sits in L2 (c <= 512: the chain wave waited 28 cycles per step for the images), a microsecond
        // per block when the rows come from HBM (the remote keepers of the multi-CU team).  Here a set is waited for with
        // vmcnt(number of younger loads): two blocks of prefetch distance, never a drain.
        typedef double d2v __attribute__((ext_vector_type(2)));
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const uint64_t qa = reinterpret_cast<uint64_t>(Q);
        u32x4 rw;
        rw.x = __builtin_amdgcn_readfirstlane(uint32_t(qa));
        rw.y = __builtin_amdgcn_readfirstlane(uint32_t(qa >> 32) & 0xffffu);
        rw.z = __builtin_amdgcn_readfirstlane(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u);
        rw.w = 0x00020000u;
        constexpr int NL = B * (R / 2);   // loads per set
        d2v A2[B][R / 2], B2[B][R / 2];
        auto fill2 = [&](d2v (&S)[B][R / 2], uint32_t off_vec, int base) {
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const uint32_t roff = uint32_t(__builtin_amdgcn_readlane(int(off_vec), base + a));
                // (the row offset comes out of a v_readlane: a VALU-written SGPR needs five wait states before a vector-memory
                //  instruction reads it, and the compiler's hazard recognizer does not look into inline assembly)
                asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(S[a][0]) : "v"(colb[0]), "s"(rw), "s"(roff) : "memory");
#pragma unroll
                for (int r2 = 1; r2 < R / 2; ++r2)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(S[a][r2]) : "v"(colb[2 * r2]), "s"(rw), "s"(roff) : "memory");
            }
        };
        auto apply2 = [&](d2v (&S)[B][R / 2], int t) -> bool {
            const unsigned long long w0 = __builtin_readcyclecounter();
            const bool okw = team_wait(&L.ctl->seqA, t + 1, L.ctl, true);
            waitB += __builtin_readcyclecounter() - w0;
            ++blocksB;
            if (!okw) {
                if (lane == 0 && k == 0 && slice0 == 0) {
                    g_team_debug[4] = waitB;
                    g_team_debug[5] = blocksB;
                }
                return false;
            }
            CP_TTRACE(1, NI == 2 && k == 0 && t == TTRACE_BLOCK);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");   // everything but the other set's loads has landed
#pragma unroll
            for (int a = 0; a < B; ++a)
#pragma unroll
                for (int r2 = 0; r2 < R / 2; ++r2) asm volatile("" : "+v"(S[a][r2]));
            CP_TTRACE(2, NI == 2 && k == 0 && t == TTRACE_BLOCK);
            const double *pb = L.pub + (t & (TeamLds<R, K, NI>::PUB_RING - 1)) * 2 * B;
#pragma unroll
            for (int a = 0; a < B; ++a) {
                if (DELTA) {
                    const double d_a = pb[a];
                    if (moves_h(d_a)) {
#pragma unroll
                        for (int r = 0; r < R; ++r) H[r] = fma(d_a, S[a][r >> 1][r & 1], H[r]);
                    }
                } else {
                    const double wo_a = pb[2 * a], wn_a = pb[2 * a + 1];
                    if (moves_h(wo_a, wn_a)) {
#pragma unroll
                        for (int r = 0; r < R; ++r) H[r] = fma(wn_a, S[a][r >> 1][r & 1], fma(-wo_a, S[a][r >> 1][r & 1], H[r]));
                    }
                }
            }
#ifdef CP_CD_TEAM_TRACE
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(H[r]));
#endif
            CP_TTRACE(3, NI == 2 && k == 0 && t == TTRACE_BLOCK);
            write_image(t + 1);
            CP_TTRACE(4, NI == 2 && k == 0 && t == TTRACE_BLOCK);
            return true;
        };
        fill2(A2, off_cur, 0);
        for (int t = 0;; t += 2) {
            const int g = t & 7;
            fill2(B2, off_cur, (g + 1) * B);
            if (!apply2(A2, t)) break;
            if (g + 2 < 8)
                fill2(A2, off_cur, (g + 2) * B);
            else
                fill2(A2, off_nxt, 0);
            if (!apply2(B2, t + 1)) break;
            if (g + 2 >= 8) {  // batch roll-over
                off_cur = off_nxt;
                off_nxt = off_n2;
                rng.next_batch();
                ++batch;
                publish_batch(batch + 2);
                off_n2 = rng.off;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
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

// ---------------------------------------------------------------------------------------------------------------------
// stager wave: the index stream's batches (coordinates, duplicate scans) for the chain wave, and per block the couplings
// Q[ii_a, ii_j] the chain wave (and, in the multi-CU team, the gatherer waves) need: the 8 x 8 pairs inside the block (l = 0)
// and between the block and each of the LAGS blocks before it, as 1 + LAGS 64-lane gathers (lane = 8 a + j) written to LDS
// in the lay-out the chain wave reads with 16-byte loads.  Requested two blocks ahead of the store, stored up to three
// blocks ahead of the chain wave.  (Keeper 0 used to do this next to its rows: it then needed ~2000 cycles per block and
// the chain wave waited for its image in four blocks out of five.)
// ---------------------------------------------------------------------------------------------------------------------
template <class LDS, int LAGS>
__device__ __forceinline__ void team_stager(const double *__restrict__ Q, int ldq, int c, uint32_t seed, LDS &L) {
    const int lane = threadIdx.x & 63;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;
    TeamCtl *ctl = L.ctl;
    IdxStream rng;
    rng.init(seed, uint32_t(c), row_stride_bytes, lane);
    uint32_t prev_idx = 0xffffffffu;
    auto publish_batch = [&](int kb) {
        L.ii[(kb % LDS::II_RING) * 64 + lane] = rng.idx;
        bool dup = false, xd = false;
        const int bs = lane & ~(B - 1);
#pragma unroll
        for (int sft = 1; sft < B; ++sft) {
            const int other = __shfl(int(rng.idx), bs | ((lane + sft) & (B - 1)), WAVE);
            dup |= (uint32_t(other) == rng.idx);
        }
#pragma unroll
        for (int sft = 0; sft < B; ++sft) {
            const int src = ((bs - B) & 63) + sft;
            const int o_same = __shfl(int(rng.idx), src, WAVE), o_prev = __shfl(int(prev_idx), src, WAVE);
            xd |= uint32_t(bs == 0 ? o_prev : o_same) == rng.idx;
        }
        const uint64_t m = __ballot(dup), mx = __ballot(xd);
        if (lane == 0) {
            L.dup[kb & 3] = m;
            L.xdup[kb & 3] = mx;
        }
        prev_idx = rng.idx;
        duo_store(&ctl->batB, kb + 1);
    };
    publish_batch(0);
    rng.next_batch();
    publish_batch(1);
    rng.next_batch();
    publish_batch(2);
    int published = 3;
    auto idx_of = [&](int v) -> uint32_t { return L.ii[((v >> 6) % LDS::II_RING) * 64 + (v & 63)]; };
    const int a = lane >> 3, j = lane & 7;
    auto request = [&](int blk, double (&qv)[LAGS + 1]) {
        const uint32_t col = idx_of(8 * blk + j) * 8u;
        qv[0] = load_q(rsrc, j > a ? idx_of(8 * blk + a) * row_stride_bytes + col : OOB, 0u);
#pragma unroll
        for (int l = 1; l <= LAGS; ++l)
            qv[l] = load_q(rsrc, blk >= l ? idx_of(8 * (blk - l) + a) * row_stride_bytes + col : OOB, 0u);
    };
    auto store = [&](int blk, const double (&qv)[LAGS + 1]) {
        double *dst = L.cpl + (blk & 3) * (B * LDS::CPL_REC) + j * LDS::CPL_REC + a;
#pragma unroll
        for (int l = 0; l <= LAGS; ++l) dst[l * B] = qv[l];
        duo_store(&ctl->cplSeq, blk + 1);
    };
    double rq[3][LAGS + 1];
    request(0, rq[0]);
    request(1, rq[1]);
    for (int blk = 0;; blk += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int b = blk + u;
            request(b + 2, rq[(u + 2) % 3]);
            // the record of block b - 4 is dead once that block is complete
            if (!team_wait(&ctl->seqA, b - 3, ctl, true)) return;
            store(b, rq[u]);
            // batch kb's slot is free once batch kb - 4 is history: keep the batches of the current block + 2 published
            const int done = duo_load(&ctl->seqA);
            while (done >= 8 * (published - 2)) {
                rng.next_batch();
                publish_batch(published);
                ++published;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// chain wave
// ---------------------------------------------------------------------------------------------------------------------
// MK: three-operation correctly rounded division (see the header); RECIP: multiply by the reciprocal (CP_CD_RECIPROCAL,
// not bit-identical to sklearn's division)
// MULTI (multi-CU team, below): the block's starting H values come from the gatherer wave's ring (L.hn) instead of the
// keepers' LDS images, the epoch-end image from the remote workgroups through global memory (box, fit)
template <class LDS, int R, int K, bool RECIP, bool DELTA, bool MK, bool MULTI = false>
__device__ __forceinline__ void team_chain(const double *__restrict__ Q, int ldq, int c, double alpha, double beta,
                                           int max_iter, double tol_scaled, double d_w_tol, double y_norm2, double *w_lds,
                                           const double *feat, LDS &L, const MultiBox *box = nullptr, int fit = 0) {
    const int lane = threadIdx.x & 63;
    const uint32_t row_stride_bytes = uint32_t(ldq) * 8u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(Q), 0, int(uint32_t(c - 1) * row_stride_bytes + uint32_t(c) * 8u), 0x00020000);
    const uint32_t rel = uint32_t(lane) & uint32_t(B - 1);
    constexpr uint32_t OOB = 0x80000000u;  // beyond num_records with or without the row offset, no 32-bit wrap
    constexpr int IMG = 64 * R * K;
    constexpr int PUBM = LDS::PUB_RING - 1;
    const int ncol64 = MULTI ? (c + WAVE - 1) / WAVE : R * K;
    TeamCtl *ctl = L.ctl;

    struct Batch {
        uint32_t ii;
        double q, Qd, den, rden;
        uint64_t dupmask, xdupmask;
    };
    auto load_batch = [&](Batch &bt, int kb) -> bool {
        if (!team_wait(&ctl->batB, kb + 1, ctl, false)) return false;
        bt.ii = L.ii[(kb % LDS::II_RING) * 64 + lane];
        bt.dupmask = L.dup[kb & 3];
        bt.xdupmask = L.xdup[kb & 3];
        const double2 qQ = *reinterpret_cast<const double2 *>(feat + 4 * bt.ii);
        const double2 dr = *reinterpret_cast<const double2 *>(feat + 4 * bt.ii + 2);
        bt.q = qQ.x;
        bt.Qd = qQ.y;
        bt.den = dr.x;
        bt.rden = dr.y;
        return true;
    };
    struct CSet {
        double qc[B];  // Q[ii_a, ii_lane] within the block (0 for finished lanes)
        double qx[B];  // Q[ii_a(previous block), ii_lane]
    };
    // the couplings of block blk, staged by keeper 0 (lane l reads the record of position l & 7: eight distinct addresses
    // per instruction, broadcast within each group of eight lanes)
    auto fill = [&](CSet &S, int blk) {
        const double *src = L.cpl + (blk & 3) * (B * LDS::CPL_REC) + rel * LDS::CPL_REC;
        auto read = [&]() {
#pragma unroll
            for (int a = 0; a < B; a += 2) {
                const double2 v = *reinterpret_cast<const double2 *>(src + a);
                const double2 x = *reinterpret_cast<const double2 *>(src + B + a);
                S.qc[a] = v.x;
                S.qc[a + 1] = v.y;
                S.qx[a] = x.x;
                S.qx[a + 1] = x.y;
            }
        };
        // the flag read and the data reads go out back to back (the LDS executes them in order, so data read behind a flag
        // that says "staged" is the staged data): one LDS round trip instead of two; not staged yet (rare) -> wait, re-read
        const int staged = duo_load(&ctl->cplSeq);
        read();
        if (__builtin_expect(staged < blk + 1, 0)) {
            team_wait(&ctl->cplSeq, blk + 1, ctl, false);
            read();
        }
    };
    // d_out: |tmp| - alpha before the clamp -- its magnitude is how far the coefficient is from the edge of its dead zone
    auto soft_step = [&](const Batch &bt, double wo_v, double Hs_v, double &d_out) -> double {
        const double Hp = fma(-wo_v, bt.Qd, Hs_v);
        const double tmp = bt.q - Hp;
        d_out = fabs(tmp) - alpha;
        if (MK) {
            const double m = fmax(d_out, 0.0);               // >= +0
            const double q0 = m * bt.rden;
            const double rem = fma(-bt.den, q0, m);
            return copysign(fma(rem, bt.rden, q0), tmp);     // den > 0: the quotient carries tmp's sign, zeros included
        }
        const double thr = copysign(fmax(d_out, 0.0), tmp);
        return RECIP ? thr * bt.den : thr / bt.den;
    };

    Batch cur, nxt;
    if (!load_batch(cur, 0) || !load_batch(nxt, 1)) return;
    int batch = 0;
    int n_iter = 0, f = 0;
    double wmax_v = 0.0, dmax_v = 0.0;
    double gap_out = tol_scaled + 1.0;
    // tie sentinels: L.edge[i] = | |tmp| - alpha | of coordinate i's latest update (every lane's LAST evaluation in a block is
    // its own final update: lanes past their step reproduce it); gap_margin = the closest the duality gap came to its threshold
    const double big = __builtin_huge_val();
    double gap_margin = big;
    double dp0[B], dp1[B];  // what the previous block published (DELTA: dp0 = differences; else dp0 = w_old, dp1 = w_new)
#pragma unroll
    for (int a = 0; a < B; ++a) dp0[a] = dp1[a] = 0.0;

    // the image after t_done blocks is complete: dual gap from it (same arithmetic as the one-wave form)
    auto epoch_end = [&](int t_done) -> bool {
        const double w_max = wave_max(wmax_v), d_w_max = wave_max(dmax_v);
        bool done = false;
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            const double *im = nullptr;
            if constexpr (MULTI) {
                if (!multi_wait_epoch_image(box, fit, n_iter + 1, lane, ctl)) return true;
            } else {
                if (!team_wait_images<K>(ctl, t_done + 1, lane)) return true;
                im = L.img + (t_done & 1) * IMG;
            }
            double s_qw = 0, s_wh = 0, s_ww = 0, s_l1 = 0, m_xta = 0;
#pragma unroll 4
            for (int r = 0; r < ncol64; ++r) {
                const int col = r * WAVE + lane;
                if (col < c) {
                    const double wv = w_lds[col], qv = feat[4 * col], hv = MULTI ? xld_d(box->himg + col) : im[col];
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
            gap_margin = fmin(gap_margin, fabs(gap - tol_scaled));
            if (gap < tol_scaled) done = true;
        }
        ++n_iter;
        wmax_v = 0.0;
        dmax_v = 0.0;
        f = 0;
        return done || n_iter == max_iter;
    };

    unsigned long long waitA = 0, repairs = 0;
    bool dead = false;   // MULTI: a wait on the other waves / workgroups ran out
    // per-lane inputs of a block, fetched while the block before it is still running
    struct Pre {
        double Hn;   // image part of H[ii] (the previous block's 8 updates are added through qx)
        double wo;   // w[ii]
        int seq;     // lane l: keeper (l % K)'s image counter as read just before Hn (evaluated when the block ends, so that
                     // the three reads go out back to back)
    };
    auto prefetch = [&](Pre &pr, const Batch &nb, int t_next) {  // for block t_next: image t_next - 1
        if constexpr (MULTI) {
            pr.seq = duo_load(&ctl->hnSlot[t_next & 3]);
            pr.Hn = L.hn[(t_next & 3) * B + rel];
        } else {
            pr.seq = duo_load(&ctl->seqB[lane % K]);
            pr.Hn = (L.img + ((t_next - 1) & 1) * IMG)[nb.ii];
        }
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
                if (DELTA) {
                    if (moves_h(dp0[a])) Hs_v = fma(dp0[a], S.qx[a], Hs_v);
                } else {
                    if (moves_h(dp0[a], dp1[a])) Hs_v = fma(dp1[a], S.qx[a], fma(-dp0[a], S.qx[a], Hs_v));
                }
            }
        }
        double wo_v = pc.wo;
        const uint64_t blockmask = ((uint64_t(1) << B) - 1) << base;
        uint64_t wmask = blockmask;
        const bool has_dup = (bt.dupmask & blockmask) != 0;
        double wn_keep = 0.0, p0_v = 0.0, p1_v = 0.0;  // what this lane publishes for its own step
        double d_v = big;                              // |tmp| - alpha of this lane's last evaluation
        if (!has_dup) {
            // every lane evaluates "its" update against its private H; lane la's is the one that counts at step a, lanes
            // before it reproduce their final value (their couplings to this and later steps read 0).  (With CP_CD_SKIP_CHAIN
            // the evaluation is repeated only after a step that moved H; by default `moved` is constant true.)
            double wn_v = soft_step(bt, wo_v, Hs_v, d_v);
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                bool moved;
                if (DELTA) {
                    const double d_a = read_lane(wn_v - wo_v, la);
                    dp0[a] = d_a;
                    moved = moves_c(d_a);
                    if (moved) Hs_v = fma(d_a, S.qc[a], Hs_v);
                } else {
                    const double wo_a = read_lane(wo_v, la), wn_a = read_lane(wn_v, la);
                    dp0[a] = wo_a;
                    dp1[a] = wn_a;
                    moved = moves_c(wo_a, wn_a);
                    if (moved) Hs_v = fma(wn_a, S.qc[a], fma(-wo_a, S.qc[a], Hs_v));
                }
                if (moved && a + 1 < B) wn_v = soft_step(bt, wo_v, Hs_v, d_v);
                if (a == PF) {
                    CP_TTRACE(5, !MULTI && t == TTRACE_BLOCK + 1);
                    prefetch(pn, nb, t + 1);
                }
            }
            wn_keep = wn_v;
            p0_v = DELTA ? wn_v - wo_v : wo_v;
            p1_v = wn_v;
        } else {  // a coordinate repeats inside the block: later visits must see the earlier result
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const int la = base + a;
                double d_a;
                const double wn_v = soft_step(bt, wo_v, Hs_v, d_a);
                const bool mine = lane == la;
                d_v = mine ? d_a : d_v;
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
            double *pb = L.pub + (t & PUBM) * 2 * B;
            if (DELTA) {
                pb[rel] = p0_v;
            } else {
                pb[2 * rel] = p0_v;
                pb[2 * rel + 1] = p1_v;
            }
            dmax_v = fmax(dmax_v, fabs(wn_keep - wo_v));
            wmax_v = fmax(wmax_v, fabs(wn_keep));
        }
        if ((wmask >> lane) & 1) {
            w_lds[bt.ii] = wn_keep;
            L.edge[bt.ii] = fabs(d_v);
        }
        duo_store(&ctl->seqA, t + 1);
        CP_TTRACE(0, !MULTI && t == TTRACE_BLOCK);
        CP_TRACE(0, MULTI && t == TRACE_BLOCK);
        CP_TRACE(7, MULTI && t == TRACE_BLOCK + XLAG);   // the block before the one the traced values feed
        CP_TTRACE(7, !MULTI && t == TTRACE_BLOCK + 1);
        // rare repairs of the prefetch: a keeper had not published image t yet, or the next block revisits a coordinate
        // this block just changed
        if (__ballot(pn.seq >= t + (MULTI ? 2 : 1)) != ~uint64_t(0)) {   // image t (MULTI: the values of block t + 1) had not
            const unsigned long long w0 = __builtin_readcyclecounter();  // been published when Hn was read
            if constexpr (MULTI) {
                dead = !team_wait(&ctl->hnSlot[(t + 1) & 3], t + 2, ctl, true);
                pn.Hn = L.hn[((t + 1) & 3) * B + rel];
            } else {
                team_wait_images<K>(ctl, t + 1, lane);
                pn.Hn = (L.img + (t & 1) * IMG)[nb.ii];
            }
            waitA += __builtin_readcyclecounter() - w0;
            CP_TTRACE(6, !MULTI && t == TTRACE_BLOCK + 1);
            ++repairs;
        }
        if ((nb.xdupmask >> nbase) & ((uint64_t(1) << B) - 1)) pn.wo = w_lds[nb.ii];
    };

    CSet SA, SB;
    Pre pa, pb2;
    fill(SA, 0);
    if constexpr (MULTI) {
        dead = !team_wait(&ctl->hnSlot[0], 1, ctl, true);
        pa.Hn = L.hn[rel];
    } else {
        team_wait_images<K>(ctl, 1, lane);
        pa.Hn = L.img[cur.ii];
    }
    pa.wo = w_lds[cur.ii];
    pa.seq = 1;
    for (int t = 0;; t += 2) {  // blocks t (set A) and t+1 (set B); 8 blocks per batch
        const int g = t & 7;
        // ---- block t ----
        fill(SB, t + 1);
        compute(SA, cur, g * B, t, pa, t > 0, cur, (g + 1) * B, pb2);
        f += B;
        if (MULTI && dead) break;
        if (f == c && epoch_end(t + 1)) break;
        // ---- block t+1 ----
        const bool roll = g + 2 >= 8;
        fill(SA, t + 2);
        if (!roll) {
            compute(SB, cur, (g + 1) * B, t + 1, pb2, true, cur, (g + 2) * B, pa);
        } else {
            compute(SB, cur, (g + 1) * B, t + 1, pb2, true, nxt, 0, pa);
        }
        f += B;
        if (MULTI && dead) break;
        if (f == c && epoch_end(t + 2)) break;
        if (roll) {
            cur = nxt;
            ++batch;
            if (!load_batch(nxt, batch + 1)) break;
        }
    }
    duo_store(&ctl->stop, 1);
    int cnt = 0;
    double edge_min = big;
#pragma unroll 4
    for (int r = 0; r < ncol64; ++r) {
        const int col = r * WAVE + lane;
        cnt += (col < c && w_lds[col] != 0.0) ? 1 : 0;
        if (col < c) edge_min = fmin(edge_min, L.edge[col]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o, WAVE);
        edge_min = fmin(edge_min, __shfl_xor(edge_min, o, WAVE));
    }
    if (lane == 0) {
        g_team_debug[2] = waitA;
        g_team_debug[3] = repairs;
        ctl->gap = gap_out;
        ctl->n_iter = duo_load(&ctl->err) ? -1 : n_iter;
        ctl->nnz = cnt;
        ctl->edge_margin = edge_min == big ? -1.0 : edge_min / alpha;
        ctl->gap_margin = gap_margin == big ? -1.0 : gap_margin / tol_scaled;
    }
}

// all K + 2 waves of the workgroup (chain, K keepers, stager) call this; returns the same FitOut in all threads.  flags: CP_CD_RECIPROCAL | CP_CD_DELTA
// (0: sklearn's operation order with the three-operation division where it is exact; 3: both rounding variants)
template <int R, int K>
__device__ __forceinline__ FitOut team_fit(int flags, int exact_div, const double *__restrict__ Q, int ldq, int c, double alpha,
                                           double beta, uint32_t seed, int max_iter, double tol_scaled, double d_w_tol,
                                           double y_norm2, double *w_lds, const double *feat, double *team_base) {
    TeamLds<R, K> L;
    L.bind(team_base);
    if (threadIdx.x == 0) {
        L.ctl->seqA = 0;
        L.ctl->batB = 0;
        L.ctl->stop = 0;
        L.ctl->err = 0;
        L.ctl->cplSeq = 0;
        for (int k = 0; k < KMAX; ++k) L.ctl->seqB[k] = 0;
    }
    for (int j = threadIdx.x; j < TeamLds<R, K>::IMG; j += blockDim.x) L.edge[j] = __builtin_huge_val();
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const bool fast = (flags & (CP_CD_RECIPROCAL | CP_CD_DELTA)) == (CP_CD_RECIPROCAL | CP_CD_DELTA);
    if (wave == K + 1) {
        team_stager<TeamLds<R, K>, 1>(Q, ldq, c, seed, L);
    } else if (wave > 0) {
        if (fast)
            team_keeper<R, K, true>(Q, ldq, c, seed, w_lds, L, wave - 1);
        else
            team_keeper<R, K, false>(Q, ldq, c, seed, w_lds, L, wave - 1);
    } else {
#define CP_CHAIN_ARGS Q, ldq, c, alpha, beta, max_iter, tol_scaled, d_w_tol, y_norm2, w_lds, feat, L
        const bool mk = !exact_div && L.ctl->mk_ok && alpha >= 0x1p-400 && alpha <= 0x1p400;
        if (fast)
            team_chain<TeamLds<R, K>, R, K, true, true, false>(CP_CHAIN_ARGS);
        else if (mk)
            team_chain<TeamLds<R, K>, R, K, false, false, true>(CP_CHAIN_ARGS);
        else
            team_chain<TeamLds<R, K>, R, K, false, false, false>(CP_CHAIN_ARGS);
#undef CP_CHAIN_ARGS
    }
    __syncthreads();
    FitOut out;
    out.gap = L.ctl->gap;
    out.n_iter = L.ctl->n_iter;
    out.nnz = L.ctl->nnz;
    out.edge_margin = L.ctl->edge_margin;
    out.gap_margin = L.ctl->gap_margin;
    __syncthreads();
    return out;
}

// LDS image: feat[4 c] | w[c] | team area.  feat[4 j + {0,1,2,3}] = { q[j], Q[j,j], d = Q[j,j] + beta (its reciprocal with
// CP_CD_RECIPROCAL), RN(1 / d) }; zero-diagonal features are skipped by sklearn (_cd_fast.pyx:651): here their update is a
// no-op through d = 1 (their row of Q, q and H entry are all zero).
template <class LDS>
__device__ __forceinline__ void team_load_features(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                                   const double *__restrict__ w_in, int c, double l2, int flags,
                                                   double *w_lds, double *feat, LDS &L) {
    if (threadIdx.x == 0) L.ctl->mk_ok = 1;
    __syncthreads();
    bool ok = true;
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        const double dj = Q[size_t(j) * ldq + j];
        w_lds[j] = w_in ? w_in[j] : 0.0;
        const double d = dj == 0.0 ? 1.0 : dj + l2;
        feat[4 * j + 0] = q[j];
        feat[4 * j + 1] = dj;
        feat[4 * j + 2] = (flags & CP_CD_RECIPROCAL) ? (dj == 0.0 ? 0.0 : 1.0 / d) : d;
        feat[4 * j + 3] = 1.0 / d;
        ok = ok && d >= 0x1p-400 && d <= 0x1p400;
    }
    if (!ok) L.ctl->mk_ok = 0;
    __syncthreads();
}

template <int R, int K>
__global__ void __launch_bounds__(64 * (K + 2)) k_cd_fit_team(const double *__restrict__ Q, int ldq, const double *__restrict__ q,
                                                              const double *__restrict__ stats, int c, double l1, double l2,
                                                              uint32_t seed, int max_iter, double tol, int flags, int exact_div,
                                                              double *__restrict__ w, DevResult *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *feat = smem, *w_lds = smem + 4 * c, *team = smem + 5 * c;
    TeamLds<R, K> L0;
    L0.bind(team);
    team_load_features(Q, ldq, q, w, c, l2, flags, w_lds, feat, L0);
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    FitOut o = team_fit<R, K>(flags, exact_div, Q, ldq, c, l1, l2, seed, max_iter, tol_scaled, tol, y_norm2, w_lds, feat, team);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        g_team_debug[0] = t1 - t0;
        g_team_debug[1] = (unsigned long long)o.n_iter * (unsigned long long)c;
    }
    for (int j = threadIdx.x; j < c; j += blockDim.x) w[j] = w_lds[j];
    if (threadIdx.x == 0) {
        res->gap = o.gap;
        res->tol_scaled = tol_scaled;
        res->n_iter = o.n_iter;
        res->nnz = o.nnz;
        res->edge_margin = o.edge_margin;
        res->gap_margin = o.gap_margin;
    }
}

// Whole alpha search of lib/decompose.py:490-525, one search per workgroup (blockIdx.x picks the argument block)
// spread >= 0: the launch has 8 workgroups per search and only workgroup (spread + job) % 8 of a search's group works --
// workgroup L of a launch is placed on XCD L % 8 (observed, not promised: nothing but speed depends on it), so concurrent
// searches, each a single workgroup whose Gram wants to stay in ONE XCD's L2, land on different XCDs instead of all on
// XCD 0.  prio: the waves raise their issue priority over co-resident throughput kernels.
template <int R, int K>
__global__ void __launch_bounds__(64 * (K + 2)) k_cd_search_team(CdSearchBatch b, int exact_div, int spread, int prio) {
    int job = blockIdx.x;
    if (spread >= 0) {
        job = blockIdx.x >> 3;
        if (int(blockIdx.x & 7) != ((spread + job) & 7)) return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    const CdSearchArgs &a = b.a[job];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int c = a.c;
    double *feat = smem, *w_lds = smem + 4 * c, *team = smem + 5 * c;
    TeamLds<R, K> L0;
    L0.bind(team);
    team_load_features(a.Q, a.ldq, a.q, nullptr, c, 0.0, a.flags, w_lds, feat, L0);
    const double y_norm2 = a.stats[0];
    const double tol_scaled = a.tol * y_norm2;
    int fit = 0;
    double left = 0.0, right = a.right0, alpha = a.right0;
    bool bracketing = true, ok = false;
    while (fit < a.max_fits) {
        alpha = bracketing ? right : (left + right) / 2;
        FitOut o = team_fit<R, K>(a.flags, exact_div, a.Q, a.ldq, c, alpha * a.M, 0.0, a.seeds[fit], a.max_iter, tol_scaled,
                                  a.tol, y_norm2, w_lds, feat, team);
        if (threadIdx.x == 0) {
            a.log[fit].gap = o.gap;
            a.log[fit].tol_scaled = tol_scaled;
            a.log[fit].n_iter = o.n_iter;
            a.log[fit].nnz = o.nnz;
            a.log[fit].edge_margin = o.edge_margin;
            a.log[fit].gap_margin = o.gap_margin;
            a.log_alpha[fit] = alpha;
        }
        ++fit;
        if (o.n_iter < 0) break;   // a hand-off between the waves failed: w and H are stale; reported through the log (n_iter = -1)
        const double tmp = double(o.nnz);
        if (bracketing) {  // decompose.py:502-515
            if (tmp < a.rank)
                bracketing = false;
            else
                right *= 2;
        } else {  // decompose.py:516-525
            if (tmp > a.rbound)
                left = alpha;
            else if (tmp < a.lbound)
                right = alpha;
            else {
                ok = true;
                break;
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        a.w[j] = w_lds[j];
        a.w_host[j] = w_lds[j];
    }
    if (threadIdx.x == 0) {
        *a.fits_used = ok ? fit : -fit;  // negative: ran out of pre-drawn seeds
        *a.alpha_out = alpha;
    }
}

// =====================================================================================================================
// multi-CU team: the keepers of a WIDE layer spread over several workgroups (= CUs)
// =====================================================================================================================
// Above c = 512 the one-workgroup team is bound by what ONE CU can pull through its vector L1: a step moves a whole row of Q
// (8 c bytes) and the L1 delivers <= 64 B / cycle -- 288 cycles per step at c = 2048 whatever the keepers do (measured 470,
// 428 with every row resident in L2).  Here the columns of H are cut in slices of 512, one REMOTE workgroup (receiver wave,
// extractor wave, two keeper waves) per slice, each on its own CU, and the HOME workgroup keeps only the serial part:
//   chain wave     team_chain<MULTI>: as in the one-workgroup team, but a block's starting H values come from L.hn;
//   stager wave    team_stager: the index stream, the batches and the couplings Q[ii_a, ii_j] of a block with itself and with
//                  the XLAG blocks before it;
//   forwarder wave copies what the chain wave publishes per block into the remotes' delta rings;
//   gatherer waves (three, taking the blocks in turn) collect the remotes' H values of a block's eight coordinates and bring
//                  them up to date with the blocks the remotes had not seen.
// The workgroups talk through global memory (MultiBox; ~630 ns per round trip inside an XCD, 850-1100 ns across XCDs:
// tools/probes/xwg_pingpong.hip), far more than the ~0.9 us a block takes.  So the remotes run XLAG blocks behind: the value
// a remote posts for block v is H[ii_j(v)] after blocks < v - XLAG (image v - XLAG), the gatherer applies blocks
// v - XLAG .. v - 2 and the chain wave block v - 1 -- each H entry still sees the same fma sequence in the same order as in the
// oracle (keeper for the old blocks, then gatherer, then chain wave), so w, n_iter and the zero pattern stay bit-identical.
// Flow control: every remote posts a (possibly empty) record for EVERY block and the gatherer waits for all G of them, so
// nobody can be more than XLAG + 1 blocks ahead of anybody else and the rings of RING = 8 slots are never overrun.
constexpr int XR = 4, XK = 2;              // a remote workgroup: two keeper waves of 256 columns each
constexpr int XSLICE = 64 * XR * XK;       // columns of H per remote workgroup
constexpr int XNI = XLAG + 2;              // images a remote keeps: the extractor reads image i while the keepers may be at i + XLAG + 1
constexpr int XMAXG = 4;                   // remote workgroups at most (c <= 2048)
constexpr int XGW = 3;                     // gatherer waves of the home workgroup (wave w takes the blocks v = w mod XGW)
#ifndef CP_CD_MULTI_WAVES
#define CP_CD_MULTI_WAVES (4 + XGW)   // wave 4 idles: it would share SIMD 0 with the chain wave (waves go to SIMD index % 4)
#endif
constexpr int XWAVES = CP_CD_MULTI_WAVES;  // waves per workgroup: home = chain, stager, forwarder, gatherers; remote = 2 + XK
constexpr unsigned long long EXITV = ~0ull;
static_assert(XLAG >= 2 && XLAG <= 7 && RING >= XLAG + 3 && XWAVES >= 2 + XK, "ring depth");
typedef TeamLds<XR, XK, XNI> RemoteLds;

struct HomeLds {
    static constexpr int PUB_RING = 8, II_RING = 4, CPL_REC = (XLAG + 1) * B;
    double *edge;      // [cpad]
    double *pub;       // [8][2 B]
    double *cpl;       // [4][B][CPL_REC]: record of position j = { Q[ii_a(blk - l), ii_j(blk)] : l = 0 .. XLAG, a = 0 .. 7 } (l = 0: 0 for j <= a)
    double *hn;        // [4][B] starting H values of a block's coordinates (everything but the block before it applied)
    uint32_t *ii;      // [4][64]
    uint64_t *dup, *xdup;
    TeamCtl *ctl;
    static __host__ __device__ constexpr int doubles(int cpad) {
        return cpad + PUB_RING * 2 * B + 4 * B * CPL_REC + 4 * B + II_RING * 32 + 8 + int(sizeof(TeamCtl) / 8) + 2;
    }
    __device__ void bind(double *base, int cpad) {
        edge = base;
        pub = edge + cpad;
        cpl = pub + PUB_RING * 2 * B;
        hn = cpl + 4 * B * CPL_REC;
        ii = reinterpret_cast<uint32_t *>(hn + 4 * B);
        dup = reinterpret_cast<uint64_t *>(ii + II_RING * 64);
        xdup = dup + 4;
        ctl = reinterpret_cast<TeamCtl *>(xdup + 4);
    }
};

__device__ __forceinline__ void team_ctl_reset(TeamCtl *ctl) {
    ctl->seqA = 0;
    ctl->batB = 0;
    ctl->stop = 0;
    ctl->err = 0;
    ctl->cplSeq = 0;
    for (int k = 0; k < 4; ++k) ctl->hnSlot[k] = 0;
    for (int k = 0; k < KMAX; ++k) ctl->seqB[k] = 0;
}

// ---- home workgroup, forwarder wave -------------------------------------------------------------------------------------------
// what the chain wave published for block t -> slot t % RING of every remote's delta ring (lane = 16 g + i).  A slot is
// written only after it has been SEEN to hold SENT (its reader put that back RING blocks ago; checked while waiting for the
// block), so the hand-off needs no ordering between different words.
template <bool DELTA>
__device__ __forceinline__ void multi_forwarder(HomeLds &L, const MultiBox *box) {
    const int lane = threadIdx.x & 63, G = box->G, g = lane >> 4, i = lane & 15;
    TeamCtl *ctl = L.ctl;
    constexpr int NV = DELTA ? B : 2 * B;
    const bool active = g < G && i < NV;
    for (int t = 0;; ++t) {
        unsigned long long *slot = box->dl + (g * RING + (t % RING)) * 2 * B + i;
        bool free_seen = false;
        for (int spin = 0;; ++spin) {
            if (!free_seen) {
                const unsigned long long raw = active ? xld(slot) : SENT;
                free_seen = __ballot(raw != SENT) == 0;
            }
            if (free_seen && duo_load(&ctl->seqA) >= t + 1) break;
            if (duo_load(&ctl->stop)) {
                vm_drain();
                return;
            }
            if (spin > XSPIN || (spin % 256 == 255 && multi_aborted(box))) {
                multi_abort(box, ctl);
                vm_drain();
                return;
            }
            if (free_seen) __builtin_amdgcn_s_sleep(1);
        }
        if (active) xst_d(slot, L.pub[(t & 7) * 2 * B + i]);
        CP_TRACE(1, t == TRACE_BLOCK);
    }
}

// ---- home workgroup, gatherer waves ---------------------------------------------------------------------------
// block v: collect every remote's record (lane = 8 g + j), take each coordinate's value from its owner, apply the blocks
// v - XLAG .. v - 2 the remotes had not seen, publish.  The chain wave applies block v - 1 itself.  One pass costs more than
// a block of the chain wave (~1000 cycles per block applied), so XGW waves take the blocks in turn; the last block applied,
// v - 2, completes about 0.6 of a block before the chain wave's prefetch asks for the values (1.6 before block v starts):
// its couplings are read before the wait, its values right after.
template <bool DELTA>
__device__ __forceinline__ void multi_gatherer(HomeLds &L, const MultiBox *box, int w) {
    const int lane = threadIdx.x & 63, j = lane & 7, G = box->G;
    TeamCtl *ctl = L.ctl;
    const bool active = lane < 8 * G;
    auto idx_of = [&](int v) -> uint32_t { return L.ii[((v >> 6) & 3) * 64 + (v & 63)]; };
    auto slot_of = [&](int v) -> unsigned long long * { return box->hv + ((lane >> 3) * RING + (v % RING)) * B + j; };
    double Hs = 0.0;
    auto apply = [&](int u, const double (&qx)[B]) {
        const double *pb = L.pub + (u & 7) * 2 * B;
        if (DELTA) {
            double d[B];
#pragma unroll
            for (int a = 0; a < B; a += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(pb + a);
                d[a] = t2.x;
                d[a + 1] = t2.y;
            }
#pragma unroll
            for (int a = 0; a < B; ++a)
                if (moves_h(d[a])) Hs = fma(d[a], qx[a], Hs);
        } else {
            double2 p[B];
#pragma unroll
            for (int a = 0; a < B; ++a) p[a] = *reinterpret_cast<const double2 *>(pb + 2 * a);   // (w_old, w_new) of step a
#pragma unroll
            for (int a = 0; a < B; ++a)
                if (moves_h(p[a].x, p[a].y)) Hs = fma(p[a].y, qx[a], fma(-p[a].x, qx[a], Hs));
        }
    };
    auto couplings = [&](int v, int l, double (&qx)[B]) {
        const double *src = L.cpl + (v & 3) * (B * HomeLds::CPL_REC) + j * HomeLds::CPL_REC + l * B;
#pragma unroll
        for (int a = 0; a < B; a += 2) {
            const double2 t2 = *reinterpret_cast<const double2 *>(src + a);
            qx[a] = t2.x;
            qx[a + 1] = t2.y;
        }
    };
    for (int v = w;; v += XGW) {
        unsigned long long *slot = active ? slot_of(v) : box->ctl;
        CP_TRACE(8, v == TRACE_BLOCK + 1 + XLAG);
        unsigned long long mine = 0;
        int looks = 0;
        Looks<4> lk;
        const int got = poll_words(lk, slot, active, mine, [&](unsigned long long) -> bool {
            return duo_load(&ctl->stop) != 0 || (++looks % 1024 == 0 && multi_aborted(box));
        });
        if (got <= 0) {
            if (got < 0) multi_abort(box, ctl);
            poll_settle(lk);
            return;
        }
        if (active) xst(slot, SENT);
        CP_TRACE(5, v == TRACE_BLOCK + 1 + XLAG);
        if (!team_wait(&ctl->batB, (v >> 3) + 1, ctl, true) || !team_wait(&ctl->cplSeq, v + 1, ctl, true)) break;
        const int owner = int(idx_of(8 * v + j)) / XSLICE;
        Hs = __shfl(__longlong_as_double((long long)mine), owner * 8 + j, WAVE);
        double qx[B];
        if (!team_wait(&ctl->seqA, v - 2, ctl, true)) break;   // blocks <= v - 3 complete (v - 3 about a block ago)
        if (v >= XLAG) {   // steady state: the operands of (up to) three old blocks at a time, then their chain of fma
            constexpr int CH = 3;
#pragma unroll
            for (int l0 = XLAG; l0 >= 3; l0 -= CH) {
                double qo[CH][B];
                double po[CH][DELTA ? B : 2 * B];
#pragma unroll
                for (int o = 0; o < CH; ++o) {
                    const int l = l0 - o;
                    if (l < 3) continue;
                    couplings(v, l, qo[o]);
                    const double *pb = L.pub + ((v - l) & 7) * 2 * B;
#pragma unroll
                    for (int a = 0; a < (DELTA ? B : 2 * B); a += 2) {
                        const double2 t2 = *reinterpret_cast<const double2 *>(pb + a);
                        po[o][a] = t2.x;
                        po[o][a + 1] = t2.y;
                    }
                }
#pragma unroll
                for (int o = 0; o < CH; ++o) {
                    if (l0 - o < 3) continue;
#pragma unroll
                    for (int a = 0; a < B; ++a) {
                        if (DELTA) {
                            if (moves_h(po[o][a])) Hs = fma(po[o][a], qo[o][a], Hs);
                        } else {
                            if (moves_h(po[o][2 * a], po[o][2 * a + 1]))
                                Hs = fma(po[o][2 * a + 1], qo[o][a], fma(-po[o][2 * a], qo[o][a], Hs));
                        }
                    }
                }
            }
        } else {
            for (int u = 0; u <= v - 3; ++u) {
                couplings(v, v - u, qx);
                apply(u, qx);
            }
        }
        CP_TRACE(9, v == TRACE_BLOCK + 1 + XLAG);
        if (v >= 2) {   // block v - 2 has just finished, or is about to: its couplings first, then the wait
            couplings(v, 2, qx);
            if (!team_wait(&ctl->seqA, v - 1, ctl, true)) break;
            apply(v - 2, qx);
        }
        if (lane < B) L.hn[(v & 3) * B + j] = Hs;
        duo_store(&ctl->hnSlot[v & 3], v + 1);
        CP_TRACE(6, v == TRACE_BLOCK + 1 + XLAG);
        poll_settle(lk);
    }
    vm_drain();
}

// ---- remote workgroup, wave 0 -------------------------------------------------------------------------------------------
template <bool DELTA>
__device__ __forceinline__ void multi_receiver(RemoteLds &L, const MultiBox *box, int g, int fit) {
    const int lane = threadIdx.x & 63, G = box->G;
    TeamCtl *ctl = L.ctl;
    constexpr int NV = DELTA ? B : 2 * B;
    // lanes < NV: the block's values; lane 32: "fit stopped"; lane 33: abort
    for (int t = 0;; ++t) {
        unsigned long long *slot = box->dl + (g * RING + (t % RING)) * 2 * B + lane;
        const unsigned long long *addr = lane < NV ? slot : lane == 32 ? box->ctl + G + g : lane == 33 ? box->ctl + 4 * G : box->ctl;
        unsigned long long raw = 0;
        Looks<4> lk;
        const int got = poll_words(lk, addr, lane < NV, raw, [&](unsigned long long v) -> bool {
            return __ballot((lane == 32 && v >= (unsigned long long)(fit + 1)) || (lane == 33 && v != 0)) != 0;
        });
        if (got <= 0) {
            if (got < 0)
                multi_abort(box, ctl);
            else
                duo_store(&ctl->stop, 1);
            poll_settle(lk);
            return;
        }
        if (lane < NV) {
            L.pub[(t & (RemoteLds::PUB_RING - 1)) * 2 * B + lane] = __longlong_as_double((long long)raw);
            xst(slot, SENT);
        }
        duo_store(&ctl->seqA, t + 1);
        CP_TRACE(2, g == 0 && t == TRACE_BLOCK);
        poll_settle(lk);   // the keepers are at work: nothing waits for this
    }
}

// ---- remote workgroup, wave 1 -------------------------------------------------------------------------------------------
__device__ __forceinline__ void multi_extractor(int c, RemoteLds &L, const MultiBox *box, int g, int fit) {
    const int lane = threadIdx.x & 63, G = box->G, slice0 = g * XSLICE;
    TeamCtl *ctl = L.ctl;
    constexpr int IMG = RemoteLds::IMG;
    const int blocks_per_epoch = c / B;
    auto idx_of = [&](int v) -> uint32_t { return L.ii[((v >> 6) % 3) * 64 + (v & 63)]; };
    auto slot_of = [&](int v) -> unsigned long long * { return box->hv + (g * RING + (v % RING)) * B + (lane & 7); };
    unsigned long long chk = xld(slot_of(0));   // the slot of the next record, looked at ahead of time: it must hold SENT
    bool ok = true;
    auto post = [&](int i, int v) {   // this slice's part of block v's starting values, from image i
        if (!ok || !team_wait(&ctl->batB, (v >> 3) + 1, ctl, true)) {
            ok = false;
            return;
        }
        unsigned long long *slot = slot_of(v);
        for (int spin = 0; __ballot(chk != SENT) != 0; ++spin) {
            if (duo_load(&ctl->stop) || spin > XSPIN) {
                if (spin > XSPIN) multi_abort(box, ctl);
                ok = false;
                return;
            }
            chk = xld(slot);
        }
        if (lane < B) {
            const int rel = int(idx_of(8 * v + lane)) - slice0;
            const double val = (rel >= 0 && rel < XSLICE) ? L.img[(i % XNI) * IMG + rel] : 0.0;
            xst_d(slot, val);
        }
        CP_TRACE(4, g == 0 && v == TRACE_BLOCK + 1 + XLAG);
        chk = xld(slot_of(v + 1));
    };
    for (int i = 0; ok; ++i) {
        for (int spin = 0;; ++spin) {   // image i of every keeper
            if (__ballot(duo_load(&ctl->seqB[lane % XK]) >= i + 1) == ~uint64_t(0)) break;
            if (duo_load(&ctl->stop)) {
                ok = false;
                break;
            }
            if (spin > (1 << 22)) {
                multi_abort(box, ctl);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) break;
        if (i == 0)
            for (int v = 0; v < XLAG; ++v) post(0, v);
        post(i, i + XLAG);
        if (ok && i > 0 && i % blocks_per_epoch == 0) {   // end of an epoch: the whole slice, for the chain wave's duality gap
            for (int col = lane; col < XSLICE; col += WAVE)
                if (slice0 + col < c) xst_d(box->himg + slice0 + col, L.img[(i % XNI) * IMG + col]);
            vm_drain();   // the image is in place before its announcement
            if (lane == 0)
                xst(box->ctl + 3 * G + g, ((unsigned long long)(fit + 1) << 24) | (unsigned long long)(i / blocks_per_epoch));
        }
    }
    vm_drain();
}

// ---- remote workgroup: all its fits ------------------------------------------------------------------------------------
__device__ __forceinline__ void multi_remote(const double *__restrict__ Q, int ldq, int c, int flags, const uint32_t *seeds,
                                             uint32_t seed_single, const MultiBox *box, int g, double *smem) {
    const int cpad = (c + 63) & ~63;
    double *w_lds = smem;
    RemoteLds L;
    L.bind(smem + cpad);
    const int wave = threadIdx.x >> 6, G = box->G;
    const bool fast = (flags & (CP_CD_RECIPROCAL | CP_CD_DELTA)) == (CP_CD_RECIPROCAL | CP_CD_DELTA);
    if (threadIdx.x == 0) xst(box->ctl + 4 * G + 1 + g, 1ull);   // resident from here to the end of the search
    for (int fit = 0;; ++fit) {
        if (threadIdx.x == 0) {
            // The home workgroup posts fit 0 only after EVERY remote of the search has reported in (multi_wait_resident,
            // ~20 s of patience): a sibling may get its CU much later than this one on a busy chip.  So this wait is as
            // patient as that one, and a remote that does give up says so (abort flag) instead of leaving silently --
            // the home workgroup would otherwise post fits to a workgroup that is gone and burn a full time-out per wait.
            int go = -1;
            bool gave_up = true;
            for (long spin = 0; spin < (10l << 20); ++spin) {
                const unsigned long long v = xld(box->ctl + g);
                if (v == EXITV) {
                    gave_up = false;
                    break;
                }
                if (v >= (unsigned long long)(fit + 1)) {
                    go = 1;
                    gave_up = false;
                    break;
                }
                if (spin % 64 == 63 && multi_aborted(box)) {
                    gave_up = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(64);
            }
            if (gave_up) multi_abort(box, L.ctl);
            L.ctl->nnz = go;
        }
        __syncthreads();
        if (L.ctl->nnz < 0) return;
        __syncthreads();
        for (int j = threadIdx.x; j < c; j += blockDim.x) w_lds[j] = xld_d(box->fitw + j);
        if (threadIdx.x == 0) team_ctl_reset(L.ctl);
        __syncthreads();
        const uint32_t seed = seeds ? seeds[fit] : seed_single;
        if (wave == 0) {
            if (fast)
                multi_receiver<true>(L, box, g, fit);
            else
                multi_receiver<false>(L, box, g, fit);
        } else if (wave == 1) {
            multi_extractor(c, L, box, g, fit);
        } else if (wave < 2 + XK) {
            if (fast)
                team_keeper<XR, XK, true, XNI, true>(Q, ldq, c, seed, w_lds, L, wave - 2, g * XSLICE);
            else
                team_keeper<XR, XK, false, XNI, true>(Q, ldq, c, seed, w_lds, L, wave - 2, g * XSLICE);
        }
        __syncthreads();
        // the fit is over: whatever is left in this workgroup's delta ring is void; tell the home workgroup
        for (int i = threadIdx.x; i < RING * 2 * B; i += blockDim.x) xst(box->dl + g * RING * 2 * B + i, SENT);
        vm_drain();
        __syncthreads();
        if (threadIdx.x == 0) xst(box->ctl + 2 * G + g, (unsigned long long)(fit + 1));
    }
}

// ---- home workgroup, once per launch: every remote workgroup is resident ----------------------------------------------------
// The waits inside a fit are bounded at about a second -- fine between workgroups that are all running, but a remote
// workgroup may have to wait for a free CU much longer than that on a chip full of other work.  So the home workgroup first
// waits, patiently (~20 s), until all its remotes have reported in; from then on everybody stays resident.
__device__ __forceinline__ void multi_wait_resident(const MultiBox *box, TeamCtl *ctl) {
    const int lane = threadIdx.x & 63, G = box->G;
    if (threadIdx.x < 64) {
        for (long spin = 0;; ++spin) {
            const unsigned long long v = lane < G ? xld(box->ctl + 4 * G + 1 + lane) : 1ull;
            if (__ballot(v == 0) == 0) break;
            if (spin > (10l << 20)) {
                multi_abort(box, ctl);
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
    __syncthreads();
}

// ---- home workgroup: one fit (all its threads call this; same FitOut in all of them) -------------------------------------
__device__ __forceinline__ FitOut multi_home_fit(int flags, int exact_div, const double *__restrict__ Q, int ldq, int c,
                                                 double alpha, double beta, uint32_t seed, int max_iter, double tol_scaled,
                                                 double d_w_tol, double y_norm2, double *w_lds, const double *feat, HomeLds &L,
                                                 const MultiBox *box, int fit) {
    const int cpad = (c + 63) & ~63, G = box->G, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < c; j += blockDim.x) xst_d(box->fitw + j, w_lds[j]);
    if (threadIdx.x == 0) team_ctl_reset(L.ctl);
    for (int j = threadIdx.x; j < cpad; j += blockDim.x) L.edge[j] = __builtin_huge_val();
    vm_drain();
    __syncthreads();
    if (threadIdx.x < G) xst(box->ctl + threadIdx.x, (unsigned long long)(fit + 1));   // w is in place: start
    const bool fast = (flags & (CP_CD_RECIPROCAL | CP_CD_DELTA)) == (CP_CD_RECIPROCAL | CP_CD_DELTA);
    if (wave == 0) {
#define CP_CHAIN_ARGS Q, ldq, c, alpha, beta, max_iter, tol_scaled, d_w_tol, y_norm2, w_lds, feat, L, box, fit
        const bool mk = !exact_div && L.ctl->mk_ok && alpha >= 0x1p-400 && alpha <= 0x1p400;
        if (fast)
            team_chain<HomeLds, 1, 1, true, true, false, true>(CP_CHAIN_ARGS);
        else if (mk)
            team_chain<HomeLds, 1, 1, false, false, true, true>(CP_CHAIN_ARGS);
        else
            team_chain<HomeLds, 1, 1, false, false, false, true>(CP_CHAIN_ARGS);
#undef CP_CHAIN_ARGS
    } else if (wave <= XGW) {   // waves 1 .. XGW: a SIMD each as long as XGW <= 3
        if (fast)
            multi_gatherer<true>(L, box, wave - 1);
        else
            multi_gatherer<false>(L, box, wave - 1);
    } else if (wave == XGW + 2) {   // (wave XGW + 1 = 4 idles;) the two light waves share SIMDs with the first two gatherers
        team_stager<HomeLds, XLAG>(Q, ldq, c, seed, L);
    } else if (wave == XGW + 3) {
        if (fast)
            multi_forwarder<true>(L, box);
        else
            multi_forwarder<false>(L, box);
    }
    __syncthreads();
    // nothing of this fit is on its way any more: stop the remotes, wait until they have cleared their rings, clear ours
    if (threadIdx.x < G) xst(box->ctl + G + threadIdx.x, (unsigned long long)(fit + 1));
    if (wave == 0) {
        for (int spin = 0;; ++spin) {
            const unsigned long long v = lane < G ? xld(box->ctl + 2 * G + lane) : ~0ull;
            if (__ballot(v >= (unsigned long long)(fit + 1)) == ~uint64_t(0)) break;
            if (spin > XSPIN || (spin % 64 == 63 && multi_aborted(box))) {
                multi_abort(box, L.ctl);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G * RING * B; i += blockDim.x) xst(box->hv + i, SENT);
    vm_drain();
    __syncthreads();
    FitOut out;
    out.gap = L.ctl->gap;
    out.n_iter = duo_load(&L.ctl->err) ? -1 : L.ctl->n_iter;
    out.nnz = L.ctl->nnz;
    out.edge_margin = L.ctl->edge_margin;
    out.gap_margin = L.ctl->gap_margin;
    __syncthreads();
    return out;
}

// the words of a job's mailbox: [dl | hv] start as SENT, the rest as 0 (k_multi_box_init)
__host__ __device__ inline size_t multi_box_sent_words(int G) { return size_t(G) * RING * 3 * B; }
__host__ __device__ inline size_t multi_box_words(int c, int G) {
    return multi_box_sent_words(G) + size_t(8 * XMAXG) + size_t((c + 63) & ~63) + size_t(G) * XSLICE;
}
__device__ __forceinline__ MultiBox multi_box_bind(unsigned long long *base, int c, int G) {
    MultiBox b;
    b.dl = base;
    b.hv = b.dl + size_t(G) * RING * 2 * B;
    b.ctl = b.hv + size_t(G) * RING * B;
    b.fitw = b.ctl + 8 * XMAXG;
    b.himg = b.fitw + ((c + 63) & ~63);
    b.G = G;
    b.slice = XSLICE;
    return b;
}
__global__ void k_multi_box_init(unsigned long long *base, size_t words_per_job, size_t sent_words, int n_jobs) {
    const size_t total = words_per_job * size_t(n_jobs);
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x)
        base[i] = (i % words_per_job) < sent_words ? SENT : 0ull;
}

struct MultiLaunch {
    unsigned long long *base;   // n_jobs mailboxes of `words` words each
    size_t words;
    int G, n_jobs, same_xcd;
    int xcd0;                   // same_xcd: job l lives on XCD (xcd0 + l) % 8 (rotated from launch to launch)
    int test_abort;             // cp_debug_cd_fail_multi: the home workgroup gives up before its first fit (tests of the fallback)
};
// workgroup -> (job, role): role 0 = home, 1 + g = remote g.  same_xcd: workgroup L of a launch lands on XCD L % 8 (observed,
// not promised: only the hand-off latency depends on it), so a job's 1 + G workgroups take ids that are 8 apart
__device__ __forceinline__ bool multi_role(const MultiLaunch &ml, int &job, int &role) {
    const int members = 1 + ml.G;
    if (ml.same_xcd) {
        const int x = blockIdx.x & 7, m = blockIdx.x >> 3;
        job = (m / members) * 8 + ((x - ml.xcd0) & 7);
        role = m % members;
    } else {
        job = blockIdx.x / members;
        role = blockIdx.x % members;
    }
    return job < ml.n_jobs;
}
inline int multi_grid(int n_jobs, int G, int same_xcd) {
    return same_xcd ? 8 * (1 + G) * ((n_jobs + 7) / 8) : n_jobs * (1 + G);
}

__global__ void __launch_bounds__(64 * XWAVES) k_cd_fit_multi(const double *__restrict__ Q, int ldq,
                                                                const double *__restrict__ q, const double *__restrict__ stats,
                                                                int c, double l1, double l2, uint32_t seed, int max_iter,
                                                                double tol, int flags, int exact_div, double *__restrict__ w,
                                                                DevResult *__restrict__ res, MultiLaunch ml) {
    int job, role;
    if (!multi_role(ml, job, role)) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const MultiBox box = multi_box_bind(ml.base, c, ml.G);
    if (role > 0) {
        multi_remote(Q, ldq, c, flags, nullptr, seed, &box, role - 1, smem);
        return;
    }
    const int cpad = (c + 63) & ~63;
    double *feat = smem, *w_lds = smem + 4 * c;
    HomeLds L;
    L.bind(smem + 5 * c, cpad);
    team_load_features(Q, ldq, q, w, c, l2, flags, w_lds, feat, L);
    multi_wait_resident(&box, L.ctl);
    if (ml.test_abort) {
        if (threadIdx.x == 0) multi_abort(&box, L.ctl);
        __syncthreads();
    }
    const double y_norm2 = stats[0];
    const double tol_scaled = tol * y_norm2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    FitOut o = multi_home_fit(flags, exact_div, Q, ldq, c, l1, l2, seed, max_iter, tol_scaled, tol, y_norm2, w_lds, feat, L, &box, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x < ml.G) xst(box.ctl + threadIdx.x, EXITV);
    if (threadIdx.x == 0) {
        g_team_debug[0] = t1 - t0;
        g_team_debug[1] = (unsigned long long)(o.n_iter > 0 ? o.n_iter : 0) * (unsigned long long)c;
    }
    if (o.n_iter >= 0)   // a failed hand-off leaves the caller's warm start as it was: the host re-runs the fit on one workgroup
        for (int j = threadIdx.x; j < c; j += blockDim.x) w[j] = w_lds[j];
    if (threadIdx.x == 0) {
        res->gap = o.gap;
        res->tol_scaled = tol_scaled;
        res->n_iter = o.n_iter;
        res->nnz = o.nnz;
        res->edge_margin = o.edge_margin;
        res->gap_margin = o.gap_margin;
    }
}

// the alpha search of lib/decompose.py:490-525 (as k_cd_search_team), one search per 1 + G workgroups
__global__ void __launch_bounds__(64 * XWAVES) k_cd_search_multi(CdSearchBatch b, int exact_div, MultiLaunch ml) {
    int job, role;
    if (!multi_role(ml, job, role)) return;
#ifdef CP_CD_MULTI_FULL_VGPR
    asm volatile("" ::: "v255");   // experiment: the kernel claims every register of its SIMDs
#endif
    const CdSearchArgs &a = b.a[job];
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int c = a.c;
    const MultiBox box = multi_box_bind(ml.base + size_t(job) * ml.words, c, ml.G);
    if (role > 0) {
        multi_remote(a.Q, a.ldq, c, a.flags, a.seeds, 0u, &box, role - 1, smem);
        return;
    }
    const int cpad = (c + 63) & ~63;
    double *feat = smem, *w_lds = smem + 4 * c;
    HomeLds L;
    L.bind(smem + 5 * c, cpad);
    team_load_features(a.Q, a.ldq, a.q, nullptr, c, 0.0, a.flags, w_lds, feat, L);
    multi_wait_resident(&box, L.ctl);
    if (ml.test_abort) {
        if (threadIdx.x == 0) multi_abort(&box, L.ctl);
        __syncthreads();
    }
    const double y_norm2 = a.stats[0];
    const double tol_scaled = a.tol * y_norm2;
    int fit = 0;
    double left = 0.0, right = a.right0, alpha = a.right0;
    bool bracketing = true, ok = false;
    while (fit < a.max_fits) {
        alpha = bracketing ? right : (left + right) / 2;
        FitOut o = multi_home_fit(a.flags, exact_div, a.Q, a.ldq, c, alpha * a.M, 0.0, a.seeds[fit], a.max_iter, tol_scaled, a.tol,
                                  y_norm2, w_lds, feat, L, &box, fit);
        if (threadIdx.x == 0) {
            a.log[fit].gap = o.gap;
            a.log[fit].tol_scaled = tol_scaled;
            a.log[fit].n_iter = o.n_iter;
            a.log[fit].nnz = o.nnz;
            a.log[fit].edge_margin = o.edge_margin;
            a.log[fit].gap_margin = o.gap_margin;
            a.log_alpha[fit] = alpha;
        }
        ++fit;
        if (o.n_iter < 0) break;   // a hand-off failed: reported through the log (n_iter = -1)
        const double tmp = double(o.nnz);
        if (bracketing) {  // decompose.py:502-515
            if (tmp < a.rank)
                bracketing = false;
            else
                right *= 2;
        } else {  // decompose.py:516-525
            if (tmp > a.rbound)
                left = alpha;
            else if (tmp < a.lbound)
                right = alpha;
            else {
                ok = true;
                break;
            }
        }
    }
    if (threadIdx.x < ml.G) xst(box.ctl + threadIdx.x, EXITV);
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += blockDim.x) {
        a.w[j] = w_lds[j];
        a.w_host[j] = w_lds[j];
    }
    if (threadIdx.x == 0) {
        *a.fits_used = ok ? fit : -fit;  // negative: ran out of pre-drawn seeds
        *a.alpha_out = alpha;
    }
}

struct TeamShape {
    int R, K;
};
TeamShape team_shape(int c) {
    // (twice the keepers with half the columns each -- (1,2) / (2,2) / (2,4) for c <= 128 / 256 / 512 -- measured equal: 276 /
    //  246 / 262 cycles per step against 251 / 255 / 253; the chain wave's dependent latency is the floor)
    if (c <= 64) return {1, 1};
    if (c <= 128) return {2, 1};
    if (c <= 256) return {4, 1};
    if (c <= 512) return {4, 2};
    if (c <= 1024) return {4, 4};
    return {6, 6};   // c <= 2304: 7 waves = at most two per SIMD, i.e. 256 registers per lane (nine waves of a (4, 8) team
                     // would share SIMDs three by three: 168 registers, and the keepers' row ring spills)
}
size_t team_lds_bytes(int c) {
    const TeamShape s = team_shape(c);
    const size_t img = size_t(64) * s.R * s.K;
    return (size_t(5) * c + 3 * img + 4 * 2 * B + 4 * B * 2 * B + 3 * 32 + 8 + sizeof(TeamCtl) / 8 + 2) * sizeof(double);
}

template <typename Kern>
hipError_t team_optin(Kern kernel, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
}

#define CP_TEAM_SWITCH(c_, CALL)                               \
    do {                                                       \
        const TeamShape ts_ = team_shape(c_);                  \
        if (ts_.K == 1 && ts_.R == 1) { CALL(1, 1); }          \
        else if (ts_.K == 1 && ts_.R == 2) { CALL(2, 1); }     \
        else if (ts_.K == 1) { CALL(4, 1); }                   \
        else if (ts_.K == 2) { CALL(4, 2); }                   \
        else if (ts_.K == 4) { CALL(4, 4); }                   \
        else { CALL(6, 6); }                                   \
    } while (0)

}  // namespace

// ---- entry points used by cd_gram.hip's dispatch ------------------------------------------------------------------
// The team kernels take: c a multiple of 8 (block size), c <= 2048, flags 0 (sklearn's operation order) or 3.
// CP_CD_TEAM=0 keeps the kernels of cd_gram.hip; CP_CD_EXACT_DIV=1 keeps the IEEE division on the chain.
bool cp_cd_team_wanted(int c, int flags) {
    static const bool on = !(getenv("CP_CD_TEAM") && atoi(getenv("CP_CD_TEAM")) == 0);
    const int f = flags & (CP_CD_RECIPROCAL | CP_CD_DELTA);
    return on && c % B == 0 && c >= B && c <= 64 * 6 * 6 && (f == 0 || f == (CP_CD_RECIPROCAL | CP_CD_DELTA));
}
static int team_exact_div() {
    static const int v = (getenv("CP_CD_EXACT_DIV") && atoi(getenv("CP_CD_EXACT_DIV")) != 0) ? 1 : 0;
    return v;
}
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// ---- multi-CU team: host side ------------------------------------------------------------------------------------------
// CP_CD_MULTI (default 1): layers of 512 < c <= 2048 channels run the multi-CU team (1 + ceil(c / 512) workgroups per search);
// CP_CD_MULTI_MIN_C moves the lower limit; CP_CD_MULTI_SAME_XCD (default 1): a search's workgroups share an XCD;
// CP_CD_MULTI_EXCLUSIVE (default 1): each of them asks for a whole CU's LDS.
static int multi_groups(int c) { return (c + XSLICE - 1) / XSLICE; }
static bool multi_wanted(int c) {
    static const int on = env_int("CP_CD_MULTI", 1), min_c = env_int("CP_CD_MULTI_MIN_C", 513);
    return on && c >= min_c && c > XSLICE / 2 && multi_groups(c) <= XMAXG && multi_groups(c) >= 1;
}
static size_t multi_lds_bytes(int c) {
    const int cpad = (c + 63) & ~63;
    const size_t home = size_t(5) * c + size_t(HomeLds::doubles(cpad)), remote = size_t(cpad) + size_t(RemoteLds::doubles());
    size_t lds = std::max(home, remote) * sizeof(double);
    static const int exclusive = env_int("CP_CD_MULTI_EXCLUSIVE", 1);
    if (exclusive) lds = std::max(lds, size_t(150) * 1024);
    return lds;
}
static int multi_prepare(cp_ctx *ctx, int c, int n_jobs, MultiLaunch &ml) {
    static const int same_xcd = env_int("CP_CD_MULTI_SAME_XCD", 1);
    ml.test_abort = ctx->cd_test_fail_multi ? 1 : 0;
    ml.G = multi_groups(c);
    ml.n_jobs = n_jobs;
    ml.same_xcd = same_xcd;
    static std::atomic<unsigned> next_xcd{0};   // concurrent searches (one launch per stream) take different XCDs
    ml.xcd0 = int(next_xcd.fetch_add(unsigned(n_jobs)) & 7u);
    ml.words = (multi_box_words(c, ml.G) + 31) & ~size_t(31);
    const size_t bytes = ml.words * size_t(n_jobs) * sizeof(unsigned long long);
    if (ctx->cd_box_bytes < bytes) {
        CP_HIP(ctx, cp_stream_wait(ctx));
        if (ctx->cd_box) CP_HIP(ctx, hipFree(ctx->cd_box));
        ctx->cd_box = nullptr;
        ctx->cd_box_bytes = 0;
        CP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->cd_box), bytes));
        ctx->cd_box_bytes = bytes;
    }
    ml.base = reinterpret_cast<unsigned long long *>(ctx->cd_box);
    const size_t total = ml.words * size_t(n_jobs);
    k_multi_box_init<<<int(std::min<size_t>((total + 255) / 256, 64)), 256, 0, ctx->stream>>>(ml.base, ml.words,
                                                                                             multi_box_sent_words(ml.G), n_jobs);
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

bool cp_cd_multi_wanted(int c) { return multi_wanted(c); }

int cp_cd_team_fit_launch(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats, int c, double l1_reg,
                          double l2_reg, uint32_t seed, int max_iter, double tol, int flags, double *w, void *dres,
                          bool allow_multi) {
    const int ex = team_exact_div();
    if (allow_multi && multi_wanted(c)) {
        MultiLaunch ml;
        CP_TRY(multi_prepare(ctx, c, 1, ml));
        const size_t mlds = multi_lds_bytes(c);
        CP_HIP(ctx, team_optin(k_cd_fit_multi, mlds));
        k_cd_fit_multi<<<multi_grid(1, ml.G, ml.same_xcd), 64 * XWAVES, mlds, ctx->stream>>>(
            Q, ldq, q, stats, c, l1_reg, l2_reg, seed, max_iter, tol, flags, ex, w, static_cast<DevResult *>(dres), ml);
        CP_LAUNCH_CHECK(ctx);
        return CP_OK;
    }
    const size_t lds = team_lds_bytes(c);
#define CP_CALL(R_, K_)                                                                                              \
    CP_HIP(ctx, team_optin(k_cd_fit_team<R_, K_>, lds));                                                             \
    k_cd_fit_team<R_, K_><<<1, 64 * (K_ + 2), lds, ctx->stream>>>(Q, ldq, q, stats, c, l1_reg, l2_reg, seed, max_iter, tol, flags, \
                                                                  ex, w, static_cast<DevResult *>(dres))
    CP_TEAM_SWITCH(c, CP_CALL);
#undef CP_CALL
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

int cp_cd_team_search_launch(cp_ctx *ctx, const void *batch, int n_jobs, int c, bool allow_multi) {
    // CP_CD_EXCLUSIVE (default 1): the workgroup asks for (almost) a whole CU's LDS, so that no other workgroup shares its
    // CU -- next to the products of other layers (or of its own layer's side stream) the chain and keeper waves otherwise
    // share their SIMDs' issue slots with MFMA-heavy waves: c = 512 search 10.5 -> 8.9 ms in a single-layer call, vgg16
    // job 33.4 -> 31.1 ms.  CP_CD_SPREAD=1: searches of concurrent launches rotate over the XCDs; CP_CD_PRIO=1: raised
    // wave priority (both measured without effect on the job: 33.6 / 33.7 ms; left as switches).
    static const int spread_on = env_int("CP_CD_SPREAD", 0), prio = env_int("CP_CD_PRIO", 0),
                     exclusive = env_int("CP_CD_EXCLUSIVE", 1);
    if (allow_multi && multi_wanted(c)) {
        MultiLaunch ml;
        CP_TRY(multi_prepare(ctx, c, n_jobs, ml));
        const size_t mlds = multi_lds_bytes(c);
        CP_HIP(ctx, team_optin(k_cd_search_multi, mlds));
        k_cd_search_multi<<<multi_grid(n_jobs, ml.G, ml.same_xcd), 64 * XWAVES, mlds, ctx->stream>>>(
            *static_cast<const CdSearchBatch *>(batch), team_exact_div(), ml);
        CP_LAUNCH_CHECK(ctx);
        return CP_OK;
    }
    static std::atomic<unsigned> next_xcd{0};
    size_t lds = team_lds_bytes(c);
    if (exclusive && !spread_on) lds = std::max(lds, size_t(150) * 1024);
    const int ex = team_exact_div();
    const int spread = spread_on ? int(next_xcd.fetch_add(unsigned(n_jobs)) & 7u) : -1;
    const int grid = spread_on ? 8 * n_jobs : n_jobs;
    const CdSearchBatch &b = *static_cast<const CdSearchBatch *>(batch);
#define CP_CALL(R_, K_)                                                  \
    CP_HIP(ctx, team_optin(k_cd_search_team<R_, K_>, lds));              \
    k_cd_search_team<R_, K_><<<grid, 64 * (K_ + 2), lds, ctx->stream>>>(b, ex, spread, prio)
    CP_TEAM_SWITCH(c, CP_CALL);
#undef CP_CALL
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

extern "C" int cp_debug_cd_team_cycles(cp_ctx *ctx, unsigned long long *out8) {
    if (!ctx || !out8) return CP_ERR_ARG;
    CP_HIP(ctx, cp_stream_wait(ctx));
#ifdef CP_CD_TEAM_TRACE
    {
        unsigned long long tr[16];
        CP_HIP(ctx, hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_team_trace), sizeof(tr)));
        fprintf(stderr, "team trace (cycles after the chain wave published block %d): keeper 0 saw it %lld, rows ready %lld, fma done %lld, "
                        "image out %lld; chain wave: prefetch of the next-but-one block's values %lld, repair finished %lld, next block "
                        "published %lld\n", TTRACE_BLOCK, (long long)(tr[1] - tr[0]), (long long)(tr[2] - tr[0]), (long long)(tr[3] - tr[0]),
                (long long)(tr[4] - tr[0]), (long long)(tr[5] - tr[0]), (long long)(tr[6] - tr[0]), (long long)(tr[7] - tr[0]));
    }
#endif
#ifdef CP_CD_MULTI_TRACE
    {
        unsigned long long tr[16];
        CP_HIP(ctx, hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_multi_trace), sizeof(tr)));
        fprintf(stderr, "multi trace (ns after the chain wave published block %d): forwarded %lld, received %lld, image %lld, posted %lld, "
                        "collected %lld (polling since %lld), old blocks applied %lld, published %lld; chain finished block %d at %lld\n", TRACE_BLOCK, (long long)(tr[1] - tr[0]) * 10,
                (long long)(tr[2] - tr[0]) * 10, (long long)(tr[3] - tr[0]) * 10, (long long)(tr[4] - tr[0]) * 10,
                (long long)(tr[5] - tr[0]) * 10, (long long)(tr[8] - tr[0]) * 10, (long long)(tr[9] - tr[0]) * 10,
                (long long)(tr[6] - tr[0]) * 10, TRACE_BLOCK + XLAG, (long long)(tr[7] - tr[0]) * 10);
    }
#endif
    CP_HIP(ctx, hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_team_debug), 8 * sizeof(unsigned long long)));
    return CP_OK;
}
