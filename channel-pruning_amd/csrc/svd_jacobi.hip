// Thin SVD of a small dense matrix by one-sided (Hestenes) Jacobi on its ROWS, replacing
// scipy.linalg.svd(x, full_matrices=False, lapack_driver='gesvd') as called by VH_decompose
// (lib/decompose.py:45-47, 100).  For M (m x n, m <= n), left rotations J^T M make the rows mutually
// orthogonal:  R M = Sigma H  with R = J^T orthogonal, so
//   singular values  sigma_k = |row k of R M|,   V = R^T (column k = row k of R),   diag(sigma) H = R M
// -- the reference needs exactly V[:, :rank] and diag(sigma) H[:rank] (decompose.py:105-112), so the rows are
// never normalised.  The scalar form (one pair of rows per workgroup, dot products taken from the rows themselves) has
// one-sided Jacobi's high relative accuracy; the block form below decides its rotations from a 16 x 16 Gram that it
// updates in place, whose off-diagonals carry noise of about eps * |largest row| * |row| -- for strongly graded rows
// (condition beyond ~1e10) that can keep it rotating, so after BLOCK_SWEEP_BOUND sweeps without convergence the remaining
// sweeps run in the scalar form.
// BLOCK form: rows in blocks of 8, one launch per round of the round-robin ordering over the blocks, one workgroup per
// block pair.  The workgroup forms the 16 x 16 Gram of its 16 rows on MFMA, diagonalises it in LDS (two-sided cyclic
// Jacobi with the same rotation formula, 8 disjoint pairs at a time), and applies the accumulated 16 x 16 rotation to the
// rows of the work matrix and of R on MFMA.  A sweep is (m/8 - 1) launches instead of (m - 1): 31 instead of 255 at
// m = 256, where the scalar form was bound by its 255 dependent launches per sweep (ITQ: 350 sweeps per layer).  Sweeps
// repeat until no block pair holds an off-diagonal above the orthogonality tolerance.
#include "cp_common.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

constexpr int JT = 256;
constexpr int BLOCK_SWEEP_BOUND = 20;   // block-form sweeps before the scalar form takes over (typical: 6-15 sweeps in all)

__device__ __forceinline__ double jblock_sum(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// round r of the tournament on m (even) players: pair k -> rows (i, j)
__device__ __forceinline__ void round_pair(int m, int r, int k, int &i, int &j) {
    if (k == 0) {
        i = m - 1;
        j = r;
    } else {
        i = (r + k) % (m - 1);
        j = (r - k + m - 1) % (m - 1);
    }
}

// floor2: rows whose squared norm is below floor2[0] count as converged (0 / null: off).  For the leading invariant subspace
// of a symmetric matrix whose spectrum decays to rounding level, pairs involving such a row only shuffle noise: they kept
// the sweep count at 15 where 6 suffice (cp_itq_iterate).
__global__ void __launch_bounds__(JT) k_jacobi_round(double *__restrict__ Wk, int n, double *__restrict__ R, int m,
                                                     int round, double tol, int *__restrict__ rotated,
                                                     const double *__restrict__ floor2) {
    __shared__ double red[4];
    __shared__ double cs[2];
    int i, j;
    round_pair(m, round, blockIdx.x, i, j);
    double *a = Wk + size_t(i) * n, *b = Wk + size_t(j) * n;
    double saa = 0, sbb = 0, sab = 0;
    for (int col = threadIdx.x; col < n; col += JT) {
        const double x = a[col], y = b[col];
        saa = fma(x, x, saa);
        sbb = fma(y, y, sbb);
        sab = fma(x, y, sab);
    }
    const double alpha = jblock_sum(saa, red), beta = jblock_sum(sbb, red), gamma = jblock_sum(sab, red);
    if (threadIdx.x == 0) {
        double c = 1.0, s = 0.0;
        const double fl = floor2 ? floor2[0] : 0.0;
        if (fabs(gamma) > tol * sqrt(alpha * beta) && gamma != 0.0 && alpha > fl && beta > fl) {
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            c = 1.0 / sqrt(1.0 + t * t);
            s = c * t;
            atomicAdd(rotated, 1);
        }
        cs[0] = c;
        cs[1] = s;
    }
    __syncthreads();
    const double c = cs[0], s = cs[1];
    if (s == 0.0) return;
    for (int col = threadIdx.x; col < n; col += JT) {
        const double x = a[col], y = b[col];
        a[col] = c * x - s * y;
        b[col] = s * x + c * y;
    }
    double *ra = R + size_t(i) * m, *rb = R + size_t(j) * m;
    for (int col = threadIdx.x; col < m; col += JT) {
        const double x = ra[col], y = rb[col];
        ra[col] = c * x - s * y;
        rb[col] = s * x + c * y;
    }
}

// out[0] = rel^2 * max_i sig[i]^2 (sig = row norms)
__global__ void __launch_bounds__(JT) k_norm_floor(const double *__restrict__ sig, int m, double rel, double *__restrict__ out) {
    __shared__ double red[4];
    double v = 0;
    for (int i = threadIdx.x; i < m; i += JT) v = fmax(v, sig[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        out[0] = rel * rel * mx * mx;
    }
}

typedef double v4f64j __attribute__((ext_vector_type(4)));
// hardware estimate + two Newton steps: full double accuracy at a fraction of the software sqrt / divide sequences
__device__ __forceinline__ double jrsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}
__device__ __forceinline__ double jrcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * fma(-x, y, 2.0);
    y = y * fma(-x, y, 2.0);
    return y;
}
constexpr int JB = 8;   // rows per block

// One block pair of one round: the pair `pair` of round `round` of the tournament over the me / JB row blocks.
__device__ __forceinline__ void jacobi_block_pair(double *__restrict__ Wk, int n, double *__restrict__ R, int me, int round,
                                                  int pair, double tol, int *__restrict__ rotated,
                                                  const double *__restrict__ floor2, int inner_sweeps) {
    __shared__ double A[16][17], Q[16][17], part[4][16][17];
    __shared__ int any_rot;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fk = lane >> 4, fi = lane & 15;
    int bi, bj;
    round_pair(me / JB, round, pair, bi, bj);
    auto row_of = [&](int mloc) { return mloc < JB ? bi * JB + mloc : bj * JB + (mloc - JB); };
    // ---- Gram of the 16 rows: wave w takes the column groups 4 w, 4 w + 16, ...  Latency-bound (a few loads per lane
    // from L2 / HBM per round): two stages of eight column groups are kept in flight, the next stage is requested before
    // the products of the current one (n = 256: everything is requested up front) ----
    {
        v4f64j acc = {0., 0., 0., 0.};
        const double *rowp = Wk + size_t(row_of(fi)) * n;
        constexpr int GU = 8;                               // column groups per stage and wave: 128 columns per stage
        double v[2][GU];
        auto fetch = [&](int stage, double *dst) {
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int c = stage * (16 * GU) + (u >> 2) * 64 + wave * 4 + 16 * (u & 3) + fk;
                dst[u] = c < n ? rowp[c] : 0.0;
            }
        };
        const int stages = (n + 16 * GU - 1) / (16 * GU);
        fetch(0, v[0]);
        for (int st = 0; st < stages; st += 2) {
            if (st + 1 < stages) fetch(st + 1, v[1]);
#pragma unroll
            for (int u = 0; u < GU; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[0][u], v[0][u], acc, 0, 0, 0);
            if (st + 2 < stages) fetch(st + 2, v[0]);
            if (st + 1 < stages) {
#pragma unroll
                for (int u = 0; u < GU; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[1][u], v[1][u], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][fk + 4 * r][fi] = acc[r];
    }
    if (tid == 0) any_rot = 0;
    __syncthreads();
    {
        const int mloc = tid >> 4, nloc = tid & 15;
        A[mloc][nloc] = part[0][mloc][nloc] + part[1][mloc][nloc] + part[2][mloc][nloc] + part[3][mloc][nloc];
        Q[mloc][nloc] = mloc == nloc ? 1.0 : 0.0;
    }
    __syncthreads();
    const double fl = floor2 ? floor2[0] : 0.0;
    // ---- diagonalise A: Q A Q^T, Q accumulated (rows of the work matrix become Q W).  ONE wave, no workgroup barriers
    // (a wave's LDS operations execute in order; 270 __syncthreads made this phase 40 us): per inner round lanes 0..7
    // compute the 8 disjoint rotations, then every lane updates two (pair, column) entries of the rows of A, of the rows
    // of Q and of the columns of A. ----
    if (wave == 0) {
        bool any = false;
        const int k = lane >> 3;
        for (int isw = 0; isw < inner_sweeps; ++isw) {
            bool rot = false;
            for (int ir = 0; ir < 15; ++ir) {
                // every lane of a pair's group computes that pair's rotation itself (no hand-off through LDS): reads of this
                // instruction precede the writes of the later ones across the whole wave
                int p, q;
                round_pair(16, ir, k, p, q);
                const double app = A[p][p], aqq = A[q][q], apq = A[p][q];
                double c = 1.0, sn = 0.0;
                if (apq * apq > tol * tol * app * aqq && app > fl && aqq > fl) {
                    const double zeta = (aqq - app) * 0.5 * jrcp(apq);
                    const double z2 = fma(zeta, zeta, 1.0);
                    const double t = copysign(1.0, zeta) * jrcp(fabs(zeta) + z2 * jrsqrt(z2));
                    c = jrsqrt(fma(t, t, 1.0));
                    sn = c * t;
                }
                rot = rot || __builtin_amdgcn_ballot_w64(sn != 0.0) != 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {   // rows p, q of A and of Q
                    const int j = (lane & 7) * 2 + e;
                    const double x = A[p][j], y = A[q][j];
                    A[p][j] = c * x - sn * y;
                    A[q][j] = sn * x + c * y;
                    const double u = Q[p][j], v = Q[q][j];
                    Q[p][j] = c * u - sn * v;
                    Q[q][j] = sn * u + c * v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int e = 0; e < 2; ++e) {   // columns p, q of A
                    const int i = (lane & 7) * 2 + e;
                    const double x = A[i][p], y = A[i][q];
                    A[i][p] = c * x - sn * y;
                    A[i][q] = sn * x + c * y;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (!rot) break;
            any = true;
        }
        if (lane == 0) any_rot = any ? 1 : 0;
    }
    __syncthreads();
    if (!any_rot) return;
    if (tid == 0) atomicAdd(rotated, 1);
    // ---- rows <- Q rows, for the work matrix (n columns) and for R (me columns) ----
    double qa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = Q[fi][4 * ks + fk];
    // 16-column tiles of [work matrix | R] dealt round-robin to the waves, TF tiles' loads (both matrices alike) in flight
    // before the first product: at n = me = 256 that is every load of the launch at once
    constexpr int TF = 8;
    const int tiles_w = (n + 15) / 16, tiles_all = tiles_w + (me + 15) / 16;
    for (int t0 = wave; t0 < tiles_all; t0 += 4 * TF) {
        double bv[TF][4];
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            const int tile = t0 + 4 * t;
            const bool in_w = tile < tiles_w;
            const double *Mat = in_w ? Wk : R;
            const int ncol = in_w ? n : me, cc = (in_w ? tile : tile - tiles_w) * 16 + fi;
            const bool live = tile < tiles_all && cc < ncol;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bv[t][ks] = live ? Mat[size_t(row_of(4 * ks + fk)) * ncol + cc] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            const int tile = t0 + 4 * t;
            const bool in_w = tile < tiles_w;
            double *Mat = in_w ? Wk : R;
            const int ncol = in_w ? n : me, cc = (in_w ? tile : tile - tiles_w) * 16 + fi;
            v4f64j acc = {0., 0., 0., 0.};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[ks], bv[t][ks], acc, 0, 0, 0);
            if (tile < tiles_all && cc < ncol) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Mat[size_t(row_of(fk + 4 * r)) * ncol + cc] = acc[r];
            }
        }
    }
}

// (JT, 2): a register budget of two waves per SIMD keeps the MFMA accumulators in VGPRs (the AccVGPR form of the f64 MFMA
// issues 1.65 x slower: profiles/r04_gemm_probe.md)
__global__ void __launch_bounds__(JT, 2) k_jacobi_block_round(double *__restrict__ Wk, int n, double *__restrict__ R, int me,
                                                           int round, double tol, int *__restrict__ rotated,
                                                           const double *__restrict__ floor2, int inner_sweeps) {
    jacobi_block_pair(Wk, n, R, me, round, blockIdx.x, tol, rotated, floor2, inner_sweeps);
}

// ALL sweeps in ONE launch (CP_JACOBI_PERSISTENT=1): the me / (2 JB) workgroups of a round stay resident and meet at a
// device-wide barrier after every round (rows written in a round are read by other workgroups in the next: agent-scope
// release before arriving, acquire after leaving); the workgroups read the sweep's rotation count themselves instead of
// the host.  59 427 launches become 206 in the ITQ profile, the time does not change (see cp_svd_rows_core).
// ctl (ints, zeroed by the host): [0] barrier arrivals (monotone), [1] sweeps run before the first one without a rotation
// (max_sweeps: none), [2] != 0: a barrier timed out (never observed; the host then reports an error instead of hanging),
// [16 + s] rotations in sweep s.
__global__ void __launch_bounds__(JT, 2) k_jacobi_block_sweeps(double *__restrict__ Wk, int n, double *__restrict__ R, int me,
                                                            double tol, int *__restrict__ ctl,
                                                            const double *__restrict__ floor2, int inner_sweeps,
                                                            int max_sweeps) {
    __shared__ int verdict;
    const int nb = me / JB, nwg = int(gridDim.x);
    int arrivals = 0;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        for (int round = 0; round < nb - 1; ++round) {
            jacobi_block_pair(Wk, n, R, me, round, blockIdx.x, tol, ctl + 16 + sweep, floor2, inner_sweeps);
            arrivals += nwg;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int ok = 0;
                for (int spin = 0; spin < (1 << 24); ++spin) {
                    if (__hip_atomic_load(ctl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= arrivals) {
                        ok = 1;
                        break;
                    }
                    if (__hip_atomic_load(ctl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) __hip_atomic_store(ctl + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                verdict = !ok ? -1 : (round == nb - 2 ? __hip_atomic_load(ctl + 16 + sweep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (verdict < 0) return;
        }
        if (verdict == 0) {          // a whole sweep without a rotation: converged
            if (blockIdx.x == 0 && threadIdx.x == 0) ctl[1] = sweep;
            return;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl[1] = max_sweeps;
}

__global__ void __launch_bounds__(JT) k_identity(double *__restrict__ R, int m) {
    const int r = blockIdx.x;
    for (int col = threadIdx.x; col < m; col += JT) R[size_t(r) * m + col] = col == r ? 1.0 : 0.0;
}

// dst[k, :cols] = src[order[k], :cols]
__global__ void __launch_bounds__(JT) k_take_rows(const double *__restrict__ src, int ld_src, const int *__restrict__ order,
                                                  int cols, double *__restrict__ dst, int ld_dst) {
    const double *a = src + size_t(order[blockIdx.x]) * ld_src;
    for (int col = threadIdx.x; col < cols; col += JT) dst[size_t(blockIdx.x) * ld_dst + col] = a[col];
}

__global__ void __launch_bounds__(JT) k_row_norms(const double *__restrict__ Wk, int n, double *__restrict__ sig) {
    __shared__ double red[4];
    const double *a = Wk + size_t(blockIdx.x) * n;
    double s = 0;
    for (int col = threadIdx.x; col < n; col += JT) s = fma(a[col], a[col], s);
    s = jblock_sum(s, red);
    if (threadIdx.x == 0) sig[blockIdx.x] = sqrt(s);
}

}  // namespace

// M DEVICE [m, n] with row stride ldm (m <= n, untouched) -> sigma [r], Vt [r, m] (row stride ldv), SH [r, n] (row
// stride ldsh); scratch from the caller (the arena is NOT re-reserved here).
int cp_svd_rows_impl(cp_ctx *ctx, const double *M, int ldm, int m, int n, int r, double *sigma, double *Vt, int ldv,
                     double *SH, int ldsh, SvdScratch &sc, int *sweeps_out) {
    return cp_svd_rows_core(ctx, M, ldm, m, n, r, sigma, Vt, ldv, SH, ldsh, sc, sweeps_out, false, 0.0, 0.0);
}

// preinit: sc.Wk [me, n] and sc.R [me, me] already hold a consistent pair (Wk = R M for an orthogonal R) -- a WARM
// START, e.g. the rotation the previous, nearby matrix ended with: Jacobi converges quadratically from there
// (2-3 sweeps instead of 8-10).  M is then not read.
int cp_svd_rows_core(cp_ctx *ctx, const double *M, int ldm, int m, int n, int r, double *sigma, double *Vt, int ldv,
                     double *SH, int ldsh, SvdScratch &sc, int *sweeps_out, bool preinit, double rel_floor, double tol_in) {
    const int me = cp_svd_me(m);  // whole 8-row blocks, an even number of them: all-zero rows play along
    double *Wk = sc.Wk, *R = sc.R, *sig = sc.sig;
    int *rotated = sc.rotated;
    CP_TRY(cp_pinned_reserve(ctx, 4096));
    if (!preinit) {
        CP_HIP(ctx, hipMemsetAsync(Wk, 0, size_t(me) * n * 8, ctx->stream));
        CP_HIP(ctx, hipMemcpy2DAsync(Wk, size_t(n) * 8, M, size_t(ldm) * 8, size_t(n) * 8, size_t(m), hipMemcpyDeviceToDevice,
                                     ctx->stream));
        k_identity<<<me, JT, 0, ctx->stream>>>(R, me);
        CP_LAUNCH_CHECK(ctx);
    }
    const double *floor2 = nullptr;   // rel_floor > 0: rows below rel_floor * (largest row norm) are left alone
    if (rel_floor > 0.0) {
        k_row_norms<<<me, JT, 0, ctx->stream>>>(Wk, n, sig);
        CP_LAUNCH_CHECK(ctx);
        double *fl = reinterpret_cast<double *>(rotated + 4);   // 16-int scratch: [0] counter, [4..5] the floor
        k_norm_floor<<<1, JT, 0, ctx->stream>>>(sig, me, rel_floor, fl);
        CP_LAUNCH_CHECK(ctx);
        floor2 = fl;
    }
    int sweeps = 0;
    // |a.b| <= tol |a||b|: rows orthogonal to rounding (tol_in > 0: the caller needs less, e.g. an invariant subspace only)
    const double tol = tol_in > 0.0 ? tol_in : std::max(1e-14, 4e-16 * std::sqrt(double(n)));
    constexpr int MAX_SWEEPS = 60;
    static const bool scalar = getenv("CP_JACOBI_SCALAR") && getenv("CP_JACOBI_SCALAR")[0] == '1';
    // opt-in: measured equal (ITQ at conv3 size 194 ms against 189 ms with per-round launches; a round is 13 us of
    // dependent work -- Gram, the single-wave 16 x 16 diagonalisation, the update -- and the device barrier costs what
    // the launch gap did), so the form without inter-workgroup waiting stays the default
    static const bool per_round = !(getenv("CP_JACOBI_PERSISTENT") && getenv("CP_JACOBI_PERSISTENT")[0] == '1');
    static const int inner = getenv("CP_JACOBI_INNER") ? atoi(getenv("CP_JACOBI_INNER")) : 1;
    const int nb = me / JB;
    // every workgroup of the one-launch form has to be resident at once (they wait for each other): 256 threads and ~10 KB
    // of LDS each, so a few per CU fit; beyond that (p > ~8000 rows) the per-round launches take over
    const bool one_launch = !scalar && !per_round && nb / 2 <= 2 * ctx->cu_count;
    int *ctl = rotated + 16;
    if (one_launch) {
        CP_HIP(ctx, hipMemsetAsync(ctl, 0, (16 + MAX_SWEEPS + 4) * sizeof(int), ctx->stream));
        k_jacobi_block_sweeps<<<nb / 2, JT, 0, ctx->stream>>>(Wk, n, R, me, tol, ctl, floor2, inner, MAX_SWEEPS);
        CP_LAUNCH_CHECK(ctx);
        CP_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctl, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        CP_HIP(ctx, cp_stream_wait(ctx));
        int head[4];
        memcpy(head, ctx->pinned, sizeof(head));
        if (head[2] != 0) return cp_set_error(ctx, CP_ERR_NUMERIC, "svd_rows: device barrier timed out");
        sweeps = head[1];
    } else {
        bool use_scalar = scalar;
        for (; sweeps < MAX_SWEEPS; ++sweeps) {
            CP_HIP(ctx, hipMemsetAsync(rotated, 0, sizeof(int), ctx->stream));
            if (!use_scalar && sweeps >= BLOCK_SWEEP_BOUND) use_scalar = true;   // graded rows: see the file header
            if (use_scalar) {
                for (int round = 0; round < me - 1; ++round) {
                    k_jacobi_round<<<me / 2, JT, 0, ctx->stream>>>(Wk, n, R, me, round, tol, rotated, floor2);
                    CP_LAUNCH_CHECK(ctx);
                }
            } else {
                for (int round = 0; round < nb - 1; ++round) {
                    k_jacobi_block_round<<<nb / 2, JT, 0, ctx->stream>>>(Wk, n, R, me, round, tol, rotated, floor2, inner);
                    CP_LAUNCH_CHECK(ctx);
                }
            }
            CP_HIP(ctx, hipMemcpyAsync(ctx->pinned, rotated, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            CP_HIP(ctx, cp_stream_wait(ctx));
            int nrot;
            memcpy(&nrot, ctx->pinned, sizeof(int));
            if (nrot == 0) break;
        }
    }
    if (sweeps_out) *sweeps_out = sweeps;
    if (sweeps >= MAX_SWEEPS) return cp_set_error(ctx, CP_ERR_NUMERIC, "svd_rows: no convergence in 60 sweeps");
    // singular values = row norms; order descending on the host (m numbers), gather the leading r rows
    k_row_norms<<<me, JT, 0, ctx->stream>>>(Wk, n, sig);
    CP_LAUNCH_CHECK(ctx);
    std::vector<double> hs(me);
    CP_HIP(ctx, hipMemcpyAsync(hs.data(), sig, size_t(me) * 8, hipMemcpyDeviceToHost, ctx->stream));
    CP_HIP(ctx, cp_stream_wait(ctx));
    std::vector<int> order(me);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return hs[x] > hs[y]; });
    std::vector<double> hsorted(r);
    for (int k = 0; k < r; ++k) hsorted[k] = hs[order[k]];
    sc.sigma_r = hsorted[r - 1];
    sc.sigma_next = r < me ? hs[order[r]] : 0.0;
    sc.sigma_sum = std::accumulate(hs.begin(), hs.end(), 0.0);
    CP_HIP(ctx, hipMemcpyAsync(sc.order, order.data(), size_t(r) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    k_take_rows<<<r, JT, 0, ctx->stream>>>(R, me, sc.order, m, Vt, ldv);
    CP_LAUNCH_CHECK(ctx);
    k_take_rows<<<r, JT, 0, ctx->stream>>>(Wk, n, sc.order, n, SH, ldsh);
    CP_LAUNCH_CHECK(ctx);
    CP_HIP(ctx, hipMemcpyAsync(sigma, hsorted.data(), size_t(r) * 8, hipMemcpyHostToDevice, ctx->stream));
    CP_HIP(ctx, cp_stream_wait(ctx));
    return CP_OK;
}

// M DEVICE [m, n] row-major f64 (m <= n, untouched).  Outputs DEVICE: sigma [r] descending, Vt [r, m] (row k =
// k-th left singular vector, i.e. V[:, k]), SH [r, n] = diag(sigma) H[:r] (k-th right singular vector scaled by
// sigma_k).  r <= m.  sweeps_out HOST (may be NULL).
extern "C" int cp_svd_rows(cp_ctx *ctx, const double *M, int m, int n, int r, double *sigma, double *Vt, double *SH,
                           int *sweeps_out) {
    if (!ctx || !M || !sigma || !Vt || !SH || m <= 0 || n < m || r <= 0 || r > m)
        return ctx ? cp_set_error(ctx, CP_ERR_ARG, "svd_rows: bad arguments (needs 0 < r <= m <= n)") : CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_TRY(cp_arena_reserve(ctx, SvdScratch::bytes(m, n) + (1 << 16)));
    SvdScratch sc;
    if (!sc.take(ctx, m, n)) return cp_set_error(ctx, CP_ERR_NOMEM, "svd_rows: arena");
    return cp_svd_rows_impl(ctx, M, n, m, n, r, sigma, Vt, m, SH, n, sc, sweeps_out);
}

extern "C" int cp_svd_rows_lowrank(cp_ctx *ctx, const double *M, int m, int n, int r, double *sigma, double *Vt, double *SH,
                                   int *sweeps_out) {
    if (!ctx || !M || !sigma || !Vt || !SH || m <= 0 || n < m || r <= 0 || r > m)
        return ctx ? cp_set_error(ctx, CP_ERR_ARG, "svd_rows: bad arguments (needs 0 < r <= m <= n)") : CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_TRY(cp_arena_reserve(ctx, SvdScratch::bytes(m, n) + (1 << 16)));
    SvdScratch sc;
    if (!sc.take(ctx, m, n)) return cp_set_error(ctx, CP_ERR_NOMEM, "svd_rows: arena");
    return cp_svd_rows_core(ctx, M, n, m, n, r, sigma, Vt, m, SH, n, sc, sweeps_out, false, 1e-13, 0.0);
}

// ---- VH_decompose helpers (lib/decompose.py:85-146) ------------------------------------------------------
namespace {

// Xv[s, r * w + wi] = sum_{ci, hi} X[s, ci, hi, wi] * Vt[r, ci * h + hi]      (np.tensordot(X, V, [[1,2],[1,2]])
// followed by the transpose / reshape of decompose.py:131-136).  One workgroup per sample, x[s] staged in LDS.
template <typename TX>
__global__ void __launch_bounds__(JT) k_vh_project(const TX *__restrict__ X, int ch, int w, const double *__restrict__ Vt,
                                                   int ldv, int rank, double *__restrict__ Xv) {
    extern __shared__ double xs[];  // [ch][w]
    const size_t s = blockIdx.x;
    for (int e = threadIdx.x; e < ch * w; e += JT) xs[e] = double(X[s * size_t(ch) * w + e]);
    __syncthreads();
    for (int o = threadIdx.x; o < rank * w; o += JT) {
        const int r = o / w, wi = o - r * w;
        const double *v = Vt + size_t(r) * ldv;
        double acc = 0.0;
        for (int k = 0; k < ch; ++k) acc = fma(xs[k * w + wi], v[k], acc);
        Xv[s * size_t(rank) * w + o] = acc;
    }
}

__global__ void __launch_bounds__(JT) k_pad_copy(const double *__restrict__ src, int rows, int cols, int ld_src,
                                                 double *__restrict__ dst, int ld_dst) {
    const int r = blockIdx.x;
    for (int col = threadIdx.x; col < ld_dst; col += JT)
        dst[size_t(r) * ld_dst + col] = (r < rows && col < cols) ? src[size_t(r) * ld_src + col] : 0.0;
}

}  // namespace

// Xv DEVICE [N, rank * w] f64 from X DEVICE [N, c, h, w] (x_dtype) and Vt DEVICE [rank, c*h] (rows = leading left
// singular vectors, as cp_svd_rows returns them): decompose.py:131-136.
extern "C" int cp_vh_project(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int h, int w, const double *Vt,
                             int rank, double *Xv) {
    if (!ctx || !X || !Vt || !Xv || N <= 0 || c <= 0 || h <= 0 || w <= 0 || rank <= 0) return CP_ERR_ARG;
    if (x_dtype != CP_F32 && x_dtype != CP_F64) return cp_set_error(ctx, CP_ERR_ARG, "vh_project: bad dtype");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int ch = c * h;
    const size_t lds = size_t(ch) * w * sizeof(double);
    if (lds > 60 * 1024) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "vh_project: c*h*w = %d too large", ch * w);
    if (x_dtype == CP_F32)
        k_vh_project<float><<<unsigned(N), JT, lds, ctx->stream>>>(static_cast<const float *>(X), ch, w, Vt, ch, rank, Xv);
    else
        k_vh_project<double><<<unsigned(N), JT, lds, ctx->stream>>>(static_cast<const double *>(X), ch, w, Vt, ch, rank, Xv);
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

// C[m, n] = A^T B with A DEVICE [k, m], B DEVICE [k, n] (row-major, any sizes): the f64 MFMA GEMM on zero-padded
// copies.  Used for the small reconstructions V diag(sigma) H (decompose.py:114, 139).
extern "C" int cp_matmul_tn(cp_ctx *ctx, const double *A, const double *B, int m, int n, int k, double *C) {
    if (!ctx || !A || !B || !C || m <= 0 || n <= 0 || k <= 0) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int mp = int(cp_align_up(size_t(m), 128)), np_ = int(cp_align_up(size_t(n), 128)), kp = int(cp_align_up(size_t(k), 16));
    const size_t ws = cp_gemm_tn_workspace(ctx, mp, np_, kp, CP_TRI_NONE);
    CP_TRY(cp_arena_reserve(ctx, (size_t(kp) * mp + size_t(kp) * np_ + size_t(mp) * np_) * 8 + ws + (1 << 16)));
    double *Ap = cp_arena_take_t<double>(ctx, size_t(kp) * mp), *Bp = cp_arena_take_t<double>(ctx, size_t(kp) * np_);
    double *Cp = cp_arena_take_t<double>(ctx, size_t(mp) * np_);
    if (!Ap || !Bp || !Cp) return cp_set_error(ctx, CP_ERR_NOMEM, "matmul_tn: arena");
    k_pad_copy<<<kp, JT, 0, ctx->stream>>>(A, k, m, m, Ap, mp);
    CP_LAUNCH_CHECK(ctx);
    k_pad_copy<<<kp, JT, 0, ctx->stream>>>(B, k, n, n, Bp, np_);
    CP_LAUNCH_CHECK(ctx);
    CP_TRY(cp_gemm_tn_f64(ctx, mp, np_, kp, 1.0, Ap, mp, Bp, np_, 0.0, Cp, np_, CP_TRI_NONE));
    CP_HIP(ctx, hipMemcpy2DAsync(C, size_t(n) * 8, Cp, size_t(np_) * 8, size_t(n) * 8, size_t(m), hipMemcpyDeviceToDevice,
                                 ctx->stream));
    return CP_OK;
}
