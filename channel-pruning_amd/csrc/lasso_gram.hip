// LASSO operands for one dictionary() call, replacing lib/decompose.py:428-437 and the
// centring + products sklearn's Lasso.fit performs on them (_base.py:108-205):
//   Z[(s,j), i] = sum_t X[samples[s], i, t] * W2[j, i, t]      M = S*n rows, c columns
//   zc = Z - colmean(Z),  yc = Y[samples].ravel() - mean
//   Q = zc^T zc (c x c),  q = zc^T yc,  yc^T yc
// Z (M x c, 131 MB at c = n = 256) is never formed.  With P = c*kk columns (i,t):
//   Xs[s, (i,t)] = X[samples[s], i, t]   (S x P)        Wf[j, (i,t)] = W2[j, i, t]   (n x P)
//   GX = Xs^T Xs,  GW = Wf^T Wf  (P x P, upper tiles)   T = Ys Wf  (S x P),  Ys = Y[samples]
//   Z^T Z [i,i'] = sum_{t,t'} GX[(i,t),(i',t')] * GW[(i,t),(i',t')]         (k_hadamard_q)
//   Z^T y [i]    = sum_t sum_s Xs[s,(i,t)] * T[s,(i,t)]                      (k_q_finish)
//   colmean(Z)[i] = sum_t (sum_s Xs[s,(i,t)]) (sum_j Wf[j,(i,t)]) / M
//   Q = Z^T Z - M zbar zbar^T,   q = Z^T y - M zbar ybar
// i.e. (S + n) P^2 / 2 MFMA flops instead of S n c^2 (1.6x fewer at k = 3, S n / (S + n) ~ 126x fewer
// at k = 1) and 2 x 21 MB of Gram tiles instead of the 131 MB of Z written and read back.
// Launches: k_lasso_prep (sampled rows of X, W2 and Y^T widened to f64 + target statistics), one paired
// MFMA launch for GX and GW, one for T, k_q_cols + k_q_finish, k_hadamard_q.
// All float64, deterministic reductions (fixed order, no atomics).
#include "cp_common.h"

#include <algorithm>

namespace {

constexpr int ZT = 256;  // threads per block in the element-wise stages

template <typename T>
__device__ __forceinline__ double ld(const T *p, size_t i) {
    return double(p[i]);
}

__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    double s = 0;
    for (int i = 0; i < int(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// All operand preparation in ONE launch (block ranges): S_pad blocks gather + widen the sampled rows of X,
// n_pad blocks widen W2, the next (S_pad/32)(n_pad/32) blocks build Yst = Y[samples]^T, and the last YB blocks
// accumulate the target statistics around a shift (the first sampled target): ypart[k] = {sum(y - shift),
// sum((y - shift)^2)} over the samples s = k mod YB, combined in fixed order by k_q_finish --
//   mean = shift + S1 / M,   yc^T yc = S2 - S1^2 / M
// (a one-pass centred sum of squares; the shift keeps the cancellation harmless).
constexpr int YB = 32;
template <typename TX, typename TW>
__global__ void __launch_bounds__(ZT) k_lasso_prep(const TX *__restrict__ X, const TW *__restrict__ W2,
                                                   const double *__restrict__ Y, const int64_t *__restrict__ samples,
                                                   int S, int n, int P, int S_pad, int n_pad, int P_pad,
                                                   double *__restrict__ Xs, double *__restrict__ Wf,
                                                   double *__restrict__ Yst, double *__restrict__ ypart) {
    __shared__ double t[32][33];
    int b = blockIdx.x;
    if (b < S_pad) {  // Xs[b, :]
        const bool live = b < S;
        const size_t base = live ? size_t(samples[b]) * P : 0;
        for (int col = threadIdx.x; col < P_pad; col += ZT)
            Xs[size_t(b) * P_pad + col] = (live && col < P) ? ld(X, base + col) : 0.0;
        return;
    }
    b -= S_pad;
    if (b < n_pad) {  // Wf[b, :]
        const bool live = b < n;
        for (int col = threadIdx.x; col < P_pad; col += ZT)
            Wf[size_t(b) * P_pad + col] = (live && col < P) ? ld(W2, size_t(b) * P + col) : 0.0;
        return;
    }
    b -= n_pad;
    const int tiles_s = S_pad / 32, tiles = tiles_s * (n_pad / 32);
    if (b < tiles) {  // Yst tile
        const int s0 = (b % tiles_s) * 32, j0 = (b / tiles_s) * 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int y = ty; y < 32; y += 8) {
            const int s = s0 + y, j = j0 + tx;
            t[y][tx] = (s < S && j < n) ? Y[samples[s] * n + j] : 0.0;
        }
        __syncthreads();
        for (int y = ty; y < 32; y += 8) Yst[size_t(j0 + y) * S_pad + s0 + tx] = t[tx][y];
        return;
    }
    b -= tiles;  // target statistics, block b of YB
    const double shift = Y[samples[0] * n];
    double s1 = 0.0, s2 = 0.0;
    for (int s = b; s < S; s += YB) {
        const double *row = Y + samples[s] * n;
        for (int j = threadIdx.x; j < n; j += ZT) {
            const double d = row[j] - shift;
            s1 += d;
            s2 = fma(d, d, s2);
        }
    }
    double *red = &t[0][0];
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        ypart[2 * b] = s1;
        ypart[2 * b + 1] = s2;
    }
}

// Column pass of q / zmean: workgroup = 64 consecutive columns (i,t) x 4 row lanes, so every row is read as one
// 512-byte run (the first form walked kk-double segments with a row stride between threads: 15x the operand bytes
// fetched at c = 512).  Per column: colX = sum_s Xs, colD = sum_s Xs T, colW = sum_j Wf; rows s = lane, lane + 4, ...
// with four loads in flight, the four row lanes combined in fixed order.
constexpr int QC = 64, QR = ZT / QC;
__global__ void __launch_bounds__(ZT) k_q_cols(const double *__restrict__ Xs, const double *__restrict__ Tm,
                                               const double *__restrict__ Wf, int S_pad, int n_pad, int ldp,
                                               double *__restrict__ colX, double *__restrict__ colD,
                                               double *__restrict__ colW) {
    __shared__ double red[3][QR][QC];
    const int cl = threadIdx.x & (QC - 1), rl = threadIdx.x / QC;
    const size_t col = size_t(blockIdx.x) * QC + cl;
    double xs = 0.0, dot = 0.0, ws = 0.0;
    int s = rl;
    for (; s + 3 * QR < S_pad; s += 4 * QR) {
        double x[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = Xs[size_t(s + u * QR) * ldp + col];
            t[u] = Tm[size_t(s + u * QR) * ldp + col];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xs += x[u];
            dot = fma(x[u], t[u], dot);
        }
    }
    for (; s < S_pad; s += QR) {
        const double x = Xs[size_t(s) * ldp + col];
        xs += x;
        dot = fma(x, Tm[size_t(s) * ldp + col], dot);
    }
    int j = rl;
    for (; j + 3 * QR < n_pad; j += 4 * QR) {
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = Wf[size_t(j + u * QR) * ldp + col];
#pragma unroll
        for (int u = 0; u < 4; ++u) ws += w[u];
    }
    for (; j < n_pad; j += QR) ws += Wf[size_t(j) * ldp + col];
    red[0][rl][cl] = xs;
    red[1][rl][cl] = dot;
    red[2][rl][cl] = ws;
    __syncthreads();
    if (rl == 0) {
        double a = 0.0, d = 0.0, w = 0.0;
#pragma unroll
        for (int r = 0; r < QR; ++r) {
            a += red[0][r][cl];
            d += red[1][r][cl];
            w += red[2][r][cl];
        }
        colX[col] = a;
        colD[col] = d;
        colW[col] = w;
    }
}

// Per channel i (one thread): zmean[i] = sum_t colX colW / M,  q[i] = sum_t colD - M zmean[i] ybar; target statistics
// from the YB partials (fixed order).
__global__ void __launch_bounds__(ZT) k_q_finish(const double *__restrict__ colX, const double *__restrict__ colD,
                                                 const double *__restrict__ colW, int c, int kk,
                                                 const double *__restrict__ ypart, const double *__restrict__ Y,
                                                 const int64_t *__restrict__ samples, int n, double M,
                                                 double *__restrict__ stats, double *__restrict__ zmean,
                                                 double *__restrict__ q) {
    const int i = blockIdx.x * ZT + threadIdx.x;
    double S1 = 0.0, S2 = 0.0;
    for (int k = 0; k < YB; ++k) {
        S1 += ypart[2 * k];
        S2 += ypart[2 * k + 1];
    }
    const double ymean = Y[samples[0] * n] + S1 / M;
    if (i == 0) {
        stats[0] = S2 - S1 * (S1 / M);
        stats[1] = ymean;
        stats[2] = M;
        stats[3] = 0.0;
    }
    if (i >= c) return;
    const size_t c0 = size_t(i) * kk;
    double acc = 0.0, dot = 0.0;
    for (int t = 0; t < kk; ++t) {
        acc = fma(colX[c0 + t], colW[c0 + t], acc);
        dot += colD[c0 + t];
    }
    const double zm = acc / M;
    zmean[i] = zm;
    q[i] = dot - M * zm * ymean;
}

// Q[i,i'] = sum_{t,t'} GX[(i,t),(i',t')] GW[(i,t),(i',t')] - M zbar_i zbar_i'  for i <= i', mirrored.
// Only the upper tiles of GX / GW exist: element (r, col) is read at (min, max).  Block = one i,
// threads over i' (a thread reads kk consecutive doubles per row: adjacent threads, adjacent segments).
__global__ void __launch_bounds__(ZT) k_hadamard_q(const double *__restrict__ GX, const double *__restrict__ GW,
                                                   int ldp, int c, int kk, const double *__restrict__ zmean,
                                                   const double *__restrict__ stats, double *__restrict__ Q, int ldq) {
    const int i = blockIdx.x, ip = blockIdx.y * ZT + threadIdx.x;
    if (ip >= c || ip < i) return;
    double acc = 0.0;
    for (int t = 0; t < kk; ++t) {
        const int r = i * kk + t;
        for (int u = 0; u < kk; ++u) {
            const int col = ip * kk + u;
            const size_t off = r <= col ? size_t(r) * ldp + col : size_t(col) * ldp + r;
            acc = fma(GX[off], GW[off], acc);
        }
    }
    const double v = fma(-stats[2] * zmean[i], zmean[ip], acc);
    Q[size_t(i) * ldq + ip] = v;
    Q[size_t(ip) * ldq + i] = v;
}

}  // namespace

namespace {
constexpr int SAMPLE_ARG_MAX = 400;     // min(400, N // 20) of decompose.py:425
struct SampleArg {
    int v[SAMPLE_ARG_MAX];
};
__global__ void __launch_bounds__(512) k_samples_from_arg(SampleArg sa, int S, int64_t *__restrict__ out) {
    const int s = threadIdx.x;
    if (s < S) out[s] = sa.v[s];
}
}  // namespace

extern "C" int cp_lasso_gram(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                             int w_dtype, int n, const double *Y, const int64_t *samples, int S, double *Q, double *q,
                             double *stats) {
    if (!ctx || !X || !W2 || !Y || !samples || !Q || !q || !stats) return CP_ERR_ARG;
    if (N <= 0 || c <= 0 || kk <= 0 || n <= 0 || S <= 0) return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: bad sizes");
    if (kk > 64) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "lasso_gram: kernel size k*k=%d > 64", kk);
    if ((x_dtype != CP_F32 && x_dtype != CP_F64) || (w_dtype != CP_F32 && w_dtype != CP_F64))
        return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: bad dtype");
    for (int s = 0; s < S; ++s)
        if (samples[s] < 0 || samples[s] >= N) return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: sample out of range");
    CP_HIP(ctx, hipSetDevice(ctx->device));

    const int64_t P64 = int64_t(c) * kk;
    if (P64 > 32768) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "lasso_gram: c*k*k = %lld too large", (long long)P64);
    const int P = int(P64), P_pad = int(cp_align_up(size_t(P), 128));
    const int S_pad = int(cp_align_up(size_t(S), 128)), n_pad = int(cp_align_up(size_t(n), 32));
    const size_t g_cnt = size_t(P_pad) * P_pad;
    size_t ws = std::max(cp_gemm_tn_workspace(ctx, P_pad, P_pad, S_pad, CP_TRI_UPPER),
                         cp_gemm_tn_workspace(ctx, P_pad, P_pad, n_pad, CP_TRI_UPPER));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, S_pad, P_pad, n_pad, CP_TRI_NONE));
    const size_t need = (2 * g_cnt + (2 * size_t(S_pad) + n_pad + 3) * P_pad + size_t(n_pad) * S_pad + size_t(c) + 64) * 8 +
                        size_t(S) * 8 + ws + (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *GX = cp_arena_take_t<double>(ctx, g_cnt);
    double *GW = cp_arena_take_t<double>(ctx, g_cnt);
    double *Xs = cp_arena_take_t<double>(ctx, size_t(S_pad) * P_pad);
    double *Tm = cp_arena_take_t<double>(ctx, size_t(S_pad) * P_pad);
    double *Wf = cp_arena_take_t<double>(ctx, size_t(n_pad) * P_pad);
    double *Yst = cp_arena_take_t<double>(ctx, size_t(n_pad) * S_pad);
    double *zmean = cp_arena_take_t<double>(ctx, c);
    double *colX = cp_arena_take_t<double>(ctx, 3 * size_t(P_pad)), *colD = colX ? colX + P_pad : nullptr,
           *colW = colX ? colX + 2 * size_t(P_pad) : nullptr;
    double *ypart = cp_arena_take_t<double>(ctx, 2 * YB);
    int64_t *dsamples = cp_arena_take_t<int64_t>(ctx, S);
    if (!GX || !GW || !Xs || !Tm || !Wf || !Yst || !zmean || !colX || !ypart || !dsamples)
        return cp_set_error(ctx, CP_ERR_NOMEM, "lasso_gram: arena");

    cp_stage_begin(ctx);
    // the sample subset (S <= 400 row indices, decompose.py:425) as a kernel argument: the first thing on a layer's stream is a
    // one-workgroup launch instead of a copy out of pageable host memory that the runtime stages through a buffer of its own
    if (S <= SAMPLE_ARG_MAX && N <= 0x7fffffffLL) {
        SampleArg sa;
        for (int s = 0; s < S; ++s) sa.v[s] = int(samples[s]);
        k_samples_from_arg<<<1, 512, 0, ctx->stream>>>(sa, S, dsamples);
        CP_LAUNCH_CHECK(ctx);
    } else {
        CP_HIP(ctx, hipMemcpyAsync(dsamples, samples, size_t(S) * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    {
        const unsigned nblk = unsigned(S_pad + n_pad + (S_pad / 32) * (n_pad / 32) + YB);
        const float *Xf = static_cast<const float *>(X), *Wf32 = static_cast<const float *>(W2);
        const double *Xd = static_cast<const double *>(X), *Wd64 = static_cast<const double *>(W2);
#define CP_PREP(TXp, TWp) \
    k_lasso_prep<<<nblk, ZT, 0, ctx->stream>>>(TXp, TWp, Y, dsamples, S, n, P, S_pad, n_pad, P_pad, Xs, Wf, Yst, ypart)
        if (x_dtype == CP_F32 && w_dtype == CP_F32)
            CP_PREP(Xf, Wf32);
        else if (x_dtype == CP_F32)
            CP_PREP(Xf, Wd64);
        else if (w_dtype == CP_F32)
            CP_PREP(Xd, Wf32);
        else
            CP_PREP(Xd, Wd64);
#undef CP_PREP
        CP_LAUNCH_CHECK(ctx);
    }
    cp_stage_mark(ctx, "lasso_prep");
    ctx->gemm_tag = CP_GEMM_LASSO_GRAM;
    ctx->gemm_mark = "lasso_gram_gemm";
    CP_TRY(cp_gemm_tn_f64_pair(ctx, P_pad, P_pad, 1.0, S_pad, Xs, Xs, GX, n_pad, Wf, Wf, GW, P_pad, P_pad, P_pad,
                               CP_TRI_UPPER));
    CP_TRY(cp_gemm_tn_f64(ctx, S_pad, P_pad, n_pad, 1.0, Yst, S_pad, Wf, P_pad, 0.0, Tm, P_pad, CP_TRI_NONE));
    cp_stage_mark(ctx, "lasso_gram_gemms");
    k_q_cols<<<P_pad / QC, ZT, 0, ctx->stream>>>(Xs, Tm, Wf, S_pad, n_pad, P_pad, colX, colD, colW);
    CP_LAUNCH_CHECK(ctx);
    k_q_finish<<<(c + ZT - 1) / ZT, ZT, 0, ctx->stream>>>(colX, colD, colW, c, kk, ypart, Y, dsamples, n,
                                                          double(S) * double(n), stats, zmean, q);
    CP_LAUNCH_CHECK(ctx);
    k_hadamard_q<<<dim3(c, (c + ZT - 1) / ZT), ZT, 0, ctx->stream>>>(GX, GW, P_pad, c, kk, zmean, stats, Q, c);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "lasso_gram_reduce");
    return CP_OK;
}
