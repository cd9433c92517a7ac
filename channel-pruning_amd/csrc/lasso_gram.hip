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

// stats = { yc^T yc, mean(y), M, 0 }.  One 1024-thread workgroup, two passes (mean, then centred
// sum of squares); thread -> (sample s = tid / 32 + 32 it, columns j = tid % 32 + 32 jt): no
// integer division in the loops, 32 consecutive doubles per sample row and wave.
__global__ void __launch_bounds__(1024) k_y_stats(const double *__restrict__ Y, const int64_t *__restrict__ samples,
                                                  int S, int n, double *__restrict__ stats) {
    __shared__ double red[16];
    const int js = threadIdx.x & 31, ss = threadIdx.x >> 5;
    double s = 0;
    for (int si = ss; si < S; si += 32) {
        const double *row = Y + samples[si] * n;
        for (int j = js; j < n; j += 32) s += row[j];
    }
    const double M = double(S) * double(n);
    const double mean = block_sum(s, red) / M;
    double v = 0;
    for (int si = ss; si < S; si += 32) {
        const double *row = Y + samples[si] * n;
        for (int j = js; j < n; j += 32) {
            const double d = row[j] - mean;
            v += d * d;
        }
    }
    const double yty = block_sum(v, red);
    if (threadIdx.x == 0) {
        stats[0] = yty;
        stats[1] = mean;
        stats[2] = M;
        stats[3] = 0.0;
    }
}

// dst[r, col] = r < rows && col < cols ? src[(idx ? idx[r] : r) * cols + col] : 0   (row gather + f64 + zero pad)
template <typename T>
__global__ void __launch_bounds__(ZT) k_gather_rows(const T *__restrict__ src, const int64_t *__restrict__ idx,
                                                    int rows, int cols, int ld_dst, double *__restrict__ dst) {
    const int r = blockIdx.x;
    const bool live = r < rows;
    const size_t base = live ? size_t(idx ? idx[r] : r) * cols : 0;
    for (int col = threadIdx.x; col < ld_dst; col += ZT)
        dst[size_t(r) * ld_dst + col] = (live && col < cols) ? ld(src, base + col) : 0.0;
}

// Yst[j, s] = Y[samples[s], j] (n_pad x S_pad, zero padded): 32 x 32 tiles through LDS
__global__ void __launch_bounds__(ZT) k_gather_yt(const double *__restrict__ Y, const int64_t *__restrict__ samples,
                                                  int S, int n, int S_pad, double *__restrict__ Yst) {
    __shared__ double t[32][33];
    const int s0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) {
        const int s = s0 + y, j = j0 + tx;
        t[y][tx] = (s < S && j < n) ? Y[samples[s] * n + j] : 0.0;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) Yst[size_t(j0 + y) * S_pad + s0 + tx] = t[tx][y];
}

// One workgroup per channel i: column sums of Xs and Wf over its kk columns, the products with T,
// then zmean[i] and q[i].  Thread = row (sample s, then filter j), kk consecutive doubles each.
__global__ void __launch_bounds__(ZT) k_q_finish(const double *__restrict__ Xs, const double *__restrict__ Tm,
                                                 const double *__restrict__ Wf, int S_pad, int n_pad, int ldp, int kk,
                                                 const double *__restrict__ stats, double *__restrict__ zmean,
                                                 double *__restrict__ q) {
    __shared__ double red[ZT / 64];
    __shared__ double cx[64], cw[64];
    const int i = blockIdx.x;
    const size_t c0 = size_t(i) * kk;
    double dot = 0.0;
    for (int t = 0; t < kk; ++t) {
        double xs = 0.0, ws = 0.0;
        for (int s = threadIdx.x; s < S_pad; s += ZT) {
            const double x = Xs[size_t(s) * ldp + c0 + t];
            xs += x;
            dot = fma(x, Tm[size_t(s) * ldp + c0 + t], dot);
        }
        for (int j = threadIdx.x; j < n_pad; j += ZT) ws += Wf[size_t(j) * ldp + c0 + t];
        xs = block_sum(xs, red);
        ws = block_sum(ws, red);
        if (threadIdx.x == 0) {
            cx[t] = xs;
            cw[t] = ws;
        }
    }
    dot = block_sum(dot, red);
    if (threadIdx.x == 0) {
        const double M = stats[2];
        double acc = 0.0;
        for (int t = 0; t < kk; ++t) acc = fma(cx[t], cw[t], acc);
        const double zm = acc / M;
        zmean[i] = zm;
        q[i] = dot - M * zm * stats[1];
    }
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
    const size_t need = (2 * g_cnt + (2 * size_t(S_pad) + n_pad) * P_pad + size_t(n_pad) * S_pad + size_t(c) + 64) * 8 +
                        size_t(S) * 8 + ws + (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *GX = cp_arena_take_t<double>(ctx, g_cnt);
    double *GW = cp_arena_take_t<double>(ctx, g_cnt);
    double *Xs = cp_arena_take_t<double>(ctx, size_t(S_pad) * P_pad);
    double *Tm = cp_arena_take_t<double>(ctx, size_t(S_pad) * P_pad);
    double *Wf = cp_arena_take_t<double>(ctx, size_t(n_pad) * P_pad);
    double *Yst = cp_arena_take_t<double>(ctx, size_t(n_pad) * S_pad);
    double *zmean = cp_arena_take_t<double>(ctx, c);
    int64_t *dsamples = cp_arena_take_t<int64_t>(ctx, S);
    if (!GX || !GW || !Xs || !Tm || !Wf || !Yst || !zmean || !dsamples)
        return cp_set_error(ctx, CP_ERR_NOMEM, "lasso_gram: arena");

    cp_stage_begin(ctx);
    CP_HIP(ctx, hipMemcpyAsync(dsamples, samples, size_t(S) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (x_dtype == CP_F32)
        k_gather_rows<float><<<S_pad, ZT, 0, ctx->stream>>>(static_cast<const float *>(X), dsamples, S, P, P_pad, Xs);
    else
        k_gather_rows<double><<<S_pad, ZT, 0, ctx->stream>>>(static_cast<const double *>(X), dsamples, S, P, P_pad, Xs);
    CP_LAUNCH_CHECK(ctx);
    if (w_dtype == CP_F32)
        k_gather_rows<float><<<n_pad, ZT, 0, ctx->stream>>>(static_cast<const float *>(W2), nullptr, n, P, P_pad, Wf);
    else
        k_gather_rows<double><<<n_pad, ZT, 0, ctx->stream>>>(static_cast<const double *>(W2), nullptr, n, P, P_pad, Wf);
    CP_LAUNCH_CHECK(ctx);
    k_gather_yt<<<dim3(S_pad / 32, n_pad / 32), ZT, 0, ctx->stream>>>(Y, dsamples, S, n, S_pad, Yst);
    CP_LAUNCH_CHECK(ctx);
    k_y_stats<<<1, 1024, 0, ctx->stream>>>(Y, dsamples, S, n, stats);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "lasso_prep");
    ctx->gemm_tag = CP_GEMM_LASSO_GRAM;
    ctx->gemm_mark = "lasso_gram_gemm";
    CP_TRY(cp_gemm_tn_f64(ctx, P_pad, P_pad, S_pad, 1.0, Xs, P_pad, Xs, P_pad, 0.0, GX, P_pad, CP_TRI_UPPER));
    CP_TRY(cp_gemm_tn_f64(ctx, P_pad, P_pad, n_pad, 1.0, Wf, P_pad, Wf, P_pad, 0.0, GW, P_pad, CP_TRI_UPPER));
    CP_TRY(cp_gemm_tn_f64(ctx, S_pad, P_pad, n_pad, 1.0, Yst, S_pad, Wf, P_pad, 0.0, Tm, P_pad, CP_TRI_NONE));
    cp_stage_mark(ctx, "lasso_gram_gemms");
    k_q_finish<<<c, ZT, 0, ctx->stream>>>(Xs, Tm, Wf, S_pad, n_pad, P_pad, kk, stats, zmean, q);
    CP_LAUNCH_CHECK(ctx);
    k_hadamard_q<<<dim3(c, (c + ZT - 1) / ZT), ZT, 0, ctx->stream>>>(GX, GW, P_pad, c, kk, zmean, stats, Q, c);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "lasso_gram_reduce");
    return CP_OK;
}
