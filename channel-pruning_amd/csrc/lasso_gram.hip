// LASSO operands for one dictionary() call, replacing lib/decompose.py:428-437 and the
// centring + products sklearn's Lasso.fit performs on them (_base.py:108-205):
//   Z[(s,j), i] = sum_t X[samples[s], i, t] * W2[j, i, t]      M = S*n rows, c columns
//   zc = Z - colmean(Z),  yc = Y[samples].ravel() - mean
//   Q = zc^T zc (c x c),  q = zc^T yc,  yc^T yc
// Stages (all float64 arithmetic, deterministic reductions):
//   k_w_transpose  W2[n,c,kk] (f32/f64) -> Wt[t][j][c_pad] f64, so that a wave reading
//                  consecutive channels reads consecutive addresses          (HBM/L2 bound)
//   k_y_stats      mean and centred sum of squares of the sampled targets
//   k_z_means      colmean(Z)[i] = (sum_t (sum_s x[s,i,t]) (sum_j w[j,i,t])) / M  -- the mean is
//                  separable, so Z is written already centred and never re-read for it
//   k_build_z      zc rows (thread = channel, coalesced 8-B stores) + per-block partials of q
//   cp_gemm_tn_f64 Q = zc^T zc on the f64 MFMA pipe                          (MFMA bound)
#include "cp_common.h"

namespace {

constexpr int ZT = 256;  // threads per block in the element-wise stages

template <typename T>
__device__ __forceinline__ double ld(const T *p, size_t i) {
    return double(p[i]);
}

template <typename TW>
__global__ void __launch_bounds__(ZT) k_w_transpose(const TW *__restrict__ W2, int n, int c, int kk, int c_pad,
                                                    double *__restrict__ Wt) {
    // Wt[(t*n + j)*c_pad + i]; grid.x over (t*n + j), threads over i
    const int tj = blockIdx.x, t = tj / n, j = tj - t * n;
    for (int i = threadIdx.x; i < c_pad; i += ZT)
        Wt[size_t(tj) * c_pad + i] = i < c ? ld(W2, (size_t(j) * c + i) * kk + t) : 0.0;
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

template <typename TX>
__global__ void __launch_bounds__(ZT) k_z_means(const TX *__restrict__ X, const int64_t *__restrict__ samples, int S,
                                                int c, int kk, const double *__restrict__ Wt, int n, int c_pad,
                                                double *__restrict__ zmean) {
    __shared__ double red[ZT / 64];
    const int i = blockIdx.x;
    double acc = 0;
    for (int t = 0; t < kk; ++t) {
        double xs = 0, ws = 0;
        for (int s = threadIdx.x; s < S; s += ZT) xs += ld(X, (size_t(samples[s]) * c + i) * kk + t);
        for (int j = threadIdx.x; j < n; j += ZT) ws += Wt[(size_t(t) * n + j) * c_pad + i];
        xs = block_sum(xs, red);
        ws = block_sum(ws, red);
        acc += xs * ws;
    }
    if (threadIdx.x == 0) zmean[i] = acc / (double(S) * double(n));
}

// grid (S, JS): block handles sample s and target rows [j0, j1); thread = channel.
template <typename TX, int KK>
__global__ void __launch_bounds__(ZT)
k_build_z(const TX *__restrict__ X, const int64_t *__restrict__ samples, int c, const double *__restrict__ Wt, int n,
          int c_pad, const double *__restrict__ Y, const double *__restrict__ stats,
          const double *__restrict__ zmean, int jchunk, double *__restrict__ Zc, double *__restrict__ qpart) {
    const int s = blockIdx.x, js = blockIdx.y;
    const int j0 = js * jchunk, j1 = min(n, j0 + jchunk);
    const int64_t row = samples[s];
    const double ymean = stats[1];
    for (int i = threadIdx.x; i < c_pad; i += ZT) {
        double x[KK];
        double mu = 0.0;
        if (i < c) {
#pragma unroll
            for (int t = 0; t < KK; ++t) x[t] = ld(X, (size_t(row) * c + i) * KK + t);
            mu = zmean[i];
        } else {
#pragma unroll
            for (int t = 0; t < KK; ++t) x[t] = 0.0;
        }
        double qacc = 0.0;
        for (int j = j0; j < j1; ++j) {
            double z = 0.0;
#pragma unroll
            for (int t = 0; t < KK; ++t) z = fma(x[t], Wt[(size_t(t) * n + j) * c_pad + i], z);
            z -= mu;  // pad channels: 0 - 0
            Zc[(size_t(s) * n + j) * c_pad + i] = z;
            qacc = fma(z, Y[row * n + j] - ymean, qacc);
        }
        qpart[(size_t(s) * gridDim.y + js) * c_pad + i] = qacc;
    }
}

// q[i] = sum_b qpart[b][i]: 64 columns x 16 row-groups per workgroup, groups combined in order
__global__ void __launch_bounds__(1024) k_reduce_q(const double *__restrict__ qpart, int nparts, int c_pad, int c,
                                                   double *__restrict__ q) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int per = (nparts + 15) / 16;
    const int b0 = g * per, b1 = min(nparts, b0 + per);
    double s = 0;
    if (i < c_pad)
        for (int b = b0; b < b1; ++b) s += qpart[size_t(b) * c_pad + i];
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && i < c) {
        double t = 0;
        for (int k = 0; k < 16; ++k) t += red[k][col];
        q[i] = t;
    }
}

__global__ void __launch_bounds__(ZT) k_copy2d(const double *__restrict__ src, int lds_, double *__restrict__ dst,
                                               int ldd, int rows, int cols) {
    const int r = blockIdx.x;
    for (int cidx = threadIdx.x; cidx < cols; cidx += ZT) dst[size_t(r) * ldd + cidx] = src[size_t(r) * lds_ + cidx];
}

template <typename TX>
int launch_build_z(cp_ctx *ctx, int kk, dim3 grid, const TX *X, const int64_t *samples, int c, const double *Wt, int n,
                   int c_pad, const double *Y, const double *stats, const double *zmean, int jchunk, double *Zc,
                   double *qpart) {
#define CP_BZ(K)                                                                                             \
    k_build_z<TX, K><<<grid, ZT, 0, ctx->stream>>>(X, samples, c, Wt, n, c_pad, Y, stats, zmean, jchunk, Zc, \
                                                    qpart)
    switch (kk) {
        case 1: CP_BZ(1); break;
        case 4: CP_BZ(4); break;
        case 9: CP_BZ(9); break;
        case 25: CP_BZ(25); break;
        case 49: CP_BZ(49); break;
        default: return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "kernel size k*k=%d not in {1,4,9,25,49}", kk);
    }
#undef CP_BZ
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

}  // namespace

extern "C" int cp_lasso_gram(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                             int w_dtype, int n, const double *Y, const int64_t *samples, int S, double *Q, double *q,
                             double *stats) {
    if (!ctx || !X || !W2 || !Y || !samples || !Q || !q || !stats) return CP_ERR_ARG;
    if (N <= 0 || c <= 0 || kk <= 0 || n <= 0 || S <= 0) return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: bad sizes");
    if ((x_dtype != CP_F32 && x_dtype != CP_F64) || (w_dtype != CP_F32 && w_dtype != CP_F64))
        return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: bad dtype");
    for (int s = 0; s < S; ++s)
        if (samples[s] < 0 || samples[s] >= N) return cp_set_error(ctx, CP_ERR_ARG, "lasso_gram: sample out of range");
    CP_HIP(ctx, hipSetDevice(ctx->device));

    const int c_pad = int(cp_align_up(size_t(c), 128));
    const int64_t M = int64_t(S) * n;
    const int64_t M_pad = int64_t(cp_align_up(size_t(M), 16));
    if (M_pad > (int64_t(1) << 30)) return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "lasso_gram: S*n too large");
    int JS = 1;
    while (int64_t(S) * JS < 1024 && JS * 2 <= n) JS *= 2;
    const int jchunk = (n + JS - 1) / JS;
    JS = (n + jchunk - 1) / jchunk;

    const size_t z_bytes = size_t(M_pad) * c_pad * 8, wt_bytes = size_t(kk) * n * c_pad * 8,
                 qp_bytes = size_t(S) * JS * c_pad * 8, qpad_bytes = size_t(c_pad) * c_pad * 8;
    const size_t need = z_bytes + wt_bytes + qp_bytes + qpad_bytes + size_t(S) * 8 + size_t(c_pad) * 8 +
                        cp_gemm_tn_workspace(ctx, c_pad, c_pad, int(M_pad), CP_TRI_LOWER_MIRROR) + (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *Zc = cp_arena_take_t<double>(ctx, size_t(M_pad) * c_pad);
    double *Wt = cp_arena_take_t<double>(ctx, size_t(kk) * n * c_pad);
    double *qpart = cp_arena_take_t<double>(ctx, size_t(S) * JS * c_pad);
    double *Qpad = cp_arena_take_t<double>(ctx, size_t(c_pad) * c_pad);
    int64_t *dsamples = cp_arena_take_t<int64_t>(ctx, S);
    double *zmean = cp_arena_take_t<double>(ctx, c_pad);
    if (!Zc || !Wt || !qpart || !Qpad || !dsamples || !zmean) return cp_set_error(ctx, CP_ERR_NOMEM, "lasso_gram: arena");

    cp_stage_begin(ctx);
    CP_HIP(ctx, hipMemcpyAsync(dsamples, samples, size_t(S) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (M_pad > M)
        CP_HIP(ctx, hipMemsetAsync(Zc + size_t(M) * c_pad, 0, size_t(M_pad - M) * c_pad * 8, ctx->stream));
    if (w_dtype == CP_F32)
        k_w_transpose<float><<<kk * n, ZT, 0, ctx->stream>>>(static_cast<const float *>(W2), n, c, kk, c_pad, Wt);
    else
        k_w_transpose<double><<<kk * n, ZT, 0, ctx->stream>>>(static_cast<const double *>(W2), n, c, kk, c_pad, Wt);
    CP_LAUNCH_CHECK(ctx);
    k_y_stats<<<1, 1024, 0, ctx->stream>>>(Y, dsamples, S, n, stats);
    CP_LAUNCH_CHECK(ctx);
    if (x_dtype == CP_F32)
        k_z_means<float><<<c, ZT, 0, ctx->stream>>>(static_cast<const float *>(X), dsamples, S, c, kk, Wt, n, c_pad,
                                                     zmean);
    else
        k_z_means<double><<<c, ZT, 0, ctx->stream>>>(static_cast<const double *>(X), dsamples, S, c, kk, Wt, n, c_pad,
                                                      zmean);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "lasso_prep");
    dim3 grid(S, JS);
    if (x_dtype == CP_F32)
        CP_TRY(launch_build_z<float>(ctx, kk, grid, static_cast<const float *>(X), dsamples, c, Wt, n, c_pad, Y, stats,
                                     zmean, jchunk, Zc, qpart));
    else
        CP_TRY(launch_build_z<double>(ctx, kk, grid, static_cast<const double *>(X), dsamples, c, Wt, n, c_pad, Y,
                                      stats, zmean, jchunk, Zc, qpart));
    k_reduce_q<<<(c + 63) / 64, 1024, 0, ctx->stream>>>(qpart, S * JS, c_pad, c, q);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "lasso_build_z");
    double *Qdst = c_pad == c ? Q : Qpad;
    ctx->gemm_tag = CP_GEMM_LASSO_GRAM;
    ctx->gemm_mark = "lasso_gram_gemm";
    CP_TRY(cp_gemm_tn_f64(ctx, c_pad, c_pad, int(M_pad), 1.0, Zc, c_pad, Zc, c_pad, 0.0, Qdst, c_pad,
                          CP_TRI_LOWER_MIRROR));
    if (Qdst != Q) {
        k_copy2d<<<c, ZT, 0, ctx->stream>>>(Qpad, c_pad, Q, c, c, c);
        CP_LAUNCH_CHECK(ctx);
    }
    cp_stage_mark(ctx, "lasso_gram_reduce");
    return CP_OK;
}
