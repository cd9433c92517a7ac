// Operand assembly kernels (HBM-bound gathers, no arithmetic to speak of):
//   cp_patch_gather  sampled-point im2col of Net.extract_XY (lib/net.py:629-657) fused with the
//                    [N*k*k, C] -> [N, C, k, k] re-layout of lib/net.py:1702 and the VGG ReLU of
//                    lib/net.py:1720
//   cp_assemble_y    Y = feats - bias (+ resY)            lib/net.py:1707, 1716-1722
#include "cp_common.h"

namespace {

// One workgroup per (batch, point, image) output row; threads sweep the C*k*k patch elements, which are
// contiguous in the output (coalesced stores); reads walk k-wide runs of the feature map: every run of a
// channel row costs one 64-byte fabric request whatever k is, so the kernel is bound by requests in flight --
// four independent loads per thread are issued before the first store.  K > 0: compile-time kernel size
// (the channel / tap split of the element index becomes multiplications); K = 0: any k.
template <int K>
__global__ void __launch_bounds__(256) k_patch_gather(const float *__restrict__ fmap, int B, int C, int H, int W,
                                                      const int *__restrict__ xs, const int *__restrict__ ys, int P, int k_rt,
                                                      int pad, int stride, int relu, float *__restrict__ out) {
    const int k = K > 0 ? K : k_rt;
    const int row = blockIdx.x;  // = (batch * P + p) * B + b
    const int bp = row / B, b = row - bp * B;      // bp = batch * P + p: index into xs / ys
    const int batch = bp / P;
    const int h0 = xs[bp] * stride - pad, w0 = ys[bp] * stride - pad;
    const int kk = k * k, total = C * kk;
    const float *src = fmap + (size_t(batch) * B + b) * C * H * W;
    float *dst = out + size_t(row) * total;
    constexpr int U = 4;
    for (int e0 = threadIdx.x; e0 < total; e0 += 256 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const int ch = e / kk, t = e - ch * kk, dh = t / k, dw = t - dh * k;
            const int hh = h0 + dh, ww = w0 + dw;
            v[u] = 0.f;
            if (e < total && hh >= 0 && hh < H && ww >= 0 && ww < W) v[u] = src[(size_t(ch) * H + hh) * W + ww];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            if (e < total) dst[e] = (relu && v[u] < 0.f) ? 0.f : v[u];
        }
    }
}

// k = 3 (every 3 x 3 consumer of the VGG / ResNet jobs): a thread moves whole k-wide RUNS -- one 12-byte load and one
// 12-byte store per (channel, patch row) -- instead of single elements.  The gather is bound by fabric requests in flight
// (every run costs a 64-byte request whatever it delivers): with one element per lane three lanes share a request and a wave
// instruction carries 21 of them; with one run per lane it carries 64, and a thread keeps U runs in flight.  Runs that touch
// the zero padding take the element-wise path (columns outside [0, W)) or are all zero (rows outside [0, H)).
struct __attribute__((packed, aligned(4))) Run3 {
    float v[3];
};
__global__ void __launch_bounds__(256) k_patch_gather_runs3(const float *__restrict__ fmap, int B, int C, int H, int W,
                                                            const int *__restrict__ xs, const int *__restrict__ ys, int P,
                                                            int pad, int stride, int relu, float *__restrict__ out) {
    const int row = blockIdx.x;  // = (batch * P + p) * B + b
    const int bp = row / B, b = row - bp * B;
    const int batch = bp / P;
    const int h0 = xs[bp] * stride - pad, w0 = ys[bp] * stride - pad;
    const int runs = C * 3;
    const float *src = fmap + (size_t(batch) * B + b) * C * H * W;
    float *dst = out + size_t(row) * runs * 3;
    const bool cols_inside = w0 >= 0 && w0 + 2 < W;
    constexpr int U = 4;
    for (int r0 = threadIdx.x; r0 < runs; r0 += 256 * U) {
        Run3 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 256 * u;
            const int ch = r / 3, dh = r - ch * 3, hh = h0 + dh;
            v[u].v[0] = v[u].v[1] = v[u].v[2] = 0.f;
            if (r < runs && hh >= 0 && hh < H) {
                const float *p = src + (size_t(ch) * H + hh) * W + w0;
                if (cols_inside) {
                    typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                    const f3u t = *reinterpret_cast<const f3u *>(p);   // dword-aligned 12-byte load
                    v[u].v[0] = t.x;
                    v[u].v[1] = t.y;
                    v[u].v[2] = t.z;
                } else {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
                        if (w0 + dw >= 0 && w0 + dw < W) v[u].v[dw] = p[dw];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + 256 * u;
            if (r < runs) {
                if (relu) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) v[u].v[dw] = v[u].v[dw] < 0.f ? 0.f : v[u].v[dw];
                }
                __builtin_memcpy(dst + size_t(r) * 3, &v[u], sizeof(Run3));
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_assemble_y(const float *__restrict__ feats, const float *__restrict__ bias,
                                                    const double *__restrict__ resY, int64_t total, int n,
                                                    double *__restrict__ Y) {
    int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    const int64_t step = int64_t(gridDim.x) * 256;
    for (; i < total; i += step) {
        double v = double(feats[i]) - double(bias[i % n]);
        if (resY) v += resY[i];
        Y[i] = v;
    }
}

}  // namespace

static int patch_gather_launch(cp_ctx *ctx, const float *fmap, int nb, int B, int C, int H, int W, const int32_t *xs,
                               const int32_t *ys, int P, int k, int pad, int stride, int relu, float *dst) {
    const size_t np = size_t(nb) * P;
    CP_TRY(cp_arena_reserve(ctx, np * 8 + 4096));
    cp_stage_begin(ctx);
    // (two small copies from the caller's pageable arrays: 12.6 us per call all in all; one packed copy from a ring of
    //  page-locked slots guarded by events was measured slower, 24 us per call)
    int *dx = cp_arena_take_t<int>(ctx, np), *dy = cp_arena_take_t<int>(ctx, np);
    CP_HIP(ctx, hipMemcpyAsync(dx, xs, np * 4, hipMemcpyHostToDevice, ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(dy, ys, np * 4, hipMemcpyHostToDevice, ctx->stream));
    cp_stage_mark(ctx, "gather_points_h2d");
    const unsigned grid = unsigned(np * B);
    if (k == 3)
        k_patch_gather_runs3<<<grid, 256, 0, ctx->stream>>>(fmap, B, C, H, W, dx, dy, P, pad, stride, relu, dst);
    else if (k == 1)
        k_patch_gather<1><<<grid, 256, 0, ctx->stream>>>(fmap, B, C, H, W, dx, dy, P, k, pad, stride, relu, dst);
    else
        k_patch_gather<0><<<grid, 256, 0, ctx->stream>>>(fmap, B, C, H, W, dx, dy, P, k, pad, stride, relu, dst);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "gather_kernel");
    return CP_OK;
}

extern "C" int cp_patch_gather(cp_ctx *ctx, const float *fmap, int B, int C, int H, int W, const int32_t *xs,
                               const int32_t *ys, int P, int k, int pad, int stride, int relu, float *X_out,
                               int64_t row0) {
    if (!ctx || !fmap || !xs || !ys || !X_out) return CP_ERR_ARG;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || k <= 0 || pad < 0 || stride <= 0 || row0 < 0)
        return cp_set_error(ctx, CP_ERR_ARG, "patch_gather: bad sizes");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    return patch_gather_launch(ctx, fmap, 1, B, C, H, W, xs, ys, P, k, pad, stride, relu, X_out + size_t(row0) * C * k * k);
}

// All batches of a layer in ONE launch: fmap DEVICE [nb, B, C, H, W], xs / ys HOST [nb * P] (batch-major); rows
// [(batch * P + p) * B + b] of X_out -- the reference's row order over the batches (lib/net.py:642-657).
extern "C" int cp_patch_gather_batches(cp_ctx *ctx, const float *fmap, int nb, int B, int C, int H, int W,
                                       const int32_t *xs, const int32_t *ys, int P, int k, int pad, int stride, int relu,
                                       float *X_out) {
    if (!ctx || !fmap || !xs || !ys || !X_out) return CP_ERR_ARG;
    if (nb <= 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || k <= 0 || pad < 0 || stride <= 0 ||
        int64_t(nb) * P * B > int64_t(0x7fffffff))
        return cp_set_error(ctx, CP_ERR_ARG, "patch_gather_batches: bad sizes");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    return patch_gather_launch(ctx, fmap, nb, B, C, H, W, xs, ys, P, k, pad, stride, relu, X_out);
}

extern "C" int cp_assemble_y(cp_ctx *ctx, const float *feats, const float *bias, const double *resY, int64_t N, int n,
                             double *Y) {
    if (!ctx || !feats || !bias || !Y || N <= 0 || n <= 0) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t total = N * n;
    int blocks = int(std::min<int64_t>((total + 255) / 256, int64_t(ctx->cu_count) * 8));
    k_assemble_y<<<blocks, 256, 0, ctx->stream>>>(feats, bias, resY, total, n, Y);
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}
