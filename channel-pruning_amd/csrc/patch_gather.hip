// Operand assembly kernels (HBM-bound gathers, no arithmetic to speak of):
//   cp_patch_gather  sampled-point im2col of Net.extract_XY (lib/net.py:629-657) fused with the
//                    [N*k*k, C] -> [N, C, k, k] re-layout of lib/net.py:1702 and the VGG ReLU of
//                    lib/net.py:1720
//   cp_assemble_y    Y = feats - bias (+ resY)            lib/net.py:1707, 1716-1722
#include "cp_common.h"

namespace {

// One workgroup per (point, image) output row; threads sweep the C*k*k patch elements, which are
// contiguous in the output (coalesced stores); reads walk k-wide runs of the feature map.
__global__ void __launch_bounds__(256) k_patch_gather(const float *__restrict__ fmap, int B, int C, int H, int W,
                                                      const int *__restrict__ xs, const int *__restrict__ ys, int k,
                                                      int pad, int stride, int relu, float *__restrict__ out) {
    const int row = blockIdx.x;  // = p * B + b
    const int p = row / B, b = row - p * B;
    const int h0 = xs[p] * stride - pad, w0 = ys[p] * stride - pad;
    const int kk = k * k, total = C * kk;
    const float *src = fmap + size_t(b) * C * H * W;
    float *dst = out + size_t(row) * total;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int ch = e / kk, t = e - ch * kk, dh = t / k, dw = t - dh * k;
        const int hh = h0 + dh, ww = w0 + dw;
        float v = 0.f;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = src[(size_t(ch) * H + hh) * W + ww];
        if (relu && v < 0.f) v = 0.f;
        dst[e] = v;
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

extern "C" int cp_patch_gather(cp_ctx *ctx, const float *fmap, int B, int C, int H, int W, const int32_t *xs,
                               const int32_t *ys, int P, int k, int pad, int stride, int relu, float *X_out,
                               int64_t row0) {
    if (!ctx || !fmap || !xs || !ys || !X_out) return CP_ERR_ARG;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || k <= 0 || pad < 0 || stride <= 0 || row0 < 0)
        return cp_set_error(ctx, CP_ERR_ARG, "patch_gather: bad sizes");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_TRY(cp_arena_reserve(ctx, size_t(P) * 8 + 4096));
    int *dx = cp_arena_take_t<int>(ctx, P), *dy = cp_arena_take_t<int>(ctx, P);
    CP_HIP(ctx, hipMemcpyAsync(dx, xs, size_t(P) * 4, hipMemcpyHostToDevice, ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(dy, ys, size_t(P) * 4, hipMemcpyHostToDevice, ctx->stream));
    float *dst = X_out + size_t(row0) * C * k * k;
    k_patch_gather<<<P * B, 256, 0, ctx->stream>>>(fmap, B, C, H, W, dx, dy, k, pad, stride, relu, dst);
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
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
