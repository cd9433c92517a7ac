// Context, error reporting, device-memory helpers, stage timing and the two roofline
// micro-probes of libcpmi355.so.
#include "cp_common.h"

#include <atomic>

#include <algorithm>
#include <vector>

#include <chrono>

int cp_set_error(cp_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

extern "C" int cp_version(void) { return CP_VERSION; }

extern "C" const char *cp_strerror(int code) {
    switch (code) {
        case CP_OK: return "ok";
        case CP_ERR_ARG: return "invalid argument";
        case CP_ERR_HIP: return "HIP runtime error";
        case CP_ERR_NOMEM: return "out of device memory";
        case CP_ERR_UNSUPPORTED: return "unsupported shape";
        case CP_ERR_NUMERIC: return "numerical breakdown";
        case CP_ERR_NODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

extern "C" const char *cp_last_error(const cp_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

extern "C" int cp_device_count(int *count) {
    if (!count) return CP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return CP_ERR_NODEVICE;
    }
    *count = n;
    return CP_OK;
}

#include <mutex>
hipStream_t cp_side_stream(cp_ctx *ctx) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    if (ctx->device < 0 || ctx->device >= 64) return ctx->stream;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[ctx->device]) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return ctx->stream;
        streams[ctx->device] = st;
    }
    return streams[ctx->device];
}

extern "C" int cp_ctx_create(int device, cp_ctx **out) {
    // CP_CTX_PRIORITY (read, never written, by the library): the default HIP priority of a context's stream
    int prio = 0;
    if (const char *pv = getenv("CP_CTX_PRIORITY")) prio = atoi(pv);
    return cp_ctx_create_priority(device, prio, out);
}

extern "C" int cp_ctx_create_priority(int device, int prio, cp_ctx **out) {
    if (!out) return CP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CP_ERR_NODEVICE;
    if (device < 0 || device >= n) return CP_ERR_ARG;
    cp_ctx *ctx = new cp_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return CP_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->cu_count = prop.multiProcessorCount;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            // The code object only carries gfx950 ISA; any other device cannot run it.
            delete ctx;
            return CP_ERR_NODEVICE;
        }
    }
    // prio < 0 = higher: a resident layer set may raise it for the layers on the job's critical path
    // (cpmi355.shard.ResidentLayerSet), whose short dependent kernels then do not queue behind the long products of layers
    // with slack
    hipError_t se;
    if (prio != 0) {
        int least = 0, greatest = 0;
        hipDeviceGetStreamPriorityRange(&least, &greatest);   // numerically lower = higher priority
        prio = std::max(greatest, std::min(least, prio));
        se = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, prio);
    } else {
        se = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) {
        delete ctx;
        return CP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    for (int i = 0; i < 2 * CP_MAX_STAGES; ++i) hipEventCreate(&ctx->ev[i]);
    *out = ctx;
    return CP_OK;
}

// A context that runs on `of`'s stream and owns no stream of its own (every HIP stream claims a hardware queue, and
// dispatch gets slow for everybody once ~20 of them exist): the per-job contexts of cp_prune_layers.  Destroy it
// before `of`.
extern "C" int cp_ctx_create_sibling(cp_ctx *of, cp_ctx **out) {
    if (!of || !out) return CP_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(of->device) != hipSuccess) return CP_ERR_HIP;
    cp_ctx *ctx = new cp_ctx();
    ctx->device = of->device;
    ctx->cu_count = of->cu_count;
    ctx->own_stream = nullptr;
    ctx->stream = of->stream;
    for (int i = 0; i < 2 * CP_MAX_STAGES; ++i) hipEventCreate(&ctx->ev[i]);
    *out = ctx;
    return CP_OK;
}

extern "C" int cp_ctx_destroy(cp_ctx *ctx) {
    if (!ctx) return CP_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    cp_precompute_release(ctx);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->layer_ws) hipFree(ctx->layer_ws);
    if (ctx->cd_box) hipFree(ctx->cd_box);
    if (ctx->gemm_cnt) hipFree(ctx->gemm_cnt);
    if (ctx->pinned) hipHostFree(ctx->pinned);
    if (ctx->stage) hipHostFree(ctx->stage);
    if (ctx->ev_upload) hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_epoch) hipEventDestroy(ctx->ev_epoch);
    for (int i = 0; i < 2 * CP_MAX_STAGES; ++i)
        if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return CP_OK;
}

extern "C" int cp_ctx_set_stream(cp_ctx *ctx, void *hip_stream) {
    if (!ctx) return CP_ERR_ARG;
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return CP_OK;
}

extern "C" int cp_sync(cp_ctx *ctx) {
    if (!ctx) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CP_OK;
}

extern "C" int cp_malloc(cp_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return cp_set_error(ctx, CP_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_free(cp_ctx *ctx, void *dptr) {
    if (!ctx) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    if (dptr) CP_HIP(ctx, hipFree(dptr));
    return CP_OK;
}

extern "C" int cp_memcpy_h2d(cp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));  // the calling thread may never have selected this device
    if (ctx->pre.ready && (dst == ctx->pre.X || dst == ctx->pre.Y)) cp_precompute_void(ctx);   // new contents
    // pageable source: hipMemcpyAsync stages it before returning, so the caller may reuse src.
    CP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CP_OK;
}

extern "C" int cp_memcpy_d2h(cp_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CP_OK;
}

extern "C" int cp_memset(cp_ctx *ctx, void *dst, int value, size_t bytes) {
    if (!ctx || (bytes && !dst)) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    CP_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return CP_OK;
}

hipError_t cp_stream_wait(cp_ctx *ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return e;
}

// ---- arena ---------------------------------------------------------------------------
int cp_arena_reserve(cp_ctx *ctx, size_t bytes) {
    bytes = cp_align_up(bytes + 4096, 1 << 20);
    if (bytes > ctx->arena_bytes) {
        CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->arena) CP_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->arena), want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(reinterpret_cast<void **>(&ctx->arena), want);
        }
        if (e != hipSuccess)
            return cp_set_error(ctx, CP_ERR_NOMEM, "arena hipMalloc(%zu): %s", want, hipGetErrorString(e));
        ctx->arena_bytes = want;
    }
    ctx->arena_used = 0;
    return CP_OK;
}

void *cp_arena_take(cp_ctx *ctx, size_t bytes) {
    size_t off = cp_align_up(ctx->arena_used, 256);
    if (off + bytes > ctx->arena_bytes) return nullptr;
    ctx->arena_used = off + bytes;
    return ctx->arena + off;
}

int cp_pinned_reserve(cp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return CP_OK;
    if (ctx->pinned) {
        CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        CP_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    bytes = cp_align_up(bytes, 4096);
    CP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pinned), bytes, hipHostMallocDefault));
    ctx->pinned_bytes = bytes;
    return CP_OK;
}

// ---- stage timing ----------------------------------------------------------------------
void cp_stage_begin(cp_ctx *ctx) {
    if (!ctx->timing || ctx->timing_gram_only || ctx->n_marks >= 2 * CP_MAX_STAGES) return;
    ctx->mark_names[ctx->n_marks] = nullptr;
    hipEventRecord(ctx->ev[ctx->n_marks], ctx->stream);
    ++ctx->n_marks;
}

void cp_stage_mark(cp_ctx *ctx, const char *name) {
    if (!ctx->timing || ctx->n_marks >= 2 * CP_MAX_STAGES) return;
    // mode 2: only the events that bracket the alpha search, the refit Gram GEMM, the factorisation chain (Cholesky steps
    // with the forward substitution riding along) and the backward substitution (every event is a packet in the stream)
    if (ctx->timing_gram_only && strcmp(name, "refit_gram_begin") != 0 && strcmp(name, "refit_gram_gemm") != 0 &&
        strcmp(name, "refit_chol_begin") != 0 && strcmp(name, "refit_cholesky") != 0 && strcmp(name, "refit_solve") != 0 &&
        strcmp(name, "cd_search_begin") != 0 && strcmp(name, "cd_alpha_search") != 0)
        return;
    ctx->mark_names[ctx->n_marks] = name;
    hipEventRecord(ctx->ev[ctx->n_marks], ctx->stream);
    ++ctx->n_marks;
}

void cp_stage_finish(cp_ctx *) {}

extern "C" int cp_enable_stage_timing(cp_ctx *ctx, int on) {
    if (!ctx) return CP_ERR_ARG;
    ctx->timing = on != 0;
    ctx->timing_gram_only = on == 2;
    ctx->n_marks = 0;
    ctx->n_stages = 0;
    if (ctx->pre.worker) cp_enable_stage_timing(ctx->pre.worker, on);
    return CP_OK;
}

extern "C" int cp_stage_epoch(cp_ctx *ctx) {
    if (!ctx) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->ev_epoch) CP_HIP(ctx, hipEventCreate(&ctx->ev_epoch));
    CP_HIP(ctx, hipEventRecord(ctx->ev_epoch, ctx->stream));
    return CP_OK;
}

// Resolves and clears the marks recorded since the previous call (synchronises the stream).  epoch_of / begin_ms (both or
// neither): when each bracket began, in ms after epoch_of's cp_stage_epoch (-1 where that cannot be told).
extern "C" int cp_last_stage_spans(cp_ctx *ctx, cp_ctx *epoch_of, int *count, float *ms, float *begin_ms) {
    if (!ctx || !count || !ms || (begin_ms != nullptr) != (epoch_of != nullptr)) return CP_ERR_ARG;
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipEvent_t epoch = epoch_of ? epoch_of->ev_epoch : nullptr;
    if (epoch && hipEventSynchronize(epoch) != hipSuccess) epoch = nullptr;
    ctx->n_stages = 0;
    auto resolve = [&](cp_ctx *c) {
        for (int i = 1; i < c->n_marks && ctx->n_stages < CP_MAX_STAGES; ++i) {
            if (!c->mark_names[i]) continue;
            float t = 0.f, b = -1.f;
            if (hipEventElapsedTime(&t, c->ev[i - 1], c->ev[i]) != hipSuccess) t = -1.f;
            if (begin_ms && (!epoch || hipEventElapsedTime(&b, epoch, c->ev[i - 1]) != hipSuccess)) b = -1.f;
            ctx->stage_names[ctx->n_stages] = c->mark_names[i];
            ctx->stage_ms[ctx->n_stages] = t;
            if (begin_ms) begin_ms[ctx->n_stages] = b;
            ++ctx->n_stages;
        }
        c->n_marks = 0;
    };
    resolve(ctx);
    if (cp_ctx *w = ctx->pre.worker) {   // the stages of the overlapped precompute (side stream) belong to this call too
        hipStreamSynchronize(w->stream);
        resolve(w);
    }
    *count = ctx->n_stages;
    for (int i = 0; i < ctx->n_stages; ++i) ms[i] = ctx->stage_ms[i];
    return CP_OK;
}

extern "C" int cp_last_stage_times(cp_ctx *ctx, int *count, float *ms) {
    return cp_last_stage_spans(ctx, nullptr, count, ms, nullptr);
}

extern "C" const char *cp_stage_name(cp_ctx *ctx, int index) {
    if (!ctx || index < 0 || index >= ctx->n_stages) return "";
    return ctx->stage_names[index];
}

// ---- roofline micro-probes -----------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));

// 8 independent accumulators per wave, 4 waves per SIMD: measures the sustained issue
// rate of v_mfma_f64_16x16x4_f64 (2*16*16*4 = 2048 FLOP per wave-instruction).
// The register budget is cut for >= 2 waves per SIMD ON PURPOSE: with a whole SIMD's register file to itself the compiler
// keeps the accumulators in AccVGPRs, and the AccVGPR form of this instruction issues once per ~107 cycles and SIMD
// (46.7 TFLOP/s) where the VGPR form -- what every kernel of the library uses -- issues once per 64-65 (77 TFLOP/s, the
// nominal figure): tools/ubench/gemm_probe.hip part (0), profiles/r04_gemm_probe.md.  Rounds 1-4 quoted the AccVGPR
// number as "what the matrix pipe delivers".
__global__ void __launch_bounds__(256, 2) k_probe_mfma_f64(double *out, int iters) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0., 0., 0., 0.};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;  // keep the chain live
}

extern "C" int cp_probe_mfma_f64(cp_ctx *ctx, double *tflops) {
    if (!ctx || !tflops) return CP_ERR_ARG;
    CP_TRY(cp_arena_reserve(ctx, 4096));
    double *out = cp_arena_take_t<double>(ctx, 8);
    const int iters = 4000, blocks = ctx->cu_count * 4;
    hipEvent_t e0, e1;
    CP_HIP(ctx, hipEventCreate(&e0));
    CP_HIP(ctx, hipEventCreate(&e1));
    k_probe_mfma_f64<<<blocks, 256, 0, ctx->stream>>>(out, 100);  // warm-up
    CP_HIP(ctx, hipEventRecord(e0, ctx->stream));
    k_probe_mfma_f64<<<blocks, 256, 0, ctx->stream>>>(out, iters);
    CP_HIP(ctx, hipEventRecord(e1, ctx->stream));
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    CP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    double flops = double(blocks) * 4 /*waves*/ * iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    return CP_OK;
}

// the same loop with clock stamps: st[wave] = {shader cycles (s_memtime), 100 MHz ticks (s_memrealtime)} around the loop
__global__ void __launch_bounds__(256, 2) k_probe_mfma_f64_clock(double *out, unsigned long long *st, int iters) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0., 0., 0., 0.};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678) out[0] = s;  // keep the chain live
    if ((threadIdx.x & 63) == 0) {
        const size_t w = size_t(blockIdx.x) * (blockDim.x / 64) + threadIdx.x / 64;
        st[2 * w] = t1 - t0;
        st[2 * w + 1] = r1 - r0;
    }
}

extern "C" int cp_probe_mfma_f64_clock(cp_ctx *ctx, double *tflops, double *ghz, double *cycles_per_mfma) {
    if (!ctx || !tflops || !ghz || !cycles_per_mfma) return CP_ERR_ARG;
    // 2 workgroups of 4 waves per CU = 2 waves per SIMD, 8 independent accumulators each, accumulators in VGPRs (see
    // k_probe_mfma_f64): 72-77 TFLOP/s at 2 / 4 waves per SIMD, 64-69 cycles per instruction and SIMD (profiles/r04_gemm_probe.md)
    const int iters = 20000, blocks = ctx->cu_count * 2, waves = blocks * 4;
    CP_TRY(cp_arena_reserve(ctx, 4096 + size_t(waves) * 16));
    double *out = cp_arena_take_t<double>(ctx, 8);
    unsigned long long *st = cp_arena_take_t<unsigned long long>(ctx, size_t(waves) * 2);
    if (!out || !st) return cp_set_error(ctx, CP_ERR_NOMEM, "probe arena");
    hipEvent_t e0, e1;
    CP_HIP(ctx, hipEventCreate(&e0));
    CP_HIP(ctx, hipEventCreate(&e1));
    k_probe_mfma_f64_clock<<<blocks, 256, 0, ctx->stream>>>(out, st, 100);  // warm-up
    CP_HIP(ctx, hipEventRecord(e0, ctx->stream));
    k_probe_mfma_f64_clock<<<blocks, 256, 0, ctx->stream>>>(out, st, iters);
    CP_HIP(ctx, hipEventRecord(e1, ctx->stream));
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    CP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    std::vector<unsigned long long> h(size_t(waves) * 2);
    CP_HIP(ctx, hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> g(waves), cyc(waves);
    for (int w = 0; w < waves; ++w) {
        cyc[w] = double(h[2 * w]);
        g[w] = h[2 * w + 1] ? double(h[2 * w]) / double(h[2 * w + 1]) * 0.1 : 0.0;
    }
    std::sort(g.begin(), g.end());
    std::sort(cyc.begin(), cyc.end());
    *tflops = double(waves) * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
    *ghz = g[waves / 2];
    *cycles_per_mfma = cyc[waves / 2] / (2.0 * iters * 8.0);   // a SIMD runs two of the waves
    return CP_OK;
}

__global__ void __launch_bounds__(256) k_probe_copy(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                    size_t n) {
    size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

extern "C" int cp_probe_hbm_copy(cp_ctx *ctx, size_t bytes, double *gbps) {
    if (!ctx || !gbps || bytes < (1 << 20)) return CP_ERR_ARG;
    bytes = bytes / 16 * 16;
    CP_TRY(cp_arena_reserve(ctx, 2 * bytes + 8192));
    float4 *src = reinterpret_cast<float4 *>(cp_arena_take(ctx, bytes));
    float4 *dst = reinterpret_cast<float4 *>(cp_arena_take(ctx, bytes));
    if (!src || !dst) return cp_set_error(ctx, CP_ERR_NOMEM, "probe arena");
    CP_HIP(ctx, hipMemsetAsync(src, 1, bytes, ctx->stream));
    size_t n = bytes / 16;
    int blocks = ctx->cu_count * 8;
    hipEvent_t e0, e1;
    CP_HIP(ctx, hipEventCreate(&e0));
    CP_HIP(ctx, hipEventCreate(&e1));
    k_probe_copy<<<blocks, 256, 0, ctx->stream>>>(src, dst, n);
    CP_HIP(ctx, hipEventRecord(e0, ctx->stream));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_probe_copy<<<blocks, 256, 0, ctx->stream>>>(src, dst, n);
    CP_HIP(ctx, hipEventRecord(e1, ctx->stream));
    CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    CP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *gbps = 2.0 * double(bytes) * reps / (ms * 1e-3) / 1e9;
    return CP_OK;
}

// ---- process-wide experiment switches (A/B measurements in one process: tools/probes/job_knobs.py) --------------------------
namespace {
std::atomic<int> g_knobs[CP_KNOB_COUNT] = {};
}
int cp_knob(int id) { return id >= 0 && id < CP_KNOB_COUNT ? g_knobs[id].load(std::memory_order_relaxed) : 0; }
// cp_debug_knob(id, value): sets switch `id` (CP_KNOB_*, cp_common.h), returns the previous value (-1: no such switch)
extern "C" int cp_debug_knob(int id, int value) {
    if (id < 0 || id >= CP_KNOB_COUNT) return -1;
    return g_knobs[id].exchange(value, std::memory_order_relaxed);
}
