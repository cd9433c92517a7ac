// single-wave issue cost of vector-memory / LDS reads (gfx950): N independent loads back to back,
// one wait at the end; (cycles(N=32) - cycles(N=8)) / 24 = issue cost per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int MODE, int N>
__global__ void __launch_bounds__(64) k(const double* Q, double* out, unsigned long long* cyc, int iters, int stride) {
    __shared__ double lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = Q[i];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Q), 0, 1 << 26, 0x00020000);
    const unsigned lane = threadIdx.x;
    double acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        unsigned base = (unsigned)(it & 7) * (unsigned)stride;  // sgpr row offset
        if (MODE == 0) {  // buffer_load_dwordx2
            v2u v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8u, base + j * 2048u, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += __hiloint2double(int(v[j][1]), int(v[j][0]));
        } else if (MODE == 1) {  // buffer_load_dwordx4
            v4u v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, base + j * 2048u, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += __hiloint2double(int(v[j][1]), int(v[j][0])) + __hiloint2double(int(v[j][3]), int(v[j][2]));
        } else if (MODE == 2) {  // ds_read_b64
            double v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = lds[((it + j) & 63) * 64 + lane];
#pragma unroll
            for (int j = 0; j < N; ++j) acc += v[j];
        } else if (MODE == 3) {  // ds_read_b128
            double2 v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = *reinterpret_cast<const double2*>(&lds[((it + j) & 31) * 128 + 2 * lane]);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += v[j].x + v[j].y;
        } else if (MODE == 4) {  // global_load_dwordx2 (flat addressing)
            double v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = Q[(base >> 3) + j * 256 + lane];
#pragma unroll
            for (int j = 0; j < N; ++j) acc += v[j];
        } else if (MODE == 5) {  // buffer_load_dword
            unsigned v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4u, base + j * 2048u, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += double(v[j]);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int N>
static double run(const double* Q, double* out, unsigned long long* cyc, int iters) {
    k<MODE, N><<<1, 64>>>(Q, out, cyc, 10, 16384);
    k<MODE, N><<<1, 64>>>(Q, out, cyc, iters, 16384);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return double(c) / iters;
}

int main() {
    double *Q, *out;
    unsigned long long* cyc;
    CHK(hipMalloc(&Q, 1 << 26));
    CHK(hipMemset(Q, 0, 1 << 26));
    CHK(hipMalloc(&out, 64 * 8));
    CHK(hipMalloc(&cyc, 8));
    const int it = 2000;
    const char* names[] = {"buffer_load_dwordx2", "buffer_load_dwordx4", "ds_read_b64", "ds_read_b128", "global_load_dwordx2", "buffer_load_dword"};
#define ROW(M) { double a = run<M, 8>(Q, out, cyc, it), b = run<M, 16>(Q, out, cyc, it), c = run<M, 32>(Q, out, cyc, it); \
    printf("%-22s N=8 %.0f  N=16 %.0f  N=32 %.0f cycles/iter -> issue %.1f cycles/instr (incl. consumer add)\n", names[M], a, b, c, (c - a) / 24.0); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    return 0;
}
