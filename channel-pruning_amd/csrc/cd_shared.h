// Primitives shared by the coordinate-descent kernels (cd_gram.hip: one- and two-wave forms; cd_team.hip: chain wave +
// several keeper waves): wave reductions, SGPR-descriptor row loads, the xorshift index stream of
// sklearn/utils/_random.pxd:20-35 produced 64 values at a time, LDS flag accesses, and the argument blocks of the
// alpha-search launches.
#pragma once
#include "cp_common.h"
#include "xorshift_jump.h"

namespace cdk {

constexpr int WAVE = 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o, WAVE);
        v = u > v ? u : v;
    }
    return v;
}

// uniform-lane read of a double held in `v` (lane index is wave-uniform)
__device__ __forceinline__ double read_lane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));

// 8-byte row-element load: SGPR descriptor + SGPR row offset + per-lane column offset.  No
// address arithmetic on the vector unit, and the load stays on the vector-memory path (an
// s_load would share lgkmcnt with the LDS reads and force full drains).
__device__ __forceinline__ double load_q(__amdgpu_buffer_rsrc_t rsrc, uint32_t col_bytes, uint32_t row_bytes) {
    const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col_bytes, row_bytes, 0);
    return __hiloint2double(int(v[1]), int(v[0]));
}

typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
// 16-byte variant: two consecutive row elements per lane (one vector-memory issue instead of two)
__device__ __forceinline__ void load_q2(__amdgpu_buffer_rsrc_t rsrc, uint32_t col_bytes, uint32_t row_bytes, double &a,
                                        double &b) {
    const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, col_bytes, row_bytes, 0);
    a = __hiloint2double(int(v[1]), int(v[0]));
    b = __hiloint2double(int(v[3]), int(v[2]));
}

#ifndef CP_CD_PACKED
#define CP_CD_PACKED 1
#endif

// Coordinate stream rand_int(c) of _cd_fast.pyx:30-32, produced 64 values at a time
// (xorshift_jump.h): lane l of `idx`/`off` holds the coordinate / row byte offset of value
// 64*batch + l; `pos` (wave-uniform) is the next lane to hand out.
struct IdxStream {
    uint32_t st, idx, off;
    uint32_t n, row_stride_bytes;
    uint64_t magic;
    __device__ __forceinline__ void derive() {
        idx = cpx::fastmod(st & 0x7fffffffu, magic, n);
        off = idx * row_stride_bytes;
    }
    __device__ __forceinline__ void init(uint32_t seed, uint32_t n_, uint32_t row_stride_bytes_, int lane) {
        n = n_;
        row_stride_bytes = row_stride_bytes_;
        magic = cpx::fastmod_magic(n_);
        uint32_t s = seed == 0 ? 1u : seed;  // _random.pxd:24-25
        for (int i = 0; i <= lane; ++i) s = cpx::xs_step(s);
        st = s;
        derive();
    }
    // value number `pos` (wave-uniform, < 64) of the current batch
    __device__ __forceinline__ void take(int pos, int &ii, uint32_t &row_off) const {
        ii = __builtin_amdgcn_readlane(int(idx), pos);
        row_off = uint32_t(__builtin_amdgcn_readlane(int(off), pos));
    }
    __device__ __forceinline__ void next_batch() {
        st = cpx::xs_jump64(st);
        derive();
    }
    // drop the first k (< 64) values of the batch: lane l takes over value l + k
    __device__ __forceinline__ void realign(int k, int lane) {
        const uint32_t rot = uint32_t(__shfl(int(st), (lane + k) & 63, WAVE));
        const uint32_t adv = cpx::xs_jump64(rot);
        st = lane + k >= 64 ? adv : rot;
        derive();
    }
};

struct FitOut {
    double gap;
    int n_iter;
    int nnz;
    double edge_margin = -1.0, gap_margin = -1.0;
};

// Flags and payloads all live in LDS, and the LDS executes one wave's instructions in program order:
// a flag written after its payload lands after it, a payload read after the flag read sees what the
// flag announced.  So the hand-offs need no s_waitcnt of their own (an acquire / release atomic would
// also drain the outstanding vector-memory prefetches) -- only the compiler must keep the order.
// (explicit LDS address space: a volatile access through a generic pointer stays a FLAT instruction
// with a full vmcnt(0) drain around it)
typedef __attribute__((address_space(3))) volatile int duo_lds_vint;
__device__ __forceinline__ int duo_load(int *p) {
    const int v = *(duo_lds_vint *)p;
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void duo_store(int *p, int v) {
    asm volatile("" ::: "memory");
    *(duo_lds_vint *)p = v;
}
struct DevResult {  // mirrors cp_cd_result
    double gap;
    double tol_scaled;
    int32_t n_iter;
    int32_t nnz;
    double edge_margin;  // tie sentinels (-1: not tracked by this kernel form)
    double gap_margin;
};

struct CdSearchArgs {
    const double *Q;
    int ldq;
    const double *q, *stats;
    int c;
    double M, right0, rank, lbound, rbound;
    const uint32_t *seeds;
    int max_fits, max_iter;
    double tol;
    int flags;
    double *w, *w_host;
    DevResult *log;
    double *log_alpha;
    int *fits_used;
    double *alpha_out;
};
constexpr int CP_CD_MAX_BATCH = 16;
struct CdSearchBatch {
    CdSearchArgs a[CP_CD_MAX_BATCH];
};


}  // namespace cdk
