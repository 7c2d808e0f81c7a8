// Batched products of the BIE's C x C matrices (include/bmc_hip.h: bmc_small_mm).
//
// The attention of a BIE (reference models/submodules.py:66-73) is evaluated without the value tensor (bmc_hip/bie.py): what
// remains of the value convolution are products of per-sample C x C matrices with the C x C weight matrix of the sample's
// weight group -- att = scale (G0 W^T + s b^T), P W, da W, P^T dM + da^T G0, ... -- 16 samples of 128^3 multiply-adds each: work
// for microseconds, where the cost is the NUMBER of launches (eight library GEMMs, their bias / outer-product updates and the
// transposes and concatenations around them were ~25 launches per BIE and 17 ms per step).  One kernel with strided operands
// covers them all:
//     C[b][i][j]  (=|+=) alpha * ( sum_t sum_k A_t[b][i][k] B_t[b][k][j]  +  u[b][i] v[b][j] )
//     vec[b][i]   (=|+=) alpha *   sum_t sum_k A_t[b][i][k] w_t[b][k]
// t = 1 or 2 terms; every operand is addressed as base + b * batch_stride + (b / batch_per_group) * group_stride + row and
// column strides, so transposes, per-group weights, and writes into a column range of a wider matrix (the concatenated
// per-sample weights of the following 1x1 launch) cost nothing.
// One workgroup = one 32 x 32 tile of one batch, 256 threads x (2 x 2) outputs, K in steps of 32 through LDS; plain fp32 FMAs in
// a fixed order (deterministic; no matrix cores: 4 MFLOP per tile).
#include "bmc_common.h"

namespace {

struct MmK { bmc_small_mm_args_t a; };

constexpr int T = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const float* opbase(const float* p, long long sb, long long sg, int b, int g) {
    return p + (long long)b * sb + (long long)g * sg;
}

__global__ __launch_bounds__(256) void small_mm_kernel(const MmK k) {
    const bmc_small_mm_args_t& a = k.a;
    // rows of As are read four k at a time (16-byte aligned: 36 floats per row), rows of Bs two columns at a time (34)
    __shared__ __attribute__((aligned(16))) float As[T][T + 4];
    __shared__ __attribute__((aligned(16))) float Bs[T][T + 2];
    __shared__ __attribute__((aligned(16))) float ws[T];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z, g = b / a.batch_per_group;
    const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, vacc[2] = {0.f, 0.f};
    const bool want_vec = a.vec_out != nullptr && blockIdx.x == 0;
    // flattened (term, K step) iteration space, software-pipelined: the global loads of step s + 1 are in flight while step s
    // is multiplied out of LDS (the kernel is a chain of load latencies otherwise: 17 us for 4 steps)
    const int ksteps = (a.K + T - 1) / T, nsteps = a.nterms * ksteps;
    float ra[4], rb[4], rw = 0.f;
    auto fetch = [&](int s_) {
        const bmc_mm_term_t& m = a.t[s_ / ksteps];
        const int k0 = (s_ % ksteps) * T;
        const float* A = opbase(m.a, m.a_sb, m.a_sg, b, g);
        const float* B = opbase(m.b, m.b_sb, m.b_sg, b, g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, hi = e >> 5, lo = e & 31;
            // the unit-stride index runs fastest over the lanes
            const int ai = m.a_sk == 1 ? hi : lo, ak = m.a_sk == 1 ? lo : hi;
            ra[q] = (i0 + ai < a.M && k0 + ak < a.K) ? ldg4(A + (long long)(i0 + ai) * m.a_si + (long long)(k0 + ak) * m.a_sk) : 0.f;
            const int bk = m.b_sj == 1 ? hi : lo, bj = m.b_sj == 1 ? lo : hi;
            rb[q] = (k0 + bk < a.K && j0 + bj < a.N) ? ldg4(B + (long long)(k0 + bk) * m.b_sk + (long long)(j0 + bj) * m.b_sj) : 0.f;
        }
        if (want_vec && tid < T) rw = k0 + tid < a.K ? ldg4(opbase(m.w, m.w_sb, m.w_sg, b, g) + (long long)(k0 + tid) * m.w_sk) : 0.f;
    };
    fetch(0);
    for (int s_ = 0; s_ < nsteps; ++s_) {
        const bmc_mm_term_t& m = a.t[s_ / ksteps];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q, hi = e >> 5, lo = e & 31;
            As[m.a_sk == 1 ? hi : lo][m.a_sk == 1 ? lo : hi] = ra[q];
            Bs[m.b_sj == 1 ? hi : lo][m.b_sj == 1 ? lo : hi] = rb[q];
        }
        if (want_vec && tid < T) ws[tid] = rw;
        __syncthreads();
        if (s_ + 1 < nsteps) fetch(s_ + 1);
#pragma unroll
        for (int kk = 0; kk < T; kk += 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&As[2 * ty][kk]), a1 = *reinterpret_cast<const f32x4*>(&As[2 * ty + 1][kk]);
            const f32x4 w4 = want_vec ? *reinterpret_cast<const f32x4*>(&ws[kk]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 bq = *reinterpret_cast<const f32x2*>(&Bs[kk + q][2 * tx]);
                acc[0][0] += a0[q] * bq[0]; acc[0][1] += a0[q] * bq[1];
                acc[1][0] += a1[q] * bq[0]; acc[1][1] += a1[q] * bq[1];
                vacc[0] += a0[q] * w4[q]; vacc[1] += a1[q] * w4[q];
            }
        }
        __syncthreads();
    }
    const float* u = a.u ? a.u + (long long)b * a.u_sb + (long long)g * a.u_sg : nullptr;
    const float* v = a.v ? a.v + (long long)b * a.v_sb + (long long)g * a.v_sg : nullptr;
    float* C = a.c ? a.c + (long long)b * a.c_sb + (long long)g * a.c_sg : nullptr;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = i0 + 2 * ty + r;
        if (i >= a.M) continue;
        if (C) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = j0 + 2 * tx + c;
                if (j >= a.N) continue;
                float val = acc[r][c];
                if (u) val += ldg4(u + i) * ldg4(v + j);
                val *= a.alpha;
                float* o = C + (long long)i * a.c_si + (long long)j * a.c_sj;
                *o = a.accumulate ? *o + val : val;
            }
        }
        if (want_vec && tx == 0) {
            float* o = a.vec_out + (long long)b * a.vo_sb + (long long)g * a.vo_sg + i;
            const float val = a.alpha * vacc[r];
            *o = a.accumulate ? *o + val : val;
        }
    }
}

}  // namespace

extern "C" int bmc_small_mm(const bmc_small_mm_args_t* host_args, bmc_stream_t stream) {
    BMC_CHECK_ARG(host_args, "bmc_small_mm: null arguments");
    const bmc_small_mm_args_t& a = *host_args;
    BMC_CHECK_ARG(a.nterms >= 1 && a.nterms <= 2 && a.nbatch >= 1 && a.batch_per_group >= 1 && a.M >= 1 && a.N >= 1 && a.K >= 1,
                  "bmc_small_mm: bad sizes (terms %d, batch %d / %d, M %d N %d K %d)", a.nterms, a.nbatch, a.batch_per_group, a.M, a.N, a.K);
    BMC_CHECK_ARG(a.c || a.vec_out, "bmc_small_mm: no output");
    BMC_CHECK_ARG((a.u == nullptr) == (a.v == nullptr), "bmc_small_mm: u and v go together");
    for (int t = 0; t < a.nterms; ++t)
        BMC_CHECK_ARG(a.t[t].a && a.t[t].b && (!a.vec_out || a.t[t].w), "bmc_small_mm: term %d: null operand (vec_out needs w in every term)", t);
    BMC_CHECK_ARG(a.nbatch <= 65535, "bmc_small_mm: more than 65535 batches");
    MmK k;
    k.a = a;
    dim3 grid((unsigned)((a.N + T - 1) / T), (unsigned)((a.M + T - 1) / T), (unsigned)a.nbatch);
    hipLaunchKernelGGL(small_mm_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
    BMC_CHECK_LAUNCH("bmc_small_mm");
    return 0;
}
