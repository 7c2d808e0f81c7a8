// Weight gradient of a dense 3x3, 128 -> 128 channel convolution through the Winograd transform F(2x2, 3x3) on the fp32
// matrix cores: the pixel-reduction GEMM of pgemm.hip (9 taps x 4 pixels = 36 multiplies per 2x2 output tile and channel
// pair) becomes 16 multiplies in the transformed domain,
//     dU[xi][nu][co][ci] = sum over tiles  dM[xi][nu][tile][co] * V[xi][nu][tile][ci],     dM = A dY A^T,  V = B^T d B,
//     dW = G^T dU G   (+ bias gradient = sum over tiles of dM[1][1], which is the plain sum of the tile's four dY pixels),
// with d the 4x4 input patch and dY the 2x2 output-gradient patch of a tile.  (The forward kernel is wino.hip.)
//
// Work split.  One workgroup = ONE row xi of the 4x4 position grid: 4 positions nu x 128 co x 128 ci = 65 536 accumulators
// = 128 per thread of an 8-wave workgroup (wave = position nu = w >> 1, output-channel half w & 1: 4 x 8 tiles of the
// 16x16x4 MFMA).  With xi fixed the transforms separate without duplicated work: the row combination of B^T (two of the
// patch's four rows) / of A (one or two of dY's two rows) is made once and the four column combinations follow from it, so
// the four workgroup types together execute exactly the 32 + 12 additions per tile and channel of the full transform.
// The tile axis is the reduction axis: `nsplit` workgroups per xi share the tiles in contiguous ranges and write
// partial sums, which wino_wgrad_reduce_kernel adds in a fixed order and transforms to the 3x3 taps (deterministic).
//
// Stage = 16 tiles (2 tile rows x 8 tiles) = 4 k-steps of the MFMA.  Thread (channel quad cq = tid & 31, tile t = tid >> 5)
// loads its two patch rows x four columns and dY rows x two columns as 16-byte quads straight from global memory (a wave =
// two adjacent tiles x all 128 channels: 512-byte coalesced; one scalar base per image row + a per-lane offset per column,
// both clamped into the image; values from outside it are replaced by zero afterwards), transforms them in registers and
// stores one 16-byte quad per position: LDS image [position][tile][channel], rows of 144 floats, so that the MFMA operand
// fragments (lane -> channel l & 15 of a 16-block, tile l >> 4 of the k-step) are conflict-free ds_read_b32 (consecutive
// rows lie 16 banks apart) and the producer's ds_write_b128 likewise.  The rows of stage s + 2 are requested in the middle
// of stage s and consumed in the middle of stage s + 1; two LDS buffers, one barrier per stage.
#include "bmc_common.h"
#include "dma_ring.h"
#include "wgrad_k.h"
#include <stdlib.h>

#ifndef BMC_WW_PIPE
#define BMC_WW_PIPE 1     // 1: rows of stage s + 2 requested in the middle of stage s; 0: rows of stage s + 1 at its top
#endif
#ifndef BMC_WW_ABL
#define BMC_WW_ABL 0      // ablation builds (tools/): 1 no MFMA, 2 no global loads, 4 no transform / LDS stores, 8 no fragment reads
#endif

namespace {

constexpr int TROW = 144;              // floats per LDS row: one tile's 128 channels + 16 (consecutive rows 16 banks apart)
constexpr int POSF = 16 * TROW;        // one position's image [16 tiles][144]
constexpr int HALF = 4 * POSF;         // the dM (or V) images of the 4 positions of a stage: 36 KB
constexpr int BUFF = 2 * HALF;         // one stage: 72 KB

// A wave-uniform pointer, made opaque (readfirstlane) so that "uniform base + this lane's 32-bit offset" survives as the
// scalar-base form of global_load (the compiler otherwise reassociates base + lane + offset into per-lane 64-bit addresses).
__device__ __forceinline__ const char* uni(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// a - b on packed pairs: the compiler packs an f32x4 addition into two v_pk_add_f32 but a subtraction into four v_sub_f32
// (and folds b * -1 + a back into one), and VALU issue slots are what the transform costs beside the fp32 MFMAs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// xi is a template parameter: which rows are loaded and with which signs they are combined is decided at compile time (the
// kernel dispatches once on its workgroup's xi) -- straight-line load / transform code, no per-load branches.
template <int xi>
__device__ __forceinline__ void wgrad_body(const WgradK& a, float* const lds, const int split) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- producer role: channels 4 cq .. 4 cq + 3 of tile t of the stage (tile row t >> 3 of the pair, tile t & 7 of the 8);
    // a wave = two horizontally adjacent tiles (lanes 0-31 / 32-63) x all 32 channel quads: 512-byte coalesced 16-byte loads
    const int cq = tid & 31, hl = (tid >> 5) & 1, t = tid >> 5;
    const int tr = wave >> 2, tcw = 2 * (wave & 3);                  // tile row, first tile column of the wave (uniform)
    // row xi of B^T = +- row ra +- row rb of the patch: (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1)
    constexpr int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    // row xi of A: (1,0) (1,1) (1,-1) (0,-1) of dY's two rows.  The two -1 entries standing alone (row 3 here, column 3 in
    // produce()) are NOT applied: the partial sums of positions with xi = 3 or nu = 3 carry the opposite sign, which
    // wino_wgrad_reduce_kernel folds into its coefficients
    const int pst = t * TROW + 4 * cq;                               // + position * POSF (+ HALF for V)
    // ---- consumer role: wave = position nu, output-channel half ch; lane -> channel li of a 16-block, tile lk of a k-step
    const int nu = wave >> 1, ch = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int aoff = nu * POSF + lk * TROW + ch * 64 + li;           // + k-step * 4 * TROW + block * 16
    const int boff = HALF + nu * POSF + lk * TROW + li;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("" : "+a"(acc[i][n]));
    };
    pin_acc();

    const int per_img = a.SY * a.SX;
    const int st0 = (int)((long long)a.nstages * split / a.nsplit), st1 = (int)((long long)a.nstages * (split + 1) / a.nsplit);
    constexpr int PS = 128;                  // pixel stride of both operands (checked by the launcher)
    const long long rowb = (long long)a.W * PS * 4;     // bytes per image row

    // (image, tile-row pair, group of 8 tiles) of the next stage to load, advanced incrementally -- no divisions in the loop
    int nb = st0 / per_img, nsy, nsx;
    {
        const int r = st0 - nb * per_img;
        nsy = r / a.SX; nsx = r - nsy * a.SX;
    }
    nb = __builtin_amdgcn_readfirstlane(nb); nsy = __builtin_amdgcn_readfirstlane(nsy); nsx = __builtin_amdgcn_readfirstlane(nsx);
    // image nb of the launch -> its segment's operand pair (a handful of scalar compares, once per image)
    auto xptr = [&](const int b) __attribute__((always_inline)) {
        int s = 0;
        while (s + 1 < a.nseg && b >= a.segb[s + 1]) ++s;
        return uni(src_batch_ptr(a.x[s], b - a.segb[s]));
    };
    auto aptr = [&](const int b) __attribute__((always_inline)) {
        int s = 0;
        while (s + 1 < a.nseg && b >= a.segb[s + 1]) ++s;
        return uni(src_batch_ptr(a.a[s], b - a.segb[s]));
    };
    const char* xbase = xptr(nb);
    const char* abase = aptr(nb);

    f32x4 xa[4], xb[4], y0[2], y1[2];        // the two patch rows; dY row 0 (row 1 for xi = 3); dY row 1 (xi = 1, 2)
    // Requests the rows of the next stage.  Every address is clamped into the image (one scalar base per image row + one
    // per-lane byte offset per patch column; dY's two columns are patch columns 1 and 2); the return value says what of it
    // lies outside the image: bits 0 / 1 / 2 / 3 = patch row a, patch row b, dY row 0, dY row 1, bits 8.. = first patch column
    // of the wave + 1, bit 31 = some column of the wave is outside.  Applied in produce() (the loads stay one batch in flight).
    auto load = [&]() __attribute__((always_inline)) -> unsigned {
        unsigned zm = 0;
        if (BMC_WW_ABL & 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[q] = xb[q] = f32x4{1.f, 2.f, 3.f, 4.f};
            y0[0] = y0[1] = y1[0] = y1[1] = f32x4{1.f, 2.f, 3.f, 4.f};
        } else {
            const int ty = 2 * nsy + tr;
            const int ixw = 2 * (8 * nsx + tcw) - 1;                       // first patch column of the wave's first tile
            const int iya = 2 * ty - 1 + ra, iyb = 2 * ty - 1 + rb;        // the two patch rows of B^T's row xi
            const int oy0 = 2 * ty + (xi == 3 ? 1 : 0), oy1 = 2 * ty + 1;  // dY rows
            const char* const pa = uni(xbase + min(max(iya, 0), a.H - 1) * rowb);
            const char* const pb = uni(xbase + min(iyb, a.H - 1) * rowb);
            const char* const q0 = uni(abase + min(oy0, a.H - 1) * rowb);
            const char* const q1 = uni(abase + min(oy1, a.H - 1) * rowb);
            zm = (iya < 0 || iya >= a.H ? 1u : 0u) | (iyb >= a.H ? 2u : 0u) | (oy0 >= a.H ? 4u : 0u) | (oy1 >= a.H ? 8u : 0u) |
                 (ixw < 0 || ixw + 5 >= a.W ? 1u << 31 : 0u);
            if (zm) zm |= (unsigned)(ixw + 1) << 8;
            unsigned vo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) vo[q] = (unsigned)min(max(ixw + 2 * hl + q, 0), a.W - 1) * (PS * 4) + (unsigned)cq * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xa[q] = ldg16(pa + vo[q]);
                xb[q] = ldg16(pb + vo[q]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                y0[q] = ldg16(q0 + vo[q + 1]);
                if (xi == 1 || xi == 2) y1[q] = ldg16(q1 + vo[q + 1]);
            }
        }
        if (++nsx == a.SX) {
            nsx = 0;
            if (++nsy == a.SY) {
                nsy = 0; ++nb;
                if (nb < a.B) {
                    xbase = xptr(nb);
                    abase = aptr(nb);
                }
            }
        }
        return zm;
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto produce = [&](float* buf, const unsigned zm) __attribute__((always_inline)) {
        if (BMC_WW_ABL & 4) return;
        if (zm) {
            asm volatile("; image border" ::: "memory");      // (keeps this a branch: if-converted it costs selects in EVERY stage)
            const int col0 = (int)((zm >> 8) & 0x3fffff) - 1 + 2 * hl;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool out = col0 + q < 0 || col0 + q >= a.W;
                if (out || (zm & 1u)) xa[q] = z;
                if (out || (zm & 2u)) xb[q] = z;
                if (q == 1 || q == 2) {
                    if (out || (zm & 4u)) y0[q - 1] = z;
                    if (out || (zm & 8u)) y1[q - 1] = z;
                }
            }
        }
        f32x4 tq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tq[q] = xi == 1 ? xa[q] + xb[q] : (xi == 2 ? sub4(xb[q], xa[q]) : sub4(xa[q], xb[q]));
        float* const vp = buf + HALF + pst;
        *reinterpret_cast<f32x4*>(vp) = sub4(tq[0], tq[2]);
        *reinterpret_cast<f32x4*>(vp + POSF) = tq[1] + tq[2];
        *reinterpret_cast<f32x4*>(vp + 2 * POSF) = sub4(tq[2], tq[1]);
        *reinterpret_cast<f32x4*>(vp + 3 * POSF) = sub4(tq[1], tq[3]);
        f32x4 s0, s1;
        if (xi == 0 || xi == 3) { s0 = y0[0]; s1 = y0[1]; }
        else if (xi == 1) { s0 = y0[0] + y1[0]; s1 = y0[1] + y1[1]; }
        else { s0 = sub4(y0[0], y1[0]); s1 = sub4(y0[1], y1[1]); }
        const f32x4 m1 = s0 + s1;
        float* const mp = buf + pst;
        *reinterpret_cast<f32x4*>(mp) = s0;
        *reinterpret_cast<f32x4*>(mp + POSF) = m1;
        *reinterpret_cast<f32x4*>(mp + 2 * POSF) = sub4(s0, s1);
        *reinterpret_cast<f32x4*>(mp + 3 * POSF) = s1;
        if (xi == 1) bsum += m1;
    };

    // Software pipeline: the rows of stage s + 1 are loaded during stage s - 1 (issued right after the registers were
    // consumed), transformed and stored in the middle of stage s: a full stage (~5 us) of latency cover with one register set.
    unsigned zmn = 0;
    if (st0 < st1) {
        const unsigned zm = load();
        produce(lds, zm);
    }
    if (BMC_WW_PIPE && st0 + 1 < st1) zmn = load();
    ring_publish();
    int it = 0;
    for (int st = st0; st < st1; ++st, ++it) {
        const float* const cur = lds + (it & 1) * BUFF;
        float* const nxt = lds + ((it & 1) ^ 1) * BUFF;
        // 128 MFMAs in two halves of four ci-blocks each; k-step ks of a fragment = tile 4 ks + lk of the stage
        // (the dM fragments are read again for the second half: nothing of the first half stays live across produce())
        float af[4][4], bf[4][4];
        auto read_ab = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (BMC_WW_ABL & 8) { af[i][ks] = 1.f + i; bf[i][ks] = 2.f + ks; asm volatile("" : "+v"(af[i][ks]), "+v"(bf[i][ks])); continue; }
                    af[i][ks] = cur[aoff + ks * 4 * TROW + i * 16];
                    bf[i][ks] = cur[boff + ks * 4 * TROW + (4 * h + i) * 16];
                }
        };
        auto mfma64 = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (BMC_WW_ABL & 1) acc[i][4 * h + n][0] += af[i][ks] * bf[n][ks];
                        else acc[i][4 * h + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][ks], bf[n][ks], acc[i][4 * h + n], 0, 0, 0);
                    }
        };
        if (!BMC_WW_PIPE && st + 1 < st1) zmn = load();
        read_ab(0);
        mfma64(0);
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < st1) produce(nxt, zmn);
        if (BMC_WW_PIPE && st + 2 < st1) zmn = load();
        __builtin_amdgcn_sched_barrier(0);
        read_ab(1);
        mfma64(1);
        pin_acc();
        ring_publish();      // (raw barrier: the loads in flight are not drained)
    }

    // ---- partial sums: part[split][xi][nu][co][ci]
    float* const P = a.part + (((long long)split * 4 + xi) * 4 + nu) * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(ch * 64 + i * 16 + 4 * lk + r) * 128 + n * 16 + li] = acc[i][n][r];
    if (xi == 1 && a.bias_part) {        // bias partial of the workgroup: the 16 tile slots added through LDS (the loop ended on a barrier)
        *reinterpret_cast<f32x4*>(lds + t * 128 + 4 * cq) = bsum;
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += lds[k * 128 + tid];
            a.bias_part[(long long)split * 128 + tid] = sum;
        }
    }
}

__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WgradK a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BUFF];     // 144 KB: one workgroup per CU
    // workgroup -> (split, xi): the four xi of a split read the same pixels -- on the same XCD (same L2) when the split
    // count allows (consecutive workgroup ids go round the 8 XCDs)
    int xi, split;
    if ((a.nsplit & 7) == 0) {
        const int j = blockIdx.x >> 3;
        xi = j & 3; split = (j >> 2) * 8 + (blockIdx.x & 7);
    } else {
        xi = blockIdx.x & 3; split = blockIdx.x >> 2;
    }
    if (xi == 0) wgrad_body<0>(a, lds, split);
    else if (xi == 1) wgrad_body<1>(a, lds, split);
    else if (xi == 2) wgrad_body<2>(a, lds, split);
    else wgrad_body<3>(a, lds, split);
}

// dW[co][k0 + ci][3][3] (+)= G^T (sum over splits of dU) G, db[co] (+)= sum of the bias partials.
// Block = 32 consecutive ci of one co x 8 parts (part p adds splits p, p + 8, ...; then a fixed-order sum over the parts).
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ dw,
                                                               int ldw, int k0, int accumulate, const float* __restrict__ bias_part,
                                                               float* __restrict__ db) {
    __shared__ float red[8][16][32];
    __shared__ float tot[16][32];
    const int ol = threadIdx.x & 31, p8 = threadIdx.x >> 5;
    if ((int)blockIdx.y == 128) {        // bias: one block, 128 channels x 2 halves of the splits
        if (blockIdx.x != 0) return;
        __shared__ float bs[2][128];
        const int cc = threadIdx.x & 127, hf = threadIdx.x >> 7;
        float s = 0.f;
        for (int i = hf; i < nsplit; i += 2) s += bias_part[(long long)i * 128 + cc];
        bs[hf][cc] = s;
        __syncthreads();
        if (hf == 0) {
            const float v = bs[0][cc] + bs[1][cc];
            db[cc] = accumulate ? db[cc] + v : v;
        }
        return;
    }
    const int co = blockIdx.y, ci = blockIdx.x * 32 + ol;
    float u[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) u[p] = 0.f;
    for (int s = p8; s < nsplit; s += 8) {
        const float* const ps = part + (long long)s * 16 * 16384 + co * 128 + ci;
#pragma unroll
        for (int p = 0; p < 16; ++p) u[p] += ps[p * 16384];
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) red[p8][p][ol] = u[p];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = 2 * p8 + h;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][p][ol];
        tot[p][ol] = t;
    }
    __syncthreads();
    // rows of G: (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1);  dW[i][j] = sum_xi,nu G[xi][i] G[nu][j] dU[xi][nu]
    // (row 3 enters with -1: the main kernel leaves out the sign of A's rows / columns 3)
    for (int tap = p8; tap < 9; tap += 8) {
        const int i = tap / 3, jj = tap - 3 * i;
        float o = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float gx = x == 0 ? (i == 0 ? 1.f : 0.f) : x == 3 ? (i == 2 ? -1.f : 0.f) : (x == 2 && i == 1 ? -0.5f : 0.5f);
            float rowv = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float gn = n == 0 ? (jj == 0 ? 1.f : 0.f) : n == 3 ? (jj == 2 ? -1.f : 0.f) : (n == 2 && jj == 1 ? -0.5f : 0.5f);
                rowv += gn * tot[x * 4 + n][ol];
            }
            o += gx * rowv;
        }
        float* const d = dw + ((long long)co * ldw + k0 + ci) * 9 + tap;
        *d = accumulate ? *d + o : o;
    }
}

}  // namespace

extern "C" int bmc_wgrad_wino_nsplit(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const long long stages = (long long)B * (((H + 1) / 2 + 1) / 2) * (((W + 1) / 2 + 7) / 8);
    const int per_xi = bmc_num_cus() / 4 > 0 ? bmc_num_cus() / 4 : 1;
    // (fewer, longer splits for small problems -- every workgroup writes a full 1 MB set of partial sums -- were measured
    //  slower: 88.6 -> 96 ms per step at 31x56 with 12 stages per workgroup)
    return (int)(stages < per_xi ? stages : per_xi);
}

extern "C" int bmc_wgrad_wino_multi(const bmc_src_t* dy, const bmc_src_t* x, const int* batches, int nseg, int H, int W, int nsplit,
                                    float* part, float* bias_part, bmc_stream_t s) {
    BMC_CHECK_ARG(dy && x && batches && part, "bmc_wgrad_wino: null argument");
    BMC_CHECK_ARG(nseg >= 1 && nseg <= BMC_WG_MAXSEG, "bmc_wgrad_wino: 1 .. %d operand pairs per launch (got %d)", BMC_WG_MAXSEG, nseg);
    WgradK k;
    k.nseg = nseg;
    k.segb[0] = 0;
    for (int i = 0; i < nseg; ++i) {
        BMC_CHECK_ARG(dy[i].ptr && x[i].ptr && batches[i] >= 1, "bmc_wgrad_wino: null operand / empty segment %d", i);
        BMC_CHECK_ARG(dy[i].nch == 128 && x[i].nch == 128, "bmc_wgrad_wino: both operands must be 128-channel windows (got %d, %d)",
                      dy[i].nch, x[i].nch);
        BMC_CHECK_ARG(dy[i].pix_stride == 128 && x[i].pix_stride == 128, "bmc_wgrad_wino: both operands must be dense in the channel "
                      "axis (pix_stride 128; got %d, %d)", dy[i].pix_stride, x[i].pix_stride);
        k.a[i] = to_dev(dy[i]); k.x[i] = to_dev(x[i]);
        k.segb[i + 1] = k.segb[i] + batches[i];
    }
    for (int i = nseg; i < BMC_WG_MAXSEG; ++i) { k.a[i] = k.a[0]; k.x[i] = k.x[0]; k.segb[i + 1] = k.segb[nseg]; }
    const int B = k.segb[nseg];
    BMC_CHECK_ARG(H >= 1 && W >= 1 && (long long)H * W * 128 < (1ll << 29), "bmc_wgrad_wino: bad geometry");
    k.B = B; k.H = H; k.W = W;
    k.SY = ((H + 1) / 2 + 1) / 2; k.SX = ((W + 1) / 2 + 7) / 8;
    const long long stages = (long long)B * k.SY * k.SX;
    BMC_CHECK_ARG(stages < (1ll << 31), "bmc_wgrad_wino: too many tiles");
    BMC_CHECK_ARG(nsplit >= 1 && nsplit <= stages, "bmc_wgrad_wino: nsplit must be in [1, %lld]", stages);
    k.nstages = (int)stages; k.nsplit = nsplit;
    k.part = part; k.bias_part = bias_part;
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3((unsigned)nsplit * 4), dim3(512), 0, (hipStream_t)s, k);
    BMC_CHECK_LAUNCH("bmc_wgrad_wino");
    return 0;
}

extern "C" int bmc_wgrad_wino(const bmc_src_t* dy, const bmc_src_t* x, int B, int H, int W, int nsplit, float* part,
                              float* bias_part, bmc_stream_t s) {
    BMC_CHECK_ARG(dy && x && B >= 1, "bmc_wgrad_wino: null argument");
    return bmc_wgrad_wino_multi(dy, x, &B, 1, H, W, nsplit, part, bias_part, s);
}

extern "C" int bmc_wgrad_wino_reduce(const float* part, int nsplit, float* dw, int ldw, int k0, int accumulate,
                                     const float* bias_part, float* db, bmc_stream_t s) {
    BMC_CHECK_ARG(part && dw && nsplit >= 1, "bmc_wgrad_wino_reduce: bad arguments");
    BMC_CHECK_ARG(ldw >= 128 && k0 >= 0 && k0 + 128 <= ldw, "bmc_wgrad_wino_reduce: columns [k0, k0 + 128) must lie inside the %d input "
                  "channels of the weight tensor", ldw);
    BMC_CHECK_ARG((bias_part == nullptr) == (db == nullptr), "bmc_wgrad_wino_reduce: bias_part and db go together");
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(4, bias_part ? 129 : 128), dim3(256), 0, (hipStream_t)s, part, nsplit, dw, ldw, k0,
                       accumulate, bias_part, db);
    BMC_CHECK_LAUNCH("bmc_wgrad_wino_reduce");
    return 0;
}
