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
// Stage = 16 tiles (2 tile rows x 8 tiles) = 4 k-steps of the MFMA.  Thread (channel c = tid & 127, tile quad tg = tid >> 7)
// loads its rows of X / dY as scalars straight from global memory (a wave = 64 consecutive channels of one pixel: 256-byte
// coalesced; one scalar base per image row + a per-lane offset per column, both clamped into the image; values from outside
// it are replaced by zero afterwards), transforms four consecutive tiles in registers and stores one 16-byte quad per position:
// LDS image [position][channel][16 tiles], rows of 64 B, quads XOR-swizzled as in dma_ring.h -- the MFMA operand
// fragments (lane -> channel l & 15, quad l >> 4, the quad's four floats = the four k-steps) are conflict-free
// ds_read_b128 and the producer's ds_write_b128 likewise.  The loads of stage s + 1 are issued before the 128 MFMAs of
// stage s and consumed in their middle; two LDS buffers, one barrier per stage.
#include "bmc_common.h"
#include "dma_ring.h"
#include "wgrad_k.h"

#ifndef BMC_WW_PIPE
#define BMC_WW_PIPE 1     // 1: rows of stage s + 2 requested in the middle of stage s; 0: rows of stage s + 1 at its top
#endif
#ifndef BMC_WW_ABL
#define BMC_WW_ABL 0      // ablation builds (tools/): 1 no MFMA, 2 no global loads, 4 no transform / LDS stores, 8 no fragment reads
#endif

namespace {

constexpr int ROWF = 16;               // floats per LDS row: one channel's 16 tiles of a stage
constexpr int POSF = 128 * ROWF;       // one position's image [128 channels][16 tiles]
constexpr int HALF = 4 * POSF;         // the dM (or V) images of the 4 positions of a stage: 32 KB
constexpr int BUFF = 2 * HALF;         // one stage: 64 KB

// A wave-uniform pointer, made opaque (readfirstlane) so that "uniform base + this lane's 32-bit offset" survives as the
// scalar-base form of global_load (the compiler otherwise reassociates base + lane + offset into per-lane 64-bit addresses).
__device__ __forceinline__ const float* uni(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

// xi is a template parameter: which rows are loaded and with which signs they are combined is decided at compile time (the
// kernel dispatches once on its workgroup's xi) -- straight-line load / transform code, no per-load branches.
template <int xi>
__device__ __forceinline__ void wgrad_body(const WgradK& a, float* const lds, const int split) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- producer role: channel c, tiles 4 tg .. 4 tg + 3 of the stage (tile row tg >> 1, tiles 4 (tg & 1) .. + 3 of its 8)
    const int c = tid & 127;
    const int tg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int tr = tg >> 1, xq = tg & 1;
    // row xi of B^T = sa * row ra + sb * row rb of the patch: (1,0,-1,0) (0,1,1,0) (0,-1,1,0) (0,1,0,-1)
    constexpr int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    // row xi of A: (1,0) (1,1) (1,-1) (0,-1) of dY's two rows.  The two -1 entries standing alone (row 3 here, column 3 in
    // produce()) are NOT applied: the partial sums of positions with xi = 3 or nu = 3 carry the opposite sign, which
    // wino_wgrad_reduce_kernel folds into its coefficients (12 VALU instructions less per thread and stage)
    const int pst = c * ROWF + ((tg ^ swz(c)) << 2);                 // + position * POSF (+ HALF for V)
    // ---- consumer role
    const int nu = wave >> 1, ch = wave & 1;
    const int li = lane & 15, lk = lane >> 4;
    const int qsw = (lk ^ swz(li)) << 2;
    const int aoff = nu * POSF + (ch * 64 + li) * ROWF + qsw;        // + cbk * 16 * ROWF
    const int boff = HALF + nu * POSF + li * ROWF + qsw;             // + nb * 16 * ROWF

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
    constexpr int PS = 128;                  // pixel stride of both operands (checked by the launcher): column steps are immediates
    const long long rowf = (long long)a.W * PS;

    // (image, tile-row pair, group of 8 tiles) of the next stage to load, advanced incrementally -- no divisions in the loop
    int nb = st0 / per_img, nsy, nsx;
    {
        const int r = st0 - nb * per_img;
        nsy = r / a.SX; nsx = r - nsy * a.SX;
    }
    nb = __builtin_amdgcn_readfirstlane(nb); nsy = __builtin_amdgcn_readfirstlane(nsy); nsx = __builtin_amdgcn_readfirstlane(nsx);
    const float* xbase = uni(src_batch_ptr(a.x, nb));
    const float* abase = uni(src_batch_ptr(a.a, nb));

    float xa[10], xb[10], y0[8], y1[8];      // (y0: dY row 0, or row 1 for xi = 3; y1: row 1 for xi = 1, 2)
    // -> 0 for an interior stage; at the image border the returned bits name what lies outside the image (bits 0-9: patch
    // columns, 10 / 11: patch rows, 12-19: dY columns, 20 / 21: dY rows) -- applied in produce(), so that the loads of a
    // stage stay one batch in flight
    auto load = [&]() __attribute__((always_inline)) -> unsigned {
        unsigned zm = 0;
        if (BMC_WW_ABL & 2) {
#pragma unroll
            for (int q = 0; q < 10; ++q) { xa[q] = 1.f; xb[q] = 2.f; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { y0[q] = 1.f; y1[q] = 2.f; }
        } else {
            const int ty = 2 * nsy + tr, tx0 = 8 * nsx + 4 * xq;
            const int ix0 = 2 * tx0 - 1;                                   // patch columns ix0 .. ix0 + 9, dY columns ix0 + 1 .. ix0 + 8
            const int iya = 2 * ty - 1 + ra, iyb = 2 * ty - 1 + rb;        // the two patch rows of B^T's row xi
            const int oy0 = 2 * ty + (xi == 3 ? 1 : 0), oy1 = 2 * ty + 1;  // dY rows
            // one scalar base per image row (clamped into the image) + one per-lane 32-bit offset per column (clamped likewise;
            // dY's columns are patch columns 1 .. 8): every address is valid, what lies outside the image is named in zm
            const float* const pa = uni(xbase + min(max(iya, 0), a.H - 1) * rowf);
            const float* const pb = uni(xbase + min(iyb, a.H - 1) * rowf);
            const float* const q0 = uni(abase + min(oy0, a.H - 1) * rowf);
            const float* const q1 = uni(abase + min(oy1, a.H - 1) * rowf);
            zm = (iya < 0 || iya >= a.H ? 1u << 10 : 0u) | (iyb >= a.H ? 1u << 11 : 0u) | (oy0 >= a.H ? 1u << 20 : 0u) |
                 (oy1 >= a.H ? 1u << 21 : 0u);
            unsigned vo[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int ix = ix0 + q;
                if (ix < 0 || ix >= a.W) zm |= (1u << q) | (q >= 1 && q <= 8 ? 1u << (11 + q) : 0u);
                vo[q] = ((unsigned)min(max(ix, 0), a.W - 1) * PS + (unsigned)c) * 4u;      // bytes
            }
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                xa[q] = ldg4(reinterpret_cast<const char*>(pa) + vo[q]);
                xb[q] = ldg4(reinterpret_cast<const char*>(pb) + vo[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                y0[q] = ldg4(reinterpret_cast<const char*>(q0) + vo[q + 1]);
                if (xi == 1 || xi == 2) y1[q] = ldg4(reinterpret_cast<const char*>(q1) + vo[q + 1]);
            }
        }
        if (++nsx == a.SX) {
            nsx = 0;
            if (++nsy == a.SY) {
                nsy = 0; ++nb;
                xbase = uni(src_batch_ptr(a.x, nb));
                abase = uni(src_batch_ptr(a.a, nb));
            }
        }
        return zm;
    };
    float bsum = 0.f;
    auto produce = [&](float* buf, const unsigned zm) __attribute__((always_inline)) {
        if (BMC_WW_ABL & 4) return;
        if (zm) {
            asm volatile("; image border" ::: "memory");      // (keeps this a branch: if-converted it costs 28 selects in EVERY stage)
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                if (zm & (1u << q | 1u << 10)) xa[q] = 0.f;
                if (zm & (1u << q | 1u << 11)) xb[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (zm & (1u << (12 + q) | 1u << 20)) y0[q] = 0.f;
                if (zm & (1u << (12 + q) | 1u << 21)) y1[q] = 0.f;
            }
        }
        float t[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) t[q] = xi == 1 ? xa[q] + xb[q] : (xi == 2 ? xb[q] - xa[q] : xa[q] - xb[q]);
        f32x4 v0, v1, v2, v3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = t[2 * j] - t[2 * j + 2];
            v1[j] = t[2 * j + 1] + t[2 * j + 2];
            v2[j] = t[2 * j + 2] - t[2 * j + 1];
            v3[j] = t[2 * j + 1] - t[2 * j + 3];
        }
        float* const vp = buf + HALF + pst;
        *reinterpret_cast<f32x4*>(vp) = v0;
        *reinterpret_cast<f32x4*>(vp + POSF) = v1;
        *reinterpret_cast<f32x4*>(vp + 2 * POSF) = v2;
        *reinterpret_cast<f32x4*>(vp + 3 * POSF) = v3;
        float s[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = (xi == 0 || xi == 3) ? y0[q] : (xi == 1 ? y0[q] + y1[q] : y0[q] - y1[q]);
        f32x4 m0, m1, m2, m3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m0[j] = s[2 * j];
            m1[j] = s[2 * j] + s[2 * j + 1];
            m2[j] = s[2 * j] - s[2 * j + 1];
            m3[j] = s[2 * j + 1];
        }
        float* const mp = buf + pst;
        *reinterpret_cast<f32x4*>(mp) = m0;
        *reinterpret_cast<f32x4*>(mp + POSF) = m1;
        *reinterpret_cast<f32x4*>(mp + 2 * POSF) = m2;
        *reinterpret_cast<f32x4*>(mp + 3 * POSF) = m3;
        if (xi == 1) bsum += (m1[0] + m1[1]) + (m1[2] + m1[3]);
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
        // 128 MFMAs in two halves of four ci-blocks each (12 fragment quads live at a time instead of all 12 + the loaded rows)
        // (the dM fragments are read again for the second half: nothing of the first half stays live across produce())
        f32x4 af[4], bf[4];
        auto read_b = [&](int h) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BMC_WW_ABL & 8) { af[i] = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(af[i])); continue; }
                af[i] = *reinterpret_cast<const f32x4*>(cur + aoff + i * 16 * ROWF);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (BMC_WW_ABL & 8) { bf[n] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(bf[n])); continue; }
                bf[n] = *reinterpret_cast<const f32x4*>(cur + boff + (4 * h + n) * 16 * ROWF);
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
        read_b(0);
        mfma64(0);
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < st1) produce(nxt, zmn);
        if (BMC_WW_PIPE && st + 2 < st1) zmn = load();
        __builtin_amdgcn_sched_barrier(0);
        read_b(1);
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
    if (xi == 1 && a.bias_part) a.bias_part[((long long)split * 4 + tg) * 128 + c] = bsum;
}

__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WgradK a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BUFF];     // 128 KB: one workgroup per CU
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
    if ((int)blockIdx.y == 128) {        // bias: one block, 128 channels x 2 halves of the (split, quad) rows
        if (blockIdx.x != 0) return;
        __shared__ float bs[2][128];
        const int cc = threadIdx.x & 127, hf = threadIdx.x >> 7;
        float s = 0.f;
        for (int i = hf; i < nsplit * 4; i += 2) s += bias_part[(long long)i * 128 + cc];
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
    return (int)(stages < per_xi ? stages : per_xi);
}

extern "C" int bmc_wgrad_wino(const bmc_src_t* dy, const bmc_src_t* x, int B, int H, int W, int nsplit, float* part,
                              float* bias_part, bmc_stream_t s) {
    BMC_CHECK_ARG(dy && x && part && dy->ptr && x->ptr, "bmc_wgrad_wino: null argument");
    BMC_CHECK_ARG(dy->nch == 128 && x->nch == 128, "bmc_wgrad_wino: both operands must be 128-channel windows (got %d, %d)", dy->nch,
                  x->nch);
    BMC_CHECK_ARG(dy->pix_stride == 128 && x->pix_stride == 128, "bmc_wgrad_wino: both operands must be dense in the channel axis "
                  "(pix_stride 128; got %d, %d)", dy->pix_stride, x->pix_stride);
    BMC_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && (long long)H * W * 128 < (1ll << 29), "bmc_wgrad_wino: bad geometry");
    WgradK k;
    k.a = to_dev(*dy); k.x = to_dev(*x);
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
