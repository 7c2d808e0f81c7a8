// Pixel-reduction GEMM on the fp32 matrix cores:
//   C[g][tap][m][n] = sum_{b in group g} sum_{y,x} A[b,y,x,m] * cat(src)[b, y+dy, x+dx, n]
// = the weight gradient of a 3x3 / 1x1 convolution (A = dY, src = the conv inputs),
// and the channel Gram matrix center . v^T of the BIE block (and its backward).
//
// The reduction axis is the pixel axis, so both MFMA operands are read "k-major" straight from NHWC tiles in LDS
// (lane i -> channel i: conflict-free ds_read_b32); the 9 taps re-use one A fragment and read the X halo tile at 9
// constant LDS offsets.  Pixel tiles (4x16 / 64 flat pixels) are dealt round-robin to `nsplit` splits; partial sums
// go to slabs and are summed by a second kernel in a fixed order (deterministic, no float atomics).
#include "bmc_common.h"
#include "pgemm_k.h"
#include "dma_ring.h"

namespace {


// LDS-DMA pixel-reduction GEMM: ONE 8-wave workgroup per CU.
//   TAPS = 9: 128 rows x 64 columns x 9 taps per workgroup (wave = 32 x 32 x 9 taps, 144 accumulator registers);
//   TAPS = 1: 128 rows x 128 columns              (wave = 32 x 64, 32 accumulator registers).
// Both operand tiles are double-buffered in LDS and filled by global_load_lds_dwordx4 (no staging registers, no
// ds_write): the fill of pixel tile t+1 is in flight under the MFMAs of tile t, one barrier per tile.  The LDS images
// are lane-linear ([px][128] for A, [halo px][64] or [px][128] for X); pixels outside the image and channels beyond
// M / N are sourced from a small zero buffer.
// TG = taps per workgroup: 9, or 3 (one tap ROW per workgroup, a.tap_groups = 3): small problems, where 256 workgroups
// are only reached by cutting the pixel axis into ~128 splits and every split writes a whole [taps][M][N] slab -- at
// 31x56 the slab writes took as long as the MFMAs, and the reduction read 75 MB per weight gradient.  Three times the
// workgroups per split = a third of the splits = a third of the slab bytes, for three times the (cheap) tile fills.
template <int TAPS, int TG = TAPS, bool TAB = false>
__global__ __launch_bounds__(512, 2) void pgemm_dma_kernel(const PgemmK a) {
    constexpr int HWD = PT_W + 2, HHT = PT_H + 2;
    constexpr int NHALO = TAPS == 9 ? HWD * HHT : PT;                   // 108 halo pixels or 64 pixels
    constexpr int XCH = TAPS == 9 ? 64 : 128;                           // columns per workgroup
    constexpr int NT = TAPS == 9 ? 1 : 2;                               // 32-column tiles per wave
    constexpr int XSH = TAPS == 9 ? 4 : 5;                              // log2(float4 per X row)
    constexpr int BUF = 2048 * 4;                                       // floats per LDS buffer (2048 float4)
    __shared__ __attribute__((aligned(16))) float lds[4 * BUF];         // A[2], X[2]  (128 KB)
    // the source table, read with a per-lane index by the boundary-tile path: from LDS (indexing the by-value kernel
    // argument made the compiler keep a copy of it in scratch memory)
    __shared__ SrcDev tab[BMC_MAX_SRC];
    const int tid = threadIdx.x, lane = tid & 63;
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int mw = wave & 3, nw = wave >> 2;

    int bid = blockIdx.x;
    const int split = bid % a.nsplit; bid /= a.nsplit;
    const int tg = TG == TAPS ? 0 : bid % (TAPS / TG);      // tap row of this workgroup
    if (TG != TAPS) bid /= TAPS / TG;
    const int nb = bid % a.n_nblk; bid /= a.n_nblk;
    const int mb = bid % a.n_mblk;
    const int g = bid / a.n_mblk;
    const int m0 = mb * 128, n0 = nb * XCH;
    const bool wave_active = m0 + 32 * mw < a.Mpad;
    const int HWp = a.H * a.W;

    f32x16 acc[TG * NT];
#pragma unroll
    for (int t = 0; t < TG * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ntiles = a.batch_per_group * a.tiles_per_img;
    // ---- tile fill by LDS-DMA.  Instructions issued beside the MFMAs cost matrix-pipe time about 1:1 (in-kernel stamps:
    // the fill of one tile took 4 000 cycles per wave = 9 % of a 3x3 tile, 21 % of a 1x1 tile, nearly all of it address
    // arithmetic), so everything that does not depend on the tile is computed ONCE per piece here -- byte offset from
    // the tile's first (halo) pixel, pixel coordinates inside the tile -- and interior tiles whose channels all exist in
    // ONE source are filled with "uniform base (SGPR pair) + per-lane 32-bit offset" DMA: no VALU per piece.
    // the source that holds this workgroup's block of columns [n0, n0 + XCH) -- if one source holds all of it
    int xsi = 0, xch0 = n0;
#pragma unroll
    for (int si = 1; si < BMC_MAX_SRC; ++si)
        if (xsi == si - 1 && xch0 >= tab[si - 1].nch && si < a.nsrc) { xch0 -= tab[si - 1].nch; xsi = si; }
    const SrcDev xs = tab[xsi];
    const bool x_one_src = xch0 + XCH <= xs.nch;
    int a_off[4], x_off[4], a_yx[4], x_yx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = i * 512 + tid, p = e >> 5, c4 = (e & 31) * 4;
        a_yx[i] = TAPS == 9 ? ((p >> 4) << 16) | (p & 15) : p;
        a_off[i] = ((TAPS == 9 ? (p >> 4) * a.W + (p & 15) : p) * a.a.pix_stride + m0 + c4) * 4;
        const int hp = e >> XSH, xc4 = (e & ((1 << XSH) - 1)) * 4;
        const int hy = TAPS == 9 ? hp / HWD : 0, hx = TAPS == 9 ? hp - hy * HWD : hp;
        x_yx[i] = TAPS == 9 ? (hy << 16) | hx : hp;
        // lanes beyond the halo (3x3: hp >= NHALO) read the tile's first pixel; their LDS rows are never used
        x_off[i] = ((TAPS == 9 ? (hp < NHALO ? hy * a.W + hx : 0) : hp) * xs.pix_stride + xch0 + xc4) * 4;
    }
    const bool all_ch = m0 + 128 <= a.M && n0 + XCH <= a.N && x_one_src &&
                        (long long)a.H * a.W * (a.a.pix_stride > xs.pix_stride ? a.a.pix_stride : xs.pix_stride) < (1ll << 28);
    // (image, tile row, tile column) of the next tile to fill, advanced incrementally (tiles are visited in order)
    const int step_img = a.nsplit / a.tiles_per_img, step_rem = a.nsplit - step_img * a.tiles_per_img;
    const int step_ty = TAPS == 9 ? step_rem / a.tiles_x : 0, step_tx = TAPS == 9 ? step_rem - step_ty * a.tiles_x : step_rem;
    int nx_bb = split / a.tiles_per_img, nx_ty, nx_tx;
    {
        const int tin = split - nx_bb * a.tiles_per_img;
        nx_ty = TAPS == 9 ? tin / a.tiles_x : 0; nx_tx = TAPS == 9 ? tin - nx_ty * a.tiles_x : tin;
    }
    auto dma = [&](const void* sbase, unsigned voff, float* ldst) {
        const unsigned long long pv = reinterpret_cast<unsigned long long>(sbase);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
        const void* const sb = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
        const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)ldst);
        // (s_nop 4: wait states between the VALU-written SGPRs / m0 and the VMEM instruction; inline asm is opaque to the
        //  hazard recognizer.  m0 is reserved and cannot be named as a clobber; nothing else in this kernel uses it.)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(la) : "memory");
    };
    auto issue = [&](int tile, int buf) {      // must be called for tiles split, split + nsplit, ... in order
        const int b = g * a.batch_per_group + nx_bb;
        int y0 = 0, x0 = 0, p0 = 0;
        if (TAPS == 9) { y0 = nx_ty * PT_H; x0 = nx_tx * PT_W; }
        else p0 = nx_tx * PT;
        nx_bb += step_img; nx_ty += step_ty; nx_tx += step_tx;
        if (TAPS == 9) {
            if (nx_tx >= a.tiles_x) { nx_tx -= a.tiles_x; ++nx_ty; }
            if (nx_ty >= a.tiles_y) { nx_ty -= a.tiles_y; ++nx_bb; }
        } else if (nx_tx >= a.tiles_per_img) { nx_tx -= a.tiles_per_img; ++nx_bb; }
        const float* ab = src_bp<TAB>(a.a, b);
        const bool interior = TAPS == 9 ? (y0 >= 1 && x0 >= 1 && y0 + PT_H + 1 <= a.H && x0 + PT_W + 1 <= a.W) : (p0 + PT <= HWp);
        if (interior && all_ch) {
            const float* const abt = ab + (TAPS == 9 ? (long long)y0 * a.W + x0 : (long long)p0) * a.a.pix_stride;
            const float* const xbt = src_bp<TAB>(xs, b) +
                                     (TAPS == 9 ? (long long)(y0 - 1) * a.W + (x0 - 1) : (long long)p0) * xs.pix_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma(abt, (unsigned)a_off[i], lds + buf * BUF + (i * 512 + wave * 64) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma(xbt, (unsigned)x_off[i], lds + (2 + buf) * BUF + (i * 512 + wave * 64) * 4);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {            // A tile: [64 px][128 ch]
            const int e = i * 512 + tid, p = e >> 5, c4 = (e & 31) * 4;
            long long pix;
            bool ok;
            if (TAPS == 9) {
                const int y = y0 + (p >> 4), x = x0 + (p & 15);
                ok = y < a.H && x < a.W;
                pix = (long long)y * a.W + x;
            } else {
                pix = p0 + p;
                ok = pix < HWp;
            }
            const float* src = a.zeros;
            if (ok && m0 + c4 < a.M) src = ab + pix * a.a.pix_stride + m0 + c4;
            dma16v(src, (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + buf * BUF + (i * 512 + wave * 64) * 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {            // X tile: [108 halo px][64 ch] (+ padding lanes) or [64 px][128 ch]
            const int e = i * 512 + tid, hp = e >> XSH, c4 = (e & ((1 << XSH) - 1)) * 4;
            long long pix;
            bool ok;
            if (TAPS == 9) {
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                ok = hp < NHALO && y >= 0 && y < a.H && x >= 0 && x < a.W;
                pix = (long long)y * a.W + x;
            } else {
                pix = p0 + hp;
                ok = pix < HWp;
            }
            int ch = n0 + c4;
            const float* src = a.zeros;
            if (ok && ch < a.N) {
                int s_i = 0;
#pragma unroll
                for (int si = 1; si < BMC_MAX_SRC; ++si)
                    if (s_i == si - 1 && ch >= tab[si - 1].nch && si < a.nsrc) { ch -= tab[si - 1].nch; s_i = si; }
                const SrcDev S = tab[s_i];
                src = src_bp<TAB>(S, b) + pix * S.pix_stride + ch;
            }
            dma16v(src, (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (2 + buf) * BUF + (i * 512 + wave * 64) * 4));
        }
    };

    if (split < ntiles) issue(split, 0);
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // vmcnt(0): the asm DMA is invisible to the compiler's own waits
    __syncthreads();
    // bias gradient = column sums of A over pixels: the first n-block adds up its A tiles straight from LDS
    // (thread -> channel tid & 127, rows (tid >> 7) * 16 .. + 16); 4 partial rows per workgroup go to bias_slabs
    const bool do_bias = a.bias_slabs != nullptr && nb == 0 && tg == 0;
    float bsum = 0.f;
    int it = 0;
    for (int tile = split; tile < ntiles; tile += a.nsplit, ++it) {
        const int cur = it & 1;
        const int next = tile + a.nsplit;
        if (next < ntiles) issue(next, cur ^ 1);
        if (do_bias) {
            const float* const bp = lds + cur * BUF + (tid >> 7) * 16 * 128 + (tid & 127);
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += bp[r * 128];
        }
        if (wave_active) {
            const float* const ap = lds + cur * BUF + lh * 128 + 32 * mw + li;
            const float* const xp = lds + (2 + cur) * BUF + lh * XCH + 32 * NT * nw + li + (TG == TAPS ? 0 : tg * HWD * XCH);
            if constexpr (TAPS == 1) {
                // 1x1: two MFMAs per pair of operand reads.  Left to itself the compiler issues each read right in front of its
                // MFMAs and waits for it (lgkmcnt(0) before every pair: an LDS round trip per 128 matrix-pipe cycles, 0.71 of the
                // MFMA rate with nothing else going on).  Operands are read a batch of QB k-steps AHEAD into a second register
                // set; the scheduling barriers keep "reads of batch b + 1, then the MFMAs of batch b" in that order.
                constexpr int QB = 4, NBAT = PT / 2 / QB;
                float av[2][QB], xv[2][QB][NT];
                auto rd = [&](const int slot, const int q0) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        av[slot][j] = ap[2 * (q0 + j) * 128];
#pragma unroll
                        for (int u = 0; u < NT; ++u) xv[slot][j][u] = xp[2 * (q0 + j) * XCH + 32 * u];
                    }
                };
                rd(0, 0);
#pragma unroll
                for (int qb = 0; qb < NBAT; ++qb) {
                    if (qb + 1 < NBAT) rd((qb + 1) & 1, (qb + 1) * QB);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < QB; ++j)
#pragma unroll
                        for (int u = 0; u < NT; ++u)
                            acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[qb & 1][j], xv[qb & 1][j][u], acc[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int q = 0; q < PT / 2; ++q) {
                const float av = ap[2 * q * 128];
#pragma unroll
                for (int tap = 0; tap < TG; ++tap)       // (TG = 3: tap row tg is folded into xp)
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        int off;
                        if (TAPS == 9) off = (((q >> 3) + tap / 3) * HWD + 2 * (q & 7) + tap % 3) * XCH;
                        else off = 2 * q * XCH + 32 * u;
                        acc[tap * NT + u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xp[off], acc[tap * NT + u], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));   // vmcnt(0): the next tile's DMA has landed
        __syncthreads();    // publishes it and fences this tile's LDS reads
    }

    if (do_bias && m0 + (tid & 127) < a.Mpad)
        a.bias_slabs[(((long long)split * a.G + g) * 4 + (tid >> 7)) * a.Mpad + m0 + (tid & 127)] = bsum;
    if (wave_active) {
        float* const sl = a.slabs + (((long long)split * a.G + g) * TAPS + (TG == TAPS ? 0 : tg * TG)) * a.Mpad * a.Npad;
#pragma unroll
        for (int tap = 0; tap < TG; ++tap)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int n = n0 + 32 * NT * nw + 32 * u + li;
                if (n < a.Npad) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + 32 * mw + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        sl[((long long)tap * a.Mpad + m) * a.Npad + n] = acc[tap * NT + u][r];
                    }
                }
            }
    }
}

// Slab reductions: 32 outputs x 8 split-parts per 256-thread block (each part sums every 8th split, then a fixed
// order LDS tree) -- deterministic, and short even when nsplit is in the hundreds.
__device__ __forceinline__ float split_sum(const float* p, long long slab, int nsplit, float* red) {
    const int part = threadIdx.x >> 5, ol = threadIdx.x & 31;
    float s = 0.f;
    if (p)
        for (int i = part; i < nsplit; i += 8) s += p[i * slab];
    red[part * 32 + ol] = s;
    __syncthreads();
    float t = 0.f;
    if (part == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + ol];
    }
    __syncthreads();
    return t;
}

// Per-group destinations (weights stacked over groups for ONE launch but owned by separate parameters -- v1 / v2,
// conv_hp / conv_hn): group g is written to gd.dw[g] / gd.db[g] instead of dw + g * M*Cin*taps / db + g * M.
struct GroupDst { float* dw[4]; float* db[4]; int n; };

// Bias gradient db[g][m] = sum over the nsplit * 4 column-sum partials the GEMM's workgroups left in bias_slabs (partial
// i = split * 4 + part lives at ((split * G + g) * 4 + part) * Mpad + m).  One block = BIAS_OUT outputs x 64 parts: a thread
// adds every 64th partial (eight loads in flight), the 64 part sums of an output meet in LDS and are added in part order -- a
// fixed order whatever the launch.  (32 outputs x 8 parts, the first form, left each thread with nsplit / 2 dependent rounds of
// loads: with 224-256 splits these few blocks ran 14-37 us and were the whole duration of a reduction whose weight part takes
// 5 us -- 715 launches per small-frame step.)
constexpr int BIAS_OUT = 4;
__device__ __forceinline__ void bias_block_sum(const float* bias_slabs, int nsplit, int G, int M, int Mpad, float* db,
                                               int accumulate, int block, const GroupDst& gd, float* red /* [256] */) {
    const long long total = (long long)G * M;
    const long long idx = (long long)block * BIAS_OUT + (threadIdx.x & (BIAS_OUT - 1));
    const bool live = idx < total;
    const int g = live ? (int)(idx / M) : 0, m = live ? (int)(idx % M) : 0;
    const int part = threadIdx.x / BIAS_OUT, nparts = 256 / BIAS_OUT, n4 = nsplit * 4;
    float s = 0.f;
    if (live) {
        auto at = [&](int i) { return ldg4(bias_slabs + (((long long)(i >> 2) * G + g) * 4 + (i & 3)) * Mpad + m); };
        int i = part;
        for (; i + 7 * nparts < n4; i += 8 * nparts) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = at(i + k * nparts);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; i < n4; i += nparts) s += at(i);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < BIAS_OUT && live) {
        float t = 0.f;
        for (int k = 0; k < nparts; ++k) t += red[k * BIAS_OUT + threadIdx.x];
        float* o = gd.n ? gd.db[g] + m : db + idx;
        *o = accumulate ? *o + t : t;
    }
}

__global__ void reduce_weight_kernel(const float* slabs, int nsplit, int G, int taps, int M, int N, int Mpad, int Npad,
                                     const int* kmap, int Cin, float* dw, int accumulate, const float* bias_slabs,
                                     float* db, int wblocks, const GroupDst gd) {
    __shared__ float red[256];
    if ((int)blockIdx.x >= wblocks) {    // trailing blocks: bias gradient
        bias_block_sum(bias_slabs, nsplit, G, M, Mpad, db, accumulate, (int)blockIdx.x - wblocks, gd, red);
        return;
    }
    const long long total = (long long)G * taps * M * N;
    const long long slab = (long long)G * taps * Mpad * Npad;
    const long long nchunk = (total + 31) / 32;
    for (long long chunk = blockIdx.x; chunk < nchunk; chunk += wblocks) {
        const long long idx = chunk * 32 + (threadIdx.x & 31);
        const float* p = nullptr;
        int m = 0, tap = 0, g = 0, ci = -1;
        if (idx < total) {
            const int n = idx % N;
            long long r = idx / N;
            m = r % M; r /= M;
            tap = r % taps;
            g = r / taps;
            ci = kmap ? kmap[n] : n;
            if (ci >= 0) p = slabs + (((long long)g * taps + tap) * Mpad + m) * Npad + n;
        }
        const float s = split_sum(p, slab, nsplit, red);
        if (threadIdx.x < 32 && ci >= 0) {
            float* o = (gd.n ? gd.dw[g] : dw + (long long)g * M * Cin * taps) + ((long long)m * Cin + ci) * taps + tap;
            *o = accumulate ? *o + s : s;
        }
    }
}

// The same reduction, 16 bytes per lane: a thread owns four consecutive columns n of one (group, tap, row) and sums
// them over the splits in a fixed order (P = 1, 2, 4 or 8 threads share the splits when there are many; their partial
// sums meet in LDS, again in a fixed order).  The 32-outputs-per-block kernel above spent its time waiting for two
// dependent 4-byte loads per thread (27 us per launch at the C2 shapes, 715 launches per step); used when N % 4 == 0.
__global__ void reduce_weight4_kernel(const float* slabs, int nsplit, int G, int taps, int M, int N, int Mpad, int Npad,
                                      const int* kmap, int Cin, float* dw, int accumulate, const float* bias_slabs,
                                      float* db, int wblocks, const GroupDst gd, int P) {
    __shared__ f32x4 red4[256];
    __shared__ float red[256];
    if ((int)blockIdx.x >= wblocks) {    // trailing blocks: bias gradient (as in reduce_weight_kernel)
        bias_block_sum(bias_slabs, nsplit, G, M, Mpad, db, accumulate, (int)blockIdx.x - wblocks, gd, red);
        return;
    }
    const int per = 256 / P, ol = threadIdx.x % per, part = threadIdx.x / per;
    const int n4s = N / 4;
    const long long total4 = (long long)G * taps * M * n4s;
    const long long slab = (long long)G * taps * Mpad * Npad;
    const long long nchunk = (total4 + per - 1) / per;
    for (long long chunk = blockIdx.x; chunk < nchunk; chunk += wblocks) {
        const long long idx = chunk * per + ol;
        const bool live = idx < total4;
        int n = 0, m = 0, tap = 0, g = 0;
        if (live) {
            n = (int)(idx % n4s) * 4;
            long long r = idx / n4s;
            m = r % M; r /= M;
            tap = r % taps;
            g = r / taps;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // destinations (and, when accumulating, their present values: loaded beside the slabs, not after them)
        float* o[4] = {nullptr, nullptr, nullptr, nullptr};
        float old[4] = {0.f, 0.f, 0.f, 0.f};
        if (live && part == 0) {
            float* const base = (gd.n ? gd.dw[g] : dw + (long long)g * M * Cin * taps) + (long long)m * Cin * taps + tap;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = kmap ? kmap[n + k] : n + k;
                if (ci >= 0) {
                    o[k] = base + (long long)ci * taps;
                    if (accumulate) old[k] = ldg4(o[k]);
                }
            }
        }
        if (live) {
            const float* p = slabs + (((long long)g * taps + tap) * Mpad + m) * Npad + n;
            int i = part;
            for (; i + 3 * P < nsplit; i += 4 * P) {     // four loads in flight, added in split order
                const f32x4 v0 = ldg16(p + (long long)i * slab), v1 = ldg16(p + (long long)(i + P) * slab);
                const f32x4 v2 = ldg16(p + (long long)(i + 2 * P) * slab), v3 = ldg16(p + (long long)(i + 3 * P) * slab);
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; i < nsplit; i += P) acc += ldg16(p + (long long)i * slab);
        }
        if (P > 1) {
            red4[threadIdx.x] = acc;
            __syncthreads();
            if (part == 0) {
                acc = red4[ol];
                for (int k = 1; k < P; ++k) acc += red4[k * per + ol];
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (o[k]) stg4(o[k], old[k] + acc[k]);
    }
}

__global__ void reduce_plain_kernel(const float* slabs, int nsplit, int G, int M, int N, int Mpad, int Npad, float scale,
                                    float* out) {
    __shared__ float red[256];
    const long long total = (long long)G * M * N;
    const long long slab = (long long)G * Mpad * Npad;
    const long long nchunk = (total + 31) / 32;
    for (long long chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
        const long long idx = chunk * 32 + (threadIdx.x & 31);
        const float* p = nullptr;
        if (idx < total) {
            const int n = idx % N;
            long long r = idx / N;
            const int m = r % M;
            const int g = r / M;
            p = slabs + ((long long)g * Mpad + m) * Npad + n;
        }
        const float s = split_sum(p, slab, nsplit, red);
        if (threadIdx.x < 32 && idx < total) out[idx] = s * scale;
    }
}

}  // namespace

extern "C" int bmc_pgemm(const bmc_pgemm_args_t* h, bmc_stream_t stream) {
    BMC_CHECK_ARG(h != nullptr, "bmc_pgemm: null args");
    BMC_CHECK_ARG(h->nsrc >= 1 && h->nsrc <= BMC_MAX_SRC, "bmc_pgemm: nsrc=%d out of range", h->nsrc);
    BMC_CHECK_ARG(h->taps == 1 || h->taps == 9, "bmc_pgemm: taps must be 1 or 9");
    BMC_CHECK_ARG(h->tap_groups == 0 || h->tap_groups == 1 || (h->tap_groups == 3 && h->taps == 9 && h->math != BMC_MATH_BF16X6),
                  "bmc_pgemm: tap_groups must be 0, 1, or 3 (3: taps = 9, fp32 or bf16 arithmetic)");
    BMC_CHECK_ARG(h->batch_per_group >= 1 && h->B % h->batch_per_group == 0, "bmc_pgemm: B %% batch_per_group != 0");
    BMC_CHECK_ARG(h->nsplit >= 1 && h->slabs, "bmc_pgemm: nsplit/slabs");
    BMC_CHECK_ARG(h->zeros != nullptr, "bmc_pgemm: the zero buffer is required");
    BMC_CHECK_ARG(h->a.ptr && h->a.nch > 0 && h->a.nch % 4 == 0 && h->a.pix_stride % 4 == 0, "bmc_pgemm: bad A operand");
    PgemmK k;
    k.a = to_dev(h->a);
    k.nsrc = h->nsrc;
    int N = 0;
    for (int i = 0; i < BMC_MAX_SRC; ++i) {
        if (i < h->nsrc) {
            BMC_CHECK_ARG(h->src[i].ptr && h->src[i].nch > 0 && h->src[i].nch % 4 == 0 && h->src[i].pix_stride % 4 == 0,
                          "bmc_pgemm: bad source %d", i);
            k.src[i] = to_dev(h->src[i]);
            N += h->src[i].nch;
        } else k.src[i] = to_dev(h->src[0]);
    }
    k.B = h->B; k.H = h->H; k.W = h->W; k.batch_per_group = h->batch_per_group;
    k.slabs = h->slabs; k.nsplit = h->nsplit; k.zeros = h->zeros; k.bias_slabs = h->bias_slabs;
    k.M = h->a.nch; k.Mpad = bmc_round_up(k.M, 32); k.N = N; k.Npad = bmc_round_up(N, 32);
    k.G = h->B / h->batch_per_group;
    k.tap_groups = h->tap_groups == 3 ? 3 : 1;
    k.n_mblk = (k.Mpad + 127) / 128;
    hipStream_t st = (hipStream_t)stream;
    BMC_CHECK_ARG(h->math == BMC_MATH_FP32 || h->math == BMC_MATH_BF16 || h->math == BMC_MATH_BF16X6,
                  "bmc_pgemm: unknown math mode %d", h->math);
    if (h->math != BMC_MATH_FP32) {
        const int cols = bmc_pgemm_cols(h->taps, h->math);
        k.n_nblk = (k.Npad + cols - 1) / cols;
        if (h->taps == 9) {
            k.tiles_x = (h->W + PT_W - 1) / PT_W; k.tiles_y = (h->H + PT_H - 1) / PT_H;
            k.tiles_per_img = k.tiles_x * k.tiles_y;
        } else {
            k.tiles_x = k.tiles_y = 0;
            k.tiles_per_img = (h->H * h->W + PT - 1) / PT;
        }
        return bmc_pgemm_bf_launch(k, h->taps, h->math == BMC_MATH_BF16 ? 1 : 3, st);
    }
    const bool tab = pgemm_uses_tables(k);
    if (h->taps == 9) {
        k.n_nblk = (k.Npad + 63) / 64;
        k.tiles_x = (h->W + PT_W - 1) / PT_W; k.tiles_y = (h->H + PT_H - 1) / PT_H;
        k.tiles_per_img = k.tiles_x * k.tiles_y;
        if (h->tap_groups == 3) {
            dim3 grid((unsigned)((long long)k.G * k.n_mblk * k.n_nblk * 3 * k.nsplit));
            if (tab) hipLaunchKernelGGL((pgemm_dma_kernel<9, 3, true>), grid, dim3(512), 0, st, k);
            else hipLaunchKernelGGL((pgemm_dma_kernel<9, 3>), grid, dim3(512), 0, st, k);
        } else {
            dim3 grid((unsigned)((long long)k.G * k.n_mblk * k.n_nblk * k.nsplit));
            if (tab) hipLaunchKernelGGL((pgemm_dma_kernel<9, 9, true>), grid, dim3(512), 0, st, k);
            else hipLaunchKernelGGL(pgemm_dma_kernel<9>, grid, dim3(512), 0, st, k);
        }
    } else {
        k.n_nblk = (k.Npad + 127) / 128;
        k.tiles_x = k.tiles_y = 0;
        k.tiles_per_img = (h->H * h->W + PT - 1) / PT;
        dim3 grid((unsigned)((long long)k.G * k.n_mblk * k.n_nblk * k.nsplit));
        if (tab) hipLaunchKernelGGL((pgemm_dma_kernel<1, 1, true>), grid, dim3(512), 0, st, k);
        else hipLaunchKernelGGL(pgemm_dma_kernel<1>, grid, dim3(512), 0, st, k);
    }
    BMC_CHECK_LAUNCH("bmc_pgemm");
    return 0;
}

// threads sharing one output's splits: at most four loads per thread, all in flight at once (the kernel is latency-bound)
// (many splits: 16 / 32 threads per output quad -- the 1x1 weight gradients have only 4 096 quads, 8 parts left half the
//  CUs without a block and every thread with 8 dependent rounds of loads)
static int reduce_parts(int nsplit) { return nsplit <= 4 ? 1 : nsplit <= 8 ? 2 : nsplit <= 16 ? 4 : nsplit <= 64 ? 8 : nsplit <= 128 ? 16 : 32; }

extern "C" int bmc_pgemm_reduce_weight(const float* slabs, int nsplit, int G, int taps, int M, int N, const int* kmap,
                                       int Cin, float* dw, int accumulate, const float* bias_slabs, float* db,
                                       bmc_stream_t stream) {
    BMC_CHECK_ARG(slabs && dw && nsplit >= 1, "bmc_pgemm_reduce_weight: bad args");
    BMC_CHECK_ARG((bias_slabs == nullptr) == (db == nullptr), "bmc_pgemm_reduce_weight: bias_slabs and db go together");
    const long long total = (long long)G * taps * M * N;
    const int wblocks = (int)((total + 31) / 32 > 8192 ? 8192 : (total + 31) / 32);
    const int bblocks = bias_slabs ? (int)(((long long)G * M + BIAS_OUT - 1) / BIAS_OUT) : 0;
    GroupDst gd = {};
    if (N % 4 == 0) {
        const int P = reduce_parts(nsplit), per = 256 / P;
        const long long nchunk = ((long long)G * taps * M * (N / 4) + per - 1) / per;
        const int wb4 = (int)(nchunk > 8192 ? 8192 : nchunk);
        hipLaunchKernelGGL(reduce_weight4_kernel, dim3(wb4 + bblocks), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, G, taps,
                           M, N, bmc_round_up(M, 32), bmc_round_up(N, 32), kmap, Cin, dw, accumulate, bias_slabs, db, wb4, gd, P);
    } else
        hipLaunchKernelGGL(reduce_weight_kernel, dim3(wblocks + bblocks), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, G,
                           taps, M, N, bmc_round_up(M, 32), bmc_round_up(N, 32), kmap, Cin, dw, accumulate, bias_slabs, db,
                           wblocks, gd);
    BMC_CHECK_LAUNCH("bmc_pgemm_reduce_weight");
    return 0;
}

extern "C" int bmc_pgemm_reduce_weight_groups(const float* slabs, int nsplit, int G, int taps, int M, int N, const int* kmap,
                                              int Cin, float* const* dw, int accumulate, const float* bias_slabs,
                                              float* const* db, bmc_stream_t stream) {
    BMC_CHECK_ARG(slabs && dw && nsplit >= 1 && G >= 1 && G <= 4, "bmc_pgemm_reduce_weight_groups: bad args (1 <= G <= 4)");
    BMC_CHECK_ARG((bias_slabs == nullptr) == (db == nullptr), "bmc_pgemm_reduce_weight_groups: bias_slabs and db go together");
    GroupDst gd = {};
    gd.n = G;
    for (int g = 0; g < G; ++g) {
        BMC_CHECK_ARG(dw[g] && (!db || db[g]), "bmc_pgemm_reduce_weight_groups: null destination for group %d", g);
        gd.dw[g] = dw[g];
        gd.db[g] = db ? db[g] : nullptr;
    }
    const long long total = (long long)G * taps * M * N;
    const int wblocks = (int)((total + 31) / 32 > 8192 ? 8192 : (total + 31) / 32);
    const int bblocks = bias_slabs ? (int)(((long long)G * M + BIAS_OUT - 1) / BIAS_OUT) : 0;
    if (N % 4 == 0) {
        const int P = reduce_parts(nsplit), per = 256 / P;
        const long long nchunk = ((long long)G * taps * M * (N / 4) + per - 1) / per;
        const int wb4 = (int)(nchunk > 8192 ? 8192 : nchunk);
        hipLaunchKernelGGL(reduce_weight4_kernel, dim3(wb4 + bblocks), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, G, taps,
                           M, N, bmc_round_up(M, 32), bmc_round_up(N, 32), kmap, Cin, nullptr, accumulate, bias_slabs, nullptr,
                           wb4, gd, P);
    } else
        hipLaunchKernelGGL(reduce_weight_kernel, dim3(wblocks + bblocks), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, G,
                           taps, M, N, bmc_round_up(M, 32), bmc_round_up(N, 32), kmap, Cin, nullptr, accumulate, bias_slabs,
                           nullptr, wblocks, gd);
    BMC_CHECK_LAUNCH("bmc_pgemm_reduce_weight_groups");
    return 0;
}

extern "C" int bmc_pgemm_reduce_plain(const float* slabs, int nsplit, int G, int M, int N, float scale, float* out,
                                      bmc_stream_t stream) {
    BMC_CHECK_ARG(slabs && out && nsplit >= 1, "bmc_pgemm_reduce_plain: bad args");
    const long long total = (long long)G * M * N;
    const int blocks = (int)((total + 31) / 32 > 8192 ? 8192 : (total + 31) / 32);
    hipLaunchKernelGGL(reduce_plain_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, G, M, N,
                       bmc_round_up(M, 32), bmc_round_up(N, 32), scale, out);
    BMC_CHECK_LAUNCH("bmc_pgemm_reduce_plain");
    return 0;
}
