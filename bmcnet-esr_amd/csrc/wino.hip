// 3x3 convolution on the fp32 matrix cores through the Winograd transform F(2x2, 3x3):
//
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray; B^T, G, A^T below)
//
// 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36: the step's dominant kernel (the 3x3 128 -> 128
// convolutions of the residual blocks, forward and data gradient: ~59 % of its FLOPs) is fp32-MFMA-bound at 0.83 of the
// matrix peak, so the only way to make it substantially faster in exact-fp32 arithmetic is to execute fewer multiplies.
// Numerics: B^T and A^T hold only 0 / +-1 (the input and output transforms are sums), G holds 1/2 (exact scalings); measured
// against float64, one convolution is 1.2-1.6x the direct fp32 kernel's error (3.5e-7 vs 2.2e-7), the whole recurrent network
// within 20 % of it (tools / DESIGN.md) -- far inside the 1e-4 budget.  Same semantics as conv.hip (bmc_conv): multi-source
// NHWC operands, per-group weights, fused bias / residual / ReLU / ReLU-mask / accumulate epilogue.
//
// Machine mapping: wino2_conv_kernel below (8 waves of v_mfma_f32_16x16x4_f32, two per SIMD).
// The bias rides in the accumulators of position (1, 1): A^T m A passes that position into all four outputs with weight 1.
#include "bmc_common.h"
#include "conv_k.h"
#include "dma_ring.h"
#include <stdlib.h>

#ifndef BMC_WINO_DMA_TAIL
#define BMC_WINO_DMA_TAIL 1      // 1: a stage's DMA requests sit between its last two MFMA groups; 0: in front of its first
#endif
#ifndef BMC_WINO_ABL
#define BMC_WINO_ABL 0     // ablation bits for tools/ builds only: 1 no MFMAs, 2 no weight DMA, 4 no halo loads, 8 no epilogue stores,
                           // 16 no patch reads / input transform, 32 no weight fragment reads, 64 no barriers, 128 no output transform
#endif

namespace {

constexpr int CK = BMC_CK;
constexpr int RS = 20;                  // floats per halo pixel in LDS (16 + 4 pad)
constexpr int TW = 16;                  // output pixel columns per workgroup tile; rows: the kernel's TH (8, or 4 for launches that
                                        // would leave CUs without a tile: small frames)
constexpr int HWD = TW + 2;
constexpr int XROW = (HWD * RS + 63) / 64 * 64;      // halo row stride (as conv.hip: rows start on a 256-byte boundary)
constexpr int XBUFA = 4096;              // floats per X buffer as allocated: 16 DMA instructions x 64 lanes x 4 floats >= 10 rows x XROW
constexpr int BN = 128;                 // output channels per workgroup tile
constexpr int WSTAGE = 4 * BN * CK;     // floats per weight stage: 4 positions (nu) x 128 rows x 16 channels
constexpr int NWR = 3, DW = 2;          // weight ring: stages, stages ahead
constexpr int VSTAGE = 4 * 32 * CK;     // floats of transformed input per stage: 4 positions (nu) x 32 tiles x 16 channels

// quad swizzle of 16-float LDS rows: dma_ring.h's table {0,2,3,1}[(row >> 2) & 3] -- conflict-free for the 16-row x 4-quad
// fragment reads of the 16x16x4 MFMA (wino2 below) AND for the 32-row reads of the 32x32x2 MFMA (any bijection of the
// four (row >> 2) & 3 classes is: the 16-lane groups of ds_read_b128 hold one row of each class per row & 3)
__device__ __forceinline__ int wswz(int row) { return swz(row); }

// packed pair arithmetic for the input transform of wino2 (exact: a - b, a * b + c)
typedef float f32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2w pk_subw(f32x2w a, f32x2w b) {
    f32x2w d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2w pk_fmaw(f32x2w a, f32x2w b, f32x2w c) {
    f32x2w d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x4 sub4w(f32x4 a, f32x4 b) {
    const f32x2w lo = pk_subw(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2w hi = pk_subw(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// 4 KB linear copy global -> LDS by this wave (4 LDS-DMA instructions, one asm block: the base pointer reaches its SGPR pair once, the instruction's immediate offset advances the global AND the LDS address alike)
__device__ __forceinline__ void dma4k(const void* gbase, unsigned lane_off, unsigned lds_addr) {
    const unsigned long long pv = reinterpret_cast<unsigned long long>(gbase);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    const unsigned long long b0 = ((unsigned long long)hi << 32) | lo;
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 m0, %2\n\ts_nop 4\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
        "global_load_lds_dwordx4 %0, %1 offset:3072"
        ::"v"(lane_off), "s"(b0), "s"(la) : "memory");
}

// ---- machine mapping: TWO waves per SIMD.
// Round 3's first kernel (removed in round 4) kept all 16 transform positions of 32 tiles x 32 channels in one wave (256 accumulator
// registers): one wave per SIMD, and every instruction that is not an MFMA -- fragment reads, DMA issue, waits, barriers, the output
// transform -- is time the matrix pipe idles (ablation: MFMAs alone 0.39 ms of the kernel's 0.60).  Here a workgroup has 8
// waves of v_mfma_f32_16x16x4_f32: wave w = output channels [16 w, 16 w + 16) x all 32 tiles (two 16-tile column blocks) x
// all 16 positions = 32 accumulator quads = 128 registers, so two waves share a SIMD and cover each other's stalls.
//   * D rows = channels (4 consecutive per lane), D columns = tiles: the output transform is still in-lane, a lane stores
//     16 bytes (4 channels) for each of the 4 pixels of its 2 tiles;
//   * a K step is 4 channels (lane group l >> 4 holds channel quad l >> 4 of the 16-channel chunk, component m feeds MFMA m):
//     per stage and wave 4 weight reads + 8 V reads (ds_read_b128) feed 32 MFMAs;
//   * every thread produces ONE quad of the next stage's V (tile tid & 31, channel quad (tid >> 5) & 3, position
//     nu = tid >> 7): its position needs two patch columns only -- 4 reads, 12 adds, 1 store;
//   * LDS, rings, DMA, barriers per stage: as above (8 waves share the copies: 4 KB of a weight stage, 2 halo instructions each).
//   * TH = 4 (one 16-tile column block per wave, 64 accumulators): half the pixels per workgroup for launches with fewer 8 x 16
//     tiles than CUs -- the same stage loop, weight ring and halo image (rows 6-9 of the image and tiles 16-31 of V are produced and
//     never read), so a workgroup's serial time is what shrinks: at 31x56 an 8-image launch is 128 tiles of 8 x 16 or 256 of 4 x 16.
template <int TH>
__global__ __launch_bounds__(512, 2) void wino2_conv_kernel(const ConvK a) {
    constexpr int HHT = TH + 2, NTB = TH / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * XBUFA + NWR * WSTAGE + 2 * VSTAGE + BMC_MAX_SRC * 8 + BN];
    float* const Xb = lds;
    float* const Wb = lds + 2 * XBUFA;
    float* const Vb = lds + 2 * XBUFA + NWR * WSTAGE;
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + 2 * XBUFA + NWR * WSTAGE + 2 * VSTAGE);
    float* const init_lds = lds + 2 * XBUFA + NWR * WSTAGE + 2 * VSTAGE + BMC_MAX_SRC * 8;
    const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Wb;
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xb;
    const bool bias_pre = a.bias != nullptr && a.batch_per_group >= a.B && a.ntn == 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    if (tid < BN) init_lds[tid] = (bias_pre && tid < a.Cout) ? a.bias[tid] : 0.f;
    __syncthreads();

    const int ntiles = a.ntiles;
    constexpr int NX_ = 8;
    const bool xcd_map = (gridDim.x % NX_) == 0 && ntiles >= (int)gridDim.x;
    const int xcd = blockIdx.x % NX_, xj = blockIdx.x / NX_, per_x = gridDim.x / NX_;
    const int t_lo = xcd_map ? (int)((long long)ntiles * xcd / NX_) : 0;
    const int t_hi = xcd_map ? (int)((long long)ntiles * (xcd + 1) / NX_) : ntiles;
    const int t_first = xcd_map ? t_lo + xj : (int)blockIdx.x;
    const int t_stride = xcd_map ? per_x : (int)gridDim.x;
    const int my_tiles = t_first < t_hi ? (t_hi - t_first + t_stride - 1) / t_stride : 0;
    if (my_tiles == 0) return;
    const int nchunks = a.nchunks;

    struct TileIt { int nt, tx, ty, b; };
    auto decode = [&](int t) {
        TileIt it;
        it.nt = t % a.ntn; t /= a.ntn;
        it.tx = t % a.tiles_x; t /= a.tiles_x;
        it.ty = t % a.tiles_y;
        it.b = t / a.tiles_y;
        return it;
    };

    // The halo and weight loaders visit t_first, t_first + t_stride, ...: they advance by the digits of t_stride with carries
    // (a decode() there is three runtime divisions of scalar code the compiler speculates into the per-stage path)
    const TileIt stp = decode(t_stride);
    auto advance = [&](TileIt it) {
        it.nt += stp.nt;
        if (it.nt >= a.ntn) { it.nt -= a.ntn; ++it.tx; }
        it.tx += stp.tx;
        if (it.tx >= a.tiles_x) { it.tx -= a.tiles_x; ++it.ty; }
        it.ty += stp.ty;
        if (it.ty >= a.tiles_y) { it.ty -= a.tiles_y; ++it.b; }
        it.b += stp.b;
        return it;
    };

    // ---- X loader: 16 DMA instructions per halo tile, 2 per wave (layout and quad stream as in the kernel above), in the
    // cheap form "uniform base (SGPR pair) + per-lane 32-bit offset": EVERY lane fetches a valid address -- pixels outside
    // the image the clamped edge pixel, pad quads the image's first quad -- and the lanes of out-of-image pixels overwrite
    // their LDS quad with zeros once the chunk has landed, before the barrier that publishes it (zero_x below; interior
    // tiles: nobody).  No per-lane 64-bit addresses, no selects: this kernel has no registers to spare (128 + 128).
    constexpr int NXD = 2;
    int xl_tile = t_first, xl_chunk = 0, s_idx = 0, c_in = 0, snch = 0;
    TileIt xl_it = decode(t_first);
    const float* sbase = nullptr;
    unsigned xoff[NXD];               // byte offset from the source's batch pointer (+ channel chunk)
    unsigned xzm = 0;                 // bit k: instruction k's quad belongs to a pixel outside the image (of the tile being LOADED)
    unsigned xzm_landed = 0;          // the same for the chunk in flight (zero_x consumes it; a tile's chunks share it)
    int sb_s = -1, sb_b = -1;         // (source, image) sbase belongs to
    auto src_select = [&]() {
        const SrcDev S = tab[s_idx];
        if (s_idx != sb_s || xl_it.b != sb_b) { sbase = src_batch_ptr(S, xl_it.b); sb_s = s_idx; sb_b = xl_it.b; }
        snch = S.nch;
        const int y0 = xl_it.ty * TH, x0 = xl_it.tx * TW;
        xzm = 0;
#pragma unroll
        for (int k = 0; k < NXD; ++k) {
            int ln;                           // (lane index re-derived here, once per tile and source: not a loop-invariant to keep
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));    // in a register -- or to spill)
            const int Q = (wave * NXD + k) * 64 + ln;
            const int hy = Q / 96, rq = Q - hy * 96, hx = rq / 5, q = rq - hx * 5;
            const bool real = hy < HHT && hx < HWD && q < 4;            // a quad some patch read will touch
            int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
            xzm |= (real && !inside) ? (1u << k) : 0u;
            y = y < 0 ? 0 : (y < a.H ? y : a.H - 1);
            x = x < 0 ? 0 : (x < a.W ? x : a.W - 1);
            xoff[k] = real ? (unsigned)(((y * a.W + x) * S.pix_stride + q * 4) * 4) : 0u;
        }
    };
    auto xl_setup = [&]() {
        s_idx = 0; c_in = 0; xl_chunk = 0;
        src_select();
    };
    auto load_x = [&](int buf) {
        const float* const base = sbase + c_in;
        xzm_landed = xzm;
#pragma unroll
        for (int k = 0; k < NXD; ++k)
            if (!(BMC_WINO_ABL & 4)) dma16(base, xoff[k], xb_lds + (unsigned)((buf * XBUFA + (wave * NXD + k) * 256) * 4));
        c_in += CK;
        if (++xl_chunk == nchunks) {
            xl_tile += t_stride;
            if (xl_tile < t_hi) xl_it = advance(xl_it);      // (past the last tile: the same tile again -- see the stage loop)
            xl_setup();
        } else if (c_in >= snch) {
            c_in = 0; ++s_idx;
            src_select();
        }
    };
    auto zero_x = [&](int buf) {      // after the chunk's DMA has landed (this wave's own pieces), before its barrier
        if (__builtin_amdgcn_ballot_w64(xzm_landed != 0) == 0) return;       // interior tile: wave-uniform skip
#pragma unroll
        for (int k = 0; k < NXD; ++k)
            if ((xzm_landed >> k) & 1) {
                f32x4 z;
                asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
                             : "=v"(z[0]), "=v"(z[1]), "=v"(z[2]), "=v"(z[3]));      // (made here: a zero quad kept live would be spilled)
                int zl;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(zl));
                *reinterpret_cast<f32x4*>(Xb + buf * XBUFA + ((wave * NXD + k) * 64 + zl) * 4) = z;
            }
    };

    // ---- W ring loader: wave w copies half (w & 1) of position nu = w >> 1 of a stage (4 KB).  Source pointer and ring slot
    // advance incrementally (the per-stage scalar work sits in front of the stage's MFMAs in BOTH waves of a SIMD at once --
    // they leave the barrier together -- so it is matrix-pipe idle time: ~120 scalar instructions per 32 MFMAs were 24 %)
    int wl_tile = t_first, wl_sub = 0, wslot = 0;
    const float* wp = nullptr;
    const long long wrow = (long long)a.Coutpad * CK;
    TileIt wl_it = decode(t_first);
    auto wl_setup = [&]() {
        const int grp = a.batch_per_group >= a.B ? 0 : wl_it.b / a.batch_per_group;
        wp = static_cast<const float*>(a.w) + (long long)grp * a.w_group_stride + (long long)wl_it.nt * BN * CK + (wave >> 1) * wrow +
             (wave & 1) * 1024;
        wl_sub = 0;
    };
    const unsigned wdst0 = wb_lds + (unsigned)(((wave >> 1) * BN * CK + (wave & 1) * 1024) * 4);
    auto issue_w = [&]() {
        if (!(BMC_WINO_ABL & 2)) dma4k(wp, (unsigned)(lane * 16), wdst0 + (unsigned)(wslot * WSTAGE * 4));
        wslot = wslot == NWR - 1 ? 0 : wslot + 1;
        wp += 4 * wrow;
        if (++wl_sub == 4 * nchunks) {
            wl_tile += t_stride;
            if (wl_tile < t_hi) wl_it = advance(wl_it);      // (past the last tile: the same tile again)
            wl_setup();
        }
    };

    // ---- fragment addressing
    const int ptile = tid & 31, ptr_ = ptile >> 3, ptc = ptile & 7;          // producer: tile, its row / column of tiles
    const int pq = (tid >> 5) & 3, pnu = __builtin_amdgcn_readfirstlane(tid >> 7);
    // the two patch columns B's column nu combines: (0, 2 : -) (1, 2 : +) (2, 1 : -) (1, 3 : -)
    const int ca = pnu == 0 ? 0 : (pnu == 2 ? 2 : 1), cb = pnu == 3 ? 3 : (pnu == 2 ? 1 : 2);
    const int poffa = (2 * ptr_) * XROW + (2 * ptc + ca) * RS + 4 * pq;       // + r * XROW
    const int poffb = (2 * ptr_) * XROW + (2 * ptc + cb) * RS + 4 * pq;
    const int vst = pnu * 32 * CK + ptile * CK + ((pq ^ wswz(ptile)) * 4);
    const int qsw = (lk ^ wswz(lj)) * 4;                                     // (16 w + lj and 16 tb + lj share (row >> 2) & 3)
    const int woff = (16 * wave + lj) * CK + qsw;                            // + nu * BN * CK
    const int voff = lj * CK + qsw;                                          // + nu * 32 * CK + tb * 16 * CK

    f32x4 acc[16][NTB];
    auto init_acc = [&]() {
        // (the lane's quad index is re-derived here, once per tile, from v_mbcnt: kept live across the stage loop the address was
        //  the one value the register allocator spilled, and its reload -- a scratch load -- waited, in order, for every DMA
        //  in flight at the top of every tile)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const f32x4 bq = *reinterpret_cast<const f32x4*>(init_lds + 16 * wave + 4 * (ln >> 4));
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) acc[p][tb] = p == 5 ? bq : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) asm volatile("" : "+a"(acc[p][tb]));
    };

    auto produce_load = [&](const float* xb, int xi, f32x4 (&d)[4]) __attribute__((always_inline)) {
        const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rb = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
        d[0] = *reinterpret_cast<const f32x4*>(xb + poffa + ra * XROW);
        d[1] = *reinterpret_cast<const f32x4*>(xb + poffa + rb * XROW);
        d[2] = *reinterpret_cast<const f32x4*>(xb + poffb + ra * XROW);
        d[3] = *reinterpret_cast<const f32x4*>(xb + poffb + rb * XROW);
    };
    // (VALU instructions are the scarce resource here -- beside fp32 MFMAs each costs ~16-20 cycles of matrix-pipe time
    //  (ablation) -- so the wave-uniform sign of the column combination is a multiplier of an exact fma, not a
    //  select between a sum and a difference: 12 instructions per thread and stage)
    // (packed arithmetic: v_pk_add_f32 with negated second operand for the differences -- the compiler emits four v_sub_f32 for
    //  an f32x4 subtraction -- and v_pk_fma_f32 for the signed sum: 6 instead of 12 VALU instructions per thread and stage)
    const float psgn = pnu == 1 ? 1.f : -1.f;
    const f32x2w psgn2 = {psgn, psgn};
    auto produce_store = [&](float* vb, int xi, const f32x4 (&d)[4]) __attribute__((always_inline)) {
        const f32x4 ta = xi == 1 ? d[0] + d[1] : sub4w(d[0], d[1]);      // (B^T d) at column ca
        const f32x4 tb = xi == 1 ? d[2] + d[3] : sub4w(d[2], d[3]);      //          at column cb
        const f32x2w lo = pk_fmaw(__builtin_shufflevector(tb, tb, 0, 1), psgn2, __builtin_shufflevector(ta, ta, 0, 1));
        const f32x2w hi = pk_fmaw(__builtin_shufflevector(tb, tb, 2, 3), psgn2, __builtin_shufflevector(ta, ta, 2, 3));
        *reinterpret_cast<f32x4*>(vb + vst) = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
    };

    // One stage = 32 MFMAs, split in two halves AROUND the barrier that publishes the next stage (conv.hip's pattern): the
    // fragments of positions nu = 0, 1 were read right after the previous barrier (under the previous stage's second half),
    // those of nu = 2, 3 are read at the top (their data has been public since that barrier) and consumed after this
    // stage's barrier -- so no MFMA ever waits for an LDS round trip that starts at a barrier, which is where all eight
    // waves of the workgroup would otherwise stall together (two waves per SIMD only cover each other when they are not in
    // lockstep).  The production of the next stage's V (4 reads, 12 adds, 1 store per thread) rides in the first half.
    f32x4 ufA[2], vfA[2][NTB];
    auto load_first = [&](const float* vb, const float* wb) __attribute__((always_inline)) {
#pragma unroll
        for (int nu = 0; nu < 2; ++nu) {
            if (BMC_WINO_ABL & 32) {
                ufA[nu] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) { vfA[nu][tb] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(vfA[nu][tb])); }
                asm volatile("" : "+v"(ufA[nu]));
                continue;
            }
            ufA[nu] = *reinterpret_cast<const f32x4*>(wb + nu * BN * CK + woff);
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) vfA[nu][tb] = *reinterpret_cast<const f32x4*>(vb + nu * 32 * CK + tb * 16 * CK + voff);
        }
    };
    auto mfma8 = [&](f32x4 (&c)[NTB], const f32x4& u, const f32x4 (&v)[NTB]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) {
                if (BMC_WINO_ABL & 1) c[tb][m] += u[m] * v[tb][m];
                else c[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[m], v[tb][m], c[tb], 0, 0, 0);
            }
    };
    // first half: everything before this stage's barrier
    auto stage_head = [&](const float* vb, const float* wb, int xi, const float* xb_n, float* vb_n, int xi_n, f32x4 (&ufB)[2],
                          f32x4 (&vfB)[2][NTB]) __attribute__((always_inline)) {
        f32x4 d[4];
        if (!(BMC_WINO_ABL & 16)) produce_load(xb_n, xi_n, d);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(acc[4 * xi + 0], ufA[0], vfA[0]);
        if (!(BMC_WINO_ABL & 16)) produce_store(vb_n, xi_n, d);
        __builtin_amdgcn_sched_barrier(0);
        // the second half's fragments: read HERE -- 8 MFMAs (256 cycles) before the barrier's lgkmcnt(0) -- and consumed
        // behind the barrier
#pragma unroll
        for (int nu = 0; nu < 2; ++nu) {
            if (BMC_WINO_ABL & 32) {
                ufB[nu] = f32x4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb) { vfB[nu][tb] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(vfB[nu][tb])); }
                asm volatile("" : "+v"(ufB[nu]));
                continue;
            }
            ufB[nu] = *reinterpret_cast<const f32x4*>(wb + (2 + nu) * BN * CK + woff);
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) vfB[nu][tb] = *reinterpret_cast<const f32x4*>(vb + (2 + nu) * 32 * CK + tb * 16 * CK + voff);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma8(acc[4 * xi + 1], ufA[1], vfA[1]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // second half: after the barrier; vb_n / wb_n: the NEXT stage's (just published) operands, or nullptr
    // (the DMA issue of the stage -- weights three stages ahead, after xi = 3 the halo two chunks ahead -- sits BETWEEN the two
    //  MFMA groups: scalar address work placed in front of a stage's first MFMA is matrix-pipe idle time, because the two
    //  waves of a SIMD leave the barrier together and reach it together)
    auto stage_tail = [&](int xi, const float* vb_n, const float* wb_n, const f32x4 (&ufB)[2], const f32x4 (&vfB)[2][NTB], int xbuf) __attribute__((always_inline)) {
        load_first(vb_n, wb_n);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(acc[4 * xi + 2], ufB[0], vfB[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (BMC_WINO_DMA_TAIL) {
            issue_w();
            if (xi == 3) load_x(xbuf);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma8(acc[4 * xi + 3], ufB[1], vfB[1]);
        __builtin_amdgcn_sched_barrier(0);
    };

    int ep_b = -1;
    const float* ep_res = nullptr;
    const float* ep_mask = nullptr;
    auto epilogue = [&](const TileIt& it) __attribute__((always_inline)) {
        pin_acc();
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (!(BMC_WINO_ABL & 128)) {
                // Y = A^T M A on whole accumulator quads (the four output channels of a lane at once): packed adds
                f32x4 ta[4], y[4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) ta[nu] = (acc[nu][tb] + acc[4 + nu][tb]) + acc[8 + nu][tb];
                y[0] = (ta[0] + ta[1]) + ta[2];
                y[1] = sub4w(sub4w(ta[1], ta[2]), ta[3]);
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) ta[nu] = sub4w(sub4w(acc[4 + nu][tb], acc[8 + nu][tb]), acc[12 + nu][tb]);
                y[2] = (ta[0] + ta[1]) + ta[2];
                y[3] = sub4w(sub4w(ta[1], ta[2]), ta[3]);
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p][tb] = y[p];
            }
            pin_acc();
        }
        const int g = a.batch_per_group >= a.B ? 0 : it.b / a.batch_per_group;
        const float* const biasg = a.bias ? a.bias + (long long)g * a.bias_group_stride : nullptr;
        float* const outb = a.out + (long long)it.b * a.out_batch_stride;
        if (it.b != ep_b) {
            ep_b = it.b;
            ep_res = a.residual.ptr ? src_batch_ptr(a.residual, it.b) : nullptr;
            ep_mask = a.mask.ptr ? src_batch_ptr(a.mask, it.b) : nullptr;
        }
        const float* const resb = ep_res;
        const float* const maskb = ep_mask;
        int eln;      // (lane index re-derived: the epilogue's lane-dependent values are not kept -- or spilled -- across the stages)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(eln));
        const int elk = eln >> 4, elj = eln & 15;
        const int co = it.nt * BN + 16 * wave + 4 * elk;
        const bool cok = co < a.Cout;
        const bool simple = !resb && !maskb && !a.accumulate && (bias_pre || !biasg);
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (biasg && !bias_pre && cok) bq = ldg16(biasg + co);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            const int t = 16 * tb + elj, tr = t >> 3, tc = t & 7;
            bool pok[4];
            int pix[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int y = it.ty * TH + 2 * tr + (p >> 1), x = it.tx * TW + 2 * tc + (p & 1);
                pok[p] = y < a.H && x < a.W && cok;
                pix[p] = y * a.W + x;
            }
            f32x4 v[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) v[p] = acc[p][tb] + bq;
            if (!simple) {
                auto fetch = [&](const float* base, int stride, f32x4 (&dd)[4], float fill) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        dd[p] = f32x4{fill, fill, fill, fill};
                        if (pok[p]) dd[p] = ldg16(base + pix[p] * stride + co);
                    }
                };
                if (resb) {
                    f32x4 dd[4];
                    fetch(resb, a.residual.pix_stride, dd, 0.f);
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[p] += dd[p];
                }
                if (a.relu) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[p][k] = fmaxf(v[p][k], 0.f);
                }
                if (maskb) {
                    f32x4 dd[4];
                    fetch(maskb, a.mask.pix_stride, dd, 1.f);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[p][k] = dd[p][k] > 0.f ? v[p][k] : 0.f;
                }
                if (a.accumulate) {
                    f32x4 dd[4];
                    fetch(outb, a.out_pix_stride, dd, 0.f);
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[p] += dd[p];
                }
            } else if (a.relu) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[p][k] = fmaxf(v[p][k], 0.f);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (pok[p] && !((BMC_WINO_ABL & 8) && v[p][0] != 12345.678f)) *reinterpret_cast<f32x4*>(outb + pix[p] * a.out_pix_stride + co) = v[p];
        }
        init_acc();
    };

    // ---- prologue
    xl_setup();
    wl_setup();
    load_x(0);
    for (int k = 0; k < (BMC_WINO_DMA_TAIL ? NWR : DW); ++k) issue_w();          // stage g's tail requests the weights of stage g + NWR
    dma_wait<0>();
    zero_x(0);
    if (BMC_WINO_DMA_TAIL) load_x(1);                 // the halo of the second chunk: in flight across the first stages
    __syncthreads();
    {
        f32x4 d[4];
        produce_load(Xb, 0, d);
        produce_store(Vb, 0, d);
    }
    __syncthreads();
    init_acc();
    load_first(Vb, Wb);

    // The loaders never stop: past the workgroup's last tile they stream that tile's operands again (valid addresses, into
    // ring slots / buffers nobody reads any more), so that no stage carries an "is there a next one" branch and every wait
    // is a constant.  The DMA still in flight is drained before the workgroup ends.
    int gs = 0, gc = 0, rslot = 0;
    for (int tile = t_first; tile < t_hi; tile += t_stride) {
        for (int c = 0; c < nchunks; ++c, ++gc) {
            const float* const xb = Xb + (gc & 1) * XBUFA;
            const float* const xbn = Xb + ((gc + 1) & 1) * XBUFA;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi, ++gs) {
                if (!BMC_WINO_DMA_TAIL) {
                    issue_w();
                    if (xi == 0) load_x((gc + 1) & 1);
                }
                const int nslot = rslot == NWR - 1 ? 0 : rslot + 1;
                f32x4 ufB[2], vfB[2][NTB];
                stage_head(Vb + (gs & 1) * VSTAGE, Wb + rslot * WSTAGE, xi, xi == 3 ? xbn : xb, Vb + ((gs + 1) & 1) * VSTAGE, (xi + 1) & 3,
                           ufB, vfB);
                // everything but the newest DMA is complete: stage gs + 1's weights have landed (younger than them: the 4
                // instructions of stage gs + 2's weights, and at xi = 0 the NXD halo instructions issued right behind those in the
                // previous stage's tail); the halo itself is complete by xi = 1, in time for the production at xi = 3
                if (BMC_WINO_DMA_TAIL ? xi == 0 : xi <= 1) dma_wait<4 + NXD>(); else dma_wait<4>();
                if (xi == 2) zero_x((gc + 1) & 1);       // the next chunk's halo has landed (it is older than stage gs + 1)
                if (BMC_WINO_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else ring_publish();
                stage_tail(xi, Vb + ((gs + 1) & 1) * VSTAGE, Wb + nslot * WSTAGE, ufB, vfB, gc & 1);
                rslot = nslot;
            }
        }
        epilogue(decode(tile));
    }
    dma_wait<0>();
}

// U = G g G^T for every (output channel, packed input channel) pair, in the kernel's streaming order.
//   forward  (transposed == 0): rows = output channels co (< Coutpad), K = packed input channels k (kmap[k] -> ci, < 0 = zero);
//   data gradient w.r.t. packed source channels [k0, k0 + nk) (transposed != 0): rows = n = k - k0 (< rows_pad), K = output
//   channels co (padded to kpad), taps mirrored: g'[ky][kx] = w[co][kmap[k0 + n]][2 - ky][2 - kx].
// out[g][K chunk][xi][nu][row][16], the 4 quads of a row XOR-swizzled with (row >> 2) & 3 (the kernel's LDS image).
__global__ void pack_wino_kernel(const float* __restrict__ w, const int* __restrict__ kmap, int G, int Cout, int Cin, int kpad,
                                 int rows_pad, int transposed, int k0, int nk, float* __restrict__ out) {
    const long long total = (long long)G * (kpad / CK) * rows_pad * CK;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = idx % CK;
        long long r = idx / CK;
        const int row = r % rows_pad; r /= rows_pad;
        const int chunk = r % (kpad / CK);
        const int g = r / (kpad / CK);
        const int k = chunk * CK + kk;
        float gt[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) gt[i / 3][i % 3] = 0.f;
        int co, ci;
        if (!transposed) { co = row; ci = kmap ? kmap[k] : (k < Cin ? k : -1); }
        else { co = k; ci = row < nk ? (kmap ? kmap[k0 + row] : k0 + row) : -1; }
        if (co < Cout && ci >= 0) {
            const float* p = w + (((long long)g * Cout + co) * Cin + ci) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) gt[i / 3][i % 3] = transposed ? p[8 - i] : p[i];
        }
        // rows of G: (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1)
        float gg[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            gg[0][b] = gt[0][b];
            gg[1][b] = 0.5f * ((gt[0][b] + gt[1][b]) + gt[2][b]);
            gg[2][b] = 0.5f * ((gt[0][b] - gt[1][b]) + gt[2][b]);
            gg[3][b] = gt[2][b];
        }
        const int pos = row * CK + (((kk >> 2) ^ ((0x1320 >> (4 * ((row >> 2) & 3))) & 3)) << 2) + (kk & 3);
        float* const o = out + (((long long)g * (kpad / CK) + chunk) * 16) * rows_pad * CK + pos;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float u0 = gg[xi][0];
            const float u1 = 0.5f * ((gg[xi][0] + gg[xi][1]) + gg[xi][2]);
            const float u2 = 0.5f * ((gg[xi][0] - gg[xi][1]) + gg[xi][2]);
            const float u3 = gg[xi][2];
            o[(long long)(xi * 4 + 0) * rows_pad * CK] = u0;
            o[(long long)(xi * 4 + 1) * rows_pad * CK] = u1;
            o[(long long)(xi * 4 + 2) * rows_pad * CK] = u2;
            o[(long long)(xi * 4 + 3) * rows_pad * CK] = u3;
        }
    }
}

}  // namespace

// Rows per workgroup tile for a launch of this geometry: 8, or 4 where the 8-row tiling leaves CUs without a tile (or with a
// badly quantised last round) and the 4-row one does not.  A 4-row workgroup costs ~0.6 of an 8-row one (same stage loop and
// weight stream, half the MFMAs).  bmc_hip/ops.py::wino_ok asks the same function (bmc_conv_wino_rows).
static int wino_rows(int B, int H, int W, int ntn, int cus) {
    const char* const e = getenv("BMC_WINO_TH");      // tests and experiments: 8 / 4 = always that tiling (read per launch: ~50 ns)
    const int mode = e ? atoi(e) : 0;
    if (mode == 8 || mode == 4) return mode;
    const long long tx = (W + TW - 1) / TW;
    const long long n8 = (long long)B * tx * ((H + 7) / 8) * ntn, n4 = (long long)B * tx * ((H + 3) / 4) * ntn;
    const long long r8 = (n8 + cus - 1) / cus, r4 = (n4 + cus - 1) / cus;
    return (r8 <= 2 && 6 * r4 < 10 * r8) ? 4 : 8;
}

extern "C" int bmc_conv_wino_rows(int B, int H, int W, int Coutpad, int cus) {
    return wino_rows(B, H, W, Coutpad / BN, cus > 0 ? cus : 256);
}

// Called by bmc_conv (conv.hip) for math == BMC_MATH_FP32_WINO once the argument block is validated.
int bmc_conv_wino_launch(ConvK k, int cus, hipStream_t st) {
    k.ntn = k.Coutpad / BN;
    const int th = wino_rows(k.B, k.H, k.W, k.ntn, cus);
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.H + th - 1) / th;
    const long long ntiles = (long long)k.B * k.tiles_x * k.tiles_y * k.ntn;
    if (ntiles >= (1ll << 31)) { bmc_set_error("bmc_conv (winograd): too many tiles"); return -1; }
    k.ntiles = (int)ntiles;
    // one workgroup per CU (148 KB of LDS), 8 waves x 128 accumulators.  (Round 3's first mapping -- 4 waves x 256 accumulators
    // on v_mfma_f32_32x32x2_f32, one wave per SIMD, 0.59 ms against this kernel's 0.45 -- was removed in round 4: git history.)
    // 32-bit per-lane DMA offsets: the launch must keep H * W * pix_stride * 4 below 2^31
    long long max_stride = 0;
    for (int i = 0; i < k.nsrc; ++i) max_stride = k.src[i].pix_stride > max_stride ? k.src[i].pix_stride : max_stride;
    if ((long long)k.H * k.W * max_stride * 4 >= (1ll << 31)) {
        bmc_set_error("bmc_conv (winograd): image too large for 32-bit offsets (H*W*pix_stride*4 must stay below 2^31)");
        return -1;
    }
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));
    if (th == 4) hipLaunchKernelGGL(wino2_conv_kernel<4>, grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL(wino2_conv_kernel<8>, grid, dim3(512), 0, st, k);
    return 0;
}

extern "C" int bmc_pack_weight_wino(const float* w, const int* kmap, int G, int Cout, int Cin, int Kpad, int Coutpad,
                                    int transposed, int k0, int nk, float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(w && out && G >= 1 && Cout >= 1 && Cin >= 1, "bmc_pack_weight_wino: bad arguments");
    BMC_CHECK_ARG(Kpad > 0 && Kpad % CK == 0 && Coutpad > 0 && Coutpad % 128 == 0,
                  "bmc_pack_weight_wino: Kpad must be a multiple of 16 and the row count a multiple of 128");
    BMC_CHECK_ARG(!transposed || (k0 >= 0 && nk >= 1 && nk <= Coutpad && Kpad >= Cout), "bmc_pack_weight_wino: bad transposed window");
    const long long total = (long long)G * (Kpad / CK) * Coutpad * CK;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)s, w, kmap, G,
                       Cout, Cin, Kpad, Coutpad, transposed, k0, nk, out);
    BMC_CHECK_LAUNCH("bmc_pack_weight_wino");
    return 0;
}
