// Weight gradient of a dense 3x3, 128 -> 128 channel convolution through the Winograd transform F(4x4, 3x3) on the fp32
// matrix cores (round 5): 36 multiplies per 4x4 output tile and channel pair -- 2.25 per output pixel, against 4 for the
// F(2x2) kernel of wino_wgrad.hip and 9 for the pixel-reduction GEMM of pgemm.hip:
//     dU[xi][nu][co][ci] = sum over tiles  dM[xi][nu][tile][co] * V[xi][nu][tile][ci],     dM = A dY A^T,  V = B^T d B,
//     dW = G^T dU G   (+ bias gradient = sum over tiles of dM[1][1] = the plain sum of the tile's 16 dY pixels),
// with d the 6x6 input patch and dY the 4x4 output-gradient patch of a tile (matrices: wino4.hip, the forward kernel).
// Numerics: 3.0e-6 rel-L2 per convolution against float64 (profiles/r04_wino_numerics.txt), contract 1e-3.
//
// Work split.  36 positions x 128 x 128 = 589 824 accumulators are eight workgroups' worth (144 per thread x 512 threads).
// What decides the split is the INPUT side: both operands are transformed per tile, and every workgroup that owns a slice of
// the output has to read and transform the pixels it needs.  A workgroup owns (xi group tg: xi in {3 tg .. 3 tg + 2}) x all six
// nu x (co half) x (ci half): it reads 64 channels of x (5 of the patch's 6 rows) and 64 channels of dY per tile -- 16 bytes per
// matrix-pipe cycle and CU, the least of the splits that fit the register file (4 positions x 128 x 128, the F(2x2) kernel's
// shape, would need 38) -- and with xi restricted to three rows the separable transforms cost each workgroup half of the full
// ones: the row combinations of three of B^T's / A's six rows, then the six column combinations of each.
// The tile axis is the reduction axis: `nsplit` workgroups per type share the stages in contiguous ranges and write partial
// sums, which wino4_wgrad_reduce_kernel adds in a fixed order and transforms to the 3x3 taps (deterministic).
//
// Stage = 4 horizontally adjacent tiles = 2 k-steps of v_mfma_f32_32x32x2_f32.  Wave (pg, cb, kb) owns the nine positions
// (3 xi) x (nu in {3 pg .. 3 pg + 2}) of the 32 x 32 block (co block cb, ci block kb): 9 x 16 accumulator registers, two waves
// per SIMD.  Per stage and wave: 18 MFMAs of 64 cycles, 18 ds_read2_b32 of operand fragments (lane -> channel l & 31, tile
// l >> 5 of the k-step: conflict-free).
// Raw pixels reach LDS by LDS-DMA, 39 pieces of 1 KB per stage spread over the eight waves (5 each): no staging registers.  Inside
// the image a piece is "scalar base of the stage + a per-lane offset that never changes"; stages that touch the image border take
// a slower path (clamped per-lane addresses; the quads of out-of-image pixels are overwritten with zeros once landed) so that
// the transforms never see the border.  Two raw buffers and two transformed images: 150 KB of LDS, one barrier per stage.
// Work that is not MFMAs is split by WAVE so that it never stands in front of MFMAs the matrix pipe is waiting for (SIMD partners
// are waves w and w + 4): waves 0-3 multiply and THEN transform the next stage on channel pairs, two items per lane -- wave 0 the
// lone xi row of x (xi = 0 / 5), waves 1 / 2 sum / difference of the two xi rows that share their sub-sums (1, 2 / 3, 4), wave 3
// all three xi of dY; waves 4-7 request the raw strips of the stage after next (all 39 DMA pieces) and THEN multiply.  While one
// partner's MFMAs run, the other one transforms or requests.
// Partial sums leave in REGISTER order (one 1 KB store per accumulator quad: [split][type][wave][position][quad][lane][4]);
// the reduction kernel knows the MFMA's D layout and reads them back as 16-byte quads of four consecutive output channels.
#include "bmc_common.h"
#include "dma_ring.h"
#include <stdlib.h>

#ifndef BMC_W4G_ABL
#define BMC_W4G_ABL 0     // ablation builds (tools/): 1 no MFMA, 2 no DMA, 4 no transforms, 8 no fragment reads, 16 every DMA from the
                          // first stage's pixels (cache hits), 32 no wait for the DMA at the end of a stage
#endif

#ifdef BMC_W4G_STAMP      // diagnostic build (tools/ only): per-wave cycle stamps of workgroup 8, iterations 40..103
__device__ unsigned long long g_w4g_stamp[8][64][8];
#define W4G_STAMP(it, k) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (it) >= 40 && (it) < 104) g_w4g_stamp[threadIdx.x >> 6][(it) - 40][(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4G_STAMP(it, k) do { } while (0)
#endif

namespace {

struct Wgrad4K {
    SrcDev a;            // dY  [B,H,W,128]
    SrcDev x;            // the convolution's input [B,H,W,128]
    int B, H, W;
    int TY, SX;          // tile rows per image, stages (groups of 4 tiles) per tile row
    int nstages, nsplit;
    float* part;         // [nsplit][36 positions][128 co][128 ci]
    float* bias_part;    // optional [nsplit][128]
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// a wave-uniform pointer, made opaque (readfirstlane) so that it lives in an SGPR pair
__device__ __forceinline__ const float* uni(const float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

constexpr int NPW = 18;                     // positions per workgroup: 3 xi x 6 nu
constexpr int TS = 4;                       // tiles per stage
constexpr int CH = 64;                      // channels of each operand per workgroup
constexpr int PIMG = TS * CH;               // floats of one position's image: [4 tiles][64 channels]
constexpr int IMG = NPW * PIMG;             // one operand's transformed image of a stage (18 KB)
constexpr int SIMG = 2 * IMG;               // dM image, then V image
constexpr int XPC = 23, YPC = 16;           // DMA pieces (4 pixels x 64 channels = 1 KB) of the raw x strip (5 rows x 18 pixels: 90
                                            // of the 92 pixel slots used) and of the raw dY strip (4 rows x 16 pixels)
constexpr int RAWX = XPC * 256;             // floats
constexpr int RAWF = (XPC + YPC) * 256;     // floats per raw buffer (39 KB)
constexpr int NPC = XPC + YPC;
constexpr int PPW = 10;                     // pieces per requesting wave: waves 4-7 carry 10 each (pieces w - 4 + 4 j; slot 39 = piece 38 once more)
constexpr int LDSF = 2 * RAWF + 2 * SIMG;   // 153 600 bytes

template <int TG>
__device__ __forceinline__ void wgrad4_body(const Wgrad4K& a, float* const lds, const int split, const int chh, const int kh) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const rawb = lds;
    float* const imgb = lds + 2 * RAWF;
    const unsigned lds_raw = (unsigned)(size_t)(__attribute__((address_space(3))) void*)rawb;

    // ---- consumer role: wave = (nu group pg, co block cb, ci block kb); lane -> channel l & 31 of the block, tile l >> 5 of a k-step
    const int pg = wave >> 2, cb = (wave >> 1) & 1, kb = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int aoff = 3 * pg * PIMG + lh * CH + 32 * cb + l31;             // + (6 u + j) * PIMG + 2 ks * CH
    const int boff = IMG + 3 * pg * PIMG + lh * CH + 32 * kb + l31;
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- roles.  The CU's vector-memory path takes one 1 KB piece per ~40-60 cycles (PMC + ablations) and an in-order wave
    // that issues a piece while requests queue there is blocked with everything behind it, its MFMAs included; a wave's
    // transform is a chain of dependent packed operations behind LDS round trips.  Neither must stand in front of MFMAs the
    // matrix pipe is waiting for, so the two kinds of work sit on DIFFERENT waves of every SIMD (partners are w and w + 4):
    //   waves 0-3: multiply, THEN transform the next stage (wave 0 the lone xi row of x, wave 1 / 2 sum / difference of the xi pair,
    //              wave 3 dY; two items per lane: tiles {0, 1} then {2, 3});
    //   waves 4-7: request the raw strips of the stage after next (all 39 pieces, 10 slots each), THEN multiply.
    // While one partner's MFMAs run, the other one transforms or requests.
    const bool requester = wave >= 4;
    unsigned poff[PPW];       // byte offset from the stage's base pixel (interior stages)
    unsigned pla[PPW];        // LDS byte address of the piece in raw buffer 0
    unsigned pxm = 0;         // bit j: piece j is an x piece
    auto piece_of = [&](const int j) { return min((wave & 3) + 4 * j, NPC - 1); };
    // pixel slot (row r, column c) of this lane's quad of piece p; false: not a pixel (the last two slots of the x strip)
    auto slot_of = [&](const int p, int& r, int& c) __attribute__((always_inline)) {
        if (p < XPC) { const int q = 4 * p + (lane >> 4); r = q / 18; c = q - 18 * r; return q < 90; }
        const int q = 4 * (p - XPC) + (lane >> 4); r = q >> 4; c = q & 15;
        return true;
    };
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = piece_of(j);
        int r, c;
        const bool real = slot_of(p, r, c);
        poff[j] = real ? (unsigned)(((r * a.W + c) * 128 + (lane & 15) * 4) * 4) : 0u;
        pla[j] = lds_raw + (unsigned)(p * 1024);
        pxm |= p < XPC ? 1u << j : 0u;
    }
    unsigned zm = 0;          // bit j: this lane's quad of piece j is a pixel outside the image (stage in flight)
    static_assert(PPW <= 32, "zm is a bit mask");

    const int per_img = a.TY * a.SX;
    const int st0 = (int)((long long)a.nstages * split / a.nsplit), st1 = (int)((long long)a.nstages * (split + 1) / a.nsplit);
    // (image, tile row, group of 4 tiles) of the next stage to request, advanced incrementally; per tile row: the image's base
    // pointers (channel half included), the first pixel of the two strips' first rows, "the rows touch the image border"
    int nb = st0 / per_img, nty, nsx;
    {
        const int r = st0 - nb * per_img;
        nty = r / a.SX; nsx = r - nty * a.SX;
    }
    nb = __builtin_amdgcn_readfirstlane(nb); nty = __builtin_amdgcn_readfirstlane(nty); nsx = __builtin_amdgcn_readfirstlane(nsx);
    const float* xbat = nullptr;
    const float* ybat = nullptr;
    const float* xrow = nullptr;
    const float* yrow = nullptr;
    bool rowborder = false;
    auto row_setup = [&]() __attribute__((always_inline)) {
        xbat = uni(src_batch_ptr(a.x, nb) + kh * CH);
        ybat = uni(src_batch_ptr(a.a, nb) + chh * CH);
        const int y0 = 4 * nty - 1 + TG;                               // first staged patch row (TG = 1: patch rows 1..5)
        rowborder = y0 < 0 || y0 + 4 >= a.H;
        xrow = xbat + (long long)y0 * a.W * 128;                       // (only used when the rows are inside the image)
        yrow = ybat + (long long)(4 * nty) * a.W * 128;
    };
    row_setup();

    // requests of the raw strips of the next stage: rq_begin fixes the stage (scalar state) and advances the cursor, rq_piece(j)
    // issues this wave's piece j into the raw buffer rq_begin named
    const float* rq_xs = nullptr;
    const float* rq_ys = nullptr;
    int rq_y0 = 0, rq_x0 = 0;
    unsigned rq_lo = 0;
    bool rq_border = false;
    auto rq_begin = [&](const int rb) __attribute__((always_inline)) {
        rq_lo = (unsigned)(rb * RAWF * 4);
        rq_y0 = 4 * nty - 1 + TG; rq_x0 = 16 * nsx - 1;
        rq_border = rowborder || nsx == 0 || rq_x0 + 17 >= a.W;
        // interior: base = the strip's first pixel; border: base = the image (per-lane offsets are absolute then)
        rq_xs = rq_border ? xbat : xrow + rq_x0 * 128;
        rq_ys = rq_border ? ybat : yrow + (rq_x0 + 1) * 128;
        zm = 0;
        if (BMC_W4G_ABL & 16) return;
        if (++nsx == a.SX) {
            nsx = 0;
            if (++nty == a.TY) { nty = 0; ++nb; }
            row_setup();
        }
    };
    auto rq_piece = [&](const int j) __attribute__((always_inline)) {
        if (BMC_W4G_ABL & 2) return;
        unsigned o = poff[j];
        if (rq_border) {
            asm volatile("; image border" ::: "memory");
            const bool isx = (pxm >> j) & 1;
            int r, c;
            const bool real = slot_of(piece_of(j), r, c);
            const int y = (isx ? rq_y0 : rq_y0 + 1 - TG) + r, x = (isx ? rq_x0 : rq_x0 + 1) + c;
            const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
            const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
            o = real ? (unsigned)(((yc * a.W + xc) * 128 + (lane & 15) * 4) * 4) : 0u;
            zm |= (real && !inside) ? (1u << j) : 0u;
        }
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(o), "s"(((pxm >> j) & 1) ? rq_xs : rq_ys),
                     "s"(pla[j] + rq_lo) : "memory");
    };
    // after the requests have landed, before the barrier that publishes them: pixels outside the image become zeros
    auto patch = [&](const int rb) __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(zm != 0) == 0) return;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
            if ((zm >> j) & 1) *reinterpret_cast<f32x4*>(rawb + (pla[j] - lds_raw) / 4 + rb * RAWF + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- producer role (waves 0-3): item = (tile, channel pair cp) of the stage; instance i of a wave: tiles 2 i + (lane >> 5)
    const int cp = lane & 31;
    f32x2 bsum = {0.f, 0.f};
    // B^T along a row of six: (4 0 -5 0 1 0) (0 -4 -4 1 1 0) (0 4 -4 -1 1 0) (0 -2 -1 2 1 0) (0 2 -1 -2 1 0) (0 4 0 -5 0 1)
    auto xcols = [&](const f32x2 (&w)[6], float* const dst) __attribute__((always_inline)) {
        const f32x2 ta = w[4] - 4.f * w[2], tb = w[3] - 4.f * w[1], tc = w[4] - w[2], te = w[3] - w[1];
        *reinterpret_cast<f32x2*>(dst) = 4.f * w[0] - 5.f * w[2] + w[4];
        *reinterpret_cast<f32x2*>(dst + PIMG) = ta + tb;
        *reinterpret_cast<f32x2*>(dst + 2 * PIMG) = ta - tb;
        *reinterpret_cast<f32x2*>(dst + 3 * PIMG) = tc + 2.f * te;
        *reinterpret_cast<f32x2*>(dst + 4 * PIMG) = tc - 2.f * te;
        *reinterpret_cast<f32x2*>(dst + 5 * PIMG) = 4.f * w[1] - 5.f * w[3] + w[5];
    };
    // A along a row of four: (1 0 0 0) (1 1 1 1) (1 -1 1 -1) (1 2 4 8) (1 -2 4 -8) (0 0 0 1); returns the nu = 1 entry
    auto ycols = [&](const f32x2 (&w)[4], float* const dst) __attribute__((always_inline)) -> f32x2 {
        const f32x2 s = w[0] + w[2], t = w[1] + w[3], p = w[0] + 4.f * w[2], q = w[1] + 4.f * w[3];
        const f32x2 m1 = s + t;
        *reinterpret_cast<f32x2*>(dst) = w[0];
        *reinterpret_cast<f32x2*>(dst + PIMG) = m1;
        *reinterpret_cast<f32x2*>(dst + 2 * PIMG) = s - t;
        *reinterpret_cast<f32x2*>(dst + 3 * PIMG) = p + 2.f * q;
        *reinterpret_cast<f32x2*>(dst + 4 * PIMG) = p - 2.f * q;
        *reinterpret_cast<f32x2*>(dst + 5 * PIMG) = w[3];
        return m1;
    };
    auto transform = [&](const float* const raw, float* const img) __attribute__((always_inline)) {
        if (BMC_W4G_ABL & 4) return;
        auto ld2 = [](const float* p) __attribute__((always_inline)) { return *reinterpret_cast<const f32x2*>(p); };
#pragma unroll
        for (int inst = 0; inst < 2; ++inst) {
            const int pt = 2 * inst + (lane >> 5);
            if (wave == 0) {
                // the lone xi row of x: xi = 0 = (4 0 -5 0 1 0) on patch rows 0, 2, 4; xi = 5 = (0 4 0 -5 0 1) on patch rows 1, 3, 5 --
                // staged rows 0, 2, 4 either way.  All 18 reads first
                const float* const s = raw + (4 * pt) * CH + 2 * cp;
                f32x2 d[3][6], w[6];
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) d[k][c] = ld2(s + (2 * k * 18 + c) * CH);
#pragma unroll
                for (int c = 0; c < 6; ++c) w[c] = 4.f * d[0][c] - 5.f * d[1][c] + d[2][c];
                xcols(w, img + IMG + ((TG == 0 ? 0 : 2) * 6) * PIMG + pt * CH + 2 * cp);
            } else if (wave < 3) {
                // the two xi rows on patch rows 1..4 (staged rows 1 - TG ..): xi = 1, 2 = (r4 - 4 r2) +- (r3 - 4 r1); xi = 3, 4 = (r4 - r2) +- 2 (r3 - r1);
                // wave 1 the sum, wave 2 the difference
                const float* const s = raw + ((1 - TG) * 18 + 4 * pt) * CH + 2 * cp;
                constexpr float al = TG == 0 ? -4.f : -1.f;
                const float ga = (TG == 0 ? 1.f : 2.f) * (wave == 2 ? -1.f : 1.f);
                f32x2 d[4][6], w[6];
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k][c] = ld2(s + (k * 18 + c) * CH);
#pragma unroll
                for (int c = 0; c < 6; ++c) w[c] = (d[3][c] + al * d[1][c]) + ga * (d[2][c] + al * d[0][c]);
                xcols(w, img + IMG + (((TG == 0 ? 1 : 0) + (wave == 2 ? 1 : 0)) * 6) * PIMG + pt * CH + 2 * cp);
            } else {
                // dY (wave 3), all three xi of the group: (y0, (y0 + y2) +- (y1 + y3)) or ((y0 + 4 y2) +- 2 (y1 + 4 y3), y3)
                const float* const s = raw + RAWX + (4 * pt) * CH + 2 * cp;
                constexpr float ka = TG == 0 ? 1.f : 4.f, la = TG == 0 ? 1.f : 2.f;
                f32x2 y[4][4], u0[4], u1[4], u2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k][c] = ld2(s + (16 * k + c) * CH);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x2 p = y[0][c] + ka * y[2][c], q = y[1][c] + ka * y[3][c];
                    if (TG == 0) { u0[c] = y[0][c]; u1[c] = p + la * q; u2[c] = p - la * q; }
                    else { u0[c] = p + la * q; u1[c] = p - la * q; u2[c] = y[3][c]; }
                }
                float* const dd = img + pt * CH + 2 * cp;
                ycols(u0, dd);
                const f32x2 m11 = ycols(u1, dd + 6 * PIMG);
                ycols(u2, dd + 12 * PIMG);
                if (TG == 0) bsum += m11;         // xi = 1, nu = 1: the sum of the tile's 16 dY pixels
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- matrix role: 9 positions x 2 k-steps on the stage's images.  The operand fragments of position i + 2 are requested
    // before the MFMAs of position i are issued (a ring of three register sets): no MFMA waits for an LDS round trip
    auto multiply = [&](const float* const img) __attribute__((always_inline)) {
        float af[3][2], bf[3][2];
        auto frag = [&](const int i) __attribute__((always_inline)) {
            const int u = i / 3, j = i - 3 * u, sl = i % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BMC_W4G_ABL & 8) { af[sl][ks] = 1.f + ks; bf[sl][ks] = 2.f + j; asm volatile("" : "+v"(af[sl][ks]), "+v"(bf[sl][ks])); continue; }
                af[sl][ks] = img[aoff + (6 * u + j) * PIMG + 2 * ks * CH];
                bf[sl][ks] = img[boff + (6 * u + j) * PIMG + 2 * ks * CH];
            }
        };
        frag(0);
        frag(1);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < 9) frag(i + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BMC_W4G_ABL & 1) acc[i][0] += af[i % 3][ks] * bf[i % 3][ks];
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i % 3][ks], bf[i % 3][ks], acc[i], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: stage st0 raw -> image 0, stage st0 + 1 requested
    const int n = st1 - st0;
    if (n > 0) {
        rq_begin(0);
        if (requester) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) rq_piece(j);
            dma_wait<0>();
            patch(0);
        }
        ring_publish();
        if (n > 1) {
            rq_begin(1);
            if (requester) {
#pragma unroll
                for (int j = 0; j < PPW; ++j) rq_piece(j);
            }
        }
        if (!requester) transform(rawb, imgb);
        dma_wait<0>();
        if (n > 1 && requester) patch(1);
        ring_publish();
    }
    for (int it = 0; it < n; ++it) {
        const float* const img = imgb + (it & 1) * SIMG;
        const bool req = it + 2 < n, more = it + 1 < n;                      // stage it + 2 -> the raw buffer stage it was made from
        W4G_STAMP(it, 0);
        if (req) rq_begin(it & 1);
        W4G_STAMP(it, 1);
        if (requester && req) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) rq_piece(j);
        }
        W4G_STAMP(it, 2);
        __builtin_amdgcn_sched_barrier(0);
        multiply(img);
        __builtin_amdgcn_sched_barrier(0);
        W4G_STAMP(it, 3);
        if (!requester && more) transform(rawb + ((it + 1) & 1) * RAWF, imgb + ((it + 1) & 1) * SIMG);
        W4G_STAMP(it, 4);
        if (!(BMC_W4G_ABL & 32)) dma_wait<0>();
        W4G_STAMP(it, 5);
        if (requester && req) patch(it & 1);
        ring_publish();
        W4G_STAMP(it, 6);
    }

    // ---- partial sums in register order: part[split][type][wave][position 3 u + j][quad m][lane][4] -- D row 8 m + 4 (l >> 5) + e
    // (co), column l & 31 (ci), e = 0..3: one coalesced 1 KB store per accumulator quad
    {
        float* const P = a.part + ((((long long)split * 8 + (TG * 4 + chh * 2 + kh)) * 8 + wave) * 9) * 1024 + lane * 4;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                stg16(P + (i * 4 + m) * 256, f32x4{acc[i][4 * m], acc[i][4 * m + 1], acc[i][4 * m + 2], acc[i][4 * m + 3]});
    }
    if (TG == 0 && kh == 0 && a.bias_part) {      // bias partial: the 4 tile slots added through LDS (the loop ended on a barrier)
        if (wave == 3) *reinterpret_cast<f32x2*>(lds + (lane >> 5) * CH + 2 * cp) = bsum;      // (the dY wave: tiles t and t + 2 per lane)
        __syncthreads();
        if (tid < CH) a.bias_part[(long long)split * 128 + 64 * chh + tid] = lds[tid] + lds[CH + tid];
    }
}

__global__ __launch_bounds__(512, 2) void wino4_wgrad_kernel(const Wgrad4K a) {
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    // workgroup -> (type, split): the eight types of a split read the same pixels -- on the same XCD (same L2) when the split
    // count allows (consecutive workgroup ids go round the 8 XCDs)
    int type, split;
    if ((a.nsplit & 7) == 0) {
        const int j = blockIdx.x >> 3;
        type = j & 7; split = (j >> 3) * 8 + (blockIdx.x & 7);
    } else {
        type = blockIdx.x & 7; split = blockIdx.x >> 3;
    }
    const int chh = (type >> 1) & 1, kh = type & 1;
    if (type & 4) wgrad4_body<1>(a, lds, split, chh, kh);
    else wgrad4_body<0>(a, lds, split, chh, kh);
}

// dW[co][k0 + ci][3][3] (+)= G^T (sum over splits of dU) G, db[co] (+)= sum of the bias partials.
// Block = four consecutive co (one accumulator quad of the main kernel) x 32 consecutive ci (one MFMA column block) x 8 position
// groups: thread (ci, pp) adds the positions p = pp, pp + 8, ... over ALL splits (fixed order), 16 bytes per load, applies
// G^T . G to what it holds, and the eight partial tap sets are added through LDS in a fixed order.
__global__ __launch_bounds__(256) void wino4_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ dw,
                                                                int ldw, int k0, int accumulate, const float* __restrict__ bias_part,
                                                                float* __restrict__ db) {
    __shared__ float red[8][36][32];
    const int ol = threadIdx.x & 31, pp = threadIdx.x >> 5;
    if ((int)blockIdx.y == 32) {         // bias: one block, 128 channels x 2 halves of the splits
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
    const int co0 = 4 * blockIdx.y, ci = blockIdx.x * 32 + ol;
    const int chh = co0 >> 6, cb = (co0 >> 5) & 1, m = (co0 >> 3) & 3, lh = (co0 >> 2) & 1;
    const int kh = ci >> 6, kb = (ci >> 5) & 1;
    // rows of G: (1/4 0 0) (-1/6 -1/6 -1/6) (-1/6 1/6 -1/6) (1/24 1/12 1/6) (1/24 -1/12 1/6) (0 0 1);
    // dW[i][j] = sum_xi,nu G[xi][i] G[nu][j] dU[xi][nu]
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    f32x4 tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p = pp; p < 36; p += 8) {
        const int xi = p / 6, nu = p - 6 * xi;
        const int type = (xi / 3) * 4 + chh * 2 + kh, wv = (nu / 3) * 4 + cb * 2 + kb, a9 = 3 * (xi % 3) + nu % 3;
        const float* ps = part + ((((long long)type * 8 + wv) * 9 + a9) * 4 + m) * 256 + (lh * 32 + ol) * 4;
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nsplit; ++s) u += ldg16(ps + (long long)s * 36 * 16384);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) tap[3 * i + jj] += (G[xi][i] * G[nu][jj]) * u;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[pp][e * 9 + t][ol] = tap[t][e];
    __syncthreads();
    for (int q = pp; q < 36; q += 8) {       // q = e * 9 + tap: output channel co0 + e
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) o += red[k][q][ol];
        const int e = q / 9, t = q - 9 * e;
        float* const d = dw + ((long long)(co0 + e) * ldw + k0 + ci) * 9 + t;
        *d = accumulate ? *d + o : o;
    }
}

}  // namespace

#ifdef BMC_W4G_STAMP
extern "C" int bmc_w4g_read_stamps(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4g_stamp), sizeof(unsigned long long) * 8 * 64 * 8) == hipSuccess ? 0 : -1;
}
#endif

static long long w4g_stages(int B, int H, int W) { return (long long)B * ((H + 3) / 4) * (((W + 3) / 4 + TS - 1) / TS); }

extern "C" int bmc_wgrad_wino4_nsplit(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    const long long stages = w4g_stages(B, H, W);
    const int per_type = bmc_num_cus() / 8 > 0 ? bmc_num_cus() / 8 : 1;
    return (int)(stages < per_type ? stages : per_type);
}

extern "C" int bmc_wgrad_wino4(const bmc_src_t* dy, const bmc_src_t* x, int B, int H, int W, int nsplit, float* part,
                               float* bias_part, bmc_stream_t s) {
    BMC_CHECK_ARG(dy && x && part && dy->ptr && x->ptr, "bmc_wgrad_wino4: null argument");
    BMC_CHECK_ARG(dy->nch == 128 && x->nch == 128, "bmc_wgrad_wino4: both operands must be 128-channel windows (got %d, %d)", dy->nch,
                  x->nch);
    BMC_CHECK_ARG(dy->pix_stride == 128 && x->pix_stride == 128, "bmc_wgrad_wino4: both operands must be dense in the channel axis "
                  "(pix_stride 128; got %d, %d)", dy->pix_stride, x->pix_stride);
    BMC_CHECK_ARG(B >= 1 && H >= 1 && W >= 1 && (long long)H * W * 128 < (1ll << 29), "bmc_wgrad_wino4: bad geometry");
    Wgrad4K k;
    k.a = to_dev(*dy); k.x = to_dev(*x);
    k.B = B; k.H = H; k.W = W;
    k.TY = (H + 3) / 4; k.SX = ((W + 3) / 4 + TS - 1) / TS;
    const long long stages = w4g_stages(B, H, W);
    BMC_CHECK_ARG(stages < (1ll << 31), "bmc_wgrad_wino4: too many tiles");
    BMC_CHECK_ARG(nsplit >= 1 && nsplit <= stages, "bmc_wgrad_wino4: nsplit must be in [1, %lld]", stages);
    k.nstages = (int)stages; k.nsplit = nsplit;
    k.part = part; k.bias_part = bias_part;
    hipLaunchKernelGGL(wino4_wgrad_kernel, dim3((unsigned)nsplit * 8), dim3(512), 0, (hipStream_t)s, k);
    BMC_CHECK_LAUNCH("bmc_wgrad_wino4");
    return 0;
}

extern "C" int bmc_wgrad_wino4_reduce(const float* part, int nsplit, float* dw, int ldw, int k0, int accumulate,
                                      const float* bias_part, float* db, bmc_stream_t s) {
    BMC_CHECK_ARG(part && dw && nsplit >= 1, "bmc_wgrad_wino4_reduce: bad arguments");
    BMC_CHECK_ARG(ldw >= 128 && k0 >= 0 && k0 + 128 <= ldw, "bmc_wgrad_wino4_reduce: columns [k0, k0 + 128) must lie inside the %d input "
                  "channels of the weight tensor", ldw);
    BMC_CHECK_ARG((bias_part == nullptr) == (db == nullptr), "bmc_wgrad_wino4_reduce: bias_part and db go together");
    hipLaunchKernelGGL(wino4_wgrad_reduce_kernel, dim3(4, bias_part ? 33 : 32), dim3(256), 0, (hipStream_t)s, part, nsplit, dw, ldw, k0,
                       accumulate, bias_part, db);
    BMC_CHECK_LAUNCH("bmc_wgrad_wino4_reduce");
    return 0;
}
