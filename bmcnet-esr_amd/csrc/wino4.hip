// 3x3 convolution on the fp32 matrix cores through the Winograd transform F(4x4, 3x3):
//
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          (Lavin & Gray, points 0, +-1, +-2, inf; matrices below)
//
// 36 multiplies per 4x4 output tile and (ci, co) pair instead of 144 (direct) or 64 (F(2x2), wino.hip): 1.78x fewer MFMA
// cycles than wino.hip's kernel, which ran the matrix pipes at 0.64-0.66 and was 45 % of the training step.
// Numerics (tests/wino_numerics.py, profiles/r04_wino_numerics.txt): one convolution 1.9e-6 rel-L2 against float64 (direct
// fp32 1.6e-7, F(2x2) 3.1e-7); two recurrent windows of the full network at 180x240 SR 1.0e-7 / 1.5e-7 (direct 4e-8 / 7e-8),
// whole gradient 9.7e-6 (direct 8.6e-6) -- inside the parity budget (1e-4 / 1e-3 contract, 2e-5 / 2e-4 regression bars).
// One restriction found by that experiment and enforced by the caller (bmc_hip/ops.py: `exact_zero`): convolutions whose
// input can hold exact zeros over whole receptive fields (raw event counts, and what follows them before the first
// normalisation, with zero biases and a zero state) keep F(2x2) in their FORWARD pass -- the reference gives exactly 0 there and
// relu'(0) = 0 gates the gradient; a 6x6 patch through rounded transformed weights gives +-1e-8, a coin flip of the mask.
// Same semantics as conv.hip (bmc_conv): multi-source NHWC operands, per-group weights, fused bias / residual / ReLU /
// ReLU-mask / accumulate epilogue.
//
// Machine mapping (v_mfma_f32_16x16x4_f32, D rows = output channels, D columns = Winograd tiles):
//   * workgroup = 8 waves = 16 CONSECUTIVE tiles of one image (flat tile index, row-major: a strip of 4 x 64 output pixels that
//     may wrap onto the next tile rows -- no padded tile rows / columns whatever the image size) x 128 output channels;
//     wave w = channels [16 w, 16 w + 16) x all 16 tiles x all 36 positions = 36 accumulator quads = 144 registers (AGPRs),
//     two waves per SIMD.  Both transforms are in-lane (a lane owns one tile's column of D).
//   * The transformed weights never touch LDS: wave w is the ONLY reader of its 16 rows of U = G g G^T, so it streams them from
//     L2 straight into MFMA A-operand registers (global_load_dwordx4, one 1 KB instruction per position, a ring of D positions
//     ahead; bmc_pack_weight_wino4 lays the 36 KB of a (chunk, wave) out in exactly that order).  No weight ring, no barrier
//     for weights, 72 KB of LDS free for the transformed input of a WHOLE 16-channel chunk:
//   * ONE barrier per chunk = per 144 MFMAs of every wave (wino.hip: per 32).  V = B^T d B of chunk c + 1 (36 positions x 16
//     tiles x 16 channels, double-buffered) is produced while chunk c is multiplied: waves 0-5 each make row xi = w of the
//     position grid for all (tile, channel quad) pairs -- 4 patch rows x 6 columns of 16-byte reads from the raw halo in LDS,
//     4 FMAs per column with wave-uniform coefficients (row xi of B^T), the fixed 6 -> 6 transform along the row, 6 stores --
//     spread over the chunk's MFMA pairs.
//   * Waves 6-7 are the loaders: the raw halo strip of chunk c + 2 (6 rows x 72 pixel slots x 20 floats, padded so that the
//     patch reads of four neighbouring tiles fall on different banks) goes straight into LDS by LDS-DMA, 17 instructions per
//     loader wave per chunk with per-lane source offsets from a small per-wave LDS table rebuilt once per tile; pixels outside
//     the image fetch the clamped edge pixel and are overwritten with zeros once landed (wino.hip's scheme).  Vector-memory
//     operations complete in order, so only these two waves have HBM-latency operations in front of their weight loads: they
//     run a ring of 12 positions (they hold no producer registers), waves 0-5 a ring of 6.  All waits are counted by hand
//     (the U loads are inline asm; an over-count can only wait longer: hidden younger operations are never waited for less).
//   * The epilogue applies A^T . A in place (in-lane, 100 quad operations), then finishes as conv.hip does: all loads, then all
//     stores, 16 bytes per lane for each of the tile's 16 pixels.
#include "bmc_common.h"
#include "conv_k.h"
#include "dma_ring.h"
#include <stdlib.h>
#include <type_traits>

#ifndef BMC_W4_PRIO
#define BMC_W4_PRIO 0
#endif
#ifndef BMC_W4_DPRIO
#define BMC_W4_DPRIO 5       // wave priority by position in the chunk (below: the chunk's pair loop); 0 = none, other values: the variants measured in NOTEBOOK.md R5.11
#endif
#ifndef BMC_W4_IL
#define BMC_W4_IL 1        // 1: the pair loop issues its side work (LDS reads, U requests, halo pieces, producer arithmetic) BETWEEN the pair's
                           // eight MFMAs, one or two instructions per MFMA gap (round 6); 0: round 4/5's form -- reads, eight MFMAs back to back,
                           // then requests and arithmetic
#endif
#ifndef BMC_W4_EPF
#define BMC_W4_EPF 1       // N > 0: N tile rows of the first epilogue operand of a launch is requested between the two passes of the output transform (round 6); 0: behind them
#endif
#ifndef BMC_W4_HSWAP
#define BMC_W4_HSWAP 0     // 1: the producer lanes of tiles 4-7 / 12-15 make the two channel halves of their item in the opposite order, which
                           // takes the 2-way bank conflict out of every patch read (round 6); 0: round 4/5's order
#endif
#ifndef BMC_W4_EPI
#define BMC_W4_EPI 0       // 0: all arithmetic, then all loads, then the 16 stores; 1: output transform, operand loads and stores row by row
                           // (round 6: measured 0.7 % SLOWER -- 0.3230 / 0.3537 against 0.3208 / 0.3513 ms, three alternating rounds; kept for the record)
#endif
#ifndef BMC_W4_XP
#define BMC_W4_XP 0        // experiments (tools/ builds only; results are wrong by design): 2 waves w and w + 4 stream the SAME rows of
                           // U (does the CU's L1 serve the SIMD partner's copy?), 3 every second U request left out (half the stream)
#endif
#ifndef BMC_W4_ABL
#define BMC_W4_ABL 0       // ablation bits (tools/ builds only): 1 no MFMAs, 2 no weight loads, 4 no halo DMA, 8 no stores,
                           // 16 no V production, 32 no V fragment reads, 64 no barriers, 128 no output transform
#endif

#ifdef BMC_W4_CLK        // diagnostic build (tools/ only): ONLY the two stamps around the whole kernel (cycles and wall time of every
#define BMC_W4_STAMP     // workgroup -> the clock the chip holds under this kernel), nothing inside the loops
#endif
#ifdef BMC_W4_STAMP      // diagnostic build (tools/ only): per-workgroup cycle stamps, read back with bmc_w4_read_stamps
__device__ unsigned long long g_w4_stamp[1024][16];
// per-wave stamps of workgroup 8 (an XCD-0 workgroup), first 24 chunks: [wave][chunk][k]
__device__ unsigned long long g_w4_wstamp[8][24][8];
#ifdef BMC_W4_CLK
#define W4_STAMP(i) do { } while (0)
#define W4_WSTAMP(chunk, k) do { } while (0)
#else
#define W4_STAMP(i) do { if (threadIdx.x == 0 && (i) < 14) g_w4_stamp[blockIdx.x][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W4_WSTAMP(chunk, k) do { if (blockIdx.x == 8 && (threadIdx.x & 63) == 0 && (chunk) < 24) g_w4_wstamp[threadIdx.x >> 6][(chunk)][(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define W4_STAMP(i) do { } while (0)
#define W4_WSTAMP(chunk, k) do { } while (0)
#endif

namespace {

__device__ float g_w4_trash[4];        // where the epilogue's lanes outside the image store (never read)

constexpr int CK = BMC_CK;
constexpr int NT = 16, NPOS = 36, BN = 128;
constexpr int RS = 20, NSLOT = 72;            // floats per pixel slot (16 + 4 pad), pixel slots per halo row
constexpr int XROWF = NSLOT * RS;             // floats per halo row
constexpr int XQUADS = 6 * XROWF / 4;         // 16-byte quads of a halo strip
constexpr int NLW = 2, PPW = 17;              // loader waves, DMA instructions (64 quads each) per loader wave and chunk
constexpr int XBUFA = NLW * PPW * 256;        // floats per X buffer as allocated
constexpr int VBUF = NPOS * NT * CK;          // floats of transformed input per chunk
constexpr int UBLK = NPOS * 256;              // floats of transformed weights per (chunk, wave): 36 positions x 1 KB
constexpr int UCH = 8 * UBLK;                 // per chunk
static_assert(XBUFA * 4 >= XQUADS * 16, "DMA coverage of the halo strip");

template <int IMM>
__device__ __forceinline__ void uload(f32x4& dst, const float* sbase, unsigned voff) {
    if (BMC_W4_ABL & 2) { asm volatile("" : "=v"(dst)); return; }
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// all but the newest N vector-memory operations of this wave are done; the two fragments are tied to the wait so that no
// use of them can be scheduled above it
template <int N>
__device__ __forceinline__ void uwait(f32x4& x, f32x4& y) {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}

// the same with the count given as a value that folds to a constant once the pair loop is unrolled
__device__ __forceinline__ void uwait_n(int n, f32x4& x, f32x4& y) {
    switch (n) {
#define W4_CASE(N) case N: uwait<N>(x, y); break;
        W4_CASE(4) W4_CASE(5) W4_CASE(6) W4_CASE(7) W4_CASE(8) W4_CASE(9) W4_CASE(10) W4_CASE(11) W4_CASE(12) W4_CASE(13) W4_CASE(14)
        W4_CASE(15) W4_CASE(16) W4_CASE(17) W4_CASE(18) W4_CASE(19) W4_CASE(20) W4_CASE(21) W4_CASE(22) W4_CASE(23) W4_CASE(24)
        W4_CASE(25) W4_CASE(26) W4_CASE(27) W4_CASE(28)
#undef W4_CASE
        default: uwait<0>(x, y); break;
    }
}

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N - 1.  (`#pragma unroll` is a request: in an experiment where the
// 18-pair body had grown past the unroller's size threshold it silently stayed a loop, the accumulators were indexed dynamically
// and moved to scratch memory -- 0.33 -> 0.39 ms without a word from the compiler.)
template <int I, int N>
struct W4For {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, I>{});
        W4For<I + 1, N>::run(f);
    }
};
template <int N>
struct W4For<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F&&) {}
};

struct W4Tile { int nt, wt, b; };

// LOADER = false: waves 0-5 (producer of row xi = wave of the position grid); true: waves 6-7 (halo DMA)
template <bool LOADER>
__device__ __forceinline__ void wino4_body(const ConvK& a, float* lds, const int wave, const int t_first, const int t_hi,
                                           const int t_stride) {
#ifndef BMC_W4_DP
#define BMC_W4_DP 6
#endif
#ifndef BMC_W4_DL
#define BMC_W4_DL 12       // (a ring depth must divide the 36 positions of a chunk: position p lives in slot p % D across chunks)
#endif
    constexpr int D = LOADER ? BMC_W4_DL : BMC_W4_DP;        // positions of U in flight
    constexpr int XLAST = (PPW - 1) / 2;      // the pairs 0 .. XLAST of a chunk issue the halo pieces 2 pp, 2 pp + 1
    static_assert(2 * (NPOS / 2 - XLAST - 1) >= PPW && XLAST + D / 2 + 1 < NPOS / 2, "table rebuild / landing fit behind the pieces");
    static_assert(NPOS % D == 0 && D % 2 == 0, "position p of every chunk lives in ring slot p % D");
    // halo pieces issued behind the requests of the pairs pp - D / 2 .. pp - 1 (pair indices modulo the chunk).  (Interleaved form:
    // a pair issues its pieces IN FRONT of its requests -- gaps 4-5, then 6-7 -- so the pieces of pair pp - D / 2 are older than
    // its requests and do not count.)
    auto xyounger = [](int pp) {
        int n = 0;
        for (int j = pp - D / 2 + (BMC_W4_IL ? 1 : 0); j < pp; ++j) {
            const int jj = (j + NPOS / 2) % (NPOS / 2);
            n += jj > XLAST ? 0 : (2 * jj + 1 < PPW ? 2 : 1);
        }
        return n;
    };
    float* const Xb = lds;
    float* const Vb = lds + 2 * XBUFA;
    int* const xtab = reinterpret_cast<int*>(lds + 2 * XBUFA + 2 * VBUF);          // [NLW][PPW][64]
    const SrcDev* const tab = reinterpret_cast<const SrcDev*>(lds + 2 * XBUFA + 2 * VBUF + NLW * PPW * 64);
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xb;

    const int lane = threadIdx.x & 63;
    const int lj = lane & 15, lk = lane >> 4;
    const int nchunks = a.nchunks;
    const int tpi = a.tiles_x * a.tiles_y;            // Winograd tiles per image
    const int wpi = (tpi + NT - 1) / NT;              // workgroup tiles per image

    // (lane-dependent addresses that are needed once or twice per chunk are re-derived from v_mbcnt where they are used: kept
    //  live across 144 MFMAs they are what the register allocator spills, and a scratch reload waits, in order, for every
    //  weight request in flight)
    auto lane_now = [&]() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto decode = [&](int t) {
        W4Tile it;
        it.nt = t % a.ntn; t /= a.ntn;
        it.wt = t % wpi;
        it.b = t / wpi;
        return it;
    };
    // A workgroup visits the tiles t_first, t_first + t_stride, ...: (nt, wt, b) advance by the digits of t_stride with carries.  No
    // integer division stays in the tile loop (three decodes per tile and wave were ~120 vector instructions, and the two reciprocals
    // they keep in registers were what the allocator spilled once the epilogue prefetched its operand: round 6).
    // tiles_x as a divisor the compiler cannot hoist: the per-tile divisions below recompute their reciprocal (~25 instructions per
    // tile) instead of keeping it in a vector register across the chunk loops, where every register is spoken for
    auto opaque = [](int v) __attribute__((always_inline)) {
        asm volatile("" : "+s"(v));
        return v;
    };
    auto tiles_x_now = [&]() __attribute__((always_inline)) { return opaque(a.tiles_x); };
    // (the same for the other per-tile divisions by launch constants: weight / bias group of an image, batch maps of the epilogue operands)
    auto group_of = [&](int b) __attribute__((always_inline)) { return a.batch_per_group >= a.B ? 0 : b / opaque(a.batch_per_group); };
    auto batch_ptr = [&](const SrcDev& sd, int b) __attribute__((always_inline)) {
        int bs = b + sd.batch_shift;
        if (sd.batch_mod > 0) bs %= opaque(sd.batch_mod);
        return sd.ptr + (long long)bs * sd.batch_stride;
    };
    const W4Tile stp = decode(t_stride);
    auto advance = [&](W4Tile it) {
        it.nt += stp.nt;
        if (it.nt >= a.ntn) { it.nt -= a.ntn; ++it.wt; }
        it.wt += stp.wt;
        if (it.wt >= wpi) { it.wt -= wpi; ++it.b; }
        it.b += stp.b;
        return it;
    };

    // ---------------------------------------------------------------- U stream (every wave: its own 16 rows)
    const unsigned uvoff = (unsigned)(lane * 16);
    auto ublock = [&](const W4Tile& it, int chunk) -> const float* {
        const int grp = group_of(it.b);
        const float* p = static_cast<const float*>(a.w) + (long long)grp * a.w_group_stride + (long long)it.nt * nchunks * UCH +
                         (long long)chunk * UCH + (BMC_W4_XP == 2 ? (wave & 3) : wave) * UBLK;
        const unsigned long long pv = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };

    // ---------------------------------------------------------------- X stream (loader waves), two chunks ahead of the MFMAs
    const int wl = wave - (8 - NLW);                  // loader index
    int xl_tile = t_first, xl_chunk = 0, s_idx = 0, c_in = 0, snch = 0, xl_buf = 0;
    W4Tile xl_it = decode(t_first);
    const float* sbase = nullptr;
    int x_stride4 = 0;                                // pixel stride of the current source in bytes
    unsigned xzm = 0, xzm_next = 0, xzm_landed = 0;   // bit k: instruction k's quad is a pixel outside the image (table in use / being rebuilt / in flight)
    bool xl_rebuild = false;                          // the stream has moved on to another tile: the table is rebuilt piece by piece
    int tb_ty0 = 0, tb_tx0 = 0, tb_st1 = 0, tb_st2 = 0, tb_st3 = 0, tb_st4 = 0;
    // once per tile: pixel index / channel quad / zero flag of every quad this wave copies
    auto table_setup = [&]() {
        const int T0 = xl_it.wt * NT;
        tb_ty0 = T0 / tiles_x_now(); tb_tx0 = T0 - tb_ty0 * a.tiles_x;
        // segments g = 0..3: the tiles of tile row ty0 + g, n_g of them from tile column (g == 0 ? tx0 : 0), pixel slots
        // [st_g, st_g + 4 n_g + 2): neighbours in a row share two columns, every row break starts a fresh 6-column patch
        int n = a.tiles_x - tb_tx0; n = n < NT ? n : NT;
        int left = NT - n;
        tb_st1 = 4 * n + 2;
        n = left < a.tiles_x ? left : a.tiles_x; left -= n;
        tb_st2 = tb_st1 + (n > 0 ? 4 * n + 2 : 0);
        n = left < a.tiles_x ? left : a.tiles_x; left -= n;
        tb_st3 = tb_st2 + (n > 0 ? 4 * n + 2 : 0);
        n = left < a.tiles_x ? left : a.tiles_x;
        tb_st4 = tb_st3 + (n > 0 ? 4 * n + 2 : 0);
        xzm_next = 0;
    };
    auto table_piece = [&](int k) __attribute__((always_inline)) {
        const int l = lane_now();
        const int Q = (wl * PPW + k) * 64 + l;
        const int r = Q / (XROWF / 4), rem = Q - r * (XROWF / 4);
        const int s = rem / 5, q = rem - 5 * s;
        const int g = (s >= tb_st1 ? 1 : 0) + (s >= tb_st2 ? 1 : 0) + (s >= tb_st3 ? 1 : 0);       // (slots >= st4 are not real)
        const int x = 4 * (g == 0 ? tb_tx0 : 0) - 1 + (s - (g == 0 ? 0 : g == 1 ? tb_st1 : g == 2 ? tb_st2 : tb_st3));
        const int y = 4 * (tb_ty0 + g) - 1 + r;
        const bool real = r < 6 && s < tb_st4 && q < 4;
        const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
        const int yc = y < 0 ? 0 : (y < a.H ? y : a.H - 1), xc = x < 0 ? 0 : (x < a.W ? x : a.W - 1);
        xzm_next |= (real && !inside) ? (1u << k) : 0u;
        xtab[(wl * PPW + k) * 64 + l] = real ? ((yc * a.W + xc) << 2) | q : 0;
    };
    auto src_select = [&]() {
        const SrcDev S = tab[s_idx];
        const unsigned long long pv = reinterpret_cast<unsigned long long>(src_batch_ptr(S, xl_it.b));
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
        sbase = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
        snch = __builtin_amdgcn_readfirstlane(S.nch);
        x_stride4 = __builtin_amdgcn_readfirstlane(S.pix_stride * 4);
    };
    auto xl_setup = [&]() {                           // prologue: the whole table at once
        s_idx = 0; c_in = 0; xl_chunk = 0;
        table_setup();
#pragma unroll 1
        for (int k = 0; k < PPW; ++k) table_piece(k);
        xzm = xzm_next;
        src_select();
    };
    // piece k of the halo strip of the stream's chunk -> buffer xl_buf: uniform base in SGPRs + per-lane byte offset.  The 17
    // offsets live in registers and change only with the tile or the source (pixel stride): recomputed from the table then,
    // at the top of a chunk (read from LDS per piece, each piece paid an LDS round trip + address arithmetic in front of its
    // DMA: ~400 cycles per piece, stamps)
    // (interleaved form: the LAST PPW - KREG offsets are not kept -- with a halo piece issued in the middle of a pair, all 17 beside
    //  144 accumulators, 12 ring slots and both V fragment buffers are 4 registers too many, and a scratch reload in front of a
    //  piece waits for every request in flight.  Those pieces take their offset from the table one pair ahead: xoj_*)
    constexpr int KREG = BMC_W4_IL ? 12 : PPW;
    unsigned xoff[KREG];
    bool xl_newoff = true;
    auto offsets_from_table = [&]() {
        const int l = lane_now();
#pragma unroll
        for (int k = 0; k < KREG; ++k) {
            const int e = xtab[(wl * PPW + k) * 64 + l];
            xoff[k] = __umul24((unsigned)(e >> 2), (unsigned)x_stride4) + (unsigned)(e & 3) * 16u;
        }
        xl_newoff = false;
    };
    int xoj_e[2] = {0, 0};                            // table entries of the two pieces the NEXT pair issues (k >= KREG)
    auto xoj_fetch = [&](int k, int slot) __attribute__((always_inline)) {
        const int l = lane_now();
        xoj_e[slot] = xtab[(wl * PPW + k) * 64 + l];
    };
    auto load_x_piece = [&](int k) __attribute__((always_inline)) {
        const unsigned la = xb_lds + (unsigned)((xl_buf * XBUFA + (wl * PPW + k) * 256) * 4);
        unsigned off;
        if (k < KREG) off = xoff[k < KREG ? k : 0];
        else { const int e = xoj_e[k & 1]; off = __umul24((unsigned)(e >> 2), (unsigned)x_stride4) + (unsigned)(e & 3) * 16u; }
        if (!(BMC_W4_ABL & 4))
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(sbase + c_in), "s"(la) : "memory");
    };
    auto load_x_begin = [&]() {
        if (xl_rebuild) { xzm = xzm_next; xl_rebuild = false; }
        if (xl_newoff) offsets_from_table();
        xzm_landed = xzm;
    };
    auto load_x_advance = [&]() {                     // all pieces of the stream's chunk are issued: on to the next one
        xl_buf ^= 1;
        c_in += CK;
        if (++xl_chunk == nchunks) {
            xl_tile += t_stride;
            if (xl_tile < t_hi) xl_it = advance(xl_it);       // (past the last tile: the same tile again -- valid addresses, nobody reads it)
            s_idx = 0; c_in = 0; xl_chunk = 0;
            table_setup();
            src_select();
            xl_rebuild = true;
            xl_newoff = true;
        } else if (c_in >= snch) {
            c_in = 0; ++s_idx;
            src_select();
            xl_newoff = true;
        }
    };
    auto load_x_all = [&]() {                         // prologue: a whole strip at once
        load_x_begin();
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            if (k >= KREG) xoj_fetch(k, k & 1);
            load_x_piece(k);
        }
        load_x_advance();
        if (xl_rebuild) {
#pragma unroll 1
            for (int k = 0; k < PPW; ++k) table_piece(k);
        }
    };
    auto zero_x = [&](int buf) {                      // after the burst has landed, before the barrier that publishes it
        if (__builtin_amdgcn_ballot_w64(xzm_landed != 0) == 0) return;
        const int l = lane_now();
#pragma unroll
        for (int k = 0; k < PPW; ++k)
            if ((xzm_landed >> k) & 1)
                *reinterpret_cast<f32x4*>(Xb + buf * XBUFA + ((wl * PPW + k) * 64 + l) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---------------------------------------------------------------- V production (waves 0-5), one chunk ahead of the MFMAs
    // row xi = wave of B^T restricted to the rows it touches: (rows, coefficients)
    const int xi = wave;
    const int pr0 = xi == 0 ? 0 : 1, pr1 = (xi == 0 || xi == 5) ? (xi == 0 ? 2 : 3) : 2, pr2 = xi == 0 ? 4 : (xi == 5 ? 5 : 3),
              pr3 = xi == 0 ? 4 : (xi == 5 ? 5 : 4);
    const float pk0 = (xi == 0 || xi == 5 || xi == 2) ? 4.f : (xi == 1 ? -4.f : (xi == 3 ? -2.f : 2.f));
    const float pk1 = (xi == 0 || xi == 5) ? -5.f : ((xi == 1 || xi == 2) ? -4.f : -1.f);
    const float pk2 = (xi == 0 || xi == 5 || xi == 1) ? 1.f : (xi == 2 ? -1.f : (xi == 3 ? 2.f : -2.f));
    const float pk3 = (xi == 0 || xi == 5) ? 0.f : 1.f;
    int prow[4];                                      // float offsets of (patch row k, this lane's tile, channel quad) in an X buffer
    auto prod_setup = [&](const W4Tile& it) {         // geometry of the tile whose chunks the producer is working on
        int pln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(pln));
        const int ptile = pln >> 2, pq = pln & 3;
        const int T0 = it.wt * NT;
        const int txn = tiles_x_now();
        const int ty0 = T0 / txn;
        const int ty = (T0 + ptile) / txn;
        const int slot = 4 * ptile + 2 * (ty - ty0);
        prow[0] = pr0 * XROWF + slot * RS + pq * 4;
        prow[1] = pr1 * XROWF + slot * RS + pq * 4;
        prow[2] = pr2 * XROWF + slot * RS + pq * 4;
        prow[3] = pr3 * XROWF + slot * RS + pq * 4;
    };
    // Bank conflicts of the patch reads.  A ds_read_b64 is served in two groups of 32 lanes = 8 tiles x 4 channel quads; a tile's slots
    // lie 4 * RS = 80 floats = 16 banks (mod 64) apart, so tiles t and t + 4 of a group met on the same banks: a 2-way conflict on
    // every one of the 48 patch reads of a chunk (PMC: 0.29 of the LDS cycles).  The eight bytes a lane reads are one HALF of its
    // 16-byte quad, and the other half's banks are free in that very instruction: the lanes of tiles 4-7 (12-15) read -- and later
    // store -- the halves in the opposite order (half h ^ 1 in pass h).  Every element of V ends up where it was.
    // hx = 2 floats for those lanes, 0 for the others; prow is shifted by +hx in front of pass 0, by -2 hx in front of pass 1 (whose
    // reads carry the immediate + 2), and back by +hx behind it.
    auto prow_shift = [&](int d) __attribute__((always_inline)) {
        if (!BMC_W4_HSWAP) return;
        const int hx = (lane_now() >> 3) & 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) prow[k] += d * hx;
    };
    // An item = row xi of (tile, channel quad) is made in two halves of two channels each (h = 0, 1): 6 + 4 + 4 register
    // pairs live instead of quads -- with 144 accumulators, the U ring and the V fragments the quads did not fit (90 spills)
    typedef float f32x2p __attribute__((ext_vector_type(2)));
    f32x2p pt[6];
    auto prod_col = [&](const float* xb, int h, int c, f32x2p (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = *reinterpret_cast<const f32x2p*>(xb + prow[k] + c * RS + 2 * h);
    };
    auto prod_ld = [&](const float* xb, int h, int c, int k, f32x2p (&d)[4]) __attribute__((always_inline)) {
        d[k] = *reinterpret_cast<const f32x2p*>(xb + prow[k] + c * RS + 2 * h);
    };
    // (an `asm volatile` that rewrites a value in place fixes WHERE the arithmetic producing it is emitted: instruction selection
    //  otherwise sinks pure arithmetic to its first use, across every sched_barrier)
    auto pin2 = [](f32x2p& x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); };
    auto prod_fma = [&](int c, const f32x2p (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) pt[c][e] = __builtin_fmaf(pk3, d[3][e], __builtin_fmaf(pk2, d[2][e], __builtin_fmaf(pk1, d[1][e], pk0 * d[0][e])));
    };
    // along the row: (4 0 -5 0 1 0) (0 -4 -4 1 1 0) (0 4 -4 -1 1 0) (0 -2 -1 2 1 0) (0 2 -1 -2 1 0) (0 4 0 -5 0 1)
    auto prod_row_a = [&](f32x2p& ta, f32x2p& tb, f32x2p& tc, f32x2p& te) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ta[e] = __builtin_fmaf(-4.f, pt[2][e], pt[4][e]);
            tb[e] = __builtin_fmaf(-4.f, pt[1][e], pt[3][e]);
            tc[e] = pt[4][e] - pt[2][e];
            te[e] = pt[3][e] - pt[1][e];
            pt[0][e] = __builtin_fmaf(4.f, pt[0][e], __builtin_fmaf(-5.f, pt[2][e], pt[4][e]));
            pt[5][e] = __builtin_fmaf(4.f, pt[1][e], __builtin_fmaf(-5.f, pt[3][e], pt[5][e]));
        }
    };
    auto prod_row_b = [&](const f32x2p& ta, const f32x2p& tb, const f32x2p& tc, const f32x2p& te) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            pt[1][e] = ta[e] + tb[e];
            pt[2][e] = ta[e] - tb[e];
            pt[3][e] = __builtin_fmaf(2.f, te[e], tc[e]);
            pt[4][e] = __builtin_fmaf(-2.f, te[e], tc[e]);
        }
    };
    // the same in pieces (interleaved form: one piece per MFMA gap); the order of the operations -- and so every rounding -- is prod_fma's
    auto prod_fma_a = [&](int c, const f32x2p (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) pt[c][e] = __builtin_fmaf(pk1, d[1][e], pk0 * d[0][e]);
    };
    auto prod_fma_b = [&](int c, const f32x2p (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) pt[c][e] = __builtin_fmaf(pk3, d[3][e], __builtin_fmaf(pk2, d[2][e], pt[c][e]));
    };
    auto prod_row_a1 = [&](f32x2p& ta, f32x2p& tb) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { ta[e] = __builtin_fmaf(-4.f, pt[2][e], pt[4][e]); tb[e] = __builtin_fmaf(-4.f, pt[1][e], pt[3][e]); }
    };
    auto prod_row_a2 = [&](f32x2p& tc, f32x2p& te) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { tc[e] = pt[4][e] - pt[2][e]; te[e] = pt[3][e] - pt[1][e]; }
    };
    auto prod_row_a3 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) pt[0][e] = __builtin_fmaf(4.f, pt[0][e], __builtin_fmaf(-5.f, pt[2][e], pt[4][e]));
    };
    auto prod_row_a4 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) pt[5][e] = __builtin_fmaf(4.f, pt[1][e], __builtin_fmaf(-5.f, pt[3][e], pt[5][e]));
    };
    auto prod_row_b1 = [&](const f32x2p& ta, const f32x2p& tb) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { pt[1][e] = ta[e] + tb[e]; pt[2][e] = ta[e] - tb[e]; }
    };
    auto prod_row_b2 = [&](const f32x2p& tc, const f32x2p& te) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { pt[3][e] = __builtin_fmaf(2.f, te[e], tc[e]); pt[4][e] = __builtin_fmaf(-2.f, te[e], tc[e]); }
    };
    auto prod_addr = [&](int h) __attribute__((always_inline)) {          // (the half of its quad this lane writes in pass h included)
        const int l = lane_now();
        int vst = xi * 6 * NT * CK + (l >> 2) * CK + (((l & 3) ^ swz(l >> 2)) * 4) + (BMC_W4_HSWAP ? (2 * h) ^ ((l >> 3) & 2) : 2 * h);
        asm volatile("" : "+v"(vst));
        return vst;
    };
    auto prod_store_at = [&](float* vb, int vst) __attribute__((always_inline)) {
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x2p*>(vb + vst + nu * NT * CK) = pt[nu];
    };
    auto prod_store = [&](float* vb, int h) __attribute__((always_inline)) { prod_store_at(vb, prod_addr(h)); };

    // ---------------------------------------------------------------- MFMA side
    const int voff = lj * CK + ((lk ^ swz(lj)) * 4);                                  // + pos * NT * CK
    f32x4 acc[NPOS];
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NPOS; ++p) asm volatile("" : "+v"(acc[p]));
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int p = 0; p < NPOS; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 ur[D];
    f32x4 vf[2][2];
    auto read_v = [&](const float* vb, int pair, f32x4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (BMC_W4_ABL & 32) { v[i] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(v[i])); continue; }
            v[i] = *reinterpret_cast<const f32x4*>(vb + (2 * pair + i) * NT * CK + voff);
        }
    };
    [[maybe_unused]] auto mfma8 = [&](f32x4& c0, f32x4& c1, const f32x4& u0, const f32x4& u1, const f32x4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (BMC_W4_ABL & 1) { c0[m] += u0[m] * v[0][m]; c1[m] += u1[m] * v[1][m]; continue; }
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(u0[m], v[0][m], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(u1[m], v[1][m], c1, 0, 0, 0);
        }
    };

    // One chunk: 18 pairs of positions.  ucur / unx: this wave's U blocks of this chunk and of the next one (positions that
    // the ring requests past 35); vb: this chunk's V; xbn / vbn: the producer's input and output (next chunk).
    auto chunk = [&](const float* ucur, const float* unx, const float* vb, const float* xbn, float* vbn, const int gbuf, const int gcx)
                     __attribute__((always_inline)) {
        f32x2p pd[4], ta, tb, tc, te;
        // the producer's timetable: half h of its item takes pairs PH0 + 8 h + (0..5: one patch column each, reads in front of
        // the pair's MFMAs, FMAs behind them), + 6 (along the row, first part), + 7 (second part, stores)
        constexpr int PH0 = 1;
        W4For<0, NPOS / 2>::run([&](auto ic) __attribute__((always_inline)) {
            constexpr int pp = decltype(ic)::value;
            constexpr int ph = pp >= PH0 + 8 ? 1 : 0, ps = pp - PH0 - 8 * ph;      // half, step within it (valid for PH0 <= pp < PH0 + 16)
            const bool pact = !LOADER && !(BMC_W4_ABL & 16) && pp >= PH0 && pp < PH0 + 16;
            constexpr int psc = ps < 0 ? 0 : (ps > 5 ? 5 : ps);        // (ps where it names a patch column; dead code otherwise)
            const int p0 = 2 * pp, p1 = p0 + 1, s0 = p0 % D, s1 = p1 % D;
            // U(p0), U(p1) have landed: younger are the D - 2 requests behind them and, on a loader wave, the halo pieces
            // issued since (behind the requests of pairs pp - D / 2 .. pp - 1: xyounger)
#if BMC_W4_DPRIO
            // Priority follows the wave's position in its chunk: 3, 2, 1, 0 by quarters of the pair loop.  The two waves of a SIMD share
            // its matrix pipe and its issue slots, arbitrated by priority, then age -- at equal priority the older wave wins every
            // conflict, finishes its 144 MFMAs in ~9.2 k cycles and waits ~3 k at the chunk's barrier while its partner works through
            // what it was denied (round 4's stamps).  With the priority falling along the chunk, whichever wave is BEHIND holds the
            // higher one: the pair stays in step and the pipe busy to the end of the chunk.  -1.3 % per launch (0.3225 -> 0.3182 ms);
            // static sets of waves at priority 1 measured nothing (R5.7).  Other values of the macro: the variants of NOTEBOOK.md R5.11
            // (1: high in the first half; 2: high in the second half; 3: first third; 4: first sixth; 6: first third at 3; 7-9: other break points)
            if (BMC_W4_DPRIO <= 3) {
                if (pp == 0) __builtin_amdgcn_s_setprio(BMC_W4_DPRIO == 2 ? 0 : 1);
                if (pp == (BMC_W4_DPRIO == 3 ? NPOS / 6 : NPOS / 4)) __builtin_amdgcn_s_setprio(BMC_W4_DPRIO == 2 ? 1 : 0);
            } else if (BMC_W4_DPRIO == 4) {
                if (pp == 0) __builtin_amdgcn_s_setprio(1);
                if (pp == 3) __builtin_amdgcn_s_setprio(0);
            } else if (BMC_W4_DPRIO == 5) {
                if (pp == 0) __builtin_amdgcn_s_setprio(3);
                if (pp == 4) __builtin_amdgcn_s_setprio(2);
                if (pp == 9) __builtin_amdgcn_s_setprio(1);
                if (pp == 13) __builtin_amdgcn_s_setprio(0);
            } else if (BMC_W4_DPRIO == 7 || BMC_W4_DPRIO == 8 || BMC_W4_DPRIO == 9) {
                constexpr int e1 = BMC_W4_DPRIO == 7 ? 3 : (BMC_W4_DPRIO == 8 ? 6 : 2), e2 = BMC_W4_DPRIO == 7 ? 7 : (BMC_W4_DPRIO == 8 ? 10 : 5),
                              e3 = BMC_W4_DPRIO == 7 ? 12 : (BMC_W4_DPRIO == 8 ? 14 : 9);
                if (pp == 0) __builtin_amdgcn_s_setprio(3);
                if (pp == e1) __builtin_amdgcn_s_setprio(2);
                if (pp == e2) __builtin_amdgcn_s_setprio(1);
                if (pp == e3) __builtin_amdgcn_s_setprio(0);
            } else if (BMC_W4_DPRIO == 6) {
                if (pp == 0) __builtin_amdgcn_s_setprio(3);
                if (pp == 6) __builtin_amdgcn_s_setprio(0);
            } else if (BMC_W4_DPRIO == 10) {
                // (round 6 experiment: the SIMD partners w / w + 4 take turns pair by pair)
                if (((pp + (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            } else if (BMC_W4_DPRIO == 11) {
                // (... on top of the priority that falls along the chunk)
                constexpr int qb = pp < 4 ? 2 : (pp < 9 ? 2 : (pp < 13 ? 1 : 0));
                if (((pp + (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(qb + 1); else __builtin_amdgcn_s_setprio(qb);
            } else if (BMC_W4_DPRIO == 12) {
                // (... in blocks of three pairs = one row of the position grid)
                if ((((pp / 3) + (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            }
#endif
            if (LOADER) uwait_n(D - 2 + xyounger(pp), ur[s0], ur[s1]); else uwait<D - 2>(ur[s0], ur[s1]);
            if (LOADER && pp == XLAST + D / 2 + 1) zero_x(gbuf);     // (the wait above was the first behind the strip's last piece: it has landed)
#if BMC_W4_IL
            // ---- interleaved form.  A wave issues in order: behind eight back-to-back MFMAs its requests, LDS reads and arithmetic
            // run with NO matrix instruction of its own in the pipe, so whenever the SIMD partner is not multiplying at that moment
            // (it is parked at the barrier, waits for U, or has reached the same point of the same program) the pipe idles.  Here
            // every MFMA is followed by one or two of those instructions (an MFMA holds the SIMD's issue for 8 of its 32 cycles):
            // a wave keeps the pipe busy by itself.  Gaps 0-1: the producer's patch reads; 2-3: the next pair's V fragments;
            // 4-5: the loaders' halo pieces; 6-7: the U requests into the two ring slots this pair has just consumed, and the
            // producer's arithmetic on what gaps 0-1 read.
            if (pp == NPOS / 2 - 1) {
                // the chunk's barrier sits in front of its LAST pair (see the other form)
                W4_WSTAMP(gcx, 4);
                if (BMC_W4_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else ring_publish();
                W4_WSTAMP(gcx, 5);
            }
            if (pp == 0) W4_WSTAMP(gcx, 0);
            if (pp == 1) W4_WSTAMP(gcx, 1);
            if (pp == 7) W4_WSTAMP(gcx, 2);
            if (pp == 12) W4_WSTAMP(gcx, 3);
            if (LOADER && pp == 0) load_x_begin();      // (may branch: a new tile's or source's offsets)
            __builtin_amdgcn_sched_barrier(0);
            {
                const f32x4 (&v)[2] = vf[pp & 1];
                f32x4 (&vn)[2] = vf[(pp + 1) & 1];
                const float* const vnb = pp < NPOS / 2 - 1 ? vb + (2 * pp + 2) * NT * CK + voff : vbn + voff;
                const bool vread = !(BMC_W4_ABL & 32);
                auto mf = [&](f32x4& c, float u, float x) __attribute__((always_inline)) {
                    if (BMC_W4_ABL & 1) c[0] += u * x; else c = __builtin_amdgcn_mfma_f32_16x16x4f32(u, x, c, 0, 0, 0);
                };
                const int q0 = p0 + D;
                const float* const ub = q0 < NPOS ? ucur + (q0 / 4) * 1024 : unx + ((q0 - NPOS) / 4) * 1024;
                const bool lo = (q0 % NPOS) % 4 == 0;
                // gap 0
                mf(acc[p0], ur[s0][0], v[0][0]);
                if (pact && ps < 6) { prod_ld(xbn, ph, psc, 0, pd); prod_ld(xbn, ph, psc, 1, pd); }
                if (pact && ps == 6) { prod_row_a1(ta, tb); pin2(ta); pin2(tb); }
                if (pact && ps == 7) { prod_row_b1(ta, tb); pin2(pt[1]); pin2(pt[2]); }
                __builtin_amdgcn_sched_barrier(0);
                // gap 1
                mf(acc[p1], ur[s1][0], v[1][0]);
                if (pact && ps < 6) { prod_ld(xbn, ph, psc, 2, pd); prod_ld(xbn, ph, psc, 3, pd); }
                if (pact && ps == 6) { prod_row_a2(tc, te); pin2(tc); pin2(te); }
                if (pact && ps == 7) { prod_row_b2(tc, te); pin2(pt[3]); pin2(pt[4]); }
                __builtin_amdgcn_sched_barrier(0);
                // gap 2
                mf(acc[p0], ur[s0][1], v[0][1]);
                if (vread) vn[0] = *reinterpret_cast<const f32x4*>(vnb);
                else { vn[0] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(vn[0])); }
                if (pact && ps == 6) { prod_row_a3(); pin2(pt[0]); }
                __builtin_amdgcn_sched_barrier(0);
                // gap 3
                mf(acc[p1], ur[s1][1], v[1][1]);
                if (vread) vn[1] = *reinterpret_cast<const f32x4*>(vnb + NT * CK);
                else { vn[1] = f32x4{4.f, 3.f, 2.f, 1.f}; asm volatile("" : "+v"(vn[1])); }
                if (pact && ps == 6) { prod_row_a4(); pin2(pt[5]); }
                __builtin_amdgcn_sched_barrier(0);
                // gap 4
                mf(acc[p0], ur[s0][2], v[0][2]);
                if (LOADER && pp <= XLAST) load_x_piece(2 * pp);
                int vst = 0;
                if (pact && ps == 7) vst = prod_addr(ph);
                __builtin_amdgcn_sched_barrier(0);
                // gap 5
                mf(acc[p1], ur[s1][2], v[1][2]);
                if (LOADER && pp <= XLAST && 2 * pp + 1 < PPW) load_x_piece(2 * pp + 1);
                if (pact && ps == 7) prod_store_at(vbn, vst);
                __builtin_amdgcn_sched_barrier(0);
                // gap 6
                mf(acc[p0], ur[s0][3], v[0][3]);
                if (lo) uload<0>(ur[s0], ub, uvoff); else uload<2048>(ur[s0], ub, uvoff);
                // (the table entries of the NEXT pair's two pieces, behind this pair's: one slot each)
                if (LOADER && pp + 1 <= XLAST && 2 * (pp + 1) >= KREG) xoj_fetch(2 * (pp + 1), 0);
                if (pact && ps < 6) { prod_fma_a(psc, pd); pin2(pt[psc]); }
                __builtin_amdgcn_sched_barrier(0);
                // gap 7
                mf(acc[p1], ur[s1][3], v[1][3]);
                if (BMC_W4_XP != 3) { if (lo) uload<1024>(ur[s1], ub, uvoff); else uload<3072>(ur[s1], ub, uvoff); }
                if (LOADER && pp + 1 <= XLAST && 2 * (pp + 1) + 1 >= KREG && 2 * (pp + 1) + 1 < PPW) xoj_fetch(2 * (pp + 1) + 1, 1);
                if (pact && ps < 6) { prod_fma_b(psc, pd); pin2(pt[psc]); }
                // (the patch reads of tiles 4-7 / 12-15 take the other half of their quads: prow_shift)
                if (!LOADER && !(BMC_W4_ABL & 16)) {
                    if (pp == PH0 - 1) prow_shift(1);
                    if (pp == PH0 + 7) prow_shift(-2);
                    if (pp == PH0 + 15) prow_shift(1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (LOADER) {
                    if (pp == XLAST + 1) load_x_advance();
                    if (pp > XLAST && xl_rebuild) {       // a new tile: its table, two pieces per pair
                        const int k = 2 * (pp - XLAST - 1);
                        if (k < PPW) table_piece(k);
                        if (k + 1 < PPW) table_piece(k + 1);
                    }
                }
            }
#else
            __builtin_amdgcn_sched_barrier(0);
            if (pp < NPOS / 2 - 1) read_v(vb, pp + 1, vf[(pp + 1) & 1]);
            if (pact && ps < 6) prod_col(xbn, ph, ps, pd);
            if (pp == 1) W4_WSTAMP(gcx, 1);
            if (pp == 7) W4_WSTAMP(gcx, 2);
            if (pp == 12) W4_WSTAMP(gcx, 3);
            if (pp == NPOS / 2 - 1) W4_WSTAMP(gcx, 4);
            if (pp == NPOS / 2 - 1) {
                // the chunk's barrier sits in front of its LAST pair: V of the next chunk is complete (lgkmcnt(0)), the halo
                // two chunks ahead has landed and is patched; behind it the next chunk's first fragments are read under
                // this pair's MFMAs (their own fragments are in registers since the previous pair)
                if (BMC_W4_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else ring_publish();
                read_v(vbn, 0, vf[0]);
                W4_WSTAMP(gcx, 5);
            }
            if (pp == 0) W4_WSTAMP(gcx, 0);
            __builtin_amdgcn_sched_barrier(0);
#if defined(BMC_W4_EXP) && BMC_W4_EXP == 1      // experiment (ablation builds only): dependent MFMAs 4 apart instead of 2
            if (pp & 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[p0 - 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[(p0 - 2) % D][m], vf[0][0][m], acc[p0 - 2], 0, 0, 0);
                    acc[p0 - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[(p0 - 1) % D][m], vf[0][1][m], acc[p0 - 1], 0, 0, 0);
                    acc[p0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[s0][m], vf[1][0][m], acc[p0], 0, 0, 0);
                    acc[p1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ur[s1][m], vf[1][1][m], acc[p1], 0, 0, 0);
                }
            }
#else
            mfma8(acc[p0], acc[p1], ur[s0], ur[s1], vf[pp & 1]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // the ring slots just consumed take positions p0 + D, p1 + D
            {
                const int q0 = p0 + D;
                if (q0 < NPOS) {
                    const float* const b = ucur + (q0 / 4) * 1024;
                    if (q0 % 4 == 0) { uload<0>(ur[s0], b, uvoff); if (BMC_W4_XP != 3) uload<1024>(ur[s1], b, uvoff); }
                    else { uload<2048>(ur[s0], b, uvoff); if (BMC_W4_XP != 3) uload<3072>(ur[s1], b, uvoff); }
                } else {
                    const float* const b = unx + ((q0 - NPOS) / 4) * 1024;
                    if ((q0 - NPOS) % 4 == 0) { uload<0>(ur[s0], b, uvoff); if (BMC_W4_XP != 3) uload<1024>(ur[s1], b, uvoff); }
                    else { uload<2048>(ur[s0], b, uvoff); if (BMC_W4_XP != 3) uload<3072>(ur[s1], b, uvoff); }
                }
            }
            if (LOADER) {
                // the halo strip two chunks ahead: two pieces behind each of the pairs 0 .. XLAST (the last one: one) -- spread
                // out, a piece costs this wave ~100 cycles between its MFMAs; as one burst the 17 cost 3 000 - 6 800 cycles, and
                // every weight request behind the burst waited for all of it (in-order completion): 7 400 cycles (stamps)
                if (pp == 0) load_x_begin();
                if (pp <= XLAST) { load_x_piece(2 * pp); if (2 * pp + 1 < PPW) load_x_piece(2 * pp + 1); }
                if (pp == XLAST + 1) load_x_advance();
                if (pp > XLAST && xl_rebuild) {       // a new tile: its table, two pieces per pair
                    const int k = 2 * (pp - XLAST - 1);
                    if (k < PPW) table_piece(k);
                    if (k + 1 < PPW) table_piece(k + 1);
                }
            }
            if (pact) {
                if (ps < 6) prod_fma(ps, pd);
                if (ps == 6) prod_row_a(ta, tb, tc, te);
                if (ps == 7) { prod_row_b(ta, tb, tc, te); prod_store(vbn, ph); }
            }
            if (!LOADER && !(BMC_W4_ABL & 16)) {
                if (pp == PH0 - 1) prow_shift(1);
                if (pp == PH0 + 7) prow_shift(-2);
                if (pp == PH0 + 15) prow_shift(1);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    int ep_b = -1;
    const float* ep_res = nullptr;
    const float* ep_mask = nullptr;
#if BMC_W4_EPI
    // Output transform Y = A^T M A (rows of A^T: (1 1 1 1 1 0) (0 1 -1 2 -2 0) (0 1 1 4 4 0) (0 1 -1 8 -8 1)) + bias / residual / ReLU / mask /
    // accumulate + stores.  Columns first, in place; then ROW BY ROW: the row's operand loads (residual / mask / previous output) go out
    // first and land under the row's arithmetic, and its four stores follow at once -- the 16 stores of a wave leave over the length of
    // the row pass instead of as one burst behind it.  (One burst of 8 x 16 KB per workgroup occupies the CU's one in-order vector-memory
    // path for ~2 k cycles, and every U request of the next tile's first pairs queues behind it: without the stores the kernel ran 12 %
    // fewer cycles, twice the epilogue's own length -- round 6's ablation at the sustained clock.)  Every element sees the same
    // operations in the same order as before: bit-identical results.
    auto epilogue = [&](const W4Tile& it) __attribute__((always_inline)) {
        pin_acc();
        if (!(BMC_W4_ABL & 128)) {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const f32x4 m0 = acc[nu], m1 = acc[6 + nu], m2 = acc[12 + nu], m3 = acc[18 + nu], m4 = acc[24 + nu], m5 = acc[30 + nu];
                const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                acc[nu] = (m0 + s1) + s2;
                acc[6 + nu] = d1 + 2.f * d2;
                acc[12 + nu] = s1 + 4.f * s2;
                acc[18 + nu] = (d1 + 8.f * d2) + m5;
            }
        }
        pin_acc();
        __builtin_amdgcn_sched_barrier(0);          // (phases are not interleaved: every one of them alone fits the register file)
        const int g = group_of(it.b);
        const float* const biasg = a.bias ? a.bias + (long long)g * a.bias_group_stride : nullptr;
        float* const outb = a.out + (long long)it.b * a.out_batch_stride;
        if (it.b != ep_b) {
            ep_b = it.b;
            ep_res = a.residual.ptr ? batch_ptr(a.residual, it.b) : nullptr;
            ep_mask = a.mask.ptr ? batch_ptr(a.mask, it.b) : nullptr;
        }
        const float* const resb = ep_res;
        const float* const maskb = ep_mask;
        // (lane index re-derived: the epilogue's lane-dependent values must not be kept -- or spilled -- across the chunks; a
        //  scratch reload inside the chunk loop waits, in order, for every weight request in flight)
        int eln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(eln));
        const int elj = eln & 15, elk = eln >> 4;
        const int co = it.nt * BN + 16 * wave + 4 * elk;
        const bool cok = co < a.Cout;
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (biasg && cok) bq = ldg16(biasg + co);
        const int T = it.wt * NT + elj;
        const int ty = T / tiles_x_now(), tx = T - ty * a.tiles_x;
        const int y0 = 4 * ty, x0 = 4 * tx;
        const int pix0 = y0 * a.W + x0;
        // pixel (i, j) of the tile: in the image?  (tiles past the image's last one have ty >= tiles_y: y0 >= H)
        auto pok = [&](int i, int j) { return cok && y0 + i < a.H && x0 + j < a.W; };
        // (no per-lane branches in here: a pixel outside the image LOADS the operand's first quad -- whatever it computes is not
        //  kept -- and STORES into a 16-byte trash word; with `if (inside) ...` around every access the row-by-row form is a maze of
        //  exec-mask blocks through which the register allocator spills 70 accumulator registers)
        auto row_loads = [&](int i, const float* base, int stride, float, f32x4 (&dd)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dd[j] = ldg16(base + (pok(i, j) ? (pix0 + i * a.W + j) * stride + co : 0));
        };
        // (a zero the compiler cannot see through, defined behind row i's arithmetic: added into the row's pixel index it keeps the
        //  address arithmetic of the row's stores BEHIND that arithmetic -- hoisted to the top of the epilogue, as the scheduler
        //  prefers, the 16 64-bit addresses are live across both transform passes and spill)
        auto late0 = [&](int i) __attribute__((always_inline)) {
            int z = 0;
            asm volatile("" : "+v"(z) : "v"(acc[4 * i + 3]));
            return z;
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(BMC_W4_ABL & 128)) {
                const f32x4 m0 = acc[6 * i], m1 = acc[6 * i + 1], m2 = acc[6 * i + 2], m3 = acc[6 * i + 3], m4 = acc[6 * i + 4], m5 = acc[6 * i + 5];
                const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                acc[4 * i] = (m0 + s1) + s2;          // (4 i + j <= 6 i + j: never overwrites an unread input of a later row)
                acc[4 * i + 1] = d1 + 2.f * d2;
                acc[4 * i + 2] = s1 + 4.f * s2;
                acc[4 * i + 3] = (d1 + 8.f * d2) + m5;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * i + j] += bq;
            if (resb) {
                f32x4 dd[4];
                row_loads(i, resb, a.residual.pix_stride, 0.f, dd);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] += dd[j];
            }
            if (a.relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[4 * i + j][k] = fmaxf(acc[4 * i + j][k], 0.f);
            }
            if (maskb) {
                f32x4 dd[4];
                row_loads(i, maskb, a.mask.pix_stride, 1.f, dd);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[4 * i + j][k] = dd[j][k] > 0.f ? acc[4 * i + j][k] : 0.f;
            }
            if (a.accumulate) {
                f32x4 dd[4];
                row_loads(i, outb, a.out_pix_stride, 0.f, dd);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] += dd[j];
            }
            const int lz = late0(i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float* const dst = pok(i, j) ? outb + ((pix0 + i * a.W + j + lz) * a.out_pix_stride + co) : g_w4_trash;
                if (!((BMC_W4_ABL & 8) && acc[4 * i + j][0] != 12345.678f)) stg16(dst, acc[4 * i + j]);
            }
        }
        init_acc();
    };

#else
    // Y = A^T M A (rows of A^T: (1 1 1 1 1 0) (0 1 -1 2 -2 0) (0 1 1 4 4 0) (0 1 -1 8 -8 1)), columns first, in place: that frees the
    // accumulators of the grid rows xi = 4, 5 -- and the FIRST of the launch's epilogue operands (residual, else mask, else previous
    // output) is requested right there, all 16 quads, to land under the ~400 vector instructions of the row pass.  (Rounds 4-5 fetched
    // it behind the arithmetic, four loads at a time: four memory round trips per tile with both waves of every SIMD waiting -- the
    // residual form of the 2B launch took 0.351 ms against 0.321 for the plain one.)  Pixels outside the image read the operand's
    // first quad; nothing computed from it is stored.  Same operations in the same order per element: bit-identical results.
    auto epilogue = [&](const W4Tile& it) __attribute__((always_inline)) {
        pin_acc();
        if (!(BMC_W4_ABL & 128)) {
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                const f32x4 m0 = acc[nu], m1 = acc[6 + nu], m2 = acc[12 + nu], m3 = acc[18 + nu], m4 = acc[24 + nu], m5 = acc[30 + nu];
                const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                acc[nu] = (m0 + s1) + s2;
                acc[6 + nu] = d1 + 2.f * d2;
                acc[12 + nu] = s1 + 4.f * s2;
                acc[18 + nu] = (d1 + 8.f * d2) + m5;
            }
        }
        const int g = group_of(it.b);
        const float* const biasg = a.bias ? a.bias + (long long)g * a.bias_group_stride : nullptr;
        float* const outb = a.out + (long long)it.b * a.out_batch_stride;
        if (it.b != ep_b) {
            ep_b = it.b;
            ep_res = a.residual.ptr ? batch_ptr(a.residual, it.b) : nullptr;
            ep_mask = a.mask.ptr ? batch_ptr(a.mask, it.b) : nullptr;
        }
        const float* const resb = ep_res;
        const float* const maskb = ep_mask;
        // (lane index re-derived: the epilogue's lane-dependent values must not be kept -- or spilled -- across the chunks; a
        //  scratch reload inside the chunk loop waits, in order, for every weight request in flight)
        int eln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(eln));
        const int elj = eln & 15, elk = eln >> 4;
        const int co = it.nt * BN + 16 * wave + 4 * elk;
        const bool cok = co < a.Cout;
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (biasg && cok) bq = ldg16(biasg + co);
        const int T = it.wt * NT + elj;
        const int ty = T / tiles_x_now(), tx = T - ty * a.tiles_x;
        const int y0 = 4 * ty, x0 = 4 * tx;
        const int pix0 = y0 * a.W + x0;
        // pixel (i, j) of the tile: in the image?  (tiles past the image's last one have ty >= tiles_y: y0 >= H)
        auto pok = [&](int i, int j) { return cok && y0 + i < a.H && x0 + j < a.W; };
        const bool lead_any = BMC_W4_EPF && (resb || maskb || a.accumulate);
        const float* const lbase = resb ? resb : (maskb ? maskb : outb);
        const int lstride = resb ? a.residual.pix_stride : (maskb ? a.mask.pix_stride : a.out_pix_stride);
        // (loader waves: the 12 halo offsets they keep in registers are recomputed from the table at the END of this epilogue instead
        //  of living through it -- 12 LDS reads per tile for 12 registers here; the table and the source stride are those the
        //  stream's next chunk uses)
        constexpr int NPF = BMC_W4_EPF;                 // tile rows of the lead operand requested ahead (their registers: the freed accumulators)
        f32x4 pf[4 * (NPF > 0 ? NPF : 1)];
        if (lead_any) {
#pragma unroll
            for (int i = 0; i < NPF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pf[4 * i + j] = ldg16(lbase + (pok(i, j) ? (pix0 + i * a.W + j) * lstride + co : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(BMC_W4_ABL & 128)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 m0 = acc[6 * i], m1 = acc[6 * i + 1], m2 = acc[6 * i + 2], m3 = acc[6 * i + 3], m4 = acc[6 * i + 4], m5 = acc[6 * i + 5];
                const f32x4 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                acc[4 * i] = (m0 + s1) + s2;          // (4 i + j <= 6 i + j: never overwrites an unread input of a later row)
                acc[4 * i + 1] = d1 + 2.f * d2;
                acc[4 * i + 2] = s1 + 4.f * s2;
                acc[4 * i + 3] = (d1 + 8.f * d2) + m5;
            }
        }
        auto fetch = [&](int i0, const float* base, int stride, float fill, auto&& apply) __attribute__((always_inline)) {
            // one tile row at a time: 4 loads, then their use (16 registers in flight)
#pragma unroll
            for (int i = i0; i < 4; ++i) {
                f32x4 dd[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dd[j] = f32x4{fill, fill, fill, fill};
                    if (pok(i, j)) dd[j] = ldg16(base + ((pix0 + i * a.W + j) * stride + co));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) apply(acc[4 * i + j], dd[j]);
            }
        };
        auto add_to = [](f32x4& v, const f32x4& d) { v += d; };
        auto mask_by = [](f32x4& v, const f32x4& d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = d[k] > 0.f ? v[k] : 0.f;
        };
        // the lead operand: rows 0 .. NPF - 1 from the registers requested ahead, the rest as before
        auto lead_apply = [&](const float* base, int stride, float fill, auto&& apply) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < 4 * NPF; ++p) apply(acc[p], pf[p]);
            fetch(NPF, base, stride, fill, apply);
        };
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] += bq;
        if (resb) {
            if (lead_any) lead_apply(resb, a.residual.pix_stride, 0.f, add_to);
            else fetch(0, resb, a.residual.pix_stride, 0.f, add_to);
        }
        if (a.relu) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[p][k] = fmaxf(acc[p][k], 0.f);
        }
        if (maskb) {
            if (lead_any && !resb) lead_apply(maskb, a.mask.pix_stride, 1.f, mask_by);
            else fetch(0, maskb, a.mask.pix_stride, 1.f, mask_by);
        }
        if (a.accumulate) {
            if (lead_any && !resb && !maskb) lead_apply(outb, a.out_pix_stride, 0.f, add_to);
            else fetch(0, outb, a.out_pix_stride, 0.f, add_to);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (pok(i, j) && !((BMC_W4_ABL & 8) && acc[4 * i + j][0] != 12345.678f))
                    stg16(outb + ((pix0 + i * a.W + j) * a.out_pix_stride + co), acc[4 * i + j]);
        init_acc();
        if (LOADER && BMC_W4_EPF) offsets_from_table();
    };

#endif
    // ---------------------------------------------------------------- prologue
    const W4Tile it0 = decode(t_first);
    const float* ublk = ublock(it0, 0);
    if (LOADER) {
        xl_setup();
        load_x_all();                                 // chunk 0 -> buffer 0
    }
    // the first D positions of U
#pragma unroll
    for (int p = 0; p < D; p += 2) {
        const float* const b = ublk + (p / 4) * 1024;
        if (p % 4 == 0) { uload<0>(ur[p], b, uvoff); uload<1024>(ur[p + 1], b, uvoff); }
        else { uload<2048>(ur[p], b, uvoff); uload<3072>(ur[p + 1], b, uvoff); }
    }
    if (LOADER) {
        dma_wait<D>();                                // the first burst (older than the D requests) has landed
        zero_x(0);
        load_x_all();                                 // chunk 1 -> buffer 1
    }
    __syncthreads();
    if (!LOADER) {
        prod_setup(it0);
        if (!(BMC_W4_ABL & 16)) {
            f32x2p pd[4], ta, tb, tc, te;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                prow_shift(h == 0 ? 1 : -2);
#pragma unroll
                for (int c = 0; c < 6; ++c) { prod_col(Xb, h, c, pd); prod_fma(c, pd); }
                prod_row_a(ta, tb, tc, te);
                prod_row_b(ta, tb, tc, te);
                prod_store(Vb, h);
            }
            prow_shift(1);
        }
    } else {
        dma_wait<0>();
        zero_x(1);
    }
    __syncthreads();
    init_acc();
    read_v(Vb, 0, vf[0]);
    W4_STAMP(1);
    [[maybe_unused]] int stamp_i = 2;

    // ---------------------------------------------------------------- main loop
    // The streams never stop: past the workgroup's last tile the loaders and the producers work on that tile again (valid
    // addresses, buffers nobody reads), so no pair carries an "is there a next one" branch and every wait is a constant.
    int gc = 0;
    W4Tile it = it0;
    for (int tile = t_first; tile < t_hi; tile += t_stride) {
        const W4Tile it_n = tile + t_stride < t_hi ? advance(it) : it;
        for (int c = 0; c < nchunks; ++c, ++gc) {
            const float* unx;
            if (c + 1 < nchunks) unx = ublk + UCH;
            else unx = ublock(it_n, 0);
            if (!LOADER && c == nchunks - 1) prod_setup(it_n);           // the producer moves on to the next tile's first chunk
            chunk(ublk, unx, Vb + (gc & 1) * VBUF, Xb + ((gc + 1) & 1) * XBUFA, Vb + ((gc + 1) & 1) * VBUF, gc & 1, gc);
            ublk = unx;
        }
        W4_STAMP(stamp_i); ++stamp_i;
        epilogue(it);
        W4_STAMP(stamp_i); ++stamp_i;
        it = it_n;
    }
    dma_wait<0>();
}

__global__ __launch_bounds__(512, 2) void wino4_conv_kernel(const ConvK a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * XBUFA + 2 * VBUF + NLW * PPW * 64 + BMC_MAX_SRC * 8];
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + 2 * XBUFA + 2 * VBUF + NLW * PPW * 64);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BMC_W4_STAMP
    if (tid == 0) { g_w4_stamp[blockIdx.x][0] = __builtin_amdgcn_s_memtime(); g_w4_stamp[blockIdx.x][14] = __builtin_amdgcn_s_memrealtime(); }
#endif
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    __syncthreads();

    const int ntiles = a.ntiles;
    constexpr int NX_ = 8;
    const bool xcd_map = (gridDim.x % NX_) == 0 && ntiles >= (int)gridDim.x;
    const int xcd = blockIdx.x % NX_, xj = blockIdx.x / NX_, per_x = gridDim.x / NX_;
    const int t_lo = xcd_map ? (int)((long long)ntiles * xcd / NX_) : 0;
    const int t_hi = xcd_map ? (int)((long long)ntiles * (xcd + 1) / NX_) : ntiles;
    const int t_first = xcd_map ? t_lo + xj : (int)blockIdx.x;
    const int t_stride = xcd_map ? per_x : (int)gridDim.x;
    if (t_first >= t_hi) return;
#if BMC_W4_PRIO
    if ((BMC_W4_PRIO >> wave) & 1) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for a set of waves (bit w = wave w)
#endif
    if (wave >= 8 - NLW) wino4_body<true>(a, lds, wave, t_first, t_hi, t_stride);
    else wino4_body<false>(a, lds, wave, t_first, t_hi, t_stride);
#ifdef BMC_W4_STAMP
    if (tid == 0) { g_w4_stamp[blockIdx.x][13] = __builtin_amdgcn_s_memtime(); g_w4_stamp[blockIdx.x][15] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// U = G g G^T (6x6 from 3x3, G rows (1/4 0 0) (-1/6 -1/6 -1/6) (-1/6 1/6 -1/6) (1/24 1/12 1/6) (1/24 -1/12 1/6) (0 0 1)), made in
// double and rounded once, in the kernel's streaming order out[g][ntile][K chunk][wave 8][position 36][lane 64][4]: lane l of
// wave w holds row co = 128 ntile + 16 w + (l & 15), channels k = 16 chunk + 4 (l >> 4) + 0..3.  Row / K conventions as
// pack_wino_kernel (wino.hip): forward rows = output channels, K = packed input channels (kmap); transposed (data gradient
// w.r.t. packed source channels [k0, k0 + nk)): rows = n = k - k0, K = output channels, taps mirrored.
__global__ void pack_wino4_kernel(const float* __restrict__ w, const int* __restrict__ kmap, int G, int Cout, int Cin, int kpad,
                                  int rows_pad, int transposed, int k0, int nk, float* __restrict__ out) {
    const long long total = (long long)G * (kpad / CK) * rows_pad * CK;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = idx % CK;
        long long r = idx / CK;
        const int row = r % rows_pad; r /= rows_pad;
        const int chunk = r % (kpad / CK);
        const int g = r / (kpad / CK);
        const int k = chunk * CK + kk;
        double gt[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) gt[i / 3][i % 3] = 0.0;
        int co, ci;
        if (!transposed) { co = row; ci = kmap ? kmap[k] : (k < Cin ? k : -1); }
        else { co = k; ci = row < nk ? (kmap ? kmap[k0 + row] : k0 + row) : -1; }
        if (co < Cout && ci >= 0) {
            const float* p = w + (((long long)g * Cout + co) * Cin + ci) * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) gt[i / 3][i % 3] = transposed ? p[8 - i] : p[i];
        }
        const double Gm[6][3] = {{0.25, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                                 {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
        double gg[6][3];
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int b = 0; b < 3; ++b) gg[x][b] = Gm[x][0] * gt[0][b] + Gm[x][1] * gt[1][b] + Gm[x][2] * gt[2][b];
        const int ntile = row / BN, wv = (row % BN) / 16, l = (row & 15) + 16 * (kk >> 2);
        float* const o = out + ((((long long)g * (rows_pad / BN) + ntile) * (kpad / CK) + chunk) * 8 + wv) * UBLK + l * 4 + (kk & 3);
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int n = 0; n < 6; ++n)
                o[(x * 6 + n) * 256] = (float)(gg[x][0] * Gm[n][0] + gg[x][1] * Gm[n][1] + gg[x][2] * Gm[n][2]);
    }
}

}  // namespace

// Called by bmc_conv (conv.hip) for math == BMC_MATH_FP32_WINO4 once the argument block is validated.
int bmc_conv_wino4_launch(ConvK k, int cus, hipStream_t st) {
    k.tiles_x = (k.W + 3) / 4;
    k.tiles_y = (k.H + 3) / 4;
    k.ntn = k.Coutpad / BN;
    if (k.tiles_x < 5) { bmc_set_error("bmc_conv (winograd F(4x4)): images narrower than 17 pixels are not served"); return -1; }
    long long max_stride = k.out_pix_stride;
    for (int i = 0; i < k.nsrc; ++i) max_stride = k.src[i].pix_stride > max_stride ? k.src[i].pix_stride : max_stride;
    if (k.residual.ptr && k.residual.pix_stride > max_stride) max_stride = k.residual.pix_stride;
    if (k.mask.ptr && k.mask.pix_stride > max_stride) max_stride = k.mask.pix_stride;
    if ((long long)k.H * k.W * max_stride * 4 >= (1ll << 31) || (long long)k.H * k.W >= (1ll << 24)) {
        bmc_set_error("bmc_conv (winograd F(4x4)): image too large for 32-bit offsets (H*W*pix_stride*4 must stay below 2^31)");
        return -1;
    }
    const long long wpi = ((long long)k.tiles_x * k.tiles_y + NT - 1) / NT;
    const long long ntiles = (long long)k.B * wpi * k.ntn;
    if (ntiles >= (1ll << 31)) { bmc_set_error("bmc_conv (winograd F(4x4)): too many tiles"); return -1; }
    k.ntiles = (int)ntiles;
#ifdef BMC_W4_STAMP
    static const int grid_cap = getenv("BMC_W4_GRID") ? atoi(getenv("BMC_W4_GRID")) : 0;      // (diagnostic builds: fewer workgroups than CUs)
    if (grid_cap > 0 && grid_cap < cus) cus = grid_cap;
#endif
    dim3 grid((unsigned)(ntiles < cus ? ntiles : cus));
    hipLaunchKernelGGL(wino4_conv_kernel, grid, dim3(512), 0, st, k);
    return 0;
}

#ifdef BMC_W4_STAMP
extern "C" int bmc_w4_read_stamps(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_stamp), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : -1;
}
extern "C" int bmc_w4_read_wstamps(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_wstamp), sizeof(unsigned long long) * 8 * 24 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int bmc_pack_weight_wino4(const float* w, const int* kmap, int G, int Cout, int Cin, int Kpad, int Coutpad,
                                     int transposed, int k0, int nk, float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(w && out && G >= 1 && Cout >= 1 && Cin >= 1, "bmc_pack_weight_wino4: bad arguments");
    BMC_CHECK_ARG(Kpad > 0 && Kpad % CK == 0 && Coutpad > 0 && Coutpad % 128 == 0,
                  "bmc_pack_weight_wino4: Kpad must be a multiple of 16 and the row count a multiple of 128");
    BMC_CHECK_ARG(!transposed || (k0 >= 0 && nk >= 1 && nk <= Coutpad && Kpad >= Cout), "bmc_pack_weight_wino4: bad transposed window");
    const long long total = (long long)G * (Kpad / CK) * Coutpad * CK;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)s, w, kmap, G,
                       Cout, Cin, Kpad, Coutpad, transposed, k0, nk, out);
    BMC_CHECK_LAUNCH("bmc_pack_weight_wino4");
    return 0;
}
