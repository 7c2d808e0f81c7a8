// Implicit-GEMM 3x3 / 1x1 convolution on the bf16 matrix cores of gfx950
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate), same operator as conv.hip:
// NHWC fp32 activations in HBM, multi-source, fused epilogue, per-group weights.
//
// NP (planes) selects the arithmetic:
//   NP = 1  operands rounded to bf16 (RNE) on their way into LDS: the "bf16 compute / fp32 accumulate" mode;
//   NP = 3  every fp32 operand is split EXACTLY into three bf16 planes x = h + m + l (8 significand bits each) and the
//           product is rebuilt from the six plane products of weight >= 2^-16: hh, hm, mh, hl, lh, mm.  The three
//           dropped ones (ml, lm, ll) are <= 2^-23 |x w| together in the worst case (~2^-26 rms), the size of ONE fp32
//           rounding of the product;
//           bf16 x bf16 products are exact in fp32 and the matrix core accumulates in fp32.  The bf16 pipe runs 16x
//           the fp32 MFMA rate, so six products cost 6/16 of the native fp32 time.
// The split happens once per element, when a tile goes from registers to LDS (activations), or once per step on the
// packed weights (bmc_split_weight); LDS holds bf16 planes only.
//
// LDS image: per plane [row][16 bf16] = 32 B rows without padding.  An MFMA fragment read is one ds_read_b128 per lane
// (lane l: row of l & 31, 16-byte half l >> 5); the hardware serves it in 16-lane groups {0-3,12-15,20-27} /
// {4-11,16-19,28-31} over 64 banks (256 B), so each group takes 8 rows from lanes 0-15 and 8 from lanes 16-31 that
// fall on the same 8 bank slots.  Swapping the two halves of every row that lanes 16-31 read (weights: rows with bit 4
// set; pixels: odd halo rows) makes all fragment reads conflict-free.
#include "conv_k.h"
#include "bf_split.h"

#ifdef BMC_BF_STAMP
// experiment builds only (tools/): cycle totals of the 1x1 kernel's step phases, wave 0 (halo role) and wave 2 (weight role)
__device__ unsigned long long g_cstamp[16 * 1024];
extern "C" int bmc_cstamp_read(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cstamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#define ST(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define ST(v)
#endif
#ifndef BMC_BF_ABL
#define BMC_BF_ABL 0   // ablation bits for experiment builds (tools/): 1 no epilogue stores, 4 no MFMAs, 8 no weight loads, 16 no activation loads (reads a zero constant: the
                       // chip then clocks ~1.3x higher on the all-zero MFMA operands -- not a latency measurement), 32 no halo split + store
#endif

namespace {

constexpr int CK = BMC_CK;
constexpr int TW = 16;
constexpr int RD = 8;   // dwords per LDS row (16 bf16)
__device__ __attribute__((aligned(16))) const float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};   // source of out-of-image lanes

__device__ __forceinline__ int swz_w(int row, int half) { return row * RD + 4 * (half ^ ((row >> 4) & 1)); }
__device__ __forceinline__ int swz_x(int hp, int hy, int half) { return hp * RD + 4 * (half ^ (hy & 1)); }

// 16 bytes per lane from global memory straight into LDS (lane-linear image at the wave-uniform LDS byte address).
// Inline asm on purpose: the compiler must not track this as an LDS store, or it drains vmcnt(0) before every later
// ds_read and the ring could never run ahead.  Completion is waited for explicitly (dma_wait) before the barrier
// that publishes the stage.
// Address = uniform base (SGPR pair) + this lane's 32-bit byte offset: no per-piece VALU.  (m0 is reserved and cannot be
// named as a clobber; nothing else in this kernel uses it.  s_nop 4: wait states for a VALU-written SGPR base / m0.)
__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void dma_wait() {   // all but the newest N vector-memory operations of this wave are done
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// Loader roles are split by wave, because a wave's vmcnt completes IN ORDER: a wave that issued the next chunk's halo
// loads (HBM, needed 8 steps later) and then waits for a weight slice (L2, needed next step) waits for the halo too.
//   waves 0-1: activation halo: global -> registers at a chunk's first tap, split into planes -> LDS at its last tap;
//   waves 2-3: weight slices: LDS-DMA into a 3-stage ring, two steps ahead.
template <int TAPS, int BN, int TH, int NP>
__global__ __launch_bounds__(256, (BN == 128 && (TH == 8 || NP == 3)) ? 2 : 3) void conv_bf_kernel(const ConvK a) {
    static_assert((BN == 32 && TH == 8) || ((BN == 128 || BN == 64) && (TH == 8 || TH == 4)), "unsupported tile shape");
    static_assert(NP == 1 || NP == 3, "planes");
    constexpr int P = TAPS == 9 ? 1 : 0;
    constexpr int HWD = TW + 2 * P, HHT = TH + 2 * P, NHALO = HWD * HHT;
    constexpr int MT = BN == 32 ? 1 : TH / 4, NT = BN == 128 ? 2 : 1;
    constexpr int NXLD = (NHALO * 4 + 127) / 128;        // float4 per loader thread (128 loader threads)
    constexpr int NDMA = NP * BN * 2 / 64;               // 1 KB wave-instructions per weight slice
    constexpr int XPL = NHALO * RD, WPL = BN * RD;       // dwords per plane
    constexpr int XBUF = NP * XPL, WBUF = NP * WPL, NSTG = 3;
    __shared__ __attribute__((aligned(16))) u32 lds[2 * XBUF + NSTG * WBUF + BMC_MAX_SRC * 8 + BN];
    u32* const Xb = lds;
    u32* const Wb = lds + 2 * XBUF;
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + 2 * XBUF + NSTG * WBUF);
    // accumulator start values (bias when it is the same for every tile of the launch, else zeros): see conv.hip
    float* const init_lds = reinterpret_cast<float*>(lds + 2 * XBUF + NSTG * WBUF + BMC_MAX_SRC * 8);
    const bool bias_pre = a.bias != nullptr && a.batch_per_group >= a.B && a.ntn == 1;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool xrole = wave < 2;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    if (tid < BN) init_lds[tid] = (bias_pre && tid < a.Cout) ? a.bias[tid] : 0.f;
    __syncthreads();

    // persistent workgroups with the XCD-aware tile walk of conv.hip
    const int ntiles = a.ntiles;
    constexpr int NX_ = 8;
    const bool xcd_map = (gridDim.x % NX_) == 0 && ntiles >= (int)gridDim.x;
    const int xcd = blockIdx.x % NX_, xj = blockIdx.x / NX_, per_x = gridDim.x / NX_;
    const int t_lo = xcd_map ? (int)((long long)ntiles * xcd / NX_) : 0;
    const int t_hi = xcd_map ? (int)((long long)ntiles * (xcd + 1) / NX_) : ntiles;
    const int t_first = xcd_map ? t_lo + xj : (int)blockIdx.x;
    const int t_stride = xcd_map ? per_x : (int)gridDim.x;
    const int my_tiles = t_first < t_hi ? (t_hi - t_first + t_stride - 1) / t_stride : 0;
    const int nsteps = a.nchunks * TAPS;
    const int total_steps = my_tiles * nsteps, total_chunks = my_tiles * a.nchunks;
    if (my_tiles == 0) return;
    const long long wstep = (long long)NP * a.Coutpad * RD;   // dwords per step in the packed planes

    // Tile index -> (channel tile, tile column, tile row, image) is a mixed-radix decode = three integer divisions, ~100
    // VALU instructions that three users (halo loader, weight loader, epilogue) would pay per tile beside the MFMAs.
    // A workgroup visits t_first, t_first + t_stride, ...: decode once, then advance digit-wise with carries.
    struct TileIt { int nt, tx, ty, b; };
    TileIt it0;
    {
        int t = t_first;
        it0.nt = t % a.ntn; t /= a.ntn;
        it0.tx = t % a.tiles_x; t /= a.tiles_x;
        it0.ty = t % a.tiles_y;
        it0.b = t / a.tiles_y;
    }
    int d_nt, d_tx, d_ty, d_b;
    {
        int t = t_stride;
        d_nt = t % a.ntn; t /= a.ntn;
        d_tx = t % a.tiles_x; t /= a.tiles_x;
        d_ty = t % a.tiles_y;
        d_b = t / a.tiles_y;
    }
    auto it_next = [&](TileIt& it) {
        it.nt += d_nt;
        int c = 0;
        if (it.nt >= a.ntn) { it.nt -= a.ntn; c = 1; }
        it.tx += d_tx + c; c = 0;
        if (it.tx >= a.tiles_x) { it.tx -= a.tiles_x; c = 1; }
        it.ty += d_ty + c; c = 0;
        if (it.ty >= a.tiles_y) { it.ty -= a.tiles_y; c = 1; }
        it.b += d_b + c;
    };
    TileIt xl_it = it0, wl_it = it0, ep_it = it0;

    // ---- X loader (waves 0-1): fp32 from HBM into registers; split into planes when written to LDS
    const int xt = tid & 127, q = tid & 3;
    int xl_tile = t_first, xl_chunk = 0, xl_b = 0, s_idx = 0, c_in = 0;
    const float* sbase = nullptr;
    int spix = 0, snch = 0;
    int xpix[NXLD];
    bool xok[NXLD];
    // Fast path (interior tiles: every halo pixel this wave stages lies in the image): the load address is a uniform
    // base + a per-lane 32-bit byte offset that is fixed for the (tile, source) -> no per-load VALU at all.  Lanes past
    // the halo (their data is never stored) read pixel 0.
    unsigned xoffb[NXLD];
    bool x_fast = false, xneed[NXLD];
    auto src_select = [&]() {
        const SrcDev S = tab[s_idx];
        sbase = src_batch_ptr(S, xl_b); spix = S.pix_stride; snch = S.nch;
#pragma unroll
        for (int n = 0; n < NXLD; ++n) xoffb[n] = TAPS == 1 ? (unsigned)(((xneed[n] ? xpix[n] * spix : 0) + q * 4) * 4) : 0u;
    };
    auto xl_setup = [&]() {      // for the tile xl_it points at
        const int y0 = xl_it.ty * TH, x0 = xl_it.tx * TW;
        xl_b = xl_it.b;
#pragma unroll
        for (int n = 0; n < NXLD; ++n) {
            const int e = xt + 128 * n, hp = e >> 2;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - P + hy, x = x0 - P + hx;
            xok[n] = hp < NHALO && y >= 0 && y < a.H && x >= 0 && x < a.W;   // out-of-image lanes read a zero buffer: no
            xpix[n] = y * a.W + x;                                           // branch, and nothing to fix up after the load
            xneed[n] = hp < NHALO;
        }
        bool allin = true;
#pragma unroll
        for (int n = 0; n < NXLD; ++n) allin = allin && (xok[n] || !xneed[n]);
        // (1x1 only: there a step is short and the halo waves' instruction count is what limits it; the 3x3 kernel measured
        //  slower with the second code path, 0.556 -> 0.587 ms)
        x_fast = TAPS == 1 && __all(allin);
        s_idx = 0; c_in = 0; xl_chunk = 0;
        src_select();
    };
    // TAPS = 1: a step is one 16-channel chunk straight from HBM, shorter than the memory latency, so the loads run XD
    // steps ahead through a register ring (slots are static: the step loop is unrolled by XD)
    // The ring's loads are issued through inline asm and waited for by hand (wait_x): left to the compiler, the wait in
    // front of a slot's ds_write comes out as vmcnt(0) (it cannot count across this loop's branches), i.e. a ring of depth 1.
    constexpr int XD = TAPS == 1 ? 3 : 1;
    constexpr bool ASMX = TAPS == 1;    // 3x3: ordinary loads; the compiler's wait (everything, at the first store) comes 3+ steps after the issue
    f32x4 xr[XD][NXLD];
    auto load_x = [&](int slot) {
        // (the byte offsets of one image of this source must fit the instruction's unsigned 32-bit lane offset)
        if (TAPS == 1 && x_fast && (long long)a.H * a.W * spix < (1ll << 29) && !(BMC_BF_ABL & 16)) {
            // uniform, but it came through LDS (the source table): tell the compiler, so that it can live in SGPRs
            const unsigned long long sbv = reinterpret_cast<unsigned long long>(sbase + c_in);
            const unsigned sb_lo = __builtin_amdgcn_readfirstlane((unsigned)sbv), sb_hi = __builtin_amdgcn_readfirstlane((unsigned)(sbv >> 32));
            const float* const sb = reinterpret_cast<const float*>(((unsigned long long)sb_hi << 32) | sb_lo);
#pragma unroll
            for (int n = 0; n < NXLD; ++n) {
                // (s_nop: a VALU-written SGPR -- the readfirstlane above -- needs 5 wait states before a VMEM instruction may
                //  use it as its scalar base, and the hazard recognizer does not look inside inline asm)
                if constexpr (ASMX) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(xr[slot][n]) : "v"(xoffb[n]), "s"(sb) : "memory");
                else xr[slot][n] = ldg16(reinterpret_cast<const char*>(sb) + xoffb[n]);
            }
        } else {
            const float* base = sbase + c_in + q * 4;
#pragma unroll
            for (int n = 0; n < NXLD; ++n) {
                const float* src = xok[n] ? base + (long long)xpix[n] * spix : g_zero4;
                if (BMC_BF_ABL & 16) src = g_zero4;
                if constexpr (ASMX) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[slot][n]) : "v"(src) : "memory");
                else xr[slot][n] = ldg16(src);
            }
        }
        c_in += CK;
        if (++xl_chunk == a.nchunks) {
            xl_tile += t_stride;
            if (xl_tile < t_hi) { it_next(xl_it); xl_setup(); }
        } else if (c_in >= snch) {
            c_in = 0; ++s_idx;
            src_select();
        }
    };
    auto pin_x = [&](int slot) {   // nothing that reads the slot may be scheduled above this point
#pragma unroll
        for (int n = 0; n < NXLD; ++n) asm volatile("" : "+v"(xr[slot][n]));
    };
    auto store_x_item = [&](int slot, int buf, int n) {
        if (BMC_BF_ABL & 32) return;
        const int e = xt + 128 * n, hp = e >> 2;
        if ((n + 1) * 128 <= NHALO * 4 || hp < NHALO) {
            u32x2 pl[NP];
            split4<NP>(xr[slot][n], pl);
            u32* const dst = Xb + buf * XBUF + swz_x(hp, hp / HWD, q >> 1) + 2 * (q & 1);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(dst + p * XPL) = pl[p];
        }
    };
    auto store_x = [&](int slot, int buf) {
#pragma unroll
        for (int n = 0; n < NXLD; ++n) store_x_item(slot, buf, n);
    };
    // ---- W loader (waves 2-3): the packed planes are already the LDS image (swizzle included): linear 1 KB pieces
    const int w2 = wave & 1;
    int wl_tile = t_first, wl_step = 0, wl_stage = 0;
    const u32* wl_base = nullptr;
    auto wl_setup = [&]() {      // for the tile wl_it points at
        const int grp = a.batch_per_group >= a.B ? 0 : wl_it.b / a.batch_per_group;
        wl_base = static_cast<const u32*>(a.w) + (long long)grp * a.w_group_stride + (long long)wl_it.nt * BN * RD;
        wl_step = 0;
    };
    const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Wb;
    // this wave issues the pieces j = w2, w2 + 2, ... of a slice; a piece's offset inside the slice never changes
    constexpr int PWMAX = (NDMA + 1) / 2;
    unsigned dma_off[PWMAX];
#pragma unroll
    for (int i = 0; i < PWMAX; ++i) {
        const int j = 2 * i + w2, e = (j < NDMA ? j : 0) * 64 + lane;
        const int pl = e / (BN * 2), within = e - pl * (BN * 2);
        dma_off[i] = (unsigned)((pl * a.Coutpad * RD + within * 4) * 4);
    }
    auto dma_w = [&]() {   // one slice into ring stage wl_stage
        const unsigned long long pv = reinterpret_cast<unsigned long long>(wl_base + (long long)wl_step * wstep);
        const unsigned p_lo = __builtin_amdgcn_readfirstlane((unsigned)pv), p_hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
        const void* const p = reinterpret_cast<const void*>(((unsigned long long)p_hi << 32) | p_lo);
#pragma unroll
        for (int i = 0; i < PWMAX; ++i) {
            const int j = 2 * i + w2;
            if (j < NDMA && !(BMC_BF_ABL & 8) && !(NDMA == 1 && w2))
                dma16(p, dma_off[i], wb_lds + (unsigned)((wl_stage * WBUF + j * 256) * 4));
        }
        wl_stage = wl_stage == NSTG - 1 ? 0 : wl_stage + 1;
        if (++wl_step == nsteps) {
            wl_tile += t_stride;
            if (wl_tile < t_hi) { it_next(wl_it); wl_setup(); }
        }
    };
    constexpr int DMA_W0 = (NDMA + 1) / 2, DMA_W1 = NDMA / 2;   // pieces per slice issued by loader wave 0 / 1
    auto wait_older_slices = [&]() {   // every slice but the one just issued has landed
        if (w2 == 0) dma_wait<DMA_W0>(); else dma_wait<DMA_W1>();
    };

    // ---- MFMA fragment addressing: A = weights (rows = output channels), B = pixels (cols)
    const int rowbase = BN == 32 ? 2 * wave : (TH / 2) * (wave >> 1);
    const int cobase = BN == 32 ? 0 : (BN / 2) * (wave & 1);
    int arow[MT], boff[NT];
    const int ahy = rowbase + (li >> 4);   // halo row of this lane's pixel for t = 0, tap row 0 (parity is what matters)
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = (rowbase + 2 * t + (li >> 4)) * HWD + (li & 15);
#pragma unroll
    for (int u = 0; u < NT; ++u) boff[u] = swz_w(cobase + 32 * u + li, lh);

    f32x16 acc[MT][NT];
    auto init_acc = [&]() {     // 16 LDS reads straight into the accumulator registers (no v_mov, no bias adds later)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(init_lds + cobase + 4 * lh + 32 * u + 8 * rq);
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[t][u][4 * rq + k] = v[k];
            }
    };
    init_acc();

    u32x4 xf[NP][MT], wf[NP][NT];
    auto read_frags = [&](const u32* xb, const u32* wb, int tapshift, int taprow) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                xf[p][t] = *reinterpret_cast<const u32x4*>(xb + p * XPL + swz_x(arow[t] + tapshift, ahy + taprow, lh));
#pragma unroll
            for (int u = 0; u < NT; ++u) wf[p][u] = *reinterpret_cast<const u32x4*>(wb + p * WPL + boff[u]);
        }
    };
    auto mma = [&](int pw, int px) {
        if (BMC_BF_ABL & 4) return;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[pw][u]),
                                                                    __builtin_bit_cast(bf16x8, xf[px][t]), acc[t][u], 0, 0, 0);
    };
    auto mma_all = [&]() {   // in the order the fragments arrive from LDS (plane 0 first)
        mma(0, 0);
        if constexpr (NP == 3) { mma(0, 1); mma(1, 0); mma(1, 1); mma(0, 2); mma(2, 0); }
    };
    auto tap_shift = [](int tap) { return TAPS == 9 ? (tap / 3) * HWD + (tap % 3) : 0; };

    // Epilogue: bias / residual / ReLU / mask / accumulate, 16-byte accesses, then clear the accumulators.
    // MFMA rows = output channels (registers), cols = pixels (lanes): each lane owns, for ITS pixel, four consecutive
    // channels per register quad.  ALL loads of the epilogue come before ALL its stores: vmcnt completes in order and
    // counts stores, so a load issued after a store waits for that store's round trip to HBM -- the former
    // (load, wait, store) per quad serialised 16 store round trips per tile.  Pass 1 finishes the values in place, one
    // 32x32 MFMA tile (4 quads = 4 independent loads per operand) at a time; pass 2 is nothing but stores.
    auto epilogue = [&](int) {    // for the tile ep_it points at; advances it
        const int b = ep_it.b, y0 = ep_it.ty * TH, x0 = ep_it.tx * TW, nt = ep_it.nt;
        it_next(ep_it);
        const int g = a.batch_per_group >= a.B ? 0 : b / a.batch_per_group;
        const float* const biasg = a.bias ? a.bias + (long long)g * a.bias_group_stride : nullptr;
        float* const outb = a.out + (long long)b * a.out_batch_stride;
        const float* const resb = a.residual.ptr ? src_batch_ptr(a.residual, b) : nullptr;
        const float* const maskb = a.mask.ptr ? src_batch_ptr(a.mask, b) : nullptr;
        const int co0 = nt * BN + cobase + 4 * lh;       // + 32 u + 8 rq
        bool pok[MT];
        int pix[MT];      // pixel index inside one image: 32-bit offsets from the (uniform) per-image base pointers
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int y = y0 + rowbase + 2 * t + (li >> 4), x = x0 + (li & 15);
            pok[t] = y < a.H && x < a.W;
            pix[t] = y * a.W + x;
        }
        // the usual case -- bias already in the accumulators, nothing to read back -- is one pass: ReLU on the way out
        const bool simple = !resb && !maskb && !a.accumulate && (bias_pre || !biasg);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if (simple) break;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f32x4 v[4];
                bool ok[4];
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    ok[rq] = pok[t] && co0 + 32 * u + 8 * rq < a.Cout;     // Cout is a multiple of 4: a quad is all-in or all-out
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[rq][k] = acc[t][u][4 * rq + k];
                }
                auto fetch = [&](const float* base, int off, f32x4 (&d)[4], float fill) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        d[rq] = f32x4{fill, fill, fill, fill};
                        if (ok[rq]) d[rq] = ldg16(base + off + co0 + 32 * u + 8 * rq);
                    }
                };
                if (biasg && !bias_pre) {     // (otherwise the accumulators started from the bias)
                    f32x4 d[4];
                    fetch(biasg, 0, d, 0.f);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) v[rq] += d[rq];
                }
                if (resb) {
                    f32x4 d[4];
                    fetch(resb, pix[t] * a.residual.pix_stride, d, 0.f);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) v[rq] += d[rq];
                }
                if (a.relu) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[rq][k] = fmaxf(v[rq][k], 0.f);
                }
                if (maskb) {
                    f32x4 d[4];
                    fetch(maskb, pix[t] * a.mask.pix_stride, d, 1.f);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[rq][k] = d[rq][k] > 0.f ? v[rq][k] : 0.f;
                }
                if (a.accumulate) {
                    f32x4 d[4];
                    fetch(outb, pix[t] * a.out_pix_stride, d, 0.f);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) v[rq] += d[rq];
                }
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[t][u][4 * rq + k] = v[rq][k];
            }
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = co0 + 32 * u + 8 * rq;
                    if ((BMC_BF_ABL & 1) && acc[t][u][4 * rq] != 12345.678f) continue;
                    if (pok[t] && co < a.Cout) {
                        f32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = acc[t][u][4 * rq + k];
                        if (simple && a.relu) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                        }
                        *reinterpret_cast<f32x4*>(outb + pix[t] * a.out_pix_stride + co) = v;
                    }
                }
        init_acc();
    };

    // ---- prologue: halo of chunk 0 in LDS, weight slices 0 and 1 in flight, slice 0 landed
    if (xrole) {
        xl_setup();
        load_x(0);
        if (ASMX) { dma_wait<0>(); pin_x(0); }
        store_x(0, 0);
        if (TAPS == 1) {
#pragma unroll
            for (int j = 1; j <= XD; ++j)
                if (j < total_chunks) load_x(j % XD);
        }
    } else {
        wl_setup();
        dma_w();
        if (total_steps > 1) { dma_w(); wait_older_slices(); } else dma_wait<0>();
    }
    __syncthreads();

    // ---- main loop, one barrier per step s:
    //   loaders : halo waves  - first tap: issue the next chunk's global loads; last tap: split + write them to LDS
    //             weight waves - issue the DMA of slice s+2 into ring stage (s+2) % 3
    //   all     : read this step's fragments, MFMAs
    //   weight waves: wait until slice s+1 has landed (slice s+2 may stay in flight); barrier
    int stage = 0;
    if constexpr (TAPS == 1) {
        int tile = t_first, cc = 0, landed = 0;   // landed: upcoming steps whose ring slot is known to be complete
#ifdef BMC_BF_STAMP
        unsigned long long st_stage = 0, st_mma = 0, st_wait = 0, st_bar = 0, st_epi = 0;
#endif
        for (int base = 0; base < total_steps; base += XD) {
#pragma unroll
            for (int d = 0; d < XD; ++d) {
                const int s = base + d;     // chunk s: LDS buffer s & 1; chunk c >= 1 travels through ring slot c % XD
                if (s < total_steps) {
                    ST(c0);
                    if (xrole) {
                        if (s + 1 < total_steps) {
                            // chunk s+1 sits in slot (d+1) % XD; younger loads in flight: chunks s+2 .. s+XD.  After a
                            // tile's epilogue (which drained everything) the next XD-1 steps need no wait at all, and
                            // by the first real wait the epilogue's stores (older, counted by vmcnt) have long drained.
                            if (landed > 0) --landed;
                            else if (s + XD < total_steps) dma_wait<NXLD * (XD - 1)>();
                            else dma_wait<0>();
                            pin_x((d + 1) % XD);
                            store_x((d + 1) % XD, (s + 1) & 1);
                        }
                        if (s + 1 + XD < total_steps) load_x((d + 1) % XD);
                    } else {
                        if (s + 2 < total_steps) dma_w();
                    }
                    ST(c1);
                    read_frags(Xb + (s & 1) * XBUF, Wb + stage * WBUF, 0, 0);
                    mma_all();
                    ST(c2);
                    const bool tile_end = cc + 1 == a.nchunks;
                    if (!xrole) {
                        // vmcnt completes in order and counts the epilogue's global stores: a wait for a slice issued
                        // AFTER them would sit out the whole store drain.  So a tile's last step waits for everything
                        // (slice s+2 included) before the stores go out, and the next step has nothing to wait for.
                        if (tile_end || s + 2 >= total_steps) dma_wait<0>();
                        else if (cc != 0 || s == 0) wait_older_slices();
                    }
                    stage = stage == NSTG - 1 ? 0 : stage + 1;
                    ST(c3);
                    __syncthreads();
                    ST(c4);
                    if (++cc == a.nchunks) {
                        // same for the halo waves: their ring loads are older than the stores about to be issued, but
                        // the compiler's static vmcnt for the next ds_write must also hold on the no-epilogue path and
                        // would sit out half the stores; draining the (nearly landed) loads first costs less
                        if (xrole) { dma_wait<0>(); landed = XD - 1; }
                        epilogue(tile);
                        tile += t_stride;
                        cc = 0;
                    }
#ifdef BMC_BF_STAMP
                    { const unsigned long long c5 = __builtin_amdgcn_s_memtime();
                      st_stage += c1 - c0; st_mma += c2 - c1; st_wait += c3 - c2; st_bar += c4 - c3; st_epi += c5 - c4; }
#endif
                }
            }
        }
#ifdef BMC_BF_STAMP
        if ((tid == 0 || tid == 128) && blockIdx.x < 1024) {
            unsigned long long* o = g_cstamp + (blockIdx.x * 2 + (tid >> 7)) * 8;
            o[0] = st_stage; o[1] = st_mma; o[2] = st_wait; o[3] = st_bar; o[4] = st_epi; o[5] = total_steps; o[6] = my_tiles;
        }
#endif
    } else {
        int gs = 0, gc = 0;
        for (int tile = t_first; tile < t_hi; tile += t_stride) {
            for (int c = 0; c < a.nchunks; ++c, ++gc) {
                const u32* const xb = Xb + (gc & 1) * XBUF;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap, ++gs) {
                    const bool last_tap = tap == TAPS - 1;
                    if (xrole) {
                        // the next chunk's halo: loads go out at the first tap; its NXLD items are split + stored ONE PER
                        // STEP over the last NXLD taps (VALU issued beside a saturated matrix pipe costs ~5x its nominal
                        // cycles: all six items at the last tap made that step several times longer than the others,
                        // with every wave waiting at its barrier).
                        if (tap == 0 && gc + 1 < total_chunks) load_x(0);
                        constexpr int FIRST = TAPS - NXLD;
                        if (tap >= FIRST && gc + 1 < total_chunks) store_x_item(0, (gc + 1) & 1, tap - FIRST);
                    } else {
                        if (gs + 2 < total_steps) dma_w();
                    }
                    read_frags(xb, Wb + stage * WBUF, tap_shift(tap), tap / 3);
                    mma_all();
                    if (!xrole) {   // (see the 1x1 loop: no wait may straddle the epilogue's stores)
                        const bool tile_end = last_tap && c + 1 == a.nchunks, tile_begin = tap == 0 && c == 0 && gs != 0;
                        if (tile_end || gs + 2 >= total_steps) dma_wait<0>();
                        else if (!tile_begin) wait_older_slices();
                    }
                    stage = stage == NSTG - 1 ? 0 : stage + 1;
                    __syncthreads();
                }
            }
            epilogue(tile);
        }
    }
}

// packed fp32 weights [S][Coutpad][16] -> bf16 planes [S][NP][Coutpad][16] in the kernel's LDS image (halves swizzled)
template <int NP>
__global__ void split_weight_kernel(const float* __restrict__ in, u32* __restrict__ out, long long nquads, int Coutpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 4-channel quad
    if (i >= nquads) return;
    const int qd = (int)(i & 3);
    const long long rowg = i >> 2;
    const int row = (int)(rowg % Coutpad);
    const long long s = rowg / Coutpad;
    u32x2 pl[NP];
    split4<NP>(*reinterpret_cast<const f32x4*>(in + i * 4), pl);
#pragma unroll
    for (int p = 0; p < NP; ++p)
        *reinterpret_cast<u32x2*>(out + ((s * NP + p) * Coutpad) * RD + swz_w(row, qd >> 1) + 2 * (qd & 1)) = pl[p];
}

}  // namespace

extern "C" int bmc_split_weight(const float* packed, void* out, long long nsteps, int Coutpad, int planes, bmc_stream_t stream) {
    BMC_CHECK_ARG(packed && out && nsteps > 0 && Coutpad > 0 && Coutpad % 32 == 0, "bmc_split_weight: bad arguments");
    BMC_CHECK_ARG(planes == 1 || planes == 3, "bmc_split_weight: planes must be 1 or 3 (got %d)", planes);
    const long long nquads = nsteps * Coutpad * 4;
    const unsigned blocks = (unsigned)((nquads + 255) / 256);
    if (planes == 1)
        hipLaunchKernelGGL(split_weight_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, (u32*)out, nquads, Coutpad);
    else
        hipLaunchKernelGGL(split_weight_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, (u32*)out, nquads, Coutpad);
    BMC_CHECK_LAUNCH("bmc_split_weight");
    return 0;
}

int bmc_conv_bf_launch(const ConvK& k, int taps, int BN, int TH, int planes, int cus, hipStream_t st) {
    const int per_cu = (BN == 128 && (TH == 8 || planes == 3)) ? 2 : 3;
    const long long max_blocks = (long long)cus * per_cu;
    dim3 grid((unsigned)(k.ntiles < max_blocks ? k.ntiles : max_blocks)), block(256);
#define BMC_LAUNCH_BF(TAPS_, BN_, TH_)                                                                     \
    do {                                                                                                   \
        if (planes == 3) hipLaunchKernelGGL((conv_bf_kernel<TAPS_, BN_, TH_, 3>), grid, block, 0, st, k);  \
        else hipLaunchKernelGGL((conv_bf_kernel<TAPS_, BN_, TH_, 1>), grid, block, 0, st, k);              \
    } while (0)
    if (taps == 9) {
        if (BN == 32) BMC_LAUNCH_BF(9, 32, 8);
        else if (BN == 64) BMC_LAUNCH_BF(9, 64, 4);
        else if (TH == 4) BMC_LAUNCH_BF(9, 128, 4);
        else BMC_LAUNCH_BF(9, 128, 8);
    } else {
        if (BN == 32) BMC_LAUNCH_BF(1, 32, 8);
        else if (BN == 64) BMC_LAUNCH_BF(1, 64, 4);
        else if (TH == 4) BMC_LAUNCH_BF(1, 128, 4);
        else BMC_LAUNCH_BF(1, 128, 8);
    }
#undef BMC_LAUNCH_BF
    BMC_CHECK_LAUNCH("bmc_conv (bf16 planes)");
    return 0;
}
