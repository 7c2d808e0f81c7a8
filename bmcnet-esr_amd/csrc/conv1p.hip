// 1x1 convolution with the weights RESIDENT IN REGISTERS (round 3): same semantics as conv1.hip / conv.hip's TAPS = 1 path
// (multi-source, per-group weights, fused bias / residual / ReLU / mask / accumulate epilogue), for the shapes the BIE block
// launches: 128-granular output channels, K = 128 or 256 packed input channels.
//
// conv1.hip streams the [128][16] weight slice of every 16-channel K step through an LDS ring: one barrier per 64 MFMAs and
// as many bytes from L2 (weights) as from HBM (pixels) -- measured 0.48 of the MFMA peak at 0.44 of the attainable HBM rate,
// bound by neither.  Here a wave owns ONE 16-channel block of the outputs for the whole launch and keeps its slice of the
// weight matrix (16 x K: K / 4 registers per lane) in VGPRs as MFMA A-operand fragments; the only stream is the pixel tile:
//   * workgroup = 8 waves = 64 consecutive pixels (flat index: a 1x1 convolution has no 2-D structure) x 128 output channels;
//     wave w = output channels [16 w, 16 w + 16) x the 64 pixels = 4 accumulator tiles of v_mfma_f32_16x16x4_f32;
//   * the pixel tile [K/16 chunks][64 px][16 ch] (rows of 64 B, quads XOR-swizzled: dma_ring.h) is filled by LDS-DMA one tile
//     ahead (two buffers), every wave fetching K/32 KB of it; ONE barrier per tile = per 128 (K = 128) or 256 MFMAs of a wave;
//   * all eight waves read the same pixel fragments (conflict-free ds_read_b128, one per 4 MFMAs);
//   * D rows = output channels: a lane ends up with 4 consecutive channels of ITS pixel -> 16-byte epilogue loads / stores.
// Weights come in bmc_pack_weight's layout ([K/16][Coutpad][16]) and are (re)loaded when the tile's weight
// group or channel tile changes (per-sample matrices of softmax(att) . v: once per image).
#include "bmc_common.h"
#include "conv_k.h"
#include "dma_ring.h"
#include <stdlib.h>

#ifndef BMC_C1P_EPF
#define BMC_C1P_EPF 1     // K = 256 instantiation: epilogue operands requested in front of the tile's MFMAs (round 6); 0: behind them
#endif
#ifndef BMC_C1P_SPREAD
#define BMC_C1P_SPREAD 1  // the next tile's DMA pieces dealt out over the MFMA groups of the tile's first half (round 6); 0: one burst behind the first group
#endif
#ifndef BMC_C1P_ABL
#define BMC_C1P_ABL 0     // ablation builds (tools/): 1 no MFMA, 2 no stores, 4 no pixel DMA, 8 no fragment reads
#endif

namespace {

constexpr int CK = BMC_CK;
constexpr int PX = 64;                 // pixels per tile
constexpr int CHF = PX * CK;           // floats per chunk image [64 px][16 ch] = 4 KB

// DEPTH = how many tiles ahead the pixel DMA runs (DEPTH + 1 buffers)
template <int NCH, int DEPTH>
__global__ __launch_bounds__(512, (NCH == 8 && DEPTH == 1) ? 4 : 2) void conv1p_kernel(const ConvK a) {
    constexpr int SLOT = NCH * CHF;                              // one pixel tile: 32 KB (K = 128) / 64 KB (K = 256)
    constexpr int NS = DEPTH + 1;
    __shared__ __attribute__((aligned(16))) float lds[NS * SLOT + BMC_MAX_SRC * 8];
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + NS * SLOT);
    const unsigned x_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    __syncthreads();

    // ---- persistent walk over tiles, XCD-contiguous ranges (as conv.hip); tile = (image b, 64-pixel run pt, channel tile nt)
    const int HW = a.H * a.W;
    const int tpi = (HW + PX - 1) / PX;                          // pixel tiles per image
    const int ntiles = a.ntiles;
    constexpr int NX_ = 8;
    const bool xcd_map = (gridDim.x % NX_) == 0 && ntiles >= (int)gridDim.x;
    const int xcd = blockIdx.x % NX_, xj = blockIdx.x / NX_, per_x = gridDim.x / NX_;
    const int t_lo = xcd_map ? (int)((long long)ntiles * xcd / NX_) : 0;
    const int t_hi = xcd_map ? (int)((long long)ntiles * (xcd + 1) / NX_) : ntiles;
    const int t_first = xcd_map ? t_lo + xj : (int)blockIdx.x;
    const int t_stride = xcd_map ? per_x : (int)gridDim.x;
    if (t_first >= t_hi) return;
    struct TileIt { int nt, pt, b; };
    auto decode = [&](int t) {
        TileIt it;
        it.nt = t % a.ntn; t /= a.ntn;
        it.pt = t % tpi;
        it.b = t / tpi;
        return it;
    };
    // tiles are visited t_first, t_first + t_stride, ...: (nt, pt, b) advance by the digits of t_stride with carries -- integer
    // divisions (~40 VALU instructions each) stay out of the tile loop: they cost 1.6 us per tile when they were in it
    const TileIt stp = decode(t_stride);
    auto advance = [&](TileIt it) {
        it.nt += stp.nt;
        if (it.nt >= a.ntn) { it.nt -= a.ntn; ++it.pt; }
        it.pt += stp.pt;
        if (it.pt >= tpi) { it.pt -= tpi; ++it.b; }
        it.b += stp.b;
        return it;
    };

    // ---- pixel-tile loader.  DMA unit u (1 KB = 16 pixels x one 16-channel chunk): chunk u >> 2, pixel group u & 3; wave w
    //      issues units w, w + 8, ...: always ITS pixel group w & 3 (lane -> pixel (w & 3) * 16 + (lane >> 2), quad lane & 3
    //      fetching the source quad (lane & 3) ^ swz(pixel)), chunks (w >> 2), (w >> 2) + 2, ...
    //      Pixels beyond the image re-read its last pixel (their result columns are never stored).
    const int lpx = (wave & 3) * 16 + (lane >> 2);
    const unsigned lq = (unsigned)(((lane & 3) ^ swz(lane >> 2)) * 4);
    // per-image loader state (recomputed only when the image changes: the batch maps of the sources are modulo operations)
    const char* xbase[NCH / 2];
    unsigned xstride[NCH / 2];
    int xl_b = -1;
    auto loader_image = [&](int b) {
        int s_idx = 0, c_in = (wave >> 2) * CK;                  // source / channel offset of this wave's first chunk
#pragma unroll
        for (int j = 0; j < NCH / 2; ++j) {
            while (c_in >= tab[s_idx].nch) { c_in -= tab[s_idx].nch; ++s_idx; }
            const SrcDev S = tab[s_idx];
            xbase[j] = reinterpret_cast<const char*>(src_batch_ptr(S, b) + c_in);
            xstride[j] = (unsigned)S.pix_stride * 4u;
            c_in += 2 * CK;
        }
        xl_b = b;
    };
    // pieces j0 .. j1 - 1 of this wave's NCH / 2 DMA instructions for one tile
    auto issue_x_part = [&](const TileIt& it, int slot, int j0, int j1) __attribute__((always_inline)) {
        if (BMC_C1P_ABL & 4) return;
        if (it.b != xl_b) loader_image(it.b);
        int p = it.pt * PX + lpx;
        p = p < HW ? p : HW - 1;
#pragma unroll
        for (int j = j0; j < j1; ++j)
            dma16(xbase[j], (unsigned)p * xstride[j] + lq * 4u,
                  x_lds + (unsigned)((slot * SLOT + ((wave >> 2) + 2 * j) * CHF + (wave & 3) * 256) * 4));
    };
    auto issue_x = [&](const TileIt& it, int slot) { issue_x_part(it, slot, 0, NCH / 2); };

    // ---- this wave's weight slice: A fragments of output channels [nt * 128 + 16 w, + 16), k-steps = the 4 floats of a quad
    f32x4 wreg[NCH];
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
    int w_grp = -1, w_nt = -1;
    const int wq = (lk ^ swz(li)) * 4;
    auto load_w = [&](int grp, int nt) {
        const float* const wb = static_cast<const float*>(a.w) + (long long)grp * a.w_group_stride +
                                ((long long)nt * 128 + 16 * wave + li) * CK + 4 * lk;
        // (loads the compiler does not track, completed inside the statement: a tracked load in this rarely taken branch makes
        //  the compiler wait vmcnt(0) before the first MFMA of EVERY tile -- for the pixel DMA just issued and the previous
        //  tile's stores, i.e. one full memory round trip per tile, which is what this kernel exists to avoid)
        auto ld8 = [](const float* p, long long st, f32x4 (&v)[NCH], int c0) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %9, off\n\tglobal_load_dwordx4 %2, %10, off\n\t"
                         "global_load_dwordx4 %3, %11, off\n\tglobal_load_dwordx4 %4, %12, off\n\tglobal_load_dwordx4 %5, %13, off\n\t"
                         "global_load_dwordx4 %6, %14, off\n\tglobal_load_dwordx4 %7, %15, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[c0]), "=&v"(v[c0 + 1]), "=&v"(v[c0 + 2]), "=&v"(v[c0 + 3]), "=&v"(v[c0 + 4]), "=&v"(v[c0 + 5]),
                           "=&v"(v[c0 + 6]), "=&v"(v[c0 + 7])
                         : "v"(p), "v"(p + st), "v"(p + 2 * st), "v"(p + 3 * st), "v"(p + 4 * st), "v"(p + 5 * st), "v"(p + 6 * st),
                           "v"(p + 7 * st)
                         : "memory");
        };
        const long long wst = (long long)a.Coutpad * CK;
#pragma unroll
        for (int c0 = 0; c0 < NCH; c0 += 8) ld8(wb + c0 * wst, wst, wreg, c0);
        const int co = nt * 128 + 16 * wave + 4 * lk;
        bq = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias && co < a.Cout)
            asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(bq) : "v"(a.bias + (long long)grp * a.bias_group_stride + co) : "memory");
        w_grp = grp; w_nt = nt;
    };

    const int xoff = li * CK + wq;                               // + chunk * CHF + pixel block * 16 * CK
    auto read_x = [&](const float* xb, int c, f32x4 (&xf)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            if (BMC_C1P_ABL & 8) { xf[pb] = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(xf[pb])); continue; }
            xf[pb] = *reinterpret_cast<const f32x4*>(xb + c * CHF + pb * 16 * CK + xoff);
        }
    };

    int tile = t_first;
    TileIt it = decode(tile);
    // the loader runs DEPTH tiles ahead
    TileIt itl = it;
    int tl = t_first, nl = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (tl < t_hi) { issue_x(itl, nl % NS); itl = advance(itl); tl += t_stride; ++nl; }
    }
    dma_wait<0>();
    ring_publish();
    int n = 0;
    // per-image epilogue state
    int ep_b = -1, grp = 0;
    float* outb = nullptr;
    const float* resb = nullptr;
    const float* maskb = nullptr;
    for (; tile < t_hi; tile += t_stride, ++n) {
        const bool more = tl < t_hi;
        if (it.b != ep_b) {
            ep_b = it.b;
            grp = a.batch_per_group >= a.B ? 0 : it.b / a.batch_per_group;
            outb = a.out + (long long)it.b * a.out_batch_stride;
            resb = a.residual.ptr ? src_batch_ptr(a.residual, it.b) : nullptr;
            maskb = a.mask.ptr ? src_batch_ptr(a.mask, it.b) : nullptr;
        }
        if (grp != w_grp || it.nt != w_nt) load_w(grp, it.nt);

        const int co = it.nt * 128 + 16 * wave + 4 * lk;
        const bool cok = co < a.Cout;
        int pix[4];
        bool ok[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            pix[pb] = it.pt * PX + pb * 16 + li;
            ok[pb] = pix[pb] < HW && cok;
        }
        // ---- epilogue operands.  K = 256 (one workgroup per CU, two waves per SIMD, 256 registers each): requested HERE, in front
        //      of the tile's 256 MFMAs, so that they land under them -- fetched behind the MFMAs (round 3-5) every wave of the CU
        //      sat through one memory round trip per tile with the matrix pipes idle (both SIMD partners reach their epilogues
        //      together: one barrier per tile keeps them in step).  K = 128 (four waves per SIMD, 128 registers): behind the MFMAs as
        //      before, one operand at a time -- the other workgroup's waves cover the round trip.
        constexpr bool EPF = BMC_C1P_EPF && NCH == 16;
        f32x4 rv[4], mv[4], ov[4];
        auto ep_load = [&](const float* base, int stride, float fill, f32x4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                v[pb] = f32x4{fill, fill, fill, fill};
                if (ok[pb]) v[pb] = ldg16(base + (long long)pix[pb] * stride + co);
            }
        };
        // (requests the compiler does not track, waited for by count where they are used -- a tracked load at the top of the tile
        //  makes it wait vmcnt(0) THERE, for the previous tile's stores: its registers might still be the target of an older load.
        //  Lanes beyond the image or the channel count read a valid element nobody keeps: their results are never stored.)
        auto ep_request = [&](const float* base, int stride, f32x4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const float* const q = base + (long long)(pix[pb] < HW ? pix[pb] : HW - 1) * stride + (cok ? co : 0);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[pb]) : "v"(q) : "memory");
            }
        };
        if (EPF) {
            if (resb) ep_request(resb, a.residual.pix_stride, rv);
            if (maskb) ep_request(maskb, a.mask.pix_stride, mv);
            if (a.accumulate) ep_request(outb, a.out_pix_stride, ov);
        }

        // ---- the tile's MFMAs: chunk c's pixel fragments are read while chunk c - 1 is multiplied
        const float* const xb = lds + (n % NS) * SLOT;
        f32x4 acc[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc[pb] = bq;
        f32x4 xfA[4], xfB[4];
        read_x(xb, 0, xfA);
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
            read_x(xb, c + 1, xfB);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    if (BMC_C1P_ABL & 1) acc[pb][0] += wreg[c][j] * xfA[pb][j];
                    else acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c][j], xfA[pb][j], acc[pb], 0, 0, 0);
                }
            if (c + 2 < NCH) read_x(xb, c + 2, xfA);
#if BMC_C1P_SPREAD
            // tile n + DEPTH into the buffer tile n - 1 was read from (before the previous barrier): requested between the MFMA
            // groups of the tile's FIRST HALF, NCH / 8 pieces per group -- as one burst behind the first group (rounds 3-5) the
            // eight pieces (~100 cycles each: two readfirstlanes, m0, the wait states) held every wave of the CU off the matrix pipe at
            // the same moment
            if (more && c < NCH / 2) {
                constexpr int PER = (NCH / 2) / (NCH / 4);          // pieces per group: NCH / 2 pieces over NCH / 4 groups
                issue_x_part(itl, nl % NS, (c / 2) * PER, (c / 2 + 1) * PER);
                if (c == NCH / 2 - 2) { itl = advance(itl); tl += t_stride; ++nl; }
            }
#else
            if (c == 0 && more) {      // tile n + DEPTH into the buffer tile n - 1 was read from (before the previous barrier); requested
                                       // here, between MFMA groups, not in front of the tile's first MFMA
                issue_x(itl, nl % NS); itl = advance(itl); tl += t_stride; ++nl;
            }
#endif
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    if (BMC_C1P_ABL & 1) acc[pb][0] += wreg[c + 1][j] * xfB[pb][j];
                    else acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[c + 1][j], xfB[pb][j], acc[pb], 0, 0, 0);
                }
        }

        // ---- epilogue
        if (EPF && (resb || maskb || a.accumulate)) {
            // younger than the requests: the next tile's pixel DMA (NCH / 2 instructions), if it was issued
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NCH / 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) asm volatile("" : "+v"(rv[pb]), "+v"(mv[pb]), "+v"(ov[pb]));
        }
        if (resb) {
            if (!EPF) ep_load(resb, a.residual.pix_stride, 0.f, rv);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[pb] += rv[pb];
        }
        if (a.relu) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[pb][k] = fmaxf(acc[pb][k], 0.f);
        }
        if (maskb) {
            if (!EPF) ep_load(maskb, a.mask.pix_stride, 1.f, mv);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[pb][k] = mv[pb][k] > 0.f ? acc[pb][k] : 0.f;
        }
        if (a.accumulate) {
            if (!EPF) ep_load(outb, a.out_pix_stride, 0.f, ov);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[pb] += ov[pb];
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
            if (ok[pb] && !((BMC_C1P_ABL & 2) && acc[pb][0] != 1.2345e30f)) stg16(outb + (long long)pix[pb] * a.out_pix_stride + co, acc[pb]);
        // the next tile's pixels have landed: everything older than this tile's (at most 4) stores and the DEPTH - 1 younger
        // tiles' DMA (NCH / 2 instructions each; in the tail, where nothing was issued, the wait is merely stricter) is complete
        if (DEPTH > 1 && more) dma_wait<4 + (DEPTH - 1) * (NCH / 2)>(); else dma_wait<4>();
        ring_publish();
        it = advance(it);
    }
}

}  // namespace

// Called by bmc_conv1_launch (conv1.hip) for Coutpad % 128 == 0 and K = 128 / 256: returns 1 if the problem was launched here.
int bmc_conv1p_launch(ConvK k, int cus, hipStream_t st) {
    static const bool off = getenv("BMC_CONV1P") && atoi(getenv("BMC_CONV1P")) == 0;
    if (off || k.Coutpad % 128 != 0 || (k.nchunks != 8 && k.nchunks != 16)) return 0;
    const long long hw = (long long)k.H * k.W;
    k.ntn = k.Coutpad / 128;
    const long long ntiles = (long long)k.B * ((hw + PX - 1) / PX) * k.ntn;
    if (ntiles >= (1ll << 31) || hw * 1024 >= (1ll << 31)) return 0;     // (per-lane DMA offsets are 32-bit: pixel * pix_stride * 4)
    for (int i = 0; i < k.nsrc; ++i)
        if (hw * k.src[i].pix_stride * 4 >= (1ll << 32)) return 0;
    k.ntiles = (int)ntiles;
    const int per_cu = k.nchunks == 8 ? 2 : 1;
    const long long max_blocks = (long long)per_cu * cus;
    dim3 grid((unsigned)(ntiles < max_blocks ? ntiles : max_blocks)), block(512);
    if (k.nchunks == 8) hipLaunchKernelGGL((conv1p_kernel<8, 1>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv1p_kernel<16, 1>), grid, block, 0, st, k);
    return 1;
}
