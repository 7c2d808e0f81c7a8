// 1x1 convolution (multi-source, per-group weights, fused bias / residual / ReLU / mask / accumulate epilogue) on the fp32
// matrix cores -- the launches of the BIE block that are not part of the fused centre chain (chain.hip): value
// projections, softmax(att) . v with per-sample matrices, unclustering, and every 1x1 data gradient
// (models/submodules.py:65-75 and their backward).  Same semantics as conv.hip's TAPS = 1 path (bmc_conv), different machine
// mapping, the one chain.hip established for HBM-heavy K <= 256 contractions:
//   * v_mfma_f32_16x16x4_f32; workgroup = 4 waves = 8 rows x 16 pixels, a wave = 2 rows x 16 pixels x ALL output channels
//     of the channel tile (NT tiles of 16; a weight fragment feeds both rows): 8 NT accumulator registers, 2 workgroups per CU;
//   * both operand streams through swizzled, unpadded LDS rings filled by LDS-DMA (dma_ring.h): X chunks (128 px x 16 ch,
//     from HBM) 3 chunks ahead, weight slices ([128][16], from L2) 2 steps ahead; waves 0-1 load X, waves 2-3 load W;
//     raw barriers, counted vmcnt -- the prefetch survives tile boundaries and epilogues;
//   * epilogue = all loads, then all stores, 16 bytes per lane (lane = one pixel x 4 consecutive channels per tile).
// conv.hip's 32x32x2 kernel keeps the small problems (few tiles: its 4-row / 64-channel tile shapes fill the chip better).
#include "bmc_common.h"
#include "conv_k.h"
#include "dma_ring.h"
#include <stdlib.h>

namespace {

constexpr int CK = BMC_CK;
constexpr int TW = 16, MP = 2, TH = 4 * MP, NPX = TW * TH;      // workgroup tile: 8 rows x 16 pixels, MP = 2 rows per wave

template <int NT>
__global__ __launch_bounds__(256, 2) void conv1_kernel(const ConvK a) {
    constexpr int BN = 16 * NT;                   // output channels per channel tile
    constexpr int DX = 3, NXR = 5, DW = 2, NWR = 4;      // ring depths: 2 workgroups per CU must fit 160 KB of LDS
    constexpr int XSLOT = NPX * CK, WSLOT = BN * CK;
    constexpr int NDX = NPX / 32;                 // DMA instructions (16 rows = 1 KB) per X wave and chunk
    constexpr int NDW = (BN + 31) / 32;           // ... per W wave and slice
    __shared__ __attribute__((aligned(16))) float lds[NXR * XSLOT + NWR * WSLOT + BMC_MAX_SRC * 8 + BN];
    float* const Xb = lds;
    float* const Wb = lds + NXR * XSLOT;
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + NXR * XSLOT + NWR * WSLOT);
    float* const init_lds = lds + NXR * XSLOT + NWR * WSLOT + BMC_MAX_SRC * 8;     // accumulator start values (bias or zeros)
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xb;
    const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Wb;
    const bool bias_pre = a.bias != nullptr && a.batch_per_group >= a.B && a.ntn == 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < BMC_MAX_SRC; ++i)
        if (tid == i) tab[i] = a.src[i];
    if (tid < BN) init_lds[tid] = (bias_pre && tid < a.Cout) ? a.bias[tid] : 0.f;
    __syncthreads();

    // ---- persistent walk over tiles, XCD-contiguous ranges (as conv.hip)
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
    const int nsteps = a.nchunks;                 // steps (= X chunks = weight slices) per tile
    const int total_steps = my_tiles * nsteps;

    struct TileIt { int nt, tx, ty, b; };
    auto decode = [&](int t) {
        TileIt it;
        it.nt = t % a.ntn; t /= a.ntn;
        it.tx = t % a.tiles_x; t /= a.tiles_x;
        it.ty = t % a.tiles_y;
        it.b = t / a.tiles_y;
        return it;
    };

    // ---- X ring loader (waves 0-1): the tile's 16-channel chunks, source after source.  Instruction i of X wave w covers
    //      tile row NDX w + i (16 pixels, 4 lanes per pixel row); lane (pixel p, position q') fetches quad q' ^ swz(p).
    //      Pixels outside the image re-read the clamped edge pixel (their result columns are never stored).
    const bool xrole = wave < 2;
    int xl_tile = t_first, s_idx = 0, c_in = 0, snch = 0, xl_cnt = 0;
    TileIt xl_it = decode(t_first);
    const float* sbase = nullptr;
    int xpixq[NDX];
    unsigned xvoff[NDX];
    const int xq = ((lane & 3) ^ swz(lane >> 2)) * 4;
    auto src_select = [&]() {
        const SrcDev S = tab[s_idx];
        sbase = src_batch_ptr(S, xl_it.b);
        snch = S.nch;
#pragma unroll
        for (int i = 0; i < NDX; ++i) xvoff[i] = (unsigned)(xpixq[i] * S.pix_stride + xq) * 4u;
    };
    auto xl_setup = [&]() {
        const int y0 = xl_it.ty * TH, x0 = xl_it.tx * TW;
#pragma unroll
        for (int i = 0; i < NDX; ++i) {
            int y = y0 + (wave & 1) * NDX + i, x = x0 + (lane >> 2);
            y = y < a.H ? y : a.H - 1;
            x = x < a.W ? x : a.W - 1;
            xpixq[i] = y * a.W + x;
        }
        s_idx = 0; c_in = 0;
        src_select();
    };
    auto issue_x = [&]() {
        const float* base = sbase + c_in;
        const unsigned dst = xb_lds + (unsigned)(((xl_cnt % NXR) * XSLOT + (wave & 1) * NDX * 256) * 4);
#pragma unroll
        for (int i = 0; i < NDX; ++i) dma16(base, xvoff[i], dst + i * 1024);
        ++xl_cnt;
        c_in += CK;
        if (xl_cnt % nsteps == 0) {              // tile finished
            xl_tile += t_stride;
            if (xl_tile < t_hi) { xl_it = decode(xl_tile); xl_setup(); }
        } else if (c_in >= snch) {
            c_in = 0; ++s_idx;
            src_select();
        }
    };
    // ---- W ring loader (waves 2-3): slice k of the tile's weight group / channel tile
    int wl_tile = t_first, wl_step = 0, wl_cnt = 0;
    const float* wl_base = nullptr;
    unsigned wvoff[NDW];
    bool wact[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        const int row = ((wave & 1) * NDW + i) * 16 + (lane >> 2);
        wact[i] = ((wave & 1) * NDW + i) * 16 < BN;
        wvoff[i] = (unsigned)(row * 64 + (((lane & 3) ^ swz(row)) * 16));
    }
    const long long wstep = (long long)a.Coutpad * CK;
    auto wl_setup = [&]() {
        const TileIt it = decode(wl_tile);
        const int grp = a.batch_per_group >= a.B ? 0 : it.b / a.batch_per_group;
        wl_base = static_cast<const float*>(a.w) + (long long)grp * a.w_group_stride + (long long)it.nt * BN * CK;
        wl_step = 0;
    };
    auto issue_w = [&]() {
        const float* p = wl_base + (long long)wl_step * wstep;
        const unsigned dst = wb_lds + (unsigned)(((wl_cnt % NWR) * WSLOT + (wave & 1) * NDW * 256) * 4);
#pragma unroll
        for (int i = 0; i < NDW; ++i)
            if (wact[i]) dma16(p, wvoff[i], dst + i * 1024);
        ++wl_cnt;
        if (++wl_step == nsteps) {
            wl_tile += t_stride;
            if (wl_tile < t_hi) wl_setup();
        }
    };

    // ---- fragments
    const int qoff = (lg ^ swz(lp)) * 4;
    const int arow = (16 * MP * wave + lp) * CK + qoff;        // + 16 m rows for the wave's m-th pixel row
    const int brow = lp * CK + qoff;
    f32x4 afA[MP], afB[MP], bfA[NT], bfB[NT];
    auto read_a = [&](const float* xb, f32x4 (&af)[MP]) {
#pragma unroll
        for (int m = 0; m < MP; ++m) af[m] = *reinterpret_cast<const f32x4*>(xb + arow + 16 * m * CK);
    };
    auto read_b = [&](const float* wb, f32x4 (&bf)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const f32x4*>(wb + brow + 16 * t * CK);
    };
    f32x4 acc[MP][NT];
    auto init_acc = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(init_lds + 16 * t + 4 * lg);
#pragma unroll
            for (int m = 0; m < MP; ++m) acc[m][t] = v;
        }
    };

    int gs = 0;        // global step: X chunk gs sits in ring slot gs % NXR, weight slice gs in slot gs % NWR
    auto loader = [&]() {
        if (xrole) {
            if (xl_cnt < total_steps) issue_x();
            if (xl_cnt - (gs + 1) >= DX) dma_wait<NDX * (DX - 1)>(); else dma_wait<0>();
        } else {
            if (wl_cnt < total_steps) issue_w();
            if (wl_cnt - (gs + 1) >= DW) dma_wait<NDW * (DW - 1)>(); else dma_wait<0>();
        }
    };
    auto step = [&](const f32x4 (&af)[MP], const f32x4 (&bf)[NT], f32x4 (&afn)[MP], f32x4 (&bfn)[NT]) {
        loader();
        ring_publish();                      // stage gs + 1 is in LDS for everybody
        if (gs + 1 < total_steps) {
            read_b(Wb + ((gs + 1) % NWR) * WSLOT, bfn);
            read_a(Xb + ((gs + 1) % NXR) * XSLOT, afn);
        }
        // a weight fragment feeds the wave's MP pixel rows: MP independent chains per tile (dependent MFMAs >= 64 cycles apart)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MP; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t][j], af[m][j], acc[m][t], 0, 0, 0);
        ++gs;
    };

    // ---- epilogue: bias / residual / ReLU / mask / accumulate; ALL loads before ALL stores (vmcnt is in-order and counts
    //      stores: a load behind a store would wait for the store's round trip)
    auto epilogue = [&](const TileIt& it) {
        const int g = a.batch_per_group >= a.B ? 0 : it.b / a.batch_per_group;
        const float* const biasg = a.bias ? a.bias + (long long)g * a.bias_group_stride : nullptr;
        float* const outb = a.out + (long long)it.b * a.out_batch_stride;
        const float* const resb = a.residual.ptr ? src_batch_ptr(a.residual, it.b) : nullptr;
        const float* const maskb = a.mask.ptr ? src_batch_ptr(a.mask, it.b) : nullptr;
        const int co0 = it.nt * BN + 4 * lg;
        const int x = it.tx * TW + lp;
        bool ok[MP][NT];
        int pix[MP];
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            const int y = it.ty * TH + MP * wave + m;
            pix[m] = y * a.W + x;
#pragma unroll
            for (int t = 0; t < NT; ++t) ok[m][t] = y < a.H && x < a.W && co0 + 16 * t < a.Cout;
        }
        auto fetch = [&](const float* base, int stride, f32x4 (&d)[MP][NT], float fill) {
#pragma unroll
            for (int m = 0; m < MP; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    d[m][t] = f32x4{fill, fill, fill, fill};
                    if (ok[m][t]) d[m][t] = *reinterpret_cast<const f32x4*>(base + (long long)pix[m] * stride + co0 + 16 * t);
                }
        };
        if (biasg && !bias_pre) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 d = co0 + 16 * t < a.Cout ? *reinterpret_cast<const f32x4*>(biasg + co0 + 16 * t) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MP; ++m) acc[m][t] += d;
            }
        }
        if (resb) {
            f32x4 d[MP][NT];
            fetch(resb, a.residual.pix_stride, d, 0.f);
#pragma unroll
            for (int m = 0; m < MP; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] += d[m][t];
        }
        if (a.relu) {
#pragma unroll
            for (int m = 0; m < MP; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[m][t][k] = fmaxf(acc[m][t][k], 0.f);
        }
        if (maskb) {
            f32x4 d[MP][NT];
            fetch(maskb, a.mask.pix_stride, d, 1.f);
#pragma unroll
            for (int m = 0; m < MP; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[m][t][k] = d[m][t][k] > 0.f ? acc[m][t][k] : 0.f;
        }
        if (a.accumulate) {
            f32x4 d[MP][NT];
            fetch(outb, a.out_pix_stride, d, 0.f);
#pragma unroll
            for (int m = 0; m < MP; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[m][t] += d[m][t];
        }
#pragma unroll
        for (int m = 0; m < MP; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (ok[m][t]) *reinterpret_cast<f32x4*>(outb + (long long)pix[m] * a.out_pix_stride + co0 + 16 * t) = acc[m][t];
        init_acc();
    };

    // ---- prologue: fill the rings, publish stage 0
    if (xrole) {
        xl_setup();
        for (int k = 0; k < DX && xl_cnt < total_steps; ++k) issue_x();
    } else {
        wl_setup();
        for (int k = 0; k < DW && wl_cnt < total_steps; ++k) issue_w();
    }
    dma_wait<0>();
    ring_publish();
    read_a(Xb, afA);
    read_b(Wb, bfA);
    init_acc();

    // Steps alternate between the fragment sets A and B.  nsteps may be odd, so the parity is carried across tiles by
    // running the step loop two steps at a time over the workgroup's whole step sequence.
    int tile = t_first, in_tile = 0;
    auto after_step = [&]() {
        if (++in_tile == nsteps) {
            epilogue(decode(tile));
            tile += t_stride;
            in_tile = 0;
        }
    };
    for (int s = 0; s < total_steps; s += 2) {
        step(afA, bfA, afB, bfB);
        after_step();
        if (s + 1 < total_steps) {
            step(afB, bfB, afA, bfA);
            after_step();
        }
    }
}

}  // namespace

// Called by bmc_conv (conv.hip) for taps == 1, fp32 arithmetic, once the argument block is validated and the geometry
// fields that do not depend on the tile shape are filled in.  Returns 1 if the problem was launched here, 0 if it is
// left to the 32x32x2 kernel (small problems, channel counts this kernel has no instantiation for).
int bmc_conv1_launch(ConvK k, int cus, hipStream_t st) {
    // K = 128 / 256 with 128-granular output channels and enough tiles: the kernel with register-resident weights (conv1p.hip)
    {
        static const long long min64 = getenv("BMC_CONV1P_MIN_TILES") ? atoll(getenv("BMC_CONV1P_MIN_TILES")) : -1;     // (tests)
        const long long tiles64 = (long long)k.B * (((long long)k.H * k.W + 63) / 64) * (k.Coutpad / 128);
        if (k.Coutpad % 128 == 0 && tiles64 >= (min64 >= 0 ? min64 : 2ll * cus) && bmc_conv1p_launch(k, cus, st)) return 1;
    }
    int NT;
    if (k.Coutpad == 32) NT = 2;
    else if (k.Coutpad % 128 == 0) NT = 8;
    else return 0;
    const int BN = 16 * NT;
    k.tiles_x = (k.W + TW - 1) / TW;
    k.tiles_y = (k.H + TH - 1) / TH;
    k.ntn = k.Coutpad / BN;
    const long long ntiles = (long long)k.B * k.tiles_x * k.tiles_y * k.ntn;
    // few tiles: conv.hip's smaller tile shapes fill the chip better (BMC_CONV1_MIN_TILES overrides the threshold: tests)
    static const long long min_tiles = getenv("BMC_CONV1_MIN_TILES") ? atoll(getenv("BMC_CONV1_MIN_TILES")) : -1;
    if (ntiles >= (1ll << 31) || ntiles < (min_tiles >= 0 ? min_tiles : 4ll * cus)) return 0;
    k.ntiles = (int)ntiles;
    const int max_blocks = 2 * cus;                                 // resident workgroups per CU (LDS rings, registers)
    dim3 grid((unsigned)(ntiles < max_blocks ? ntiles : max_blocks)), block(256);
    if (NT == 8) hipLaunchKernelGGL((conv1_kernel<8>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv1_kernel<2>), grid, block, 0, st, k);
    return 1;
}
