// Fused "centre" chain of the BIE block on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//
//   forward  (models/submodules.py:63-64, 127-140):   z = convf(cat[s0, s1]) + b_f            1x1, 2C -> C
//                                                      yhat = (z - mean_c z) * rstd            LayerNorm2d, eps inside the sqrt
//                                                      centre = clustering(yhat*gamma + beta)  1x1, C -> C
//   backward (autograd of the same, LayerNormFunction.backward :141-154):
//                                                      dy = W_c^T dcentre;  dz = LN'(dy);  d s0 += W_f[:, :C]^T dz;  d s1 = W_f[:, C:]^T dz
//
// One launch per direction instead of conv -> LayerNorm -> conv (forward) and dgrad -> LayerNorm-backward -> 3 dgrads
// (backward): z, y = LN(z) and dy never travel to HBM.  What makes the fusion cheap on this machine: the 32x32x2 fp32
// MFMA takes ONE register per operand per lane, and its result layout (rows = channels in the 16 registers, columns =
// pixels on the lanes, lane half h owning rows 4h..4h+3 of every 8) is exactly the B-operand layout of a following
// MFMA that sums over the channel index, with the k order (c, c + 4) that the weight fragment reads (4 consecutive k
// at offset 4h) already use.  So the second GEMM of the chain takes its pixel operand straight from the first one's
// accumulators: no LDS round trip, no shuffles.  The per-pixel LayerNorm reductions are in-lane sums over 64
// registers plus one exchange between lanes l and l + 32.
//
// Workgroup = 4 waves = 8x16 pixels; wave = 2 rows x 16 pixels = 32 pixels x ALL C channels (NU = C/32 MFMA tiles), so
// a pixel's channels live in two lanes of one wave.  K is walked in 16-channel steps; a step's [C][16] weight slice
// streams through a double-buffered LDS image (loaded two steps ahead, one barrier per step), input pixels of the
// first GEMM are staged through LDS like in conv.hip (coalesced 16-byte loads, conflict-free ds_read_b128).
// Persistent workgroups (at C = 128 one per CU: two 64-register result tiles + operands per lane do not fit the 256
// registers that two waves per SIMD would leave), load pipeline running across tile boundaries, XCD-contiguous tile ranges.
//
// Backward works on PAIRS of tiles (batch b and b + n, the two polarity halves of the twin layout) so that the two
// contributions to the shared stream's gradient d s0[b] are summed in registers: no read-modify-write of d s0.
// Weight gradients stay with the pixel-reduction GEMM (pgemm.hip): it reads (dcentre, yhat) and (dz, s0, s1); the
// LayerNorm affine gradients follow algebraically from its results (bmc_chain_affine_grads below).
#include "bmc_common.h"
#include "dma_ring.h"

#ifndef BMC_CHAIN_ABL
#define BMC_CHAIN_ABL 0     // ablation bits for tools/ builds only: 1 no MFMAs, 2 no DMA issue, 4 no global stores,
                            // 8 no LayerNorm math, 16 no barriers, 32 no fragment reads, 64 no DMA waits
#endif

namespace {

constexpr int CK = BMC_CK;
constexpr int TW = 16, TH = 4, NPX = TW * TH;      // workgroup tile: 4 rows x 16 pixels, one row per wave
__device__ __attribute__((aligned(16))) const float g_zero4c[4] = {0.f, 0.f, 0.f, 0.f};

struct ChainK {
    SrcDev src[2];        // fwd: s0, s1 (C channels each, with their batch maps); bwd: src[0] = dcentre (C channels)
    const float* w;       // weight stream of one unit, slices of [C][16] floats in consumption order
    const float* vec[4];  // fwd: b_f, b_c, gamma, beta;  bwd: -, -, gamma, -
    float eps;
    float* out0;          // fwd: yhat   [B][H][W][C]     bwd: dz    [2n][H][W][C]
    float* out1;          // fwd: centre [B][H][W][C]     bwd: d s1  [2n][H][W][C]  (written at batch (bb + n) % 2n)
    float* out2;          // fwd: rstd   [B][H][W]        bwd: d s0  [n][H][W][C]
    const float* in0;     // bwd: yhat
    const float* in1;     // bwd: rstd
    SrcDev res;           // bwd: term added to d s0 (upstream gradient of the skip connection), or ptr == nullptr
    int nb;               // fwd: launch batches B; bwd: n (pairs)
    int n;                // bwd: batch distance of the two halves
    int H, W, tiles_x, tiles_y, nunits;
};

// Two (or more) independent workgroups per CU hide each other's LayerNorm / epilogue / barrier phases (a first version
// with 32x32x2 MFMAs needed 128 accumulator registers per lane for its two result tiles = one workgroup per CU, and ran
// its matrix pipes only 60 % busy: every non-MFMA instruction of the single wave per SIMD was exposed).  Both operand
// streams run through LDS rings filled by LDS-DMA several steps ahead -- X chunks (64 px x 16 ch = 4 KB, from HBM) DX
// chunks ahead, weight slices (C x 16, from L2) DW steps ahead.  Roles by wave (a wave's vmcnt completes in order: a
// wave waiting for a weight slice would wait for its younger X chunks too): waves 0-1 fill the X ring, waves 2-3 the W
// ring; all four compute.  Rows are 64 B without padding; the 16-byte quads of a row are XOR-swizzled with
// SWZ[(row >> 2) & 3] on the DMA's SOURCE address and on the fragment reads (conflict-free ds_read_b128 for the
// 16-row x 4-quad fragment shape), the LDS destination of a DMA stays lane-linear.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    if (BMC_CHAIN_ABL & 1) { c[0] += a * b; return c; }      // ablation: one VALU fma instead, operands stay live
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int NT, bool BWD>
__global__ __launch_bounds__(256, 2) void chain_kernel(const ChainK a) {
    constexpr int C = 16 * NT;
    constexpr int NR = NT;                        // steps of one register-operand GEMM (K = C, 16 per step)
    constexpr int NS = BWD ? 5 * NR : 3 * NR;     // steps per unit
    constexpr int NXC = 2 * NR;                   // X chunks per unit (two segments of C channels)
    constexpr int DX = 6, NXR = 8, DW = 3, NWR = 5;
    constexpr int XSLOT = NPX * CK, WSLOT = C * CK;          // floats per ring slot
    constexpr int NDX = NPX / 32;                 // DMA instructions (16 rows = 1 KB each) per X wave and chunk
    constexpr int NDW = (C + 31) / 32;            // ... per W wave and slice
    __shared__ __attribute__((aligned(16))) float lds[NXR * XSLOT + NWR * WSLOT + 2 * 8 + 5 * C];
    float* const Xb = lds;
    float* const Wb = lds + NXR * XSLOT;
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + NXR * XSLOT + NWR * WSLOT);
    float* const img = lds + NXR * XSLOT + NWR * WSLOT + 2 * 8;     // [0] b_f | 0   [1] b_c | 0   [2] gamma   [3] beta   [4] zeros
    const unsigned xb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Xb;
    const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Wb;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lp = lane & 15, lg = lane >> 4;     // pixel of the wave's row / channel quad: D rows 4*lg .. 4*lg+3
    if (tid < 2) tab[tid] = a.src[tid];
    for (int i = tid; i < 5 * C; i += 256) {
        const int k = i / C, c = i - k * C;
        img[i] = (k < 4 && a.vec[k]) ? a.vec[k][c] : 0.f;
    }
    __syncthreads();

    // ---- persistent walk over units, XCD-contiguous ranges (as conv.hip)
    const int nunits = a.nunits;
    constexpr int NX_ = 8;
    const bool xcd_map = (gridDim.x % NX_) == 0 && nunits >= (int)gridDim.x;
    const int xcd = blockIdx.x % NX_, xj = blockIdx.x / NX_, per_x = gridDim.x / NX_;
    const int t_lo = xcd_map ? (int)((long long)nunits * xcd / NX_) : 0;
    const int t_hi = xcd_map ? (int)((long long)nunits * (xcd + 1) / NX_) : nunits;
    const int t_first = xcd_map ? t_lo + xj : (int)blockIdx.x;
    const int t_stride = xcd_map ? per_x : (int)gridDim.x;
    const int my_units = t_first < t_hi ? (t_hi - t_first + t_stride - 1) / t_stride : 0;
    if (my_units == 0) return;
    const int total_steps = my_units * NS, total_xchunks = my_units * NXC;

    struct UnitIt { int tx, ty, b; };
    auto decode = [&](int t) {
        UnitIt it;
        it.tx = t % a.tiles_x; t /= a.tiles_x;
        it.ty = t % a.tiles_y;
        it.b = t / a.tiles_y;
        return it;
    };

    // ---- X ring loader (waves 0-1): chunks of 16 channels in consumption order; a unit has two segments of C channels
    //      (fwd: s0 then s1 at batch b; bwd: dcentre at batch b then at batch b + n).  Instruction i of X wave w covers the
    //      16 tile pixels [(NDX w + i) * 16, +16) = tile row NDX w + i, 4 lanes per pixel row; lane (pixel p, position q')
    //      fetches quad q' ^ swz(p).  Pixels outside the image re-read the clamped edge pixel: their columns of the result
    //      are never stored, and a pixel's column never mixes with another's.
    const bool xrole = wave < 2;
    int xl_unit = t_first, xl_seg = 0, c_in = 0, xl_cnt = 0;
    UnitIt xl_it = decode(t_first);
    const float* sbase = nullptr;
    int xpixq[NDX];               // pixel index of this lane's row per instruction (tile-dependent)
    unsigned xvoff[NDX];          // byte offset from the segment's batch pointer (segment-dependent: pixel stride)
    const int xq = ((lane & 3) ^ swz(lane >> 2)) * 4;         // row index within the instruction = lane / 4
    auto seg_select = [&]() {
        const SrcDev S = tab[BWD ? 0 : xl_seg];
        sbase = src_batch_ptr(S, BWD ? xl_it.b + xl_seg * a.n : xl_it.b);
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
        xl_seg = 0; c_in = 0;
        seg_select();
    };
    auto issue_x = [&]() {       // the next chunk of the stream -> ring slot xl_cnt % NXR
        const float* base = sbase + c_in;
        const unsigned dst = xb_lds + (unsigned)(((xl_cnt % NXR) * XSLOT + (wave & 1) * NDX * 256) * 4);
#pragma unroll
        for (int i = 0; i < NDX; ++i)
            if (!(BMC_CHAIN_ABL & 2)) dma16(base, xvoff[i], dst + i * 1024);
        ++xl_cnt;
        c_in += CK;
        if (c_in >= C) {
            c_in = 0;
            if (++xl_seg == 2) {
                xl_unit += t_stride;
                if (xl_unit < t_hi) { xl_it = decode(xl_unit); xl_setup(); }
            } else {
                seg_select();
            }
        }
    };
    // ---- W ring loader (waves 2-3): NS slices per unit, the same for every unit
    int wl_step = 0, wl_cnt = 0;
    unsigned wvoff[NDW];
    bool wact[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        const int row = ((wave & 1) * NDW + i) * 16 + (lane >> 2);
        wact[i] = ((wave & 1) * NDW + i) * 16 < C;
        wvoff[i] = (unsigned)(row * 64 + (((lane & 3) ^ swz(row)) * 16));
    }
    auto issue_w = [&]() {
        const float* p = a.w + (long long)wl_step * WSLOT;
        const unsigned dst = wb_lds + (unsigned)(((wl_cnt % NWR) * WSLOT + (wave & 1) * NDW * 256) * 4);
#pragma unroll
        for (int i = 0; i < NDW; ++i)
            if (wact[i] && !(BMC_CHAIN_ABL & 2)) dma16(p, wvoff[i], dst + i * 1024);
        ++wl_cnt;
        if (++wl_step == NS) wl_step = 0;
    };

    // ---- fragments: pixel row 16*wave + lp / weight row 16*t + lp; this lane's quad lg, swizzled by the row
    const int qoff = (lg ^ swz(lp)) * 4;          // (16*wave + lp) and (16*t + lp) have the same (row >> 2) & 3
    const int arow = (16 * wave + lp) * CK + qoff;
    const int brow = lp * CK + qoff;              // + 16*t rows
    f32x4 afA, afB, bfA[NT], bfB[NT];             // two fragment sets: the next step's reads fly under this step's MFMAs
    auto read_a = [&](const float* xb, f32x4& af) {
        if (BMC_CHAIN_ABL & 32) { af = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(af)); return; }
        af = *reinterpret_cast<const f32x4*>(xb + arow);
    };
    auto read_b = [&](const float* wb, f32x4 (&bf)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (BMC_CHAIN_ABL & 32) { bf[t] = f32x4{1.f, 2.f, 3.f, 4.f}; asm volatile("" : "+v"(bf[t])); continue; }
            bf[t] = *reinterpret_cast<const f32x4*>(wb + brow + 16 * t * CK);
        }
    };

    f32x4 Ra[NT], Rt[NT], Rz[BWD ? NT : 1];
    auto init_acc = [&](f32x4 (&R)[NT], int which) {
#pragma unroll
        for (int t = 0; t < NT; ++t) R[t] = *reinterpret_cast<const f32x4*>(img + which * C + 16 * t + 4 * lg);
    };

    // ---- pipeline state
    int gs = 0;        // global step of this workgroup: its weight slice sits in ring slot gs % NWR
    int xc = 0;        // next X chunk to be consumed: ring slot xc % NXR
    // Loader part of a step (before the barrier that publishes the next stage): issue what lies D stages ahead, then
    // wait until the NEXT stage's data has landed (everything but the D - 1 younger stages).  x_step: this step consumed
    // an X chunk (the X ring advances only then).
    auto loader = [&](bool x_step) {
        if (xrole) {
            if (x_step && xl_cnt < total_xchunks) issue_x();
            if (BMC_CHAIN_ABL & 64) {}
            else if (xl_cnt - (xc + (x_step ? 1 : 0)) >= DX) dma_wait<NDX * (DX - 1)>(); else dma_wait<0>();
        } else {
            if (wl_cnt < total_steps) issue_w();
            if (BMC_CHAIN_ABL & 64) {}
            else if (wl_cnt - (gs + 1) >= DW) dma_wait<NDW * (DW - 1)>(); else dma_wait<0>();
        }
    };
    auto publish = [&]() {      // raw barrier: no vmcnt(0) drain of the rings' prefetch (a __syncthreads() would)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(BMC_CHAIN_ABL & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // Two steps whose pixel operand comes from LDS (chunks xc, xc + 1), fragment sets A then B.  next_lds: the step after
    // the pair also reads an X chunk.  (LDS phases always have an even number of steps: NR is even for C >= 32.)
    auto lds_pair = [&](f32x4 (&acc)[NT], bool next_lds) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4& af = h ? afB : afA;
            f32x4(&bf)[NT] = h ? bfB : bfA;
            f32x4& afn = h ? afA : afB;
            f32x4(&bfn)[NT] = h ? bfA : bfB;
            const bool has_next = gs + 1 < total_steps;
            const bool nl = h == 0 || next_lds;
            loader(true);
            publish();                      // stage gs + 1 (and chunk xc + 1) is in LDS for everybody
            if (has_next) {
                read_b(Wb + ((gs + 1) % NWR) * WSLOT, bfn);
                if (nl) read_a(Xb + ((xc + 1) % NXR) * XSLOT, afn);
            }
            // tile pairs outermost: a pair's two chains alternate (dependent MFMAs 64 cycles apart, latency 40) and its
            // weight fragments die after 8 MFMAs, so the registers of set A and of the incoming set B overlap
#pragma unroll
            for (int t = 0; t < NT; t += 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[t] = mfma16(bf[t][j], af[j], acc[t]);
                    acc[t + 1] = mfma16(bf[t + 1][j], af[j], acc[t + 1]);
                }
            ++gs; ++xc;
        }
    };
    // NR steps whose pixel operand is the register tile `op` (a previous result); next_lds refers to the step after the
    // last one (the first step of the next LDS phase consumes chunk xc).
    auto reg_steps = [&](const f32x4 (&op)[NT], f32x4 (&acc)[NT], bool next_lds) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            f32x4(&bf)[NT] = (c & 1) ? bfB : bfA;
            f32x4(&bfn)[NT] = (c & 1) ? bfA : bfB;
            f32x4& afn = (c & 1) ? afA : afB;
            const bool has_next = gs + 1 < total_steps;
            const bool nl = (c == NR - 1) && next_lds;
            loader(false);
            publish();
            if (has_next) {
                read_b(Wb + ((gs + 1) % NWR) * WSLOT, bfn);
                if (nl) read_a(Xb + (xc % NXR) * XSLOT, afn);
            }
#pragma unroll
            for (int t = 0; t < NT; t += 2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[t] = mfma16(bf[t][j], op[c][j], acc[t]);
                    acc[t + 1] = mfma16(bf[t + 1][j], op[c][j], acc[t + 1]);
                }
            ++gs;
        }
    };

    // ---- per-lane output geometry of the current unit: pixel (row wave, column lp) of the tile, channels 16 t + 4 lg ..
    UnitIt ep_it = decode(t_first);
    int ep_unit = t_first;
    bool pok = false;
    int pix = 0;
    auto ep_setup = [&]() {
        const int y = ep_it.ty * TH + wave, x = ep_it.tx * TW + lp;
        pok = y < a.H && x < a.W;
        pix = y * a.W + x;
    };
    const long long img_elems = (long long)a.H * a.W;
    auto store_tile = [&](const f32x4 (&R)[NT], float* base) {     // base: batch plane [H][W][C]
        if (!pok || ((BMC_CHAIN_ABL & 4) && R[0][0] != 12345.678f)) return;
        float* p = base + (long long)pix * C + 4 * lg;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(p + 16 * t) = R[t];
    };
    auto wsum = [&](float v) {      // over the four lanes (channel quads) that share a pixel
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    };

    // ---- prologue: fill the rings DX / DW stages deep, publish stage 0
    if (xrole) {
        xl_setup();
        for (int k = 0; k < DX && xl_cnt < total_xchunks; ++k) issue_x();
    } else {
        for (int k = 0; k < DW && wl_cnt < total_steps; ++k) issue_w();
    }
    dma_wait<0>();
    publish();
    read_a(Xb, afA);
    read_b(Wb, bfA);
    init_acc(Ra, BWD ? 4 : 0);
    // Steps alternate between the fragment sets A and B; every phase has an even number of steps, so each phase starts on A.
    static_assert(NR % 2 == 0, "C must be a multiple of 32");

    for (; ep_unit < t_hi; ep_unit += t_stride) {
        ep_it = decode(ep_unit);
        ep_setup();
        const bool more_units = ep_unit + t_stride < t_hi;
        if constexpr (!BWD) {
            // ---- z = convf(cat[s0, s1]) + b_f
            for (int i = 0; i < NXC; i += 2) lds_pair(Ra, i + 2 < NXC);
            // ---- LayerNorm over the C channels of each pixel: 4 NT registers in each of the pixel's four lanes
            float rstd = 1.f;
            if (!(BMC_CHAIN_ABL & 8)) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) s += (Ra[t][0] + Ra[t][1]) + (Ra[t][2] + Ra[t][3]);
                const float mu = wsum(s) * (1.f / C);
                float q = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = Ra[t][k] - mu;
                        Ra[t][k] = d;
                        q += d * d;
                    }
                rstd = 1.f / sqrtf(wsum(q) * (1.f / C) + a.eps);
#pragma unroll
                for (int t = 0; t < NT; ++t) Ra[t] *= rstd;
            }
            store_tile(Ra, a.out0 + (long long)ep_it.b * img_elems * C);
            if (pok && lg == 0) a.out2[(long long)ep_it.b * img_elems + pix] = rstd;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 gq = *reinterpret_cast<const f32x4*>(img + 2 * C + 16 * t + 4 * lg);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(img + 3 * C + 16 * t + 4 * lg);
                Ra[t] = Ra[t] * gq + bq;
            }
            // ---- centre = clustering(y) + b_c, pixel operand = the registers of y
            init_acc(Rt, 1);
            reg_steps(Ra, Rt, more_units);
            store_tile(Rt, a.out1 + (long long)ep_it.b * img_elems * C);
            init_acc(Ra, 0);
        } else {
            // dz of BOTH halves stays in registers (Rz: batch b, Ra: batch b + n): the shared stream's gradient needs
            // W_f[:, :C]^T (dz[b] + dz[b + n]) -- ONE register-operand GEMM on the sum instead of two (5 C^2 instead of
            // 6 C^2 multiply-adds per pixel pair half), and d s0 is written once.
            for (int half = 0; half < 2; ++half) {
                const int bb = ep_it.b + half * a.n;
                // ---- dy = W_c^T dcentre
                for (int i = 0; i < NR; i += 2) lds_pair(Ra, half == 0 || i + 2 < NR);
                // ---- LayerNorm backward: g = dy*gamma, dz = rstd * (g - yhat*mean(g*yhat) - mean(g))
                const float* const yh = a.in0 + (long long)bb * img_elems * C + (long long)pix * C + 4 * lg;
                const float rstd = pok ? a.in1[(long long)bb * img_elems + pix] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 gq = *reinterpret_cast<const f32x4*>(img + 2 * C + 16 * t + 4 * lg);
                    const f32x4 yq = ldg16(pok ? yh + 16 * t : g_zero4c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float g = Ra[t][k] * gq[k];
                        Ra[t][k] = g;
                        s1 += g;
                        s2 += g * yq[k];
                    }
                }
                const float m1 = wsum(s1) * (1.f / C), m2 = wsum(s2) * (1.f / C);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f32x4 yq = ldg16(pok ? yh + 16 * t : g_zero4c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) Ra[t][k] = rstd * (Ra[t][k] - yq[k] * m2 - m1);
                }
                store_tile(Ra, a.out0 + (long long)bb * img_elems * C);
                if (half == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) Rz[t] = Ra[t];
                    init_acc(Ra, 4);
                }
            }
            // ---- d s1, written for the batch the forward READ s1 from: batch b's dz goes to b + n and vice versa
            init_acc(Rt, 4);
            reg_steps(Rz, Rt, false);                                          // W_f[:, C:]^T dz[b]
            store_tile(Rt, a.out1 + (long long)(ep_it.b + a.n) * img_elems * C);
            init_acc(Rt, 4);
            reg_steps(Ra, Rt, false);                                          // W_f[:, C:]^T dz[b + n]
            store_tile(Rt, a.out1 + (long long)ep_it.b * img_elems * C);
            // ---- d s0 = add + W_f[:, :C]^T (dz[b] + dz[b + n])
#pragma unroll
            for (int t = 0; t < NT; ++t) Rz[t] += Ra[t];
            init_acc(Rt, 4);
            if (a.res.ptr && pok) {
                const float* rp = src_batch_ptr(a.res, ep_it.b) + (long long)pix * a.res.pix_stride + 4 * lg;
#pragma unroll
                for (int t = 0; t < NT; ++t) Rt[t] = ldg16(rp + 16 * t);
            }
            reg_steps(Rz, Rt, more_units);
            store_tile(Rt, a.out2 + (long long)ep_it.b * img_elems * C);
            init_acc(Ra, 4);
        }
    }
}

// LayerNorm affine gradients from the pixel-reduction GEMM of the clustering convolution taken on yhat:
//   G[co][ci] = sum_px dcentre[px][co] yhat[px][ci],   dbc[co] = sum_px dcentre[px][co]
//   dW_c[co][ci] = gamma[ci] G[co][ci] + dbc[co] beta[ci]        (y = yhat*gamma + beta)
//   dgamma[ci]   = sum_co W_c[co][ci] G[co][ci]                    (= sum_px dy*yhat, dy = W_c^T dcentre)
//   dbeta[ci]    = sum_co W_c[co][ci] dbc[co]                      (= sum_px dy)
// Block = 32 columns ci x 32 row groups (co = part, part + 32, ...: four dependent global round trips per thread, the
// kernel is pure latency): partial column sums meet in LDS.  Results are written
// (accumulate = 0) or added (1) to dwc / dgamma / dbeta (dwc may be G itself); dbc_out, if given, receives dbc the same
// way (the clustering bias gradient on its way to its accumulator).
__global__ __launch_bounds__(1024) void affine_grads_kernel(const float* G, const float* __restrict__ dbc, const float* __restrict__ Wc,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int C, float* dwc,
                                    float* __restrict__ dbc_out, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    int accumulate) {
    constexpr int NPART = 32;
    __shared__ float red[2][NPART][32];
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5, ci = blockIdx.x * 32 + cl;
    float dg = 0.f, db = 0.f;
    if (ci < C) {
        const float gm = gamma[ci], bt = beta[ci];
        for (int co = part; co < C; co += NPART) {
            const float g = G[co * C + ci], w = Wc[co * C + ci], d = dbc[co];
            dg += w * g;
            db += w * d;
            const float v = gm * g + d * bt;
            dwc[co * C + ci] = accumulate ? dwc[co * C + ci] + v : v;
        }
    }
    red[0][part][cl] = dg;
    red[1][part][cl] = db;
    __syncthreads();
    if (part == 0 && ci < C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < NPART; ++k) { a += red[0][k][cl]; b += red[1][k][cl]; }
        dgamma[ci] = accumulate ? dgamma[ci] + a : a;
        dbeta[ci] = accumulate ? dbeta[ci] + b : b;
        if (dbc_out) dbc_out[ci] = accumulate ? dbc_out[ci] + dbc[ci] : dbc[ci];
    }
}

int launch(const ChainK& k, int C, bool bwd, hipStream_t st) {
    const int cus = bmc_num_cus();
    const int max_blocks = 2 * cus;      // resident workgroups per CU (LDS rings: ~75 KB per workgroup at C = 128)
    dim3 grid((unsigned)(k.nunits < max_blocks ? k.nunits : max_blocks)), block(256);
#define BMC_LAUNCH_CHAIN(NU_)                                                                    \
    do {                                                                                          \
        if (bwd) hipLaunchKernelGGL((chain_kernel<NU_, true>), grid, block, 0, st, k);            \
        else hipLaunchKernelGGL((chain_kernel<NU_, false>), grid, block, 0, st, k);               \
    } while (0)
    if (C == 128) BMC_LAUNCH_CHAIN(8);
    else if (C == 64) BMC_LAUNCH_CHAIN(4);
    else BMC_LAUNCH_CHAIN(2);
#undef BMC_LAUNCH_CHAIN
    return 0;
}

bool src_ok(const bmc_src_t& s, int C) {
    return s.ptr && s.nch == C && s.pix_stride % 4 == 0 && s.batch_stride % 4 == 0 && ((uintptr_t)s.ptr & 15) == 0;
}

}  // namespace

extern "C" int bmc_chain_fwd(const bmc_chain_fwd_args_t* h, bmc_stream_t stream) {
    BMC_CHECK_ARG(h != nullptr, "bmc_chain_fwd: null args");
    BMC_CHECK_ARG(h->C == 32 || h->C == 64 || h->C == 128, "bmc_chain_fwd: C=%d unsupported (32, 64, 128)", h->C);
    BMC_CHECK_ARG(h->B > 0 && h->H > 0 && h->W > 0, "bmc_chain_fwd: bad shape");
    BMC_CHECK_ARG(src_ok(h->s0, h->C) && src_ok(h->s1, h->C), "bmc_chain_fwd: sources must carry C channels, 16-byte aligned");
    BMC_CHECK_ARG(h->wstream && h->bias_f && h->bias_c && h->gamma && h->beta && h->yhat && h->rstd && h->centre,
                  "bmc_chain_fwd: null pointer");
    ChainK k = {};
    k.src[0] = to_dev(h->s0); k.src[1] = to_dev(h->s1);
    k.w = h->wstream;
    k.vec[0] = h->bias_f; k.vec[1] = h->bias_c; k.vec[2] = h->gamma; k.vec[3] = h->beta;
    k.eps = h->eps;
    k.out0 = h->yhat; k.out1 = h->centre; k.out2 = h->rstd;
    k.nb = h->B; k.n = 0; k.H = h->H; k.W = h->W;
    k.tiles_x = (h->W + TW - 1) / TW; k.tiles_y = (h->H + TH - 1) / TH;
    const long long nu = (long long)h->B * k.tiles_x * k.tiles_y;
    BMC_CHECK_ARG(nu < (1ll << 31), "bmc_chain_fwd: too many tiles");
    k.nunits = (int)nu;
    launch(k, h->C, false, (hipStream_t)stream);
    BMC_CHECK_LAUNCH("bmc_chain_fwd");
    return 0;
}

extern "C" int bmc_chain_bwd(const bmc_chain_bwd_args_t* h, bmc_stream_t stream) {
    BMC_CHECK_ARG(h != nullptr, "bmc_chain_bwd: null args");
    BMC_CHECK_ARG(h->C == 32 || h->C == 64 || h->C == 128, "bmc_chain_bwd: C=%d unsupported (32, 64, 128)", h->C);
    BMC_CHECK_ARG(h->n > 0 && h->H > 0 && h->W > 0, "bmc_chain_bwd: bad shape");
    BMC_CHECK_ARG(src_ok(h->dcentre, h->C), "bmc_chain_bwd: dcentre must carry C channels, 16-byte aligned");
    BMC_CHECK_ARG(h->wstream && h->gamma && h->yhat && h->rstd && h->dz && h->ds1 && h->ds0, "bmc_chain_bwd: null pointer");
    BMC_CHECK_ARG(!h->ds0_add.ptr || (h->ds0_add.pix_stride % 4 == 0 && ((uintptr_t)h->ds0_add.ptr & 15) == 0),
                  "bmc_chain_bwd: ds0_add must be 16-byte granular");
    ChainK k = {};
    k.src[0] = to_dev(h->dcentre); k.src[1] = k.src[0];
    k.w = h->wstream;
    k.vec[2] = h->gamma;
    k.out0 = h->dz; k.out1 = h->ds1; k.out2 = h->ds0;
    k.in0 = h->yhat; k.in1 = h->rstd;
    k.res = to_dev(h->ds0_add);
    k.nb = h->n; k.n = h->n; k.H = h->H; k.W = h->W;
    k.tiles_x = (h->W + TW - 1) / TW; k.tiles_y = (h->H + TH - 1) / TH;
    const long long nu = (long long)h->n * k.tiles_x * k.tiles_y;
    BMC_CHECK_ARG(nu < (1ll << 31), "bmc_chain_bwd: too many tiles");
    k.nunits = (int)nu;
    launch(k, h->C, true, (hipStream_t)stream);
    BMC_CHECK_LAUNCH("bmc_chain_bwd");
    return 0;
}

extern "C" int bmc_chain_affine_grads(const float* G, const float* dbc, const float* Wc, const float* gamma, const float* beta,
                                      int C, float* dwc, float* dbc_out, float* dgamma, float* dbeta, int accumulate,
                                      bmc_stream_t stream) {
    BMC_CHECK_ARG(G && dbc && Wc && gamma && beta && dwc && dgamma && dbeta && C > 0, "bmc_chain_affine_grads: bad arguments");
    hipLaunchKernelGGL(affine_grads_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, G, dbc, Wc, gamma, beta, C, dwc, dbc_out,
                       dgamma, dbeta, accumulate);
    BMC_CHECK_LAUNCH("bmc_chain_affine_grads");
    return 0;
}
