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

namespace {

constexpr int CK = BMC_CK;
constexpr int RS = 20;            // LDS row stride in floats (16 + 4 pad)
constexpr int TW = 16, TH = 8, NPX = TW * TH;
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

template <int NU, bool BWD>
__global__ __launch_bounds__(256, NU == 4 ? 1 : 2) void chain_kernel(const ChainK a) {
    constexpr int C = 32 * NU;
    constexpr int NR = C / CK;                    // steps of one register-operand GEMM (K = C)
    constexpr int NS = BWD ? 6 * NR : 3 * NR;     // steps per unit
    constexpr int NXC = 2 * NR;                   // X chunks per unit (two segments of C channels)
    constexpr int NXLD = 2;                       // 128 px x 4 float4 / 256 threads
    constexpr int NWLD = (C * 4 + 255) / 256;
    constexpr int XBUF = NPX * RS, WBUF = C * RS;
    __shared__ __attribute__((aligned(16))) float lds[2 * XBUF + 2 * WBUF + 2 * 8 + 5 * C];
    float* const Xb = lds;
    float* const Wb = lds + 2 * XBUF;
    SrcDev* const tab = reinterpret_cast<SrcDev*>(lds + 2 * XBUF + 2 * WBUF);
    float* const img = lds + 2 * XBUF + 2 * WBUF + 2 * 8;     // [0] b_f | 0   [1] b_c | 0   [2] gamma   [3] beta   [4] zeros

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
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
    const int q4 = (tid & 3) * 4;

    // ---- X loader: chunks of 16 channels in consumption order; a unit has two segments of C channels
    //      (fwd: s0 then s1 at batch b; bwd: dcentre at batch b then at batch b + n)
    int xl_unit = t_first, xl_seg = 0, c_in = 0, xl_cnt = 0;
    UnitIt xl_it = decode(t_first);
    const float* sbase = nullptr;
    int spix = 0;
    int xpix[NXLD];
    bool xok[NXLD];
    auto seg_select = [&]() {
        const SrcDev S = tab[BWD ? 0 : xl_seg];
        sbase = src_batch_ptr(S, BWD ? xl_it.b + xl_seg * a.n : xl_it.b);
        spix = S.pix_stride;
    };
    auto xl_setup = [&]() {
        const int y0 = xl_it.ty * TH, x0 = xl_it.tx * TW;
#pragma unroll
        for (int n = 0; n < NXLD; ++n) {
            const int hp = (tid + 256 * n) >> 2;
            const int y = y0 + hp / TW, x = x0 + hp % TW;
            xok[n] = y < a.H && x < a.W;
            xpix[n] = y * a.W + x;
        }
        xl_seg = 0; c_in = 0;
        seg_select();
    };
    f32x4 xr[NXLD], wr[NWLD];
    auto load_x = [&]() {
        const float* base = sbase + c_in + q4;
#pragma unroll
        for (int n = 0; n < NXLD; ++n) {
            const float* src = xok[n] ? base + (long long)xpix[n] * spix : g_zero4c;
            xr[n] = *reinterpret_cast<const f32x4*>(src);
        }
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
    auto store_x = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NXLD; ++n) {
            const int hp = (tid + 256 * n) >> 2;
            *reinterpret_cast<f32x4*>(Xb + buf * XBUF + hp * RS + q4) = xr[n];
        }
    };
    // ---- W loader: NS slices per unit, the same for every unit
    int wl_step = 0;
    auto load_w = [&]() {
        const float* p = a.w + (long long)wl_step * (C * CK);
#pragma unroll
        for (int n = 0; n < NWLD; ++n) {
            const int e = tid + 256 * n;
            const int ec = (n + 1) * 256 <= C * 4 ? e : (e < C * 4 ? e : C * 4 - 1);
            wr[n] = *reinterpret_cast<const f32x4*>(p + ec * 4);
        }
        if (++wl_step == NS) wl_step = 0;
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NWLD; ++n) {
            const int e = tid + 256 * n;
            if ((n + 1) * 256 <= C * 4 || e < C * 4) *reinterpret_cast<f32x4*>(Wb + buf * WBUF + (e >> 2) * RS + q4) = wr[n];
        }
    };

    // ---- fragments
    const int aoff = ((2 * wave + (li >> 4)) * TW + (li & 15)) * RS + 4 * lh;
    int boff[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) boff[u] = (32 * u + li) * RS + 4 * lh;
    f32x4 af0, af1, bf0[NU], bf1[NU];
    auto read_a = [&](const float* xb, int kg, f32x4& af) { af = *reinterpret_cast<const f32x4*>(xb + aoff + 8 * kg); };
    auto read_b = [&](const float* wb, int kg, f32x4 (&bf)[NU]) {
#pragma unroll
        for (int u = 0; u < NU; ++u) bf[u] = *reinterpret_cast<const f32x4*>(wb + boff[u] + 8 * kg);
    };

    f32x16 Ra[NU], Rt[NU], Rx[BWD ? NU : 1];
    auto init_acc = [&](f32x16 (&R)[NU], int which) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(img + which * C + 32 * u + 8 * rq + 4 * lh);
#pragma unroll
                for (int k = 0; k < 4; ++k) R[u][4 * rq + k] = v[k];
            }
    };
    auto mfma_lds = [&](const f32x4& af, const f32x4 (&bf)[NU], f32x16 (&acc)[NU]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[u][j], af[j], acc[u], 0, 0, 0);
    };

    // ---- pipeline state
    int gs = 0;        // global step of this workgroup (W buffer parity)
    int xc = 0;        // next X chunk to be consumed (its LDS buffer is xc & 1)
    bool x_full;       // the X register slot holds a loaded, not yet stored chunk

    // One step whose pixel operand comes from LDS (chunk xc).  next_lds: the following step also reads an X chunk.
    auto lds_step = [&](f32x16 (&acc)[NU], bool next_lds) {
        const bool has_next = gs + 1 < total_steps;
        bool stored = false;
        read_a(Xb + (xc & 1) * XBUF, 1, af1);
        read_b(Wb + (gs & 1) * WBUF, 1, bf1);
        if (has_next) store_w((gs + 1) & 1);
        if (has_next && next_lds && x_full) { store_x((xc + 1) & 1); stored = true; }
        mfma_lds(af0, bf0, acc);
        __syncthreads();
        if (has_next) {
            read_b(Wb + ((gs + 1) & 1) * WBUF, 0, bf0);
            if (next_lds) read_a(Xb + ((xc + 1) & 1) * XBUF, 0, af0);
        }
        if (gs + 2 < total_steps) load_w();
        if (stored) { x_full = xl_cnt < total_xchunks; if (x_full) load_x(); }
        mfma_lds(af1, bf1, acc);
        ++gs; ++xc;
    };
    // NR steps whose pixel operand is the register tile `op` (a previous result); next_lds refers to the step after the
    // last one (the first step of the next LDS phase consumes chunk xc).
    auto reg_steps = [&](const f32x16 (&op)[NU], f32x16 (&acc)[NU], bool next_lds) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const bool has_next = gs + 1 < total_steps;
            const bool nl = (c == NR - 1) && next_lds;
            bool stored = false;
            read_b(Wb + (gs & 1) * WBUF, 1, bf1);
            if (has_next) store_w((gs + 1) & 1);
            if (has_next && nl && x_full) { store_x(xc & 1); stored = true; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf0[u][j], op[c >> 1][4 * (2 * (c & 1) + 0) + j], acc[u], 0, 0, 0);
            __syncthreads();
            if (has_next) {
                read_b(Wb + ((gs + 1) & 1) * WBUF, 0, bf0);
                if (nl) read_a(Xb + (xc & 1) * XBUF, 0, af0);
            }
            if (gs + 2 < total_steps) load_w();
            if (stored) { x_full = xl_cnt < total_xchunks; if (x_full) load_x(); }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1[u][j], op[c >> 1][4 * (2 * (c & 1) + 1) + j], acc[u], 0, 0, 0);
            ++gs;
        }
    };

    // ---- per-lane output geometry of the current unit
    UnitIt ep_it = decode(t_first);
    int ep_unit = t_first;
    bool pok = false;
    int pix = 0;
    auto ep_setup = [&]() {
        const int y = ep_it.ty * TH + 2 * wave + (li >> 4), x = ep_it.tx * TW + (li & 15);
        pok = y < a.H && x < a.W;
        pix = y * a.W + x;
    };
    const long long img_elems = (long long)a.H * a.W;
    auto store_tile = [&](const f32x16 (&R)[NU], float* base) {     // base: batch plane [H][W][C]
        if (!pok) return;
        float* p = base + (long long)pix * C + 4 * lh;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = R[u][4 * rq + k];
                *reinterpret_cast<f32x4*>(p + 32 * u + 8 * rq) = v;
            }
    };

    // ---- prologue
    xl_setup();
    load_x();
    load_w();
    store_x(0);
    store_w(0);
    if (total_steps > 1) load_w();
    x_full = xl_cnt < total_xchunks;
    if (x_full) load_x();
    __syncthreads();
    read_a(Xb, 0, af0);
    read_b(Wb, 0, bf0);
    init_acc(Ra, BWD ? 4 : 0);

    for (; ep_unit < t_hi; ep_unit += t_stride) {
        ep_it = decode(ep_unit);
        ep_setup();
        const bool more_units = ep_unit + t_stride < t_hi;
        if constexpr (!BWD) {
            // ---- z = convf(cat[s0, s1]) + b_f
            for (int i = 0; i < NXC; ++i) lds_step(Ra, i + 1 < NXC);
            // ---- LayerNorm over the C channels of each pixel: registers of lanes l and l ^ 32
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += Ra[u][r];
            s += __shfl_xor(s, 32);
            const float mu = s * (1.f / C);
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = Ra[u][r] - mu;
                    Ra[u][r] = d;
                    q += d * d;
                }
            q += __shfl_xor(q, 32);
            const float rstd = 1.f / sqrtf(q * (1.f / C) + a.eps);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) Ra[u][r] *= rstd;
            store_tile(Ra, a.out0 + (long long)ep_it.b * img_elems * C);
            if (pok && lh == 0) a.out2[(long long)ep_it.b * img_elems + pix] = rstd;
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 gq = *reinterpret_cast<const f32x4*>(img + 2 * C + 32 * u + 8 * rq + 4 * lh);
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(img + 3 * C + 32 * u + 8 * rq + 4 * lh);
#pragma unroll
                    for (int k = 0; k < 4; ++k) Ra[u][4 * rq + k] = Ra[u][4 * rq + k] * gq[k] + bq[k];
                }
            // ---- centre = clustering(y) + b_c, pixel operand = the registers of y
            init_acc(Rt, 1);
            reg_steps(Ra, Rt, more_units);
            store_tile(Rt, a.out1 + (long long)ep_it.b * img_elems * C);
            init_acc(Ra, 0);
        } else {
            for (int half = 0; half < 2; ++half) {
                const int bb = ep_it.b + half * a.n;
                // ---- dy = W_c^T dcentre
                for (int i = 0; i < NR; ++i) lds_step(Ra, i + 1 < NR);
                // ---- LayerNorm backward: g = dy*gamma, dz = rstd * (g - yhat*mean(g*yhat) - mean(g))
                const float* const yh = a.in0 + (long long)bb * img_elems * C + (long long)pix * C + 4 * lh;
                const float rstd = pok ? a.in1[(long long)bb * img_elems + pix] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 gq = *reinterpret_cast<const f32x4*>(img + 2 * C + 32 * u + 8 * rq + 4 * lh);
                        const f32x4 yq = *reinterpret_cast<const f32x4*>(pok ? yh + 32 * u + 8 * rq : g_zero4c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float g = Ra[u][4 * rq + k] * gq[k];
                            Ra[u][4 * rq + k] = g;
                            s1 += g;
                            s2 += g * yq[k];
                        }
                    }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                const float m1 = s1 * (1.f / C), m2 = s2 * (1.f / C);
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const f32x4 yq = *reinterpret_cast<const f32x4*>(pok ? yh + 32 * u + 8 * rq : g_zero4c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) Ra[u][4 * rq + k] = rstd * (Ra[u][4 * rq + k] - yq[k] * m2 - m1);
                    }
                store_tile(Ra, a.out0 + (long long)bb * img_elems * C);
                // ---- d s1 (written for the batch the forward READ s1 from) and d s0 (both halves summed in Rx)
                if (half == 0) {
                    init_acc(Rt, 4);
                    reg_steps(Ra, Rt, false);                                  // W_f[:, C:]^T dz
                    store_tile(Rt, a.out1 + (long long)((bb + a.n) % (2 * a.n)) * img_elems * C);
                    init_acc(Rt, 4);
                    reg_steps(Ra, Rt, true);                                   // W_f[:, :C]^T dz (first half)
#pragma unroll
                    for (int u = 0; u < NU; ++u) Rx[u] = Rt[u];
                } else {
#pragma unroll
                    for (int u = 0; u < NU; ++u) Rt[u] = Rx[u];
                    reg_steps(Ra, Rt, false);                                  // + W_f[:, :C]^T dz (second half)
                    if (a.res.ptr && pok) {
                        const float* rp = src_batch_ptr(a.res, ep_it.b) + (long long)pix * a.res.pix_stride + 4 * lh;
#pragma unroll
                        for (int u = 0; u < NU; ++u)
#pragma unroll
                            for (int rq = 0; rq < 4; ++rq) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(rp + 32 * u + 8 * rq);
#pragma unroll
                                for (int k = 0; k < 4; ++k) Rt[u][4 * rq + k] += v[k];
                            }
                    }
                    store_tile(Rt, a.out2 + (long long)ep_it.b * img_elems * C);
                    init_acc(Rt, 4);
                    reg_steps(Ra, Rt, more_units);                             // W_f[:, C:]^T dz
                    store_tile(Rt, a.out1 + (long long)((bb + a.n) % (2 * a.n)) * img_elems * C);
                }
                init_acc(Ra, 4);
            }
        }
    }
}

// LayerNorm affine gradients from the pixel-reduction GEMM of the clustering convolution taken on yhat:
//   G[co][ci] = sum_px dcentre[px][co] yhat[px][ci],   dbc[co] = sum_px dcentre[px][co]
//   dW_c[co][ci] = gamma[ci] G[co][ci] + dbc[co] beta[ci]        (y = yhat*gamma + beta)
//   dgamma[ci]   = sum_co W_c[co][ci] G[co][ci]                    (= sum_px dy*yhat, dy = W_c^T dcentre)
//   dbeta[ci]    = sum_co W_c[co][ci] dbc[co]                      (= sum_px dy)
// One block, thread per ci; results are written (accumulate = 0) or added (1) to dwc / dgamma / dbeta (dwc may be G
// itself); dbc_out, if given, receives dbc the same way (the clustering bias gradient on its way to its accumulator).
__global__ void affine_grads_kernel(const float* G, const float* __restrict__ dbc, const float* __restrict__ Wc,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int C, float* dwc,
                                    float* __restrict__ dbc_out, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                    int accumulate) {
    for (int ci = threadIdx.x; ci < C; ci += blockDim.x) {
        if (dbc_out) dbc_out[ci] = accumulate ? dbc_out[ci] + dbc[ci] : dbc[ci];
        float dg = 0.f, db = 0.f;
        const float gm = gamma[ci], bt = beta[ci];
        for (int co = 0; co < C; ++co) {
            const float g = G[co * C + ci], w = Wc[co * C + ci], d = dbc[co];
            dg += w * g;
            db += w * d;
            const float v = gm * g + d * bt;
            dwc[co * C + ci] = accumulate ? dwc[co * C + ci] + v : v;
        }
        dgamma[ci] = accumulate ? dgamma[ci] + dg : dg;
        dbeta[ci] = accumulate ? dbeta[ci] + db : db;
    }
}

int launch(const ChainK& k, int C, bool bwd, hipStream_t st) {
    const int cus = bmc_num_cus();
    const int max_blocks = (C == 128 ? 1 : 2) * cus;      // resident workgroups per CU (registers: 128 accumulators per lane at C = 128)
    dim3 grid((unsigned)(k.nunits < max_blocks ? k.nunits : max_blocks)), block(256);
#define BMC_LAUNCH_CHAIN(NU_)                                                                    \
    do {                                                                                          \
        if (bwd) hipLaunchKernelGGL((chain_kernel<NU_, true>), grid, block, 0, st, k);            \
        else hipLaunchKernelGGL((chain_kernel<NU_, false>), grid, block, 0, st, k);               \
    } while (0)
    if (C == 128) BMC_LAUNCH_CHAIN(4);
    else if (C == 64) BMC_LAUNCH_CHAIN(2);
    else BMC_LAUNCH_CHAIN(1);
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
    hipLaunchKernelGGL(affine_grads_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, G, dbc, Wc, gamma, beta, C, dwc, dbc_out,
                       dgamma, dbeta, accumulate);
    BMC_CHECK_LAUNCH("bmc_chain_affine_grads");
    return 0;
}
