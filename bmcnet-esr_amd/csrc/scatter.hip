// Event -> two-channel count image (dataloader/encodings.py:241-269, 290-305), many frames per launch.
// Integer bin indices, +p*p contributions (exact integers for p = +-1) accumulated with
// global_atomic_add_f32: the sum is order independent below 2^24, hence bit-exact.
#include "bmc_common.h"

namespace {

__global__ void events_kernel(float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ ps,
                              const long long* __restrict__ offsets, int H, int W, float* __restrict__ out, int mutate) {
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    float* const img = out + (long long)f * 2 * H * W;
    for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (long long)gridDim.x * blockDim.x) {
        float x = xs[e], y = ys[e];
        const float p = ps[e];
        // first events_to_image() call (positive channel): out-of-range events are zeroed IN PLACE
        // (encodings.py:249-254) -- weight dropped, coordinates reset to (0, 0)
        const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (oob) {
            x = 0.f; y = 0.f;
            if (mutate) { xs[e] = 0.f; ys[e] = 0.f; }
        }
        const int xi = (int)x;            // .long(): truncation (encodings.py:260-263)
        const int yi = H - (int)y - 1;    // vertical flip (encodings.py:265)
        const float wpos = (!oob && p > 0.f) ? p * p : 0.f;  // ps * mask_pos, mask_pos = p where p >= 0
        // second call (negative channel) sees the already-reset coordinates: nothing is masked any more,
        // so a formerly out-of-range negative event counts at [H-1, 0]
        const float wneg = p < 0.f ? p * p : 0.f;
        if (wpos != 0.f) atomicAdd(img + (long long)yi * W + xi, wpos);
        if (wneg != 0.f) atomicAdd(img + (long long)H * W + (long long)yi * W + xi, wneg);
    }
}

// Raw HDF5 columns (generate_dataset/tools/event_packagers.py:128-156: xs/ys int16, ps float64) -> count images, with
// the dataset's flip augmentation (dataloader/h5dataset.py:559-578) folded into the index arithmetic:
// flags bit0: x = W-1-x, bit1: y = H-1-y, bit2: p = -p; then event_formatting's float32 cast
// (dataloader/base_dataset.py:24-31) and the events_to_channels() semantics above.
__global__ void encode_raw_kernel(const short* __restrict__ xs, const short* __restrict__ ys, const double* __restrict__ ps,
                                  const long long* __restrict__ offsets, const unsigned char* __restrict__ flips, int H, int W,
                                  float* __restrict__ out) {
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    const int fl = flips ? flips[f] : 0;
    float* const img = out + (long long)f * 2 * H * W;
    for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (long long)gridDim.x * blockDim.x) {
        double xd = (double)xs[e], yd = (double)ys[e], pd = ps[e];
        if (fl & 1) xd = (double)(W - 1) - xd;
        if (fl & 2) yd = (double)(H - 1) - yd;
        if (fl & 4) pd = pd * -1.0;
        float x = (float)xd, y = (float)yd;
        const float p = (float)pd;
        const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (oob) { x = 0.f; y = 0.f; }
        const int xi = (int)x, yi = H - (int)y - 1;
        const float wpos = (!oob && p > 0.f) ? p * p : 0.f;
        const float wneg = p < 0.f ? p * p : 0.f;
        if (wpos != 0.f) atomicAdd(img + (long long)yi * W + xi, wpos);
        if (wneg != 0.f) atomicAdd(img + (long long)H * W + (long long)yi * W + xi, wneg);
    }
}

// ---- the same two encoders without scattered global float atomics (large frames) --------------------------------------
// A 720x960 ground-truth frame holds ~0.4 M events on random pixels: every lane of a global_atomic_add_f32 hits its own
// 64-byte segment, and MI355X executes float atomics at the memory side at ~0.08 TB/s in that shape
// (MI355X_MICROARCH.md, Global float atomics): 0.68 ms for the 36 HR frames of a C2 step, 10x the time their bytes need.
// Here the count image of a (frame, row band) is built in LDS by ONE workgroup (LDS float adds) and written out once with
// coalesced stores (which is also the zero fill: no memset); events reach their band's workgroup through a counting
// sort on the band index:
//   count : a workgroup walks 4 096-event chunks of a frame, histograms their bands in LDS and adds the histogram to the
//           frame's band counts (integer atomics, <= nbands per workgroup);
//   scan  : exclusive prefix sum over the (frame, band) count table -> segment starts;
//   place : the same walk again: per chunk a workgroup reserves its slots per band with one integer atomic each and drops
//           8-byte records (plane pixel, weight) there; zero-weight events are dropped, out-of-range coordinates are
//           reset in the caller's arrays if asked to;
//   gather: one workgroup per (frame, band) adds its records into the LDS image and stores it.
// Sums are order independent for the integer-valued weights p*p of +-1 polarities (exact below 2^24), as before.
constexpr int BIN_LDS = 32768;          // floats of LDS image per workgroup (128 KB): 2 channels x R rows x W
constexpr int BIN_CHUNK = 4096;         // events per chunk (256 threads x 16)
constexpr int BIN_MAXBANDS = 1024;

struct F32Events {       // events as event_formatting() leaves them (dataloader/base_dataset.py:24-31)
    float* xs; float* ys; const float* ps; int mutate;
    __device__ __forceinline__ void load(long long e, int, int, int, float& x, float& y, float& p) const { x = xs[e]; y = ys[e]; p = ps[e]; }
    __device__ __forceinline__ void reset(long long e) const { if (mutate) { xs[e] = 0.f; ys[e] = 0.f; } }
};
struct RawEvents {       // raw dataset columns + flip flags (see encode_raw_kernel)
    const short* xs; const short* ys; const double* ps; const unsigned char* flips;
    __device__ __forceinline__ void load(long long e, int f, int H, int W, float& x, float& y, float& p) const {
        const int fl = flips ? flips[f] : 0;
        double xd = (double)xs[e], yd = (double)ys[e], pd = ps[e];
        if (fl & 1) xd = (double)(W - 1) - xd;
        if (fl & 2) yd = (double)(H - 1) - yd;
        if (fl & 4) pd = pd * -1.0;
        x = (float)xd; y = (float)yd; p = (float)pd;
    }
    __device__ __forceinline__ void reset(long long) const {}
};
// -> plane pixel (channel * H*W + row * W + column) and weight of the event's one contribution, or weight 0 (none):
// positive events count in channel 0 unless out of range; negative ones in channel 1, out-of-range ones at [H-1, 0]
__device__ __forceinline__ void event_record(float x, float y, float p, int H, int W, bool& oob, int& pix, float& wgt) {
    oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
    if (oob) { x = 0.f; y = 0.f; }
    const int xi = (int)x, yi = H - (int)y - 1;
    const bool neg = p < 0.f;
    wgt = (neg || (!oob && p > 0.f)) ? p * p : 0.f;
    pix = (neg ? H * W : 0) + yi * W + xi;
}
// PLACE = false: count pass (table += histogram).  PLACE = true: table holds the segment starts; records are written.
template <typename EV, bool PLACE>
__global__ __launch_bounds__(256) void bin_events_kernel(const EV ev, const long long* __restrict__ offsets, int H, int W, int R,
                                                          int nbands, int* __restrict__ table, int* __restrict__ cursors,
                                                          unsigned long long* __restrict__ records) {
    __shared__ int hist[BIN_MAXBANDS];
    __shared__ int base[BIN_MAXBANDS];
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    for (int i = threadIdx.x; i < nbands; i += 256) hist[i] = 0;
    __syncthreads();
    const int HW = H * W;
    for (long long c0 = e0 + (long long)blockIdx.x * BIN_CHUNK; c0 < e1; c0 += (long long)gridDim.x * BIN_CHUNK) {
        int band[BIN_CHUNK / 256], rank[BIN_CHUNK / 256], pix[BIN_CHUNK / 256];
        float wgt[BIN_CHUNK / 256];
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / 256; ++k) {
            const long long e = c0 + k * 256 + threadIdx.x;
            band[k] = -1;
            if (e < e1) {
                float x, y, p;
                ev.load(e, f, H, W, x, y, p);
                bool oob;
                event_record(x, y, p, H, W, oob, pix[k], wgt[k]);
                if (PLACE && oob) ev.reset(e);
                if (wgt[k] != 0.f) {
                    const int q = pix[k] >= HW ? pix[k] - HW : pix[k];
                    band[k] = (q / W) / R;
                    rank[k] = atomicAdd(&hist[band[k]], 1);
                }
            }
        }
        if (!PLACE) continue;         // the count pass keeps adding to one histogram
        __syncthreads();
        for (int i = threadIdx.x; i < nbands; i += 256) {
            const int h = hist[i];
            base[i] = h ? atomicAdd(cursors + (long long)f * nbands + i, h) : 0;
            hist[i] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BIN_CHUNK / 256; ++k)
            if (band[k] >= 0) {
                const long long slot = (long long)table[(long long)f * nbands + band[k]] + base[band[k]] + rank[k];
                records[slot] = ((unsigned long long)__float_as_uint(wgt[k]) << 32) | (unsigned)pix[k];
            }
        // (the next chunk's LDS adds come after the barrier above; its reservation after the one at the loop's first barrier)
    }
    if (!PLACE) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbands; i += 256)
            if (hist[i]) atomicAdd(table + (long long)f * nbands + i, hist[i]);
    }
}
// exclusive prefix sum of table[0 .. n) in place, table[n] = total; one workgroup (n is a few thousand at most)
__global__ __launch_bounds__(1024) void bin_scan_kernel(int* __restrict__ table, int n) {
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024;
    const int lo = threadIdx.x * per < n ? threadIdx.x * per : n, hi = lo + per < n ? lo + per : n;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += table[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
        table[n] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = table[i]; table[i] = run; run += v; }
}
__global__ __launch_bounds__(1024) void bin_gather_kernel(const unsigned long long* __restrict__ records, const int* __restrict__ starts,
                                                          int H, int W, int R, int nbands, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float img[BIN_LDS];
    const int f = blockIdx.y, b = blockIdx.x;
    const int r0 = b * R, rows = r0 + R <= H ? R : H - r0;
    const int n1 = rows * W;                       // floats per channel in this band
    for (int i = threadIdx.x; i < 2 * n1; i += 1024) img[i] = 0.f;
    __syncthreads();
    const int s0 = starts[(long long)f * nbands + b], s1 = starts[(long long)f * nbands + b + 1];   // (entry ntab = the total)
    const int HW = H * W;
    for (int i = s0 + threadIdx.x; i < s1; i += 1024) {
        const unsigned long long rec = records[i];
        const int pix = (int)(unsigned)rec;
        const int ch = pix >= HW, q = pix - ch * HW - r0 * W;
        atomicAdd(&img[ch * n1 + q], __uint_as_float((unsigned)(rec >> 32)));
    }
    __syncthreads();
    float* const o = out + (long long)f * 2 * HW + (long long)r0 * W;
    for (int ch = 0; ch < 2; ++ch)
        for (int i = threadIdx.x; i < n1; i += 1024) o[(long long)ch * HW + i] = img[ch * n1 + i];
}
template <typename EV>
int launch_binned(const EV& ev, const long long* offsets, long long nevents, int nframes, int H, int W, float* out, void* ws,
                  long long ws_bytes, hipStream_t st, const char* name) {
    const int R = BIN_LDS / 2 / W < H ? BIN_LDS / 2 / W : H, nbands = R > 0 ? (H + R - 1) / R : 0;
    BMC_CHECK_ARG(R >= 1 && nbands <= BIN_MAXBANDS, "%s: frame %dx%d does not fit the LDS band scheme", name, H, W);
    BMC_CHECK_ARG(nevents >= 0 && nevents < (1ll << 31) && (long long)H * W < (1ll << 30), "%s: too many events / pixels", name);
    const long long ntab = (long long)nframes * nbands;
    BMC_CHECK_ARG(ntab <= 1024 * 1024, "%s: too many (frame, band) pairs", name);
    const long long head = ((2 * ntab + 1) * 4 + 7) / 8 * 8;
    BMC_CHECK_ARG(ws && ws_bytes >= head + nevents * 8 && ((uintptr_t)ws & 7) == 0,
                  "%s: workspace of %lld bytes needed (8-byte aligned)", name, head + nevents * 8);
    int* const table = static_cast<int*>(ws);                       // [ntab + 1]: counts, then segment starts (+ total)
    int* const cursors = table + ntab + 1;                          // [ntab]
    unsigned long long* const records = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + head);
    hipError_t e = hipMemsetAsync(table, 0, (size_t)(2 * ntab + 1) * 4, st);
    if (e != hipSuccess) { bmc_set_error("%s: memset failed: %s", name, hipGetErrorString(e)); return -2; }
    // workgroups per frame: enough chunks in flight for the longest frame the total allows, at most 128
    long long per = (nevents + BIN_CHUNK - 1) / BIN_CHUNK;
    if (per > 128) per = 128;
    if (per < 1) per = 1;
    const dim3 grid((unsigned)per, (unsigned)nframes);
    hipLaunchKernelGGL((bin_events_kernel<EV, false>), grid, dim3(256), 0, st, ev, offsets, H, W, R, nbands, table, cursors, records);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, st, table, (int)ntab);
    hipLaunchKernelGGL((bin_events_kernel<EV, true>), grid, dim3(256), 0, st, ev, offsets, H, W, R, nbands, table, cursors, records);
    hipLaunchKernelGGL(bin_gather_kernel, dim3((unsigned)nbands, (unsigned)nframes), dim3(1024), 0, st, records, table, H, W, R, nbands, out);
    return 0;
}

// Temporal-bilinear voxel grid (dataloader/encodings.py:272-287): bin b receives p * max(0, 1 - |t*(bins-1) - b|)
// through events_to_image(), i.e. with the vertical flip and with the same side effect as above: the FIRST bin's call
// zeroes out-of-range events and resets their coordinates in place, so in every later bin they are no longer masked
// and deposit their weight at [H-1, 0].  Weights are arbitrary floats, so the summation ORDER is part of the result:
// the reference's index_put_(accumulate=True) adds a pixel's events in event order (sequential CPU kernel).  Same
// order here, without float atomics (deterministic, bit-identical run to run and to the single-threaded reference):
//   count   : events per pixel (integer atomics);
//   scan    : exclusive prefix sum over the pixels of a frame (one workgroup per frame) -> segment offsets;
//   fill    : every event drops its index into its pixel's segment (arrival order, arbitrary);
//   reduce  : one thread per pixel sorts its segment by event index (segments are a handful of events) and sums the
//             bins' weights in that order.
__device__ __forceinline__ int voxel_pixel(float x, float y, int H, int W, bool& oob) {
    oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
    if (oob) { x = 0.f; y = 0.f; }
    return (H - (int)y - 1) * W + (int)x;
}
__global__ void voxel_count_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                   const long long* __restrict__ offsets, int H, int W, int* __restrict__ cnt) {
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    int* const c = cnt + (long long)f * (H * W + 1);
    for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (long long)gridDim.x * blockDim.x) {
        bool oob;
        atomicAdd(c + voxel_pixel(xs[e], ys[e], H, W, oob), 1);
    }
}
// in-place exclusive scan of cnt[f][0 .. HW] (entry HW receives the total); cur[f][q] = start of segment q (fill cursor)
__global__ void voxel_scan_kernel(int* __restrict__ cnt, int* __restrict__ cur, int HW) {
    int* const c = cnt + (long long)blockIdx.x * (HW + 1);
    int* const u = cur + (long long)blockIdx.x * (HW + 1);
    const int per = (HW + blockDim.x - 1) / blockDim.x;
    const int lo = threadIdx.x * per, hi = lo + per < HW ? lo + per : HW;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += c[i];
    __shared__ int part[1024];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { const int v = part[i]; part[i] = run; run += v; }
        c[HW] = run;
        u[HW] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = c[i]; c[i] = run; u[i] = run; run += v; }
}
__global__ void voxel_fill_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                  const long long* __restrict__ offsets, int H, int W, int* __restrict__ cur,
                                  int* __restrict__ idx) {
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    int* const u = cur + (long long)f * (H * W + 1);
    for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (long long)gridDim.x * blockDim.x) {
        bool oob;
        const int q = voxel_pixel(xs[e], ys[e], H, W, oob);
        idx[e0 + atomicAdd(u + q, 1)] = (int)(e - e0);
    }
}
// Long segments (hot pixels, and pixel (H-1)*W, where EVERY out-of-range event of a frame lands) are put in event order
// here, by the whole workgroup, before the per-pixel pass below: its single-thread insertion sort is O(k^2) moves on an
// unsorted segment (50 k events ~ 1e9 dependent global accesses) but O(k) on a sorted one.  One workgroup per frame
// collects the segments longer than VOX_LONG (up to 1 024 of them; any beyond that keep the slow path, still correct)
// and sorts each with a bitonic network written for arbitrary lengths: every compare-exchange moves the larger key to the
// higher index, so virtual +infinity padding above the segment's end never moves (out-of-range partners are skipped).
constexpr int VOX_LONG = 32;
__global__ __launch_bounds__(1024) void voxel_sort_long_kernel(const long long* __restrict__ offsets, int HW,
                                                               const int* __restrict__ seg, int* __restrict__ idx) {
    __shared__ int longq[1024];
    __shared__ int nlong;
    const int f = blockIdx.x;
    const int* const sg = seg + (long long)f * (HW + 1);
    int* const id = idx + offsets[f];
    if (threadIdx.x == 0) nlong = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < HW; q += 1024)
        if (sg[q + 1] - sg[q] > VOX_LONG) {
            const int s = atomicAdd(&nlong, 1);
            if (s < 1024) longq[s] = q;
        }
    __syncthreads();
    const int n = nlong < 1024 ? nlong : 1024;
    for (int s = 0; s < n; ++s) {
        const int a = sg[longq[s]], k = sg[longq[s] + 1] - a;
        int* const v = id + a;
        for (int size = 2; (size >> 1) < k; size <<= 1) {
            // first step of a merge: partner = mirror image inside the block of `size`; then halving strides
            for (int stride = size >> 1, first = 1; stride > 0; stride >>= 1, first = 0) {
                for (int t = threadIdx.x; t < (k + 1) / 2 + stride; t += 1024) {      // t enumerates the lower partners
                    const int blk = t / stride, off = t - blk * stride;
                    const int i = blk * 2 * stride + off;
                    const int j = first ? blk * 2 * stride + (2 * stride - 1 - off) : i + stride;
                    if (i < k && j < k) {
                        const int x = v[i], y = v[j];
                        if (x > y) { v[i] = y; v[j] = x; }
                    }
                }
                __syncthreads();
            }
        }
    }
}
__global__ void voxel_reduce_kernel(float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ ts,
                                    const float* __restrict__ ps, const long long* __restrict__ offsets, int bins, int H, int W,
                                    const int* __restrict__ seg, int* __restrict__ idx, float* __restrict__ out, int mutate) {
    const int f = blockIdx.y, HW = H * W;
    const long long e0 = offsets[f];
    const int* const sg = seg + (long long)f * (HW + 1);
    float* const vox = out + (long long)f * bins * HW;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < HW; q += gridDim.x * blockDim.x) {
        const int a = sg[q], b = sg[q + 1];
        int* const id = idx + e0;
        for (int i = a + 1; i < b; ++i) {          // insertion sort by event index: restores event order
            const int v = id[i];
            int j = i - 1;
            while (j >= a && id[j] > v) { id[j + 1] = id[j]; --j; }
            id[j + 1] = v;
        }
        for (int bi = 0; bi < bins; ++bi) {
            float acc = 0.f;
            for (int i = a; i < b; ++i) {
                const long long e = e0 + id[i];
                const float x = xs[e], y = ys[e];
                const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
                const float t = ts[e] * (float)(bins - 1);
                const float wgt = ps[e] * fmaxf(0.f, 1.0f - fabsf(t - (float)bi));
                if (!(oob && bi == 0)) acc += wgt;
            }
            vox[(long long)bi * HW + q] = acc;
        }
        if (mutate && q == (H - 1) * W)            // every out-of-range event sits in this pixel's segment: reset them
            for (int i = a; i < b; ++i) {
                const long long e = e0 + id[i];
                const float x = xs[e], y = ys[e];
                if ((x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f)) { xs[e] = 0.f; ys[e] = 0.f; }
            }
    }
}

// Event stack without polarity split (dataloader/encodings.py:202-238).  Three small kernels:
//   stack_search : 2 threads per bin run the reference's hand-written binary search (:75-97, quirks included: the
//                  first probe that EQUALS the bound wins, side 'right' returns r) on the float32 bounds the host
//                  computed with the reference's own float32 expressions -> [beg, end) per bin;
//   stack_scatter: bin b adds p at [(long) y, (long) x] (no vertical flip here) for its in-range events;
//   stack_mutate : events_to_image_torch() works on views of the caller's arrays, so every out-of-range event that
//                  falls in some bin's range ends up with xs = ys = ps = 0 (run after the scatter: the scatter reads
//                  the original values; a masked event contributes nothing in any bin either way).
__device__ int stack_bsearch(const float* __restrict__ t, int l, int r, float x, bool left) {
    while (l <= r) {
        if (t[l] == x) return l;
        if (t[r] == x) return r;
        const int mid = l + (r - l) / 2;
        const float mv = t[mid];
        if (mv == x) return mid;
        if (mv < x) l = mid + 1; else r = mid - 1;
    }
    return left ? l : r;
}
__global__ void stack_search_kernel(const float* __restrict__ ts, long long n, const float* __restrict__ tstart,
                                    const float* __restrict__ tend, int bins, int* __restrict__ ranges) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * bins) return;
    const int b = i >> 1;
    if (i & 1) ranges[i] = stack_bsearch(ts, 0, (int)n - 1, tend[b], false) + 1;
    else ranges[i] = stack_bsearch(ts, 0, (int)n - 1, tstart[b], true);
}
__global__ void stack_scatter_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ ps,
                                     const int* __restrict__ ranges, int H, int W, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int beg = ranges[2 * b], end = ranges[2 * b + 1];
    float* const img = out + (long long)b * H * W;
    for (int e = beg + blockIdx.x * blockDim.x + threadIdx.x; e < end; e += gridDim.x * blockDim.x) {
        const float x = xs[e], y = ys[e], p = ps[e];
        const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (!oob && p != 0.f) atomicAdd(img + (long long)(int)y * W + (int)x, p);
    }
}
__global__ void stack_mutate_kernel(float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ ps,
                                    const int* __restrict__ ranges, int H, int W) {
    const int b = blockIdx.y;
    const int beg = ranges[2 * b], end = ranges[2 * b + 1];
    for (int e = beg + blockIdx.x * blockDim.x + threadIdx.x; e < end; e += gridDim.x * blockDim.x) {
        const float x = xs[e], y = ys[e];
        // a neighbouring bin may have zeroed this event already (ranges can overlap by an event): (0, 0) is in range
        if ((x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f)) { xs[e] = 0.f; ys[e] = 0.f; ps[e] = 0.f; }
    }
}


// Event stack WITH polarity split (dataloader/encodings.py:151-199): per bin two count images [2][bins][H][W] =
// (positives, negatives), no vertical flip, weights p*p.  Per bin the reference calls events_to_image_torch twice on
// VIEWS of the caller's coordinates with a temporary weight vector: the first (positive) call of the FIRST bin that
// covers an event resets its out-of-range coordinates to (0, 0) and masks only that call; in every later call
// (the same bin's negative image, and both images of any later bin that covers the event again) it is in range at
// [0, 0].  Counts are integers: float atomics are exact.
__global__ void stack_pol_scatter_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ ps,
                                         const int* __restrict__ ranges, int bins, int H, int W, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int beg = ranges[2 * b], end = ranges[2 * b + 1];
    float* const pos = out + (long long)b * H * W;
    float* const neg = out + ((long long)bins + b) * H * W;
    for (int e = beg + blockIdx.x * blockDim.x + threadIdx.x; e < end; e += gridDim.x * blockDim.x) {
        const float x = xs[e], y = ys[e], p = ps[e];
        const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (p == 0.f) continue;
        if (!oob) {
            atomicAdd((p > 0.f ? pos : neg) + (long long)(int)y * W + (int)x, p * p);
        } else {
            bool first = true;          // did an earlier bin already cover (and reset) this event?
            for (int bb = 0; bb < b; ++bb) first = first && !(e >= ranges[2 * bb] && e < ranges[2 * bb + 1]);
            if (p < 0.f) atomicAdd(neg, p * p);
            else if (!first) atomicAdd(pos, p * p);
        }
    }
}
__global__ void stack_pol_mutate_kernel(float* __restrict__ xs, float* __restrict__ ys, const int* __restrict__ ranges, int H,
                                        int W) {
    const int b = blockIdx.y;
    const int beg = ranges[2 * b], end = ranges[2 * b + 1];
    for (int e = beg + blockIdx.x * blockDim.x + threadIdx.x; e < end; e += gridDim.x * blockDim.x) {
        const float x = xs[e], y = ys[e];
        if ((x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f)) { xs[e] = 0.f; ys[e] = 0.f; }
    }
}

// Binary event mask (dataloader/encodings.py:308-332): out-of-range events get xs = ys = ps = 0 in place, then
// mask[(long) y][(long) x] = |p| with index_put_(accumulate=False): for a pixel hit several times the LAST event in
// order wins (sequential semantics; a zeroed out-of-range event can clear [0, 0]).  Deterministic here: atomicMax of the
// event index per pixel, then the winner writes.
__global__ void mask_owner_kernel(const float* __restrict__ xs, const float* __restrict__ ys, long long n, int H, int W,
                                  int* __restrict__ owner) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        float x = xs[e], y = ys[e];
        if ((x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f)) { x = 0.f; y = 0.f; }
        atomicMax(owner + (long long)(int)y * W + (int)x, (int)e + 1);
    }
}
__global__ void mask_write_kernel(float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ ps, long long n, int H, int W,
                                  const int* __restrict__ owner, float* __restrict__ out, int mutate) {
    const long long hw = (long long)H * W;
    const long long total = hw > n ? hw : n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < hw) {
            const int o = owner[i];
            float v = 0.f;
            if (o > 0) {
                const float x = xs[o - 1], y = ys[o - 1];
                const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
                v = oob ? 0.f : fabsf(ps[o - 1]);
            }
            out[i] = v;
        }
    }
}
__global__ void mask_mutate_kernel(float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ ps, long long n, int H,
                                   int W) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const float x = xs[e], y = ys[e];
        if ((x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f)) { xs[e] = 0.f; ys[e] = 0.f; ps[e] = 0.f; }
    }
}

}  // namespace

extern "C" int bmc_events_to_voxel(float* xs, float* ys, const float* ts, const float* ps, const long long* offsets,
                                   long long nevents, int nframes, int bins, int H, int W, float* out, int mutate, int* ws,
                                   bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && bins >= 1 && H > 0 && W > 0 && out && ws, "bmc_events_to_voxel: bad shape / null workspace");
    BMC_CHECK_ARG(nevents >= 0 && nevents < (1ll << 31), "bmc_events_to_voxel: event count out of range");
    hipStream_t st = (hipStream_t)s;
    if (nframes == 0) return 0;
    const long long segn = (long long)nframes * (H * W + 1);
    int* const seg = ws;                 // [nframes][HW + 1] segment offsets
    int* const cur = ws + segn;          // [nframes][HW + 1] fill cursors
    int* const idx = ws + 2 * segn;      // [nevents] event indices grouped by pixel
    hipError_t e = hipMemsetAsync(seg, 0, (size_t)segn * sizeof(int), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_voxel: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(voxel_count_kernel, dim3(64, nframes), dim3(256), 0, st, xs, ys, offsets, H, W, seg);
    hipLaunchKernelGGL(voxel_scan_kernel, dim3(nframes), dim3(1024), 0, st, seg, cur, H * W);
    hipLaunchKernelGGL(voxel_fill_kernel, dim3(64, nframes), dim3(256), 0, st, xs, ys, offsets, H, W, cur, idx);
    hipLaunchKernelGGL(voxel_sort_long_kernel, dim3(nframes), dim3(1024), 0, st, offsets, H * W, seg, idx);
    hipLaunchKernelGGL(voxel_reduce_kernel, dim3((H * W + 255) / 256, nframes), dim3(256), 0, st, xs, ys, ts, ps, offsets, bins,
                       H, W, seg, idx, out, mutate);
    BMC_CHECK_LAUNCH("bmc_events_to_voxel");
    return 0;
}

extern "C" int bmc_encode_raw_events(const short* xs, const short* ys, const double* ps, const long long* offsets,
                                     const unsigned char* flips, int nframes, int H, int W, float* out, bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && H > 0 && W > 0 && out, "bmc_encode_raw_events: bad shape");
    hipStream_t st = (hipStream_t)s;
    if (nframes == 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nframes * 2 * H * W * sizeof(float), st);
    if (e != hipSuccess) { bmc_set_error("bmc_encode_raw_events: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(encode_raw_kernel, dim3(64, nframes), dim3(256), 0, st, xs, ys, ps, offsets, flips, H, W, out);
    BMC_CHECK_LAUNCH("bmc_encode_raw_events");
    return 0;
}

extern "C" long long bmc_events_binned_ws_bytes(long long nevents, int nframes, int H, int W) {
    const int R = BIN_LDS / 2 / W < H ? BIN_LDS / 2 / W : H;
    if (R < 1 || nframes < 0 || nevents < 0) return -1;
    const long long ntab = (long long)nframes * ((H + R - 1) / R);
    return ((2 * ntab + 1) * 4 + 7) / 8 * 8 + nevents * 8;
}

extern "C" int bmc_events_to_channels_binned(float* xs, float* ys, const float* ps, const long long* offsets, long long nevents,
                                             int nframes, int H, int W, float* out, int mutate, void* ws, long long ws_bytes,
                                             bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && H > 0 && W > 0 && out, "bmc_events_to_channels_binned: bad shape");
    if (nframes == 0) return 0;
    F32Events ev{xs, ys, ps, mutate};
    const int rc = launch_binned(ev, offsets, nevents, nframes, H, W, out, ws, ws_bytes, (hipStream_t)s, "bmc_events_to_channels_binned");
    if (rc) return rc;
    BMC_CHECK_LAUNCH("bmc_events_to_channels_binned");
    return 0;
}

extern "C" int bmc_encode_raw_events_binned(const short* xs, const short* ys, const double* ps, const long long* offsets,
                                            const unsigned char* flips, long long nevents, int nframes, int H, int W, float* out,
                                            void* ws, long long ws_bytes, bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && H > 0 && W > 0 && out, "bmc_encode_raw_events_binned: bad shape");
    if (nframes == 0) return 0;
    RawEvents ev{xs, ys, ps, flips};
    const int rc = launch_binned(ev, offsets, nevents, nframes, H, W, out, ws, ws_bytes, (hipStream_t)s, "bmc_encode_raw_events_binned");
    if (rc) return rc;
    BMC_CHECK_LAUNCH("bmc_encode_raw_events_binned");
    return 0;
}

extern "C" int bmc_events_to_channels(float* xs, float* ys, const float* ps, const long long* offsets, int nframes, int H,
                                      int W, float* out, int mutate, bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && H > 0 && W > 0 && out, "bmc_events_to_channels: bad shape");
    hipStream_t st = (hipStream_t)s;
    if (nframes == 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nframes * 2 * H * W * sizeof(float), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_channels: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(events_kernel, dim3(64, nframes), dim3(256), 0, st, xs, ys, ps, offsets, H, W, out, mutate);
    BMC_CHECK_LAUNCH("bmc_events_to_channels");
    return 0;
}

extern "C" int bmc_events_to_stack(float* xs, float* ys, const float* ts, float* ps, long long n, const float* tstart,
                                   const float* tend, int bins, int H, int W, float* out, int* ranges, int mutate,
                                   bmc_stream_t s) {
    BMC_CHECK_ARG(bins >= 1 && H > 0 && W > 0 && out && ranges && tstart && tend, "bmc_events_to_stack: bad arguments");
    BMC_CHECK_ARG(n >= 0 && n < (1ll << 31), "bmc_events_to_stack: event count out of range");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)bins * H * W * sizeof(float), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_stack: memset failed: %s", hipGetErrorString(e)); return -2; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(stack_search_kernel, dim3((2 * bins + 63) / 64), dim3(64), 0, st, ts, n, tstart, tend, bins, ranges);
    hipLaunchKernelGGL(stack_scatter_kernel, dim3(64, bins), dim3(256), 0, st, xs, ys, ps, ranges, H, W, out);
    if (mutate) hipLaunchKernelGGL(stack_mutate_kernel, dim3(64, bins), dim3(256), 0, st, xs, ys, ps, ranges, H, W);
    BMC_CHECK_LAUNCH("bmc_events_to_stack");
    return 0;
}

extern "C" int bmc_events_to_stack_polarity(float* xs, float* ys, const float* ts, const float* ps, long long n,
                                            const float* tstart, const float* tend, int bins, int H, int W, float* out,
                                            int* ranges, int mutate, bmc_stream_t s) {
    BMC_CHECK_ARG(bins >= 1 && H > 0 && W > 0 && out && ranges && tstart && tend, "bmc_events_to_stack_polarity: bad arguments");
    BMC_CHECK_ARG(n >= 0 && n < (1ll << 31), "bmc_events_to_stack_polarity: event count out of range");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)2 * bins * H * W * sizeof(float), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_stack_polarity: memset failed: %s", hipGetErrorString(e)); return -2; }
    if (n == 0) return 0;
    hipLaunchKernelGGL(stack_search_kernel, dim3((2 * bins + 63) / 64), dim3(64), 0, st, ts, n, tstart, tend, bins, ranges);
    hipLaunchKernelGGL(stack_pol_scatter_kernel, dim3(64, bins), dim3(256), 0, st, xs, ys, ps, ranges, bins, H, W, out);
    if (mutate) hipLaunchKernelGGL(stack_pol_mutate_kernel, dim3(64, bins), dim3(256), 0, st, xs, ys, ranges, H, W);
    BMC_CHECK_LAUNCH("bmc_events_to_stack_polarity");
    return 0;
}

extern "C" int bmc_events_to_mask(float* xs, float* ys, float* ps, long long n, int H, int W, float* out, int* ws, int mutate,
                                  bmc_stream_t s) {
    BMC_CHECK_ARG(H > 0 && W > 0 && out && ws, "bmc_events_to_mask: bad arguments");
    BMC_CHECK_ARG(n >= 0 && n < (1ll << 31) - 1, "bmc_events_to_mask: event count out of range");
    hipStream_t st = (hipStream_t)s;
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)H * W * sizeof(int), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_mask: memset failed: %s", hipGetErrorString(e)); return -2; }
    if (n > 0) hipLaunchKernelGGL(mask_owner_kernel, dim3(256), dim3(256), 0, st, xs, ys, n, H, W, ws);
    hipLaunchKernelGGL(mask_write_kernel, dim3(256), dim3(256), 0, st, xs, ys, ps, n, H, W, ws, out, mutate);
    if (mutate && n > 0) hipLaunchKernelGGL(mask_mutate_kernel, dim3(256), dim3(256), 0, st, xs, ys, ps, n, H, W);
    BMC_CHECK_LAUNCH("bmc_events_to_mask");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The torch-tensor encodings of the reference (dataloader/encodings.py:16-73 events_to_image_torch with bilinear
// interpolation, :100-148 events_to_voxel_torch).  Nothing in the reference calls them; they complete the encodings row.
// Arbitrary float weights -> the summation order is part of the result, and it is the order of the reference's CPU
// index_put_(accumulate=True): one pass per bilinear corner ((0,0), (0,1), (1,0), (1,1)), the events in order within a
// pass.  Same ordered machinery as the voxel grid above: events are binned by their CELL (floor pixel) with integer
// atomics, every cell's segment is put in event order, and every OUTPUT pixel then gathers -- pass by pass -- from the up
// to four cells that reach it.  Deterministic, bit-identical to the single-threaded reference; float arithmetic is written
// with explicit round-to-nearest intrinsics in the reference's operation order (no fused multiply-add).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ void one_frame_offsets_kernel(long long* off, long long n) { off[0] = 0; off[1] = n; }

// events_to_image_torch :33-38 (reset out-of-range events; MUTATES xs, ys, ps) and :48-64: cell of every event.
// mode bit 0: bilinear, bit 1: padding, bit 2: clip_out_of_range.  cell[e] = iy * iw + ix; wts[e] = ps * mask (bilinear).
__global__ void img_cells_kernel(float* __restrict__ xs, float* __restrict__ ys, float* __restrict__ ps, long long n, int H, int W,
                                 int ih, int iw, int mode, int* __restrict__ cell, int* __restrict__ cnt,
                                 unsigned char* __restrict__ bad_out, int mutate_ps, const int* __restrict__ live) {
#pragma clang fp contract(off)
    const bool mutate = live == nullptr || *live != 0;      // (events_to_voxel_torch returns before touching anything when ts is all zero)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        float x = xs[e], y = ys[e];
        const bool bad = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (bad) {
            x = 0.f; y = 0.f;
            if (mutate) {
                xs[e] = 0.f; ys[e] = 0.f;
                if (mutate_ps) ps[e] = 0.f;
            }
        }
        if (bad_out) bad_out[e] = bad ? 1 : 0;
        int ix, iy;
        if (mode & 1) {
            float m = 1.f;
            if (mode & 4) {
                const float clipx = (float)(iw - 1), clipy = (float)(ih - 1);
                m = (x >= clipx ? 0.f : 1.f) * (y >= clipy ? 0.f : 1.f);
            }
            ix = (int)(floorf(x) * m);
            iy = (int)(floorf(y) * m);
        } else {
            ix = (int)x; iy = (int)y;            // .long(): truncation; coordinates are >= 0 here
        }
        const int q = iy * iw + ix;
        cell[e] = q;
        atomicAdd(cnt + q, 1);
    }
}
__global__ void cell_fill_kernel(const int* __restrict__ cell, long long n, int* __restrict__ cur, int* __restrict__ idx) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        idx[atomicAdd(cur + cell[e], 1)] = (int)e;
}
// every cell's segment into event order (insertion sort: segments are short, the long ones were sorted cooperatively before)
__global__ void cell_sort_kernel(const int* __restrict__ seg, int ncell, int* __restrict__ idx) {
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ncell; q += gridDim.x * blockDim.x) {
        const int a = seg[q], b = seg[q + 1];
        for (int i = a + 1; i < b; ++i) {
            const int v = idx[i];
            int j = i - 1;
            while (j >= a && idx[j] > v) { idx[j + 1] = idx[j]; --j; }
            idx[j + 1] = v;
        }
    }
}
__global__ void img_gather_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ ps, int H, int W,
                                  int ih, int iw, int mode, const int* __restrict__ seg, const int* __restrict__ idx,
                                  float* __restrict__ out) {
#pragma clang fp contract(off)      // every product is rounded before it is added, as the reference's tensor expressions are (the
                                    // *_rn intrinsics are plain operators here: the compiler would fuse w * fx * fy + acc)
    const int npx = ih * iw;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < npx; o += gridDim.x * blockDim.x) {
        const int oy = o / iw, ox = o - oy * iw;
        float acc = 0.f;
        if (!(mode & 1)) {                       // interpolation=None: img[ys, xs] += ps, events in order
            for (int i = seg[o]; i < seg[o + 1]; ++i) acc = acc + ps[idx[i]];
            out[o] = acc;
            continue;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {            // interpolate_to_image (:6-13): pass c adds corner (c >> 1, c & 1) of every event
            const int cy = oy - (c >> 1), cx = ox - (c & 1);
            if (cy < 0 || cx < 0) continue;
            const int q = cy * iw + cx;
            for (int i = seg[q]; i < seg[q + 1]; ++i) {
                const int e = idx[i];
                const float x = xs[e], y = ys[e];
                float m = 1.f;
                if (mode & 4) m = (x >= (float)(iw - 1) ? 0.f : 1.f) * (y >= (float)(ih - 1) ? 0.f : 1.f);
                const float dx = x - floorf(x), dy = y - floorf(y);
                const float w = ps[e] * m;
                const float fx = (c & 1) ? dx : 1.0f - dx, fy = (c >> 1) ? dy : 1.0f - dy;
                const float t = (w * fx) * fy;                               // weights * (1 - dxs | dxs) * (1 - dys | dys)
                acc = acc + t;
            }
        }
        out[o] = acc;
    }
}
// events_to_voxel_torch :121-139: flag[0] = any timestamp != 0 (timestamps are >= 0: "ts.sum() == 0" <=> all zero)
__global__ void any_nonzero_kernel(const float* __restrict__ ts, long long n, int* __restrict__ flag) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        if (ts[e] != 0.f) { *flag = 1; break; }
}
__global__ void voxel_torch_gather_kernel(const float* __restrict__ ts, const float* __restrict__ ps, long long n, int bins, int H, int W,
                                          const int* __restrict__ seg, const int* __restrict__ idx,
                                          const unsigned char* __restrict__ bad, const int* __restrict__ flag, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int HW = H * W;
    const float t0 = ts[0];
    const float dt = (ts[n - 1] - t0) + 1e-6f;
    const float bm1 = (float)(bins - 1);
    const bool live = *flag != 0;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < HW; o += gridDim.x * blockDim.x) {
        for (int bi = 0; bi < bins; ++bi) {
            float acc = 0.f;
            if (live)
                for (int i = seg[o]; i < seg[o + 1]; ++i) {
                    const int e = idx[i];
                    const float tn = ((ts[e] - t0) / dt) * bm1;
                    float wgt = ps[e] * fmaxf(0.f, 1.0f - fabsf(tn - (float)bi));
                    if (bad[e] && bi == 0) wgt = 0.f;        // the first bin's call zeroes the weights of out-of-range events (:38)
                    acc = acc + wgt;
                }
            out[(long long)bi * HW + o] = acc;
        }
    }
}

int ordered_cells(float* xs, float* ys, float* ps, long long n, int H, int W, int ih, int iw, int mode, int* ws, unsigned char* bad,
                  int mutate_ps, const int* live, hipStream_t st, int** seg_out, int** idx_out) {
    const int ncell = ih * iw;
    // ws: seg[ncell + 1] | cur[ncell + 1] | idx[n] | cell[n] | offsets (2 x int64, 8-byte aligned)
    int* const seg = ws;
    int* const cur = ws + (ncell + 1);
    int* const idx = ws + 2 * (ncell + 1);
    int* const cell = idx + n;
    long long* const off = reinterpret_cast<long long*>(ws + ((2ll * (ncell + 1) + 2 * n + 1) / 2) * 2);
    hipError_t e = hipMemsetAsync(seg, 0, (size_t)(ncell + 1) * sizeof(int), st);
    if (e != hipSuccess) { bmc_set_error("events_to_image_torch: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(one_frame_offsets_kernel, dim3(1), dim3(1), 0, st, off, n);
    hipLaunchKernelGGL(img_cells_kernel, dim3(256), dim3(256), 0, st, xs, ys, ps, n, H, W, ih, iw, mode, cell, seg, bad, mutate_ps, live);
    hipLaunchKernelGGL(voxel_scan_kernel, dim3(1), dim3(1024), 0, st, seg, cur, ncell);
    hipLaunchKernelGGL(cell_fill_kernel, dim3(256), dim3(256), 0, st, cell, n, cur, idx);
    hipLaunchKernelGGL(voxel_sort_long_kernel, dim3(1), dim3(1024), 0, st, off, ncell, seg, idx);
    hipLaunchKernelGGL(cell_sort_kernel, dim3((ncell + 255) / 256), dim3(256), 0, st, seg, ncell, idx);
    *seg_out = seg; *idx_out = idx;
    return 0;
}

}  // namespace

extern "C" long long bmc_events_torch_ws_ints(long long n, int H, int W) {
    return 2ll * ((long long)(H + 1) * (W + 1) + 1) + 2 * n + 8 + (n + 3) / 4 + 4;
}

extern "C" int bmc_events_to_image_torch(float* xs, float* ys, float* ps, long long n, int H, int W, int clip_out_of_range,
                                         int bilinear, int padding, float* out, int* ws, bmc_stream_t s) {
    BMC_CHECK_ARG(H > 0 && W > 0 && out && ws && n >= 0 && n < (1ll << 30), "bmc_events_to_image_torch: bad arguments");
    BMC_CHECK_ARG(!(bilinear && !padding && !clip_out_of_range), "bmc_events_to_image_torch: bilinear interpolation without padding needs "
                  "clip_out_of_range (the reference itself indexes out of bounds for an event in the last row / column)");
    hipStream_t st = (hipStream_t)s;
    const int ih = (bilinear && padding) ? H + 1 : H, iw = (bilinear && padding) ? W + 1 : W;
    const int mode = (bilinear ? 1 : 0) | (padding ? 2 : 0) | (clip_out_of_range ? 4 : 0);
    int *seg, *idx;
    if (ordered_cells(xs, ys, ps, n, H, W, ih, iw, mode, ws, nullptr, 1, nullptr, st, &seg, &idx)) return -2;
    hipLaunchKernelGGL(img_gather_kernel, dim3((ih * iw + 255) / 256), dim3(256), 0, st, xs, ys, ps, H, W, ih, iw, mode, seg, idx, out);
    BMC_CHECK_LAUNCH("bmc_events_to_image_torch");
    return 0;
}

extern "C" int bmc_events_to_voxel_torch(float* xs, float* ys, const float* ts, const float* ps, long long n, int bins, int H, int W,
                                         float* out, int* ws, bmc_stream_t s) {
    BMC_CHECK_ARG(bins >= 1 && H > 0 && W > 0 && out && ws && n >= 0 && n < (1ll << 30), "bmc_events_to_voxel_torch: bad arguments");
    hipStream_t st = (hipStream_t)s;
    if (n <= 3) {                                    // :121-122
        hipError_t e = hipMemsetAsync(out, 0, (size_t)bins * H * W * sizeof(float), st);
        if (e != hipSuccess) { bmc_set_error("bmc_events_to_voxel_torch: memset failed: %s", hipGetErrorString(e)); return -2; }
        return 0;
    }
    const long long base = 2ll * ((long long)H * W + 1) + 2 * n + 8;
    unsigned char* const bad = reinterpret_cast<unsigned char*>(ws + base);
    int* const flag = ws + base + (n + 3) / 4;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_voxel_torch: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(any_nonzero_kernel, dim3(64), dim3(256), 0, st, ts, n, flag);
    int *seg, *idx;
    // (coordinates are reset in place by the first bin's events_to_image_torch call; its weights are a temporary: ps stays)
    if (ordered_cells(xs, ys, const_cast<float*>(ps), n, H, W, H, W, 0, ws, bad, 0, flag, st, &seg, &idx)) return -2;
    hipLaunchKernelGGL(voxel_torch_gather_kernel, dim3((H * W + 255) / 256), dim3(256), 0, st, ts, ps, n, bins, H, W, seg, idx, bad, flag,
                       out);
    BMC_CHECK_LAUNCH("bmc_events_to_voxel_torch");
    return 0;
}
