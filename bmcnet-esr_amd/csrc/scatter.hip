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

// Temporal-bilinear voxel grid (dataloader/encodings.py:272-287): bin b receives p * max(0, 1 - |t*(bins-1) - b|)
// through events_to_image(), i.e. with the vertical flip and with the same side effect as above: the FIRST bin's call
// zeroes out-of-range events and resets their coordinates in place, so in every later bin they are no longer masked
// and deposit their weight at [H-1, 0].  Weights are arbitrary floats: the sum is accumulated with float atomics, so
// (like the reference's multi-threaded index_put_) it is defined up to summation order.
__global__ void voxel_kernel(float* __restrict__ xs, float* __restrict__ ys, const float* __restrict__ ts,
                             const float* __restrict__ ps, const long long* __restrict__ offsets, int bins, int H, int W,
                             float* __restrict__ out, int mutate) {
    const int f = blockIdx.y;
    const long long e0 = offsets[f], e1 = offsets[f + 1];
    float* const vox = out + (long long)f * bins * H * W;
    for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += (long long)gridDim.x * blockDim.x) {
        float x = xs[e], y = ys[e];
        const float p = ps[e], t = ts[e] * (float)(bins - 1);
        const bool oob = (x >= (float)W) | (x < 0.f) | (y >= (float)H) | (y < 0.f);
        if (oob) {
            x = 0.f; y = 0.f;
            if (mutate) { xs[e] = 0.f; ys[e] = 0.f; }
        }
        const long long pix = (long long)(H - (int)y - 1) * W + (int)x;
        for (int b = 0; b < bins; ++b) {
            const float wgt = p * fmaxf(0.f, 1.0f - fabsf(t - (float)b));
            if (wgt != 0.f && !(oob && b == 0)) atomicAdd(vox + (long long)b * H * W + pix, wgt);
        }
    }
}

}  // namespace

extern "C" int bmc_events_to_voxel(float* xs, float* ys, const float* ts, const float* ps, const long long* offsets,
                                   int nframes, int bins, int H, int W, float* out, int mutate, bmc_stream_t s) {
    BMC_CHECK_ARG(nframes >= 0 && bins >= 1 && H > 0 && W > 0 && out, "bmc_events_to_voxel: bad shape");
    hipStream_t st = (hipStream_t)s;
    if (nframes == 0) return 0;
    hipError_t e = hipMemsetAsync(out, 0, (size_t)nframes * bins * H * W * sizeof(float), st);
    if (e != hipSuccess) { bmc_set_error("bmc_events_to_voxel: memset failed: %s", hipGetErrorString(e)); return -2; }
    hipLaunchKernelGGL(voxel_kernel, dim3(64, nframes), dim3(256), 0, st, xs, ys, ts, ps, offsets, bins, H, W, out, mutate);
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
