/*
 * bmc_hip.h -- C ABI of libbmc_hip.so, the MI355X (gfx950) kernel library behind
 * the BMCNet bilateral event-SR hot path.
 *
 * The reference (Lqm26/BMCNet-ESR) has no FFI: its hot path is a chain of ATen
 * calls issued from Python.  Each entry point below replaces the
 * ATen call sites cited next to it; the Python host side
 * (bmcnet-esr_amd/bmc_hip) binds them with ctypes and wraps fwd/bwd pairs in
 * torch.autograd.Function objects behind the reference's own nn.Module
 * signatures.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory unless a
 *    parameter is documented as a host struct;
 *  - activations are fp32 NHWC: element (b,y,x,c) of a tensor lives at
 *    ptr[b*batch_stride + (y*W + x)*pix_stride + c];
 *  - nothing allocates: workspaces are passed in;
 *  - every launch goes to the hipStream_t given (pass torch's current stream);
 *  - return 0 on success, <0 on error (bmc_last_error() has the text).
 */
#ifndef BMC_HIP_H
#define BMC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bmc_stream_t; /* hipStream_t */

#define BMC_MAX_SRC 6
#define BMC_CK 16 /* channel granule: every source's nch is a multiple of 16 */

/* A channel-slice of an NHWC tensor used as one operand of a multi-source
 * convolution (what the reference builds with torch.cat, models/BMCNet.py:60-73,
 * models/submodules.py:63-64,75).  Batch index b of the launch reads batch
 * (b + batch_shift) % batch_mod of this tensor (shared / swapped operands of the
 * weight-shared twin branches). */
typedef struct bmc_src {
    const float* ptr;
    long long batch_stride; /* floats */
    int pix_stride;         /* floats */
    int nch;                /* channels consumed from ptr, multiple of 16 */
    int batch_shift;
    int batch_mod;          /* >= 1, or BMC_SRC_TABLE: `ptr` is a device table of per-image base pointers (bmc_ptr_table) */
} bmc_src_t;
/* batch_mod == BMC_SRC_TABLE: image b of the launch starts at ((const float* const*)ptr)[b] -- the images of one operand gathered
 * from several tensors (the uses of ONE weight by the weight-sharing blocks of a window reduced by one weight-gradient launch,
 * models/BMCNet.py:19-32); pix_stride / nch as usual, batch_stride / batch_shift unused. */
#define BMC_SRC_TABLE (-1)


/* ---- library ---- */
int bmc_version(void);
const char* bmc_last_error(void);
/* A new stream of the device's lowest priority (hipDeviceGetStreamPriorityRange), never destroyed by the library.  The training
 * step issues its weight-gradient kernels there (the reference has no counterpart: autograd runs train.py:233's backward on
 * one stream): they fill the CUs the data-gradient chain leaves idle without delaying it. */
int bmc_stream_create_low_priority(bmc_stream_t* out);

/* ---- event -> count image: dataloader/encodings.py:241-269,290-305 -------
 * events_to_channels() for `nframes` frames in one launch.  Frame f owns events
 * [offsets[f], offsets[f+1]) of xs/ys/ps (fp32, as event_formatting() leaves
 * them, dataloader/base_dataset.py:24-31) and writes out[f] = [2,H,W] fp32
 * (zero-filled here).  Bit-exact with the reference including its quirk: an
 * out-of-range event is dropped from channel 0, but its x,y are reset to 0 in
 * place so a NEGATIVE one lands on [H-1,0] of channel 1.  If mutate != 0 xs/ys
 * are updated in place as the reference does to its caller's tensors. */
int bmc_events_to_channels(float* xs, float* ys, const float* ps, const long long* offsets,
                           int nframes, int H, int W, float* out, int mutate, bmc_stream_t s);

/* The same two encoders for LARGE frames (the 720x960 ground-truth frames of the training step) without scattered global
 * float atomics: events are counting-sorted by row band (integer atomics on a small table), one workgroup per
 * (frame, band) accumulates its band of the count image in LDS and stores it once (that store is also the zero fill).
 * Same results bit for bit (integer-valued sums), same in-place side effect on xs / ys with mutate.  nevents =
 * offsets[nframes]; ws: 8-byte aligned workspace of bmc_events_binned_ws_bytes(nevents, nframes, H, W) bytes
 * ((frame, band) count table and cursors + 8 bytes per event).  W <= 16384. */
long long bmc_events_binned_ws_bytes(long long nevents, int nframes, int H, int W);
int bmc_events_to_channels_binned(float* xs, float* ys, const float* ps, const long long* offsets, long long nevents,
                                  int nframes, int H, int W, float* out, int mutate, void* ws, long long ws_bytes,
                                  bmc_stream_t s);
int bmc_encode_raw_events_binned(const short* xs, const short* ys, const double* ps, const long long* offsets,
                                 const unsigned char* flips, long long nevents, int nframes, int H, int W, float* out,
                                 void* ws, long long ws_bytes, bmc_stream_t s);

/* The torch-tensor encodings of the reference (no caller in the reference; they complete the encodings row):
 * events_to_image_torch (dataloader/encodings.py:16-73) on one event list -> out [H][W], or [H+1][W+1] for bilinear
 * interpolation with padding.  bilinear != 0: interpolate_to_image (:6-13), sub-pixel positions spread over the four
 * neighbours; bilinear == 0: img[ys.long(), xs.long()] += ps.  Out-of-range events are reset IN PLACE to (0, 0) with weight 0
 * (:33-38) -- xs, ys, ps are mutated like the reference's tensors.  The sums are taken in the order of the reference's CPU
 * index_put_(accumulate=True) (one pass per corner, events in order): deterministic and bit-identical to it.
 * ws: bmc_events_torch_ws_ints(n, H, W) ints.  bilinear without padding requires clip_out_of_range.
 * events_to_voxel_torch (:100-148), temporal_bilinear=True: out [bins][H][W]; ts sorted ascending and >= 0; zeros for
 * n <= 3 or all-zero timestamps (:121-122); xs, ys are reset in place by the first bin's call (ps is not touched). */
long long bmc_events_torch_ws_ints(long long n, int H, int W);
int bmc_events_to_image_torch(float* xs, float* ys, float* ps, long long n, int H, int W, int clip_out_of_range, int bilinear,
                              int padding, float* out, int* ws, bmc_stream_t s);
int bmc_events_to_voxel_torch(float* xs, float* ys, const float* ts, const float* ps, long long n, int bins, int H, int W,
                              float* out, int* ws, bmc_stream_t s);

/* events_to_voxel(): temporal-bilinear voxel grid [nframes][bins][H][W] (dataloader/encodings.py:272-287; ts already
 * normalised to [0,1] by event_formatting).  Same coordinate conventions and first-call side effect as above.  The
 * weights are arbitrary floats, so the order of summation is part of the result: a pixel's events are added in EVENT
 * ORDER, as the reference's sequential index_put_ does -- deterministic (no float atomics) and bit-identical to the
 * single-threaded reference.  nevents = offsets[nframes]; ws: workspace of 2*nframes*(H*W + 1) + nevents ints. */
int bmc_events_to_voxel(float* xs, float* ys, const float* ts, const float* ps, const long long* offsets,
                        long long nevents, int nframes, int bins, int H, int W, float* out, int mutate, int* ws,
                        bmc_stream_t s);

/* events_to_stack_no_polarity() (dataloader/encodings.py:202-238): `bins` temporal bins over one event window, each the
 * signed per-pixel sum of the polarities of its events at [(long) y, (long) x] (no vertical flip).  tstart / tend [bins]:
 * the float32 bin bounds ts[0] + delta_t*bi and tstart + delta_t, computed by the caller with the reference's float32
 * expressions; the bins' event ranges come from the reference's own binary search (:75-97, quirks included) run on
 * the device, into `ranges` [2*bins] ints.  If mutate != 0, out-of-range events covered by a bin get xs = ys = ps = 0 in
 * place, as the reference does to its caller's tensors.  Exact for +-1 polarities (integer-valued sums). */
int bmc_events_to_stack(float* xs, float* ys, const float* ts, float* ps, long long n, const float* tstart,
                        const float* tend, int bins, int H, int W, float* out, int* ranges, int mutate, bmc_stream_t s);

/* events_to_stack_polarity() (dataloader/encodings.py:151-199): as bmc_events_to_stack but two COUNT images per bin,
 * out [2][bins][H][W] = (positives, negatives), weights p*p.  The reference's first (positive) call of the first bin that
 * covers an event resets its out-of-range coordinates (in place, if mutate) and masks only that call: an out-of-range
 * negative event counts at [0,0] of the negative image, and an out-of-range event that a later, overlapping bin covers
 * again counts at [0,0] whatever its sign.  ps is never modified. */
int bmc_events_to_stack_polarity(float* xs, float* ys, const float* ts, const float* ps, long long n, const float* tstart,
                                 const float* tend, int bins, int H, int W, float* out, int* ranges, int mutate,
                                 bmc_stream_t s);

/* events_to_mask() (dataloader/encodings.py:308-332; the hot-pixel filter's input, dataloader/h5dataset.py:528-546):
 * out [H][W], out[(long) y][(long) x] = |p| of the LAST event (in order) that maps there -- index_put_(accumulate=False)
 * semantics, made deterministic with an integer atomicMax over event indices (ws: H*W ints).  Out-of-range events
 * count as (0, 0) with p = 0; if mutate != 0 their xs / ys / ps are zeroed in place as the reference does. */
int bmc_events_to_mask(float* xs, float* ys, float* ps, long long n, int H, int W, float* out, int* ws, int mutate,
                       bmc_stream_t s);

/* Sequence encoder on raw dataset columns: what H5Dataset.__getitem__ does per frame on the CPU
 * (dataloader/h5dataset.py:261-316: get_events :407-414 -> augment_event :559-578 -> event_formatting
 * base_dataset.py:24-31 -> events_to_channels), for all frames of a batch in one launch.
 * xs/ys int16, ps float64 (generate_dataset/tools/event_packagers.py:128-156); flips[f] bit0 horizontal
 * (x = W-1-x), bit1 vertical (y = H-1-y), bit2 polarity (p = -p); flips NULL = no augmentation. */
int bmc_encode_raw_events(const short* xs, const short* ys, const double* ps, const long long* offsets,
                          const unsigned char* flips, int nframes, int H, int W, float* out, bmc_stream_t s);

/* ---- weight packing ------------------------------------------------------
 * Conv weights [G][Cout][Cin][taps] (taps = kh*kw = 1 or 9; nn.Conv2d layout)
 * -> MFMA staging layout [G][Kpad/16][taps][Coutpad][16] where packed input
 * channel k holds reference input channel kmap[k] (or zero when kmap[k] < 0).
 * transpose != 0 builds the data-gradient operator instead: output channels =
 * packed k (Kpad rounded up to Coutpad_t), reduction over Cout, taps mirrored. */
int bmc_pack_weight(const float* w, const int* kmap, int G, int Cout, int Cin, int taps,
                    int Kpad, int Coutpad, float* out, bmc_stream_t s);
int bmc_pack_weight_t(const float* w, const int* kmap, int G, int Cout, int Cin, int taps,
                      int k0, int nk, int nkpad, int Coutpad16, float* out, bmc_stream_t s);
/* Packed fp32 weights [nsteps][Coutpad][16] (nsteps = G * Kpad/16 * taps) -> `planes` bf16 planes
 * [nsteps][planes][Coutpad][16] for bmc_conv with math = BMC_MATH_BF16 (planes 1) / BMC_MATH_BF16X6 (planes 3);
 * `out` holds nsteps*planes*Coutpad*16 bf16 values (2 bytes each); inside a 32-byte row the two 16-byte halves are
 * swapped when (row & 16), the kernel's conflict-free LDS image (the weight stream is a linear LDS-DMA copy of it). */
#define BMC_MATH_FP32 0
#define BMC_MATH_BF16 1
#define BMC_MATH_BF16X6 3
#define BMC_MATH_FP32_WINO 4 /* bmc_conv only: fp32 MFMA through the Winograd transform F(2x2, 3x3), below */
#define BMC_MATH_FP32_WINO4 5 /* bmc_conv only: the same through F(4x4, 3x3) (bmc_pack_weight_wino4) */
int bmc_split_weight(const float* packed, void* out, long long nsteps, int Coutpad, int planes, bmc_stream_t s);

/* Conv weights [G][Cout][Cin][3][3] -> the TRANSFORMED weights U = G g G^T of Winograd's F(2x2, 3x3) minimal filtering
 * (16 values per (co, ci) pair instead of 9), in the streaming order of bmc_conv with math = BMC_MATH_FP32_WINO:
 * [G][Kpad/16][4 (xi)][4 (nu)][Coutpad][16], quads of a 16-float row XOR-swizzled with (row >> 2) & 3.
 *   transposed == 0: rows = output channels (Coutpad: multiple of 128), K = packed input channels through kmap (as
 *                    bmc_pack_weight);  w_group_stride of the launch = Kpad * Coutpad * 16 floats;
 *   transposed != 0: the data-gradient operator w.r.t. packed source channels [k0, k0 + nk): rows = those channels (padded
 *                    to Coutpad), K = the Cout output channels (padded to Kpad), taps mirrored (as bmc_pack_weight_t). */
int bmc_pack_weight_wino(const float* w, const int* kmap, int G, int Cout, int Cin, int Kpad, int Coutpad, int transposed,
                         int k0, int nk, float* out, bmc_stream_t s);

/* Rows per workgroup tile (8 or 4) that bmc_conv with math = BMC_MATH_FP32_WINO uses for a launch of B images of H x W pixels and
 * Coutpad output channels on a device of `cus` compute units (<= 0: 256): 4 where the 8 x 16-pixel tiling leaves CUs without a
 * tile and the 4 x 16 one does not (small frames).  The host's routing rule (bmc_hip/ops.py::wino_ok) counts tiles with it. */
int bmc_conv_wino_rows(int B, int H, int W, int Coutpad, int cus);

/* The same for F(4x4, 3x3) (36 values per (co, ci) pair, made in double and rounded once; points 0, +-1, +-2, inf), in the
 * streaming order of bmc_conv with math = BMC_MATH_FP32_WINO4: [G][Coutpad/128][Kpad/16][8 (wave)][36 (position)][64 (lane)][4],
 * lane l of wave w = row 128 ntile + 16 w + (l & 15), channels 16 chunk + 4 (l >> 4) + 0..3 -- the MFMA A-operand image a wave
 * loads with one 1 KB instruction per position.  Arguments as bmc_pack_weight_wino; w_group_stride of the launch =
 * Kpad * Coutpad * 36 floats.  Replaces the weight operand of the 3x3 F.conv2d calls at models/submodules.py:31-35 (the
 * residual blocks: 100 of the 103 3x3 convolutions of a window, models/BMCNet.py:19-32) and models/BMCNet.py:64-82. */
int bmc_pack_weight_wino4(const float* w, const int* kmap, int G, int Cout, int Cin, int Kpad, int Coutpad, int transposed,
                          int k0, int nk, float* out, bmc_stream_t s);

/* ---- implicit-GEMM convolution (fp32 MFMA) -------------------------------
 * Replaces F.conv2d at models/submodules.py:25-26,33-34,44-53,63-67,75 and
 * models/BMCNet.py:40-53,64-82 together with the torch.cat / relu / residual
 * add around them, and torch.bmm(softmax, v) at models/submodules.py:72-73
 * (a 1x1 convolution with per-sample weights).  Also the data gradient of all
 * of these (same kernel, transposed weights).
 *   out[b,y,x,co] = epi( sum_{tap,k} W[g][k][tap][co] * cat(src)[b, y+dy, x+dx, k] + bias[g][co] )
 *   epi(v) = relu? max(v,0) : v, after adding residual[b,y,x,co] if given;
 *   g = b / batch_per_group. */
typedef struct bmc_conv_args {
    int nsrc;
    bmc_src_t src[BMC_MAX_SRC];
    const void* wpacked;        /* from bmc_pack_weight (math 0) or bmc_split_weight of it (math 1, 3) */
    const float* bias;          /* [G][Cout] or NULL */
    long long w_group_stride;   /* 32-bit words between groups in wpacked (math 0: floats; math 1 / 3: floats * planes / 2) */
    int bias_group_stride;
    int batch_per_group;        /* >= 1 */
    float* out;
    long long out_batch_stride;
    int out_pix_stride;
    int B, H, W;
    int Cout, Coutpad;          /* Coutpad: multiple of 32 (of 128 when > 32) */
    int taps;                   /* 1 or 9 */
    int relu;
    bmc_src_t residual;         /* ptr NULL -> none; nch ignored */
    bmc_src_t mask;             /* ptr NULL -> none; out = mask > 0 ? v : 0 (ReLU backward) */
    int accumulate;             /* out += v */
    int math;                   /* BMC_MATH_FP32 (0): v_mfma_f32_32x32x2_f32 on the fp32 operands;
                                   BMC_MATH_BF16 (1): operands rounded to bf16, fp32 accumulate (v_mfma_f32_32x32x16_bf16);
                                   BMC_MATH_BF16X6 (3): each fp32 operand split exactly into three bf16 planes, six plane
                                   products with fp32 accumulate -- fp32-equivalent (dropped terms ~ one fp32 rounding per product);
                                   BMC_MATH_FP32_WINO (4): taps = 9, Coutpad % 128 == 0, wpacked from bmc_pack_weight_wino: fp32 MFMA
                                   on Winograd-transformed operands, F(2x2, 3x3): 16 instead of 36 multiplies per 2x2 output
                                   tile and channel pair, fp32 throughout (error vs float64 ~1.5x the direct fp32 kernel's) */
} bmc_conv_args_t;
int bmc_conv(const bmc_conv_args_t* host_args, bmc_stream_t s);

/* ---- pixel-reduction GEMM: weight gradients and channel Gram matrices ----
 *   C[g][tap][m][n] = sum_{b in group g} sum_{y,x} A[b,y,x,m] * cat(src)[b,y+dy,x+dx,n]
 * Replaces the weight-gradient half of conv backward (ATen autograd) and
 * torch.bmm(center, v) at models/submodules.py:69-70 (+ its backward).
 * Split over pixels: partial sums go to `slabs`
 * [nsplit][G][taps][Mpad][Npad] (Mpad/Npad = M/N rounded up to 32), which one of
 * the reduce calls below then sums (deterministic order). */
typedef struct bmc_pgemm_args {
    bmc_src_t a;                /* nch = M */
    int nsrc;
    bmc_src_t src[BMC_MAX_SRC]; /* sum nch = N */
    int B, H, W, taps;
    int batch_per_group;
    float* slabs;
    int nsplit;
    const float* zeros;         /* >= 64 bytes of zeros in device memory (source for out-of-image LDS-DMA lanes) */
    float* bias_slabs;          /* optional [nsplit][G][4][Mpad]: column sums of A (the bias gradient of the same conv),
                                   taken from the A tiles the kernel stages anyway; summed by bmc_pgemm_reduce_weight */
    int math;                   /* BMC_MATH_* as for bmc_conv; bf16 modes: v_mfma_f32_32x32x16_bf16 on planes read
                                   with the transposing LDS load */
    int tap_groups;             /* 0 / 1: a workgroup accumulates all taps; 3 (taps = 9, fp32 or bf16 arithmetic): one tap row per
                                   workgroup = three times the workgroups per pixel split, for small images (fewer,
                                   smaller slab writes); same slab layout */
} bmc_pgemm_args_t;
int bmc_pgemm(const bmc_pgemm_args_t* host_args, bmc_stream_t s);
/* slabs -> dW[Cout][Cin][taps] (nn.Conv2d layout) through kmap; beta 0/1 = overwrite/accumulate */
int bmc_pgemm_reduce_weight(const float* slabs, int nsplit, int G, int taps, int M, int N, const int* kmap,
                            int Cin, float* dw, int accumulate, const float* bias_slabs /* or NULL */,
                            float* db /* [G][M] or NULL */, bmc_stream_t s);
/* the same with one destination per group (HOST arrays of G <= 4 device pointers): weights that ONE grouped launch stacks
 * but that belong to separate parameters (v1 / v2, conv_hp / conv_hn) -- each group's sum goes (=|+=) straight to its own
 * dw[g] [M][Cin][taps] / db[g] [M]. */
int bmc_pgemm_reduce_weight_groups(const float* slabs, int nsplit, int G, int taps, int M, int N, const int* kmap,
                                   int Cin, float* const* dw, int accumulate, const float* bias_slabs, float* const* db,
                                   bmc_stream_t s);
/* slabs -> out[G][M][N] * scale */
int bmc_pgemm_reduce_plain(const float* slabs, int nsplit, int G, int M, int N, float scale,
                           float* out, bmc_stream_t s);

/* ---- weight gradient of a dense 3x3, 128 -> 128 channel convolution through the Winograd transform F(2x2, 3x3) ----
 * The same sum as bmc_pgemm (taps = 9) + bmc_pgemm_reduce_weight for this shape -- F.conv2d's weight / bias gradient at
 * models/submodules.py:25-26,33-34 (the residual blocks: 88 % of the network's 3x3 weight-gradient work) -- with 16 instead
 * of 36 multiplies per 2x2 output tile and channel pair on the fp32 MFMA:
 *   dU[xi][nu] = sum over tiles (A dY A^T)[xi][nu]^T (B^T d B)[xi][nu],  dW = G^T dU G,  db = sum of dY;
 * fp32 throughout, deterministic (partial sums per workgroup, added in a fixed order).
 *   dy, x:   NHWC tensors [B,H,W,128] (nch == pix_stride == 128; batch stride / shift / modulus as everywhere);
 *            dy = gradient of the convolution's output, x = its input
 *   nsplit:  workgroups per position row, 1 .. bmc_wgrad_wino_nsplit(B, H, W) (which returns the count that fills the chip)
 *   part:    workspace of nsplit * 16 * 128 * 128 floats;  bias_part: NULL or nsplit * 128 floats
 * bmc_wgrad_wino_reduce:  dw[co][k0 + ci][3][3] (=|+=) the gradient, dw = a [128][ldw][3][3] weight tensor of which the launch's
 * input channels are columns [k0, k0 + 128);  db[128] (=|+=) the bias gradient (bias_part and db go together). */
int bmc_wgrad_wino_nsplit(int B, int H, int W);
int bmc_wgrad_wino(const bmc_src_t* dy, const bmc_src_t* x, int B, int H, int W, int nsplit, float* part, float* bias_part,
                   bmc_stream_t s);
int bmc_wgrad_wino_reduce(const float* part, int nsplit, float* dw, int ldw, int k0, int accumulate, const float* bias_part,
                          float* db, bmc_stream_t s);
/* The same sum over SEVERAL (dy, x) operand pairs in one launch -- the uses of one weight by the weight-sharing blocks of a
 * window (models/BMCNet.py:19-32: `para_reschunk` holds ONE ParallelBlk n_b times), which autograd would add one by one:
 * dy[i], x[i] with batches[i] images each, i < nseg <= 8; nsplit <= bmc_wgrad_wino_nsplit(sum of batches, H, W); the result
 * goes through bmc_wgrad_wino_reduce as before (one partial-sum set, one reduction instead of nseg). */
int bmc_wgrad_wino_multi(const bmc_src_t* dy, const bmc_src_t* x, const int* batches, int nseg, int H, int W, int nsplit,
                         float* part, float* bias_part, bmc_stream_t s);

/* ---- the same weight gradient through F(4x4, 3x3) (round 5): 36 multiplies per 4x4 output tile and channel pair, i.e. 2.25 per
 * output pixel against 4 -- the transform of bmc_conv's BMC_MATH_FP32_WINO4 forward / data-gradient kernel applied to
 * F.conv2d's weight gradient (models/submodules.py:25-26,33-34).  Same operands, same reduce semantics, same determinism;
 * 3.0e-6 rel-L2 per convolution against float64 (contract 1e-3).
 *   nsplit:  workgroups per output slice, 1 .. bmc_wgrad_wino4_nsplit(B, H, W) (8 slices x nsplit workgroups fill the chip)
 *   part:    workspace of nsplit * 36 * 128 * 128 floats;  bias_part: NULL or nsplit * 128 floats */
int bmc_wgrad_wino4_nsplit(int B, int H, int W);
int bmc_wgrad_wino4(const bmc_src_t* dy, const bmc_src_t* x, int B, int H, int W, int nsplit, float* part, float* bias_part,
                    bmc_stream_t s);
int bmc_wgrad_wino4_reduce(const float* part, int nsplit, float* dw, int ldw, int k0, int accumulate, const float* bias_part,
                           float* db, bmc_stream_t s);

/* table[i] = ptrs[i], i < n <= 256: a device table of per-image base pointers for bmc_src_t's BMC_SRC_TABLE mode, written by a
 * kernel on stream s (the pointers travel in its argument block: no host buffer has to outlive the call). */
int bmc_ptr_table(const unsigned long long* ptrs, int n, unsigned long long* table, bmc_stream_t s);

/* ---- streaming kernels ---------------------------------------------------*/
/* out[i] = sum_{k < groups} in[k*n + i] (fixed order): gradient of an operand shared by several batch groups of a launch */
int bmc_group_sum(const float* in, int groups, long long n, float* out, bmc_stream_t s);
/* column sums over pixels (bias gradients): out[c] (+)= sum_p x[p*pix_stride + c]; ws >= 2048*C floats */
int bmc_colsum(const float* x, long long npix, int pix_stride, int C, float* ws, float* out,
               int accumulate, bmc_stream_t s);
/* ReLU backward: g = y > 0 ? dy : 0 (F.relu at models/BMCNet.py:64-80, submodules.py:33) */
int bmc_relu_bwd(const float* dy, const float* y, float* g, long long n, bmc_stream_t s);
/* LayerNorm2d over channels per pixel: models/submodules.py:127-140 (fwd), :141-154 (bwd).
 * stats = [npix][2] (mean, rstd).  bwd: gx, and dgamma/dbeta (+)= via ws (>= 2*1024*C floats). */
int bmc_layernorm_fwd(const float* x, const float* gamma, const float* beta, long long npix, int C,
                      float eps, float* y, float* stats, bmc_stream_t s);
int bmc_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                      long long npix, int C, float* dx, float* ws, float* dgamma, float* dbeta,
                      int accumulate, bmc_stream_t s);
/* row softmax of [rows][C] (torch.softmax(att, -1), models/submodules.py:72-73) and its backward
 * dA = P * (dP - rowsum(dP*P)) * scale_out */
int bmc_softmax_fwd(const float* a, long long rows, int C, float* p, bmc_stream_t s);
int bmc_softmax_bwd(const float* p, const float* dp, long long rows, int C, float scale_out,
                    float* da, bmc_stream_t s);

/* ---- head / tail of a recurrent window -----------------------------------
 * bmc_pack_inputs: models/BMCNet.py:106-112 -- polarity split + x3 repeat of the
 * two frames into two NHWC tensors of 16 channels each
 * [f1,f1,f1,f2,f2,f2,0...] (p: polarity 0, n: polarity 1).  x is [B,2,T,H,W]
 * with arbitrary element strides (sb,sc,st,sy,sx). */
int bmc_pack_inputs(const float* x, long long sb, long long sc, long long st, long long sy, long long sx,
                    int B, int H, int W, int repeat, float* xin_p, float* xin_n, bmc_stream_t s);
/* HR NCHW [B,C,rH,rW] -> LR NHWC [B,H,W,C*r*r] (pixel_unshuffle, models/submodules.py:80-92;
 * also the backward of the head).  split = S > 1 stores the LR tensor as S batch-stacked channel groups
 * [S*B][H][W][C*r*r/S] (group s of sample b at batch s*B + b): with S = 2 that is [o[:, :s^2]; o[:, s^2:]], the operand
 * layout of the input-fusion convolutions (models/BMCNet.py:63) -- no torch.cat between the two.  */
int bmc_unshuffle_to_nhwc(const float* hr, int B, int C, int H, int W, int r, float* lr, int split, bmc_stream_t s);
/* LR NHWC [B,H,W,C*r*r] -> HR NCHW [B,C,rH,rW] (+ bilinear x r of base[B,C,H,W] given with strides
 * (sb,sc,sy,sx), align_corners=False) -- F.pixel_shuffle + F.interpolate + add, models/BMCNet.py:119;
 * base NULL -> pure shuffle (backward of pixel_unshuffle). */
int bmc_shuffle_to_hr(const float* lr, int B, int C, int H, int W, int r, const float* base,
                      long long sb, long long sc, long long sy, long long sx, float* hr, int split /* as above */,
                      bmc_stream_t s);

/* Head + loss of one window in one pass: pred = pixel_shuffle(x_o, r) + bilinear(base) (models/BMCNet.py:119) written to hr
 * [B,C,rH,rW], and loss[0] = mean((pred - gt)^2) (nn.MSELoss, train.py:233,647) from per-block partial sums
 * (partials: >= 2048 floats of workspace; fixed-order reduction).  gt: [B][C][rH][rW] with batch stride gt_batch_stride. */
int bmc_head_mse_fwd(const float* lr, int B, int C, int H, int W, int r, const float* base, long long sb, long long sc,
                     long long sy, long long sx, const float* gt, long long gt_batch_stride, float* hr, float* partials,
                     float* loss, bmc_stream_t s);
/* Its backward: dlr [B,H,W,C*r*r] = pixel_unshuffle(dpred + (2 gloss[0] / numel) (pred - gt)); dpred NULL = no gradient
 * from the next window, gloss (DEVICE scalar) NULL = no gradient from the loss. */
int bmc_head_mse_bwd(const float* dpred, const float* pred, const float* gt, long long gt_batch_stride, const float* gloss,
                     int B, int C, int H, int W, int r, float* dlr, bmc_stream_t s);

/* ---- fused "centre" chain of the BIE block ----------------------------------
 * forward: centre = clustering(LayerNorm2d(convf(cat[s0, s1])))  -- models/submodules.py:63-64 with LayerNormFunction
 * (:127-140) between the two 1x1 convolutions -- in ONE launch: z and y = LN(z) never reach HBM (the second GEMM takes
 * its pixel operand from the first one's accumulator registers).  Saved for backward: yhat = (z - mean)/sqrt(var + eps)
 * (before the affine) and rstd.  C = channels of s0, s1, yhat, centre: 32, 64 or 128.  All outputs are contiguous
 * [B][H][W][C] (rstd [B][H][W]).
 * wstream: 3C/16 slices of [C][16] floats = bmc_pack_weight(W_f, Kpad = 2C, Coutpad = C) followed by
 * bmc_pack_weight(W_c, Kpad = C, Coutpad = C). */
typedef struct bmc_chain_fwd_args {
    bmc_src_t s0, s1;           /* nch = C each; launch batch b reads them through their batch maps */
    const float* wstream;
    const float* bias_f;        /* [C] convf bias */
    const float* bias_c;        /* [C] clustering bias */
    const float* gamma;         /* [C] LayerNorm2d weight */
    const float* beta;          /* [C] LayerNorm2d bias */
    float eps;
    float* yhat;
    float* rstd;
    float* centre;
    int B, C, H, W;
} bmc_chain_fwd_args_t;
int bmc_chain_fwd(const bmc_chain_fwd_args_t* host_args, bmc_stream_t s);

/* backward of the same chain for the twin layout (2n launch batches; batch bb read s0 at bb % n and s1 at (bb + n) % 2n):
 *   dy = W_c^T dcentre;  dz = rstd * (g - yhat*mean_c(g*yhat) - mean_c(g)), g = dy*gamma  (LayerNormFunction.backward, :141-154);
 *   dz  [2n][H][W][C]  is written for the weight-gradient GEMM of convf;
 *   ds1 [(bb + n) % 2n] = W_f[:, C:]^T dz[bb];
 *   ds0 [b] = ds0_add[b] + W_f[:, :C]^T (dz[b] + dz[b + n])   (dz of both halves summed in registers: one GEMM, ds0 written once).
 * Weight / bias / affine gradients: bmc_pgemm on (dcentre, yhat) and (dz, s0, s1) + bmc_chain_affine_grads.
 * wstream: 5C/16 slices of [C][16]: T(W_c), T(W_c), T1(W_f), T1(W_f), T0(W_f) with
 * T(.) = bmc_pack_weight_t(., nkpad = C) and T0 / T1 the operators of W_f's first / second C input channels. */
typedef struct bmc_chain_bwd_args {
    bmc_src_t dcentre;          /* nch = C, 2n launch batches */
    const float* wstream;
    const float* gamma;
    const float* yhat;          /* [2n][H][W][C] from bmc_chain_fwd */
    const float* rstd;          /* [2n][H][W] */
    float* dz;
    float* ds1;
    float* ds0;
    bmc_src_t ds0_add;          /* ptr NULL -> none */
    int n, C, H, W;
} bmc_chain_bwd_args_t;
int bmc_chain_bwd(const bmc_chain_bwd_args_t* host_args, bmc_stream_t s);
/* G[C][C] = bmc_pgemm(dcentre, yhat) reduced with bmc_pgemm_reduce_weight, dbc[C] its bias column sums ->
 * dwc (=|+=) gamma[ci] G[co][ci] + dbc[co] beta[ci];  dgamma[ci] (=|+=) sum_co W_c[co][ci] G[co][ci];  dbeta (=|+=) W_c^T dbc
 * (accumulate 0 | 1; dwc may alias G); dbc_out, if not NULL, (=|+=) dbc. */
int bmc_chain_affine_grads(const float* G, const float* dbc, const float* Wc, const float* gamma, const float* beta, int C,
                           float* dwc, float* dbc_out, float* dgamma, float* dbeta, int accumulate, bmc_stream_t s);

/* ---- batched products of C x C matrices: the BIE's attention without the value tensor --------------------------------
 * models/submodules.py:63-73 computes v = conv1x1(x) and uses it twice, att = scale * bmm(center, v^T) and
 * out = bmm(softmax(att), v).  Both are linear in v = W_v x + b_v, so with G0 = center^T x and s = the column sums of center
 * (one bmc_pgemm launch with bias slabs on x instead of v)
 *     att = scale * (G0 W_v^T + s b_v^T),     out = (P W_v) x + P b_v
 * and v is never formed (bmc_hip/bie.py); the backward likewise needs only C x C matrices.  This entry point evaluates
 * those products for all samples in one launch:
 *     C[b][i][j]  (=|+=) alpha * ( sum_t sum_k A_t[b][i][k] B_t[b][k][j]  +  u[b][i] v[b][j] )
 *     vec[b][i]   (=|+=) alpha *   sum_t sum_k A_t[b][i][k] w_t[b][k]
 * for b < nbatch, i < M, j < N, k < K; t < nterms (1 or 2).  Every operand X of batch b (weight group g = b / batch_per_group)
 * starts at X.ptr + b * X_sb + g * X_sg (floats) and is indexed with its own row / column strides: transposed and per-group
 * operands, and results written into a column range of a wider matrix, need no copies.  u / v (together) and vec_out (with a w
 * in every term) are optional; c may be NULL when only vec_out is wanted.  fp32 FMAs in a fixed order. */
typedef struct {
    const float* a; long long a_sb, a_sg; int a_si, a_sk;   /* A[b][i][k] */
    const float* b; long long b_sb, b_sg; int b_sk, b_sj;   /* B[b][k][j] */
    const float* w; long long w_sb, w_sg; int w_sk;         /* optional w[b][k] */
} bmc_mm_term_t;
typedef struct {
    int nterms;
    bmc_mm_term_t t[2];
    int nbatch, batch_per_group;
    int M, N, K;
    float alpha;
    const float* u; long long u_sb, u_sg;                   /* optional u[b][i] (unit stride) */
    const float* v; long long v_sb, v_sg;                   /*          v[b][j] (unit stride) */
    float* c; long long c_sb, c_sg; int c_si, c_sj;         /* C[b][i][j] */
    float* vec_out; long long vo_sb, vo_sg;                 /* vec[b][i] (unit stride) */
    int accumulate;
} bmc_small_mm_args_t;
int bmc_small_mm(const bmc_small_mm_args_t* host_args, bmc_stream_t s);

/* ---- loss-side resize ------------------------------------------------------
 * F.interpolate(prediction, size=gt.size()[-2:], mode='bicubic', align_corners=False): train.py:227-231,
 * infer_BMCNet.py:77-78 (taken when scale * round(sensor / scale) != sensor, dataloader/h5dataset.py:88-100; EventZoom:
 * 124x224 -> 124x222).  x: `planes` contiguous [H][W] planes (NCHW with planes = B*C) -> y [planes][Ho][Wo].
 * ATen semantics (A = -0.75, source coordinate fma(in/out, dst + 0.5, -0.5), clamped tap indices).  bwd is the transposed
 * operator as a gather: gx[planes][H][W] is OVERWRITTEN, deterministic. */
int bmc_bicubic_resize_fwd(const float* x, long long planes, int H, int W, int Ho, int Wo, float* y, bmc_stream_t s);
int bmc_bicubic_resize_bwd(const float* gy, long long planes, int H, int W, int Ho, int Wo, float* gx, bmc_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* BMC_HIP_H */
