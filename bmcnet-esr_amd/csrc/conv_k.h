// Kernel-argument block shared by the convolution kernels (conv.hip: fp32 MFMA; conv_bf.hip: bf16-plane MFMA).
#pragma once
#include "bmc_common.h"

struct ConvK {
    int nsrc;
    SrcDev src[BMC_MAX_SRC];
    const void* w;     // packed weights: fp32 [S][Coutpad][16] or bf16 planes [S][NP][Coutpad][16] (conv_bf.hip)
    const float* bias;
    long long w_group_stride;
    int bias_group_stride;
    int batch_per_group;
    float* out;
    long long out_batch_stride;
    int out_pix_stride;
    int B, H, W, Cout, Coutpad;
    int relu;
    SrcDev residual;
    SrcDev mask;
    int accumulate;
    int tiles_x, tiles_y, ntn, nchunks, ntiles;
};

// conv_bf.hip: launch the bf16-plane kernel (planes = 1: bf16 operands; 3: fp32 operands split into three bf16 planes,
// six plane products -- fp32-equivalent result) for an already validated argument block.
int bmc_conv_bf_launch(const ConvK& k, int taps, int BN, int TH, int planes, int cus, hipStream_t st);

// conv1.hip: 1x1 convolution on the 16x16x4 fp32 MFMA with LDS-DMA operand rings (large problems).  Returns 1 if it
// launched the problem, 0 if it is left to conv.hip's kernel.
int bmc_conv1_launch(ConvK k, int cus, hipStream_t st);

// conv1p.hip: the same for Coutpad % 128 == 0 and K = 128 / 256 with the weights resident in registers (tried first by
// bmc_conv1_launch).  Returns 1 if it launched the problem.
int bmc_conv1p_launch(ConvK k, int cus, hipStream_t st);

// wino.hip: 3x3 convolution through the Winograd transform F(2x2, 3x3) on the fp32 MFMA (math = BMC_MATH_FP32_WINO; weights from
// bmc_pack_weight_wino).  Returns 0, or < 0 with the error text set.
int bmc_conv_wino_launch(ConvK k, int cus, hipStream_t st);
// wino4.hip: the same through F(4x4, 3x3) (math = BMC_MATH_FP32_WINO4; weights from bmc_pack_weight_wino4; images at least 17
// pixels wide, 32-bit pixel offsets).  Returns 0, or < 0 with the error text set.
int bmc_conv_wino4_launch(ConvK k, int cus, hipStream_t st);
