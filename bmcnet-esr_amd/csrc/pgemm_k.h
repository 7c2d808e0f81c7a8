// Kernel-argument block shared by the pixel-reduction GEMM kernels (pgemm.hip: fp32 MFMA; pgemm_bf.hip: bf16 planes).
#pragma once
#include "bmc_common.h"

constexpr int PT_H = 4, PT_W = 16, PT = PT_H * PT_W;  // 64-pixel tile

struct PgemmK {
    SrcDev a;
    int nsrc;
    SrcDev src[BMC_MAX_SRC];
    int B, H, W;
    int batch_per_group;
    float* slabs;
    int nsplit;
    const float* zeros;
    float* bias_slabs;   // optional: per-workgroup column sums of the A tiles (bias gradient partials)
    int M, Mpad, N, Npad;
    int n_nblk, n_mblk, G;
    int tiles_x, tiles_y, tiles_per_img;
    int tap_groups;      // 3: one tap row per workgroup (bmc_pgemm_args_t.tap_groups), else all taps
};

// Does any operand of the launch name its images through a pointer table (BMC_SRC_TABLE)?  Such launches take the kernels' TAB
// instantiations (bmc_common.h, src_bp).
static inline bool pgemm_uses_tables(const PgemmK& k) {
    bool t = src_is_table(k.a);
    for (int i = 0; i < k.nsrc; ++i) t = t || src_is_table(k.src[i]);
    return t;
}

// pgemm_bf.hip: bf16-plane variant (planes = 1: bf16 operands; 3: exact 3-way split, six plane products).
int bmc_pgemm_cols(int taps, int math);   // columns of C per workgroup for (taps, math mode)
int bmc_pgemm_bf_launch(const PgemmK& k, int taps, int planes, hipStream_t st);
