// Kernel-argument block of the Winograd weight-gradient kernel (wino_wgrad.hip).
#pragma once
#include "bmc_common.h"

struct WgradK {
    SrcDev a;            // dY  [B,H,W,128]
    SrcDev x;            // the convolution's input [B,H,W,128]
    int B, H, W;
    int SY, SX;          // stages per image: pairs of tile rows x groups of 8 tiles
    int nstages, nsplit;
    float* part;         // [nsplit][4 xi][4 nu][128 co][128 ci]
    float* bias_part;    // optional [nsplit][128]
};
