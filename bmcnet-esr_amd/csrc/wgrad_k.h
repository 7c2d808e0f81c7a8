// Kernel-argument block of the Winograd weight-gradient kernel (wino_wgrad.hip).
#pragma once
#include "bmc_common.h"

constexpr int BMC_WG_MAXSEG = 8;

struct WgradK {
    // the launch sums over `nseg` (dY, x) operand pairs -- several uses of ONE weight (the five weight-sharing blocks of a window,
    // models/BMCNet.py:19-32) reduced by one launch: segment s holds the images [segb[s], segb[s + 1]) of the launch
    SrcDev a[BMC_WG_MAXSEG];     // dY  [B_s,H,W,128]
    SrcDev x[BMC_WG_MAXSEG];     // the convolution's input [B_s,H,W,128]
    int segb[BMC_WG_MAXSEG + 1];
    int nseg;
    int B, H, W;                 // B = all images of the launch
    int SY, SX;          // stages per image: pairs of tile rows x groups of 8 tiles
    int nstages, nsplit;
    float* part;         // [nsplit][4 xi][4 nu][128 co][128 ci]
    float* bias_part;    // optional [nsplit][128]
};
