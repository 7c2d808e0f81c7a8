// LDS-DMA ring helpers shared by the 16x16x4-MFMA kernels (chain.hip, conv1.hip).
#pragma once
#include "bmc_common.h"

// 16 bytes per lane from global memory straight into LDS (lane-linear image at the wave-uniform LDS byte address):
// address = uniform base (SGPR pair) + this lane's 32-bit byte offset.  Inline asm on purpose -- the compiler must not
// track these as LDS stores (it would drain vmcnt(0) before every later ds_read and the rings could never run ahead);
// completion is waited for with counted vmcnt (dma_wait) before the barrier that publishes a stage.
// (readfirstlane: the base pointer and the LDS address are wave-uniform, but must BE in SGPRs; s_nop 4: wait states between
// the VALU-written SGPRs / m0 and the VMEM instruction -- inline asm is opaque to the hazard recognizer.  m0 is reserved and
// cannot be named as a clobber; nothing else in these kernels uses it.)
__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_addr) {
    const unsigned long long pv = reinterpret_cast<unsigned long long>(gbase);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pv), hi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    const void* const sb = reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo);
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(la) : "memory");
}
// The same with a full per-lane 64-bit source address (boundary tiles: lanes outside the image point at a zero buffer).
// Every LDS-DMA of these kernels goes through one of the two asm forms, so that every write of m0 is in hand-written
// asm: mixing them with __builtin_amdgcn_global_load_lds (whose m0 setup the compiler may hoist, believing an asm
// statement leaves m0 alone -- it cannot be declared as a clobber) would be a latent mis-addressing hazard.
__device__ __forceinline__ void dma16v(const void* vaddr, unsigned lds_addr) {
    const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(la) : "memory");
}
template <int N>
__device__ __forceinline__ void dma_wait() {   // all but the newest N vector-memory operations of this wave are done
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// Raw barrier: no vmcnt(0) drain of the rings' prefetch (a __syncthreads() would); LDS reads of this wave retire first.
__device__ __forceinline__ void ring_publish() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// Rows are 64 B (16 floats) without padding; the 16-byte quads of a row are XOR-swizzled with SWZ[(row >> 2) & 3] on the
// DMA's SOURCE address and on the fragment reads: conflict-free ds_read_b128 for the 16-row x 4-quad fragment shape of
// the 16x16x4 MFMA operands (lane l reads row l & 15, quad l >> 4), while the LDS destination of a DMA stays lane-linear.
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (4 * ((row >> 2) & 3))) & 3; }      // {0, 2, 3, 1}
