// Merged log-prob step: which kernels of the other translation units the merge sink (merged_kernels.hip) recognises.
#pragma once
#include "jf_common.h"

namespace jf {

// addresses of the stand-alone kernels whose launches a merge may absorb (defined next to the kernels)
const void* gfb_inv_kernel_f32(int D);                 // gf_kernels.hip:          gfb_chain_inv_kernel<float, D>, D = 1..8
const void* gfbg_inv_kernel_f32(int D);                //                          gfbg_chain_inv_kernel<float, D, G>, D = 2..4 (G lanes per row)
const void* cond_f_inv_kernel_f32();                   // cond_manifold_kernels.hip: cond_mchain_kernel<float, FFam, 256, false>

}  // namespace jf
