// The dual-number replay of a manifold chain's backward (manifold_bwd_kernels.hip) under C++ names: since round 6 the C entry points
// jf_{r,o,m,f}_chain_inv_bwd_* are the reverse-mode kernels of manifold_rev_kernels.hip, which hand over to these when JF_M_BWD_DUAL=1 (the check).
#pragma once
#include "jf_common.h"

namespace jf {
#define JF_DECLARE_DUAL_BWD(fam, T, suffix)                                                                                                   \
    int dual_##fam##_chain_inv_bwd_##suffix(const T* x, int64_t xs, const T* p, int64_t ps, int32_t pb, int64_t B, int32_t n, const jf_##fam##_layer* L, \
                                            const T* gxo, int64_t gxos, const T* gld, const T* gblp, T* gx, int64_t gxs, T* gp, int64_t gps, int32_t* st, void* s);
JF_DECLARE_DUAL_BWD(r, float, f32)
JF_DECLARE_DUAL_BWD(r, double, f64)
JF_DECLARE_DUAL_BWD(o, float, f32)
JF_DECLARE_DUAL_BWD(o, double, f64)
JF_DECLARE_DUAL_BWD(m, float, f32)
JF_DECLARE_DUAL_BWD(m, double, f64)
JF_DECLARE_DUAL_BWD(f, float, f32)
JF_DECLARE_DUAL_BWD(f, double, f64)
#undef JF_DECLARE_DUAL_BWD
}  // namespace jf
