// Conditional MANIFOLD block in ONE launch: the default amortisation MLP (Linear -> tanh -> Linear, main/default.py:656-670) + the chain of
// 'r' / 'o' / 'm' / 'f' layers it parametrises (main/default.py:998-1031), log-prob direction:  jf_cond_{r,o,m,f}_chain_inv_*.
//
// These blocks are small (4 -> 128 -> 10 for the 'f' block of the metric configuration) and were two latency-bound launches (0.17 + 0.06 ms
// per 2^20 rows against ~0.03 ms of arithmetic).  Here a wave owns 64 rows: per 16-row tile it computes the hidden activations on the
// matrix cores into registers (transposed products as in mlp_kernels.hip: the first result is the second's B operand), multiplies them
// with W2 (<= 64 output columns, whole matrix in LDS) and drops the 16 x N parameter rows into its LDS tile; then the 64 lanes run the
// layers lane-per-row on that tile exactly as the stand-alone chain kernel does (same Fam::apply device code).
//
// Round 4, float32: the SECOND product (128 -> N) runs on the f16 matrix pipe with the arithmetic of the fused g block (jf_cond_split.h: every
// f32 operand as two f16 pieces scaled into the normal range, three v_mfma_f32_16x16x32_f16 passes, f32 accumulation; error below a plain f32
// matrix product's own rounding).  Exact-f32 MFMA issues at the VECTOR rate on CDNA4 and each of its operands was a scalar LDS read: the block
// spent 0.06 of its 0.108 ms per 2^20 rows there.  W2 is cut into fragments by the workgroup itself (<= 64 x 128 values, L2-resident).
#include "jf_cond_mchain.h"
#include "jf_merge.h"

namespace jf {

// (CmArgs and the kernel body: jf_cond_mchain.h)
// NT threads per workgroup (256 in float32; 128 in float64, whose lane-private knot tables are twice as large)
// (four workgroups per CU instead of three, round 5: 104 VGPRs allow it, the `f` block measures 0.107 instead of 0.105 ms -- not taken)
// (float32: a register budget for three waves per SIMD -- 140 + 32 AGPRs -> 112 VGPRs, the `f` block of C3 0.109 -> 0.105 ms per 2^20 rows)
template <typename T, class Fam, int NT, bool FWD = false>
__global__ void __launch_bounds__(NT, NT == 256 ? 3 : 1) cond_mchain_kernel(const CmArgs<T, typename Fam::CLayer> a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    cond_mchain_body<T, Fam, NT, FWD>(a, (int)blockIdx.x, (int)gridDim.x, smem_raw);
}


template <typename T, class Fam, bool FWD = false>
static int cond_mchain(const T* in, int64_t in_stride, const T* W1, int64_t w1s, const T* b1, const T* W2, int64_t w2s, const T* b2, int32_t K1, int32_t H,
                       const T* x, int64_t xs, const T* ld_in, int64_t B, int32_t n_layers, const typename Fam::CLayer* layers, T* x_out, int64_t xos,
                       T* ld_out, const T* blp_in, T* blp_out, int32_t* status, void* stream) {
    if (!in || !W1 || !b1 || !W2 || !x || !x_out || !ld_out || !layers) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_MCHAIN) return JF_ERR_BADARG;
    if (K1 > CM_K1MAX || H > CM_HMAX) return JF_ERR_UNSUPPORTED;
    CmArgs<T, typename Fam::CLayer> a{};
    int col = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!Fam::sane(layers[l])) return JF_ERR_BADARG;
        a.L[l] = layers[l];
        a.col0[l] = col;
        col += Fam::row_len(layers[l]);
        if constexpr (std::is_same<Fam, FFam>::value) { if (layers[l].correlated) return JF_ERR_UNSUPPORTED; }
    }
    if (col < 1 || col > CM_NMAX) return JF_ERR_UNSUPPORTED;
    if (B == 0) return JF_OK;
    a.in = in; a.in_stride = in_stride; a.W1 = W1; a.w1s = w1s; a.b1 = b1; a.W2 = W2; a.w2s = w2s; a.b2 = b2; a.K1 = K1; a.H = H; a.N = col;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.n_layers = n_layers; a.dim = Fam::DIM;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    const int np = (col + 15) / 16 * 16;
    a.tile_stride = np + 1;
    a.scratch = 0;
    int n_spl = 0;
    for (int l = 0; l < n_layers; ++l) n_spl += Fam::n_bins(layers[l]);
    a.tab = 0;                                         // lane-private knot tables: none for a spline-free chain (default 'f', 'm'), else 3 (bins + 1)
    if (n_spl > 0) {                                   // words where the family states its bin count (r, o), the 16-bin maximum otherwise
        a.tab = 1;
        for (int l = 0; l < n_layers; ++l) { const int w = fam_tab_words<Fam>::of(layers[l]); a.tab = w > a.tab ? w : a.tab; }
    }
    const int k1p = (K1 + 3) / 4 * 4, ldk = k1p + 1;
    constexpr int NT = sizeof(T) == 4 ? 256 : 128;
    // (float32: the W2 region holds np / 16 x 4 k-steps x 2 pieces x 1 KiB of f16 fragments = np x 128 floats, less than the np x 129 reserved;
    //  8 more elements behind the bias for the workgroup's absmax reduction)
    const size_t lds = ((size_t)CM_HMAX * ldk + CM_HMAX + (size_t)np * (CM_HMAX + 1) + np + 8 + (size_t)NT * ldk + (size_t)NT * a.tile_stride +
                        (size_t)NT * a.tab) * sizeof(T);
    if (lds > 160 * 1024) return JF_ERR_UNSUPPORTED;
    auto k = cond_mchain_kernel<T, Fam, NT, FWD>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // one resident round of workgroups (occupancy x CUs, at most one per row tile)
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const int64_t n_tiles = (B + NT - 1) / NT, resident = (int64_t)cus * per_cu;
    jf::launch(k, dim3((unsigned)(n_tiles < resident ? n_tiles : resident)), dim3(NT), lds, (hipStream_t)stream, a);
    return check_launch();
}

const void* cond_f_inv_kernel_f32() { return (const void*)cond_mchain_kernel<float, FFam, 256, false>; }

}  // namespace jf

using namespace jf;

#define JF_DEFINE_COND_MCHAIN(fam, Fam, T, suffix)                                                                                                 \
    extern "C" int jf_cond_##fam##_chain_inv_##suffix(const T* in, int64_t is, const T* W1, int64_t w1s, const T* b1, const T* W2, int64_t w2s,       \
                                                      const T* b2, int32_t K1, int32_t H, const T* x, int64_t xs, const T* ld_in, int64_t B,         \
                                                      int32_t n, const jf_##fam##_layer* L, T* xo, int64_t xos, T* ldo, const T* bi, T* bo,          \
                                                      int32_t* st, void* s) {                                                                        \
        return cond_mchain<T, Fam>(in, is, W1, w1s, b1, W2, w2s, b2, K1, H, x, xs, ld_in, B, n, L, xo, xos, ldo, bi, bo, st, s);                     \
    }
#define JF_DEFINE_COND_MCHAIN_FWD(fam, Fam, T, suffix)                                                                                             \
    extern "C" int jf_cond_##fam##_chain_fwd_##suffix(const T* in, int64_t is, const T* W1, int64_t w1s, const T* b1, const T* W2, int64_t w2s,       \
                                                      const T* b2, int32_t K1, int32_t H, const T* z, int64_t zs, const T* ld_in, int64_t B,         \
                                                      int32_t n, const jf_##fam##_layer* L, T* xo, int64_t xos, T* ldo, int32_t* st, void* s) {      \
        return cond_mchain<T, Fam, true>(in, is, W1, w1s, b1, W2, w2s, b2, K1, H, z, zs, ld_in, B, n, L, xo, xos, ldo, nullptr, nullptr, st, s);     \
    }
JF_DEFINE_COND_MCHAIN_FWD(r, RFam, float, f32)
JF_DEFINE_COND_MCHAIN_FWD(r, RFam, double, f64)
JF_DEFINE_COND_MCHAIN_FWD(o, OFam, float, f32)
JF_DEFINE_COND_MCHAIN_FWD(o, OFam, double, f64)
JF_DEFINE_COND_MCHAIN_FWD(m, MFam, float, f32)
JF_DEFINE_COND_MCHAIN_FWD(m, MFam, double, f64)
JF_DEFINE_COND_MCHAIN_FWD(f, FFam, float, f32)
JF_DEFINE_COND_MCHAIN_FWD(f, FFam, double, f64)
JF_DEFINE_COND_MCHAIN(r, RFam, float, f32)
JF_DEFINE_COND_MCHAIN(r, RFam, double, f64)
JF_DEFINE_COND_MCHAIN(o, OFam, float, f32)
JF_DEFINE_COND_MCHAIN(o, OFam, double, f64)
JF_DEFINE_COND_MCHAIN(m, MFam, float, f32)
JF_DEFINE_COND_MCHAIN(m, MFam, double, f64)
JF_DEFINE_COND_MCHAIN(f, FFam, float, f32)
JF_DEFINE_COND_MCHAIN(f, FFam, double, f64)
