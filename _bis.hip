// Conditional e-block in ONE launch, second generation: split-bf16 matrix arithmetic + register-resident parameters.
//
//   jf_cond_gf_pack_f32            W2 / b2 of the amortisation MLP -> the packed image the kernel streams (once per weight version)
//   jf_cond_gf_chain_inv_split_f32 log-prob direction of a conditional Euclidean block = mlp_predictors[i](...) (main/default.py:656-670,
//                                  946-962) + the per-block layer loop of all_layer_inverse (main/default.py:998-1031)
//
// What changed against cond_kernels.hip (exact-f32 MFMA, parameter tile in LDS), and why:
//  * f32-input MFMA runs at the VECTOR rate on CDNA4 (64 flop/clk/SIMD) and does not overlap with VALU work, so the 128 -> 548 product
//    alone cost 0.95 ms per 2^20 rows.  Here every f32 operand is split into three bf16 pieces (8 + 8 + 8 significant bits:
//    v = hi + mid + lo exactly, each piece rounded to nearest), and the product is evaluated as the six bf16 MFMAs whose piece
//    indices sum to <= 2, accumulated in f32 (v_mfma_f32_16x16x32_bf16).  Every bf16 x bf16 product is exact in f32; the dropped terms are
//    <= 3 * 2^-24 |w||h| -- the same size as the rounding of one f32 multiply.  Six passes at 16x the f32 rate = 0.37x the matrix time.
//    The first layer (K1 <= 28 inputs) stays on exact f32 MFMA: it is 1 % of the flops.
//  * The MFMA result layout IS the flow layout.  W2's rows are permuted on the host side of the launch (pack kernel) so that the
//    accumulator registers of lane (row n = lane % 16, coordinate d = lane / 16) hold exactly the parameters that lane needs for its
//    coordinate: register 4 t + r of column tile t = slot (4 t + r) of { mean_0..9, log_width_0..9, log_weight_0..9, householder_0..3,
//    offset, pad }.  The parameter block therefore never exists outside the register file: no LDS tile, no 34 ds_read per lane and layer.
//    The three reductions over a row's coordinates (Householder dots, sum of log-derivatives) run over lanes {l, l^16, l^32, l^48} with
//    v_permlane16_swap / v_permlane32_swap (2 swaps + 2 adds).
//  * W2 is streamed as ready-made MFMA A-fragments (1 KiB per fragment, lane-contiguous 16 bytes: conflict-free ds_read_b128, straight
//    memcpy from the packed image), 3 column tiles x 4 k-steps x 3 pieces = 36 KiB per chunk, shared by the 4 waves of a workgroup.
//
// Supported: float32, D in {3, 4}, layers with the reference's default options (K = 10 components, smooth-saturation widths, fitted and
// regulated weights, <= 4 Householder reflections), H <= 128, K1 <= 28.  Everything else: jf_cond_gf_chain_inv_* / jf_mlp2 + jf_gf_chain_inv.
#include "jf_cond_split.h"
#include <cstdlib>

namespace jf {

// ---------------------------------------------------------------------------------------------------------- packing
struct CsPackArgs {
    const float* W2; int64_t w2s; const float* b2;
    int H, D, n_layers, N;
    CsPackLayer L[JF_MAX_CHAIN];
    unsigned char* out;
};

// JF_SPLIT_F16X2: largest |W2| entry (bit pattern of a non-negative float orders like an unsigned) into the 16-byte tail of the image
__global__ void __launch_bounds__(256) cs_absmax_kernel(const CsPackArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    if (idx < (int64_t)a.N * a.H) m = fabsf(a.W2[(idx / a.H) * a.w2s + idx % a.H]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(a.out + (size_t)a.n_layers * CS_CPL * CS_CHUNK16_BYTES), __builtin_bit_cast(unsigned, m));
}
// power-of-two scale of W2 for the f16 pieces: the largest entry lands in [2^14, 2^15)
__device__ __forceinline__ int cs_w_exponent(float wmax) { return (wmax > 0.f && wmax < INFINITY) ? 14 - ilogbf(wmax) : 0; }

// one thread per (chunk, tile, k-step, lane): writes the three pieces' fragments (16 bytes each); the first 48 threads of a chunk's
// first k-step also write the bias
template <int NP> __global__ void __launch_bounds__(256) cs_pack_kernel(const CsPackArgs a) {
    using G = CsGeom<NP>;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int s = rest % CS_KSTEPS; rest /= CS_KSTEPS;
    const int tc = rest % CS_CT; rest /= CS_CT;
    const int chunk = rest;                                         // consumption order: last layer first
    if (chunk >= a.n_layers * CS_CPL) return;
    const int l = a.n_layers - 1 - chunk / CS_CPL;
    const int tile = (chunk % CS_CPL) * CS_CT + tc;
    const CsPackLayer o = a.L[l];
    const int m = lane & 15, q = lane >> 4;
    const int col = cs_slot_column(o, a.D, 4 * tile + (m & 3), m >> 2);
    bf16x8 f[NP];
    int e = 0;
    if constexpr (NP == 2) e = cs_w_exponent(*reinterpret_cast<const float*>(a.out + (size_t)a.n_layers * CS_CPL * G::CHUNK));
    const float wscale = ldexpf(1.0f, e);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 16 * (2 * s + (i >> 2)) + 4 * q + (i & 3);    // hidden unit of k-slot i of lane group q in k-step s (matches the h layout)
        const float w = (col >= 0 && k < a.H) ? a.W2[(int64_t)(o.col0 + col) * a.w2s + k] : 0.0f;
        if constexpr (NP == 3) {
            __bf16 p0, p1, p2;
            cs_split(w, p0, p1, p2);
            f[0][i] = p0; f[1][i] = p1; f[2][i] = p2;
        } else {
            const float ws = w * wscale;                            // exact (power of two)
            const _Float16 hi = (_Float16)ws;
            const _Float16 lo = (_Float16)(ws - (float)hi);
            f[0][i] = __builtin_bit_cast(__bf16, hi); f[1][i] = __builtin_bit_cast(__bf16, lo);
        }
    }
    unsigned char* base = a.out + (size_t)chunk * G::CHUNK;
#pragma unroll
    for (int p = 0; p < NP; ++p)
        *reinterpret_cast<bf16x8*>(base + (size_t)((tc * CS_KSTEPS + s) * NP + p) * CS_FRAG + lane * 16) = f[p];
    if (s == 0 && lane < 16) {
        const float b = (col >= 0 && a.b2 != nullptr) ? a.b2[o.col0 + col] : 0.0f;
        reinterpret_cast<float*>(base + G::W)[tc * 16 + m] = NP == 2 ? b * ldexpf(1.0f, e + 14) : b;      // f16 pieces: in the accumulators' units
    }
    if constexpr (NP == 2) {
        if (s == 0 && tc == 0 && lane < 4) reinterpret_cast<float*>(base + G::W + CS_B_BYTES)[lane] = lane == 0 ? ldexpf(1.0f, -(e + 14)) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------- the fused kernel
struct CsArgs {
    const float* in; int64_t in_stride;
    const float* W1; int64_t w1s; const float* b1;
    const unsigned char* packed;
    int K1, H;
    const float* x; int64_t xs;
    const float* ld_in;
    int64_t B;
    int D, n_layers;
    CsLayer L[JF_MAX_CHAIN];
    float* x_out; int64_t xos;
    float* ld_out;
    const float* blp_in; float* blp_out;
    int32_t* status;
    float* aux;                                // SAVE: what the adjoint launch starts from (see cond_bwd_kernels.hip), else unused
};

// RG = row groups (16 rows each) per wave.  With RG = 2 every A fragment read from LDS feeds two MFMAs (half the ds_read_b128 per row,
// six independent accumulators per piece product instead of three) and the chunk barriers are paid once per 128 rows instead of 64.
// SAVE (log-prob direction with gradients wanted): every layer's input coordinate and mixture sums go to a.aux, 5 floats per (layer, row,
// coordinate lane) -- 320 bytes per row of a 4-layer block instead of the 2.2 KB parameter row the adjoint would otherwise need.
template <int RG, bool FWD, bool SAVE = false, int NP = CS_NP> __global__ void __launch_bounds__(256, 2) cond_gf_split_kernel(const CsArgs a) {
}
// ---------------------------------------------------------------------------------------------------------- host side
static int cs_forced_rg = 0;                     // 0: by batch size

static bool cs_arith_ok(int arithmetic) { return arithmetic == JF_SPLIT_BF16X3 || arithmetic == JF_SPLIT_F16X2; }
static int64_t cs_image_bytes(int n_layers, int arithmetic) {
    return arithmetic == JF_SPLIT_F16X2 ? (int64_t)n_layers * CS_CPL * CS_CHUNK16_BYTES + 16 : (int64_t)n_layers * CS_CPL * CS_CHUNK_BYTES;
}

static int cs_pack(const float* W2, int64_t w2s, const float* b2, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers, int arithmetic,
                   void* packed, void* stream) {
    if (!W2 || !layers || !packed || !cs_arith_ok(arithmetic)) return JF_ERR_BADARG;
    if (!width_ok(H) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (H > CS_HMAX) return JF_ERR_UNSUPPORTED;
    CsPackArgs a{};
    int col = 0;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        if (!cs_layer_supported(h, D)) return JF_ERR_UNSUPPORTED;
        CsPackLayer& o = a.L[l];
        const int kd = h.num_kde * D;
        o.col0 = col; o.hh = h.hh_iter; o.model_offset = h.model_offset;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + h.hh_iter * D;
        o.off_lw = o.off_mean + kd;
        o.off_ln = o.off_lw + kd;
        col += o.off_ln + kd;
    }
    a.W2 = W2; a.w2s = w2s; a.b2 = b2; a.H = H; a.D = D; a.n_layers = n_layers; a.N = col; a.out = static_cast<unsigned char*>(packed);
    const int threads = n_layers * CS_CPL * CS_CT * CS_KSTEPS * 64;
    hipStream_t st = (hipStream_t)stream;
    if (arithmetic == JF_SPLIT_F16X2) {
        if (hipMemsetAsync(a.out + (size_t)n_layers * CS_CPL * CS_CHUNK16_BYTES, 0, 16, st) != hipSuccess) return JF_ERR_LAUNCH;
        hipLaunchKernelGGL(cs_absmax_kernel, dim3((unsigned)(((int64_t)col * H + 255) / 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(cs_pack_kernel<2>, dim3((threads + 255) / 256), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(cs_pack_kernel<3>, dim3((threads + 255) / 256), dim3(256), 0, st, a);
    }
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

template <bool FWD, bool SAVE, int NP>
static int cs_launch(const CsArgs& a, int64_t B, hipStream_t st) {
    // phase 1's scratch overlays chunk buffer 1 and may be larger than it (K1 = 28 with two row groups: 30 KB)
    const int k1p = (a.K1 + 3) / 4 * 4;
    auto lds_of = [&](int rg) {
        const size_t scratch = ((size_t)(CS_ROWS1 * rg + CS_HMAX) * (k1p + 1) + CS_HMAX) * 4;
        const size_t second = scratch > (size_t)CsGeom<NP>::CHUNK ? (scratch + 15) / 16 * 16 : (size_t)CsGeom<NP>::CHUNK;
        return (size_t)CsGeom<NP>::CHUNK + second;
    };
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)cond_gf_split_kernel<1, FWD, SAVE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)cond_gf_split_kernel<2, FWD, SAVE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    // two row groups per wave once that still leaves every CU several workgroups; JF_CS_RG=1|2 (environment, read once) or
    // jf_cond_gf_split_row_groups() force a variant (A/B timing: scripts/probe/rg_sweep.py; both variants in one process: the stress tests)
    static const int env_rg = getenv("JF_CS_RG") ? atoi(getenv("JF_CS_RG")) : 0;
    const int force_rg = cs_forced_rg ? cs_forced_rg : env_rg;
    const bool two = force_rg ? force_rg == 2 : B >= (int64_t)CS_ROWS1 * 2 * 1024;
    if (two) hipLaunchKernelGGL((cond_gf_split_kernel<2, FWD, SAVE, NP>), dim3((unsigned)((B + 2 * CS_ROWS1 - 1) / (2 * CS_ROWS1))), dim3(256), lds_of(2), st, a);
    else hipLaunchKernelGGL((cond_gf_split_kernel<1, FWD, SAVE, NP>), dim3((unsigned)((B + CS_ROWS1 - 1) / CS_ROWS1)), dim3(256), lds_of(1), st, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

template <bool FWD>
static int cs_chain(const float* in, int64_t in_stride, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1, int32_t H,
                    const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                    int64_t xos, float* ld_out, const float* blp_in, float* blp_out, int32_t* status, void* stream, float* aux = nullptr,
                    int arithmetic = JF_SPLIT_BF16X3) {
    if (!in || !W1 || !b1 || !packed || !x || !x_out || !ld_out || !layers || !cs_arith_ok(arithmetic)) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (K1 > CS_K1MAX || H > CS_HMAX || (reinterpret_cast<uintptr_t>(packed) & 15u)) return JF_ERR_UNSUPPORTED;
    CsArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        if (!cs_layer_supported(h, D)) return JF_ERR_UNSUPPORTED;
        CsLayer& o = a.L[l];
        o.hh = h.hh_iter; o.model_offset = h.model_offset; o.inv_type = h.inverse_function_type;
        o.wmin = (float)h.width_min; o.inv_wmax = (float)(1.0 / h.width_max); o.nmin = (float)h.norm_min; o.nmax = (float)h.norm_max;
    }
    if (aux && (FWD || (reinterpret_cast<uintptr_t>(aux) & 15u))) return JF_ERR_BADARG;
    if (B == 0) return JF_OK;
    a.in = in; a.in_stride = in_stride; a.W1 = W1; a.w1s = w1s; a.b1 = b1; a.packed = static_cast<const unsigned char*>(packed); a.K1 = K1; a.H = H;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.D = D; a.n_layers = n_layers;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status; a.aux = aux;
    hipStream_t st = (hipStream_t)stream;
    if constexpr (!FWD) {
        if (aux) return arithmetic == JF_SPLIT_F16X2 ? cs_launch<false, true, 2>(a, B, st) : cs_launch<false, true, 3>(a, B, st);
    }
    return arithmetic == JF_SPLIT_F16X2 ? cs_launch<FWD, false, 2>(a, B, st) : cs_launch<FWD, false, 3>(a, B, st);
}

}  // namespace jf

extern "C" {
int64_t jf_cond_gf_packed_bytes2(int32_t D, int32_t n_layers, const jf_gf_layer* layers, int32_t arithmetic) {
    if (!layers || n_layers < 1 || n_layers > JF_MAX_CHAIN || !jf::cs_arith_ok(arithmetic)) return JF_ERR_BADARG;
    for (int l = 0; l < n_layers; ++l)
        if (!jf::cs_layer_supported(layers[l], D)) return JF_ERR_UNSUPPORTED;
    return jf::cs_image_bytes(n_layers, arithmetic);
}
int64_t jf_cond_gf_packed_bytes(int32_t D, int32_t n_layers, const jf_gf_layer* layers) { return jf_cond_gf_packed_bytes2(D, n_layers, layers, JF_SPLIT_BF16X3); }
int jf_cond_gf_split_row_groups(int32_t rg) {
    const int prev = jf::cs_forced_rg;
    if (rg >= 0 && rg <= 2) jf::cs_forced_rg = rg;
    return prev;
}
int jf_cond_gf_pack2_f32(const float* W2, int64_t w2s, const float* b2, int32_t H, int32_t D, int32_t n, const jf_gf_layer* L, int32_t arithmetic,
                         void* packed, void* s) {
    return jf::cs_pack(W2, w2s, b2, H, D, n, L, arithmetic, packed, s);
}
int jf_cond_gf_pack_f32(const float* W2, int64_t w2s, const float* b2, int32_t H, int32_t D, int32_t n, const jf_gf_layer* L, void* packed, void* s) {
    return jf::cs_pack(W2, w2s, b2, H, D, n, L, JF_SPLIT_BF16X3, packed, s);
}
int jf_cond_gf_chain_inv_split_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1,
                                   int32_t H, const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n,
                                   const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int32_t* st, void* s) {
    return jf::cs_chain<false>(in, is, W1, w1s, b1, packed, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
int jf_cond_gf_chain_inv_split_save_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1,
                                        int32_t H, const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n,
                                        const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, float* aux,
                                        int32_t* st, void* s) {
    if (!aux) return JF_ERR_BADARG;
    return jf::cs_chain<false>(in, is, W1, w1s, b1, packed, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s, aux);
}
int jf_cond_gf_chain_split2_f32(int32_t direction, int32_t arithmetic, const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1,
                                const void* packed, int32_t K1, int32_t H, const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D,
                                int32_t n, const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, float* aux,
                                int32_t* st, void* s) {
    if (direction == JF_DIR_INV) return jf::cs_chain<false>(in, is, W1, w1s, b1, packed, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s, aux, arithmetic);
    if (direction == JF_DIR_FWD && !bi && !bo && !aux)
        return jf::cs_chain<true>(in, is, W1, w1s, b1, packed, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, nullptr, nullptr, st, s, nullptr, arithmetic);
    return JF_ERR_BADARG;
}
int64_t jf_cond_gf_aux_floats(int64_t B, int32_t n_layers) {
    return (jf::rows_ok(B) && n_layers >= 1 && n_layers <= JF_MAX_CHAIN) ? (int64_t)n_layers * B * 20 : (int64_t)JF_ERR_BADARG;
}
int jf_cond_gf_chain_fwd_split_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1,
                                   int32_t H, const float* z, int64_t zs, const float* ld_in, int64_t B, int32_t D, int32_t n,
                                   const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, int32_t* st, void* s) {
    return jf::cs_chain<true>(in, is, W1, w1s, b1, packed, K1, H, z, zs, ld_in, B, D, n, L, xo, xos, ldo, nullptr, nullptr, st, s);
}
}
