// Conditional e-block in ONE launch, third generation: the split-bf16 block of cond_split_kernels.hip re-tiled for 32x32x16 MFMAs and
// scheduled so that the matrix pipe and the vector ALU of every SIMD are busy at the same time.
//
//   jf_cond_gf_pp_pack_f32          W2 / b2 of the amortisation MLP -> the packed image this kernel streams (once per weight version)
//   jf_cond_gf_chain_inv_pp_f32     log-prob direction of a conditional Euclidean block = mlp_predictors[i](...) (main/default.py:656-670,
//                                   946-962) + the per-block layer loop of all_layer_inverse (main/default.py:998-1031)
//
// Why (measurements: scripts/probe/mfma_valu.hip).  On a gfx950 SIMD one bf16 MFMA blocks vector issue for ~12 cycles whatever
// its shape, so a 16x16x32 MFMA (16 cycles of matrix pipe) leaves ~4 cycles for other instructions and a 32x32x16 MFMA (32 cycles) ~20; and the
// two waves a SIMD can hold at 256 registers each ran in lockstep in the second-generation kernel -- matrix phase beside matrix phase (pipe
// saturated, vector ALU idle), flow phase beside flow phase (the reverse): 23 k cycles per layer and wave pair where the pipe alone needs 14 k.
// Here a workgroup is TWO TEAMS of four waves, wave w of team A and wave w of team B on the same SIMD, and the teams run the same program half
// a layer apart: while one team multiplies (32x32x16 tiles, one MFMA per 32 cycles) the other evaluates the mixture / inverse-CDF arithmetic of
// its previous layer in the issue slots the MFMAs leave free.  The workgroup is persistent (one per CU, row tiles in a grid-stride loop), so
// the pipeline fills once per launch, not once per tile.
//
// Layout.  A wave owns 32 rows; lane (j = lane % 32, q = lane / 32) owns coordinates 2q and 2q+1 of row j.  The 32x32 MFMA result register
// r of column tile t in that lane is result row 8 (r / 4) + 4 q + r % 4 of the tile; the pack kernel permutes W2's rows so that this is
// parameter slot (16 t + r) % 36 of coordinate 2 q + (16 t + r) / 36: five tiles (80 registers, 72 used) carry the two coordinates' 36
// slots { mean_0..9, log_width_0..9, log_weight_0..9, householder_0..3, offset, pad }.  The parameter block therefore lives in the result
// registers, as in the second generation; sums over a row's coordinates are one in-lane add and one v_permlane32_swap.
// The hidden activations are the MFMA B operands (three bf16 pieces each, 96 registers); W2 streams through LDS as ready-made A fragments,
// one 60 KiB chunk (5 tiles x 4 k-steps x 3 pieces) per half layer, double buffered by LDS-DMA.  The first layer (K1 <= 32 inputs) runs on
// the same split-bf16 arithmetic, its weights split once per workgroup into LDS.
//
// Schedule (one step = the stretch between two workgroup barriers; team B runs two steps behind team A):
//   per tile and layer (last layer first):  Ma Mb Fa Fb      Ma / Mb = the two K-halves of the layer's product (chunk 2 li / 2 li + 1)
//                                                            Fa = offset, Householder reflections, mixture + inverse-CDF stage of coordinate 2q
//                                                            Fb = the same for coordinate 2q+1, log-det; after the first layer: stores, then
//                                                                 the next tile's inputs and hidden activations
// At every step exactly one team is in an M step, so one chunk per step streams in (the four waves of the OTHER team issue its DMA pieces one
// step ahead).
//
// Supported: float32, D in {3, 4}, layers with the reference's default options (as cond_split_kernels.hip), H <= 128, K1 <= 32.
#include "jf_cond_regs.h"
#include "jf_mfma.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace jf {

constexpr int PP_TILES = 5;                          // 32-column MFMA tiles per layer
constexpr int PP_KSTEPS = 8;                         // 128 hidden units = 8 x 16
constexpr int PP_HALF = 4;                           // k-steps per chunk
constexpr int PP_NP = 3;                             // bf16 pieces per f32 operand
constexpr int PP_FRAG = 1024;                        // bytes of one A fragment (64 lanes x 8 bf16)
constexpr int PP_CHUNK_BYTES = PP_TILES * PP_HALF * PP_NP * PP_FRAG;      // 61440
constexpr int PP_CHUNK_PIECES = PP_CHUNK_BYTES / 1024;                    // 60 LDS-DMA instructions of 1 KiB
constexpr int PP_BIAS = PP_TILES * 32;               // bias floats per layer (result-row order)
constexpr int PP_ROWS_WAVE = 32, PP_ROWS_TEAM = 4 * PP_ROWS_WAVE, PP_ROWS_WG = 2 * PP_ROWS_TEAM;
constexpr int PP_HMAX = 128, PP_K1MAX = 32;
constexpr int PP_W1_BYTES = 4 * 2 * PP_NP * PP_FRAG; // first layer: 4 hidden tiles x 2 k-steps x 3 pieces
// first layer: weights and bias are multiplied by 2 log2(e) when they are prepared, so that tanh(v) = 1 - 2 / (2^z + 1) with z the MFMA
// result itself (bias = accumulator init): v_exp_f32, add, v_rcp_f32, fma -- the scaling multiply and the bias add of every hidden unit are gone
constexpr float PP_TANH_SCALE = 2.8853900817779268f;
constexpr int PP_LDS_BYTES = 2 * PP_CHUNK_BYTES + PP_W1_BYTES + PP_HMAX * 4 + JF_MAX_CHAIN * PP_BIAS * 4;
static_assert(PP_CHUNK_PIECES % 4 == 0, "a chunk is fetched by four waves");
static_assert(PP_LDS_BYTES <= 160 * 1024, "one workgroup per CU must fit the LDS");

// ---------------------------------------------------------------------------------------------------------- packing
struct PpPackArgs {
    const float* W2; int64_t w2s; const float* b2;
    int H, D, n_layers;
    CsPackLayer L[JF_MAX_CHAIN];
    unsigned char* out;
};

// original column (inside the layer's row) of result row m of column tile `tile`, or -1 (padding / coordinate beyond D)
__device__ __forceinline__ int pp_row_column(const CsPackLayer& o, int D, int tile, int m) {
    const int q = (m >> 2) & 1, reg = 16 * tile + 4 * (m >> 3) + (m & 3);       // the lane group and register the MFMA puts this row in
    if (reg >= 2 * CS_SLOTS) return -1;
    const int c = reg / CS_SLOTS;
    return cs_slot_column(o, D, reg - c * CS_SLOTS, 2 * q + c);
}

// hidden unit that k-slot i of lane group q stands for in k-step s: the order the first layer's result registers come in (32 x 32 result
// tile T = s / 2, registers 8 (s % 2) .. 8 (s % 2) + 7 of lane group q)
__device__ __forceinline__ int pp_hidden_of(int s, int q, int i) { return 32 * (s >> 1) + 16 * (s & 1) + 8 * (i >> 2) + 4 * q + (i & 3); }

// one thread per (chunk, tile, k-step of the chunk, lane): the three pieces' fragments (16 bytes each); chunk = (layer in consumption order, K half)
__global__ void __launch_bounds__(256) pp_pack_kernel(const PpPackArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int sp = rest % PP_HALF; rest /= PP_HALF;
    const int t = rest % PP_TILES; rest /= PP_TILES;
    const int chunk = rest;
    if (chunk >= 2 * a.n_layers) return;
    const int li = chunk >> 1, half = chunk & 1, l = a.n_layers - 1 - li, s = PP_HALF * half + sp;
    const CsPackLayer o = a.L[l];
    const int m = lane & 31, q = lane >> 5;
    const int col = pp_row_column(o, a.D, t, m);
    bf16x8 f[PP_NP];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = pp_hidden_of(s, q, i);
        const float w = (col >= 0 && k < a.H) ? a.W2[(int64_t)(o.col0 + col) * a.w2s + k] : 0.0f;
        __bf16 p0, p1, p2;
        cs_split(w, p0, p1, p2);
        f[0][i] = p0; f[1][i] = p1; f[2][i] = p2;
    }
    unsigned char* base = a.out + (size_t)chunk * PP_CHUNK_BYTES;
#pragma unroll
    for (int p = 0; p < PP_NP; ++p)
        *reinterpret_cast<bf16x8*>(base + (size_t)((t * PP_HALF + sp) * PP_NP + p) * PP_FRAG + lane * 16) = f[p];
    if (sp == 0 && half == 0 && lane < 32) {
        float* bias = reinterpret_cast<float*>(a.out + (size_t)2 * a.n_layers * PP_CHUNK_BYTES);
        bias[li * PP_BIAS + t * 32 + m] = (col >= 0 && a.b2 != nullptr) ? a.b2[o.col0 + col] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------- helpers
// sum over the two lanes of a row (l and l ^ 32): v_permlane32_swap exchanges the upper half of vdst with the lower half of src, so with both
// operands = v the results are "the lower lane's value" and "the upper lane's value" in every lane.  Inline asm: see cond_split_kernels.hip.
__device__ __forceinline__ float pp_xsum(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float pp_xmax(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// tanh(v) from z = 2 log2(e) v:  1 - 2 / (2^z + 1)  (v_exp_f32 / v_rcp_f32; saturates cleanly: 2^z = inf -> 1, 2^z = 0 -> -1)
#ifdef PP_PROBE_P1_NOTANH                                         // timing probes of the first-layer phase (scripts/probe/pp_trace.sh)
__device__ __forceinline__ float pp_tanh_scaled(float z) { return z; }
#else
__device__ __forceinline__ float pp_tanh_scaled(float z) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(z) + 1.0f); }
#endif

// exact 3-way split by truncation of two values at a time into packed bf16 pairs (and / sub / and / sub + one v_perm_b32 per packed pair)
__device__ __forceinline__ void pp_split_pair(float h0, float h1, unsigned& w0, unsigned& w1, unsigned& w2) {
#ifdef PP_PROBE_P1_NOSPLIT
    w0 = __builtin_bit_cast(unsigned, h0); w1 = __builtin_bit_cast(unsigned, h1); w2 = w0 ^ w1; return;
#endif
    const unsigned a0 = __builtin_bit_cast(unsigned, h0), a1 = __builtin_bit_cast(unsigned, h1);
    const float r0 = h0 - __builtin_bit_cast(float, a0 & 0xffff0000u), r1 = h1 - __builtin_bit_cast(float, a1 & 0xffff0000u);
    const unsigned c0 = __builtin_bit_cast(unsigned, r0), c1 = __builtin_bit_cast(unsigned, r1);
    const float s0 = r0 - __builtin_bit_cast(float, c0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, c1 & 0xffff0000u);
    w0 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    w1 = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    w2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}

#ifdef PP_TRACE
// private builds only (scripts/probe/pp_trace.sh): s_memtime at every step boundary of workgroup 0, [wave][step] -> cycles
__device__ long long pp_trace_buf[8 * 512];
#endif

// ---------------------------------------------------------------------------------------------------------- the matrix step
// One K-half of a layer's product: acc[t] += W2-tile(t, k-steps of the half) x h, k-step by k-step.  Inside a k-step the 30 MFMAs (5 tiles x
// 6 piece products) go round the FIVE accumulators, product by product.  Product order (W piece, h piece):
//   (lo,hi) (hi,lo) (mid,mid) | (mid,hi) (hi,mid) | (hi,hi)          -- the three of size 2^-16 first, then 2^-8, then the leading one
// Ten fragment registers (40 VGPRs): slot S[t] holds tile t's lo fragment for round 0, is refilled with the mid fragment right behind that
// MFMA (needed 10 MFMAs later, rounds 2 and 3) and with the NEXT k-step's lo fragment behind round 3; X[t] holds the hi fragment (rounds
// 1, 4, 5) and is refilled for the next k-step behind round 5.  80 accumulators + 96 registers of h + 40 of fragments leave the ~20 the
// rest of the kernel holds across this step; with 15 fragment registers the step reloaded spilled values from scratch memory (a round trip
// of ~1 k cycles each) in the middle of the MFMA stream.
// The waits are exact counts (LDS returns in order).  One asm statement per k-step: left to the scheduler every read was sunk to its use
// (an `s_waitcnt lgkmcnt(0)` in front of each MFMA) whatever the source order, sched_barrier or sched_group_barrier said.
// Fragment (t, sp, p) of a chunk (p: 0 hi, 1 mid, 2 lo) sits at ((t * 4 + sp) * 3 + p) KiB.
template <int HALF, int SP>
__device__ __forceinline__ void pp_kstep(f32x16 (&acc)[PP_TILES], bf16x8 (&S)[PP_TILES], bf16x8 (&X)[PP_TILES], const bf16x8 (&hB)[PP_KSTEPS][PP_NP],
                                         unsigned wbase) {
    constexpr int s = PP_HALF * HALF + SP;
    const unsigned cur = wbase + SP * PP_NP * PP_FRAG;
    if constexpr (SP + 1 < PP_HALF) {
        asm volatile(
            "s_nop 1\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h0], %[c0]\n\t"
            "ds_read_b128 %[s0], %[cur] offset:1024\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h0], %[c1]\n\t"
            "ds_read_b128 %[s1], %[cur] offset:13312\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h0], %[c2]\n\t"
            "ds_read_b128 %[s2], %[cur] offset:25600\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h0], %[c3]\n\t"
            "ds_read_b128 %[s3], %[cur] offset:37888\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h0], %[c4]\n\t"
            "ds_read_b128 %[s4], %[cur] offset:50176\n\t"
            "s_waitcnt lgkmcnt(5)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h2], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h2], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h2], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h2], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h2], %[c4]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h1], %[c0]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h1], %[c1]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h1], %[c2]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h1], %[c3]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h1], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h0], %[c0]\n\t"
            "ds_read_b128 %[s0], %[cur] offset:5120\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h0], %[c1]\n\t"
            "ds_read_b128 %[s1], %[cur] offset:17408\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h0], %[c2]\n\t"
            "ds_read_b128 %[s2], %[cur] offset:29696\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h0], %[c3]\n\t"
            "ds_read_b128 %[s3], %[cur] offset:41984\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h0], %[c4]\n\t"
            "ds_read_b128 %[s4], %[cur] offset:54272\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h1], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h1], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h1], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h1], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h1], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h0], %[c0]\n\t"
            "ds_read_b128 %[x0], %[cur] offset:3072\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h0], %[c1]\n\t"
            "ds_read_b128 %[x1], %[cur] offset:15360\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h0], %[c2]\n\t"
            "ds_read_b128 %[x2], %[cur] offset:27648\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h0], %[c3]\n\t"
            "ds_read_b128 %[x3], %[cur] offset:39936\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h0], %[c4]\n\t"
            "ds_read_b128 %[x4], %[cur] offset:52224"
            : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
              [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [x0] "+v"(X[0]), [x1] "+v"(X[1]), [x2] "+v"(X[2]), [x3] "+v"(X[3]), [x4] "+v"(X[4])
            : [h0] "v"(hB[s][0]), [h1] "v"(hB[s][1]), [h2] "v"(hB[s][2]), [cur] "v"(cur));
        pp_kstep<HALF, SP + 1>(acc, S, X, hB, wbase);
    } else {
        asm volatile(
            "s_nop 1\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h0], %[c0]\n\t"
            "ds_read_b128 %[s0], %[cur] offset:1024\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h0], %[c1]\n\t"
            "ds_read_b128 %[s1], %[cur] offset:13312\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h0], %[c2]\n\t"
            "ds_read_b128 %[s2], %[cur] offset:25600\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h0], %[c3]\n\t"
            "ds_read_b128 %[s3], %[cur] offset:37888\n\t"
            "s_waitcnt lgkmcnt(9)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h0], %[c4]\n\t"
            "ds_read_b128 %[s4], %[cur] offset:50176\n\t"
            "s_waitcnt lgkmcnt(5)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h2], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h2], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h2], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h2], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h2], %[c4]\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h1], %[c0]\n\t"
            "s_waitcnt lgkmcnt(3)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h1], %[c1]\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h1], %[c2]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h1], %[c3]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h1], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[s0], %[h0], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[s1], %[h0], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[s2], %[h0], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[s3], %[h0], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[s4], %[h0], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h1], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h1], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h1], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h1], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h1], %[c4]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c0], %[x0], %[h0], %[c0]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c1], %[x1], %[h0], %[c1]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c2], %[x2], %[h0], %[c2]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c3], %[x3], %[h0], %[c3]\n\t"
            "v_mfma_f32_32x32x16_bf16 %[c4], %[x4], %[h0], %[c4]\n\t"
            "s_nop 15\n\t"
            "s_nop 15"
            : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]),
              [s0] "+v"(S[0]), [s1] "+v"(S[1]), [s2] "+v"(S[2]), [s3] "+v"(S[3]), [s4] "+v"(S[4]), [x0] "+v"(X[0]), [x1] "+v"(X[1]), [x2] "+v"(X[2]), [x3] "+v"(X[3]), [x4] "+v"(X[4])
            : [h0] "v"(hB[s][0]), [h1] "v"(hB[s][1]), [h2] "v"(hB[s][2]), [cur] "v"(cur));
    }
}

// wbase = LDS byte address of the chunk + 16 * lane
template <int HALF>
__device__ __forceinline__ void pp_mstep(f32x16 (&acc)[PP_TILES], const bf16x8 (&hB)[PP_KSTEPS][PP_NP], unsigned wbase) {
    bf16x8 S[PP_TILES], X[PP_TILES];
#pragma unroll
    for (int t = 0; t < PP_TILES; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(S[t]) : "v"(wbase), "n"((t * PP_HALF * PP_NP + 2) * PP_FRAG));
#pragma unroll
    for (int t = 0; t < PP_TILES; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(X[t]) : "v"(wbase), "n"(t * PP_HALF * PP_NP * PP_FRAG));
    pp_kstep<HALF, 0>(acc, S, X, hB, wbase);
}

// ---------------------------------------------------------------------------------------------------------- the fused kernel
struct PpArgs {
    const float* in; int64_t in_stride;
    const float* W1; int64_t w1s; const float* b1;
    const unsigned char* packed;
    int K1, H;
    const float* x; int64_t xs;
    const float* ld_in;
    int64_t B;
    int D, n_layers, n_tiles;
    CsLayer L[JF_MAX_CHAIN];
    float* x_out; int64_t xos;
    float* ld_out;
    const float* blp_in; float* blp_out;
    int32_t* status;
};

__global__ void __launch_bounds__(512, 2) cond_gf_pp_kernel(const PpArgs a) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* W1f = smem + 2 * PP_CHUNK_BYTES;
    const unsigned lds_base = (unsigned)(size_t)(cs_lptr)smem;      // LDS byte address of smem[0] (ds_read in the asm blocks)
    float* b1s = reinterpret_cast<float*>(W1f + PP_W1_BYTES);
    float* bias = b1s + PP_HMAX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: the step dispatch below must be scalar branches, the DMA's LDS address an SGPR
    const int team = wave >> 2, tw = wave & 3;
    const int lj = lane & 31, lq = lane >> 5;
    const int64_t last = a.B - 1;
    const int D = a.D, L = a.n_layers;
    const int my_tiles = ((int)blockIdx.x < a.n_tiles) ? (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int n_ph = 4 * L, my_steps = n_ph * my_tiles, n_steps = my_steps + 2;

    // ---- chunk streaming by LDS-DMA (buffer form: resource + per-lane offset fixed, chunk / piece offset scalar); wave w moves pieces w, w + 8, ...
    const __amdgpu_buffer_rsrc_t packed_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packed), 0, 2 * L * PP_CHUNK_BYTES, 0x00027000);
    // One step ahead of its use a chunk is fetched by the FOUR WAVES OF THE TEAM THAT IS NOT MULTIPLYING in that step (15 pieces of 1 KiB each):
    // issuing an LDS-DMA instruction costs the wave 60-180 cycles, which the flow steps have to spare and the matrix steps do not (with all 8
    // waves issuing, 8 pieces opened every matrix step: 5.0 k cycles per step for 3.9 k of MFMAs).
    auto dma = [&](int chunk, int b) {
        unsigned char* l = smem + b * PP_CHUNK_BYTES;
        const int g = chunk * PP_CHUNK_BYTES;
#pragma unroll
        for (int u = 0; u < PP_CHUNK_PIECES / 4; ++u) {
            const int pc = tw + 4 * u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(packed_rsrc, (cs_lptr)(l + pc * 1024), 16, lane * 16, g + pc * 1024, 0, 0);
        }
    };
    // chunk consumed at global step sg (team A is at its step sg, team B at its step sg - 2; M steps are the steps with (step % 4) < 2), or -1
    auto chunk_for = [&](int sg) {
        int s = sg;
        if (!(s < my_steps && (s & 3) < 2)) s = sg - 2;
        if (s < 0 || s >= my_steps || (s & 3) >= 2) return -1;
        return 2 * ((s % n_ph) >> 2) + (s & 1);
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (my_steps > 0 && team == 1) dma(0, 0);                        // the first chunk: team B (idle in the first two steps)

    // ---- once per workgroup: first-layer weights as split A fragments, its bias, the packed biases of the second layer
    {
        const int T = tid >> 7, s1 = (tid >> 6) & 1, m = lane & 31, q = lane >> 5;     // 512 threads = 4 tiles x 2 k-steps x 64 lanes
        const int hrow = 32 * T + m;
        bf16x8 f[PP_NP];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 16 * s1 + 8 * q + i;
            const float t = a.W1[(int64_t)(hrow < a.H ? hrow : a.H - 1) * a.w1s + (k < a.K1 ? k : 0)];
            const float w = (hrow < a.H && k < a.K1) ? PP_TANH_SCALE * t : 0.f;
            __bf16 p0, p1, p2;
            cs_split(w, p0, p1, p2);
            f[0][i] = p0; f[1][i] = p1; f[2][i] = p2;
        }
#pragma unroll
        for (int p = 0; p < PP_NP; ++p) *reinterpret_cast<bf16x8*>(W1f + ((T * 2 + s1) * PP_NP + p) * PP_FRAG + lane * 16) = f[p];
        if (tid < PP_HMAX) b1s[tid] = tid < a.H ? PP_TANH_SCALE * a.b1[tid < a.H ? tid : 0] : 0.f;
        const float* gb = reinterpret_cast<const float*>(a.packed + (size_t)2 * L * PP_CHUNK_BYTES);
        for (int i = tid; i < L * PP_BIAS; i += 512) bias[i] = gb[i];
    }
    __syncthreads();

    // ---- the team's program.  Loop nest = data lifetimes: the hidden activations live for one tile, the accumulators / parameter registers
    //      for one layer (nothing large is carried around a loop back-edge -- the one-loop-with-step-dispatch form of this schedule made the
    //      register allocator hold old and new copies of both and spill hundreds of registers)
    const bool live0 = 2 * lq < D, live1 = 2 * lq + 1 < D;
    const int d0 = live0 ? 2 * lq : D - 1, d1 = live1 ? 2 * lq + 1 : D - 1;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};       // piece products with pa + pb <= 2, smallest first

    int sg = 2 * team;                                             // global step of the team's next step
    auto step_begin = [&](bool multiplying) {
        const int cn = chunk_for(sg + 1);
        // in flight during this step; its buffer was last read in step sg - 1.  Team B multiplies the chunk team A multiplied two steps
        // earlier, still in the same buffer: only the chunks of team A's matrix steps (global steps 0, 1 mod 4) are fetched
        if (cn >= 0 && !multiplying && ((sg + 1) & 3) < 2) dma(cn, (sg + 1) & 1);
    };
#ifdef PP_TRACE
    int trace_i = 0;
    auto stamp = [&]() { if (blockIdx.x == 0 && lane == 0 && trace_i < 512) pp_trace_buf[wave * 512 + trace_i++] = __builtin_amdgcn_s_memtime(); };
    auto step_end = [&]() { stamp(); landed(); stamp(); ++sg; };
#else
    auto step_end = [&]() { landed(); ++sg; };
#endif
    landed();                                                      // chunk 0 is in buffer 0; W1f / b1s / bias are in place
    if (team == 1) {                                               // team B runs two steps behind: its first two steps are empty
        sg = 0;
        step_begin(false); step_end();
        step_begin(false); step_end();
    }

    // parameter slot `slot` of the lane's coordinate c (compile-time indices after unrolling)
#define PP_P(c, slot) acc[(CS_SLOTS * (c) + (slot)) >> 4][(CS_SLOTS * (c) + (slot)) & 15]

    const int nks1 = a.K1 > 16 ? 2 : 1;
    // per-tile state: carried around the tile loop, (re)defined by begin_tile() -- for tile 0 before the loop, for tile n + 1 at the end of
    // tile n's last flow step, where the team's matrix registers are idle and its partner team keeps the matrix pipe busy
    bf16x8 hB[PP_KSTEPS][PP_NP];                                   // hidden activations of the wave's 32 rows as MFMA B operands
    float xc[2] = {0.f, 0.f}, ld = 0.f;
    int rows_w = 0;
    int64_t row0w = 0;
    auto rsrc_of = [&](const float* base, int64_t stride_elems, int width_elems) {
        const int bytes = rows_w > 0 ? (int)(((int64_t)(rows_w - 1) * stride_elems + width_elems) * 4) : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row0w * stride_elems), 0, bytes, 0x00027000);
    };
#ifdef PP_TRACE
#define PP_STAMP1(i) if (blockIdx.x == 0 && lane == 0 && n == 2) pp_trace_buf[wave * 512 + 496 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define PP_STAMP1(i)
#endif
    // inputs of a tile: loads only (issued BEFORE the previous tile's stores, so that waiting for them does not wait for the stores)
    // (the fetched values are locals of the call site: as state carried around the tile loop they stayed allocated through every step)
    struct TileIn { int rows_n; int64_t row0n; float px0, px1, pld, pin[2][8]; };
    auto load_tile = [&](int n, TileIn& ti) {
        int& rows_n = ti.rows_n; int64_t& row0n = ti.row0n; float& px0 = ti.px0; float& px1 = ti.px1; float& pld = ti.pld;
        float (&pin)[2][8] = ti.pin;
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1)
#pragma unroll
            for (int i = 0; i < 8; ++i) pin[s1][i] = 0.f;
        // the wave's 32 rows through buffer descriptors: scalar 64-bit base (first row of the wave), 32-bit per-lane offsets, and the
        // descriptor's range check stands in for the row < B test (loads past the end return 0, stores are dropped) -- per-lane 64-bit
        // row addresses cost ~20 long-lived registers, which the phase spilled
        row0n = ((int64_t)blockIdx.x + (int64_t)n * gridDim.x) * PP_ROWS_WG + team * PP_ROWS_TEAM + tw * PP_ROWS_WAVE;
        const int64_t left = a.B - row0n;
        rows_n = left <= 0 ? 0 : (left < PP_ROWS_WAVE ? (int)left : PP_ROWS_WAVE);
        auto rsrc_n = [&](const float* base, int64_t stride_elems, int width_elems) {
            const int bytes = rows_n > 0 ? (int)(((int64_t)(rows_n - 1) * stride_elems + width_elems) * 4) : 0;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + row0n * stride_elems), 0, bytes, 0x00027000);
        };
        const __amdgpu_buffer_rsrc_t rs_x = rsrc_n(a.x, a.xs, D), rs_in = rsrc_n(a.in, a.in_stride, a.K1);
        // offsets from an opaque copy of the lane's row: as loop invariants the eleven per-lane offsets were hoisted out of the tile loop,
        // spilled, and reloaded from scratch one by one in front of each load (11 dependent memory round trips, ~4 k cycles per tile)
        int ljv = lj;
        asm volatile("" : "+v"(ljv));
        px0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (ljv * (int)a.xs + d0) * 4, 0, 0));
        px1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (ljv * (int)a.xs + d1) * 4, 0, 0));
        pld = 0.f;
        if (a.ld_in) pld = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_n(a.ld_in, 1, 1), ljv * 4, 0, 0));
        // input k = 16 s1 + 8 lq + i of the row: one per-lane offset, the rest an immediate.  Inputs beyond K1 are read as whatever follows in
        // memory (the next row; zero past the descriptor's end) and masked where they are used
        const int in_off = (ljv * (int)a.in_stride + 8 * lq) * 4;
#pragma unroll
        for (int s1 = 0; s1 < 2; ++s1) {
            if (s1 >= nks1) break;
#pragma unroll
            for (int i = 0; i < 8; ++i) pin[s1][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, in_off + (16 * s1 + i) * 4, 0, 0));
        }
    };
    // first layer of the tile whose inputs load_tile() fetched: h^T = tanh(W1 x^T + b1) on split-bf16 MFMA
    auto begin_tile = [&](int n, const TileIn& ti) {
        PP_STAMP1(0)
        row0w = ti.row0n; rows_w = ti.rows_n;
        xc[0] = ti.px0; xc[1] = ti.px1; ld = ti.pld;
        const float (&pin)[2][8] = ti.pin;
        PP_STAMP1(1)
        {
            // the row's inputs as B operands (k-slot i of lane group lq in k-step s1 = input 16 s1 + 8 lq + i), three bf16 pieces each
            bf16x8 xb[2][PP_NP];
#pragma unroll
            for (int s1 = 0; s1 < 2; ++s1) {
                if (s1 >= nks1) break;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (16 * s1 + 8 * lq + i) < a.K1 ? pin[s1][i] : 0.f;
                u32x4 q0, q1, q2;
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    unsigned w0, w1, w2;
                    pp_split_pair(v[i], v[i + 1], w0, w1, w2);
                    q0[i >> 1] = w0; q1[i >> 1] = w1; q2[i >> 1] = w2;
                }
                xb[s1][0] = __builtin_bit_cast(bf16x8, q0); xb[s1][1] = __builtin_bit_cast(bf16x8, q1); xb[s1][2] = __builtin_bit_cast(bf16x8, q2);
            }
            PP_STAMP1(2)
            // all four 32-unit tiles of hidden units: 24 (48) MFMAs going round four accumulators (an MFMA on the accumulator of the previous
            // one waits for it), started before the tanh arithmetic so that the matrix pipe -- which the partner team keeps busy in this
            // step -- works them in while the vector ALU runs.  Accumulator init = (scaled) bias of the register's hidden unit.
            // h[T][r] = 2 log2(e) x pre-activation of hidden unit 32 T + 8 (r / 4) + 4 lq + r % 4 of row lj; k-slot i of k-step 2 T + u of the
            // second layer <-> register 8 u + i
            f32x16 h[4];
#pragma unroll
            for (int T = 0; T < 4; ++T)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + 32 * T + 8 * g + 4 * lq);
#pragma unroll
                    for (int c = 0; c < 4; ++c) h[T][4 * g + c] = b[c];
                }
#pragma unroll
            for (int s1 = 0; s1 < 2; ++s1) {
                if (s1 >= nks1) break;
                bf16x8 A[4][PP_NP];
#pragma unroll
                for (int T = 0; T < 4; ++T)
#pragma unroll
                    for (int p = 0; p < PP_NP; ++p) A[T][p] = *reinterpret_cast<const bf16x8*>(W1f + ((T * 2 + s1) * PP_NP + p) * PP_FRAG + lane * 16);
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int T = 0; T < 4; ++T) h[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[T][PA[i]], xb[s1][PB[i]], h[T], 0, 0, 0);
            }
            PP_STAMP1(3)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    u32x4 q0, q1, q2;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const int r = 8 * u + i;
                        unsigned w0, w1, w2;
                        pp_split_pair(pp_tanh_scaled(h[T][r]), pp_tanh_scaled(h[T][r + 1]), w0, w1, w2);
                        q0[i >> 1] = w0; q1[i >> 1] = w1; q2[i >> 1] = w2;
                    }
                    hB[2 * T + u][0] = __builtin_bit_cast(bf16x8, q0); hB[2 * T + u][1] = __builtin_bit_cast(bf16x8, q1);
                    hB[2 * T + u][2] = __builtin_bit_cast(bf16x8, q2);
                }
                PP_STAMP1(4 + T)
            }
        }
    };
    if (my_tiles > 0) { TileIn ti; load_tile(0, ti); begin_tile(0, ti); }

    for (int n = 0; n < my_tiles; ++n) {
        const bool row_valid = lj < rows_w;
        step_begin(true);
        for (int li = 0; li < L; ++li) {
            const int l = L - 1 - li;
            const CsLayer o = a.L[l];                              // uniform index: scalar loads from the kernarg segment
            f32x16 acc[PP_TILES];                                  // accumulators of the layer = its parameter registers afterwards
            // ---- Ma: bias, first K half
            if (li > 0) step_begin(true);
#pragma unroll
            for (int t = 0; t < PP_TILES; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + li * PP_BIAS + t * 32 + 8 * g + 4 * lq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
                }
            pp_mstep<0>(acc, hB, lds_base + lane * 16);
            step_end();
            // ---- Mb: second K half
            step_begin(true);
            pp_mstep<1>(acc, hB, lds_base + PP_CHUNK_BYTES + lane * 16);
            step_end();
            // ---- Fa: offset, reflections (both coordinates), mixture + inverse-CDF stage of coordinate 2 lq
            step_begin(false);
            float g0 = xc[0] - PP_P(0, CS_SLOT_OFF), g1 = xc[1] - PP_P(1, CS_SLOT_OFF);     // euclidean_base.py:40-45
#pragma unroll
            for (int i = 0; i < CS_HH; ++i) {
                if (i < o.hh) {                                    // x <- Q^T x (gaussianization_flow.py:1038), H_i = I - 2 v v^T / |v|^2
                    const float v0 = live0 ? PP_P(0, CS_SLOT_ROT + i) : 0.f, v1 = live1 ? PP_P(1, CS_SLOT_ROT + i) : 0.f;
                    const float n2 = pp_xsum(v0 * v0 + v1 * v1), dot = pp_xsum(v0 * g0 + v1 * g1);
                    const float f = 2.0f * dot * M<float>::rcp(n2);
                    g0 -= f * v0; g1 -= f * v1;
                }
            }
            float y0, logd0;
#ifdef PP_PROBE_NO_FLOW                                            // timing probe only (scripts/probe/pp_trace.sh): the M steps without a busy partner
            y0 = g0 + PP_P(0, 3); logd0 = PP_P(0, 17);
#else
            {
                float R[CS_SLOTS];
#pragma unroll
                for (int k = 0; k < CS_SLOTS; ++k) R[k] = PP_P(0, k);
                const MixQ<float> q = cs_mixture(R, o, g0, live0);
                const IcdfOut<float> sy = gf_icdf<float>(o.inv_type, q);
                y0 = sy.y; logd0 = sy.logd;
            }
#endif
            step_end();
            // ---- Fb: coordinate 2 lq + 1, log-det
            step_begin(false);
#ifdef PP_PROBE_NO_FLOW
            xc[0] = y0; xc[1] = g1 + PP_P(1, 5); ld += logd0 + PP_P(1, 20);
#else
            {
                float R[CS_SLOTS];
#pragma unroll
                for (int k = 0; k < CS_SLOTS; ++k) R[k] = PP_P(1, k);
                const MixQ<float> q = cs_mixture(R, o, g1, live1);
                const IcdfOut<float> sy = gf_icdf<float>(o.inv_type, q);
                xc[0] = y0; xc[1] = sy.y;
                ld += pp_xsum((live0 ? logd0 : 0.f) + (live1 ? sy.logd : 0.f));
            }
#endif
            if (l == 0) {                                          // the tile is done
                TileIn ti;
                if (n + 1 < my_tiles) load_tile(n + 1, ti);
                const __amdgpu_buffer_rsrc_t rs_o = rsrc_of(a.x_out, a.xos, D);        // rows past B: dropped by the range check
                // unconditional stores (a lane without a second coordinate repeats its first one: d1 = d0 there; both lanes of a row write the
                // row's log-det): with stores under per-lane branches the wait for the next tile's loads became a wait for everything
                // (live0 holds for every lane when D >= 3; d1 == d0 where live1 does not)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, xc[0]), rs_o, (lj * (int)a.xos + d0) * 4, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, live1 ? xc[1] : xc[0]), rs_o, (lj * (int)a.xos + d1) * 4, 0, 0);
                float sb = 0.f;
                if (a.blp_out) sb = pp_xsum((live0 ? -0.5f * xc[0] * xc[0] - M<float>::HALF_LN_2PI : 0.f) +
                                            (live1 ? -0.5f * xc[1] * xc[1] - M<float>::HALF_LN_2PI : 0.f));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ld), rsrc_of(a.ld_out, 1, 1), lj * 4, 0, 0);
                if (a.blp_out) {
                    float bi = 0.f;
                    if (a.blp_in) bi = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_of(a.blp_in, 1, 1), lj * 4, 0, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, sb + bi), rsrc_of(a.blp_out, 1, 1), lj * 4, 0, 0);
                }
                const float bad = pp_xmax(((live0 && !M<float>::finite(xc[0])) || (live1 && !M<float>::finite(xc[1]))) ? 1.f : 0.f);
                status_add(a.status, JF_STATUS_NONFINITE, row_valid && lq == 0 && (bad > 0.f || !M<float>::finite(ld)));
                if (n + 1 < my_tiles) begin_tile(n + 1, ti);
            }
            step_end();
        }
    }
    if (team == 0 && my_steps > 0) {                               // team A is done two steps before team B
        step_begin(false); step_end();
        step_begin(false); step_end();
    }
#undef PP_P
}

// ---------------------------------------------------------------------------------------------------------- host side
static bool pp_layer_supported(const jf_gf_layer& h, int D) {
    return h.num_kde == CS_K && h.hh_iter >= 0 && h.hh_iter <= CS_HH && h.nonlinear_stretch_type == JF_GF_STRETCH_CLASSIC &&
           h.rotation_mode == JF_GF_ROT_HOUSEHOLDER && !h.center_mean && !h.add_skewness &&
           h.width_mode == JF_GF_WIDTH_SMOOTH_SATURATION && !h.clamp_widths && h.fit_normalization && h.regulate_normalization &&
           h.width_min > 0 && h.width_max > 0 && D >= 3 && D <= 4;
}

static int64_t pp_packed_bytes(int n_layers) { return (int64_t)n_layers * (2 * PP_CHUNK_BYTES + PP_BIAS * 4); }

static int pp_pack(const float* W2, int64_t w2s, const float* b2, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers, void* packed,
                   void* stream) {
    if (!W2 || !layers || !packed) return JF_ERR_BADARG;
    if (!width_ok(H) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (H > PP_HMAX) return JF_ERR_UNSUPPORTED;
    PpPackArgs a{};
    int col = 0;
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        if (!pp_layer_supported(h, D)) return JF_ERR_UNSUPPORTED;
        CsPackLayer& o = a.L[l];
        const int kd = h.num_kde * D;
        o.col0 = col; o.hh = h.hh_iter; o.model_offset = h.model_offset;
        o.off_rot = h.model_offset ? D : 0;
        o.off_mean = o.off_rot + h.hh_iter * D;
        o.off_lw = o.off_mean + kd;
        o.off_ln = o.off_lw + kd;
        col += o.off_ln + kd;
    }
    a.W2 = W2; a.w2s = w2s; a.b2 = b2; a.H = H; a.D = D; a.n_layers = n_layers; a.out = static_cast<unsigned char*>(packed);
    const int threads = 2 * n_layers * PP_TILES * PP_HALF * 64;
    jf::launch(pp_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

static int pp_chain(const float* in, int64_t in_stride, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1, int32_t H,
                    const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n_layers, const jf_gf_layer* layers, float* x_out,
                    int64_t xos, float* ld_out, const float* blp_in, float* blp_out, int32_t* status, void* stream) {
    if (!in || !W1 || !b1 || !packed || !x || !x_out || !ld_out || !layers) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (K1 > PP_K1MAX || H > PP_HMAX || (reinterpret_cast<uintptr_t>(packed) & 15u)) return JF_ERR_UNSUPPORTED;
    if (B > ((int64_t)1 << 31) * PP_ROWS_WG / 4) return JF_ERR_UNSUPPORTED;
    PpArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        if (!pp_layer_supported(h, D)) return JF_ERR_UNSUPPORTED;
        CsLayer& o = a.L[l];
        o.hh = h.hh_iter; o.model_offset = h.model_offset; o.inv_type = h.inverse_function_type;
        o.wmin = (float)h.width_min; o.inv_wmax = (float)(1.0 / h.width_max); o.nmin = (float)h.norm_min; o.nmax = (float)h.norm_max;
    }
    if (B == 0) return JF_OK;
    a.in = in; a.in_stride = in_stride; a.W1 = W1; a.w1s = w1s; a.b1 = b1; a.packed = static_cast<const unsigned char*>(packed); a.K1 = K1; a.H = H;
    a.x = x; a.xs = xs; a.ld_in = ld_in; a.B = B; a.D = D; a.n_layers = n_layers;
    a.x_out = x_out; a.xos = xos; a.ld_out = ld_out; a.blp_in = blp_in; a.blp_out = blp_out; a.status = status;
    a.n_tiles = (int)((B + PP_ROWS_WG - 1) / PP_ROWS_WG);
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return JF_ERR_LAUNCH;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        (void)hipFuncSetAttribute((const void*)cond_gf_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
    }
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;          // persistent: one workgroup per CU (the LDS admits no second one)
    jf::launch(cond_gf_pp_kernel, dim3((unsigned)grid), dim3(512), PP_LDS_BYTES, (hipStream_t)stream, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

}  // namespace jf

extern "C" {
#ifdef PP_TRACE
int jf_pp_trace_read(long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(jf::pp_trace_buf), sizeof(long long) * n) == hipSuccess ? 0 : -1;
}
#endif
int64_t jf_cond_gf_pp_packed_bytes(int32_t D, int32_t n_layers, const jf_gf_layer* layers) {
    if (!layers || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    for (int l = 0; l < n_layers; ++l)
        if (!jf::pp_layer_supported(layers[l], D)) return JF_ERR_UNSUPPORTED;
    return jf::pp_packed_bytes(n_layers);
}
int jf_cond_gf_pp_pack_f32(const float* W2, int64_t w2s, const float* b2, int32_t H, int32_t D, int32_t n, const jf_gf_layer* L, void* packed, void* s) {
    return jf::pp_pack(W2, w2s, b2, H, D, n, L, packed, s);
}
int jf_cond_gf_chain_inv_pp_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const void* packed, int32_t K1,
                                int32_t H, const float* x, int64_t xs, const float* ld_in, int64_t B, int32_t D, int32_t n,
                                const jf_gf_layer* L, float* xo, int64_t xos, float* ldo, const float* bi, float* bo, int32_t* st, void* s) {
    return jf::pp_chain(in, is, W1, w1s, b1, packed, K1, H, x, xs, ld_in, B, D, n, L, xo, xos, ldo, bi, bo, st, s);
}
}
