// Adjoint of the conditional e-block in ONE launch (float32, the shapes cond_split_kernels.hip takes): what torch.autograd replays for
// mlp_predictors[i](...) + the block's layer loop of all_layer_inverse (main/default.py:656-670, 946-962, 998-1031; gaussianization_flow.py:
// 995-1114; euclidean_base.py:34-51) when the loss depends on (x_out, log_det_out, base_logp_out) -- the training step of
// examples/jammy_flows.py:381-412.
//
//   jf_cond_gf_bwd_packed_bytes / jf_cond_gf_bwd_pack_f32   W2^T as MFMA A-fragments (M = hidden unit, K = parameter slot of a coordinate)
//   jf_cond_gf_chain_inv_split_bwd_f32                      g_x (B, D), g_h (B, H), the activations h (B, H) and the parameter-row gradient in
//                                                           PACKED column order (B, n_layers * 144) for the two batch-reducing products
//                                                           (g_W2, g_b2: jf_linear_wgrad_split on the packed rows; the caller gathers the
//                                                           548 rows of W2 out of the 576 packed ones)
//
// Round 2 ran this block's backward as six launches around a materialised (B, 548) parameter block: h = tanh(..) (jf_linear), the block again
// (jf_linear_split), the chain's adjoint with the block staged through LDS (jf_gf_chain_inv_bwd, itself re-running the chain), g_h = g_params W2
// (jf_linear_split), the weight gradient, the hidden layer's adjoint: 1.22 of the 2.47 ms of a C3 training step at 2^18 rows.  Here
//  * the forward launch of a training step (cond_gf_split_kernel<.., SAVE>) leaves every layer's input coordinate and mixture sums behind:
//    20 floats per row and layer;
//  * this kernel recomputes h and each layer's parameters exactly as the forward kernel does (f16-pair MFMA, result layout = flow layout),
//    so the parameter row of lane (row, coordinate) is in 36 registers when the layer's adjoint starts, and the adjoint OVERWRITES it in
//    place with the gradient of those 36 raw parameters;
//  * in that layout the gradient row is already an MFMA B operand: lane (row n, coordinate q) supplies k-slots 8 q .. 8 q + 7 of k-step s
//    = its own slots 8 s .. 8 s + 7.  g_h^T (hidden x rows) += W2_l^T (hidden x slots) g_P^T (slots x rows) therefore needs no transpose and
//    no LDS: 5 k-steps x 8 hidden tiles x 3 piece products per layer, accumulated over the layers in 32 registers whose layout is the one
//    phase 1 produced h in.  Arithmetic: f16 pairs as in the forward kernel (jf_cond_split.h); W2^T carries the forward image's power-of-two
//    scale, a row's gradient its own (largest entry of the row and layer into [2^14, 2^15)), and the accumulator is moved into and out of
//    the layer's units around each layer's products (exact);
//  * the layers run first to last (the reverse of the log-prob direction), the gradient of a layer's input coordinate feeding the next.
#include "jf_cond_split.h"
#include "jf_gf_bwd.h"

namespace jf {

constexpr int CB_KSTEPS = 5;                        // 36 slots (+ 4 of padding) = 5 x 8 k-slots per coordinate lane
constexpr int CB_JT = CS_HMAX / 16;                 // hidden tiles
constexpr int CB_T_BYTES = CB_JT * CS_NP16 * CS_FRAG; // 16384: one k-step of W2_l^T (f16 pairs)
constexpr int CB_STEPS = CS_CPL + CB_KSTEPS;        // chunk steps per layer (even: a layer starts in buffer 0)
constexpr int CB_PROW = 4 * CS_SLOTS;               // packed gradient columns per layer
constexpr int CB_BUF = CS_CHUNK16_BYTES;           // one of the two chunk buffers
static_assert(CB_STEPS % 2 == 0 && CB_T_BYTES <= CB_BUF, "double buffer of the forward kernel");

// ---------------------------------------------------------------------------------------------------------- packing
struct CbPackArgs {
    const float* W2; int64_t w2s;
    int H, D, n_layers, N;
    CsPackLayer L[JF_MAX_CHAIN];
    unsigned char* out;
};

// one thread per (layer, k-step, hidden tile, lane): fragment value i of lane (m, q) = W2[column of slot 8 s + i of coordinate q][16 j + m]
__global__ void __launch_bounds__(256) cb_pack_kernel(const CbPackArgs a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int j = rest % CB_JT; rest /= CB_JT;
    const int s = rest % CB_KSTEPS; rest /= CB_KSTEPS;
    const int l = rest;
    if (l >= a.n_layers) return;
    const CsPackLayer o = a.L[l];
    const int m = lane & 15, q = lane >> 4;
    const int k = 16 * j + m;
    float* tail = reinterpret_cast<float*>(a.out + (size_t)a.n_layers * CB_KSTEPS * CB_T_BYTES);
    const float wmax = tail[0];
    const int e = cs_w_exponent(wmax);                               // the forward image's scale
    const float wscale = ldexpf(1.0f, e);
    bf16x8 f[CS_NP16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int slot = 8 * s + i;
        const int col = slot < CS_SLOTS ? cs_slot_column(o, a.D, slot, q) : -1;
        const float w = (col >= 0 && k < a.H) ? a.W2[(int64_t)(o.col0 + col) * a.w2s + k] : 0.0f;
        const float ws = w * wscale;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        f[0][i] = __builtin_bit_cast(__bf16, hi); f[1][i] = __builtin_bit_cast(__bf16, lo);
    }
    unsigned char* base = a.out + (size_t)(l * CB_KSTEPS + s) * CB_T_BYTES;
#pragma unroll
    for (int p = 0; p < CS_NP16; ++p) *reinterpret_cast<bf16x8*>(base + (size_t)(j * CS_NP16 + p) * CS_FRAG + lane * 16) = f[p];
    if (idx == 0) tail[1] = (float)e;
}

// ---------------------------------------------------------------------------------------------------------- one layer's adjoint on a register row
// P: raw parameters of the lane's coordinate (slot order of jf_cond_regs.h) -> gradient of the loss with respect to them (zero in unused slots
// and in lanes without a coordinate).  Returns the gradient of the layer's input coordinate.  gy: gradient of the layer's output coordinate,
// gl: gradient of log_det.  The arithmetic of gf_layer_bwd_fast / gf_layer_bwd (gf_bwd_kernels.hip) for the options the split kernels take.
__device__ __forceinline__ float cb_layer_bwd(float (&P)[CS_SLOTS], const CsLayer& o, bool live, float x_in, const CsSums& m, float gy, float gl) {
    using Mf = M<float>;
    float xr[CS_HH];
    float x = x_in - P[CS_SLOT_OFF];
#pragma unroll
    for (int i = 0; i < CS_HH; ++i) {
        xr[i] = x;
        if (i < o.hh) {
            const float v = live ? P[CS_SLOT_ROT + i] : 0.f;
            const float n2 = cs_rsum(v * v), dot = cs_rsum(v * x);
            x -= 2.0f * dot * Mf::rcp(n2) * v;
        }
    }
    const bool ok = m.C > LinRange<float>::lo && m.S > LinRange<float>::lo && m.P > LinRange<float>::lo && m.P < LinRange<float>::hi;
    float gx = 0.f;
    if (__all(ok || !live)) {
        MixQ<float> q;
        q.lc = Mf::log_fast(m.C); q.ls = Mf::log_fast(m.S); q.lp = Mf::log_fast(m.P); q.cdf = m.C; q.sf = m.S;
        const IcdfOut<float> s = gf_icdf<float>(o.inv_type, q);
        const IcdfCoef<float> c = gf_icdf_coeffs<float>(o.inv_type, q, s.y);
        const float g_lc = gy * c.Ay + gl * c.AH, g_ls = gy * c.By + gl * c.BH, g_lp = gl;
        const float Gsum = g_lc + g_ls + g_lp;
        const float icg = g_lc * Mf::rcp(m.C), isg = g_ls * Mf::rcp(m.S), ipg = g_lp * Mf::rcp(m.P);
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const float mu = P[CS_SLOT_MEAN + k], rw = P[CS_SLOT_LW + k], rn = P[CS_SLOT_LN + k];
            const float e = Mf::exp_fast(-rw);
            const float ae = o.inv_wmax + e;
            const float r2 = Mf::rcp(ae * (o.wmin * ae + 1.0f));
            const float iw = ae * ae * r2, dliw = -e * r2;           // 1 / w,  d log(1 / w) / d raw
            const float sgn = Mf::rcp(1.0f + Mf::exp_fast(-rn));
            const float pik = (o.nmin + o.nmax * sgn) * m.invN;
            const float u = (x - mu) * iw;
            const float t = Mf::exp_fast(-fabsf(u));
            const float hi = Mf::rcp(1.0f + t), lo = t * hi;
            const bool pos = u >= 0.f;
            const float sg = pos ? hi : lo, sgc = pos ? lo : hi;
            const float a = sg * icg, b = sgc * isg, cp = sg * sgc * iw * ipg;           // g . responsibility / pi_k
            const float gu = pik * (a * sgc - b * sg + cp * (sgc - sg));
            gx += gu * iw;
            P[CS_SLOT_MEAN + k] = live ? -gu * iw : 0.f;
            P[CS_SLOT_LW + k] = live ? (gu * u + pik * cp) * dliw : 0.f;
            P[CS_SLOT_LN + k] = live ? (a + b + cp - Gsum) * (o.nmax * sgn * (1.0f - sgn) * m.invN) : 0.f;
        }
    } else {
        // Some row of the wave sits where the linear-space sums under- or overflow.  Rounds 2-3 took the responsibilities in log space here
        // (gf_layer_bwd: 8 exp / log and 8 divisions per component, after a full cs_mixture); on the SURVEY inputs, two thirds of whose rows
        // sit beyond 12 sigma after three layers, nearly every wave holds such a row.  Sums scaled by e^{m}, m = distance to the nearest
        // component, are enough (gf_layer_bwd_bcast, gf_bwd_kernels.hip, has the algebra): every ratio has e^{-m} on both sides.
        float iw[CS_K], u[CS_K], sgn[CS_K];
        float mm = INFINITY;
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const float e = Mf::exp_fast(-P[CS_SLOT_LW + k]);
            const float ae = o.inv_wmax + e;
            iw[k] = ae * Mf::rcp(o.wmin * ae + 1.0f);
            sgn[k] = Mf::rcp(1.0f + Mf::exp_fast(-P[CS_SLOT_LN + k]));
            u[k] = (x - P[CS_SLOT_MEAN + k]) * iw[k];
            mm = fminf(mm, fabsf(u[k]));
        }
        const float em = Mf::exp_fast(-mm);                        // may underflow to 0: the unscaled parts then stand alone
        float Cu = 0.f, Cq = 0.f, Su = 0.f, Sq = 0.f, Pq = 0.f;
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const float pk = (o.nmin + o.nmax * sgn[k]) * m.invN;
            const float tp = Mf::exp_fast(mm - fabsf(u[k]));
            const float h = Mf::rcp(1.0f + tp * em);
            const float c1 = pk * h, c2 = c1 * tp;
            if (u[k] >= 0.f) { Cu += c1; Sq += c2; } else { Su += c1; Cq += c2; }
            Pq += c2 * h * iw[k];
        }
        MixQ<float> q;
        q.cdf = Cu + em * Cq;
        q.sf = Su + em * Sq;
        q.lc = Cu > 0.f ? Mf::log_fast(q.cdf) : Mf::log_fast(Cq) - mm;
        q.ls = Su > 0.f ? Mf::log_fast(q.sf) : Mf::log_fast(Sq) - mm;
        q.lp = Mf::log_fast(Pq) - mm;
        const IcdfOut<float> s = gf_icdf<float>(o.inv_type, q);
        const IcdfCoef<float> c = gf_icdf_coeffs<float>(o.inv_type, q, s.y);
        const float g_lc = gy * c.Ay + gl * c.AH, g_ls = gy * c.By + gl * c.BH, g_lp = gl;
        const float Gsum = g_lc + g_ls + g_lp;
        const float r1c = Cu > 0.f ? Mf::rcp(q.cdf) : 0.f, r1s = Su > 0.f ? Mf::rcp(q.sf) : 0.f;      // 1 / cdf, 1 / sf: components on their side
        const float a2c = Cu > 0.f ? em * r1c : Mf::rcp(Cq), a2s = Su > 0.f ? em * r1s : Mf::rcp(Sq);  // e^{-m} / cdf, e^{-m} / sf
        const float ap = Mf::rcp(Pq);
#pragma unroll
        for (int k = 0; k < CS_K; ++k) {
            const float e = Mf::exp_fast(-P[CS_SLOT_LW + k]);
            const float ae = o.inv_wmax + e;
            const float dliw = -e * Mf::rcp(ae * (o.wmin * ae + 1.0f));
            const float pk = (o.nmin + o.nmax * sgn[k]) * m.invN;
            const float tp = Mf::exp_fast(mm - fabsf(u[k]));
            const float h = Mf::rcp(1.0f + tp * em);
            const bool pos = u[k] >= 0.f;
            const float th = tp * h, w2 = th * h;                  // s (1 - s) = em w2
            const float sk = pos ? h : em * th;                    // sigma(u)
            const float pp = w2 * iw[k] * ap;                      // s (1 - s) / (w pdf)
            const float sc = pos ? h * r1c : th * a2c;             // s / cdf
            const float ss = pos ? th * a2s : h * r1s;             // (1 - s) / sf
            const float gu = pk * (w2 * (g_lc * a2c - g_ls * a2s) + g_lp * pp * (1.0f - 2.0f * sk));
            gx += gu * iw[k];
            P[CS_SLOT_MEAN + k] = live ? -gu * iw[k] : 0.f;
            P[CS_SLOT_LW + k] = live ? (gu * u[k] + g_lp * pk * pp) * dliw : 0.f;
            P[CS_SLOT_LN + k] = live ? (g_lc * sc + g_ls * ss + g_lp * pp - Gsum) * (o.nmax * sgn[k] * (1.0f - sgn[k]) * m.invN) : 0.f;
        }
    }
    // reflections, last first:  y = x - c v, c = 2 (v.x)/(v.v):  g_x = H g,  g_v = -c g - (2 (v.g)/n) x + (4 (v.x)(v.g)/n^2) v
    float g = live ? gx : 0.f;
#pragma unroll
    for (int i = CS_HH - 1; i >= 0; --i) {
        if (i < o.hh) {
            const float v = live ? P[CS_SLOT_ROT + i] : 0.f;
            const float n = cs_rsum(v * v), sx = cs_rsum(v * xr[i]), vg = cs_rsum(v * g);
            const float rn = Mf::rcp(n);
            P[CS_SLOT_ROT + i] = live ? -2.0f * sx * rn * g - 2.0f * vg * rn * xr[i] + 4.0f * sx * vg * rn * rn * v : 0.f;
            g -= 2.0f * vg * rn * v;
        } else {
            P[CS_SLOT_ROT + i] = 0.f;
        }
    }
    P[CS_SLOT_OFF] = (o.model_offset && live) ? -g : 0.f;
    P[CS_SLOTS - 1] = 0.f;
    return g;
}

// ---------------------------------------------------------------------------------------------------------- the kernel
struct CbArgs {
    const float* in; int64_t in_stride;
    const float* W1; int64_t w1s; const float* b1;
    const unsigned char* packed;               // forward image (jf_cond_gf_pack_f32)
    const unsigned char* packedT;              // W2^T image (jf_cond_gf_bwd_pack_f32)
    int K1, H;
    const float* z; int64_t zs;                // the block's output coordinates (x_out of the forward launch)
    const float* aux;                          // what the forward launch saved
    int64_t B;
    int D, n_layers;
    CsLayer L[JF_MAX_CHAIN];
    const float* g_xout; int64_t gxos;
    const float* g_ld; const float* g_blp;
    float* g_x; int64_t gxs;
    float* g_pp; int64_t gpps;                 // (B, n_layers * 144): [layer][coordinate lane][slot]
    float* h_out; int64_t hs;                  // (B, H)
    float* g_h; int64_t ghs;                   // (B, H)
    float* g_absmax;                           // optional: max |g_pp| over the launch (atomic max of its bit pattern; the caller zeroes it)
};

__global__ void __launch_bounds__(256, 2) cond_gf_split_bwd_kernel(const CbArgs a) {
    constexpr int MT = 16;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned char* Ws0 = smem_raw;                                 // two chunk buffers
    float* Xs = reinterpret_cast<float*>(smem_raw + CB_BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * CS_ROWS1;
    const int64_t last = a.B - 1;
    const int D = a.D;
    const __amdgpu_buffer_rsrc_t p_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packed), 0, a.n_layers * CS_CPL * CS_CHUNK16_BYTES, 0x00027000);
    const __amdgpu_buffer_rsrc_t t_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packedT), 0, a.n_layers * CB_KSTEPS * CB_T_BYTES, 0x00027000);
    const int lane_off = wave * 1024 + lane * 16;
    // step c of layer l: c < 3 parameter chunk c of the layer (forward image, stored last layer first), else k-step c - 3 of W2_l^T
    auto dma = [&](int l, int c, int buf) {
        unsigned char* dst = Ws0 + buf * CB_BUF;
        if (c < CS_CPL) cs_dma_chunk(p_rsrc, dst, ((a.n_layers - 1 - l) * CS_CPL + c) * CS_CHUNK16_BYTES, lane_off, wave, lane, CS_W16_BYTES, CS_B16_BYTES);
        else cs_dma_chunk(t_rsrc, dst, (l * CB_KSTEPS + c - CS_CPL) * CB_T_BYTES, lane_off, wave, lane, CB_T_BYTES, 0);
    };
    dma(0, 0, 0);                                                  // lands in buffer 0 while phase 1 works in buffer 1

    const int w_exp = (int)reinterpret_cast<const float*>(a.packedT + (size_t)a.n_layers * CB_KSTEPS * CB_T_BYTES)[1];   // W2^T's scale 2^w_exp
    bf16x8 hB[1][CS_KSTEPS][CS_NP16];
    cs_hidden<1, true, CS_NP16>(a.in, a.in_stride, a.W1, a.w1s, a.b1, a.K1, a.H, row0, last, Xs, hB, a.h_out, a.hs);

    // ---- flow state: lane = (row li of the wave's 16, coordinate lq)
    const bool live = lq < D;
    const int d = live ? lq : D - 1;
    const int64_t row = row0 + wave * MT + li;
    const bool row_valid = row <= last;
    const int64_t rrow = row_valid ? row : last;
    const float gl = a.g_ld ? a.g_ld[rrow] : 0.f;
    float gy = a.g_xout ? a.g_xout[rrow * a.gxos + d] : 0.f;
    if (a.g_blp) gy -= a.g_blp[rrow] * a.z[rrow * a.zs + d];        // base_logp_out = base_logp_in + sum_d (-z_d^2 / 2 - ln sqrt(2 pi))
    if (!live) gy = 0.f;
    f32x4 gh[CB_JT];                                               // g_h^T: register r of tile j = hidden unit 16 j + 4 lq + r of row li
#pragma unroll
    for (int j = 0; j < CB_JT; ++j) gh[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                        // lo x hi, hi x lo, hi x hi

    landed();                                                      // chunk 0 is in buffer 0 and every wave is done with the phase-1 scratch
    for (int l = 0; l < a.n_layers; ++l) {
        // what the forward launch kept of this layer for the lane
        const int64_t slot = ((int64_t)l * a.B + rrow) * 4 + lq;
        const f32x4 sv = reinterpret_cast<const f32x4*>(a.aux)[slot];
        const float x_in = a.aux[(int64_t)a.n_layers * a.B * 16 + slot];
        float P[CS_SLOTS];
#pragma unroll
        for (int c = 0; c < CS_CPL; ++c) {
            dma(l, c + 1, (c + 1) & 1);                            // in flight while this chunk is multiplied
            const unsigned char* Ws = Ws0 + (c & 1) * CB_BUF;
            const float* Bs = reinterpret_cast<const float*>(Ws + CS_W16_BYTES);
            f32x4 acc[CS_CT];
#pragma unroll
            for (int t = 0; t < CS_CT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(Bs + t * 16 + 4 * lq);      // bias in the accumulators' units
            bf16x8 A[2][CS_CT][CS_NP16];
            auto load_a = [&](int s, int buf) {
#pragma unroll
                for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                    for (int p = 0; p < CS_NP16; ++p)
                        A[buf][t][p] = *reinterpret_cast<const bf16x8*>(Ws + ((t * CS_KSTEPS + s) * CS_NP16 + p) * CS_FRAG + lane * 16);
            };
            load_a(0, 0);
#pragma unroll
            for (int s = 0; s < CS_KSTEPS; ++s) {
                const int b = s & 1;
                if (s + 1 < CS_KSTEPS) load_a(s + 1, b ^ 1);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int t = 0; t < CS_CT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[b][t][PA[i]]), __builtin_bit_cast(f16x8, hB[0][s][PB[i]]),
                                                                        acc[t], 0, 0, 0);
            }
            const float inv = Bs[CS_B_BYTES / 4];                    // 2^-(e + 14)
#pragma unroll
            for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[4 * (c * CS_CT + t) + r] = acc[t][r] * inv;
            if (c + 1 < CS_CPL) landed();
        }
        // ---- the layer's adjoint: P becomes the gradient row
        const CsLayer o = a.L[l];
        gy = cb_layer_bwd(P, o, live, x_in, CsSums{sv[0], sv[1], sv[2], sv[3]}, gy, gl);
        if (!live) gy = 0.f;
        if (row_valid) {
            float* dst = a.g_pp + row * a.gpps + l * CB_PROW + lq * CS_SLOTS;
#pragma unroll
            for (int t = 0; t < CS_TILES; ++t) *reinterpret_cast<f32x4*>(dst + 4 * t) = f32x4{P[4 * t], P[4 * t + 1], P[4 * t + 2], P[4 * t + 3]};
        }
        // the row's scale for this layer: largest gradient entry into [2^14, 2^15); the accumulator goes into the products' units
        float gmax = 0.f;
#pragma unroll
        for (int k = 0; k < CS_SLOTS; ++k) gmax = fmaxf(gmax, fabsf(P[k]));
        gmax = cs_rmax(gmax);
        if (a.g_absmax != nullptr) {                               // the weight-gradient product on f16 pairs scales the packed rows by one power of two
            float wm = row_valid ? gmax : 0.f;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
            // (a plain read first: after the first waves almost none raises the maximum, and 65 K atomics on one address cost 0.3 ms)
            if (lane == 0 && wm < INFINITY && __builtin_bit_cast(unsigned, wm) > __atomic_load_n(reinterpret_cast<unsigned*>(a.g_absmax), __ATOMIC_RELAXED))
                atomicMax(reinterpret_cast<unsigned*>(a.g_absmax), __builtin_bit_cast(unsigned, wm));
        }
        // (exponents clamped to +-60 like W2's: the two together stay inside the f32 exponent range, and an entry 2^-60 below that is noise)
        const int g_exp = (gmax > 0.f && gmax < INFINITY) ? max(-60, min(60, 14 - ((int)((__builtin_bit_cast(unsigned, gmax) >> 23) & 0xff) - 127))) : 0;
        const float g_scale = __builtin_bit_cast(float, (unsigned)(127 + g_exp) << 23);
        {
            const float up = ldexpf(1.0f, w_exp + g_exp);
#pragma unroll
            for (int j = 0; j < CB_JT; ++j) gh[j] *= up;
        }
        landed();                                                  // k-step 0 of W2_l^T is in buffer 1
        // ---- g_h^T += W2_l^T g_P^T
#pragma unroll
        for (int s = 0; s < CB_KSTEPS; ++s) {
            const int c = CS_CPL + s;
            if (s + 1 < CB_KSTEPS) dma(l, c + 1, (c + 1) & 1);
            else if (l + 1 < a.n_layers) dma(l + 1, 0, 0);
            const unsigned char* Ws = Ws0 + (c & 1) * CB_BUF;
            // the lane's slots 8 s .. 8 s + 7 as f16 pairs
            using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
            u32x4 q0, q1;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const float v0 = 8 * s + i < CS_SLOTS ? P[8 * s + i < CS_SLOTS ? 8 * s + i : 0] : 0.f;
                const float v1 = 8 * s + i + 1 < CS_SLOTS ? P[8 * s + i + 1 < CS_SLOTS ? 8 * s + i + 1 : 0] : 0.f;
                unsigned ph, pl;
                cs_split16(v0 * g_scale, v1 * g_scale, ph, pl);
                q0[i >> 1] = ph; q1[i >> 1] = pl;
            }
            const f16x8 Gb[CS_NP16] = {__builtin_bit_cast(f16x8, q0), __builtin_bit_cast(f16x8, q1)};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x8 A[CB_JT / 2][CS_NP16];
#pragma unroll
                for (int jj = 0; jj < CB_JT / 2; ++jj)
#pragma unroll
                    for (int p = 0; p < CS_NP16; ++p)
                        A[jj][p] = *reinterpret_cast<const bf16x8*>(Ws + ((half * (CB_JT / 2) + jj) * CS_NP16 + p) * CS_FRAG + lane * 16);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int jj = 0; jj < CB_JT / 2; ++jj)
                        gh[half * (CB_JT / 2) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[jj][PA[i]]), Gb[PB[i]],
                                                                                             gh[half * (CB_JT / 2) + jj], 0, 0, 0);
            }
            if (s + 1 < CB_KSTEPS || l + 1 < a.n_layers) landed();
        }
        {
            const float down = ldexpf(1.0f, -(w_exp + g_exp));
#pragma unroll
            for (int j = 0; j < CB_JT; ++j) gh[j] *= down;
        }
    }

    if (row_valid) {
        if (live) a.g_x[row * a.gxs + d] = gy;
#pragma unroll
        for (int j = 0; j < CB_JT; ++j) {
            const int c = j * MT + 4 * lq;
            if (c < a.H) *reinterpret_cast<f32x4*>(a.g_h + row * a.ghs + c) = gh[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
static int cb_pack(const float* W2, int64_t w2s, int32_t H, int32_t D, int32_t n_layers, const jf_gf_layer* layers, void* packed, void* stream) {
    if (!W2 || !layers || !packed) return JF_ERR_BADARG;
    if (!width_ok(H) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (H > CS_HMAX || H % 4) return JF_ERR_UNSUPPORTED;
    CbPackArgs a{};
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
    a.W2 = W2; a.w2s = w2s; a.H = H; a.D = D; a.n_layers = n_layers; a.N = col; a.out = static_cast<unsigned char*>(packed);
    const int threads = n_layers * CB_KSTEPS * CB_JT * 64;
    hipStream_t st = (hipStream_t)stream;
    jf::launch(cs_absmax_kernel, dim3(1), dim3(1024), 0, st, W2, w2s, col, (int)H,
                       reinterpret_cast<float*>(a.out + (size_t)n_layers * CB_KSTEPS * CB_T_BYTES));
    jf::launch(cb_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

static int cb_chain(const float* in, int64_t in_stride, const float* W1, int64_t w1s, const float* b1, const void* packed, const void* packedT,
                    int32_t K1, int32_t H, const float* z, int64_t zs, const float* aux, int64_t B, int32_t D, int32_t n_layers,
                    const jf_gf_layer* layers, const float* g_xout, int64_t gxos, const float* g_ld, const float* g_blp, float* g_x, int64_t gxs,
                    float* g_pp, int64_t gpps, float* h_out, int64_t hs, float* g_h, int64_t ghs, float* g_absmax, void* stream) {
    if (!in || !W1 || !b1 || !packed || !packedT || !z || !aux || !layers || !g_x || !g_pp || !h_out || !g_h) return JF_ERR_BADARG;
    if (!width_ok(K1) || !width_ok(H) || !rows_ok(B) || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    if (K1 > CS_K1MAX || H > CS_HMAX || H % 4) return JF_ERR_UNSUPPORTED;
    const uintptr_t al = reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(packedT) | reinterpret_cast<uintptr_t>(aux) |
                         reinterpret_cast<uintptr_t>(g_pp) | reinterpret_cast<uintptr_t>(h_out) | reinterpret_cast<uintptr_t>(g_h);
    if ((al & 15u) || gpps % 4 || hs % 4 || ghs % 4 || gpps < (int64_t)n_layers * CB_PROW || hs < H || ghs < H) return JF_ERR_UNSUPPORTED;
    CbArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        const jf_gf_layer& h = layers[l];
        if (!cs_layer_supported(h, D)) return JF_ERR_UNSUPPORTED;
        CsLayer& o = a.L[l];
        o.hh = h.hh_iter; o.model_offset = h.model_offset; o.inv_type = h.inverse_function_type;
        o.wmin = (float)h.width_min; o.inv_wmax = (float)(1.0 / h.width_max); o.nmin = (float)h.norm_min; o.nmax = (float)h.norm_max;
    }
    if (B == 0) return JF_OK;
    a.in = in; a.in_stride = in_stride; a.W1 = W1; a.w1s = w1s; a.b1 = b1;
    a.packed = static_cast<const unsigned char*>(packed); a.packedT = static_cast<const unsigned char*>(packedT); a.K1 = K1; a.H = H;
    a.z = z; a.zs = zs; a.aux = aux; a.B = B; a.D = D; a.n_layers = n_layers;
    a.g_xout = g_xout; a.gxos = gxos; a.g_ld = g_ld; a.g_blp = g_blp; a.g_x = g_x; a.gxs = gxs; a.g_pp = g_pp; a.gpps = gpps;
    a.h_out = h_out; a.hs = hs; a.g_h = g_h; a.ghs = ghs; a.g_absmax = g_absmax;
    const size_t lds = 2 * CB_BUF;                                 // (phase 1's scratch, <= 22.8 KB at K1 = 28, fits buffer 1)
    static LdsAttrOnce attr;
    attr.set((const void*)cond_gf_split_bwd_kernel, (int)lds);
    jf::launch(cond_gf_split_bwd_kernel, dim3((unsigned)((B + CS_ROWS1 - 1) / CS_ROWS1)), dim3(256), lds, (hipStream_t)stream, a);
    return hipPeekAtLastError() == hipSuccess ? JF_OK : JF_ERR_LAUNCH;
}

}  // namespace jf

extern "C" {
int64_t jf_cond_gf_bwd_packed_bytes(int32_t D, int32_t n_layers, const jf_gf_layer* layers) {
    if (!layers || n_layers < 1 || n_layers > JF_MAX_CHAIN) return JF_ERR_BADARG;
    for (int l = 0; l < n_layers; ++l)
        if (!jf::cs_layer_supported(layers[l], D)) return JF_ERR_UNSUPPORTED;
    return (int64_t)n_layers * jf::CB_KSTEPS * jf::CB_T_BYTES + 16;
}
int jf_cond_gf_bwd_pack_f32(const float* W2, int64_t w2s, int32_t H, int32_t D, int32_t n, const jf_gf_layer* L, void* packed, void* s) {
    return jf::cb_pack(W2, w2s, H, D, n, L, packed, s);
}
int jf_cond_gf_chain_inv_split_bwd_f32(const float* in, int64_t is, const float* W1, int64_t w1s, const float* b1, const void* packed,
                                       const void* packedT, int32_t K1, int32_t H, const float* z, int64_t zs, const float* aux, int64_t B,
                                       int32_t D, int32_t n, const jf_gf_layer* L, const float* g_xout, int64_t gxos, const float* g_ld,
                                       const float* g_blp, float* g_x, int64_t gxs, float* g_pp, int64_t gpps, float* h_out, int64_t hs,
                                       float* g_h, int64_t ghs, float* g_absmax, void* s) {
    return jf::cb_chain(in, is, W1, w1s, b1, packed, packedT, K1, H, z, zs, aux, B, D, n, L, g_xout, gxos, g_ld, g_blp, g_x, gxs, g_pp, gpps,
                        h_out, hs, g_h, ghs, g_absmax, s);
}
}
