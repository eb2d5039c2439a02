// The fused conditional e-block (amortisation MLP + g layers, split-bf16 / split-f16 matrix arithmetic, parameters in registers): kernel
// arguments and the body of cond_gf_split_kernel as a device function -- shared by the stand-alone kernel (cond_split_kernels.hip, which
// documents the design) and the merged log-prob step (merged_kernels.hip).  main/default.py:656-670, 946-962, 998-1031.
#pragma once
#include "jf_cond_split.h"

namespace jf {

// ---------------------------------------------------------------------------------------------------------- the fused kernel
constexpr int CS_MAX_PRE = 4;
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
    CondIn cin;                                // n > 0: the MLP's input rows are these segments of the targets / conditional input (jf_cond_in.h)
    // log-prob direction, the LAST block of a pdf: the log-dets / base log-probs of the blocks before it (each block evaluated on its own),
    // added in list order in front of this block's -- ld_out / blp_out then hold the pdf's totals and total = blp_out + ld_out: the sums
    // jf_combine_rows would make in a launch of its own (main/default.py:1110-1117)
    const float* ld_pre[CS_MAX_PRE]; const float* blp_pre[CS_MAX_PRE];
    int n_ld_pre, n_blp_pre;
    float* total;
};

// RG = row groups (16 rows each) per wave.  With RG = 2 every A fragment read from LDS feeds two MFMAs (half the ds_read_b128 per row,
// six independent accumulators per piece product instead of three) and the chunk barriers are paid once per 128 rows instead of 64.
// SAVE (log-prob direction with gradients wanted): every layer's input coordinate and mixture sums go to a.aux, 5 floats per (layer, row,
// coordinate lane) -- 320 bytes per row of a 4-layer block instead of the 2.2 KB parameter row the adjoint would otherwise need.
// Occupancy: the f16-pair log-prob variants are held to 168 VGPRs (8 spilled with two row groups) so that THREE workgroups share a CU (3 x 49.6 KB
// of LDS): a third wave per SIMD fills issue slots the other two leave while they sit in the same phase -- 0.66 -> 0.60 ms per 2^20 rows on
// the same box.  The bf16-triple variants (246 VGPRs) and the sampling direction (solver loops) keep two.
// `block`: the workgroup's index among the block's workgroups
template <int RG, bool FWD, bool SAVE, int NP>
__device__ __forceinline__ void cond_gf_split_body(const CsArgs& a, const int block, unsigned char* smem_raw) {
    using G = CsGeom<NP>;
    constexpr int CS_ROWS = CS_ROWS1 * RG;
    constexpr int MT = 16;
    unsigned char* Ws0 = smem_raw;                                 // two packed chunks (double buffer)
    float* Xs = reinterpret_cast<float*>(smem_raw + G::CHUNK);   // phase 1 only (overlays buffer 1 while chunk 0 lands in buffer 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int64_t row0 = (int64_t)block * CS_ROWS;
    const int64_t last = a.B - 1;
    const int D = a.D;
    // ---- chunk streaming: LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no register hop) into the buffer that is not being
    //      multiplied; wave w moves KiB pieces w, w + 4, ... of the chunk, the bias tail goes with the last piece of wave 0
    // buffer form (buffer_load_dwordx4 ... offen lds): resource + per-lane byte offset are fixed for the whole kernel, the chunk / piece offset
    // is a scalar -- no VALU address arithmetic per DMA instruction (the flat global_load_lds form spent ~8 vector integer instructions on each
    // of its 64-bit addresses, 230 per layer)
    const __amdgpu_buffer_rsrc_t packed_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.packed), 0, a.n_layers * CS_CPL * G::CHUNK, 0x00027000);
    const int lane_off = wave * 1024 + lane * 16;
    auto dma = [&](int chunk) {
        // the packed image is in the log-prob direction's consumption order (last layer first); the sampling direction walks the layers forwards
        const int img = FWD ? (a.n_layers - 1 - chunk / CS_CPL) * CS_CPL + chunk % CS_CPL : chunk;
        const int g = img * G::CHUNK;
        cs_dma_chunk(packed_rsrc, Ws0 + (chunk & 1) * G::CHUNK, g, lane_off, wave, lane, G::W, G::B);
    };
    dma(0);                                                        // lands in buffer 0 while phase 1 works in buffer 1

    // ---- phase 1: h^T = tanh(W1 x^T + b1) for the wave's rows as MFMA B operands, three bf16 pieces (jf_cond_split.h)
    bf16x8 hB[RG][CS_KSTEPS][NP];
    cs_hidden<RG, false, NP>(a.in, a.in_stride, a.W1, a.w1s, a.b1, a.K1, a.H, row0, last, Xs, hB, nullptr, 0, a.cin.n ? &a.cin : nullptr);
    // ---- flow state: lane = (row li of the row group's 16, coordinate lq)
    const bool live = lq < D, leader = lq == 0;
    int d = live ? lq : D - 1;
    int64_t row[RG]; bool row_valid[RG];
    float x[RG], ld[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        row[g] = row0 + (wave * RG + g) * MT + li;
        row_valid[g] = row[g] <= last;
        const int64_t rrow = row_valid[g] ? row[g] : last;
        x[g] = a.x[rrow * a.xs + d];
        ld[g] = a.ld_in ? a.ld_in[rrow] : 0.f;
    }

    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const int n_chunks = a.n_layers * CS_CPL;
    landed();                                                      // chunk 0 is in buffer 0 and every wave is done with Xs / W1s / b1s (buffer 1)
    int chunk = 0;
    for (int li = 0; li < a.n_layers; ++li) {
        const int l = FWD ? li : a.n_layers - 1 - li;
        float P[RG][CS_SLOTS];
#pragma unroll
        for (int c = 0; c < CS_CPL; ++c, ++chunk) {
            if (chunk + 1 < n_chunks) dma(chunk + 1);              // in flight while this chunk is multiplied
            const unsigned char* Ws = Ws0 + (chunk & 1) * G::CHUNK;
            const float* Bs = reinterpret_cast<const float*>(Ws + G::W);
            f32x4 acc[RG][CS_CT];
#pragma unroll
            for (int t = 0; t < CS_CT; ++t) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(Bs + t * 16 + 4 * lq);    // bias of columns 4 lq .. 4 lq + 3
#pragma unroll
                for (int g = 0; g < RG; ++g) acc[g][t] = bias;
            }
            // A fragments one k-step ahead of the MFMAs that consume them (the LDS latency of a k-step's 9 reads hides behind the previous
            // k-step's MFMAs instead of being waited for in front of each MFMA)
            bf16x8 A[2][CS_CT][NP];
            auto load_a = [&](int s, int buf) {
#pragma unroll
                for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        A[buf][t][p] = *reinterpret_cast<const bf16x8*>(Ws + ((t * CS_KSTEPS + s) * NP + p) * CS_FRAG + lane * 16);
            };
            load_a(0, 0);
#pragma unroll
            for (int s = 0; s < CS_KSTEPS; ++s) {
                const int b = s & 1;
                if (s + 1 < CS_KSTEPS) load_a(s + 1, b ^ 1);
                if constexpr (NP == 3) {
                    // products with piece indices pa + pb <= 2, smallest first
                    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                            for (int g = 0; g < RG; ++g)
                                acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[b][t][PA[i]], hB[g][s][PB[i]], acc[g][t], 0, 0, 0);
                } else {
                    // lo x hi, hi x lo, hi x hi
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                            for (int g = 0; g < RG; ++g) {
                                const f16x8 af = __builtin_bit_cast(f16x8, A[b][t][i == 0 ? 1 : 0]), bf = __builtin_bit_cast(f16x8, hB[g][s][i == 1 ? 1 : 0]);
                                acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[g][t], 0, 0, 0);
                            }
                }
            }
            if constexpr (NP == 3) {
#pragma unroll
                for (int g = 0; g < RG; ++g)
#pragma unroll
                    for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) P[g][4 * (c * CS_CT + t) + r] = acc[g][t][r];
            } else {
                const float inv = Bs[CS_B_BYTES / 4];                // 2^-(e + 14): the scales of W2 and h undone (exact)
#pragma unroll
                for (int g = 0; g < RG; ++g)
#pragma unroll
                    for (int t = 0; t < CS_CT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) P[g][4 * (c * CS_CT + t) + r] = acc[g][t][r] * inv;
            }
            if (c + 1 < CS_CPL) landed();                          // next chunk in place, every wave has read this one
        }
        // ---- flow phase on the lane's register rows (raw parameters; jf_gf.h arithmetic)
        const CsLayer o = a.L[l];                                  // uniform index: scalar loads from the kernarg segment
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            if constexpr (!FWD) {
                float xg = x[g] - P[g][CS_SLOT_OFF];               // euclidean_base.py:40-45 (zero column when the layer models no offset)
#pragma unroll
                for (int i = 0; i < CS_HH; ++i) {
                    if (i < o.hh) {                                // x <- Q^T x (gaussianization_flow.py:1038), H_i = I - 2 v v^T / |v|^2
                        const float v = live ? P[g][CS_SLOT_ROT + i] : 0.f;
                        const float n2 = cs_rsum(v * v), dot = cs_rsum(v * xg);
                        xg -= 2.0f * dot * M<float>::rcp(n2) * v;
                    }
                }
                CsSums sums;
                const MixQ<float> q = cs_mixture(P[g], o, xg, live, SAVE ? &sums : nullptr);
                if constexpr (SAVE) {
                    {   // (row number re-derived from the lane index: see the epilogue)
                        int t2 = tid;
                        asm volatile("" : "+v"(t2));
                        row[g] = row0 + ((t2 >> 6) * RG + g) * MT + (t2 & 15);
                        row_valid[g] = row[g] <= last;
                    }
                    if (row_valid[g]) {
                        const int64_t slot = ((int64_t)l * a.B + row[g]) * 4 + lq;
                        reinterpret_cast<f32x4*>(a.aux)[slot] = f32x4{sums.C, sums.S, sums.P, sums.invN};
                        a.aux[(int64_t)a.n_layers * a.B * 16 + slot] = x[g];
                    }
                }
                const IcdfOut<float> sy = gf_icdf<float>(o.inv_type, q);
                x[g] = sy.y;
                ld[g] += cs_rsum(live ? sy.logd : 0.f);
            } else {
                // sampling direction (gaussianization_flow.py:911-989): regulate the row once in its registers, solve stage(mixture(x)) = z by
                // 25 bisection + <= 20 Newton steps, log-det from the solution, then x <- Q x and the offset (euclidean_base.py:63-68)
                cs_derive(P[g], o);
                float slogd;
                float xs = cs_solve<float>(P[g], o.inv_type, live, x[g], row_valid[g], leader, a.status, [](float v) { return cs_rsum(v); },
                                    [](float v) { return cs_rmax(v); }, nullptr, &slogd);
                ld[g] -= cs_rsum(live ? slogd : 0.f);
#pragma unroll
                for (int i = CS_HH - 1; i >= 0; --i) {
                    if (i < o.hh) {
                        const float v = live ? P[g][CS_SLOT_ROT + i] : 0.f;
                        const float n2 = cs_rsum(v * v), dot = cs_rsum(v * xs);
                        xs -= 2.0f * dot * M<float>::rcp(n2) * v;
                    }
                }
                x[g] = xs + P[g][CS_SLOT_OFF];
            }
        }
        landed();
    }

    // The epilogue's row numbers are derived AGAIN from the lane index, behind an empty asm the compiler cannot see through: the copies made
    // before phase 1 (two 64-bit row numbers, the coordinate index) otherwise stay in registers through every layer, and at this kernel's 168
    // registers (three workgroups per CU) they were what spilled to scratch (8 registers: 36 B per lane written and read back, 75 MB per
    // 2^20-row launch on the write counters).
    if constexpr (!FWD) {                                          // (the sampling variants gain registers from it: 141 -> 153; they have none to spare either)
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int lane2 = t2 & 63, wave2 = t2 >> 6, li2 = lane2 & 15, lq2 = lane2 >> 4;
        d = lq2 < D ? lq2 : D - 1;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            row[g] = row0 + (wave2 * RG + g) * MT + li2;
            row_valid[g] = row[g] <= last;
        }
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        if (row_valid[g] && live) a.x_out[row[g] * a.xos + d] = x[g];
        if constexpr (FWD) {
            if (row_valid[g] && leader) a.ld_out[row[g]] = ld[g];
        } else {
            float sb = 0.f;
            if (a.blp_out) sb = cs_rsum(live ? -0.5f * x[g] * x[g] - M<float>::HALF_LN_2PI : 0.f);
            if (row_valid[g] && leader) {
                float ldv = ld[g], bv = sb + (a.blp_in ? a.blp_in[row[g]] : 0.f);
                if (a.n_ld_pre > 0) {                               // uniform: list order, this block last (bit for bit jf_combine_rows)
                    float t = a.ld_pre[0][row[g]];
                    for (int i = 1; i < a.n_ld_pre; ++i) t += a.ld_pre[i][row[g]];
                    ldv = t + ldv;
                }
                if (a.n_blp_pre > 0) {
                    float t = a.blp_pre[0][row[g]];
                    for (int i = 1; i < a.n_blp_pre; ++i) t += a.blp_pre[i][row[g]];
                    bv = t + bv;
                }
                a.ld_out[row[g]] = ldv;
                if (a.blp_out) a.blp_out[row[g]] = bv;
                if (a.total) a.total[row[g]] = bv + ldv;
            }
            const float bad = cs_rmax((live && !M<float>::finite(x[g])) ? 1.f : 0.f);
            status_add(a.status, JF_STATUS_NONFINITE, row_valid[g] && leader && (bad > 0.f || !M<float>::finite(ld[g])));
        }
    }
}

}  // namespace jf
