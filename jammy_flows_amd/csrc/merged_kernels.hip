// The two SIDE blocks of a log-prob step in one launch: jf_merge_begin / jf_merge_end.
//
// The sub-manifold blocks of the log-prob direction are independent given the targets (every amortisation MLP reads the targets and the
// conditional input only, main/default.py:946-962), and each already is one launch here.  At the shard sizes of the 8-GPU strong-scaling
// measurement (2^17 rows per GPU) the two small blocks of pdf("e4+s2+e4", "gggg+f+gggg") -- the broadcast g chain and the `f` block -- are bound
// by the latency of their own dependent chains, not by the chip: 0.030 + 0.024 ms where linear scaling from 2^20 rows would give 0.015 + 0.013
// (profiles/r04_rows_sweep.md).  Side streams do not help (each kernel claims the whole chip).
//
// The host side brackets the two blocks' ordinary entry points with jf_merge_begin / jf_merge_end: between them jf::launch hands every launch
// to a sink (the mechanism step plans record with), and jf_merge_end issues ONE grid that holds the workgroups of both -- the broadcast chain's
// first, then the `f` block's -- so both chains are resident together (four workgroups per CU) and overlap each other's latency: 0.036 ms for
// the pair at 2^17 rows instead of 0.055, one launch ramp instead of two.  The workgroups run the blocks' own device code (jf_gfb.h,
// jf_cond_mchain.h: the stand-alone kernels are wrappers of the same bodies): every row gets bit for bit the result of the separate launches.
// The fused conditional block follows in a launch of its own and adds the two blocks' sums in its epilogue, as before.
//
// What a merge accepts: exactly one launch of the broadcast g chain, log-prob direction, float32, D <= 4 (gfb_chain_inv_kernel or its
// lanes-per-row form gfbg_chain_inv_kernel) and one of cond_mchain_kernel<float, FFam> with one layer, over the same B rows.  Anything else:
// jf_merge_end issues the captured launches one by one, in order, and returns JF_MERGE_DECLINED -- the results are in place either way.
//
// Tried and dropped (round 5, profiles/r05_merge_experiments.md): the fused conditional block in the same grid, with the per-row sums added by
// the last wave to arrive at a 256-row group.  A grid that carries the fused block holds EVERY workgroup to its budget (168 registers, 50 KB
// of LDS: three workgroups per CU), which starves the latency-bound side blocks of the waves they need: 0.125 ms at 2^17 rows (separate
// launches: 0.126, this file: 0.111), slower above, ahead only below 2^14 rows (0.040 vs 0.046 ms).
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jf_cond_mchain.h"
#include "jf_gfb.h"
#include "jf_merge.h"

namespace jf {

// (one-layer copy of the `f` block's arguments: an `f` layer descriptor is 872 bytes, a launch has 4 KB of arguments)
using CmArgs1 = CmArgs<float, FFam::CLayer, 1>;
struct MergeHead { int n_g, n_f; };          // workgroups of the g chain (the grid's first n_g) and of the f block

// DB: dimension of the broadcast g chain (template parameter of its body), GL: its lanes per row (1, or the power of two that holds DB)
template <int DB, int GL>
__global__ void __launch_bounds__(256, 4) merged_side_kernel(const GfChainArgs<float> ga, const CmArgs1 fa, const MergeHead h) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int b = (int)blockIdx.x;
    if (b < h.n_g) {
        if constexpr (GL == 1) gfb_chain_inv_body<float, DB>(ga, b, smem_raw);
        else gfbg_chain_inv_body<float, DB, GL>(ga, b, smem_raw);
    } else {
        cond_mchain_body<float, FFam, 256, false, CmArgs1>(fa, b - h.n_g, h.n_f, smem_raw);
    }
}

// ---------------------------------------------------------------------------------------------------------- host side
namespace {

struct Captured {
    const void* fn; dim3 grid, block; size_t lds;
    std::vector<std::vector<unsigned char>> args;
    std::vector<size_t> sizes, aligns;
};

struct MergeSink : PlanSink {
    PlanSink* outer = nullptr;
    std::vector<Captured> caps;
    void add_launch(const void* fn, dim3 grid, dim3 block, size_t lds, void** args, const size_t* sizes, const size_t* aligns, int n) override {
        Captured c;
        c.fn = fn; c.grid = grid; c.block = block; c.lds = lds;
        for (int i = 0; i < n; ++i) {
            const unsigned char* p = static_cast<const unsigned char*>(args[i]);
            c.args.emplace_back(p, p + sizes[i]);
            c.sizes.push_back(sizes[i]); c.aligns.push_back(aligns[i]);
        }
        caps.push_back(std::move(c));
    }
    void forget(const PlanSink* dead) override { if (outer == dead) outer = nullptr; }
};
thread_local MergeSink* g_merge = nullptr;

// the captured launches, one by one, to whoever would have received them (an outer plan recording, or the GPU)
int replay(MergeSink& ms, hipStream_t st) {
    int rc = JF_OK;
    for (Captured& c : ms.caps) {
        std::vector<void*> ptrs;
        for (auto& a : c.args) ptrs.push_back(a.data());
        if (ms.outer) {
            std::vector<size_t> sz = c.sizes, al = c.aligns;
            sz.push_back(0); al.push_back(0);
            ms.outer->add_launch(c.fn, c.grid, c.block, c.lds, ptrs.data(), sz.data(), al.data(), (int)c.args.size());
        } else if (hipLaunchKernel(c.fn, c.grid, c.block, ptrs.data(), c.lds, st) != hipSuccess) {
            rc = JF_ERR_LAUNCH;
        }
    }
    return rc;
}

struct SideArgs { GfChainArgs<float> g; CmArgs1 f; MergeHead h; };
static_assert(sizeof(SideArgs) <= 4096, "a launch carries at most 4 KB of arguments");

template <int DB, int GL> int launch_side(const SideArgs& a, size_t lds, hipStream_t st) {
    static LdsAttrOnce attr;
    attr.set((const void*)merged_side_kernel<DB, GL>, 80 * 1024);
    jf::launch(merged_side_kernel<DB, GL>, dim3((unsigned)(a.h.n_g + a.h.n_f)), dim3(256), lds, st, a.g, a.f, a.h);
    return check_launch();
}

}  // namespace

}  // namespace jf

using namespace jf;

extern "C" {

int jf_merge_begin(void) {
    if (g_merge) return JF_ERR_BADARG;                               // no nesting
    MergeSink* ms = new (std::nothrow) MergeSink();
    if (!ms) return JF_ERR_LAUNCH;
    ms->outer = plan_sink();
    plan_sink() = ms;
    g_merge = ms;
    return JF_OK;
}

int jf_merge_abort(void) {
    if (!g_merge) return JF_ERR_BADARG;
    if (plan_sink() == g_merge) plan_sink() = g_merge->outer;
    delete g_merge;
    g_merge = nullptr;
    return JF_OK;
}

int jf_merge_captured(void) { return g_merge ? (int)g_merge->caps.size() : JF_ERR_BADARG; }

int jf_merge_end(void* stream) {
    if (!g_merge || plan_sink() != g_merge) return JF_ERR_BADARG;
    MergeSink* ms = g_merge;
    plan_sink() = ms->outer;
    g_merge = nullptr;
    struct Done { MergeSink* p; ~Done() { delete p; } } done{ms};
    hipStream_t st = (hipStream_t)stream;
    if (ms->caps.empty()) return JF_OK;

    // ---- can these launches share a grid?
    static const int disabled = getenv("JF_MERGE_OFF") ? atoi(getenv("JF_MERGE_OFF")) : 0;
    bool ok = !disabled && ms->caps.size() == 2;
    SideArgs a{};
    int db = 0, gl = 1;
    size_t lds = 0;
    bool have_g = false, have_f = false;
    for (size_t i = 0; ok && i < ms->caps.size(); ++i) {
        const Captured& c = ms->caps[i];
        if (c.args.size() != 1 || c.block.x != 256 || c.block.y != 1 || c.block.z != 1 || c.grid.y != 1 || c.grid.z != 1) { ok = false; break; }
        lds = c.lds > lds ? c.lds : lds;
        bool is_g = false;
        for (int d = 1; d <= 4 && !is_g; ++d) {
            if (c.fn == gfb_inv_kernel_f32(d)) { is_g = true; db = d; gl = 1; }
            else if (d >= 2 && c.fn == gfbg_inv_kernel_f32(d)) { is_g = true; db = d; gl = d == 2 ? 2 : 4; }
        }
        if (is_g) {
            if (have_g || c.args[0].size() != sizeof(a.g)) { ok = false; break; }
            have_g = true;
            std::memcpy(&a.g, c.args[0].data(), sizeof(a.g));
            a.h.n_g = (int)c.grid.x;
        } else if (c.fn == cond_f_inv_kernel_f32()) {
            CmArgs<float, FFam::CLayer> w;
            if (have_f || c.args[0].size() != sizeof(w)) { ok = false; break; }
            have_f = true;
            std::memcpy(&w, c.args[0].data(), sizeof(w));
            if (w.n_layers != 1) { ok = false; break; }
            a.f.in = w.in; a.f.in_stride = w.in_stride; a.f.W1 = w.W1; a.f.w1s = w.w1s; a.f.b1 = w.b1; a.f.W2 = w.W2; a.f.w2s = w.w2s; a.f.b2 = w.b2;
            a.f.K1 = w.K1; a.f.H = w.H; a.f.N = w.N; a.f.x = w.x; a.f.xs = w.xs; a.f.ld_in = w.ld_in; a.f.B = w.B; a.f.n_layers = 1; a.f.dim = w.dim;
            a.f.tile_stride = w.tile_stride; a.f.scratch = w.scratch; a.f.tab = w.tab; a.f.col0[0] = w.col0[0]; a.f.L[0] = w.L[0];
            a.f.x_out = w.x_out; a.f.xos = w.xos; a.f.ld_out = w.ld_out; a.f.blp_in = w.blp_in; a.f.blp_out = w.blp_out; a.f.status = w.status;
            a.h.n_f = (int)c.grid.x;
        } else {
            ok = false;
        }
    }
    ok = ok && have_g && have_f && a.g.B == a.f.B;
    if (!ok) {
        const int rc = replay(*ms, st);
        return rc == JF_OK ? JF_MERGE_DECLINED : rc;
    }
    switch (db * 10 + gl) {
        case 11: return launch_side<1, 1>(a, lds, st);
        case 21: return launch_side<2, 1>(a, lds, st);
        case 22: return launch_side<2, 2>(a, lds, st);
        case 31: return launch_side<3, 1>(a, lds, st);
        case 34: return launch_side<3, 4>(a, lds, st);
        case 44: return launch_side<4, 4>(a, lds, st);
        default: return launch_side<4, 1>(a, lds, st);
    }
}

}  // extern "C"
