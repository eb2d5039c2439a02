// Step plans: a whole evaluation step (every launch of pdf.forward / pdf.sample for one input shape) recorded once and re-issued from C in ONE call.
//
// The reference walks its sub-manifolds in Python and issues hundreds of ATen launches per step (main/default.py:879-1057: all_layer_inverse,
// :1059-1117: forward).  This library's host side needs 4 launches for the same step, but each went through its own ctypes call with Python
// bookkeeping around it (~30 us each): at 2^17 rows the step was bound by the host, not by the GPU -- which is exactly the
// per-GPU batch of the 8-GPU strong-scaling measurement.  A plan removes the host from the step:
//
//   record   jf_plan_record_begin(p); <any sequence of jf_* entry points on this thread>; jf_plan_record_end(p)
//            jf::launch (jf_common.h) hands every kernel launch to the plan instead of the GPU: kernel address, grid, block, LDS bytes and a
//            byte copy of every kernel argument.  Host-side decisions of the entry points (kernel variant, grid size, occupancy queries,
//            attribute setting) are therefore taken once, at record time.
//   rebind   buffers the caller wants to exchange between replays (inputs, outputs) are declared as SLOTS (jf_plan_add_slot: base, bytes)
//            before recording.  At record_end every 8-byte word of every recorded argument that points into a slot becomes a relocation
//            (slot, offset into the slot); jf_plan_launch writes base_of_slot_now + offset there.  Everything else (weights, packed images,
//            intermediate buffers owned by the caller's plan object) keeps its recorded address.
//   replay   jf_plan_launch(p, slot_bases, n, stream): hipLaunchKernel per recorded launch (+ the recorded memsets / device-to-host copies)
//            on `stream`, no allocation, no synchronisation.  ~3 us of host time per launch.
//   timing   jf_plan_set_timing(p, 1): a replay also records a HIP event before and after every op ON THE LAUNCH STREAM;
//            jf_plan_read_timing returns the summed elapsed time per op (what bench.py's roofline needs: per-kernel time inside the timed region).
//
// A plan is replayed by one host thread at a time (the relocations are written into the plan's own argument storage).
#include <hip/hip_ext.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "jf_common.h"

namespace jf {

namespace {
thread_local PlanSink* g_sink = nullptr;
}
PlanSink*& plan_sink() { return g_sink; }

}  // namespace jf

extern "C" {
// which iteration rule the solvers of THIS library were built with: 0 = the product rules, 1 = the reference's own (libjammy_hip_audit.so,
// -DJF_NEWTON_RULE_REFERENCE: 25 bisections on [-1e5, 1e5], Newton until 1e-14 / 20 steps, no float32 floor, 'v' until 1e-12)
#ifdef JF_NEWTON_RULE_REFERENCE
int jf_get_newton_rule(void) { return 1; }
#else
int jf_get_newton_rule(void) { return 0; }
#endif
}

struct jf_plan : jf::PlanSink {
    enum Kind { LAUNCH = 0, MEMSET = 1, COPY_D2H = 2, FORK = 3, JOIN = 4 };
    static constexpr int MAX_LANES = 4;                            // lane 0 = the caller's stream
    struct Op {
        int kind;
        int lane;
        int any_order;                                             // LAUNCH: issued without the queue's barrier bit (hipExtAnyOrderLaunch): may overlap its predecessors
        const void* fn; dim3 grid, block; unsigned lds;
        int first_arg, n_args;                                     // LAUNCH: indices into arg_off; MEMSET / COPY: two pointer "arguments" (dst, src)
        int value; int64_t bytes;
    };
    struct Slot { uint64_t base; int64_t bytes; };
    struct Reloc { size_t off; int slot; uint64_t delta; };
    std::vector<Op> ops;
    std::vector<size_t> arg_off;                                   // offset of every argument inside blob
    std::vector<unsigned char> blob;                               // argument bytes (each argument at its own alignment, at least 8)
    std::vector<void*> ptrs;                                       // &blob[arg_off[i]], rebuilt at record_end
    std::vector<Slot> slots;
    std::vector<Reloc> relocs;
    bool recording = false, finalised = false;
    int error = 0;
    int cur_lane = 0;                                              // lane of the ops being recorded
    int cur_any_order = 0;
    hipStream_t lane_stream[MAX_LANES] = {};                       // [0] unused; created at the first replay that needs them
    hipEvent_t fork_event = nullptr, lane_event[MAX_LANES] = {};
    int lanes_device = -1;
    // timing
    bool timing = false;
    int timing_every = 1;                                          // events on every n-th replay while timing is on
    int64_t replay_no = 0;
    bool has_lanes = false, has_any_order = false;
    std::vector<std::vector<hipEvent_t>> pending;                  // per timed replay: 2 n_ops events
    std::vector<hipEvent_t> pool;
    std::vector<double> ms_sum;
    int64_t timed_replays = 0;

    size_t push_arg(const void* src, size_t size, size_t align) {
        if (align < 8) align = 8;
        size_t off = (blob.size() + align - 1) / align * align;
        blob.resize(off + (size + 7) / 8 * 8, 0);
        std::memcpy(blob.data() + off, src, size);
        arg_off.push_back(off);
        return off;
    }
    void add_launch(const void* fn, dim3 grid, dim3 block, size_t lds, void** args, const size_t* sizes, const size_t* aligns, int n) override {
        if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;       // (an empty launch is an error for hipLaunchKernel; entry points return before it for B = 0)
        Op o{};
        o.kind = LAUNCH; o.lane = cur_lane; o.any_order = cur_any_order; o.fn = fn; o.grid = grid; o.block = block; o.lds = (unsigned)lds; o.first_arg = (int)arg_off.size(); o.n_args = n;
        for (int i = 0; i < n; ++i) push_arg(args[i], sizes[i], aligns[i]);
        ops.push_back(o);
    }
    int add_mem(int kind, const void* dst, const void* src, int value, int64_t bytes) {
        if (!recording || !dst || bytes < 0 || (kind == COPY_D2H && !src)) return JF_ERR_BADARG;
        Op o{};
        o.kind = kind; o.lane = cur_lane; o.first_arg = (int)arg_off.size(); o.n_args = 2; o.value = value; o.bytes = bytes;
        push_arg(&dst, sizeof(void*), alignof(void*));
        push_arg(&src, sizeof(void*), alignof(void*));
        ops.push_back(o);
        return JF_OK;
    }
    void finalise() {
        ptrs.resize(arg_off.size());
        for (size_t i = 0; i < arg_off.size(); ++i) ptrs[i] = blob.data() + arg_off[i];
        relocs.clear();
        // every aligned 8-byte word of the argument storage that is an address inside a slot.  Device virtual addresses are 47-bit values far
        // above any count, stride or float pair a kernel argument holds, so a word in [base, base + bytes) IS a pointer into that buffer.
        for (const Op& o : ops) {
            for (int a = 0; a < o.n_args; ++a) {
                if (o.kind == COPY_D2H && a == 0) continue;          // the host destination is not a slot address
                const size_t begin = arg_off[o.first_arg + a];
                const size_t end = (size_t)(o.first_arg + a + 1) < arg_off.size() ? arg_off[o.first_arg + a + 1] : blob.size();
                for (size_t off = begin; off + 8 <= end; off += 8) {
                    uint64_t v;
                    std::memcpy(&v, blob.data() + off, 8);
                    for (size_t s = 0; s < slots.size(); ++s)
                        if (v >= slots[s].base && v < slots[s].base + (uint64_t)slots[s].bytes) {
                            relocs.push_back({off, (int)s, v - slots[s].base});
                            break;
                        }
                }
            }
        }
        // a fork / join concerns the side lanes used between the fork and its join
        for (size_t i = 0; i < ops.size(); ++i) {
            if (ops[i].kind != FORK) continue;
            int mask = 0;
            size_t j = i + 1;
            for (; j < ops.size() && ops[j].kind != JOIN; ++j) mask |= 1 << ops[j].lane;
            ops[i].value = mask;
            if (j < ops.size()) ops[j].value = mask;
        }
        for (const Op& o : ops) { has_lanes = has_lanes || o.kind == FORK; has_any_order = has_any_order || (o.kind == LAUNCH && o.any_order); }
        ms_sum.assign(ops.size(), 0.0);
        finalised = true;
    }
    // side streams + their events, on the device the plan is replayed on (created once)
    bool ensure_lanes() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (lanes_device == dev) return true;
        if (lanes_device >= 0) return false;                        // a plan stays on the device of its first replay
        if (hipEventCreateWithFlags(&fork_event, hipEventDisableTiming) != hipSuccess) return false;
        for (int l = 1; l < MAX_LANES; ++l) {
            if (hipStreamCreateWithFlags(&lane_stream[l], hipStreamNonBlocking) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&lane_event[l], hipEventDisableTiming) != hipSuccess) return false;
        }
        lanes_device = dev;
        return true;
    }
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ~jf_plan() override {
        for (auto& v : pending) for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : pool) if (e) (void)hipEventDestroy(e);
        if (fork_event) (void)hipEventDestroy(fork_event);
        for (int l = 1; l < MAX_LANES; ++l) {
            if (lane_event[l]) (void)hipEventDestroy(lane_event[l]);
            if (lane_stream[l]) (void)hipStreamDestroy(lane_stream[l]);
        }
    }
};

// Plans are named by HANDLES, not by addresses: a C caller (or a fuzzer, tests/test_abi_asan.py) that passes a stale or random value gets
// JF_ERR_BADARG instead of a wild dereference.  handle = tag | (generation << 24) | index.
namespace {
constexpr int64_t PLAN_TAG = (int64_t)0x4a46 << 48;
struct Registry {
    std::mutex mu;
    std::vector<jf_plan*> plans;
    std::vector<uint32_t> gen;
} g_reg;
jf_plan* lookup(int64_t h) {
    if ((h & ~(int64_t)0xffffffffffff) != PLAN_TAG) return nullptr;
    const uint32_t idx = (uint32_t)(h & 0xffffff), gen = (uint32_t)((h >> 24) & 0xffffff);
    std::lock_guard<std::mutex> lock(g_reg.mu);
    if (idx >= g_reg.plans.size() || g_reg.gen[idx] != gen) return nullptr;
    return g_reg.plans[idx];
}
}  // namespace

extern "C" {

int64_t jf_plan_create(void) {
    jf_plan* p = new (std::nothrow) jf_plan();
    if (!p) return JF_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_reg.mu);
    size_t idx = 0;
    while (idx < g_reg.plans.size() && g_reg.plans[idx]) ++idx;
    if (idx == g_reg.plans.size()) {
        if (idx >= 0xffffff) { delete p; return JF_ERR_UNSUPPORTED; }
        g_reg.plans.push_back(nullptr); g_reg.gen.push_back(0);
    }
    g_reg.plans[idx] = p;
    g_reg.gen[idx] = (g_reg.gen[idx] + 1) & 0xffffff;
    return PLAN_TAG | ((int64_t)g_reg.gen[idx] << 24) | (int64_t)idx;
}

int32_t jf_plan_destroy(int64_t h) {
    // lookup and removal under ONE lock: of two threads destroying the same handle exactly one finds it.  A plan is replayed by one
    // host thread at a time (see the top of this file); destroying it while another thread is inside jf_plan_launch is the caller's race.
    jf_plan* p = nullptr;
    {
        if ((h & ~(int64_t)0xffffffffffff) != PLAN_TAG) return JF_ERR_BADARG;
        const uint32_t idx = (uint32_t)(h & 0xffffff), gen = (uint32_t)((h >> 24) & 0xffffff);
        std::lock_guard<std::mutex> lock(g_reg.mu);
        if (idx >= g_reg.plans.size() || g_reg.gen[idx] != gen || !g_reg.plans[idx]) return JF_ERR_BADARG;
        p = g_reg.plans[idx];
        g_reg.plans[idx] = nullptr;
    }
    if (jf::plan_sink() == p) jf::plan_sink() = nullptr;
    else if (jf::plan_sink()) jf::plan_sink()->forget(p);           // (a merge capture begun inside this plan's recording)
    delete p;
    return JF_OK;
}

int32_t jf_plan_add_slot(int64_t h, const void* base, int64_t bytes) {
    jf_plan* p = lookup(h);
    if (!p || !base || bytes <= 0 || p->finalised) return JF_ERR_BADARG;
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    for (const auto& s : p->slots)                                  // overlapping slots would make a pointer's owner ambiguous
        if (b < s.base + (uint64_t)s.bytes && s.base < b + (uint64_t)bytes) return JF_ERR_BADARG;
    p->slots.push_back({b, bytes});
    return (int32_t)p->slots.size() - 1;
}

int32_t jf_plan_record_begin(int64_t h) {
    jf_plan* p = lookup(h);
    if (!p || p->finalised || p->recording || jf::plan_sink() != nullptr) return JF_ERR_BADARG;
    p->recording = true;
    jf::plan_sink() = p;
    return JF_OK;
}

int32_t jf_plan_record_end(int64_t h) {
    jf_plan* p = lookup(h);
    if (!p || !p->recording || jf::plan_sink() != p) return JF_ERR_BADARG;
    jf::plan_sink() = nullptr;
    p->recording = false;
    p->finalise();
    return (int32_t)p->ops.size();
}

int32_t jf_plan_add_memset(int64_t h, void* dst, int32_t value, int64_t bytes) {
    jf_plan* p = lookup(h);
    return p ? p->add_mem(jf_plan::MEMSET, dst, nullptr, value, bytes) : JF_ERR_BADARG;
}

int32_t jf_plan_add_copy_to_host(int64_t h, void* host_dst, const void* src, int64_t bytes) {
    jf_plan* p = lookup(h);
    return p ? p->add_mem(jf_plan::COPY_D2H, host_dst, src, 0, bytes) : JF_ERR_BADARG;
}

// lanes: ops recorded after jf_plan_set_lane(p, l) are issued on side stream l (0 = the caller's stream).  Between a jf_plan_add_fork and the next
// jf_plan_add_join the ops of different lanes must be independent of each other; the fork makes every side stream wait for what the caller's
// stream has been given so far, the join makes the caller's stream wait for the side streams.
int32_t jf_plan_set_lane(int64_t h, int32_t lane) {
    jf_plan* p = lookup(h);
    if (!p || !p->recording || lane < 0 || lane >= jf_plan::MAX_LANES) return JF_ERR_BADARG;
    p->cur_lane = lane;
    return JF_OK;
}
// launches recorded while any_order is on are issued without the queue's barrier bit (hipExtAnyOrderLaunch): they may start before, and run
// beside, the launches recorded before them -- for the INDEPENDENT blocks of a step at small batches, where each kernel leaves the chip
// half empty while it drains.  The first launch after any_order is switched off again is an ordinary (ordered) one: it waits for all of them.
int32_t jf_plan_set_any_order(int64_t h, int32_t on) {
    jf_plan* p = lookup(h);
    if (!p || !p->recording) return JF_ERR_BADARG;
    p->cur_any_order = on != 0;
    return JF_OK;
}
static int32_t plan_sync_op(jf_plan* p, int kind) {
    if (!p || !p->recording) return JF_ERR_BADARG;
    jf_plan::Op o{};
    o.kind = kind; o.first_arg = (int)p->arg_off.size(); o.n_args = 0; o.value = 0;
    p->ops.push_back(o);
    return JF_OK;
}
int32_t jf_plan_add_fork(int64_t h) { return plan_sync_op(lookup(h), jf_plan::FORK); }
int32_t jf_plan_add_join(int64_t h) { return plan_sync_op(lookup(h), jf_plan::JOIN); }

int32_t jf_plan_num_ops(int64_t h) { const jf_plan* p = lookup(h); return p ? (int32_t)p->ops.size() : JF_ERR_BADARG; }
int32_t jf_plan_num_relocations(int64_t h) { const jf_plan* p = lookup(h); return (p && p->finalised) ? (int32_t)p->relocs.size() : JF_ERR_BADARG; }

int32_t jf_plan_launch(int64_t h, const void* const* slot_bases, int32_t n_slots, void* stream) {
    jf_plan* p = lookup(h);
    if (!p || !p->finalised || n_slots != (int32_t)p->slots.size() || (n_slots > 0 && !slot_bases)) return JF_ERR_BADARG;
    for (int i = 0; i < n_slots; ++i)
        if (!slot_bases[i]) return JF_ERR_BADARG;
    for (const auto& r : p->relocs) {
        const uint64_t v = reinterpret_cast<uint64_t>(slot_bases[r.slot]) + r.delta;
        std::memcpy(p->blob.data() + r.off, &v, 8);
    }
    hipStream_t st = (hipStream_t)stream;
    // timed replays.  One stream, ordered launches: n + 1 events recorded BETWEEN the ops (the end of op i is the start of op i + 1; the cheapest
    // form: a timed C3 step costs ~2 % more than an untimed one).  Plans with lanes / any-order launches: a (start, end) pair per op through
    // hipExtLaunchKernel (an event recorded behind such a launch would serialise it) -- ~12 us per launch, for experiments only.
    std::vector<hipEvent_t>* ev = nullptr;
    const bool pairs = p->has_lanes || p->has_any_order;
    if (p->timing && (p->replay_no++ % p->timing_every) == 0) {
        p->pending.emplace_back();
        ev = &p->pending.back();
        const size_t n_ev = pairs ? 2 * p->ops.size() : p->ops.size() + 1;
        for (size_t i = 0; i < n_ev; ++i) ev->push_back(p->get_event());
        if (!pairs && (*ev)[0]) (void)hipEventRecord((*ev)[0], st);
    }
    int rc = JF_OK;
    bool lanes = false;
    for (const auto& o : p->ops) lanes = lanes || o.kind == jf_plan::FORK;
    if (lanes && !p->ensure_lanes()) return JF_ERR_LAUNCH;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const jf_plan::Op& o = p->ops[i];
        hipError_t e = hipSuccess;
        hipStream_t os = (lanes && o.lane > 0) ? p->lane_stream[o.lane] : st;
        if (ev && pairs && o.kind != jf_plan::LAUNCH && (*ev)[2 * i]) (void)hipEventRecord((*ev)[2 * i], os);
        if (o.kind == jf_plan::FORK) {                              // the side streams continue from here: everything issued so far on the caller's stream comes first
            e = hipEventRecord(p->fork_event, st);
            for (int l = 1; l < jf_plan::MAX_LANES && e == hipSuccess; ++l)
                if (o.value >> l & 1) e = hipStreamWaitEvent(p->lane_stream[l], p->fork_event, 0);
        } else if (o.kind == jf_plan::JOIN) {                       // the caller's stream continues after every side stream used since the fork
            for (int l = 1; l < jf_plan::MAX_LANES && e == hipSuccess; ++l)
                if (o.value >> l & 1) {
                    e = hipEventRecord(p->lane_event[l], p->lane_stream[l]);
                    if (e == hipSuccess) e = hipStreamWaitEvent(st, p->lane_event[l], 0);
                }
        } else if (o.kind == jf_plan::LAUNCH) {
            // (timed replays take the kernel's own start / stop events: an event recorded behind an any-order launch would be a barrier)
            if (o.any_order || (ev && pairs))
                e = hipExtLaunchKernel(o.fn, o.grid, o.block, p->ptrs.data() + o.first_arg, o.lds, os, (ev && pairs) ? (*ev)[2 * i] : nullptr,
                                       (ev && pairs) ? (*ev)[2 * i + 1] : nullptr, o.any_order ? hipExtAnyOrderLaunch : 0);
            else
                e = hipLaunchKernel(o.fn, o.grid, o.block, p->ptrs.data() + o.first_arg, o.lds, os);
        } else {
            void* dst; const void* src;
            std::memcpy(&dst, p->ptrs[o.first_arg], 8);
            std::memcpy(&src, p->ptrs[o.first_arg + 1], 8);
            e = o.kind == jf_plan::MEMSET ? hipMemsetAsync(dst, o.value, (size_t)o.bytes, os)
                                          : hipMemcpyAsync(dst, src, (size_t)o.bytes, hipMemcpyDeviceToHost, os);
        }
        if (e != hipSuccess) rc = JF_ERR_LAUNCH;
        if (ev && pairs && o.kind != jf_plan::LAUNCH && (*ev)[2 * i + 1]) (void)hipEventRecord((*ev)[2 * i + 1], os);
        if (ev && !pairs && (*ev)[i + 1]) (void)hipEventRecord((*ev)[i + 1], st);
    }
    return rc;
}

// debugging aid: the 8-byte words of op `op`'s argument storage -> out[0 .. n) (n returned, at most cap); relocated words carry their CURRENT value
int32_t jf_plan_debug_words(int64_t h, int32_t op, uint64_t* out, int32_t cap) {
    const jf_plan* p = lookup(h);
    if (!p || !p->finalised || op < 0 || op >= (int32_t)p->ops.size() || !out || cap < 0) return JF_ERR_BADARG;
    const jf_plan::Op& o = p->ops[op];
    const size_t begin = p->arg_off[o.first_arg];
    const size_t end = (size_t)(o.first_arg + o.n_args) < p->arg_off.size() ? p->arg_off[o.first_arg + o.n_args] : p->blob.size();
    int32_t n = 0;
    for (size_t off = begin; off + 8 <= end && n < cap; off += 8, ++n) {
        std::memcpy(out + n, p->blob.data() + off, 8);
        for (const auto& r : p->relocs)
            if (r.off == off) out[n] |= (uint64_t)(r.slot + 1) << 56;   // tag: relocated into slot (top byte)
    }
    return n;
}

// on = 0: off; on = n >= 1: events on every n-th replay
int32_t jf_plan_set_timing(int64_t h, int32_t on) {
    jf_plan* p = lookup(h);
    if (!p || !p->finalised || on < 0) return JF_ERR_BADARG;
    p->timing = on != 0;
    p->timing_every = on > 0 ? on : 1;
    p->replay_no = 0;
    return JF_OK;
}

// waits for the events of every timed replay so far (the caller has normally synchronised already), adds their per-op elapsed times to the
// running sums and returns them: ms_sum_per_op[n_ops], *replays = number of timed replays the sums cover.  reset != 0 clears the sums afterwards.
int32_t jf_plan_read_timing(int64_t h, double* ms_sum_per_op, int32_t n_ops, int64_t* replays, int32_t reset) {
    jf_plan* p = lookup(h);
    if (!p || !p->finalised || !ms_sum_per_op || n_ops != (int32_t)p->ops.size()) return JF_ERR_BADARG;
    for (auto& ev : p->pending) {
        bool ok = true;
        for (hipEvent_t e : ev) ok = ok && e != nullptr;
        if (ok) {
            const bool pairs = ev.size() == 2 * p->ops.size() && p->ops.size() > 1;
            for (size_t i = 0; i < p->ops.size(); ++i) {
                float ms = 0.f;
                hipEvent_t a = pairs ? ev[2 * i] : ev[i], b = pairs ? ev[2 * i + 1] : ev[i + 1];
                if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) p->ms_sum[i] += ms;
            }
            ++p->timed_replays;
        }
        for (hipEvent_t e : ev) if (e) p->pool.push_back(e);
    }
    p->pending.clear();
    for (int i = 0; i < n_ops; ++i) ms_sum_per_op[i] = p->ms_sum[i];
    if (replays) *replays = p->timed_replays;
    if (reset) { p->ms_sum.assign(p->ops.size(), 0.0); p->timed_replays = 0; }
    return JF_OK;
}

}  // extern "C"
