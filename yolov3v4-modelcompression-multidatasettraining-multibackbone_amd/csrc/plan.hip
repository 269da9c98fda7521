// Native replay of a recorded launch sequence: one call per forward instead of the reference's
// per-layer Python dispatch (models.py:524-545).  Host-only code; compiled with the kernels so the
// shared library is self-contained.
#include <string.h>

#include <new>
#include <vector>

#include "common.h"

namespace yh {
thread_local AsyncReduce* g_async_reduce = nullptr;
}

namespace {

union AnyDesc {
    yh_conv_desc conv;
    yh_stem_desc stem;
    yh_pool_desc pool;
    yh_copy_desc copy;
    yh_add_desc add;
    yh_decode_desc decode;
    yh_dw_desc dw;
    yh_se_desc se;
    yh_qcopy_desc qcopy;
    yh_qadd_desc qadd;
    yh_bn_desc bn;
    yh_wgrad_desc wgrad;
    yh_resample_desc resample;
    yh_cast_desc cast;
    yh_layout_desc layout;
    yh_pool_bwd_desc pool_bwd;
    yh_pack_batch_desc pack_batch;
    yh_dw_bwd_desc dw_bwd;
    yh_se_bwd_desc se_bwd;
    yh_stem_bwd_desc stem_bwd;
};

struct Fixup {
    int field_offset;
    int slot;
    int64_t byte_offset;
};

struct Op {
    int kind;
    AnyDesc d;
    std::vector<Fixup> fixups;
    int lane = 0;                 // 0: the caller's stream; 1: the plan's side stream (yh_plan_set_lane)
    std::vector<int> deps;        // ops of the OTHER lane this one waits for (yh_plan_add_dep); same-lane order is the stream's
    bool has_dependents = false;  // some op of the other lane waits for this one: record its event
};

size_t desc_size(int kind) {
    switch (kind) {
        case YH_OP_CONV: return sizeof(yh_conv_desc);
        case YH_OP_STEM: return sizeof(yh_stem_desc);
        case YH_OP_POOL: return sizeof(yh_pool_desc);
        case YH_OP_COPY: return sizeof(yh_copy_desc);
        case YH_OP_ADD: return sizeof(yh_add_desc);
        case YH_OP_DECODE: return sizeof(yh_decode_desc);
        case YH_OP_DW: return sizeof(yh_dw_desc);
        case YH_OP_SE: return sizeof(yh_se_desc);
        case YH_OP_QCOPY: return sizeof(yh_qcopy_desc);
        case YH_OP_QPOOL: return sizeof(yh_pool_desc);
        case YH_OP_QADD: return sizeof(yh_qadd_desc);
        case YH_OP_BN_STATS: case YH_OP_BN_FINALIZE: case YH_OP_BN_ACT_FWD: case YH_OP_BN_BWD_REDUCE: case YH_OP_BN_BWD_APPLY:
            return sizeof(yh_bn_desc);
        case YH_OP_WGRAD: case YH_OP_STEM_WGRAD: return sizeof(yh_wgrad_desc);
        case YH_OP_DILATE2: case YH_OP_UPSAMPLE2_BWD: return sizeof(yh_resample_desc);
        case YH_OP_CAST_F32: return sizeof(yh_cast_desc);
        case YH_OP_NCHW_TO_NHWC: return sizeof(yh_layout_desc);
        case YH_OP_POOL_BWD: return sizeof(yh_pool_bwd_desc);
        case YH_OP_PACK_BATCH: return sizeof(yh_pack_batch_desc);
        case YH_OP_DW_WGRAD: case YH_OP_DW_DGRAD: return sizeof(yh_dw_bwd_desc);
        case YH_OP_SE_BWD: return sizeof(yh_se_bwd_desc);
        case YH_OP_STEM_BWD: return sizeof(yh_stem_bwd_desc);
        default: return 0;
    }
}

int launch(int kind, const AnyDesc& d, void* stream) {
    switch (kind) {
        case YH_OP_CONV: return yh_conv2d_fwd(&d.conv, stream);
        case YH_OP_STEM: return yh_conv2d_stem_fwd(&d.stem, stream);
        case YH_OP_POOL: return yh_maxpool2d_fwd(&d.pool, stream);
        case YH_OP_COPY: return yh_copy_channels(&d.copy, stream);
        case YH_OP_ADD: return yh_add_channels(&d.add, stream);
        case YH_OP_DECODE: return yh_yolo_decode(&d.decode, stream);
        case YH_OP_DW: return yh_dwconv2d_fwd(&d.dw, stream);
        case YH_OP_SE: return yh_se_fwd(&d.se, stream);
        case YH_OP_QCOPY: return yh_qcopy(&d.qcopy, stream);
        case YH_OP_QPOOL: return yh_qpool(&d.pool, stream);
        case YH_OP_QADD: return yh_qadd(&d.qadd, stream);
        case YH_OP_BN_STATS: return yh_bn_stats(&d.bn, stream);
        case YH_OP_BN_FINALIZE: return yh_bn_finalize(&d.bn, stream);
        case YH_OP_BN_ACT_FWD: return yh_bn_act_fwd(&d.bn, stream);
        case YH_OP_BN_BWD_REDUCE: return yh_bn_act_bwd_reduce(&d.bn, stream);
        case YH_OP_BN_BWD_APPLY: return yh_bn_act_bwd_apply(&d.bn, stream);
        case YH_OP_WGRAD: return yh_conv2d_wgrad(&d.wgrad, stream);
        case YH_OP_STEM_WGRAD: return yh_stem_wgrad(&d.wgrad, stream);
        case YH_OP_DILATE2: return yh_dilate2(&d.resample, stream);
        case YH_OP_UPSAMPLE2_BWD: return yh_upsample2_bwd(&d.resample, stream);
        case YH_OP_CAST_F32: return yh_cast_f32(&d.cast, stream);
        case YH_OP_POOL_BWD: return yh_maxpool2d_bwd(&d.pool_bwd, stream);
        case YH_OP_PACK_BATCH: return yh_pack_batch(d.pack_batch.items, d.pack_batch.n_items, stream);
        case YH_OP_DW_WGRAD: return yh_dw_wgrad(&d.dw_bwd, stream);
        case YH_OP_DW_DGRAD: return yh_dw_dgrad(&d.dw_bwd, stream);
        case YH_OP_SE_BWD: return yh_se_bwd(&d.se_bwd, stream);
        case YH_OP_STEM_BWD: return yh_stem_bwd(&d.stem_bwd, stream);
        case YH_OP_NCHW_TO_NHWC:
            return yh_nchw_to_nhwc(d.layout.x, d.layout.y, d.layout.n, d.layout.c, d.layout.h, d.layout.w_in, d.layout.c_pad,
                                   d.layout.ldy, d.layout.dtype, stream);
        default: return YH_EINVAL;
    }
}

}  // namespace

struct yh_plan {
    std::vector<Op> ops;
    std::vector<void*> slots;
    // optional per-op HIP-event timing (bench.py roofline leg): events[2*i], events[2*i+1] bracket op i
    bool timing = false;
    std::vector<hipEvent_t> events;
    // optional hipGraph of one replay (launch-bound small batches): valid for the slot pointers bound at capture
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // second lane: ops the planner marked as independent of the main chain (the weight gradients of the training backward) run
    // on this stream, ordered against the main lane by explicit dependencies only; every range joins before it returns
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> done;   // done[i]: recorded after op i when has_dependents (created lazily, timing disabled)
    hipEvent_t join = nullptr;
    // weight-gradient reduce launches on their own stream (yh_plan_set_async_reduce; common.h AsyncReduce)
    bool async_reduce = false;
    yh::AsyncReduce ar;
    void drop_graph() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
        graph_exec = nullptr;
        graph = nullptr;
    }
    ~yh_plan() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
        if (join) (void)hipEventDestroy(join);
        if (side) (void)hipStreamDestroy(side);
        if (ar.main_done) (void)hipEventDestroy(ar.main_done);
        if (ar.red_done) (void)hipEventDestroy(ar.red_done);
        if (ar.side) (void)hipStreamDestroy(ar.side);
        drop_graph();
    }
};

extern "C" int yh_abi_version(void) { return YH_ABI_VERSION; }

extern "C" const char* yh_error_string(int code) {
    switch (code) {
        case YH_OK: return "ok";
        case YH_EINVAL: return "invalid argument";
        case YH_EALIGN: return "misaligned pointer, pitch or channel count";
        case YH_EUNSUPPORTED: return "unsupported configuration";
        case YH_ENOMEM: return "out of host memory";
        case YH_ERANGE: return "index out of range";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" yh_plan* yh_plan_create(void) { return new (std::nothrow) yh_plan(); }

extern "C" void yh_plan_destroy(yh_plan* p) { delete p; }

extern "C" int yh_plan_add(yh_plan* p, int op_kind, const void* desc, int desc_bytes) {
    if (!p || !desc) return YH_EINVAL;
    const size_t want = desc_size(op_kind);
    if (want == 0 || (size_t)desc_bytes != want) return YH_EINVAL;
    Op op;
    op.kind = op_kind;
    memset(&op.d, 0, sizeof(op.d));
    memcpy(&op.d, desc, want);
    try {
        p->ops.push_back(op);
    } catch (...) {
        return YH_ENOMEM;
    }
    return (int)p->ops.size() - 1;
}

extern "C" int yh_plan_add_fixup(yh_plan* p, int op_index, int field_offset, int slot, int64_t byte_offset) {
    if (!p) return YH_EINVAL;
    if (op_index < 0 || op_index >= (int)p->ops.size() || slot < 0 || slot >= 64) return YH_ERANGE;
    Op& op = p->ops[op_index];
    if (field_offset < 0 || (size_t)field_offset + sizeof(void*) > desc_size(op.kind) || field_offset % (int)sizeof(void*)) return YH_ERANGE;
    try {
        op.fixups.push_back(Fixup{field_offset, slot, byte_offset});
        if ((int)p->slots.size() <= slot) p->slots.resize(slot + 1, nullptr);
    } catch (...) {
        return YH_ENOMEM;
    }
    return YH_OK;
}

extern "C" int yh_plan_bind_slot(yh_plan* p, int slot, void* ptr) {
    if (!p) return YH_EINVAL;
    if (slot < 0 || slot >= 64) return YH_ERANGE;
    try {
        if ((int)p->slots.size() <= slot) p->slots.resize(slot + 1, nullptr);
    } catch (...) {
        return YH_ENOMEM;
    }
    p->slots[slot] = ptr;
    return YH_OK;
}

extern "C" int yh_plan_num_ops(const yh_plan* p) { return p ? (int)p->ops.size() : YH_EINVAL; }

extern "C" int yh_plan_set_lane(yh_plan* p, int op_index, int lane) {
    if (!p) return YH_EINVAL;
    if (op_index < 0 || op_index >= (int)p->ops.size() || lane < 0 || lane > 1) return YH_ERANGE;
    p->ops[op_index].lane = lane;
    return YH_OK;
}

extern "C" int yh_plan_add_dep(yh_plan* p, int op_index, int dep_index) {
    if (!p) return YH_EINVAL;
    if (op_index < 0 || op_index >= (int)p->ops.size() || dep_index < 0 || dep_index >= op_index) return YH_ERANGE;
    try {
        p->ops[op_index].deps.push_back(dep_index);
    } catch (...) {
        return YH_ENOMEM;
    }
    p->ops[dep_index].has_dependents = true;
    return YH_OK;
}

extern "C" int yh_plan_run_range(yh_plan* p, int first, int last, void* stream) {
    if (!p) return YH_EINVAL;
    if (first < 0 || last > (int)p->ops.size() || first > last) return YH_ERANGE;
    hipStream_t main_s = (hipStream_t)stream;
    bool side_used = false;
    for (int i = first; i < last; ++i) {
        const Op& op = p->ops[i];
        AnyDesc d = op.d;
        for (const Fixup& f : op.fixups) {
            void* base = p->slots[f.slot];
            if (!base) return YH_EINVAL;                                   // never bound: a bug in the caller
            void* v = base == YH_SLOT_NULL ? nullptr : (void*)((char*)base + f.byte_offset);   // bound to "nothing": optional outputs
            memcpy((char*)&d + f.field_offset, &v, sizeof(void*));
        }
        hipStream_t s = main_s;
        if (op.lane == 1) {
            if (!p->side) {
                const hipError_t e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
                if (e != hipSuccess) return (int)e;
            }
            s = p->side;
            side_used = true;
        }
        for (int dep : op.deps) {
            // dependencies reaching into an earlier range are already satisfied: every range joins its lanes before returning
            if (dep < first || p->ops[dep].lane == op.lane) continue;
            const hipError_t e = hipStreamWaitEvent(s, p->done[dep], 0);
            if (e != hipSuccess) return (int)e;
        }
        if (p->timing) (void)hipEventRecord(p->events[2 * i], s);
        // (per-op timing keeps the reduce launches inside the op's bracket: the numbers of the bench line stay additive)
        const bool ar_on = p->async_reduce && !p->timing && op.kind == YH_OP_WGRAD && op.lane == 0;
        if (ar_on) yh::g_async_reduce = &p->ar;
        const int rc = launch(op.kind, d, s);
        yh::g_async_reduce = nullptr;
        if (p->timing) (void)hipEventRecord(p->events[2 * i + 1], s);
        if (rc != YH_OK) return rc;
        if (op.has_dependents) {
            if (p->done.size() < p->ops.size()) {
                try {
                    p->done.resize(p->ops.size(), nullptr);
                } catch (...) {
                    return YH_ENOMEM;
                }
            }
            if (!p->done[i]) {
                const hipError_t e = hipEventCreateWithFlags(&p->done[i], hipEventDisableTiming);
                if (e != hipSuccess) return (int)e;
            }
            const hipError_t e = hipEventRecord(p->done[i], s);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (p->ar.pending) {      // join: dW is complete for whatever follows on the caller's stream
        const hipError_t e = hipStreamWaitEvent(main_s, p->ar.red_done, 0);
        if (e != hipSuccess) return (int)e;
        p->ar.pending = false;
    }
    if (side_used) {   // join: whatever follows on the caller's stream sees the side lane's results
        if (!p->join) {
            const hipError_t e = hipEventCreateWithFlags(&p->join, hipEventDisableTiming);
            if (e != hipSuccess) return (int)e;
        }
        hipError_t e = hipEventRecord(p->join, p->side);
        if (e == hipSuccess) e = hipStreamWaitEvent(main_s, p->join, 0);
        if (e != hipSuccess) return (int)e;
    }
    return YH_OK;
}

// Weight-gradient reduce launches of this plan on their own stream.  The CALLER guarantees that the plan's weight-gradient ops do not
// share their workspace with any other op (engine/train.py gives them SLOT_WS2).
extern "C" int yh_plan_set_async_reduce(yh_plan* p, int enable) {
    if (!p) return YH_EINVAL;
    if (enable && !p->ar.side) {
        hipError_t e = hipStreamCreateWithFlags(&p->ar.side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ar.main_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ar.red_done, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
    }
    p->async_reduce = enable != 0;
    return YH_OK;
}

extern "C" int yh_plan_set_timing(yh_plan* p, int enable) {
    if (!p) return YH_EINVAL;
    if (enable && p->events.size() < 2 * p->ops.size()) {
        try {
            p->events.reserve(2 * p->ops.size());
        } catch (...) {
            return YH_ENOMEM;
        }
        while (p->events.size() < 2 * p->ops.size()) {
            hipEvent_t e;
            const hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return (int)rc;
            p->events.push_back(e);
        }
    }
    p->timing = enable != 0;
    return YH_OK;
}

extern "C" int yh_plan_get_timings(yh_plan* p, float* ms_out, int n) {
    if (!p || !ms_out) return YH_EINVAL;
    if (n != (int)p->ops.size() || p->events.size() < 2 * p->ops.size()) return YH_ERANGE;
    for (int i = 0; i < n; ++i) {
        const hipError_t rc = hipEventElapsedTime(&ms_out[i], p->events[2 * i], p->events[2 * i + 1]);
        if (rc != hipSuccess) return (int)rc;
    }
    return YH_OK;
}

extern "C" int yh_plan_run(yh_plan* p, void* stream) {
    if (!p) return YH_EINVAL;
    return yh_plan_run_range(p, 0, (int)p->ops.size(), stream);
}


// ---- hipGraph replay -----------------------------------------------------------------------------------
// Captures one replay of the plan (with the currently bound slot pointers) on `stream` into an executable
// graph; yh_plan_graph_launch then costs one graph launch instead of ~100 kernel launches.  The caller keeps
// the slot buffers static between capture and launches (engine/plan.py stages the frame into a fixed buffer).
extern "C" int yh_plan_graph_capture(yh_plan* p, void* stream) {
    if (!p || !stream) return YH_EINVAL;  // capture needs a real (non-null) stream
    p->drop_graph();
    const bool was_timing = p->timing;
    p->timing = false;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) { p->timing = was_timing; return (int)e; }
    const int rc = yh_plan_run_range(p, 0, (int)p->ops.size(), stream);
    e = hipStreamEndCapture(s, &p->graph);
    p->timing = was_timing;
    if (rc != YH_OK) { p->drop_graph(); return rc; }
    if (e != hipSuccess) { p->drop_graph(); return (int)e; }
    e = hipGraphInstantiate(&p->graph_exec, p->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { p->drop_graph(); return (int)e; }
    return YH_OK;
}

extern "C" int yh_plan_graph_launch(yh_plan* p, void* stream) {
    if (!p || !p->graph_exec) return YH_EINVAL;
    const hipError_t e = hipGraphLaunch(p->graph_exec, (hipStream_t)stream);
    return e == hipSuccess ? YH_OK : (int)e;
}

extern "C" int yh_plan_graph_reset(yh_plan* p) {
    if (!p) return YH_EINVAL;
    p->drop_graph();
    return YH_OK;
}
