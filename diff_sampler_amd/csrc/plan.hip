// Native launch plans: the op list of one network evaluation recorded once and walked in C (ds_plan_*), optionally replayed from a
// hipGraph.  Host code only -- every operation is one of the library's own entry points.
#include <cstring>
#include <new>
#include <vector>

#include "ds_common.h"

namespace {

union ArgBlob {
    ds_conv_args conv; ds_gemm_args gemm; ds_norm_args norm; ds_gn_finalize_args fin; ds_attn_args attn;
    ds_layernorm_args ln; ds_geglu_args geglu; ds_noise_embed_args ne; ds_stem_im2col_args stem;
};

struct Node { int op; ArgBlob a; };

size_t arg_size(int op) {
    switch (op) {
        case DS_OP_CONV2D: return sizeof(ds_conv_args);
        case DS_OP_GEMM: return sizeof(ds_gemm_args);
        case DS_OP_GN_STATS: case DS_OP_NORM_ACT: return sizeof(ds_norm_args);
        case DS_OP_GN_FINALIZE: return sizeof(ds_gn_finalize_args);
        case DS_OP_ATTENTION: case DS_OP_ATTENTION_F16: return sizeof(ds_attn_args);
        case DS_OP_LAYERNORM: case DS_OP_LAYERNORM_F16: case DS_OP_LAYERNORM_F16IO: return sizeof(ds_layernorm_args);
        case DS_OP_GEGLU: return sizeof(ds_geglu_args);
        case DS_OP_NOISE_EMBED: return sizeof(ds_noise_embed_args);
        case DS_OP_STEM_IM2COL: return sizeof(ds_stem_im2col_args);
        default: return 0;
    }
}

int issue(const Node& n, void* stream) {
    switch (n.op) {
        case DS_OP_CONV2D: return ds_conv2d_nhwc(&n.a.conv, stream);
        case DS_OP_GEMM: return ds_gemm_nt_batched(&n.a.gemm, stream);
        case DS_OP_GN_STATS: return ds_gn_stats(&n.a.norm, stream);
        case DS_OP_NORM_ACT: return ds_norm_act(&n.a.norm, stream);
        case DS_OP_GN_FINALIZE: return ds_gn_finalize(&n.a.fin, stream);
        case DS_OP_ATTENTION: return ds_attention(&n.a.attn, stream);
        case DS_OP_ATTENTION_F16: return ds_attention_f16(&n.a.attn, stream);
        case DS_OP_LAYERNORM: { const ds_layernorm_args& l = n.a.ln;
            return ds_layernorm_rows(l.x, l.ldx, l.gamma, l.beta, l.eps, l.y, l.ldy, l.rows, l.cols, stream); }
        case DS_OP_LAYERNORM_F16: { const ds_layernorm_args& l = n.a.ln;
            return ds_layernorm_rows_f16(l.x, l.ldx, l.gamma, l.beta, l.eps, l.y, l.ldy, l.rows, l.cols, stream); }
        case DS_OP_LAYERNORM_F16IO: { const ds_layernorm_args& l = n.a.ln;
            return ds_layernorm_rows_f16io(l.x, l.ldx, l.gamma, l.beta, l.eps, l.y, l.ldy, l.rows, l.cols, stream); }
        case DS_OP_GEGLU: { const ds_geglu_args& g = n.a.geglu; return ds_geglu(g.x, g.ldx, g.y, g.ldy, g.rows, g.inner, stream); }
        case DS_OP_NOISE_EMBED: { const ds_noise_embed_args& e = n.a.ne;
            return ds_noise_embed(e.sigma, e.bs, e.freqs, e.nch, e.swap, e.out, e.out_ld, stream); }
        case DS_OP_STEM_IM2COL: { const ds_stem_im2col_args& s = n.a.stem;
            return ds_stem_im2col(s.x, s.sigma, s.sigma_rows, s.sigma_data, s.n, s.c, s.h, s.w, s.out, s.kpad, stream); }
        default: return DS_E_ARG;
    }
}

}  // namespace

struct ds_plan {
    std::vector<Node> ops;
    int last_failed = -1;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

static void drop_graph(ds_plan* p) {
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
}

extern "C" int ds_plan_create(ds_plan** out) {
    if (!out) return DS_E_ARG;
    *out = new (std::nothrow) ds_plan();
    return *out ? DS_OK : DS_E_ARG;
}

extern "C" int ds_plan_add(ds_plan* plan, int op, const void* args, unsigned long long args_bytes) {
    const size_t want = arg_size(op);
    if (!plan || !args || want == 0 || args_bytes != want) return DS_E_ARG;
    Node n;
    std::memset(&n, 0, sizeof(n));
    n.op = op;
    std::memcpy(&n.a, args, want);
    plan->ops.push_back(n);
    drop_graph(plan);                       // a captured graph no longer describes the plan
    return DS_OK;
}

extern "C" int ds_plan_size(const ds_plan* plan) { return plan ? (int)plan->ops.size() : 0; }

extern "C" int ds_plan_run(ds_plan* plan, void* stream) {
    if (!plan) return DS_E_ARG;
    plan->last_failed = -1;
    const size_t n = plan->ops.size();
    for (size_t i = 0; i < n; ++i) {
        const int rc = issue(plan->ops[i], stream);
        if (rc) { plan->last_failed = (int)i; return rc; }
    }
    return DS_OK;
}

extern "C" int ds_plan_last_failed(const ds_plan* plan) { return plan ? plan->last_failed : -1; }

extern "C" int ds_plan_graph_capture(ds_plan* plan, void* stream) {
    if (!plan || !stream) return DS_E_ARG;          // the legacy default stream cannot be captured
    drop_graph(plan);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return (int)e;
    const int rc = ds_plan_run(plan, stream);
    e = hipStreamEndCapture(s, &plan->graph);
    if (rc) { drop_graph(plan); return rc; }
    if (e != hipSuccess) { plan->graph = nullptr; return (int)e; }
    e = hipGraphInstantiate(&plan->exec, plan->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { drop_graph(plan); return (int)e; }
    return DS_OK;
}

extern "C" int ds_plan_graph_launch(ds_plan* plan, void* stream) {
    if (!plan || !plan->exec) return DS_E_ARG;
    const hipError_t e = hipGraphLaunch(plan->exec, (hipStream_t)stream);
    return e == hipSuccess ? DS_OK : (int)e;
}

extern "C" void ds_plan_destroy(ds_plan* plan) {
    if (!plan) return;
    drop_graph(plan);
    delete plan;
}
