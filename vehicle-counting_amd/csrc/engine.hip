// Engine: owns the device, the stream, the detector (YOLOv5 v6.0 graph) and the ReID net, builds their
// execution plans natively and exposes them through the C ABI in include/vcount_hip.h.
//
// Graph sources restated here (independently of oracle/yolov5.py, they only meet at the parameter names):
//   ultralytics/yolov5 v6.0 models/yolov5{s,m,l}.yaml + models/common.py (Conv, Bottleneck, C3, SPPF, Concat) +
//   models/yolo.py (Detect), loaded by /root/reference/networks/yolo.py:58;
//   /root/reference/networks/deepsort/deep/model.py:5-98 (BasicBlock, make_layers, Net(reid=True)).
// Concat never copies: producers write straight into channel slices of the consumer's buffer.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>

#include "engine.h"

namespace vc {

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

int dev_alloc(vc_engine* e, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    VC_HIP(hipMalloc(p, bytes));
    e->allocs.push_back(*p);
    return VC_OK;
}
// Replace *p (an allocation of this engine, or null) by a fresh block of `bytes`; the old block is freed.  The caller makes sure no
// kernel still uses it.  Contents are not carried over.
int dev_realloc(vc_engine* e, void** p, size_t bytes) {
    void* fresh = nullptr;
    VC_HIP(hipMalloc(&fresh, bytes ? bytes : 16));
    bool swapped = false;
    for (void*& q : e->allocs)
        if (*p && q == *p) { (void)hipFree(q); q = fresh; swapped = true; break; }
    if (!swapped) e->allocs.push_back(fresh);
    *p = fresh;
    return VC_OK;
}
int host_alloc(vc_engine* e, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    VC_HIP(hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocPortable));
    e->host_allocs.push_back(*p);
    return VC_OK;
}

ProfScope::ProfScope(vc_engine* e_, int cat_, double flops_, double bytes_, hipStream_t s_)
    : e(e_), cat(cat_), flops(flops_), bytes(bytes_), s(s_ ? s_ : e_->stream) {
    if (e->profiling) hipEventRecord(e->ev0, s);
}
ProfScope::~ProfScope() {
    if (!e->profiling) return;
    hipEventRecord(e->ev1, s);
    hipEventSynchronize(e->ev1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e->ev0, e->ev1);
    ProfCat& c = e->prof[cat];
    c.ms += ms; c.flops += flops; c.bytes += bytes; c.launches += 1;
    c.flops_dense += flops; c.bytes_dense += bytes;
    e->last_ms = ms;
}

// ------------------------------------------------------------------------------------------------ weights
static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static int pack_and_upload(vc_engine* e, ConvParam& p, int prec, float act_scale = 1.0f) {
    VC_CHECK(p.set, VC_ERR_STATE, "parameter '%s' was never set", p.name.c_str());
    p.prec = prec;
    const int ch = prec == PREC_F32 ? 4 : prec == PREC_FP8 ? 16 : 8;
    const int kt = conv_k_tile(prec);
    int kw_eff = p.kw;
    if (p.pair_stem) {                       // two 4-channel pixels per chunk: kernel becomes kh x kw/2 over 8 channels
        p.cin_eff = 8;
        kw_eff = p.kw / 2;
    } else {
        p.cin_eff = round_up(p.I, ch);
    }
    p.K = p.kh * kw_eff * p.cin_eff;
    p.Kp = round_up(p.K, kt);
    p.cout_pad = round_up(p.O, 128);
    std::vector<float> packed((size_t)p.cout_pad * p.Kp, 0.f);
    for (int n = 0; n < p.O; ++n)
        for (int c = 0; c < p.I; ++c)
            for (int r = 0; r < p.kh; ++r)
                for (int s = 0; s < p.kw; ++s) {
                    const float v = p.w[(((size_t)n * p.I + c) * p.kh + r) * p.kw + s];
                    size_t k;
                    if (p.pair_stem) k = ((size_t)r * kw_eff + s / 2) * 8 + (s % 2) * 4 + c;
                    else k = ((size_t)r * p.kw + s) * p.cin_eff + c;
                    packed[(size_t)n * p.Kp + k] = v;
                }
    std::vector<float> bias(p.cout_pad, 0.f);
    for (int n = 0; n < p.O; ++n) bias[n] = p.b[n];
    const size_t nel = packed.size();
    VC_TRY(dev_alloc(e, &p.d_w, nel * elem_size(prec)));
    VC_TRY(dev_alloc(e, (void**)&p.d_b, bias.size() * sizeof(float)));
    if (prec == PREC_F32) {
        VC_HIP(hipMemcpy(p.d_w, packed.data(), nel * 4, hipMemcpyHostToDevice));
    } else if (prec == PREC_FP8) {
        // OCP e4m3fn weights with one scale per output channel (|w|max -> 448); the kernel multiplies the accumulator by
        // weight scale x activation scale in its epilogue (the MFMA's own block scales stay 1)
        std::vector<uint8_t> h(nel);
        std::vector<float> sc(p.cout_pad, 0.f);
        for (int n = 0; n < p.O; ++n) {
            float amax = 0.f;
            for (int k = 0; k < p.Kp; ++k) amax = std::max(amax, std::fabs(packed[(size_t)n * p.Kp + k]));
            const float sw = amax > 0.f ? amax / 448.0f : 1.0f;
            for (int k = 0; k < p.Kp; ++k) h[(size_t)n * p.Kp + k] = f32_to_e4m3(packed[(size_t)n * p.Kp + k] / sw);
            sc[n] = sw * act_scale;
        }
        for (size_t i = (size_t)p.O * p.Kp; i < nel; ++i) h[i] = 0;
        VC_HIP(hipMemcpy(p.d_w, h.data(), nel, hipMemcpyHostToDevice));
        VC_TRY(dev_alloc(e, (void**)&p.d_scale, sc.size() * sizeof(float)));
        VC_HIP(hipMemcpy(p.d_scale, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint16_t> h(nel);
        for (size_t i = 0; i < nel; ++i) h[i] = f32_to_bf16(packed[i]);
        VC_HIP(hipMemcpy(p.d_w, h.data(), nel * 2, hipMemcpyHostToDevice));
    }
    VC_HIP(hipMemcpy(p.d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    return VC_OK;
}

// ------------------------------------------------------------------------------------------------ plan helpers
static View mkview(const View& buf, int B, int H, int W, int C, int co) {
    View v = buf;
    v.B = B; v.H = H; v.W = W; v.C = C; v.co = buf.co + co;
    return v;
}

static int alloc_buf(vc_engine* e, std::map<std::string, View>& m, const std::string& name, size_t pixels, int C, int es) {
    View v{};
    v.cs = C; v.co = 0; v.C = C; v.es = es;
    VC_TRY(dev_alloc(e, &v.ptr, pixels * C * es));
    m[name] = v;
    return VC_OK;
}

struct PlanBuilder {
    vc_engine* e;
    Net* net;
    std::vector<Op>* ops;
    int prec;
    int status = VC_OK;

    // out.H/W are filled from the conv arithmetic; returns the output view with its dims set
    View conv(const std::string& pname, View in, View out, int k, int s, int p, int act, const View* res = nullptr,
              int res_mode = RES_NONE, bool out_f32 = false) {
        auto it = net->index.find(pname);
        if (it == net->index.end()) { set_error("plan: unknown parameter %s", pname.c_str()); status = VC_ERR_NOTFOUND; return out; }
        const ConvParam& cp = net->params[it->second];
        Op op{};
        op.kind = Op::CONV;
        op.param = it->second;
        ConvP& c = op.conv;
        c.in = in.ptr; c.w = cp.d_w; c.bias = cp.d_b; c.out = out.ptr;
        c.B = in.B;
        if (cp.pair_stem) {
            c.H = in.H; c.W = in.W / 2; c.Cin = 8; c.in_cs = 8; c.in_co = 0;
            c.kh = cp.kh; c.kw = cp.kw / 2; c.sh = s; c.sw = 1; c.ph = p; c.pw = 1;
            c.Ho = (in.H + 2 * p - k) / s + 1; c.Wo = (in.W + 2 * p - k) / s + 1;
        } else {
            c.H = in.H; c.W = in.W; c.Cin = cp.cin_eff; c.in_cs = in.cs; c.in_co = in.co;
            c.kh = k; c.kw = k; c.sh = s; c.sw = s; c.ph = p; c.pw = p;
            c.Ho = (in.H + 2 * p - k) / s + 1; c.Wo = (in.W + 2 * p - k) / s + 1;
        }
        c.Cout = cp.O; c.out_cs = out.cs; c.out_co = out.co;
        c.K = cp.K; c.Kp = cp.Kp;
        c.act = act; c.res_mode = res_mode; c.out_f32 = out_f32 ? 1 : 0; c.prec = cp.prec;
        c.scale = cp.d_scale; c.act_scale = e->act_scale; c.inv_act_scale = 1.0f / e->act_scale; c.out_bf16 = 0;
        if (res) { c.res = res->ptr; c.res_cs = res->cs; c.res_co = res->co; }
        c.M = c.B * c.Ho * c.Wo;
        c.cfg = -1;
        op.C = cp.I * cp.kh * cp.kw;          // logical K for algorithmic FLOPs
        ops->push_back(op);
        out.B = in.B; out.H = c.Ho; out.W = c.Wo; out.C = cp.O;
        return out;
    }
};

// ------------------------------------------------------------------------------------------------ YOLOv5 v6.0
static const double kYoloMult[3][2] = {{0.33, 0.50}, {0.67, 0.75}, {1.0, 1.0}};   // depth, width (s, m, l)
static const float kAnchors[3][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};

static void yolo_c3_params(Net& n, int idx, int cin, int cout, int reps) {
    const std::string p = "model." + std::to_string(idx);
    const int h = cout / 2;
    const int a = n.add(p + ".cv1.conv", h, cin, 1, 1);
    const int b = n.add(p + ".cv2.conv", h, cin, 1, 1);
    n.add(p + ".cv3.conv", cout, 2 * h, 1, 1);
    const int f = n.add(p + ".cv12", 2 * h, cin, 1, 1);        // cv1 and cv2 read the same input: one launch, two destinations
    n.params[f].fuse_a = a; n.params[f].fuse_b = b; n.params[f].hidden = true;
    for (int j = 0; j < reps; ++j) {
        n.add(p + ".m." + std::to_string(j) + ".cv1.conv", h, h, 1, 1);
        n.add(p + ".m." + std::to_string(j) + ".cv2.conv", h, h, 3, 3);
    }
}

static void yolo_define(vc_engine* e) {
    const double gd = kYoloMult[e->cfg.yolo_variant][0], gw = kYoloMult[e->cfg.yolo_variant][1];
    const int base[5] = {64, 128, 256, 512, 1024};
    for (int i = 0; i < 5; ++i) e->ch[i] = (int)std::ceil(base[i] * gw / 8.0) * 8;
    const int r[4] = {3, 6, 9, 3};
    for (int i = 0; i < 4; ++i) e->rep[i] = std::max((int)std::nearbyint(r[i] * gd), 1);
    Net& n = e->yolo;
    const int* c = e->ch;
    n.add("model.0.conv", c[0], 3, 6, 6);
    n.params.back().pair_stem = (e->aux_prec == PREC_BF16);       // the stem stays bf16 in the fp8 mode
    n.add("model.1.conv", c[1], c[0], 3, 3); yolo_c3_params(n, 2, c[1], c[1], e->rep[0]);
    n.add("model.3.conv", c[2], c[1], 3, 3); yolo_c3_params(n, 4, c[2], c[2], e->rep[1]);
    n.add("model.5.conv", c[3], c[2], 3, 3); yolo_c3_params(n, 6, c[3], c[3], e->rep[2]);
    n.add("model.7.conv", c[4], c[3], 3, 3); yolo_c3_params(n, 8, c[4], c[4], e->rep[3]);
    n.add("model.9.cv1.conv", c[4] / 2, c[4], 1, 1);
    n.add("model.9.cv2.conv", c[4], c[4] * 2, 1, 1);
    n.add("model.10.conv", c[3], c[4], 1, 1); yolo_c3_params(n, 13, 2 * c[3], c[3], e->rep[0]);
    n.add("model.14.conv", c[2], c[3], 1, 1); yolo_c3_params(n, 17, 2 * c[2], c[2], e->rep[0]);
    n.add("model.18.conv", c[2], c[2], 3, 3); yolo_c3_params(n, 20, 2 * c[2], c[3], e->rep[0]);
    n.add("model.21.conv", c[3], c[3], 3, 3); yolo_c3_params(n, 23, 2 * c[3], c[4], e->rep[0]);
    const int no = 3 * (e->cfg.num_classes + 5);
    for (int i = 0; i < 3; ++i) {
        const int h = n.add("model.24.m." + std::to_string(i), no, c[2 + i], 1, 1);
        const int o = n.add("model.24.m." + std::to_string(i) + ".obj", 8, c[2 + i], 1, 1);      // sparse head: the three objectness rows, padded to 8
        n.params[o].obj_of = h; n.params[o].hidden = true;
    }
}

static int yolo_alloc(vc_engine* e) {
    const int es = elem_size(e->prec), aes = elem_size(e->aux_prec);
    const int S = round_up(e->cfg.img_size, 32);
    const size_t B = e->cfg.max_batch;
    auto px = [&](int stride) { return B * (size_t)(S / stride) * (S / stride); };
    const int* c = e->ch;
    auto& m = e->ybuf;
    auto c3 = [&](int idx, int stride, int cout) -> int {
        const std::string p = "c3_" + std::to_string(idx);
        const int h = cout / 2;
        VC_TRY(alloc_buf(e, m, p + ".a", px(stride), h, es));
        VC_TRY(alloc_buf(e, m, p + ".b", px(stride), h, es));
        VC_TRY(alloc_buf(e, m, p + ".t", px(stride), h, es));
        VC_TRY(alloc_buf(e, m, p + ".cat", px(stride), 2 * h, es));
        return VC_OK;
    };
    VC_TRY(alloc_buf(e, m, "in", px(1), 4, aes));
    VC_TRY(alloc_buf(e, m, "l0", px(2), c[0], aes));
    if (e->prec == PREC_FP8) VC_TRY(alloc_buf(e, m, "l0q", px(2), c[0], es));       // the bf16 stem output converted to fp8
    VC_TRY(alloc_buf(e, m, "l1", px(4), c[1], es)); VC_TRY(c3(2, 4, c[1])); VC_TRY(alloc_buf(e, m, "l2", px(4), c[1], es));
    VC_TRY(alloc_buf(e, m, "l3", px(8), c[2], es)); VC_TRY(c3(4, 8, c[2])); VC_TRY(alloc_buf(e, m, "cat16", px(8), 2 * c[2], es));
    VC_TRY(alloc_buf(e, m, "l5", px(16), c[3], es)); VC_TRY(c3(6, 16, c[3])); VC_TRY(alloc_buf(e, m, "cat12", px(16), 2 * c[3], es));
    VC_TRY(alloc_buf(e, m, "l7", px(32), c[4], es)); VC_TRY(c3(8, 32, c[4])); VC_TRY(alloc_buf(e, m, "l8", px(32), c[4], es));
    VC_TRY(alloc_buf(e, m, "sppcat", px(32), 2 * c[4], es));
    VC_TRY(alloc_buf(e, m, "l9", px(32), c[4], es));
    VC_TRY(alloc_buf(e, m, "cat22", px(32), 2 * c[3], es));
    VC_TRY(c3(13, 16, c[3])); VC_TRY(alloc_buf(e, m, "l13", px(16), c[3], es));
    VC_TRY(alloc_buf(e, m, "cat19", px(16), 2 * c[2], es));
    VC_TRY(c3(17, 8, c[2])); VC_TRY(alloc_buf(e, m, "l17", px(8), c[2], es));
    VC_TRY(c3(20, 16, c[3])); VC_TRY(alloc_buf(e, m, "l20", px(16), c[3], es));
    VC_TRY(c3(23, 32, c[4])); VC_TRY(alloc_buf(e, m, "l23", px(32), c[4], es));
    const int no = 3 * (e->cfg.num_classes + 5);
    const int lcs = round_up(no, 8);
    const int strides[3] = {8, 16, 32};
    for (int i = 0; i < 3; ++i) VC_TRY(dev_alloc(e, (void**)&e->d_logits[i], px(strides[i]) * lcs * sizeof(float)));
    if (e->prec == PREC_BF16) {
        // sparse head: per level an 8-channel objectness plane and room for EVERY pixel of the batch in the gathered list (a
        // pooled B x max_candidates capacity could overflow on one busy frame where the dense head, which only counts anchors with
        // obj x cls > conf against max_candidates per frame, succeeds; 0.6 GB at B = 128 / 640 x 640 of 288 GB buys "cannot overflow")
        for (int i = 0; i < 3; ++i) {
            const size_t ppf = (size_t)(S / strides[i]) * (S / strides[i]);
            e->hc_cap[i] = (int)(B * ppf);
            VC_TRY(dev_alloc(e, &e->d_obj[i], px(strides[i]) * 8 * 2));
            VC_TRY(dev_alloc(e, (void**)&e->d_hc_list[i], (size_t)e->hc_cap[i] * sizeof(int)));
            VC_TRY(dev_alloc(e, &e->d_hc_x[i], (size_t)e->hc_cap[i] * c[2 + i] * 2));
            VC_TRY(dev_alloc(e, &e->d_hc_logits[i], (size_t)e->hc_cap[i] * lcs * 2));
        }
        e->want_hc_count = true;                            // carved out of the zero block below (one memset per pass clears all three)
        VC_TRY(host_alloc(e, (void**)&e->h_hc_ring, (size_t)vc_engine::HC_RING * 4 * sizeof(int)));
        memset(e->h_hc_ring, 0, (size_t)vc_engine::HC_RING * 4 * sizeof(int));
    }
    // post-processing
    const size_t mc = e->cfg.max_candidates, md = e->cfg.max_det;
    auto& pb = e->post;
    VC_TRY(dev_alloc(e, (void**)&pb.cand_box, B * mc * 4 * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&pb.cand_conf, B * mc * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&pb.cand_cls, B * mc * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&pb.cand_idx, B * mc * sizeof(int)));
    // per-pass counters in ONE block: [sparse-head row counts, 64 B][cand_count][overflow] -- a single memset at the head of the detector's
    // chain of dependent launches instead of three
    e->zero_bytes = 64 + 2 * B * sizeof(int);
    VC_TRY(dev_alloc(e, (void**)&e->d_zero, e->zero_bytes));
    if (e->want_hc_count) e->d_hc_count = (int*)e->d_zero;
    pb.cand_count = (int*)((char*)e->d_zero + 64);
    VC_TRY(dev_alloc(e, (void**)&pb.sort_box, B * mc * 4 * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&pb.sort_conf, B * mc * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&pb.sort_cls, B * mc * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&pb.mask, B * mc * (mc / 64) * sizeof(unsigned long long)));
    VC_TRY(dev_alloc(e, (void**)&pb.det, B * md * 6 * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&pb.det_count, B * sizeof(int)));
    pb.overflow = pb.cand_count + B;
    VC_TRY(dev_alloc(e, (void**)&e->d_geom, B * 5 * sizeof(float)));
    VC_TRY(host_alloc(e, (void**)&e->h_geom, 2 * B * 5 * sizeof(float)));
    for (int k = 0; k < 2; ++k) {
        VC_TRY(host_alloc(e, (void**)&e->h_det2[k], B * md * 6 * sizeof(float)));
        VC_TRY(host_alloc(e, (void**)&e->h_det_count2[k], B * sizeof(int)));
    }
    VC_TRY(host_alloc(e, (void**)&e->h_det, B * md * 6 * sizeof(float)));
    VC_TRY(host_alloc(e, (void**)&e->h_det_count, B * sizeof(int)));
    return VC_OK;
}

// C3(c1 -> c2, n x Bottleneck(e=1.0, shortcut), cv3 over concat(m(cv1 x), cv2 x)); models/common.py::C3
static View yolo_c3(PlanBuilder& pb, vc_engine* e, int idx, View x, View out, int cout, int reps, bool shortcut) {
    const std::string p = "model." + std::to_string(idx), bn = "c3_" + std::to_string(idx);
    const int h = cout / 2;
    auto& m = e->ybuf;
    View cat = mkview(m[bn + ".cat"], x.B, x.H, x.W, 2 * h, 0);
    View y = pb.conv(p + ".cv12", x, mkview(m[bn + ".a"], x.B, x.H, x.W, h, 0), 1, 1, 0, ACT_SILU);
    {   // channels [h, 2h) of the fused launch are C3.cv2 -> second half of the concat buffer
        ConvP& c = pb.ops->back().conv;
        View second = mkview(cat, x.B, x.H, x.W, h, h);
        c.out2 = second.ptr; c.out2_cs = second.cs; c.out2_co = second.co; c.split = h;
    }
    y.C = h;
    for (int j = 0; j < reps; ++j) {
        View t = pb.conv(p + ".m." + std::to_string(j) + ".cv1.conv", y, mkview(m[bn + ".t"], x.B, x.H, x.W, h, 0), 1, 1, 0, ACT_SILU);
        View dst = j == reps - 1 ? mkview(cat, x.B, x.H, x.W, h, 0) : mkview(m[bn + ((j % 2 == 0) ? ".b" : ".a")], x.B, x.H, x.W, h, 0);
        y = pb.conv(p + ".m." + std::to_string(j) + ".cv2.conv", t, dst, 3, 1, 1, ACT_SILU, shortcut ? &y : nullptr,
                    shortcut ? RES_AFTER_ACT : RES_NONE);
    }
    return pb.conv(p + ".cv3.conv", cat, out, 1, 1, 0, ACT_SILU);
}

static int yolo_build_ops(vc_engine* e, int B, int Hn, int Wn, std::vector<Op>& ops) {
    PlanBuilder pb{e, &e->yolo, &ops, e->prec};
    auto& m = e->ybuf;
    const int* c = e->ch;
    auto full = [&](const char* name, int stride, int C) { return mkview(m[name], B, Hn / stride, Wn / stride, C, 0); };
    View* lv = e->layer_view;
    View x = full("in", 1, 4);
    lv[0] = x = pb.conv("model.0.conv", x, full("l0", 2, c[0]), 6, 2, 2, ACT_SILU);
    if (e->prec == PREC_FP8) {             // 3-channel stem in bf16 (K = 108), its output quantised once for the fp8 layers
        Op op{}; op.kind = Op::TO_FP8; op.a = x; op.b = full("l0q", 2, c[0]); op.b.B = x.B; op.b.H = x.H; op.b.W = x.W; ops.push_back(op);
        x = op.b;
    }
    lv[1] = x = pb.conv("model.1.conv", x, full("l1", 4, c[1]), 3, 2, 1, ACT_SILU);
    lv[2] = x = yolo_c3(pb, e, 2, x, full("l2", 4, c[1]), c[1], e->rep[0], true);
    lv[3] = x = pb.conv("model.3.conv", x, full("l3", 8, c[2]), 3, 2, 1, ACT_SILU);
    if (pb.status == VC_OK) ops.back().sole_reader_next = 1;      // "l3" is read by C3.cv1 | cv2 of layer 4 (the next op) and by nothing else
    View cat16 = full("cat16", 8, 2 * c[2]);
    lv[4] = x = yolo_c3(pb, e, 4, x, mkview(cat16, B, Hn / 8, Wn / 8, c[2], c[2]), c[2], e->rep[1], true);
    lv[5] = x = pb.conv("model.5.conv", x, full("l5", 16, c[3]), 3, 2, 1, ACT_SILU);
    if (pb.status == VC_OK) ops.back().sole_reader_next = 1;
    View cat12 = full("cat12", 16, 2 * c[3]);
    lv[6] = x = yolo_c3(pb, e, 6, x, mkview(cat12, B, Hn / 16, Wn / 16, c[3], c[3]), c[3], e->rep[2], true);
    lv[7] = x = pb.conv("model.7.conv", x, full("l7", 32, c[4]), 3, 2, 1, ACT_SILU);
    if (pb.status == VC_OK) ops.back().sole_reader_next = 1;
    lv[8] = x = yolo_c3(pb, e, 8, x, full("l8", 32, c[4]), c[4], e->rep[3], true);
    {   // SPPF (models/common.py::SPPF, k=5)
        View sc = full("sppcat", 32, 2 * c[4]);
        pb.conv("model.9.cv1.conv", x, mkview(sc, B, Hn / 32, Wn / 32, c[4] / 2, 0), 1, 1, 0, ACT_SILU);
        Op op{}; op.kind = Op::SPPF; op.a = sc; op.C = c[4] / 2; ops.push_back(op);
        lv[9] = x = pb.conv("model.9.cv2.conv", sc, full("l9", 32, c[4]), 1, 1, 0, ACT_SILU);
    }
    View cat22 = full("cat22", 32, 2 * c[3]);
    lv[10] = x = pb.conv("model.10.conv", x, mkview(cat22, B, Hn / 32, Wn / 32, c[3], c[3]), 1, 1, 0, ACT_SILU);
    { Op op{}; op.kind = Op::UPSAMPLE; op.a = x; op.b = mkview(cat12, B, Hn / 16, Wn / 16, c[3], 0); ops.push_back(op); lv[11] = op.b; }
    lv[12] = cat12;
    lv[13] = x = yolo_c3(pb, e, 13, cat12, full("l13", 16, c[3]), c[3], e->rep[0], false);
    View cat19 = full("cat19", 16, 2 * c[2]);
    lv[14] = x = pb.conv("model.14.conv", x, mkview(cat19, B, Hn / 16, Wn / 16, c[2], c[2]), 1, 1, 0, ACT_SILU);
    { Op op{}; op.kind = Op::UPSAMPLE; op.a = x; op.b = mkview(cat16, B, Hn / 8, Wn / 8, c[2], 0); ops.push_back(op); lv[15] = op.b; }
    lv[16] = cat16;
    // Detect.m[i]: 1x1 conv + bias (models/yolo.py::Detect).  fp32 mode: fp32 logits.  bf16 mode: bf16 logits like every other
    // activation, and the launch covers round_up(no, 8) output channels (the packed weight / bias rows past `no` are zero) so
    // that the 255-channel head takes the 16-byte-store epilogue; FLOPs are still counted for `no` channels (Op::cout_logical).
    const int no = 3 * (e->cfg.num_classes + 5), lcs = round_up(no, 8);
    e->sparse_pass = e->prec == PREC_BF16 && e->opt.sparse_head && !e->want_pred_debug && e->d_hc_count;
    // sparse Detect head (detect_post.hip): objectness conv over every pixel -> gather the pixels that can pass conf_thres -> the
    // full head on the gathered rows (row count on the device).  The work reported is what RUNS (the gathered rows); the dense
    // head's figures travel beside it as dense_* (round 3 credited the launch with the dense head's work: a 6 us launch showed
    // 3.4 x the chip's peak in the per-layer table).  The ops of level i follow the layer that produces its map: P3's and P4's are
    // marked Op::side and run on the head stream beside layers 18 - 23 (three short dependent launches per level, 0.11 ms of
    // objectness conv alone at 80^2, that used to sit at the END of the detector's chain of ~63 dependent launches).
    auto sparse_head = [&](int i, const View& x) {
        if (!e->sparse_pass || pb.status != VC_OK) return;
        const size_t first = ops.size();
        const std::string hp = "model.24.m." + std::to_string(i);
        const double es = 2.0, Mh = (double)x.B * x.H * x.W;
        View ov{}; ov.ptr = e->d_obj[i]; ov.cs = 8; ov.co = 0;
        pb.conv(hp + ".obj", x, ov, 1, 1, 0, ACT_NONE);
        if (pb.status != VC_OK) return;
        ops.back().flops_override = 2.0 * Mh * 3 * x.C;                                    // the three objectness rows (the launch covers 8 zero-padded channels)
        ops.back().bytes_override = (Mh * x.C + 8.0 * x.C) * es + Mh * 8 * es;
        ops.back().dense_flops = 0; ops.back().dense_bytes = 0;                            // the dense head has no such launch: its work is on the head conv below
        { Op op{}; op.kind = Op::HEAD_COMPACT; op.level = i; op.a = x; ops.push_back(op); }
        View gx{}; gx.ptr = e->d_hc_x[i]; gx.B = 1; gx.H = 1; gx.W = e->hc_cap[i]; gx.C = x.C; gx.cs = x.C; gx.co = 0;
        View go{}; go.ptr = e->d_hc_logits[i]; go.cs = lcs; go.co = 0;
        pb.conv(hp, gx, go, 1, 1, 0, ACT_NONE);
        if (pb.status != VC_OK) return;
        Op& hop = ops.back();
        hop.cout_logical = hop.conv.Cout; hop.conv.Cout = lcs;
        hop.conv.m_dev = e->d_hc_count + i;
        // executed: the head on the gathered rows only (row count on the device, read back after the pass)
        hop.rows_level = i;
        hop.flops_override = 0; hop.bytes_override = (double)no * x.C * es;                // weights
        hop.flops_per_row = 2.0 * no * x.C; hop.bytes_per_row = (x.C + (double)lcs) * es;  // one gathered feature row in, one logit row out
        hop.dense_flops = 2.0 * Mh * no * x.C;
        hop.dense_bytes = (Mh * x.C + (double)no * x.C) * es + Mh * lcs * es;              // the dense head's in + weights + out
        if (i < 2) for (size_t j = first; j < ops.size(); ++j) ops[j].side = 1;
    };
    View p3 = lv[17] = yolo_c3(pb, e, 17, cat16, full("l17", 8, c[2]), c[2], e->rep[0], false);
    sparse_head(0, p3);
    lv[18] = pb.conv("model.18.conv", p3, mkview(cat19, B, Hn / 16, Wn / 16, c[2], 0), 3, 2, 1, ACT_SILU);
    lv[19] = cat19;
    View p4 = lv[20] = yolo_c3(pb, e, 20, cat19, full("l20", 16, c[3]), c[3], e->rep[0], false);
    sparse_head(1, p4);
    lv[21] = pb.conv("model.21.conv", p4, mkview(cat22, B, Hn / 32, Wn / 32, c[3], 0), 3, 2, 1, ACT_SILU);
    lv[22] = cat22;
    View p5 = lv[23] = yolo_c3(pb, e, 23, cat22, full("l23", 32, c[4]), c[4], e->rep[0], false);
    sparse_head(2, p5);
    const View heads[3] = {p3, p4, p5};
    for (int i = 0; i < 3 && !e->sparse_pass; ++i) {
        View o{}; o.ptr = e->d_logits[i]; o.cs = lcs; o.co = 0;
        const bool wide = e->prec == PREC_F32;
        pb.conv("model.24.m." + std::to_string(i), heads[i], o, 1, 1, 0, ACT_NONE, nullptr, RES_NONE, wide);
        if (!wide && pb.status == VC_OK) { Op& op = ops.back(); op.cout_logical = op.conv.Cout; op.conv.Cout = lcs; }
        if (e->prec == PREC_FP8 && pb.status == VC_OK) ops.back().conv.out_bf16 = 1;        // logits leave the fp8 domain as bf16
    }
    return pb.status;
}

// Pick the fastest tile configuration for a conv launch by timing every candidate once per (layer shape, problem-size
// bucket).  Results are identical across the implicit-GEMM configurations (same K order, fp32 accumulate); the halo-staged 3x3
// variants sum the K tiles slice-major and can differ from them in the last bf16 bit of a few values.
// VC_TUNE_CACHE=<file> persists the choices across processes (used for clean rocprofv3 passes: a first run writes the file, the
// profiled run reads it and launches no tuning candidates; also what makes a bf16 engine bit-reproducible across processes).
// Within a process the choices are shared by all engines of a device (g_tuned): a second engine neither re-times the candidates
// nor picks the other member of a near-tie, so two engines of one process agree bit for bit.
static std::string tune_key(const ConvP& c) {
    int bucket = 1;
    while (bucket < c.M) bucket <<= 1;
    char k[160];
    snprintf(k, sizeof(k), "p%d_ci%d_co%d_k%dx%d_s%d_h%d_w%d_K%d_m%d_sp%d%s", c.prec, c.Cin, c.Cout, c.kh, c.kw, c.sh, c.H, c.W, c.K, bucket, c.split,
             c.in_up ? "_up" : "");      // (the upsample fold-in runs on a subset of the tile configurations)
    return k;
}

static std::mutex g_tune_mu;
static std::map<std::string, int> g_tuned;          // "d<device>_<tune_key>" -> tile configuration

static void tune_cache_load(vc_engine* e) {
    const char* path = getenv("VC_TUNE_CACHE");
    if (!path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    char k[200];
    int cfg;
    while (fscanf(f, "%199s %d", k, &cfg) == 2) e->tuned[k] = cfg;
    fclose(f);
}

static void tune_cache_save(vc_engine* e) {
    const char* path = getenv("VC_TUNE_CACHE");
    if (!path || !e->tuned_dirty) return;
    FILE* f = fopen(path, "w");
    if (!f) return;
    for (const auto& kv : e->tuned) fprintf(f, "%s %d\n", kv.first.c_str(), kv.second);
    fclose(f);
}

static int tuned_cfg(vc_engine* e, const ConvP& c, hipStream_t s) {
    static const bool enabled = !(getenv("VC_AUTOTUNE") && atoi(getenv("VC_AUTOTUNE")) == 0);
    if (!enabled) return -1;
    const std::string key = tune_key(c);
    auto it = e->tuned.find(key);
    if (it != e->tuned.end()) return it->second;
    const std::string gkey = "d" + std::to_string(e->cfg.device) + "_" + key;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto g = g_tuned.find(gkey);
        if (g != g_tuned.end()) { e->tuned[key] = g->second; e->tuned_dirty = true; return g->second; }
    }
    int best = -1;
    float best_ms = 1e30f;
    static const bool stream_on = getenv("VC_CONV_STREAM") && atoi(getenv("VC_CONV_STREAM")) != 0;   // conv1x1_stream_kernel: wins alone, loses beside the ReID queue (conv_igemm.hip)
    for (int cfg = 0; cfg < conv_num_cfgs(); ++cfg) {
        if (conv_stream_cfg(cfg) && !stream_on) continue;
        if (launch_conv_cfg(c, cfg, s) != VC_OK) continue;          // warm-up (instruction cache, L2)
        float tmin = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e->ev0, s);
            launch_conv_cfg(c, cfg, s);
            hipEventRecord(e->ev1, s);
            hipEventSynchronize(e->ev1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e->ev0, e->ev1);
            tmin = std::min(tmin, ms);
        }
        static const bool tlog = getenv("VC_TUNE_LOG") != nullptr;      // diagnostics: every candidate's time
        if (tlog) fprintf(stderr, "[vc tune] %s cfg %d %.4f ms\n", key.c_str(), cfg, tmin);
        if (tmin < best_ms) { best_ms = tmin; best = cfg; }
    }
    e->tuned[key] = best;
    e->tuned_dirty = true;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tuned.emplace(gkey, best);
    }
    return best;
}

// in-flight profiling of one conv launch (see vc_engine::prof_async): the launch itself reports its start / stop timestamps
// into an event pair (hipExtLaunchKernel), no host wait
static void conv_timer_arm(vc_engine* e, ConvP& cp, double flops, double bytes, const Op& op) {
    if (!e->prof_async || e->profiling || e->prof_used >= e->prof_pairs.size()) return;
    vc_engine::ProfPair& pp = e->prof_pairs[e->prof_used++];
    pp.flops = flops; pp.bytes = bytes;
    pp.rows = op.rows_level >= 0 && e->h_hc_ring ? e->h_hc_ring + (size_t)e->hc_ring_cur * 4 + op.rows_level : nullptr;
    pp.flops_per_row = op.flops_per_row; pp.bytes_per_row = op.bytes_per_row;
    pp.flops_dense = op.dense_flops >= 0 ? op.dense_flops : -1; pp.bytes_dense = op.dense_bytes >= 0 ? op.dense_bytes : -1;   // -1: same as executed
    cp.ev_start = pp.a; cp.ev_stop = pp.b;
}

static int run_ops_body(vc_engine* e, std::vector<Op>& ops, int aux_cat, hipStream_t s_main, bool& side_used);
// *side_used (nullable) = the ops forked onto the head stream.  The join (head stream -> main stream) is recorded HERE, on the error path too:
// a pass that failed after the fork must not leave head ops running beside the next pass's memset of the counters they write.
static int run_ops(vc_engine* e, std::vector<Op>& ops, int aux_cat, hipStream_t s_main, bool* side_used = nullptr) {
    bool forked = false;
    const int st = run_ops_body(e, ops, aux_cat, s_main, forked);
    if (forked) {                                            // the P3 / P4 head ops ran on the head stream: whatever follows on the main stream waits for them
        const hipError_t e1 = hipEventRecord(e->ev_join, e->hstream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s_main, e->ev_join, 0) : e1;
        if (st == VC_OK && e2 != hipSuccess) { set_error("head stream join failed: %s", hipGetErrorString(e2)); return VC_ERR_HIP; }
    }
    if (side_used) *side_used = forked;
    return st;
}
static int run_ops_body(vc_engine* e, std::vector<Op>& ops, int aux_cat, hipStream_t s_main, bool& side_used) {
    // Op::side ops go to the head stream (detector passes only; not while every launch is bracketed by blocking events): the first of a run
    // of them waits for everything the main stream has been given so far, the caller joins the head stream before it reads their results
    const bool side_ok = s_main == e->dstream && e->hstream && e->opt.head_side && !e->profiling;
    bool prev_side = false;
    const Op* pending_up = nullptr;          // an UPSAMPLE op whose consumer (the next op) reads the half-size map itself
    for (size_t oi = 0; oi < ops.size(); ++oi) {
        Op& op = ops[oi];
        const bool on_side = side_ok && op.side;
        if (on_side && !prev_side) {
            VC_HIP(hipEventRecord(e->ev_fork, s_main));
            VC_HIP(hipStreamWaitEvent(e->hstream, e->ev_fork, 0));
            side_used = true;
        }
        prev_side = on_side;
        hipStream_t s = on_side ? e->hstream : s_main;
        switch (op.kind) {
            case Op::CONV: {
                const double es = elem_size(op.conv.prec);
                const double fl = op.flops_override >= 0 ? op.flops_override : 2.0 * op.conv.M * (double)(op.cout_logical ? op.cout_logical : op.conv.Cout) * op.C;
                double by = op.bytes_override >= 0 ? op.bytes_override
                                                   : ((double)op.conv.B * op.conv.H * op.conv.W * op.conv.Cin + (double)op.conv.Cout * op.conv.K) * es +
                                                         (double)op.conv.M * op.conv.Cout * (op.conv.out_f32 ? 4 : es);
                if (pending_up) by -= 0.75 * (double)op.conv.M * pending_up->a.C * es;   // (folded upsample: three quarters of that half of the input are never read)
                ConvP cp = op.conv;
                if (pending_up) {                                             // Upsample + Concat folded into this pointwise conv (conv_igemm_kernel<..., UP>)
                    const View& u = pending_up->a;
                    cp.in_up = u.ptr; cp.up_C = u.C; cp.up_cs = u.cs; cp.up_co = u.co;
                    pending_up = nullptr;
                }
                double fl_exec = -1;                                          // executed FLOPs when they differ from `fl` (device-side row count)
                static const bool stem_direct_on = !(getenv("VC_STEM_DIRECT") && atoi(getenv("VC_STEM_DIRECT")) == 0);
                static const bool reid_stem_on = !(getenv("VC_REID_STEM_FUSED") && atoi(getenv("VC_REID_STEM_FUSED")) == 0);
                const Op* nx = oi + 1 < ops.size() ? &ops[oi + 1] : nullptr;
                conv_timer_arm(e, cp, fl, by, op);
                const int front_fused_mode = e->opt.front_fused;   // 0 off, 1 stream path, 2 always
                const bool front_ok = stem_direct_on && front_fused_mode > 0 && nx && nx->kind == Op::CONV && front_fused_applicable(cp, nx->conv);
                bool u8_src = e->stem_src && cp.in == e->ybuf["in"].ptr;   // the letterbox was skipped for this pass (run_detector_dev)
                if (u8_src && !(stem_direct_on && (stem_u8_applicable(cp, e->stem_geom) || (front_ok && front_fused_resize_ok(e->stem_geom))))) {
                    // run_detector_dev decided the fold-in from the engine's widths, this op list decides which kernel runs: should they ever
                    // disagree (a plan or layout change), the letterbox kernel runs now and the pass goes on from its tensor
                    ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, s);
                    VC_TRY(launch_letterbox(e->stem_src, e->ybuf["in"].ptr, cp.B, e->stem_geom, e->aux_prec, s));
                    e->stem_src = nullptr; e->in_stale = false;
                    u8_src = false;
                }
                const bool fuse_front = front_ok && (front_fused_mode == 2 || u8_src);
                const bool c3_fused_on = e->opt.c3_fused != 0;
                const bool fuse_c3 = c3_fused_on && oi + 3 < ops.size() && ops[oi + 1].kind == Op::CONV && ops[oi + 2].kind == Op::CONV && ops[oi + 3].kind == Op::CONV &&
                                     c3_fused_applicable(cp, ops[oi + 1].conv, ops[oi + 2].conv, ops[oi + 3].conv);
                if (fuse_c3) {                                                // the first C3 block in one kernel (c3_fused.hip): y1, y2, b1, m stay in LDS
                    double flc = fl;
                    for (int j = 1; j <= 3; ++j) flc += 2.0 * ops[oi + j].conv.M * (double)ops[oi + j].conv.Cout * ops[oi + j].C;
                    const ConvP& o3 = ops[oi + 3].conv;
                    double wbytes = 0;
                    for (int j = 0; j <= 3; ++j) wbytes += (double)ops[oi + j].conv.Cout * ops[oi + j].conv.K * es;
                    const double byc = (double)cp.B * cp.H * cp.W * cp.Cin * es + wbytes + (double)o3.M * o3.Cout * es;
                    cp.cfg = 103; cp.ablate = e->opt.c3_ablate;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = flc; e->prof_pairs[e->prof_used - 1].bytes = byc; }   // the pair armed above times all four layers
                    {
                        ProfScope ps(e, VC_PROF_CONV, flc, byc, s);
                        VC_TRY(launch_c3_fused(cp, ops[oi + 1].conv, ops[oi + 2].conv, ops[oi + 3].conv, s));
                    }
                    if (e->profiling && e->op_log.size() < (1u << 20)) {
                        char line[256];
                        snprintf(line, sizeof(line), "conv M=%d N=%d K=%d k=1x1 s=1 cfg=103 ms=%.4f tflops=%.1f\n", cp.M, 64, 64, e->last_ms, flc / (e->last_ms * 1e-3) / 1e12);
                        e->op_log += line;
                    }
                    oi += 3;                                                  // m.cv1, m.cv2 and cv3 are done
                    break;
                }
                if (e->opt.reid_block_fused != 0 && nx && nx->kind == Op::CONV && reid_block_fused_applicable(cp, nx->conv)) {   // a 64-channel ReID BasicBlock in one kernel (reid_block_fused.hip)
                    const ConvP& o2 = nx->conv;
                    const double flb = fl + 2.0 * o2.M * (double)o2.Cout * nx->C;
                    const double byb = ((double)cp.B * cp.H * cp.W * cp.Cin + (double)cp.Cout * cp.K + (double)o2.Cout * o2.K) * es + (double)o2.M * o2.Cout * es;   // x in, both weights, y out
                    cp.cfg = 106;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = flb; e->prof_pairs[e->prof_used - 1].bytes = byb; }
                    {
                        ProfScope ps(e, VC_PROF_CONV, flb, byb, s);
                        VC_TRY(launch_reid_block_fused(cp, o2, s));
                    }
                    if (e->profiling && e->op_log.size() < (1u << 20)) {
                        char line[256];
                        snprintf(line, sizeof(line), "conv M=%d N=%d K=%d k=3x3 s=1 cfg=106 ms=%.4f tflops=%.1f\n", cp.M, 64, 1152, e->last_ms, flb / (e->last_ms * 1e-3) / 1e12);
                        e->op_log += line;
                    }
                    ++oi;                                                     // conv2 is done
                    break;
                }
                const bool bneck_fused_on = e->opt.bneck_fused != 0;
                if (bneck_fused_on && e->opt.bneck_cv3 != 0 && oi + 2 < ops.size() && ops[oi + 1].kind == Op::CONV && ops[oi + 2].kind == Op::CONV &&
                    bneck_cv3_fused_applicable(cp, ops[oi + 1].conv, ops[oi + 2].conv)) {   // the last 64-channel Bottleneck of a C3 + the block's cv3 in one kernel
                    const ConvP &o2 = ops[oi + 1].conv, &o3 = ops[oi + 2].conv;
                    const double flb = fl + 2.0 * o2.M * (double)o2.Cout * ops[oi + 1].C + 2.0 * o3.M * (double)o3.Cout * ops[oi + 2].C;
                    const double byb = ((double)cp.B * cp.H * cp.W * (cp.Cin + 64) + (double)cp.Cout * cp.K + (double)o2.Cout * o2.K + (double)o3.Cout * o3.K) * es + (double)o3.M * o3.Cout * es;
                    cp.cfg = 105;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = flb; e->prof_pairs[e->prof_used - 1].bytes = byb; }
                    {
                        ProfScope ps(e, VC_PROF_CONV, flb, byb, s);
                        VC_TRY(launch_bneck_cv3_fused(cp, o2, o3, s));
                    }
                    if (e->profiling && e->op_log.size() < (1u << 20)) {
                        char line[256];
                        snprintf(line, sizeof(line), "conv M=%d N=%d K=%d k=3x3 s=1 cfg=105 ms=%.4f tflops=%.1f\n", cp.M, 128, 768, e->last_ms, flb / (e->last_ms * 1e-3) / 1e12);
                        e->op_log += line;
                    }
                    oi += 2;                                                  // the 3x3 conv and cv3 are done
                    break;
                }
                if (bneck_fused_on && nx && nx->kind == Op::CONV && bneck_fused_applicable(cp, nx->conv)) {   // 64-channel Bottleneck in one kernel (bneck_fused.hip)
                    const ConvP& o2 = nx->conv;
                    const double flb = fl + 2.0 * o2.M * (double)o2.Cout * nx->C;
                    const double byb = (double)cp.B * cp.H * cp.W * cp.Cin * es * (o2.res_mode != RES_NONE ? 1.0 : 1.0) + ((double)cp.Cout * cp.K + (double)o2.Cout * o2.K) * es +
                                       (double)o2.M * o2.Cout * es;
                    cp.cfg = 104;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = flb; e->prof_pairs[e->prof_used - 1].bytes = byb; }
                    {
                        ProfScope ps(e, VC_PROF_CONV, flb, byb, s);
                        VC_TRY(launch_bneck_fused(cp, o2, s));
                    }
                    if (e->profiling && e->op_log.size() < (1u << 20)) {
                        char line[256];
                        snprintf(line, sizeof(line), "conv M=%d N=%d K=%d k=3x3 s=1 cfg=104 ms=%.4f tflops=%.1f\n", cp.M, 64, 640, e->last_ms, flb / (e->last_ms * 1e-3) / 1e12);
                        e->op_log += line;
                    }
                    ++oi;                                                     // the 3x3 conv is done
                    break;
                }
                if (e->opt.fuse_s2_pw != 0 && op.sole_reader_next && nx && nx->kind == Op::CONV && s2halo_pw_applicable(cp, nx->conv)) {
                    // a 3x3 / s2 conv and the pointwise conv that alone reads it in one launch (conv3x3s2_halo_kernel<..., F2>): YOLOv5s layer 3 + C3.cv1 | cv2 of layer 4
                    const ConvP& o2 = nx->conv;
                    const double flb = fl + 2.0 * o2.M * (double)o2.Cout * nx->C;
                    const double byb = ((double)cp.B * cp.H * cp.W * cp.Cin + (double)cp.Cout * cp.K + (double)o2.Cout * o2.K) * es + (double)o2.M * o2.Cout * es;   // x in, both weights, cv1 | cv2 out
                    cp.cfg = 107;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = flb; e->prof_pairs[e->prof_used - 1].bytes = byb; }
                    {
                        ProfScope ps(e, VC_PROF_CONV, flb, byb, s);
                        VC_TRY(launch_s2halo_pw(cp, o2, s));
                    }
                    if (e->profiling && e->op_log.size() < (1u << 20)) {
                        char line[256];
                        snprintf(line, sizeof(line), "conv M=%d N=%d K=%d k=3x3 s=2 cfg=107 ms=%.4f tflops=%.1f\n", cp.M, 128, cp.K + 128, e->last_ms, flb / (e->last_ms * 1e-3) / 1e12);
                        e->op_log += line;
                    }
                    static const bool also_store = getenv("VC_S2PW_STORE") && atoi(getenv("VC_S2PW_STORE")) != 0;   // diagnostics: the fused kernel also stores the 3x3's output
                    if (!also_store) e->s2pw_folded.push_back(op.conv);                        // its own output stays unwritten (vc_detect_debug_layer produces it on demand)
                    ++oi;                                                     // the pointwise conv is done
                    break;
                }
                if (fuse_front) {                                             // YOLO layers 0 + 1 in one kernel (front_fused.hip): layer 0 never reaches HBM
                    const Op& o1 = *nx;
                    const double fl1 = 2.0 * o1.conv.M * (double)o1.conv.Cout * o1.C;
                    const double by01 = ((double)cp.B * cp.H * cp.W * cp.Cin + (double)cp.Cout * cp.K + (double)o1.conv.Cout * o1.conv.K) * es + (double)o1.conv.M * o1.conv.Cout * es;
                    cp.cfg = 102; cp.ablate = e->opt.ff_ablate;
                    if (cp.ev_start && e->prof_used > 0) { e->prof_pairs[e->prof_used - 1].flops = fl + fl1; e->prof_pairs[e->prof_used - 1].bytes = by01; }   // the pair armed above now times both layers
                    ProfScope ps(e, VC_PROF_CONV, fl + fl1, by01, s);
                    VC_TRY(launch_front_fused(cp, o1.conv, u8_src ? e->stem_src : nullptr, e->stem_geom, s));
                    e->l0_stale = true;                                       // layer 0 lived in LDS only
                    ++oi;                                                     // the 3x3 conv is done
                } else if (stem_direct_on && stem_direct_applicable(cp)) {   // YOLO stem, bf16: direct convolution (stem_direct.hip)
                    cp.cfg = 100;
                    ProfScope ps(e, VC_PROF_CONV, fl, by, s);
                    // fp8 engine: the stem also writes the e4m3 copy the next layer reads (TO_FP8 of its own output, folded in)
                    const View* q8 = nx && nx->kind == Op::TO_FP8 && nx->a.ptr == cp.out && nx->a.co == cp.out_co ? &nx->b : nullptr;
                    if (u8_src) VC_TRY(launch_stem_direct_u8(cp, e->stem_src, e->stem_geom, s, q8, 1.0f / e->act_scale));   // letterbox folded in
                    else VC_TRY(launch_stem_direct(cp, s, q8, 1.0f / e->act_scale));
                    if (q8) ++oi;                                            // the conversion op is done
                } else if (reid_stem_on && nx && nx->kind == Op::MAXPOOL && nx->a.ptr == cp.out && reid_stem_applicable(cp, nx->b.cs, nx->b.co)) {
                    cp.cfg = 101;                                            // ReID stem: conv + ReLU + MaxPool in one kernel (reid_stem.hip)
                    ProfScope ps(e, VC_PROF_CONV, fl, by, s);
                    VC_TRY(launch_reid_stem_pool(cp, nx->b.ptr, s));
                    ++oi;                                                    // the pool op is done
                } else {
                    if (op.tuned == -2) op.tuned = cp.m_dev ? -1 : tuned_cfg(e, cp, s);   // a device-side row count: the implicit-GEMM heuristic (the only family that honours it)
                    cp.cfg = op.tuned;
                    {
                        ProfScope ps(e, VC_PROF_CONV, fl, by, s);
                        VC_TRY(launch_conv(cp, s));
                    }
                    if (e->profiling && (op.rows_level >= 0 || op.dense_flops >= 0)) {   // blocking profile: the scope has synchronised
                        ProfCat& pc = e->prof[VC_PROF_CONV];
                        double fe = fl, be = by;
                        if (op.rows_level >= 0) {                            // executed work = fixed part + gathered rows x per-row work
                            int rows = 0;
                            VC_HIP(hipMemcpy(&rows, e->d_hc_count + op.rows_level, sizeof(int), hipMemcpyDeviceToHost));
                            rows = std::min(rows, e->hc_cap[op.rows_level]);
                            fe += rows * op.flops_per_row; be += rows * op.bytes_per_row;
                            pc.flops += fe - fl; pc.bytes += be - by;
                            fl_exec = fe;
                        }
                        // the scope credited (fl, by) to the dense-equivalent accumulators as well: replace them by the dense head's figures
                        pc.flops_dense += (op.dense_flops >= 0 ? op.dense_flops : fe) - fl;
                        pc.bytes_dense += (op.dense_bytes >= 0 ? op.dense_bytes : be) - by;
                    }
                }
                if (e->profiling && e->op_log.size() < (1u << 20)) {
                    char line[256];
                    const ConvP& c = op.conv;
                    snprintf(line, sizeof(line), "%s M=%d N=%d K=%d k=%dx%d s=%d cfg=%d ms=%.4f tflops=%.1f\n", c.M > 0 ? "conv" : "?", c.M, c.Cout, c.K,
                             c.kh, c.kw, c.sh, cp.cfg, e->last_ms, (fl_exec >= 0 ? fl_exec : fl) / (e->last_ms * 1e-3) / 1e12);
                    e->op_log += line;
                }
                break;
            }
            case Op::SPPF: { ProfScope ps(e, aux_cat, 0, 0, s); VC_TRY(launch_sppf_pool(op.a, op.C, e->prec, e->opt.sppf_sep, s)); break; }
            case Op::UPSAMPLE: {
                // bf16: when the consumer is the pointwise conv over [upsampled | skip] (C3.cv1 | cv2 of layers 13 / 17) it reads the half-size
                // map itself; the slice of the concat buffer stays unwritten (vc_detect_debug_layer produces it on demand)
                const Op* nx = oi + 1 < ops.size() ? &ops[oi + 1] : nullptr;
                const bool fold = e->opt.fuse_upsample && e->prec == PREC_BF16 && nx && nx->kind == Op::CONV && nx->conv.prec == PREC_BF16 && nx->conv.kh == 1 &&
                                  nx->conv.kw == 1 && nx->conv.sh == 1 && nx->conv.ph == 0 && !nx->conv.m_dev && nx->conv.in == op.b.ptr && nx->conv.in_co == op.b.co &&
                                  nx->conv.in_cs == op.b.cs && nx->conv.H == 2 * op.a.H && nx->conv.W == 2 * op.a.W && op.a.C % 64 == 0 && nx->conv.Cin % 64 == 0 &&
                                  op.a.C < nx->conv.Cin && op.a.cs % 8 == 0 && op.a.co % 8 == 0;
                if (fold) { pending_up = &op; e->up_folded.push_back({op.a, op.b}); break; }
                ProfScope ps(e, aux_cat, 0, 0, s);
                VC_TRY(launch_upsample2x(op.a, op.b, e->prec, s));
                break;
            }
            case Op::MAXPOOL: { ProfScope ps(e, aux_cat, 0, 0, s); VC_TRY(launch_maxpool3s2(op.a, op.b, e->aux_prec, s)); break; }   // ReID only
            case Op::TO_FP8: { ProfScope ps(e, aux_cat, 0, 0, s); VC_TRY(launch_bf16_to_fp8(op.a, op.b, 1.0f / e->act_scale, s)); break; }
            case Op::HEAD_COMPACT: {
                ProfScope ps(e, aux_cat, 0, 0, s);
                const int i = op.level, M = op.a.B * op.a.H * op.a.W;
                VC_TRY(launch_head_compact(e->d_obj[i], op.a, M, op.a.H * op.a.W, e->cfg.conf_thres, e->hc_cap[i], e->d_hc_count + i, e->d_hc_list[i], e->d_hc_x[i],
                                           e->post.overflow, s));
                break;
            }
        }
    }
    return VC_OK;
}

// AutoShape geometry: models/common.py::AutoShape.forward + utils/augmentations.py::letterbox (auto=False)
static int py_round(double x) { return (int)std::nearbyint(x); }     // banker's rounding like Python 3 round()

static void autoshape_net_size(const int* h, const int* w, int n, int size, int& nh, int& nw) {
    double mh = 0, mw = 0;
    for (int i = 0; i < n; ++i) {
        const double g = (double)size / std::max(h[i], w[i]);
        mh = std::max(mh, h[i] * g); mw = std::max(mw, w[i] * g);
    }
    nh = (int)std::ceil(mh / 32.0) * 32; nw = (int)std::ceil(mw / 32.0) * 32;
}

static LetterboxGeom letterbox_geom(int h0, int w0, int nh, int nw, bool swap_rb) {
    LetterboxGeom g{};
    g.src_h = h0; g.src_w = w0; g.net_h = nh; g.net_w = nw;
    const double r = std::min((double)nh / h0, (double)nw / w0);
    g.unpad_w = py_round(w0 * r); g.unpad_h = py_round(h0 * r);
    const double dw = (nw - g.unpad_w) / 2.0, dh = (nh - g.unpad_h) / 2.0;
    g.top = py_round(dh - 0.1); g.left = py_round(dw - 0.1);
    g.swap_rb = swap_rb ? 1 : 0;
    return g;
}

static int yolo_forward(vc_engine* e, int B, int nh, int nw) {
    hipStream_t ds = e->dstream;
    const bool want_sparse = e->prec == PREC_BF16 && e->opt.sparse_head && !e->want_pred_debug && e->d_hc_count;
    YoloPlan& plan = e->yolo_plans[{B, nh, nw, want_sparse ? 1 : 0}];
    if (plan.ops.empty()) {
        const int st = yolo_build_ops(e, B, nh, nw, plan.ops);
        if (st != VC_OK) { plan.ops.clear(); return st; }
        memcpy(plan.layer_view, e->layer_view, sizeof(plan.layer_view));
        plan.sparse = e->sparse_pass;
    } else {
        memcpy(e->layer_view, plan.layer_view, sizeof(plan.layer_view));
        e->sparse_pass = plan.sparse;
    }
    std::vector<Op>& ops = plan.ops;
    e->l0_stale = false;
    e->up_folded.clear();
    e->s2pw_folded.clear();
    if (e->sparse_pass) {                                    // the compaction sets overflow flags and counts: clear them ahead of the ops
        e->hc_ring_cur = (int)(e->hc_ring_seq++ % vc_engine::HC_RING);
        VC_HIP(hipMemsetAsync(e->d_zero, 0, e->zero_bytes, ds));
    }
    VC_TRY(run_ops(e, ops, VC_PROF_DETECT_AUX, ds));          // (joins the head stream before it returns: the decode reads the head ops' rows)
    if (e->sparse_pass && e->prof_async && e->h_hc_ring)     // executed-work accounting: this pass's gathered-row counts, next to the event pairs
        VC_HIP(hipMemcpyAsync(e->h_hc_ring + (size_t)e->hc_ring_cur * 4, e->d_hc_count, 4 * sizeof(int), hipMemcpyDeviceToHost, ds));
    // decode + NMS
    const int nc = e->cfg.num_classes, no = nc + 5, lcs = round_up(3 * no, 8);
    DecodeLevel lv[3];
    int base = 0;
    const int strides[3] = {8, 16, 32};
    for (int i = 0; i < 3; ++i) {
        lv[i].logits = e->d_logits[i]; lv[i].bf16 = e->prec == PREC_F32 ? 0 : 1; lv[i].ny = nh / strides[i]; lv[i].nx = nw / strides[i]; lv[i].cs = lcs;
        lv[i].stride = (float)strides[i];
        for (int a = 0; a < 3; ++a) { lv[i].anchor_w[a] = e->anchors[i][2 * a]; lv[i].anchor_h[a] = e->anchors[i][2 * a + 1]; }
        lv[i].base = base;
        base += 3 * lv[i].ny * lv[i].nx;
    }
    e->last_B = B; e->last_nh = nh; e->last_nw = nw; e->last_ntotal = base;
    float* dbg = nullptr;
    if (e->want_pred_debug) {
        if (!e->d_pred_debug) {
            const int S = round_up(e->cfg.img_size, 32);
            const size_t nmax = (size_t)3 * ((S / 8) * (S / 8) + (S / 16) * (S / 16) + (S / 32) * (S / 32));
            VC_TRY(dev_alloc(e, (void**)&e->d_pred_debug, (size_t)e->cfg.max_batch * nmax * no * sizeof(float)));
        }
        dbg = e->d_pred_debug;
    }
    if (e->sparse_pass) {
        for (int i = 0; i < 3; ++i) lv[i].logits = e->d_hc_logits[i];
        ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, ds);
        VC_TRY(launch_decode_sparse(lv, e->d_hc_count, e->d_hc_list, e->hc_cap, nc, e->cfg.conf_thres, e->cfg.max_candidates, e->post, ds));
    } else {
        ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, ds);
        VC_TRY(launch_decode(lv, 3, B, nc, e->cfg.conf_thres, e->cfg.max_candidates, e->post, dbg, base, ds));
    }
    { ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, ds); VC_TRY(launch_nms(B, e->cfg.max_candidates, e->cfg.max_det, e->cfg.iou_thres, e->d_geom, e->post, ds)); }
    return VC_OK;
}

int run_detector_dev(vc_engine* e, const uint8_t* d_frames, int B, int h, int w, bool swap_rb) {
    VC_CHECK(e->finalized && e->cfg.with_detector, VC_ERR_STATE, "detector not finalized");
    VC_CHECK(B >= 1, VC_ERR_ARG, "a batch needs at least one frame (got %d)", B);
    VC_CHECK(B <= e->cfg.max_batch, VC_ERR_CAPACITY, "batch %d exceeds max_batch %d", B, e->cfg.max_batch);
    VC_CHECK(h >= 1 && w >= 1, VC_ERR_ARG, "frame size %d x %d", h, w);
    int nh, nw;
    autoshape_net_size(&h, &w, 1, e->cfg.img_size, nh, nw);
    const LetterboxGeom g = letterbox_geom(h, w, nh, nw, swap_rb);
    float* hg = e->h_geom + (size_t)(e->geom_seq++ & 1) * e->cfg.max_batch * 5;      // two pinned slots: two submissions may be in flight
    scale_geom_host(ScaleGeom{nh, nw, h, w}, hg);
    for (int b = 1; b < B; ++b) memcpy(hg + (size_t)b * 5, hg, 5 * sizeof(float));
    VC_HIP(hipMemcpyAsync(e->d_geom, hg, (size_t)B * 5 * sizeof(float), hipMemcpyHostToDevice, e->dstream));
    // bf16, frame already at network scale: the stem reads the u8 frames itself (same arithmetic per pixel, bit-identical stem
    // output) and the 8-byte-per-pixel letterboxed tensor is neither written nor read back
    static const bool fuse_on = !(getenv("VC_STEM_U8") && atoi(getenv("VC_STEM_U8")) == 0) && !(getenv("VC_STEM_DIRECT") && atoi(getenv("VC_STEM_DIRECT")) == 0);
    const bool same_scale = g.unpad_h == g.src_h && g.unpad_w == g.src_w && g.src_w % 2 == 0 && g.left % 2 == 0;
    // frames that need the resize (1280 x 720 -> 384 x 640, Q8): only front_fused_kernel evaluates it at patch-build time, so the fold-in
    // needs that kernel to be the one that runs (YOLOv5s widths, bf16 engine, option on) and the tile's source footprint to fit its staging
    const bool resize_ok = !same_scale && ((uintptr_t)d_frames & 3) == 0 && e->prec == PREC_BF16 && e->opt.front_fused > 0 && e->ch[0] == 32 && e->ch[1] == 64 && front_fused_resize_ok(g);
    const bool fuse = fuse_on && e->aux_prec == PREC_BF16 && (same_scale || resize_ok) && e->ch[0] % 16 == 0 && e->ch[0] <= 64;
    e->stem_src = fuse ? d_frames : nullptr;
    e->stem_geom = g;
    e->in_stale = fuse;
    if (!fuse) { ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, e->dstream); VC_TRY(launch_letterbox(d_frames, e->ybuf["in"].ptr, B, g, e->aux_prec, e->dstream)); }
    const int st = yolo_forward(e, B, nh, nw);
    if (fuse && e->in_stale) e->stem_src = d_frames;    // kept for vc_detect_debug_layer(-1) (not when run_ops fell back to the letterbox kernel)
    return st;
}

// ------------------------------------------------------------------------------------------------ ReID net
static const struct { const char* name; int cin, cout; bool down; } kReidBlocks[8] = {
    {"layer1.0", 64, 64, false}, {"layer1.1", 64, 64, false}, {"layer2.0", 64, 128, true}, {"layer2.1", 128, 128, false},
    {"layer3.0", 128, 256, true}, {"layer3.1", 256, 256, false}, {"layer4.0", 256, 512, true}, {"layer4.1", 512, 512, false}};

static void reid_define(vc_engine* e) {
    Net& n = e->reid;
    n.add("conv", 64, 3, 3, 3);
    for (const auto& b : kReidBlocks) {
        n.add(std::string(b.name) + ".conv1", b.cout, b.cin, 3, 3);
        n.add(std::string(b.name) + ".conv2", b.cout, b.cout, 3, 3);
        if (b.down || b.cin != b.cout) n.add(std::string(b.name) + ".downsample", b.cout, b.cin, 1, 1);
    }
}

static int reid_cpad(int prec) { return prec == PREC_F32 ? 4 : 8; }

static int reid_alloc(vc_engine* e) {
    const int es = elem_size(e->aux_prec);
    const size_t K = e->cfg.max_crops;
    auto& m = e->rbuf;
    VC_TRY(alloc_buf(e, m, "in", K * 50 * 50, reid_cpad(e->aux_prec), es));
    VC_TRY(alloc_buf(e, m, "r0", K * 50 * 50, 64, es));
    VC_TRY(alloc_buf(e, m, "x0", K * 25 * 25, 64, es));
    int hw = 25;
    for (int i = 0; i < 8; ++i) {
        const auto& b = kReidBlocks[i];
        if (b.down) hw = (hw - 1) / 2 + 1;
        const std::string p = b.name;
        VC_TRY(alloc_buf(e, m, p + ".t", K * hw * hw, b.cout, es));
        VC_TRY(alloc_buf(e, m, p + ".y", K * hw * hw, b.cout, es));
        if (b.down || b.cin != b.cout) VC_TRY(alloc_buf(e, m, p + ".d", K * hw * hw, b.cout, es));
    }
    VC_TRY(dev_alloc(e, (void**)&e->d_crops, K * 5 * sizeof(int)));
    VC_TRY(host_alloc(e, (void**)&e->h_crops, K * 5 * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&e->d_feat, K * VC_FEAT_DIM * sizeof(float)));
    for (int q = 0; q < 3; ++q) {                      // stream path: ReID of batches i+1 / i+2 runs while batch i is tracked
        VC_TRY(dev_alloc(e, (void**)&e->d_feat2[q], K * VC_FEAT_DIM * sizeof(float)));
        VC_TRY(dev_alloc(e, (void**)&e->d_crops2[q], K * 5 * sizeof(int)));
        VC_TRY(host_alloc(e, (void**)&e->h_crops2[q], K * 5 * sizeof(int)));
    }
    VC_TRY(host_alloc(e, (void**)&e->h_feat, K * VC_FEAT_DIM * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&e->d_reid_in_nchw, K * 3 * 50 * 50 * sizeof(float)));
    return VC_OK;
}

// forward from the pre-filled "in" buffer (k x 50 x 50 x cpad) to feat_out.  A conv launch addresses at most 2^24 output pixels
// (conv_check), and the first layer has 2500 per crop: more than VC_REID_CHUNK crops run as several passes over slices of "in"
// (the other activation buffers are reused; the passes are ordered on the stream).
#define VC_REID_CHUNK 6400
#define VC_REID_PLAN_CACHE_MAX_K 256       // crop counts whose op plans are kept (k0 is 0 for these: one chunk)
static int reid_forward_chunk(vc_engine* e, int k0, int k, hipStream_t rs, float* feat_out);
static int reid_forward(vc_engine* e, int k, hipStream_t rs, float* feat_out) {
    // ... and its 50 x 50 x 64 output must stay below the 2 GiB a buffer descriptor addresses (fp32: 3200 crops)
    const int chunk = std::min(VC_REID_CHUNK, (int)(((1ull << 31) - 1) / ((size_t)2500 * 64 * elem_size(e->aux_prec))) / 64 * 64);
    for (int k0 = 0; k0 < k; k0 += chunk)
        VC_TRY(reid_forward_chunk(e, k0, std::min(chunk, k - k0), rs, feat_out + (size_t)k0 * VC_FEAT_DIM));
    return VC_OK;
}
static int reid_forward_chunk(vc_engine* e, int k0, int k, hipStream_t rs, float* feat_out) {
    // The plan cache pays at batch 1 (a few crops per frame, the same counts again and again).  A batch of the stream path has thousands of
    // distinct crop counts: those build a transient plan (21 ops, tens of microseconds beside a multi-millisecond pass) instead of
    // filling the map (~10 KB per entry, a full clear at 8192 entries).
    ReidPlan transient;
    const bool cached = k <= VC_REID_PLAN_CACHE_MAX_K;
    ReidPlan& plan = cached ? e->reid_plans[{k0, k}] : transient;
    std::vector<Op>& ops = plan.ops;
    if (!ops.empty()) {
        VC_TRY(run_ops(e, ops, VC_PROF_REID_AUX, rs));
        ProfScope ps(e, VC_PROF_REID_AUX, 0, 0, rs);
        return launch_avgpool_l2norm(plan.out, feat_out, e->aux_prec, rs);
    }
    PlanBuilder pb{e, &e->reid, &ops, e->aux_prec};
    auto& m = e->rbuf;
    View x = mkview(m["in"], k, 50, 50, reid_cpad(e->aux_prec), 0);
    x.ptr = (char*)x.ptr + (size_t)k0 * 50 * 50 * reid_cpad(e->aux_prec) * elem_size(e->aux_prec);
    x = pb.conv("conv", x, mkview(m["r0"], k, 50, 50, 64, 0), 3, 1, 1, ACT_RELU);             // model.py:51-55
    { Op op{}; op.kind = Op::MAXPOOL; op.a = x; op.b = mkview(m["x0"], k, 25, 25, 64, 0); ops.push_back(op); x = op.b; }   // :58
    for (const auto& b : kReidBlocks) {                                                        // BasicBlock.forward, model.py:30-38
        const std::string p = b.name;
        const int s = b.down ? 2 : 1;
        View t = pb.conv(p + ".conv1", x, mkview(m[p + ".t"], k, 0, 0, b.cout, 0), 3, s, 1, ACT_RELU);
        View sc = x;
        if (b.down || b.cin != b.cout) sc = pb.conv(p + ".downsample", x, mkview(m[p + ".d"], k, 0, 0, b.cout, 0), 1, s, 0, ACT_NONE);
        x = pb.conv(p + ".conv2", t, mkview(m[p + ".y"], k, 0, 0, b.cout, 0), 3, 1, 1, ACT_RELU, &sc, RES_BEFORE_ACT);
    }
    if (pb.status != VC_OK) { const int st = pb.status; ops.clear(); return st; }
    plan.out = x;
    VC_TRY(run_ops(e, ops, VC_PROF_REID_AUX, rs));
    { ProfScope ps(e, VC_PROF_REID_AUX, 0, 0, rs); VC_TRY(launch_avgpool_l2norm(x, feat_out, e->aux_prec, rs)); }               // model.py:70,93
    return VC_OK;
}

int run_reid_on(vc_engine* e, const uint8_t* d_frames, int H, int W, int k, const int* d_crops, float* feat_out, hipStream_t rs) {
    VC_CHECK(e->finalized && e->cfg.with_reid, VC_ERR_STATE, "ReID net not finalized");
    VC_CHECK(k <= e->cfg.max_crops, VC_ERR_CAPACITY, "%d crops exceed max_crops %d", k, e->cfg.max_crops);
    if (k <= 0) return VC_OK;
    { ProfScope ps(e, VC_PROF_REID_AUX, 0, 0, rs);
      VC_TRY(launch_crop_resize(d_frames, H, W, d_crops, k, e->rbuf["in"].ptr, reid_cpad(e->aux_prec), e->aux_prec, rs, e->opt.crop_per_pixel != 0)); }
    return reid_forward(e, k, rs, feat_out);
}

// blocking-API flavour (vc_embed, vc_deepsort_update, vc_videotracker_run): tracker stream, shared crop / feature buffers
int run_reid_dev(vc_engine* e, const uint8_t* d_frames, int H, int W, int k) {
    VC_HIP(hipStreamSynchronize(e->rstream));          // the activation buffers are shared with the stream path's ReID
    return run_reid_on(e, d_frames, H, W, k, e->d_crops, e->d_feat, e->stream);
}

}  // namespace vc

// ================================================================================================ C ABI
using namespace vc;

extern "C" {

int vc_version(void) { return 100; }
const char* vc_last_error(void) { return vc::last_error(); }
int vc_device_count(int* n) {
    VC_CHECK(n, VC_ERR_ARG, "null out pointer");
    *n = 0;
    VC_HIP(hipGetDeviceCount(n));
    return VC_OK;
}

int vc_engine_config_default(vc_engine_config* c) {
    VC_CHECK(c, VC_ERR_ARG, "null config");
    memset(c, 0, sizeof(*c));
    c->device = 0; c->precision = VC_PREC_BF16; c->yolo_variant = 0; c->num_classes = 80; c->img_size = 640;
    c->max_batch = 16; c->max_frame_h = 720; c->max_frame_w = 1280;
    c->conf_thres = 0.25f; c->iou_thres = 0.45f; c->max_det = 300; c->max_candidates = 4096;
    c->max_crops = 1024; c->max_tracks = 4096; c->nn_budget_cap = 100; c->with_detector = 1; c->with_reid = 1;
    c->max_trackers = 256; c->tracks_per_tracker = 512;
    return VC_OK;
}

int vc_engine_create(const vc_engine_config* cfg, vc_engine** out) {
    VC_CHECK(cfg && out, VC_ERR_ARG, "null argument");
    VC_CHECK(cfg->yolo_variant >= 0 && cfg->yolo_variant <= 2, VC_ERR_ARG, "yolo_variant must be 0..2");
    VC_CHECK(cfg->precision == VC_PREC_BF16 || cfg->precision == VC_PREC_F32 || cfg->precision == VC_PREC_FP8, VC_ERR_ARG, "bad precision");
    VC_CHECK(cfg->max_candidates % 64 == 0 && cfg->max_candidates >= 64 && cfg->max_candidates <= 8192, VC_ERR_ARG,
             "max_candidates must be a multiple of 64 in [64, 8192]");
    VC_CHECK(cfg->max_batch >= 1 && cfg->num_classes >= 1 && cfg->max_det >= 1, VC_ERR_ARG, "bad sizes");
    // The engine runs four streams side by side (detector, ReID, tracker, host-frame copies) next to the application's own.  The HIP
    // runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default): two of the engine's streams on one queue
    // serialise, e.g. a batch's PCIe copy in front of the detector kernels of the batch before it.  Measured on the second engine of a
    // process (the first happens to get distinct queues): 15.1 k -> 17.3 k frames/s with host frames, 17.7 k -> 18.6 k device-resident.
    // GPU_MAX_HW_QUEUES >= 8 is therefore a DEPLOYMENT requirement; the runtime reads it once, when it builds its queue
    // pool, so the embedding application has to export it before its first HIP call.  The library does not touch the process
    // environment (setenv is not thread-safe against concurrent getenv); the Python binding sets a default at import
    // (_lib.py), and an engine created without it says so once when VC_VERBOSE is set.
    {
        static bool warned = false;
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (!warned && (!q || atoi(q) < 8) && getenv("VC_VERBOSE")) {     // opt-in (a library does not write to the process's stderr unasked)
            warned = true;
            fprintf(stderr, "libvcount_hip: GPU_MAX_HW_QUEUES is %s; export GPU_MAX_HW_QUEUES=8 before the first HIP call or the engine's streams share hardware queues (INTEGRATION.md)\n", q ? q : "unset");
        }
    }
    int ndev = 0;
    VC_HIP(hipGetDeviceCount(&ndev));
    VC_CHECK(cfg->device >= 0 && cfg->device < ndev, VC_ERR_HIP, "device %d not present (%d visible)", cfg->device, ndev);
    VC_HIP(hipSetDevice(cfg->device));
    vc_engine* e = new vc_engine();
    e->cfg = *cfg;
    e->prec = cfg->precision;
    e->aux_prec = cfg->precision == VC_PREC_FP8 ? (int)PREC_BF16 : cfg->precision;      // stem, logits, pools of the ReID net, ReID convs
    e->act_scale = getenv("VC_FP8_ACT_SCALE") ? (float)atof(getenv("VC_FP8_ACT_SCALE")) : 1.0f;
    {
        auto env_int = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        e->opt.c3_fused = env_int("VC_C3_FUSED", 1); e->opt.bneck_fused = env_int("VC_BNECK_FUSED", 1); e->opt.bneck_cv3 = env_int("VC_BNECK_CV3", 1);
        e->opt.front_fused = env_int("VC_FRONT_FUSED", 1); e->opt.crop_per_pixel = getenv("VC_CROP_PER_PIXEL") ? 1 : 0;
        e->opt.sparse_head = env_int("VC_SPARSE_HEAD", 1); e->opt.reid_block_fused = env_int("VC_REID_BLOCK_FUSED", 1); e->opt.head_side = env_int("VC_HEAD_SIDE", 1); e->opt.fuse_upsample = env_int("VC_FUSE_UPSAMPLE", 1); e->opt.sppf_sep = env_int("VC_SPPF_SEP", 1); e->opt.fuse_s2_pw = env_int("VC_FUSE_S2PW", 1);
    }
    memcpy(e->anchors, kAnchors, sizeof(kAnchors));
    int st = VC_OK;
    do {
        // the tracker's kernels are tiny and latency-critical (the host waits on them every frame): highest priority,
        // so they are not queued behind the conv waves of the detector / ReID streams they run beside
        int plo = 0, phi = 0;
        hipDeviceGetStreamPriorityRange(&plo, &phi);             // numerically lower = higher priority
        // Optional CU partition (VC_TRACK_CUS=n): the first n CUs are reserved for the tracker stream so that its kernels
        // never wait for a conv workgroup to drain; the conv streams get the remaining CUs.
        const int reserve = getenv("VC_TRACK_CUS") ? atoi(getenv("VC_TRACK_CUS")) : 0;
        bool ok;
        if (reserve > 0) {
            hipDeviceProp_t prop;
            hipGetDeviceProperties(&prop, cfg->device);
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mt(words, 0), mc(words, 0);
            for (int c = 0; c < ncu; ++c) (c < reserve ? mt : mc)[c / 32] |= 1u << (c % 32);
            ok = hipExtStreamCreateWithCUMask(&e->stream, words, mt.data()) == hipSuccess &&
                 hipExtStreamCreateWithCUMask(&e->dstream, words, mc.data()) == hipSuccess &&
                 hipExtStreamCreateWithCUMask(&e->rstream, words, mc.data()) == hipSuccess;
        } else {
            ok = hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, phi) == hipSuccess &&
                 hipStreamCreateWithPriority(&e->dstream, hipStreamNonBlocking, plo) == hipSuccess &&
                 hipStreamCreateWithPriority(&e->rstream, hipStreamNonBlocking, plo) == hipSuccess;
        }
        if (!ok) { set_error("stream create failed"); st = VC_ERR_HIP; break; }
        // (detector and ReID on ONE queue -- no co-running conv kernels, on the theory that persistent grids sized for the whole chip do not
        // share it well -- measured 16.9 k against 18.5 k frames/s, two alternating runs each: the overlap pays; removed)
        if (hipEventCreateWithFlags(&e->ev_reid[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_reid[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_reid[2], hipEventDisableTiming) != hipSuccess) { set_error("event create failed"); st = VC_ERR_HIP; break; }
        if (hipEventCreateWithFlags(&e->ev_det[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_det[1], hipEventDisableTiming) != hipSuccess) { set_error("event create failed"); st = VC_ERR_HIP; break; }
        if (hipStreamCreateWithPriority(&e->hstream, hipStreamNonBlocking, plo) != hipSuccess || hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) { set_error("head stream create failed"); st = VC_ERR_HIP; break; }
        if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess) { set_error("event create failed"); st = VC_ERR_HIP; break; }
        tune_cache_load(e);
        if (cfg->with_detector) yolo_define(e);
        if (cfg->with_reid) reid_define(e);
        st = tracker_init_pool(e);
    } while (0);
    if (st != VC_OK) { vc_engine_destroy(e); return st; }
    *out = e;
    return VC_OK;
}

int vc_engine_destroy(vc_engine* e) {
    if (!e) return VC_OK;
    tune_cache_save(e);
    hipSetDevice(e->cfg.device);
    vc_comm_destroy(e);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->dstream) hipStreamSynchronize(e->dstream);
    if (e->rstream) hipStreamSynchronize(e->rstream);
    if (e->cstream) hipStreamSynchronize(e->cstream);
    for (void* p : e->allocs) hipFree(p);
    for (void* p : e->host_allocs) hipHostFree(p);
    for (auto& pp : e->prof_pairs) { hipEventDestroy(pp.a); hipEventDestroy(pp.b); }
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->dstream) hipStreamDestroy(e->dstream);
    if (e->rstream) hipStreamDestroy(e->rstream);
    if (e->hstream) hipStreamDestroy(e->hstream);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    for (hipEvent_t ev : e->ev_det) if (ev) hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_reid) if (ev) hipEventDestroy(ev);
    for (vc::TrackStage& ts : e->tstage) if (ts.done) hipEventDestroy(ts.done);
    for (hipEvent_t ev : e->ev_ingest) if (ev) hipEventDestroy(ev);
    if (e->cstream) hipStreamDestroy(e->cstream);
    delete e;
    return VC_OK;
}

static Net* pick_net(vc_engine* e, int net) { return net == VC_NET_YOLO ? &e->yolo : (net == VC_NET_REID ? &e->reid : nullptr); }

int vc_engine_param_count(const vc_engine* e, int net, int* n) {
    VC_CHECK(e && n, VC_ERR_ARG, "null argument");
    const Net* nn = pick_net(const_cast<vc_engine*>(e), net);
    VC_CHECK(nn, VC_ERR_ARG, "bad net id %d", net);
    int cnt = 0;
    for (const auto& p : nn->params) cnt += p.hidden ? 0 : 1;
    *n = cnt;
    return VC_OK;
}

int vc_engine_param_info(const vc_engine* e, int net, int index, char* name, int name_cap, int dims[4]) {
    VC_CHECK(e && name && dims, VC_ERR_ARG, "null argument");
    const Net* nn = pick_net(const_cast<vc_engine*>(e), net);
    VC_CHECK(nn && index >= 0, VC_ERR_ARG, "bad net/index");
    int real = -1;
    for (size_t i = 0, k = 0; i < nn->params.size(); ++i) {
        if (nn->params[i].hidden) continue;
        if ((int)k++ == index) { real = (int)i; break; }
    }
    VC_CHECK(real >= 0, VC_ERR_ARG, "parameter index %d out of range", index);
    const ConvParam& p = nn->params[real];
    snprintf(name, name_cap, "%s", p.name.c_str());
    dims[0] = p.O; dims[1] = p.I; dims[2] = p.kh; dims[3] = p.kw;
    return VC_OK;
}

int vc_engine_set_param(vc_engine* e, int net, const char* name, const float* w, const float* bias) {
    VC_CHECK(e && name && w && bias, VC_ERR_ARG, "null argument");
    VC_CHECK(!e->finalized, VC_ERR_STATE, "engine already finalized");
    Net* nn = pick_net(e, net);
    VC_CHECK(nn, VC_ERR_ARG, "bad net id %d", net);
    auto it = nn->index.find(name);
    VC_CHECK(it != nn->index.end(), VC_ERR_NOTFOUND, "no parameter named '%s'", name);
    ConvParam& p = nn->params[it->second];
    VC_CHECK(!p.hidden, VC_ERR_NOTFOUND, "no parameter named '%s'", name);
    p.w.assign(w, w + (size_t)p.O * p.I * p.kh * p.kw);
    p.b.assign(bias, bias + p.O);
    p.set = true;
    return VC_OK;
}

// Detect anchors in pixels, [level P3,P4,P5][anchor][w,h] (ultralytics `model.24.anchors * stride`).  Custom-trained checkpoints
// (the reference loads 'custom' weights, /root/reference/networks/yolo.py:58) may carry autoanchor-evolved values.
int vc_engine_set_anchors(vc_engine* e, const float* anchors18) {
    VC_CHECK(e && anchors18, VC_ERR_ARG, "null argument");
    for (int i = 0; i < 18; ++i) VC_CHECK(anchors18[i] > 0.f && anchors18[i] < 1e5f, VC_ERR_ARG, "anchor %d = %g is not a positive pixel size", i, (double)anchors18[i]);
    memcpy(e->anchors, anchors18, 18 * sizeof(float));
    return VC_OK;
}

int vc_engine_finalize(vc_engine* e) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    VC_CHECK(!e->finalized, VC_ERR_STATE, "engine already finalized");
    VC_HIP(hipSetDevice(e->cfg.device));
    for (auto& p : e->yolo.params) {
        if (p.fuse_a < 0) continue;
        const ConvParam &a = e->yolo.params[p.fuse_a], &b = e->yolo.params[p.fuse_b];
        VC_CHECK(a.set && b.set, VC_ERR_STATE, "parameters '%s' / '%s' were never set", a.name.c_str(), b.name.c_str());
        p.w = a.w; p.w.insert(p.w.end(), b.w.begin(), b.w.end());
        p.b = a.b; p.b.insert(p.b.end(), b.b.begin(), b.b.end());
        p.set = true;
    }
    for (auto& p : e->yolo.params) {                          // sparse head: the objectness rows of a Detect head as an 8-channel conv
        if (p.obj_of < 0) continue;
        const ConvParam& hd = e->yolo.params[p.obj_of];
        VC_CHECK(hd.set, VC_ERR_STATE, "parameter '%s' was never set", hd.name.c_str());
        const int no = hd.O / 3;
        p.w.assign((size_t)8 * p.I, 0.f); p.b.assign(8, 0.f);
        for (int a = 0; a < 3; ++a) {
            memcpy(&p.w[(size_t)a * p.I], &hd.w[(size_t)(a * no + 4) * hd.I], (size_t)p.I * sizeof(float));
            p.b[a] = hd.b[a * no + 4];
        }
        p.set = true;
    }
    {
        std::vector<char> part(e->yolo.params.size(), 0);
        for (auto& p : e->yolo.params) if (p.fuse_a >= 0) { part[p.fuse_a] = 1; part[p.fuse_b] = 1; }
        for (size_t i = 0; i < e->yolo.params.size(); ++i) {
            if (part[i]) { VC_CHECK(e->yolo.params[i].set, VC_ERR_STATE, "parameter '%s' was never set", e->yolo.params[i].name.c_str()); continue; }
            ConvParam& cp = e->yolo.params[i];
            // fp8 mode: every detector conv but the 3-channel stem (Cin must be a multiple of 16 for 16-byte K chunks)
            const bool fp8 = e->prec == PREC_FP8 && cp.I % 16 == 0;
            VC_TRY(pack_and_upload(e, cp, fp8 ? (int)PREC_FP8 : e->aux_prec, e->act_scale));
        }
    }
    for (auto& p : e->reid.params) VC_TRY(pack_and_upload(e, p, e->aux_prec));
    e->d_frames_bytes = (size_t)e->cfg.max_batch * e->cfg.max_frame_h * e->cfg.max_frame_w * 3;     // staging for host frames
    VC_TRY(dev_alloc(e, (void**)&e->d_frames, e->d_frames_bytes));
    if (e->cfg.with_detector) VC_TRY(yolo_alloc(e));
    if (e->cfg.with_reid) VC_TRY(reid_alloc(e));
    for (auto& p : e->yolo.params) { p.w.clear(); p.w.shrink_to_fit(); }
    for (auto& p : e->reid.params) { p.w.clear(); p.w.shrink_to_fit(); }
    e->finalized = true;
    return VC_OK;
}

// Kernel-selection switches of a live engine (the parity tests compare a fused kernel with the launches it replaces on the same
// engine).  Initial values come from the environment at vc_engine_create; the launch path itself never calls getenv (what is left in
// the launchers are function-local statics initialised once per process: VC_CONV_RESERVE, VC_CONV_PERSIST and the A/B switches).
int vc_engine_set_option(vc_engine* e, const char* name, int value) {
    VC_CHECK(e && name, VC_ERR_ARG, "null argument");
    const std::string n = name;
    if (n == "c3_fused") e->opt.c3_fused = value;
    else if (n == "bneck_fused") e->opt.bneck_fused = value;
    else if (n == "bneck_cv3") e->opt.bneck_cv3 = value;
    else if (n == "front_fused") e->opt.front_fused = value;
    else if (n == "crop_per_pixel") e->opt.crop_per_pixel = value;
    else if (n == "sparse_head") e->opt.sparse_head = value;
    else if (n == "reid_block_fused") e->opt.reid_block_fused = value;
    else if (n == "head_side") e->opt.head_side = value;
    else if (n == "fuse_upsample") e->opt.fuse_upsample = value;
    else if (n == "sppf_sep") e->opt.sppf_sep = value;
    else if (n == "fuse_s2_pw") e->opt.fuse_s2_pw = value;
    else if (n == "ff_ablate") e->opt.ff_ablate = value;            // diagnostics (wrong results): tools/ff_ablate.py
    else if (n == "c3_ablate") e->opt.c3_ablate = value;
    else if (n == "dot_arena_mb") { VC_CHECK(value >= 0, VC_ERR_ARG, "dot_arena_mb must be >= 0"); e->dot_arena_max_floats = (size_t)value * 262144; }
    else { set_error("unknown option '%s'", name); return VC_ERR_NOTFOUND; }
    // A cached op resolved its tile configuration for the kernel variant the options selected at its first launch (the upsample fold-in
    // runs on a subset of the tiles and has its own tune key): every switch sends the ops back to tuned_cfg.
    for (auto& kv : e->yolo_plans) for (Op& op : kv.second.ops) op.tuned = -2;
    for (auto& kv : e->reid_plans) for (Op& op : kv.second.ops) op.tuned = -2;
    return VC_OK;
}

int vc_tune_export(vc_engine* e, char* buf, size_t cap, size_t* size) {
    VC_CHECK(e && size, VC_ERR_ARG, "null argument");
    std::string text;
    for (const auto& kv : e->tuned) text += kv.first + " " + std::to_string(kv.second) + "\n";
    *size = text.size() + 1;
    if (!buf) return VC_OK;
    VC_CHECK(cap >= text.size() + 1, VC_ERR_CAPACITY, "vc_tune_export: %zu bytes needed", text.size() + 1);
    memcpy(buf, text.c_str(), text.size() + 1);
    return VC_OK;
}

int vc_tune_import(vc_engine* e, const char* text) {
    VC_CHECK(e && text, VC_ERR_ARG, "null argument");
    const char* p = text;
    int n = 0, cfg = 0, used = 0;
    char key[200];
    while (sscanf(p, "%199s %d%n", key, &cfg, &used) == 2) {
        VC_CHECK(cfg >= -1 && cfg < conv_num_cfgs(), VC_ERR_ARG, "vc_tune_import: tile configuration %d of '%s' does not exist in this build", cfg, key);
        e->tuned[key] = cfg;
        p += used; ++n;
    }
    while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') ++p;
    VC_CHECK(*p == 0, VC_ERR_ARG, "vc_tune_import: malformed text after %d entries", n);
    {   // the engines of this process on the same device follow (tuned_cfg consults g_tuned before timing anything)
        std::lock_guard<std::mutex> lk(g_tune_mu);
        for (const auto& kv : e->tuned) g_tuned["d" + std::to_string(e->cfg.device) + "_" + kv.first] = kv.second;
    }
    if (n) e->tuned_dirty = true;
    // cached op plans resolved their tile configuration at their first launch: they look the (imported) choice up again
    for (auto& kv : e->yolo_plans) for (Op& op : kv.second.ops) op.tuned = -2;
    for (auto& kv : e->reid_plans) for (Op& op : kv.second.ops) op.tuned = -2;
    return VC_OK;
}

int vc_engine_sync(vc_engine* e) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    VC_HIP(hipStreamSynchronize(e->stream));
    VC_HIP(hipStreamSynchronize(e->dstream));
    VC_HIP(hipStreamSynchronize(e->rstream));
    return VC_OK;
}

// ---- detect --------------------------------------------------------------------------------------------
int vc_detect(vc_engine* e, const uint8_t* const* rgb, const int* h, const int* w, int n, float* out_det, int* out_count) {
    VC_CHECK(e && rgb && h && w && out_det && out_count, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_detector, VC_ERR_STATE, "detector not finalized");
    VC_CHECK(n >= 1, VC_ERR_ARG, "a batch needs at least one image (got %d)", n);
    VC_CHECK(n <= e->cfg.max_batch, VC_ERR_CAPACITY, "batch %d exceeds max_batch %d", n, e->cfg.max_batch);
    for (int i = 0; i < n; ++i) VC_CHECK(rgb[i] && h[i] >= 1 && w[i] >= 1, VC_ERR_ARG, "image %d: null pointer or size %d x %d", i, h[i], w[i]);
    VC_HIP(hipSetDevice(e->cfg.device));
    int nh, nw;
    autoshape_net_size(h, w, n, e->cfg.img_size, nh, nw);
    size_t off = 0;
    std::vector<float> hg((size_t)n * 5);
    std::vector<size_t> offs(n);
    for (int i = 0; i < n; ++i) {
        VC_CHECK(h[i] > 0 && w[i] > 0, VC_ERR_ARG, "empty image %d", i);
        const size_t bytes = (size_t)h[i] * w[i] * 3;
        VC_CHECK(off + bytes <= e->d_frames_bytes, VC_ERR_CAPACITY, "frames exceed the staging buffer (max_frame_h/w)");
        VC_HIP(hipMemcpyAsync(e->d_frames + off, rgb[i], bytes, hipMemcpyHostToDevice, e->dstream));
        offs[i] = off; off += bytes;
        scale_geom_host(ScaleGeom{nh, nw, h[i], w[i]}, &hg[(size_t)i * 5]);
    }
    VC_HIP(hipMemcpyAsync(e->d_geom, hg.data(), hg.size() * sizeof(float), hipMemcpyHostToDevice, e->dstream));
    VC_HIP(hipStreamSynchronize(e->dstream));        // hg / rgb[] are caller or stack memory
    const size_t px = (size_t)nh * nw * 4 * elem_size(e->aux_prec);
    // this pass reads the letterboxed tensor written below: a u8 source left behind by an earlier stream pass (kept for
    // vc_detect_debug_layer(-1)) must not be picked up by the stem kernels again
    e->stem_src = nullptr; e->in_stale = false;
    for (int i = 0; i < n; ++i) {
        const LetterboxGeom g = letterbox_geom(h[i], w[i], nh, nw, false);
        ProfScope ps(e, VC_PROF_DETECT_AUX, 0, 0, e->dstream);
        VC_TRY(launch_letterbox(e->d_frames + offs[i], (char*)e->ybuf["in"].ptr + (size_t)i * px, 1, g, e->aux_prec, e->dstream));
    }
    VC_TRY(yolo_forward(e, n, nh, nw));
    const int md = e->cfg.max_det;
    VC_HIP(hipMemcpyAsync(e->h_det, e->post.det, (size_t)n * md * 6 * sizeof(float), hipMemcpyDeviceToHost, e->dstream));
    VC_HIP(hipMemcpyAsync(e->h_det_count, e->post.det_count, n * sizeof(int), hipMemcpyDeviceToHost, e->dstream));
    VC_HIP(hipStreamSynchronize(e->dstream));
    for (int i = 0; i < n; ++i)
        VC_CHECK(e->h_det_count[i] >= 0, VC_ERR_CAPACITY, "image %d: more than max_candidates (%d) boxes passed conf_thres; raise vc_engine_config.max_candidates "
                 "(upstream keeps up to max_nms = 30000)", i, e->cfg.max_candidates);
    memcpy(out_det, e->h_det, (size_t)n * md * 6 * sizeof(float));
    memcpy(out_count, e->h_det_count, n * sizeof(int));
    return VC_OK;
}

int vc_detect_debug_shape(const vc_engine* e, int* net_h, int* net_w, int* n_candidates) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    if (net_h) *net_h = e->last_nh;
    if (net_w) *net_w = e->last_nw;
    if (n_candidates) *n_candidates = e->last_ntotal;
    return VC_OK;
}

static int read_view_f32(vc_engine* e, const View& v, float* out, size_t cap, int dims[4]) {
    const size_t n = (size_t)v.B * v.H * v.W * v.C;
    VC_CHECK(n <= cap, VC_ERR_CAPACITY, "debug buffer too small: need %zu floats", n);
    const int es = v.es ? v.es : elem_size(e->prec);
    const size_t rows = (size_t)v.B * v.H * v.W;
    std::vector<uint8_t> raw(rows * v.cs * es);
    VC_HIP(hipStreamSynchronize(e->stream));
    VC_HIP(hipStreamSynchronize(e->dstream));
    VC_HIP(hipMemcpy(raw.data(), v.ptr, raw.size(), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < rows; ++r)
        for (int c = 0; c < v.C; ++c) {
            const size_t o = r * v.cs + v.co + c;
            out[r * v.C + c] = es == 4 ? ((const float*)raw.data())[o] : es == 2 ? bf16_to_f32(((const uint16_t*)raw.data())[o]) : e4m3_to_f32(raw[o]) * e->act_scale;
        }
    dims[0] = v.B; dims[1] = v.H; dims[2] = v.W; dims[3] = v.C;
    return VC_OK;
}

int vc_detect_debug_layer(vc_engine* e, int layer, float* out, size_t cap, int dims[4]) {
    VC_CHECK(e && out && dims, VC_ERR_ARG, "null argument");
    VC_CHECK(e->last_B > 0, VC_ERR_STATE, "no detector run yet");
    VC_CHECK(layer >= -1 && layer < 24, VC_ERR_ARG, "layer must be -1..23");
    if (layer == -1) {
        if (e->in_stale && e->stem_src) {         // the last pass folded the letterbox into the stem: produce the tensor now (the frames must still be there)
            VC_TRY(launch_letterbox(e->stem_src, e->ybuf["in"].ptr, e->last_B, e->stem_geom, e->aux_prec, e->dstream));
            VC_HIP(hipStreamSynchronize(e->dstream));
            e->in_stale = false;
        }
        View v = mkview(e->ybuf["in"], e->last_B, e->last_nh, e->last_nw, 3, 0);
        return read_view_f32(e, v, out, cap, dims);
    }
    if ((layer == 11 || layer == 12 || layer == 15 || layer == 16) && !e->up_folded.empty()) {   // the last pass folded the upsampling into its consumer
        for (const auto& ab : e->up_folded) VC_TRY(launch_upsample2x(ab.first, ab.second, e->prec, e->dstream));
        VC_HIP(hipStreamSynchronize(e->dstream));
        e->up_folded.clear();
    }
    if ((layer == 3 || layer == 5 || layer == 7) && !e->s2pw_folded.empty()) {   // the last pass kept a stride-2 conv's output on chip for its one reader
        for (ConvP c : e->s2pw_folded) { c.cfg = 49; VC_TRY(launch_conv(c, e->dstream)); }   // the fused kernel's own tile and K order (conv3x3s2_halo_kernel<256, 128, 4, 1, 2>)
        VC_HIP(hipStreamSynchronize(e->dstream));
        e->s2pw_folded.clear();
    }
    VC_CHECK(!(layer == 0 && e->l0_stale), VC_ERR_STATE,
             "layer 0 was not written by the last pass (front_fused_kernel keeps it in LDS); vc_engine_set_option(e, \"front_fused\", 0) and run again");
    return read_view_f32(e, e->layer_view[layer], out, cap, dims);
}

int vc_detect_debug_pred(vc_engine* e, float* out, size_t cap) {
    VC_CHECK(e, VC_ERR_ARG, "null argument");
    if (!out) { e->want_pred_debug = true; return VC_OK; }       // arm: the next run writes the full decoded tensor
    VC_CHECK(e->d_pred_debug && e->last_B > 0, VC_ERR_STATE, "arm with out=NULL and run the detector first");
    const size_t n = (size_t)e->last_B * e->last_ntotal * (e->cfg.num_classes + 5);
    VC_CHECK(n <= cap, VC_ERR_CAPACITY, "need %zu floats", n);
    VC_HIP(hipStreamSynchronize(e->dstream));
    VC_HIP(hipMemcpy(out, e->d_pred_debug, n * sizeof(float), hipMemcpyDeviceToHost));
    return VC_OK;
}

int vc_embed_debug_input(vc_engine* e, int k, float* out, size_t cap, int dims[4]) {
    VC_CHECK(e && out && dims, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_reid, VC_ERR_STATE, "ReID net not finalized");
    VC_CHECK(k > 0 && k <= e->cfg.max_crops, VC_ERR_ARG, "k must be 1..max_crops");
    return read_view_f32(e, mkview(e->rbuf["in"], k, VC_REID_SIZE, VC_REID_SIZE, 3, 0), out, cap, dims);
}

// ---- embed ---------------------------------------------------------------------------------------------
// deep_sort.py:89-95 _xywh_to_xyxy: int() truncation toward zero, clamp to [0, W-1] / [0, H-1]
static void crop_corners(const double* b, int W, int H, int out[4]) {
    const double x = b[0], y = b[1], w = b[2], h = b[3];
    out[0] = std::max((int)(x - w / 2), 0);
    out[2] = std::min((int)(x + w / 2), W - 1);
    out[1] = std::max((int)(y - h / 2), 0);
    out[3] = std::min((int)(y + h / 2), H - 1);
}

int vc_embed(vc_engine* e, const uint8_t* bgr, int h, int w, const double* boxes, int k, float* out_feat) {
    VC_CHECK(e && bgr && (k == 0 || (boxes && out_feat)), VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_reid, VC_ERR_STATE, "ReID net not finalized");
    VC_CHECK(k <= e->cfg.max_crops, VC_ERR_CAPACITY, "%d crops exceed max_crops %d", k, e->cfg.max_crops);
    if (k == 0) return VC_OK;
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t bytes = (size_t)h * w * 3;
    VC_CHECK(bytes <= e->d_frames_bytes, VC_ERR_CAPACITY, "frame exceeds the staging buffer");
    for (int i = 0; i < k; ++i) {
        int c[4];
        crop_corners(boxes + (size_t)i * 4, w, h, c);
        VC_CHECK(c[2] > c[0] && c[3] > c[1], VC_ERR_ARG, "box %d gives an empty crop (the reference's cv2.resize raises here)", i);
        e->h_crops[i * 5 + 0] = 0; e->h_crops[i * 5 + 1] = c[0]; e->h_crops[i * 5 + 2] = c[1]; e->h_crops[i * 5 + 3] = c[2]; e->h_crops[i * 5 + 4] = c[3];
    }
    VC_HIP(hipMemcpyAsync(e->d_frames, bgr, bytes, hipMemcpyHostToDevice, e->stream));
    VC_HIP(hipMemcpyAsync(e->d_crops, e->h_crops, (size_t)k * 5 * sizeof(int), hipMemcpyHostToDevice, e->stream));
    VC_TRY(run_reid_dev(e, e->d_frames, h, w, k));
    VC_HIP(hipMemcpyAsync(e->h_feat, e->d_feat, (size_t)k * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    VC_HIP(hipStreamSynchronize(e->stream));
    memcpy(out_feat, e->h_feat, (size_t)k * VC_FEAT_DIM * sizeof(float));
    return VC_OK;
}

int vc_embed_tensor(vc_engine* e, const float* x, int k, float* out_feat) {
    VC_CHECK(e && x && out_feat, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_reid, VC_ERR_STATE, "ReID net not finalized");
    VC_CHECK(k >= 1, VC_ERR_ARG, "at least one crop is needed (got %d)", k);
    VC_CHECK(k <= e->cfg.max_crops, VC_ERR_CAPACITY, "%d crops exceed max_crops %d", k, e->cfg.max_crops);
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_HIP(hipMemcpyAsync(e->d_reid_in_nchw, x, (size_t)k * 3 * 2500 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    VC_HIP(hipStreamSynchronize(e->rstream));
    VC_TRY(launch_nchw_to_nhwc_pad(e->d_reid_in_nchw, k, 3, 50, 50, e->rbuf["in"].ptr, reid_cpad(e->aux_prec), e->aux_prec, e->stream));
    VC_TRY(reid_forward(e, k, e->stream, e->d_feat));
    VC_HIP(hipMemcpyAsync(e->h_feat, e->d_feat, (size_t)k * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    VC_HIP(hipStreamSynchronize(e->stream));
    memcpy(out_feat, e->h_feat, (size_t)k * VC_FEAT_DIM * sizeof(float));
    return VC_OK;
}

// ---- measurement ---------------------------------------------------------------------------------------
int vc_profile_enable(vc_engine* e, int on) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    e->profiling = on == 1;                 // 1: blocking per-launch events for every category (serialises the streams)
    e->prof_async = on == 2;                // 2: in-flight event pairs around conv launches only, resolved by vc_profile_read
    if (on == 2 && e->prof_pairs.empty()) {
        e->prof_pairs.resize(16384);
        for (auto& pp : e->prof_pairs) { VC_HIP(hipEventCreate(&pp.a)); VC_HIP(hipEventCreate(&pp.b)); }
    }
    if (on == 2) e->prof_used = 0;
    return VC_OK;
}
int vc_profile_reset(vc_engine* e) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    for (auto& c : e->prof) c = ProfCat{};
    e->op_log.clear();
    return VC_OK;
}
int vc_profile_ops(vc_engine* e, char* buf, size_t cap) {
    VC_CHECK(e && buf && cap > 0, VC_ERR_ARG, "bad argument");
    snprintf(buf, cap, "%s", e->op_log.c_str());
    return VC_OK;
}
int vc_profile_read(vc_engine* e, int cat, double* ms, int64_t* launches, double* flops, double* bytes) {
    VC_CHECK(e && cat >= 0 && cat < VC_PROF_NCAT, VC_ERR_ARG, "bad category");
    if (cat == VC_PROF_CONV && e->prof_used > 0) {          // resolve the in-flight pairs into the conv category
        VC_HIP(hipStreamSynchronize(e->dstream)); VC_HIP(hipStreamSynchronize(e->rstream)); VC_HIP(hipStreamSynchronize(e->stream));
        std::vector<std::pair<float, float>> iv;           // (start, stop) of every launch relative to the first one's start
        iv.reserve(e->prof_used);
        for (size_t i = 0; i < e->prof_used; ++i) {
            float t = 0.f, t0 = 0.f;
            if (hipEventElapsedTime(&t, e->prof_pairs[i].a, e->prof_pairs[i].b) != hipSuccess) continue;
            ProfCat& c = e->prof[VC_PROF_CONV];
            const vc_engine::ProfPair& pp = e->prof_pairs[i];
            const double rows = pp.rows ? (double)*pp.rows : 0.0;              // the pass's D2H copy of the row count has completed (streams synchronised above)
            const double fe = pp.flops + rows * pp.flops_per_row, be = pp.bytes + rows * pp.bytes_per_row;
            c.ms += t; c.flops += fe; c.bytes += be; c.launches += 1;
            c.flops_dense += pp.flops_dense >= 0 ? pp.flops_dense : fe; c.bytes_dense += pp.bytes_dense >= 0 ? pp.bytes_dense : be;
            if (hipEventElapsedTime(&t0, e->prof_pairs[0].a, e->prof_pairs[i].a) == hipSuccess) iv.emplace_back(t0, t0 + t);
        }
        // how much of the wall-clock window between the first start and the last stop had at least one conv kernel running
        e->prof_conv_union_ms = e->prof_conv_span_ms = 0;
        if (!iv.empty()) {
            std::sort(iv.begin(), iv.end());
            float lo = iv[0].first, hi = iv[0].second, cs = iv[0].first, ce = iv[0].second;
            double uni = 0;
            for (size_t i = 1; i < iv.size(); ++i) {
                lo = std::min(lo, iv[i].first); hi = std::max(hi, iv[i].second);
                if (iv[i].first > ce) { uni += ce - cs; cs = iv[i].first; ce = iv[i].second; }
                else ce = std::max(ce, iv[i].second);
            }
            uni += ce - cs;
            e->prof_conv_union_ms = uni; e->prof_conv_span_ms = hi - lo;
        }
        e->prof_used = 0;
    }
    if (ms) *ms = e->prof[cat].ms;
    if (launches) *launches = e->prof[cat].launches;
    if (flops) *flops = e->prof[cat].flops;
    if (bytes) *bytes = e->prof[cat].bytes;
    return VC_OK;
}

// The same category with the sparse Detect head credited as the DENSE head it replaces (the algorithmic work of the reference's
// Detect.m[i] over every pixel): kept apart from vc_profile_read, whose figures are the work the launches execute.
int vc_profile_read_dense(vc_engine* e, int cat, double* flops_dense, double* bytes_dense) {
    VC_CHECK(e && cat >= 0 && cat < VC_PROF_NCAT && flops_dense && bytes_dense, VC_ERR_ARG, "bad argument");
    VC_CHECK(!(cat == VC_PROF_CONV && e->prof_used > 0), VC_ERR_STATE, "call vc_profile_read first (it resolves the in-flight event pairs)");
    *flops_dense = e->prof[cat].flops_dense; *bytes_dense = e->prof[cat].bytes_dense;
    return VC_OK;
}

int vc_profile_conv_busy(vc_engine* e, double* union_ms, double* span_ms) {
    VC_CHECK(e && union_ms && span_ms, VC_ERR_ARG, "null argument");
    *union_ms = e->prof_conv_union_ms; *span_ms = e->prof_conv_span_ms;
    return VC_OK;
}

// ---- single-function entry points --------------------------------------------------------------------------
int vc_conv2d_host(const vc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res, float* y) {
    VC_CHECK(d && x && w && bias && y, VC_ERR_ARG, "null argument");
    const int prec = d->precision, es = elem_size(prec), ch = prec == PREC_F32 ? 4 : prec == PREC_FP8 ? 16 : 8;
    vc_engine tmp;                           // only used as an allocation list
    ConvParam p;
    p.name = "host"; p.O = d->cout; p.I = d->cin; p.kh = d->kh; p.kw = d->kw; p.set = true;
    p.w.assign(w, w + (size_t)d->cout * d->cin * d->kh * d->kw);
    p.b.assign(bias, bias + d->cout);
    int st = pack_and_upload(&tmp, p, prec);
    const int cin_eff = p.cin_eff;
    const int Ho = (d->h + 2 * d->pad - d->kh) / d->stride + 1, Wo = (d->w + 2 * d->pad - d->kw) / d->stride + 1;
    const size_t npix_in = (size_t)d->b * d->h * d->w, npix_out = (size_t)d->b * Ho * Wo;
    const int cout_s = round_up(d->cout, prec == PREC_FP8 ? 8 : 4);
    void *dx = nullptr, *dy = nullptr, *dr = nullptr;
    auto upload = [&](const float* src, size_t npix, int C, int Cs, void** dst) -> int {
        std::vector<uint8_t> buf(npix * Cs * es, 0);
        for (size_t i = 0; i < npix; ++i)
            for (int c = 0; c < C; ++c) {
                if (es == 4) ((float*)buf.data())[i * Cs + c] = src[i * C + c];
                else if (es == 2) ((uint16_t*)buf.data())[i * Cs + c] = f32_to_bf16(src[i * C + c]);
                else buf[i * Cs + c] = f32_to_e4m3(src[i * C + c]);                 // activation scale 1
            }
        VC_TRY(dev_alloc(&tmp, dst, buf.size()));
        VC_HIP(hipMemcpy(*dst, buf.data(), buf.size(), hipMemcpyHostToDevice));
        return VC_OK;
    };
    if (st == VC_OK) st = upload(x, npix_in, d->cin, cin_eff, &dx);
    if (st == VC_OK && res) st = upload(res, npix_out, d->cout, cout_s, &dr);
    if (st == VC_OK) st = dev_alloc(&tmp, &dy, npix_out * cout_s * es);
    if (st == VC_OK) {
        ConvP c{};
        c.in = dx; c.w = p.d_w; c.bias = p.d_b; c.res = dr; c.out = dy;
        c.B = d->b; c.H = d->h; c.W = d->w; c.Cin = cin_eff; c.in_cs = cin_eff; c.in_co = 0;
        c.Ho = Ho; c.Wo = Wo; c.Cout = d->cout; c.out_cs = cout_s; c.out_co = 0; c.res_cs = cout_s; c.res_co = 0;
        c.kh = d->kh; c.kw = d->kw; c.sh = c.sw = d->stride; c.ph = c.pw = d->pad;
        c.K = p.K; c.Kp = p.Kp; c.act = d->act; c.res_mode = res ? d->res_mode : RES_NONE; c.out_f32 = 0; c.prec = prec;
        c.scale = p.d_scale; c.act_scale = 1.0f; c.inv_act_scale = 1.0f; c.out_bf16 = 0;
        if (prec == PREC_FP8) c.Cout = round_up(d->cout, 8);      // the fp8 epilogue stores 8 channels at a time (weight / bias / scale rows are zero-padded)
        c.M = d->b * Ho * Wo;
        c.cfg = getenv("VC_CONV_CFG") ? atoi(getenv("VC_CONV_CFG")) : -1;
        c.ablate = getenv("VC_CONV_ABLATE") ? atoi(getenv("VC_CONV_ABLATE")) : 0;
        c.slots = getenv("VC_CONV_SLOTS") ? atoi(getenv("VC_CONV_SLOTS")) : 0;       // tests: a short persistent grid walks all tiles
        long long* dbg = nullptr;
        const size_t dbg_n = (size_t)1 << 16;
        if (getenv("VC_CONV_DBG") && hipMalloc((void**)&dbg, dbg_n * 64) == hipSuccess) { hipMemset(dbg, 0, dbg_n * 64); c.dbg = dbg; }
        st = launch_conv(c, nullptr);
        if (st == VC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("conv kernel failed: %s", hipGetErrorString(hipGetLastError())); st = VC_ERR_HIP; }
        if (st == VC_OK && getenv("VC_CONV_TIME")) {      // diagnostics: the launch alone on the GPU, best and mean of n repetitions
            const int reps = std::max(1, atoi(getenv("VC_CONV_TIME")));
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e30f, sum = 0.f;
            for (int r = 0; r < reps; ++r) {
                hipEventRecord(e0, nullptr);
                launch_conv(c, nullptr);
                hipEventRecord(e1, nullptr);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms); sum += ms;
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
            fprintf(stderr, "[vc conv time] cfg %d M %d N %d K %d: best %.4f ms mean %.4f ms, %.1f TFLOP/s\n", c.cfg, c.M, c.Cout, c.K, best, sum / reps,
                    2.0 * c.M * c.Cout * c.K / (best * 1e-3) / 1e12);
        }
        if (dbg) {       // per-workgroup phase times (100 MHz timestamps), averaged
            std::vector<long long> h(dbg_n * 8);
            hipMemcpy(h.data(), dbg, dbg_n * 64, hipMemcpyDeviceToHost);
            double acc[4] = {0, 0, 0, 0}; long n = 0; long long lo = INT64_MAX, hi = 0;
            for (size_t b = 0; b < dbg_n; ++b) {
                const long long* t = &h[b * 8];
                if (!t[0] || !t[4]) continue;
                for (int i = 0; i < 4; ++i) acc[i] += (double)(t[i + 1] - t[i]);
                lo = std::min(lo, t[0]); hi = std::max(hi, t[4]); ++n;
            }
            std::vector<double> st, du;
            for (size_t b = 0; b < dbg_n; ++b) { const long long* t = &h[b * 8]; if (t[0] && t[4]) { st.push_back((double)(t[0] - lo) / 100); du.push_back((double)(t[4] - t[0]) / 100); } }
            std::sort(st.begin(), st.end()); std::sort(du.begin(), du.end());
            if (n) fprintf(stderr, "[vc conv dbg] start offsets us p10/p50/p90/max %.1f %.1f %.1f %.1f ; lifetimes us p10/p50/p90/max %.1f %.1f %.1f %.1f\n",
                           st[n / 10], st[n / 2], st[n * 9 / 10], st[n - 1], du[n / 10], du[n / 2], du[n * 9 / 10], du[n - 1]);
            if (n) fprintf(stderr, "[vc conv dbg] %ld workgroups, us per workgroup: prologue %.2f first-tile %.2f k-loop %.2f epilogue %.2f ; first start -> last end %.2f us\n",
                           n, acc[0] / n / 100, acc[1] / n / 100, acc[2] / n / 100, acc[3] / n / 100, (double)(hi - lo) / 100);
            for (int w = 0; w < 4; ++w) {   // conv3x3_halo_v2_kernel: per-step cycle stamps of four workgroups
                const long long* tr = &h[400000 + w * 4096];
                if (!tr[0]) continue;
                int steps = 0; while (steps < 800 && tr[steps * 5]) ++steps;
                double d[5] = {0, 0, 0, 0, 0}; int n3 = 0;
                for (int k = 9; k + 1 < steps; ++k, ++n3) { for (int j = 0; j < 4; ++j) d[j] += (double)(tr[k * 5 + j + 1] - tr[k * 5 + j]); d[4] += (double)(tr[(k + 1) * 5] - tr[k * 5 + 4]); }
                if (n3) fprintf(stderr, "[vc conv dbg] wg %d: cycles per step (steps 9..%d), stamp intervals 0-1 %.0f, 1-2 %.0f, 2-3 %.0f, 3-4 %.0f, 4-next %.0f\n", w * 128, steps - 2, d[0] / n3, d[1] / n3, d[2] / n3, d[3] / n3, d[4] / n3);
                if (steps > 30) { fprintf(stderr, "[vc conv dbg]   steps 18..26 total cycles:"); for (int k = 18; k < 27; ++k) fprintf(stderr, " %lld", tr[(k + 1) * 5] - tr[k * 5]); fprintf(stderr, "\n"); }
            }
            hipFree(dbg);
        }
    }
    if (st == VC_OK) {
        std::vector<uint8_t> buf(npix_out * cout_s * es);
        if (hipMemcpy(buf.data(), dy, buf.size(), hipMemcpyDeviceToHost) != hipSuccess) { set_error("copy back failed"); st = VC_ERR_HIP; }
        for (size_t i = 0; i < npix_out && st == VC_OK; ++i)
            for (int c = 0; c < d->cout; ++c)
                y[i * d->cout + c] = es == 4 ? ((const float*)buf.data())[i * cout_s + c] : es == 2 ? bf16_to_f32(((const uint16_t*)buf.data())[i * cout_s + c]) : e4m3_to_f32(buf[i * cout_s + c]);
    }
    for (void* q : tmp.allocs) hipFree(q);
    tmp.allocs.clear();
    return st;
}

int vc_letterbox_host(const uint8_t* rgb, int h, int w, int net_h, int net_w, int precision, float* out) {
    VC_CHECK(rgb && out, VC_ERR_ARG, "null argument");
    vc_engine tmp;
    tmp.prec = precision;
    const LetterboxGeom g = letterbox_geom(h, w, net_h, net_w, false);
    void *ds = nullptr, *dd = nullptr;
    const int es = elem_size(precision);
    int st = dev_alloc(&tmp, &ds, (size_t)h * w * 3);
    if (st == VC_OK) st = dev_alloc(&tmp, &dd, (size_t)net_h * net_w * 4 * es);
    if (st == VC_OK && hipMemcpy(ds, rgb, (size_t)h * w * 3, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload failed"); st = VC_ERR_HIP; }
    if (st == VC_OK) st = launch_letterbox((const uint8_t*)ds, dd, 1, g, precision, nullptr);
    if (st == VC_OK) {
        std::vector<uint8_t> buf((size_t)net_h * net_w * 4 * es);
        if (hipMemcpy(buf.data(), dd, buf.size(), hipMemcpyDeviceToHost) != hipSuccess) { set_error("letterbox failed: %s", hipGetErrorString(hipGetLastError())); st = VC_ERR_HIP; }
        for (size_t i = 0; i < (size_t)net_h * net_w && st == VC_OK; ++i)
            for (int c = 0; c < 3; ++c)
                out[i * 3 + c] = es == 4 ? ((const float*)buf.data())[i * 4 + c] : bf16_to_f32(((const uint16_t*)buf.data())[i * 4 + c]);
    }
    for (void* q : tmp.allocs) hipFree(q);
    tmp.allocs.clear();
    return st;
}

}  // extern "C"
