// Model graphs of the EEG2Video generation hot path on one HIP stream (see model.h).
#include "model.h"
#include "h16.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace e2v;

// =====================================================================================================
// expected state-dict keys (SURVEY App. D for the UNet, App. C.5 for the VAE) -- mirrors
// eeg2video_amd/weights.py; the test-suite checks both against the reference's own state dict.
// =====================================================================================================
namespace {

struct KeySink {
    e2v_ctx* c;
    void add(const std::string& k, std::vector<int64_t> shape) {
        WTensor t;
        t.shape = std::move(shape);
        t.numel = 1;
        for (auto d : t.shape) t.numel *= (size_t)d;
        c->keys.push_back(k);
        c->raw.emplace(k, std::move(t));
    }
    void conv(const std::string& n, int co, int ci, int k) { add(n + ".weight", {co, ci, k, k}); add(n + ".bias", {co}); }
    void lin(const std::string& n, int co, int ci, bool bias = true) {
        add(n + ".weight", {co, ci});
        if (bias) add(n + ".bias", {co});
    }
    void norm(const std::string& n, int ch) { add(n + ".weight", {ch}); add(n + ".bias", {ch}); }
    void resnet3d(const std::string& p, int cin, int cout, int temb) {      // resnet.py:141-172
        norm(p + ".norm1", cin); conv(p + ".conv1", cout, cin, 3);
        if (temb > 0) lin(p + ".time_emb_proj", cout, temb);
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3);
        if (cin != cout) conv(p + ".conv_shortcut", cout, cin, 1);
    }
    void transformer3d(const std::string& p, int ch, int cross) {            // attention.py:58-87,158-202
        norm(p + ".norm", ch); conv(p + ".proj_in", ch, ch, 1);
        const std::string b = p + ".transformer_blocks.0";
        const char* names[3] = {"attn1", "attn2", "attn_temp"};
        const int kv[3] = {ch, cross, ch};
        for (int i = 0; i < 3; ++i) {
            const std::string a = b + "." + names[i];
            lin(a + ".to_q", ch, ch, false); lin(a + ".to_k", ch, kv[i], false); lin(a + ".to_v", ch, kv[i], false);
            lin(a + ".to_out.0", ch, ch);
        }
        for (const char* n : {"norm1", "norm2", "norm3", "norm_temp"}) norm(b + "." + n, ch);
        lin(b + ".ff.net.0.proj", 8 * ch, ch); lin(b + ".ff.net.2", ch, 4 * ch);
        conv(p + ".proj_out", ch, ch, 1);
    }
    void vae_mid(const std::string& p, int ch) {
        resnet3d(p + ".resnets.0", ch, ch, 0);
        const std::string a = p + ".attentions.0";
        norm(a + ".group_norm", ch);
        for (const char* n : {"query", "key", "value", "proj_attn"}) lin(a + "." + n, ch, ch);
        resnet3d(p + ".resnets.1", ch, ch, 0);
    }
};

std::string idx(const std::string& a, int i, const std::string& b, int j) {
    return a + "." + std::to_string(i) + "." + b + "." + std::to_string(j);
}

}  // namespace

void e2v_ctx::expected_keys() {
    KeySink s{this};
    const int* boc = cfg.block_out_channels;
    const int temb = boc[0] * 4, cross = cfg.cross_attention_dim, L = cfg.layers_per_block;
    s.conv("conv_in", boc[0], cfg.in_channels, 3);
    s.lin("time_embedding.linear_1", temb, boc[0]);
    s.lin("time_embedding.linear_2", temb, temb);
    int out_c = boc[0];
    for (int i = 0; i < 4; ++i) {                      // unet.py:113-139; blocks 0-2 carry attention
        const int in_c = out_c;
        out_c = boc[i];
        for (int j = 0; j < L; ++j) {
            s.resnet3d(idx("down_blocks", i, "resnets", j), j == 0 ? in_c : out_c, out_c, temb);
            if (i < 3) s.transformer3d(idx("down_blocks", i, "attentions", j), out_c, cross);
        }
        if (i != 3) s.conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out_c, out_c, 3);
    }
    s.resnet3d("mid_block.resnets.0", boc[3], boc[3], temb);
    s.transformer3d("mid_block.attentions.0", boc[3], cross);
    s.resnet3d("mid_block.resnets.1", boc[3], boc[3], temb);
    const int rev[4] = {boc[3], boc[2], boc[1], boc[0]};
    out_c = rev[0];
    for (int i = 0; i < 4; ++i) {                      // unet.py:164-202; blocks 1-3 carry attention
        const int prev = out_c;
        out_c = rev[i];
        const int in_c = rev[std::min(i + 1, 3)];
        for (int j = 0; j < L + 1; ++j) {
            const int skip = (j == L) ? in_c : out_c;  // unet_blocks.py:431-432
            const int rin = (j == 0) ? prev : out_c;
            s.resnet3d(idx("up_blocks", i, "resnets", j), rin + skip, out_c, temb);
            if (i > 0) s.transformer3d(idx("up_blocks", i, "attentions", j), out_c, cross);
        }
        if (i != 3) s.conv("up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out_c, out_c, 3);
    }
    s.norm("conv_norm_out", boc[0]);
    s.conv("conv_out", cfg.out_channels, boc[0], 3);

    // ---- VAE ("vae." prefix) ----
    const int* vb = cfg.vae_block_out_channels;
    const int VL = cfg.vae_layers_per_block, lat = cfg.vae_latent_channels;
    s.conv("vae.encoder.conv_in", vb[0], cfg.vae_in_channels, 3);
    out_c = vb[0];
    for (int i = 0; i < 4; ++i) {
        const int in_c = out_c;
        out_c = vb[i];
        for (int j = 0; j < VL; ++j) s.resnet3d(idx("vae.encoder.down_blocks", i, "resnets", j), j == 0 ? in_c : out_c, out_c, 0);
        if (i != 3) s.conv("vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out_c, out_c, 3);
    }
    s.vae_mid("vae.encoder.mid_block", vb[3]);
    s.norm("vae.encoder.conv_norm_out", vb[3]);
    s.conv("vae.encoder.conv_out", 2 * lat, vb[3], 3);
    const int vrev[4] = {vb[3], vb[2], vb[1], vb[0]};
    s.conv("vae.decoder.conv_in", vrev[0], lat, 3);
    s.vae_mid("vae.decoder.mid_block", vrev[0]);
    out_c = vrev[0];
    for (int i = 0; i < 4; ++i) {
        const int in_c = out_c;
        out_c = vrev[i];
        for (int j = 0; j < VL + 1; ++j) s.resnet3d(idx("vae.decoder.up_blocks", i, "resnets", j), j == 0 ? in_c : out_c, out_c, 0);
        if (i != 3) s.conv("vae.decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out_c, out_c, 3);
    }
    s.norm("vae.decoder.conv_norm_out", vb[0]);
    s.conv("vae.decoder.conv_out", cfg.vae_in_channels, vb[0], 3);
    s.conv("vae.quant_conv", 2 * lat, 2 * lat, 1);
    s.conv("vae.post_quant_conv", lat, lat, 1);

    // ---- Semantic Predictor ("semantic." prefix; nn.Sequential indices of train_semantic_predictor.py:14-28) ----
    if (cfg.sem_in_features > 0 && cfg.sem_hidden > 0) {
        const int dims[6] = {cfg.sem_in_features, cfg.sem_hidden, cfg.sem_hidden, cfg.sem_hidden, cfg.sem_hidden,
                             cfg.sem_tokens * cfg.cross_attention_dim};
        for (int i = 0; i < 5; ++i) s.lin("semantic.mlp." + std::to_string(2 * i), dims[i + 1], dims[i]);
    }
}

float* e2v_ctx::dev_alloc(size_t floats) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(floats, 1) * sizeof(float);
    if (dry_run()) return dry_fake_ptr(bytes);               // e2v_op_describe_dispatch: an address nobody dereferences, not owned
    E2V_HIP(hipMalloc(&p, bytes));
    (alloc_part >= 0 ? owned_part[alloc_part] : owned).push_back(static_cast<float*>(p));
    owned_bytes[p] = bytes;
    weight_bytes += bytes;
    return static_cast<float*>(p);
}

void e2v_ctx::free_part(int part) {
    for (float* p : owned_part[part]) {
        auto it = owned_bytes.find(p);
        if (it != owned_bytes.end()) { weight_bytes -= it->second; owned_bytes.erase(it); }
        (void)hipFree(p);
    }
    owned_part[part].clear();
}

bool e2v_ctx::conv_has_wino(const ConvW& w, int m) const {
    if (!w.stride1 || conv_algo == 1) return false;
    const int cmin = std::min(w.cin, w.cout);
    if (m == 2) return conv_algo == 2 || (conv_algo == 0 && cmin >= wino_min_c);
    return conv_algo == 3 || (conv_algo == 0 && wino_f4 && cmin >= wino4_min_c);
}

// First use of a layout: allocate it under the owning finalize() group (freed when that part is finalized again) and derive
// it from the torch-layout weight on the calling stream -- ordered before the launch that asked for it.
void e2v_ctx::conv_form(const ConvW& w, ConvForm f, hipStream_t s) {
    const int saved = alloc_part;
    struct Restore { e2v_ctx* c; int v; ~Restore() { c->alloc_part = v; } } restore{this, saved};
    alloc_part = w.part;
    auto split3 = [&](const float* u, size_t n) -> const void* {
        float* d = dev_alloc((3 * n + 1) / 2);
        split_bf16x3(u, d, n, n, s);
        return d;
    };
    switch (f) {
        case FORM_DIRECT32:
            if (!w.w) {
                float* d = dev_alloc((size_t)w.cout * w.ldw);
                pack_conv3x3(w.raw, d, w.cout, w.cin, 32, s);
                w.w = d;
            }
            break;
        case FORM_BF16:
        case FORM_F16: {
            const void*& dst = f == FORM_F16 ? w.w16h : w.w16;
            if (!dst) {
                const size_t n = (size_t)w.cout * w.ldw16;
                Act tmp(pool, (int64_t)((n + 1023) / 1024), 1024);            // fp32 staging of the 64-channel-chunk layout
                pack_conv3x3(w.raw, tmp.p, w.cout, w.cin, 64, s);
                float* d16 = dev_alloc((n + 1) / 2);
                to_h16(tmp.p, d16, n, f == FORM_F16 ? H16_FP16 : H16_BF16, s);
                dst = d16;
            }
            break;
        }
        case FORM_BF16_UP2:
        case FORM_F16_UP2: {
            const void*& dst = f == FORM_F16_UP2 ? w.w16h_up2 : w.w16_up2;
            if (!dst) {
                const size_t n = 4 * (size_t)w.cout * conv_up2x_packed_ld(w.cin);
                Act tmp(pool, (int64_t)((n + 1023) / 1024), 1024);            // fp32 staging: the tap sums are formed in fp32, rounded once
                pack_conv_up2x(w.raw, tmp.p, w.cout, w.cin, s);
                float* d16 = dev_alloc((n + 1) / 2);
                to_h16(tmp.p, d16, n, f == FORM_F16_UP2 ? H16_FP16 : H16_BF16, s);
                dst = d16;
            }
            break;
        }
        case FORM_WINO2:
        case FORM_WINO4: {
            const int m = f == FORM_WINO4 ? 4 : 2;
            const size_t n = (size_t)(m + 2) * (m + 2) * w.cout * w.cin;
            const float*& u = m == 4 ? w.wino4 : w.wino;
            const void*& u3 = m == 4 ? w.wino4_x3 : w.wino_x3;
            if (!u) {
                float* d = dev_alloc(n);
                wino_pack_weights(w.raw, d, w.cout, w.cin, m, s);
                u = d;
            }
            if (x3_compute && !u3) u3 = split3(u, n);
            break;
        }
    }
}

// fp16 mode: the IEEE-half copy of a linear's weight, derived on first use from the fp32 matrix finalize() kept (same layout as the
// bf16 copy: rows zero-padded to in16), filed under the finalize() group that owns the layer.
const void* e2v_ctx::lin_f16(const LinW& w, hipStream_t s) {
    if (w.w16h) return w.w16h;
    const int saved = alloc_part;
    struct Restore { e2v_ctx* c; int v; ~Restore() { c->alloc_part = v; } } restore{this, saved};
    alloc_part = w.part;
    const size_t n = (size_t)w.out * w.in16;
    float* d16 = dev_alloc((n + 1) / 2);
    if (w.in16 == w.in) {
        to_h16(w.w, d16, n, H16_FP16, s);
    } else {
        Act tmp(pool, (int64_t)((n + 1023) / 1024), 1024);
        pad_cols(w.w, w.in, tmp.p, w.in16, w.out, s);
        to_h16(tmp.p, d16, n, H16_FP16, s);
    }
    w.w16h = d16;
    return d16;
}

void e2v_ctx::enter_stream(hipStream_t s) {
    if (has_last_stream && s != last_stream) {
        if (!stream_ev) E2V_HIP(hipEventCreateWithFlags(&stream_ev, hipEventDisableTiming));
        E2V_HIP(hipEventRecord(stream_ev, last_stream));
        E2V_HIP(hipStreamWaitEvent(s, stream_ev, 0));
    }
    last_stream = s;
    has_last_stream = true;
}

// =====================================================================================================
// finalize: torch layouts -> kernel layouts
// =====================================================================================================
namespace {

struct Packer {
    e2v_ctx* c;
    hipStream_t s = nullptr;
    const WTensor& t(const std::string& k) {
        auto it = c->raw.find(k);
        E2V_REQUIRE(it != c->raw.end(), E2V_ENOWEIGHT, "unknown key " + k);
        if (dry_run()) {                                     // shapes only: the dispatch of a configuration does not depend on values
            if (!it->second.d) it->second.d = dry_fake_ptr(it->second.numel * sizeof(float));
            return it->second;
        }
        E2V_REQUIRE(it->second.loaded, E2V_ENOWEIGHT, "state-dict key not loaded: " + k);
        return it->second;
    }
    const void* half(const float* w, size_t n) {             // bf16 copy of a (packed) weight for the bf16-MFMA mode
        float* d = c->dev_alloc((n + 1) / 2);
        to_bf16(w, d, n, s);
        return d;
    }
    // every form of one [out][in] matrix: fp32 as it is, bf16 with rows padded to a multiple of 8 (16-byte pieces), f32x3 planes
    LinW mk_lin(const float* w, const float* b, int in, int out) {
        const int in16 = (in + 7) / 8 * 8;
        const void* w16;
        if (in16 == in) {
            w16 = half(w, (size_t)out * in);
        } else {
            float* tmp = nullptr;
            E2V_HIP(hipMalloc((void**)&tmp, (size_t)out * in16 * sizeof(float)));
            pad_cols(w, in, tmp, in16, out, s);
            w16 = half(tmp, (size_t)out * in16);
            E2V_HIP(hipStreamSynchronize(s));
            if (!dry_run()) (void)hipFree(tmp);
        }
        LinW l{w, b, in, out, w16, split3(w, (size_t)out * in)};
        l.in16 = in16;
        l.part = c->alloc_part >= 0 ? c->alloc_part : 0;
        return l;
    }
    const void* split3(const float* w, size_t n) {           // three bf16 planes (f32x3 mode only)
        if (!c->x3_compute) return nullptr;
        float* d = c->dev_alloc((3 * n + 1) / 2);
        split_bf16x3(w, d, n, n, s);
        return d;
    }
    NormW norm(const std::string& n) { return NormW{t(n + ".weight").d, t(n + ".bias").d, (int)t(n + ".weight").shape[0]}; }
    LinW lin(const std::string& n, bool bias = true) {       // Linear or 1x1 conv: [out][in] as it is
        const WTensor& w = t(n + ".weight");
        return mk_lin(w.d, bias ? t(n + ".bias").d : nullptr, (int)w.shape[1], (int)w.shape[0]);
    }
    // 3x3 conv: the torch-layout weight stays; kernel layouts are built on first use (e2v_ctx::conv_form)
    ConvW conv3(const std::string& n, bool stride1 = true) {
        const WTensor& w = t(n + ".weight");
        ConvW cw;
        cw.raw = w.d; cw.b = t(n + ".bias").d;
        cw.cout = (int)w.shape[0]; cw.cin = (int)w.shape[1];
        cw.cin_pad = (cw.cin + 3) / 4 * 4; cw.cin_pad16 = (cw.cin + 7) / 8 * 8;
        cw.ldw = conv3x3_packed_ld(cw.cin, 32); cw.ldw16 = conv3x3_packed_ld(cw.cin, 64);
        cw.stride1 = stride1 && cw.cin % 4 == 0 && cw.cout % 4 == 0;
        cw.part = c->alloc_part >= 0 ? c->alloc_part : 0;
        return cw;
    }
    LinW fuse_rows(const std::vector<std::string>& names, bool bias) {   // stack Linear weights along `out`
        int in = 0, out = 0;
        for (auto& n : names) { const WTensor& w = t(n + ".weight"); in = (int)w.shape[1]; out += (int)w.shape[0]; }
        float* d = c->dev_alloc((size_t)out * in);
        float* b = bias ? c->dev_alloc(out) : nullptr;
        int r = 0;
        for (auto& n : names) {
            const WTensor& w = t(n + ".weight");
            copy_rows(w.d, in, d + (size_t)r * in, in, (int)w.shape[0], in, s);
            if (bias) copy_rows(t(n + ".bias").d, (int)w.shape[0], b + r, (int)w.shape[0], 1, (int)w.shape[0], s);
            r += (int)w.shape[0];
        }
        return mk_lin(d, b, in, out);
    }
    LinW geglu(const std::string& n) {        // rows [value 0..4C) | gate 0..4C)] -> per 64: [32 value | 32 gate]
        const WTensor& w = t(n + ".weight");
        const WTensor& bsrc = t(n + ".bias");
        const int out = (int)w.shape[0], in = (int)w.shape[1], half = out / 2;
        E2V_REQUIRE(half % 32 == 0, E2V_EINVAL, "GEGLU inner width must be a multiple of 32");
        float* d = c->dev_alloc((size_t)out * in);
        float* b = c->dev_alloc(out);
        for (int q = 0; q < half / 32; ++q) {
            copy_rows(w.d + (size_t)(q * 32) * in, in, d + (size_t)(q * 64) * in, in, 32, in, s);
            copy_rows(w.d + (size_t)(half + q * 32) * in, in, d + (size_t)(q * 64 + 32) * in, in, 32, in, s);
            copy_rows(bsrc.d + q * 32, 32, b + q * 64, 32, 1, 32, s);
            copy_rows(bsrc.d + half + q * 32, 32, b + q * 64 + 32, 32, 1, 32, s);
        }
        return mk_lin(d, b, in, out);
    }
    ResW resnet(const std::string& p, bool temb) {
        ResW r;
        r.n1 = norm(p + ".norm1"); r.c1 = conv3(p + ".conv1");
        if (temb) r.temb = lin(p + ".time_emb_proj");
        r.n2 = norm(p + ".norm2"); r.c2 = conv3(p + ".conv2");
        r.cin = r.c1.cin; r.cout = r.c1.cout;
        if (c->raw.count(p + ".conv_shortcut.weight")) r.sc = lin(p + ".conv_shortcut");
        return r;
    }
    TransW transformer(const std::string& p) {
        TransW w;
        const std::string b = p + ".transformer_blocks.0";
        w.norm = norm(p + ".norm"); w.proj_in = lin(p + ".proj_in"); w.proj_out = lin(p + ".proj_out");
        w.ln1 = norm(b + ".norm1"); w.ln2 = norm(b + ".norm2"); w.ln3 = norm(b + ".norm3"); w.lnt = norm(b + ".norm_temp");
        w.a1_qkv = fuse_rows({b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"}, false);
        w.a1_out = lin(b + ".attn1.to_out.0");
        w.a2_q = lin(b + ".attn2.to_q", false);
        w.a2_kv = fuse_rows({b + ".attn2.to_k", b + ".attn2.to_v"}, false);
        w.a2_out = lin(b + ".attn2.to_out.0");
        w.ff1 = geglu(b + ".ff.net.0.proj"); w.ff2 = lin(b + ".ff.net.2");
        w.at_qkv = fuse_rows({b + ".attn_temp.to_q", b + ".attn_temp.to_k", b + ".attn_temp.to_v"}, false);
        w.at_out = lin(b + ".attn_temp.to_out.0");
        w.C = w.norm.c;
        return w;
    }
    VAEAttnW vae_attn(const std::string& a) {
        VAEAttnW w;
        w.norm = norm(a + ".group_norm");
        w.qkv = fuse_rows({a + ".query", a + ".key", a + ".value"}, true);
        w.proj = lin(a + ".proj_attn");
        w.C = w.norm.c;
        return w;
    }
};

}  // namespace

void e2v_ctx::finalize(int which) {
    Packer P{this};
    const int L = cfg.layers_per_block;
    struct PartGuard { e2v_ctx* c; ~PartGuard() { c->alloc_part = -1; } } part_guard{this};
    E2V_HIP(hipDeviceSynchronize());                 // a part finalized before may still be in use by queued work
    if (which & 1) {
        free_part(0);
        alloc_part = 0;
        UNetW u;
        u.conv_in = P.conv3("conv_in");
        u.te1 = P.lin("time_embedding.linear_1"); u.te2 = P.lin("time_embedding.linear_2");
        for (int i = 0; i < 4; ++i) {
            UNetW::Block b;
            for (int j = 0; j < L; ++j) {
                b.res.push_back(P.resnet(idx("down_blocks", i, "resnets", j), true));
                if (i < 3) b.attn.push_back(P.transformer(idx("down_blocks", i, "attentions", j)));
            }
            if (i != 3) { b.resample = true; b.rs = P.conv3("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", false); }
            u.down.push_back(std::move(b));
        }
        u.mid_r0 = P.resnet("mid_block.resnets.0", true);
        u.mid_attn = P.transformer("mid_block.attentions.0");
        u.mid_r1 = P.resnet("mid_block.resnets.1", true);
        for (int i = 0; i < 4; ++i) {
            UNetW::Block b;
            for (int j = 0; j < L + 1; ++j) {
                b.res.push_back(P.resnet(idx("up_blocks", i, "resnets", j), true));
                if (i > 0) b.attn.push_back(P.transformer(idx("up_blocks", i, "attentions", j)));
            }
            if (i != 3) { b.resample = true; b.rs = P.conv3("up_blocks." + std::to_string(i) + ".upsamplers.0.conv"); }
            u.up.push_back(std::move(b));
        }
        u.norm_out = P.norm("conv_norm_out");
        u.conv_out = P.conv3("conv_out");
        unet = std::move(u);
    }
    if (which & 2) {
        free_part(1);
        alloc_part = 1;
        VAEW v;
        const int VL = cfg.vae_layers_per_block;
        v.post_quant = P.lin("vae.post_quant_conv");
        v.dec_in = P.conv3("vae.decoder.conv_in");
        v.dec_mid0 = P.resnet("vae.decoder.mid_block.resnets.0", false);
        v.dec_attn = P.vae_attn("vae.decoder.mid_block.attentions.0");
        v.dec_mid1 = P.resnet("vae.decoder.mid_block.resnets.1", false);
        for (int i = 0; i < 4; ++i) {
            VAEW::Block b;
            for (int j = 0; j < VL + 1; ++j) b.res.push_back(P.resnet(idx("vae.decoder.up_blocks", i, "resnets", j), false));
            if (i != 3) { b.resample = true; b.rs = P.conv3("vae.decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv"); }
            v.dec_up.push_back(std::move(b));
        }
        v.dec_norm_out = P.norm("vae.decoder.conv_norm_out");
        v.dec_out = P.conv3("vae.decoder.conv_out");
        v.quant = P.lin("vae.quant_conv");
        v.enc_in = P.conv3("vae.encoder.conv_in");
        for (int i = 0; i < 4; ++i) {
            VAEW::Block b;
            for (int j = 0; j < VL; ++j) b.res.push_back(P.resnet(idx("vae.encoder.down_blocks", i, "resnets", j), false));
            if (i != 3) { b.resample = true; b.rs = P.conv3("vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", false); }
            v.enc_down.push_back(std::move(b));
        }
        v.enc_mid0 = P.resnet("vae.encoder.mid_block.resnets.0", false);
        v.enc_attn = P.vae_attn("vae.encoder.mid_block.attentions.0");
        v.enc_mid1 = P.resnet("vae.encoder.mid_block.resnets.1", false);
        v.enc_norm_out = P.norm("vae.encoder.conv_norm_out");
        v.enc_out = P.conv3("vae.encoder.conv_out");
        vae = std::move(v);
    }
    if (which & 4) {
        E2V_REQUIRE(cfg.sem_in_features > 0 && cfg.sem_hidden > 0, E2V_ESTATE, "the config has no semantic predictor");
        free_part(2);
        alloc_part = 2;
        sem.clear();
        // the first layer's K (310) is padded to a multiple of 8 (320 -> 312 would do for fp32's 16-byte rows, but the bf16
        // tile reads 8-element pieces): pad FIRST, then derive the bf16 / split copies from the padded matrix so that every
        // form of the layer has the same row length
        sem_in_pad = (cfg.sem_in_features + 7) / 8 * 8;
        for (int i = 0; i < 5; ++i) {
            const std::string n = "semantic.mlp." + std::to_string(2 * i);
            if (i == 0 && sem_in_pad != cfg.sem_in_features) {
                const WTensor& w = P.t(n + ".weight");
                const int out = (int)w.shape[0];
                float* wp = dev_alloc((size_t)out * sem_in_pad);
                pad_cols(w.d, cfg.sem_in_features, wp, sem_in_pad, out, nullptr);
                sem.push_back(P.mk_lin(wp, P.t(n + ".bias").d, sem_in_pad, out));
            } else {
                sem.push_back(P.lin(n));
            }
        }
    }
    alloc_part = -1;
    E2V_HIP(hipStreamSynchronize(nullptr));
    E2V_HIP(hipGetLastError());
    if (dry_run()) {                                         // nothing was allocated: nothing to drop
        if (which & 1) unet_ready = true;
        if (which & 2) vae_ready = true;
        return;
    }
    // the torch-layout copies of re-laid-out tensors are no longer needed
    auto drop = [&](const std::string& k) {
        auto it = raw.find(k);
        if (it != raw.end() && it->second.d) {
            (void)hipFree(it->second.d);
            weight_bytes -= it->second.numel * sizeof(float);
            it->second.d = nullptr;
            it->second.loaded = false;
        }
    };
    for (const auto& k : keys) {
        const bool is_vae = k.rfind("vae.", 0) == 0;
        if (k.rfind("semantic.", 0) == 0) continue;
        if ((is_vae && !(which & 2)) || (!is_vae && !(which & 1))) continue;
        const WTensor& w = raw[k];
        const bool conv3 = w.shape.size() == 4 && w.shape[2] == 3;
        const bool fused = k.find(".to_q.") != std::string::npos || k.find(".to_k.") != std::string::npos ||
                           k.find(".to_v.") != std::string::npos || k.find(".ff.net.0.proj.") != std::string::npos ||
                           k.find(".query.") != std::string::npos || k.find(".key.") != std::string::npos ||
                           k.find(".value.") != std::string::npos;
        const bool keep_q = k.find(".attn2.to_q.") != std::string::npos;
        (void)conv3;                            // 3x3 conv weights stay: their kernel layouts are derived from them on first use
        if (fused && !keep_q) drop(k);
    }
    if (which & 1) unet_ready = true;
    if (which & 2) vae_ready = true;
    if (which & 4) sem_ready = true;
}

// =====================================================================================================
// graph runner
// =====================================================================================================
namespace {

struct Geo { int nimg, H, W; };     // images (= n*F frames) and their map size

struct Runner {
    e2v_ctx* c;
    hipStream_t s;
    int res_idx = 0, tr_idx = 0;      // position in the per-generate caches (resnets with a time embedding / transformers)
    Pool& pool() { return c->pool; }
    // bf16-activation mode (e2v_set_compute_dtype(E2V_BF16)): every activation the graph stores in HBM is bf16; accumulation,
    // GroupNorm / LayerNorm statistics, softmax and the time-embedding rows stay fp32
    bool bf() const { return c->bf16_compute; }
    int h16() const { return c->bf16_compute ? c->h16_mode : 0; }      // the mode flag of every 16-bit launch: 1 bf16, 2 fp16 (h16.h)
    bool fp16() const { return h16() == H16_FP16; }
    const void* w16_of(const LinW& w) { return fp16() ? c->lin_f16(w, s) : w.w16; }
    Act new_act(int64_t rows, int C) { return Act(pool(), rows, C, bf()); }
    // fp32 rows -> bf16 rows padded to `cpad` channels (boundary tensors: latents, z, images, conditioning)
    Act to_act16(const float* x, int cols, int cpad, int64_t rows) {
        Act o(pool(), rows, cpad, true);
        pad_cols(x, cols, o.p, cpad, rows, s, h16());
        return o;
    }

    // Small-batch family: ask for split-K where the launch leaves most of the chip idle AND K is deep enough to pay for the fp32
    // partials (each run at least E2V_SPLITK_MIN_DEPTH deep: the partial planes cost 8 bytes per output and run, a run of depth d
    // 2 d flops per output).  Returns the workspace the launch needs (kept alive by the caller until the launch is queued).
    Act sk_setup(IgemmArgs& g) {
        Act ws;
        static const int* const on = knob("E2V_SPLITK", 1);
        static const int* const min_depth = E2V_AB_KNOB("E2V_SPLITK_MIN_DEPTH", 2048);     // (1024 / 640 and 400 tiles measured +-0: `make ab` only)
        static const int* const max_tiles = E2V_AB_KNOB("E2V_SPLITK_MAX_TILES", 256);
        if (!c->small_family || !g.a_bf16 || !*on || *min_depth < 64) return ws;
        const long K = (long)g.taps * (g.c0 + g.c1);
        const long tiles = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
        if (tiles >= *max_tiles || K < 2L * *min_depth) return ws;
        const int want = (int)std::min<long>(std::min<long>((512 + tiles - 1) / tiles, K / *min_depth), 16);
        const int runs = splitk_plan(g, want);
        if (runs < 2) return ws;
        ws = Act(pool(), (int64_t)runs * g.M, g.N);
        g.sk = want; g.sk_ws = ws.p;
        return ws;
    }

    void gn_ws(int samples, int P, int C) {
        const size_t need_p = (size_t)samples * groupnorm_chunks(P) * C * 2;
        const size_t need_s = (size_t)samples * C * 2;
        if (need_p > c->gn_part_floats) {
            E2V_HIP(hipStreamSynchronize(s));
            if (c->gn_part && !dry_run()) (void)hipFree(c->gn_part);
            E2V_HIP(hipMalloc((void**)&c->gn_part, need_p * sizeof(float)));
            if (dry_run()) c->gn_part = dry_fake_ptr(need_p * sizeof(float));       // (pointers steer the graph: a null workspace reads as "no GroupNorm in front")
            c->gn_part_floats = need_p;
        }
        if (need_s > c->gn_scale_floats) {
            E2V_HIP(hipStreamSynchronize(s));
            if (c->gn_scale && !dry_run()) (void)hipFree(c->gn_scale);
            E2V_HIP(hipMalloc((void**)&c->gn_scale, need_s * sizeof(float)));
            if (dry_run()) c->gn_scale = dry_fake_ptr(need_s * sizeof(float));
            c->gn_scale_floats = need_s;
        }
    }

    // rb0 / rb1: row-block sums that came with the source tensors (Act::rb; null: none) -- the statistics pass over such a source is
    // skipped (bf16 mode, P a multiple of 64)
    Act gn(const NormW& w, const float* x0, int c0, const float* x1, int c1, int samples, int P, int groups, float eps,
           bool act, const float* rb0 = nullptr, const float* rb1 = nullptr) {
        const int C = c0 + c1;
        E2V_REQUIRE(C == w.c && C % groups == 0 && c0 % 4 == 0 && c1 % 4 == 0, E2V_ESHAPE, "GroupNorm channel mismatch");
        gn_ws(samples, P, C);
        Act out = new_act((int64_t)samples * P, C);
        GroupNormArgs a;
        a.bf16 = h16();
        a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1; a.ld0 = c0; a.ld1 = c1;
        a.gamma = w.g; a.beta = w.b; a.out = out.p; a.ldo = C;
        a.samples = samples; a.P = P; a.groups = groups; a.eps = eps; a.silu = act ? 1 : 0;
        a.ws_part = c->gn_part; a.ws_scale = c->gn_scale;
        if (bf() && P % 64 == 0) { a.rb0 = rb0; a.rb1 = c1 > 0 ? rb1 : nullptr; }
        // (16-bit modes only: the fp32 mode keeps one summation order at every batch.  E2V_GN_FUSED_SMALL = 3: the one-kernel form in
        // BOTH families -- the A/B that priced it at B = 32, profiles/r05_shape_ab_b32_gn_fused.log)
        static const int* const gn_fused = knob("E2V_GN_FUSED_SMALL", 1);
        a.small_chunks = (c->small_family && bf()) ? 1 : 0;
        a.fused_small = (bf() && (c->small_family || *gn_fused == 3)) ? 1 : 0;
        groupnorm(a, s);
        return out;
    }

    // statistics only: (scale, shift) per (slab, channel) left in c->gn_scale for the conv's input transform
    void gn_stats(const NormW& w, const float* x0, int c0, const float* x1, int c1, int samples, int P, int groups, float eps) {
        const int C = c0 + c1;
        E2V_REQUIRE(C == w.c && C % groups == 0 && c0 % 4 == 0 && c1 % 4 == 0, E2V_ESHAPE, "GroupNorm channel mismatch");
        gn_ws(samples, P, C);
        GroupNormArgs a;
        a.bf16 = h16();
        a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1; a.ld0 = c0; a.ld1 = c1;
        a.gamma = w.g; a.beta = w.b; a.samples = samples; a.P = P; a.groups = groups; a.eps = eps;
        a.ws_part = c->gn_part; a.ws_scale = c->gn_scale;
        groupnorm_stats(a, s);
    }

    // does this conv run in Winograd form?  (fp32 arithmetic only: in bf16 the transforms would dominate and the
    // rounding of transformed inputs costs accuracy)
    // returns the output tile size: 0 = direct, 2 = F(2x2,3x3), 4 = F(4x4,3x3).  fp32 arithmetic only: in bf16 the
    // transforms would dominate and rounding the transformed inputs costs accuracy.
    int winograd(const ConvW& w, int stride, int pad, int Hi, int Wi, int Ho, int Wo) const {
        if (c->conv_algo == 1 || c->bf16_compute || stride != 1 || pad != 1 || Hi != Ho || Wi != Wo) return 0;
        if (c->conv_algo == 2) return c->conv_has_wino(w, 2) ? 2 : 0;
        if (c->conv_algo == 3) return c->conv_has_wino(w, 4) ? 4 : 0;
        const double padded = (double)((Ho + 3) / 4 * 4) * ((Wo + 3) / 4 * 4);
        if (c->conv_has_wino(w, 4) && padded <= c->wino_f4_pad * Ho * Wo) return 4;
        return c->conv_has_wino(w, 2) ? 2 : 0;
    }

    Act ln(const NormW& w, const Act& x) {
        E2V_REQUIRE(x.C == w.c && x.C % 4 == 0 && x.C <= 1280, E2V_ESHAPE, "LayerNorm width unsupported");
        Act out = new_act(x.rows, x.C);
        layernorm(x.p, x.C, w.g, w.b, out.p, x.C, (int)x.rows, x.C, 1e-5f, s, h16());
        return out;
    }

    // out[M][N] = A[M][K] W^T + b (+ resid); geglu: out[M][N/2].
    // bf16-activation mode: A, resid and out are bf16 unless `f32_io` (the time-embedding MLP: fp32 rows in, fp32 rows out,
    // fp32 arithmetic) or `out_f32` (a boundary tensor: VAE moments, attention scores)
    Act linear(const LinW& w, const float* a, int lda, int64_t M, const float* resid = nullptr, int ldr = 0,
               bool geglu = false, const float* a1 = nullptr, int c1 = 0, int lda1 = 0, bool f32_io = false, bool out_f32 = false) {
        const bool b16 = bf() && !f32_io;
        const int K = b16 ? w.in16 : w.in;
        const int K0 = K - c1;
        const int gran = b16 ? 8 : 4;
        E2V_REQUIRE(K0 > 0 && K0 % gran == 0 && c1 % gran == 0 && lda % gran == 0, E2V_ESHAPE, "linear: K must be a multiple of 4 (bf16: 8)");
        Act out(pool(), M, geglu ? w.out / 2 : w.out, b16 && !out_f32);
        IgemmArgs g;
        g.a0 = a; g.c0 = K0; g.lda0 = lda; g.a1 = a1; g.c1 = c1; g.lda1 = lda1;
        g.w = w.w; g.ldw = w.in; g.ldw16 = w.in16; g.out = out.p; g.ldc = out.C; g.bias = w.b;
        g.resid = resid; g.ldr = ldr; g.M = (int)M; g.N = w.out; g.taps = 1; g.geglu = geglu ? 1 : 0;
        if (b16) { g.w16 = w16_of(w); g.a_bf16 = h16(); g.out_f32 = out_f32 ? 1 : 0; g.resid_bf16 = resid ? 1 : 0; }
        if (c->x3_compute && w.w3) { g.x3 = 1; g.w3 = w.w3; g.w3_plane = (long long)w.out * w.in; }
        Act skws = sk_setup(g);
        igemm(g, s);
        return out;
    }

    // 3x3 conv over `geo.nimg` images; (Hi, Wi) = logical input size (after nearest resize), (Ho, Wo) output size
    // gn_P > 0: the input is the RAW tensor and GroupNorm's affine + SiLU (statistics already in c->gn_scale, slabs of
    // gn_P rows) is applied on the way in -- only valid when winograd() says so
    Act conv3(const ConvW& w, const float* x0, int c0, const float* x1, int c1, Geo geo, int Hi, int Wi, int Ho, int Wo,
              int stride, int pad, const float* rowbias = nullptr, int rows_per_sample = 1, const float* resid = nullptr,
              int gn_P = 0, int rb_ld = -1, bool out_f32 = false, bool want_rb = false) {   // rb_ld: stride between the samples' rowbias rows (0: one row for all)
        // want_rb: a GroupNorm follows -- leave the row-block sums with the output (Act::rb) where the tensor can carry them
        if (rb_ld < 0) rb_ld = w.cout;
        E2V_REQUIRE(c0 + c1 == (bf() ? w.cin_pad16 : w.cin_pad), E2V_ESHAPE, "conv: input channels do not match the weight");
        Act out(pool(), (int64_t)geo.nimg * Ho * Wo, w.cout, bf() && !out_f32);
        if (const int wm = winograd(w, stride, pad, Hi, Wi, Ho, Wo)) {
            WinoArgs a;
            a.m = wm;
            a.x0 = x0; a.c0 = c0; a.ld0 = c0; a.x1 = x1; a.c1 = c1; a.ld1 = c1;
            a.nimg = geo.nimg; a.Hs = geo.H; a.Ws = geo.W; a.Ho = Ho; a.Wo = Wo;
            if (Hi != geo.H || Wi != geo.W) {
                a.upsample = 1;
                a.ups_h = (float)geo.H / (float)Hi;        // torch: scale = (float)input_size / output_size
                a.ups_w = (float)geo.W / (float)Wi;
            }
            if (gn_P > 0) { a.gn_scsh = c->gn_scale; a.gn_P = gn_P; a.gn_silu = 1; }
            c->conv_form(w, wm == 4 ? e2v_ctx::FORM_WINO4 : e2v_ctx::FORM_WINO2, s);
            a.U = wm == 4 ? w.wino4 : w.wino; a.N = w.cout; a.out = out.p; a.ldc = w.cout; a.bias = w.b;
            if (c->x3_compute) a.U3 = wm == 4 ? w.wino4_x3 : w.wino_x3;
            a.rowbias = rowbias; a.rb_ld = rb_ld; a.rows_per_sample = rows_per_sample; a.resid = resid; a.ldr = w.cout;
            const int chunk = wino_chunk_images(a, c->wino_ws_floats);
            const size_t need = wino_workspace_floats(a, chunk);
            Act ws(pool(), (int64_t)((need + 1023) / 1024), 1024);
            wino_conv3x3(a, ws.p, chunk, s);
            return out;
        }
        E2V_REQUIRE(gn_P == 0, E2V_EINVAL, "conv: fused GroupNorm needs the Winograd path");
        IgemmArgs g;
        g.a0 = x0; g.c0 = c0; g.lda0 = c0; g.a1 = x1; g.c1 = c1; g.lda1 = c1;
        E2V_REQUIRE(c1 == 0 || c0 % 32 == 0, E2V_ESHAPE, "conv: the concat seam must be a multiple of 32 channels");
        g.ldw = w.ldw; g.ldw16 = w.ldw16; g.out = out.p; g.ldc = w.cout; g.bias = w.b;
        g.rowbias = rowbias; g.rb_ld = rb_ld; g.rows_per_sample = rows_per_sample;
        g.resid = resid; g.ldr = w.cout;
        g.M = (int)out.rows; g.N = w.cout; g.taps = 9;
        if (bf()) { g.a_bf16 = h16(); g.out_f32 = out_f32 ? 1 : 0; g.resid_bf16 = resid ? 1 : 0; }
        g.Ho = Ho; g.Wo = Wo; g.Hi = Hi; g.Wi = Wi; g.Hs = geo.H; g.Ws = geo.W; g.stride = stride; g.pad = pad;
        if (Hi != geo.H || Wi != geo.W) {
            g.upsample = 1;
            g.ups_h = (float)geo.H / (float)Hi;        // torch: scale = (float)input_size / output_size
            g.ups_w = (float)geo.W / (float)Wi;
        }
        // (small-batch family: the four parity launches of a small layer are a handful of 256-row tiles each -- 131 us per launch
        // measured at B = 1 on the 9x16 -> 18x32 conv -- while the gather form takes split-K: the sub-pixel form needs >= 128 tiles)
        const bool up2x_small = c->small_family && (long)((out.rows / 4 + 255) / 256) * ((w.cout + 319) / 320) < 128;
        if (bf() && c0 == w.cin && !up2x_small && bgemm_up2x_applies(g)) {       // exact 2x resize (Upsample3D): four 2x2 convs on the source map
            c->conv_form(w, fp16() ? e2v_ctx::FORM_F16_UP2 : e2v_ctx::FORM_BF16_UP2, s);
            bgemm_up2x_launch(g, fp16() ? w.w16h_up2 : w.w16_up2, s);
            return out;
        }
        c->conv_form(w, fp16() ? e2v_ctx::FORM_F16 : bf() ? e2v_ctx::FORM_BF16 : e2v_ctx::FORM_DIRECT32, s);
        g.w = w.w; g.w16 = fp16() ? w.w16h : w.w16;
        // GroupNorm statistics from the producer -- BUILT, BIT-EXACT, MEASURED AND NOT ADOPTED (`make ab` builds, E2V_GN_RB = 1; DESIGN
        // section 9, profiles/r04_shape_ab_gn_producer_sums.log): the sums are ALWAYS the canonical ones -- from the epilogue when the
        // launch runs on the 256-row staged kernel (E2V_GN_RB_EPILOGUE = 0: never, for the bit-identity test), else from rowblock_sums
        // over the stored tensor -- so a result never depends on which kernel (i.e. which batch size) served the layer.  At B = 32 the
        // GroupNorms save 4.0 ms per step + decode and the convs that take the sums lose 6.2 (+4..8 % each: the 256-row kernel is one
        // workgroup per CU, nothing hides a longer tile tail, while the statistics pass it replaces streams at 5 TB/s).
        static const int* const rb_on = E2V_AB_KNOB("E2V_GN_RB", 0);
        static const int* const rb_epi = E2V_AB_KNOB("E2V_GN_RB_EPILOGUE", 1);
        bool epilogue_sums = false;
        if (*rb_on && want_rb && h16() == H16_BF16 && !out_f32 && rbsum_capable(out.rows, w.cout)) {
            out.rb = pool().get((size_t)(out.rows / 64) * w.cout * 2);
            if (*rb_epi) { g.rbsum = out.rb; epilogue_sums = igemm_writes_rbsum(g); if (!epilogue_sums) g.rbsum = nullptr; }
        }
        Act skws = sk_setup(g);
        igemm(g, s);
        if (epilogue_sums) dry_tag(" +rbsum");
        if (out.rb && !epilogue_sums) rowblock_sums(out.p, w.cout, w.cout, out.rows, rbsum_rows_per_pass(w.cout), out.rb, s);
        return out;
    }

    // ResnetBlock3D.forward (resnet.py:174-204); x1 = skip tensor concatenated on channels (may be null).
    // samples x P rows share GroupNorm statistics and one time-embedding row; geo describes the frames.
    // rb0 / rb1: row-block sums of x0 / x1 (Act::rb of the tensors; null: none).  The output carries its own when P allows.
    Act resnet(const ResW& w, const float* x0, int c0, const float* x1, int c1, int samples, int P, Geo geo, int groups,
               float eps, const float* temb_silu, int temb_dim, const float* rb0 = nullptr, const float* rb1 = nullptr) {
        const bool rbw = bf() && P % 64 == 0;            // the GroupNorms that read conv1's / this block's output can use the sums
        E2V_REQUIRE(c0 + c1 == w.cin, E2V_ESHAPE, "resnet: channel mismatch");
        Act tp;
        const float* tpp = nullptr;
        int tp_ld = w.cout;
        if (w.temb.w) {
            if (c->step_cache_on) {      // e2v_generate: time_emb_proj(SiLU(emb(t))) of every step was computed up front; one row
                tpp = c->temb_cache[res_idx++].p + (size_t)c->step_cache_step * w.cout;    // serves all samples (stride 0)
                tp_ld = 0;
            } else {
                E2V_REQUIRE(temb_silu != nullptr, E2V_EINVAL, "resnet needs a time embedding");
                tp = linear(w.temb, temb_silu, temb_dim, samples, nullptr, 0, false, nullptr, 0, 0, true);   // :183 (fp32 rows)
                tpp = tp.p;
            }
        }
        Act h1;
        if (winograd(w.c1, 1, 1, geo.H, geo.W, geo.H, geo.W)) {        // GroupNorm + SiLU applied inside the conv's input transform
            gn_stats(w.n1, x0, c0, x1, c1, samples, P, groups, eps);                                  // :177-178
            h1 = conv3(w.c1, x0, c0, x1, c1, geo, geo.H, geo.W, geo.H, geo.W, 1, 1, tpp, P, nullptr, P, tp_ld);   // :180,186
        } else {
            Act hn = gn(w.n1, x0, c0, x1, c1, samples, P, groups, eps, true, rb0, rb1);              // :177-178
            h1 = conv3(w.c1, hn.p, w.cin, nullptr, 0, geo, geo.H, geo.W, geo.H, geo.W, 1, 1,          // :180,186
                       tpp, P, nullptr, 0, tp_ld, false, rbw);
        }
        Act sc;
        const float* resid = x0;
        if (w.sc.w) {                                                                                 // :199-200
            sc = linear(w.sc, x0, c0, (int64_t)samples * P, nullptr, 0, false, x1, c1, c1);
            resid = sc.p;
        } else {
            E2V_REQUIRE(c1 == 0, E2V_ESHAPE, "identity shortcut with a concatenated input");
        }
        if (winograd(w.c2, 1, 1, geo.H, geo.W, geo.H, geo.W)) {
            gn_stats(w.n2, h1.p, w.cout, nullptr, 0, samples, P, groups, eps);                        // :188,194
            return conv3(w.c2, h1.p, w.cout, nullptr, 0, geo, geo.H, geo.W, geo.H, geo.W, 1, 1, nullptr, 1, resid, P);   // :197,202
        }
        Act h2 = gn(w.n2, h1.p, w.cout, nullptr, 0, samples, P, groups, eps, true, h1.rb);           // :188,194
        h1.reset();
        return conv3(w.c2, h2.p, w.cout, nullptr, 0, geo, geo.H, geo.W, geo.H, geo.W, 1, 1, nullptr, 1, resid, 0, -1, false, rbw);   // :197,202
    }

    // Transformer3DModel.forward + BasicTransformerBlock.forward (attention.py:89-136, 232-269)
    Act twice(const Act& a) {                        // [a ; a] along the rows
        Act d(pool(), 2 * a.rows, a.C, a.bf16);
        const size_t bytes = a.bytes();
        E2V_HIP(hipMemcpyAsync(d.p, a.p, bytes, hipMemcpyDeviceToDevice, s));
        E2V_HIP(hipMemcpyAsync(reinterpret_cast<char*>(d.p) + bytes, a.p, bytes, hipMemcpyDeviceToDevice, s));
        return d;
    }

    // n_shared > 0: `x` holds only n_shared = n / 2 samples whose two copies (the uncond / cond halves of a classifier-free
    // guidance batch, pipeline_tuneeeg2video.py:313) would be bit-identical up to the first use of the conditioning: the
    // per-frame GroupNorm, proj_in and the whole sparse-causal self-attention run once, then the tokens are duplicated.
    // `cond` is in the activation type (bf16 mode: already converted, see unet_forward_cl).
    Act transformer(const TransW& w, const Act& x_in, int n, int F, int HW, const float* cond, int T, int heads, int groups,
                    int n_shared = 0) {
        const int C = w.C, D = C / heads;
        const int64_t rows = (int64_t)n * F * HW;
        const int n1 = n_shared > 0 ? n_shared : n;                       // samples up to and including attn1
        const int64_t rows1 = (int64_t)n1 * F * HW;
        E2V_REQUIRE(n_shared == 0 || 2 * n_shared == n, E2V_EINVAL, "shared prefix needs n == 2 * n_shared");
        const Act* xp = &x_in;
        const float scale = 1.0f / std::sqrt((float)D);
        E2V_REQUIRE(C % heads == 0 && D % 8 == 0, E2V_EINVAL, "attention head dim must be a multiple of 8");
        const int io16 = h16();
        Act t;
        {
            Act hn = gn(w.norm, xp->p, C, nullptr, 0, n1 * F, HW, groups, 1e-6f, false, xp->rb);      // :99 (per frame)
            t = linear(w.proj_in, hn.p, C, rows1);                                                    // :101-103
        }
        {   // attn1: sparse-causal self-attention                                                      :234-243
            Act nrm = ln(w.ln1, t);
            Act qkv = linear(w.a1_qkv, nrm.p, C, rows1);
            nrm.reset();
            Act ao = new_act(rows1, C);
            AttnArgs a;
            a.q = qkv.p; a.ldq = 3 * C; a.k = qkv.at(C); a.v = qkv.at(2 * C); a.ldkv = 3 * C; a.o = ao.p; a.ldo = C;
            a.n = n1; a.F = F; a.heads = heads; a.D = D; a.Nq = HW; a.Nk = HW; a.mode = 0; a.scale = scale;
            a.io_bf16 = io16;
            a.x3 = c->x3_compute ? 1 : 0;
            flash_attention(a, s);
            qkv.reset();
            t = linear(w.a1_out, ao.p, C, rows1, t.p, C);
        }
        Act x2;
        if (n_shared > 0) {                          // from here on the two halves see different conditioning
            t = twice(t);
            x2 = twice(*xp);
            xp = &x2;
        }
        {   // attn2: cross-attention to the cond tokens (identical for the F frames of a sample)         :245-255
            Act nrm = ln(w.ln2, t);
            Act q = linear(w.a2_q, nrm.p, C, rows);
            nrm.reset();
            Act kv_own;                        // e2v_generate computes to_k / to_v of the conditioning once for all steps
            const bool cached = c->step_cache_on;
            if (!cached) kv_own = linear(w.a2_kv, cond, w.a2_kv.in, (int64_t)n * T);
            const Act& kv = cached ? c->kv_cache[tr_idx] : kv_own;
            ++tr_idx;
            Act ao = new_act(rows, C);
            AttnArgs a;
            a.q = q.p; a.ldq = C; a.k = kv.p; a.v = kv.at(C); a.ldkv = 2 * C; a.o = ao.p; a.ldo = C;
            a.n = n; a.F = F; a.heads = heads; a.D = D; a.Nq = HW; a.Nk = T; a.mode = 1; a.scale = scale;
            a.io_bf16 = io16;
            a.x3 = c->x3_compute ? 1 : 0;
            flash_attention(a, s);
            t = linear(w.a2_out, ao.p, C, rows, t.p, C);
        }
        {   // GEGLU feed-forward                                                                          :258
            Act nrm = ln(w.ln3, t);
            Act hgate = linear(w.ff1, nrm.p, C, rows, nullptr, 0, true);
            nrm.reset();
            t = linear(w.ff2, hgate.p, 4 * C, rows, t.p, C);
        }
        {   // temporal attention over the F frames of each pixel                                         :261-267
            E2V_REQUIRE(F <= 8, E2V_EINVAL, "temporal attention supports at most 8 frames");
            Act nrm = ln(w.lnt, t);
            Act qkv = linear(w.at_qkv, nrm.p, C, rows);
            nrm.reset();
            Act ao = new_act(rows, C);
            temporal_attention(qkv.p, 3 * C, ao.p, C, n, F, HW, heads, D, scale, s, io16);
            qkv.reset();
            t = linear(w.at_out, ao.p, C, rows, t.p, C);
        }
        return linear(w.proj_out, t.p, C, rows, xp->p, C);                                            // :123,130
    }

    // AttentionBlock of the VAE mid block (diffusers 0.11.1): one head over H*W tokens, per image.  The scores and the softmax
    // are fp32 in both modes; in bf16-activation mode the probabilities are rounded once, on their way into the PV product.
    Act vae_attention(const VAEAttnW& w, const Act& x, int nimg, int HW, int groups, float eps) {
        const int C = w.C;
        E2V_REQUIRE(HW % (bf() ? 8 : 4) == 0, E2V_EINVAL, "VAE attention needs H*W to be a multiple of 4 (bf16: 8)");
        const int64_t rows = (int64_t)nimg * HW;
        Act qkv;
        {
            Act hn = gn(w.norm, x.p, C, nullptr, 0, nimg, HW, groups, eps, false);
            qkv = linear(w.qkv, hn.p, C, rows);
        }
        Act sc(pool(), rows, HW);                                   // fp32 scores
        {
            IgemmArgs g;
            g.a0 = qkv.p; g.c0 = C; g.lda0 = 3 * C; g.w = qkv.at(C); g.ldw = 3 * C;
            g.out = sc.p; g.ldc = HW; g.M = HW; g.N = HW; g.taps = 1; g.alpha = 1.0f / std::sqrt((float)C);
            g.batch = nimg; g.sa0 = (long long)HW * 3 * C; g.sw = (long long)HW * 3 * C; g.sout = (long long)HW * HW;
            if (bf()) { g.a_bf16 = h16(); g.out_f32 = 1; g.w16 = qkv.at(C); g.ldw16 = 3 * C; }
            igemm(g, s);
        }
        Act p16;
        if (bf()) p16 = Act(pool(), rows, HW, true);
        softmax_rows(sc.p, HW, (int)rows, HW, s, bf() ? p16.p : nullptr, bf() ? h16() : H16_BF16);
        Act vt = new_act((int64_t)nimg * C, HW);
        transpose2d(qkv.at(2 * C), 3 * C, vt.p, HW, HW, C, nimg, (long long)HW * 3 * C, (long long)C * HW, s, bf() ? 1 : 0);
        qkv.reset();
        Act o = new_act(rows, C);
        {
            IgemmArgs g;
            g.a0 = bf() ? p16.p : sc.p; g.c0 = HW; g.lda0 = HW; g.w = vt.p; g.ldw = HW; g.out = o.p; g.ldc = C;
            g.M = HW; g.N = C; g.taps = 1; g.batch = nimg;
            g.sa0 = (long long)HW * HW; g.sw = (long long)C * HW; g.sout = (long long)HW * C;
            if (bf()) { g.a_bf16 = h16(); g.w16 = vt.p; g.ldw16 = HW; }
            igemm(g, s);
        }
        return linear(w.proj, o.p, C, rows, x.p, C);
    }
};

int down_size(int x) { return (x - 1) / 2 + 1; }     // 3x3, stride 2, padding 1

}  // namespace

// -----------------------------------------------------------------------------------------------------
// UNet3DConditionModel.forward (unet.py:278-413), channel-last in and out
// -----------------------------------------------------------------------------------------------------
Act e2v_ctx::unet_forward_cl(const float* sample_cl, const int64_t* host_t, int n_t, const float* cond, int N, int F,
                             int H, int W, int T, hipStream_t s, bool cfg_pair, const float* host_tf) {
    E2V_REQUIRE(unet_ready, E2V_ESTATE, "UNet weights are not finalized");
    E2V_REQUIRE(n_t == 1 || n_t == N, E2V_EINVAL, "timesteps must have 1 or N entries");
    // cfg_pair: `sample_cl` holds N / 2 samples standing for [x ; x] (one timestep): everything before the first use of the
    // conditioning -- conv_in, the first resnet, the first transformer block up to its cross-attention -- is computed once
    E2V_REQUIRE(!cfg_pair || (N % 2 == 0 && n_t == 1 && !unet.down[0].attn.empty()), E2V_EINVAL, "cfg_pair needs an even batch, one timestep and a first block with attention");
    const int N1 = cfg_pair ? N / 2 : N;
    static const int* const fam_clips = knob("E2V_SMALL_FAMILY_CLIPS", 4);      // (where the boundary lies was an A/B; 0: the large family at every batch -- the parity tests hold BOTH families to the oracle with one clip)
    small_family = N <= 2 * *fam_clips;                       // B <= 4 clips with their guidance pairs (model.h)
    Runner R{this, s};
    const int groups = cfg.norm_num_groups;
    // heads of the blocks of resolution level l (unet.py:110-111,131,151,165,194: down block l, the mid block = level 3, up block 3 - l)
    auto heads_at = [&](int l) { return cfg.attention_heads_per_block[l] > 0 ? cfg.attention_heads_per_block[l] : cfg.attention_heads; };
    const float eps = cfg.norm_eps;
    const int boc0 = cfg.block_out_channels[0], temb_dim = boc0 * 4;

    // test aid (e2v_op_unet_forward_taps): activation [n * F * h * w][C] -> fp32 NCFHW at the sink's cursor
    auto tap = [&](const Act& a, int n, int C, int f, int h, int w) {
        TapSink& t = *tap_sink;
        const int64_t fhw = (int64_t)f * h * w, count = (int64_t)n * C * fhw;
        E2V_REQUIRE(t.count < 16 && t.used + count <= t.cap, E2V_EINVAL, "tap buffer too small");
        const float* rows = a.p;
        Act wide;
        if (a.bf16) {
            wide = Act(pool, a.rows, a.C);
            cvt_rows(a.p, a.C, h16_mode, wide.p, a.C, 0, a.rows, a.C, a.C, s);
            rows = wide.p;
        }
        cl_to_ncfhw(rows, a.C, t.buf + t.used, n, C, (int)fhw, 1.0f, 0.0f, 0, 0.f, 0.f, s);
        const int64_t shp[5] = {n, C, f, h, w};
        for (int i = 0; i < 5; ++i) t.shapes[t.count][i] = shp[i];
        t.used += count; ++t.count;
    };
    // time embedding (unet.py:324-345): sinusoid -> linear_1 -> SiLU -> linear_2; resnets consume SiLU(emb)
    if (d_timesteps_cap < N) {
        E2V_HIP(hipStreamSynchronize(s));
        if (d_timesteps) (void)hipFree(d_timesteps);
        E2V_HIP(hipMalloc((void**)&d_timesteps, sizeof(long long) * N));
        d_timesteps_cap = N;
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "int64");
    if (host_tf) E2V_HIP(hipMemcpyAsync(d_timesteps, host_tf, sizeof(float) * n_t, hipMemcpyHostToDevice, s));
    else E2V_HIP(hipMemcpyAsync(d_timesteps, host_t, sizeof(int64_t) * n_t, hipMemcpyHostToDevice, s));
    Act temb_silu;
    if (!step_cache_on) {
        Act sin(pool, N, boc0);
        timestep_sinusoid(d_timesteps, n_t, sin.p, N, boc0, cfg.flip_sin_to_cos, cfg.freq_shift, s, host_tf ? 1 : 0);
        Act e1 = R.linear(unet.te1, sin.p, boc0, N, nullptr, 0, false, nullptr, 0, 0, true);        // fp32 rows in both modes
        silu(e1.p, e1.p, (long long)N * temb_dim, s);
        Act emb = R.linear(unet.te2, e1.p, temb_dim, N, nullptr, 0, false, nullptr, 0, 0, true);
        if (tap_sink) tap(emb, N, temb_dim, 1, 1, 1);                                              // taps["emb"] (unet.py:345)
        silu(emb.p, emb.p, (long long)N * temb_dim, s);
        temb_silu = std::move(emb);
    }
    // bf16-activation mode: the two fp32 boundary tensors are converted once -- the latents (zero-padded to 8 channels, the
    // 16-byte granule of the bf16 kernels) and, when no per-run cache holds to_k / to_v already, the conditioning
    Act sample16, cond16;
    const int cin_act = R.bf() ? unet.conv_in.cin_pad16 : unet.conv_in.cin_pad;
    if (R.bf()) {
        sample16 = R.to_act16(sample_cl, unet.conv_in.cin_pad, cin_act, (int64_t)N1 * F * H * W);
        sample_cl = sample16.p;
        if (!step_cache_on) {
            cond16 = R.to_act16(cond, cfg.cross_attention_dim, cfg.cross_attention_dim, (int64_t)N * T);
            cond = cond16.p;
        }
    }

    int hs[4], ws[4];
    hs[0] = H; ws[0] = W;
    for (int i = 1; i < 4; ++i) { hs[i] = down_size(hs[i - 1]); ws[i] = down_size(ws[i - 1]); }

    struct Skip { Act a; int lvl; };
    std::vector<Skip> skips;
    skips.reserve(16);
    auto P_of = [&](int l) { return F * hs[l] * ws[l]; };
    auto geo_of = [&](int l) { return Geo{N * F, hs[l], ws[l]}; };

    Act x = R.conv3(unet.conv_in, sample_cl, cin_act, nullptr, 0, Geo{N1 * F, hs[0], ws[0]}, H, W, H, W, 1, 1);   // :358
    Act x_shared;                                     // cfg_pair: the N / 2-sample conv_in output feeding the first resnet
    if (cfg_pair) { x_shared = std::move(x); x = R.twice(x_shared); }
    // The skip tensors alias the running activation in the reference; here the running tensor is moved
    // into the skip list and read from there (no copy) -- `x` then points at the list's last entry.
    auto keep = [&](Act&& a, int lvl) -> const Act& {
        skips.push_back(Skip{std::move(a), lvl});
        return skips.back().a;
    };
    const Act* cur = &keep(std::move(x), 0);                                                         // :361
    for (int i = 0; i < 4; ++i) {                                                                    // :362-373
        const UNetW::Block& b = unet.down[i];
        for (size_t j = 0; j < b.res.size(); ++j) {
            Act h;
            if (cfg_pair && i == 0 && j == 0) {      // shared prefix: N / 2 samples through the resnet and up to attn2
                h = R.resnet(b.res[j], x_shared.p, x_shared.C, nullptr, 0, N1, P_of(i), Geo{N1 * F, hs[0], ws[0]}, groups, eps,
                             temb_silu.p, temb_dim);
                x_shared.reset();
                h = R.transformer(b.attn[j], h, N, F, hs[i] * ws[i], cond, T, heads_at(i), groups, N1);
            } else {
                h = R.resnet(b.res[j], cur->p, cur->C, nullptr, 0, N, P_of(i), geo_of(i), groups, eps, temb_silu.p, temb_dim, cur->rb);
                if (!b.attn.empty()) h = R.transformer(b.attn[j], h, N, F, hs[i] * ws[i], cond, T, heads_at(i), groups);
            }
            cur = &keep(std::move(h), i);
        }
        if (b.resample) {                                                                            // resnet.py:99-107
            Act d = R.conv3(b.rs, cur->p, cur->C, nullptr, 0, geo_of(i), hs[i], ws[i], hs[i + 1], ws[i + 1], 2, 1, nullptr, 1, nullptr, 0, -1, false,
                            R.bf() && P_of(i + 1) % 64 == 0);          // (the next level's first GroupNorm and, as a skip, a later one read it)
            cur = &keep(std::move(d), i + 1);
        }
        if (tap_sink) { const int l = b.resample ? i + 1 : i; tap(*cur, N, cfg.block_out_channels[i], F, hs[l], ws[l]); }       // taps["down{i}"]
    }
    // mid (unet_blocks.py:199-205)
    Act h = R.resnet(unet.mid_r0, cur->p, cur->C, nullptr, 0, N, P_of(3), geo_of(3), groups, eps, temb_silu.p, temb_dim, cur->rb);
    h = R.transformer(unet.mid_attn, h, N, F, hs[3] * ws[3], cond, T, heads_at(3), groups);
    h = R.resnet(unet.mid_r1, h.p, h.C, nullptr, 0, N, P_of(3), geo_of(3), groups, eps, temb_silu.p, temb_dim, h.rb);
    if (tap_sink) tap(h, N, cfg.block_out_channels[3], F, hs[3], ws[3]);                               // taps["mid"]
    // up (unet.py:381-404)
    for (int i = 0; i < 4; ++i) {
        const UNetW::Block& b = unet.up[i];
        const int lvl = 3 - i;
        for (size_t j = 0; j < b.res.size(); ++j) {
            E2V_REQUIRE(!skips.empty() && skips.back().lvl == lvl, E2V_ESTATE, "skip bookkeeping out of step");
            Skip sk = std::move(skips.back());                                                       // unet_blocks.py:485-487
            skips.pop_back();
            Act o = R.resnet(b.res[j], h.p, h.C, sk.a.p, sk.a.C, N, P_of(lvl), geo_of(lvl), groups, eps, temb_silu.p, temb_dim, h.rb, sk.a.rb);
            if (!b.attn.empty()) o = R.transformer(b.attn[j], o, N, F, hs[lvl] * ws[lvl], cond, T, heads_at(lvl), groups);
            h = std::move(o);
        }
        if (b.resample) {   // Upsample3D: nearest to the next skip's (f,h,w) then 3x3 conv (resnet.py:58-69, unet.py:389-390)
            h = R.conv3(b.rs, h.p, h.C, nullptr, 0, geo_of(lvl), hs[lvl - 1], ws[lvl - 1], hs[lvl - 1], ws[lvl - 1], 1, 1);
        }
        if (tap_sink) { const int l = b.resample ? lvl - 1 : lvl; tap(h, N, cfg.block_out_channels[lvl], F, hs[l], ws[l]); }   // taps["up{i}"]
    }
    Act hn = R.gn(unet.norm_out, h.p, h.C, nullptr, 0, N, P_of(0), groups, eps, true, h.rb);           // :406-407
    h.reset();
    Act out = R.conv3(unet.conv_out, hn.p, hn.C, nullptr, 0, geo_of(0), H, W, H, W, 1, 1, nullptr, 1, nullptr, 0, -1, true);   // :408 (eps: fp32)
    E2V_HIP(hipGetLastError());
    return out;
}

// What a denoising loop can compute once instead of once per step ("precompute all 50 x 22 vectors once per
// run"): the time-embedding MLP and every resnet's time_emb_proj for all timesteps (one GEMM of `steps` rows per resnet
// instead of `steps` GEMMs of N identical rows), and to_k / to_v of the conditioning of every cross-attention.  Row-wise
// the arithmetic is the uncached one, so results are bit-identical.
void e2v_ctx::build_step_caches(const int64_t* ts, int steps, const float* cond, int N, int T, hipStream_t s) {
    Runner R{this, s};
    temb_cache.clear();
    kv_cache.clear();
    const int boc0 = cfg.block_out_channels[0], temb_dim = boc0 * 4;
    if (d_timesteps_cap < steps) {
        E2V_HIP(hipStreamSynchronize(s));
        if (d_timesteps) (void)hipFree(d_timesteps);
        E2V_HIP(hipMalloc((void**)&d_timesteps, sizeof(long long) * steps));
        d_timesteps_cap = steps;
    }
    E2V_HIP(hipMemcpyAsync(d_timesteps, ts, sizeof(int64_t) * steps, hipMemcpyHostToDevice, s));
    Act sin(pool, steps, boc0);
    timestep_sinusoid(d_timesteps, steps, sin.p, steps, boc0, cfg.flip_sin_to_cos, cfg.freq_shift, s);
    Act e1 = R.linear(unet.te1, sin.p, boc0, steps, nullptr, 0, false, nullptr, 0, 0, true);
    silu(e1.p, e1.p, (long long)steps * temb_dim, s);
    Act emb = R.linear(unet.te2, e1.p, temb_dim, steps, nullptr, 0, false, nullptr, 0, 0, true);
    silu(emb.p, emb.p, (long long)steps * temb_dim, s);
    Act cond16;
    if (R.bf()) {
        cond16 = R.to_act16(cond, cfg.cross_attention_dim, cfg.cross_attention_dim, (int64_t)N * T);
        cond = cond16.p;
    }
    auto add_res = [&](const ResW& r) {
        if (r.temb.w) temb_cache.push_back(R.linear(r.temb, emb.p, temb_dim, steps, nullptr, 0, false, nullptr, 0, 0, true));
    };
    auto add_tr = [&](const TransW& w) { kv_cache.push_back(R.linear(w.a2_kv, cond, w.a2_kv.in, (int64_t)N * T)); };
    for (const auto& b : unet.down) { for (const auto& r : b.res) add_res(r); for (const auto& a : b.attn) add_tr(a); }
    add_res(unet.mid_r0); add_tr(unet.mid_attn); add_res(unet.mid_r1);
    for (const auto& b : unet.up) { for (const auto& r : b.res) add_res(r); for (const auto& a : b.attn) add_tr(a); }
}

// -----------------------------------------------------------------------------------------------------
// AutoencoderKL (diffusers 0.11.1).  Images are independent: "samples" = frames, F = 1.
// -----------------------------------------------------------------------------------------------------
void e2v_ctx::vae_decode_frames(const float* z_cl, int nf, int h, int w, float* out_cl, hipStream_t s, bool small) {
    E2V_REQUIRE(vae_ready, E2V_ESTATE, "VAE weights are not finalized");
    small_family = small;                                     // the CALL decodes at most four clips (the caller knows: a pass may be a call's ragged last group)
    Runner R{this, s};
    const int g = cfg.vae_norm_num_groups;
    const float eps = cfg.vae_norm_eps;
    const int lat = cfg.vae_latent_channels;
    int H = h, W = w;
    Act z16;
    int zc = lat;
    if (R.bf()) {                                // bf16-activation mode: z zero-padded to the 8-channel granule
        zc = vae.post_quant.in16;
        z16 = R.to_act16(z_cl, lat, zc, (int64_t)nf * H * W);
        z_cl = z16.p;
    }
    Act x = R.linear(vae.post_quant, z_cl, zc, (int64_t)nf * H * W);
    if (R.bf() && x.C != vae.dec_in.cin_pad16) {         // 4 -> 8 channels for the 3x3 conv's 16-byte pieces
        Act xp(pool, x.rows, vae.dec_in.cin_pad16, true);
        E2V_HIP(hipMemsetAsync(xp.p, 0, xp.bytes(), s));
        E2V_HIP(hipMemcpy2DAsync(xp.p, (size_t)xp.C * 2, x.p, (size_t)x.C * 2, (size_t)x.C * 2, (size_t)x.rows, hipMemcpyDeviceToDevice, s));
        x = std::move(xp);
    }
    x = R.conv3(vae.dec_in, x.p, x.C, nullptr, 0, Geo{nf, H, W}, H, W, H, W, 1, 1);
    x = R.resnet(vae.dec_mid0, x.p, x.C, nullptr, 0, nf, H * W, Geo{nf, H, W}, g, eps, nullptr, 0);
    x = R.vae_attention(vae.dec_attn, x, nf, H * W, g, eps);
    x = R.resnet(vae.dec_mid1, x.p, x.C, nullptr, 0, nf, H * W, Geo{nf, H, W}, g, eps, nullptr, 0);
    for (size_t i = 0; i < vae.dec_up.size(); ++i) {
        const VAEW::Block& b = vae.dec_up[i];
        for (const ResW& r : b.res) x = R.resnet(r, x.p, x.C, nullptr, 0, nf, H * W, Geo{nf, H, W}, g, eps, nullptr, 0);
        if (b.resample) {        // F.interpolate(scale_factor=2, nearest) + 3x3 conv
            x = R.conv3(b.rs, x.p, x.C, nullptr, 0, Geo{nf, H, W}, 2 * H, 2 * W, 2 * H, 2 * W, 1, 1);
            H *= 2; W *= 2;
        }
    }
    Act hn = R.gn(vae.dec_norm_out, x.p, x.C, nullptr, 0, nf, H * W, g, eps, true);
    x.reset();
    Act y = R.conv3(vae.dec_out, hn.p, hn.C, nullptr, 0, Geo{nf, H, W}, H, W, H, W, 1, 1, nullptr, 1, nullptr, 0, -1, true);   // frames: fp32
    E2V_HIP(hipMemcpyAsync(out_cl, y.p, (size_t)y.rows * y.C * sizeof(float), hipMemcpyDeviceToDevice, s));
    E2V_HIP(hipGetLastError());
}

void e2v_ctx::vae_encode_frames(const float* img_cl4, int n, int H0, int W0, float* moments_cl, hipStream_t s) {
    E2V_REQUIRE(vae_ready, E2V_ESTATE, "VAE weights are not finalized");
    small_family = n <= 12;
    Runner R{this, s};
    const int g = cfg.vae_norm_num_groups;
    const float eps = cfg.vae_norm_eps;
    int H = H0, W = W0;
    Act img16;
    int ic = vae.enc_in.cin_pad;
    if (R.bf()) {
        ic = vae.enc_in.cin_pad16;
        img16 = R.to_act16(img_cl4, vae.enc_in.cin_pad, ic, (int64_t)n * H * W);
        img_cl4 = img16.p;
    }
    Act x = R.conv3(vae.enc_in, img_cl4, ic, nullptr, 0, Geo{n, H, W}, H, W, H, W, 1, 1);
    for (size_t i = 0; i < vae.enc_down.size(); ++i) {
        const VAEW::Block& b = vae.enc_down[i];
        for (const ResW& r : b.res) x = R.resnet(r, x.p, x.C, nullptr, 0, n, H * W, Geo{n, H, W}, g, eps, nullptr, 0);
        if (b.resample) {        // F.pad(x, (0,1,0,1)) then 3x3 stride-2 conv without padding
            const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
            x = R.conv3(b.rs, x.p, x.C, nullptr, 0, Geo{n, H, W}, H, W, Ho, Wo, 2, 0);
            H = Ho; W = Wo;
        }
    }
    x = R.resnet(vae.enc_mid0, x.p, x.C, nullptr, 0, n, H * W, Geo{n, H, W}, g, eps, nullptr, 0);
    x = R.vae_attention(vae.enc_attn, x, n, H * W, g, eps);
    x = R.resnet(vae.enc_mid1, x.p, x.C, nullptr, 0, n, H * W, Geo{n, H, W}, g, eps, nullptr, 0);
    Act hn = R.gn(vae.enc_norm_out, x.p, x.C, nullptr, 0, n, H * W, g, eps, true);
    x.reset();
    Act y = R.conv3(vae.enc_out, hn.p, hn.C, nullptr, 0, Geo{n, H, W}, H, W, H, W, 1, 1);
    Act m = R.linear(vae.quant, y.p, y.C, y.rows, nullptr, 0, false, nullptr, 0, 0, false, true);          // moments: fp32
    E2V_HIP(hipMemcpyAsync(moments_cl, m.p, (size_t)m.rows * m.C * sizeof(float), hipMemcpyDeviceToDevice, s));
    E2V_HIP(hipGetLastError());
}
