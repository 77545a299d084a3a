// extern "C" surface of libeeg2video_hip.so (include/eeg2video_hip.h, include/eeg2video_hip_ops.h).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <mutex>

#include "../../include/eeg2video_hip_ops.h"
#include "h16.h"
#include "model.h"
#include "prof.h"

using namespace e2v;

namespace {

std::string g_create_error;

template <typename Fn>
e2v_status guarded(e2v_ctx* ctx, Fn&& fn) {
    try {
        if (ctx && !dry_run()) {                             // (a dry run -- e2v_op_describe_dispatch -- makes no HIP call)
            E2V_REQUIRE(ctx->device >= 0, E2V_ESTATE, "host-only context (device = -1): no GPU work possible");
            E2V_HIP(hipSetDevice(ctx->device));
        }
        fn();
        return E2V_OK;
    } catch (const Error& e) {
        if (ctx) ctx->err = e.what(); else g_create_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->err = e.what(); else g_create_error = e.what();
        return E2V_EINVAL;
    }
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1F, man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            exp = 127 - 15 + 1;
            while (!(man & 0x400)) { man <<= 1; --exp; }
            bits = sign | (exp << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

void make_alphas(e2v_ctx* c) {
    const int T = c->cfg.num_train_timesteps;
    c->alphas.resize(T);
    const float start = (float)std::sqrt(c->cfg.beta_start), end = (float)std::sqrt(c->cfg.beta_end);
    const float step = (end - start) / (float)(T - 1);
    float prod = 1.0f;
    for (int i = 0; i < T; ++i) {
        const float b = (i < T / 2) ? start + step * (float)i : end - step * (float)(T - i - 1);
        prod *= 1.0f - b * b;
        c->alphas[i] = prod;
    }
}

// Every entry point that enqueues work names its stream through here: the ctx's workspace cache hands a freed block to
// the next launch on the assumption that launches of one ctx are stream-ordered, so when a call arrives on a DIFFERENT
// stream than the previous one, the new stream is first made to wait for everything the old one was given.
hipStream_t S(e2v_ctx* c, e2v_stream s) {
    hipStream_t hs = static_cast<hipStream_t>(s);
    c->enter_stream(hs);
    return hs;
}

}  // namespace

extern "C" {

void e2v_default_config(e2v_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->in_channels = 4; c->out_channels = 4;
    const int boc[4] = {320, 640, 1280, 1280};
    std::memcpy(c->block_out_channels, boc, sizeof(boc));
    c->layers_per_block = 2; c->cross_attention_dim = 768; c->attention_heads = 8;
    c->norm_num_groups = 32; c->norm_eps = 1e-5f; c->flip_sin_to_cos = 1; c->freq_shift = 0.f;
    c->vae_in_channels = 3; c->vae_latent_channels = 4;
    const int vb[4] = {128, 256, 512, 512};
    std::memcpy(c->vae_block_out_channels, vb, sizeof(vb));
    c->vae_layers_per_block = 2; c->vae_norm_num_groups = 32; c->vae_norm_eps = 1e-6f; c->vae_scaling_factor = 0.18215;
    c->num_train_timesteps = 1000; c->beta_start = 0.00085; c->beta_end = 0.012; c->steps_offset = 1;
    c->sem_in_features = 310; c->sem_hidden = 10000; c->sem_tokens = 77;
}

int64_t e2v_config_size(void) { return (int64_t)sizeof(e2v_config); }

const char* e2v_version(void) { return "eeg2video_hip 0.1 (gfx950, fp32 MFMA)"; }

e2v_status e2v_create(const e2v_config* cfg, int device, e2v_ctx** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return E2V_EINVAL; }
    *out = nullptr;
    e2v_ctx* c = nullptr;
    e2v_status st = guarded(nullptr, [&] {
        if (device != -1) {      // device = -1: host-only context (key scheme + DDIM schedule), no HIP call at all
            int ndev = 0;
            E2V_HIP(hipGetDeviceCount(&ndev));
            E2V_REQUIRE(device >= 0 && device < ndev, E2V_EINVAL, "no such HIP device");
            E2V_HIP(hipSetDevice(device));
        }
        E2V_REQUIRE(cfg->attention_heads > 0 && cfg->norm_num_groups > 0 && cfg->layers_per_block > 0, E2V_EINVAL, "bad config");
        for (int i = 0; i < 4; ++i) {
            const int heads = cfg->attention_heads_per_block[i] > 0 ? cfg->attention_heads_per_block[i] : cfg->attention_heads;
            E2V_REQUIRE(cfg->attention_heads_per_block[i] >= 0 && cfg->block_out_channels[i] % cfg->norm_num_groups == 0 &&
                            cfg->block_out_channels[i] % 32 == 0 && cfg->block_out_channels[i] % heads == 0 &&
                            (cfg->block_out_channels[i] / heads) % 8 == 0,
                        E2V_EINVAL, "block_out_channels must be multiples of 32, of norm_num_groups and of 8*heads");
            E2V_REQUIRE(flash_attention_supports(cfg->block_out_channels[i] / heads), E2V_EINVAL,
                        "block_out_channels / attention heads must be a head dim with a kernel instance (8, 16, 32, 40, 64, 80, 160)");
            E2V_REQUIRE(cfg->vae_block_out_channels[i] % cfg->vae_norm_num_groups == 0 && cfg->vae_block_out_channels[i] % 4 == 0,
                        E2V_EINVAL, "vae_block_out_channels must be multiples of 4 and of vae_norm_num_groups");
        }
        E2V_REQUIRE(cfg->cross_attention_dim % 4 == 0 && cfg->in_channels % 4 == 0, E2V_EINVAL,
                    "cross_attention_dim and in_channels must be multiples of 4");
        c = new e2v_ctx();
        c->cfg = *cfg;
        c->device = device;
        c->expected_keys();
        make_alphas(c);
        if (const char* e = std::getenv("E2V_CONV_ALGO")) c->conv_algo = std::atoi(e);
        if (const char* e = std::getenv("E2V_F32X3")) c->x3_compute = std::atoi(e) != 0;
        if (const char* e = std::getenv("E2V_WINO_MIN_C")) c->wino_min_c = std::atoi(e);
        if (const char* e = std::getenv("E2V_WINO_F4")) c->wino_f4 = std::atoi(e) != 0;
        if (const char* e = std::getenv("E2V_WINO_F4_PAD")) c->wino_f4_pad = std::atof(e);
        if (const char* e = std::getenv("E2V_WINO_F4_MIN_C")) c->wino4_min_c = std::atoi(e);
        if (const char* e = std::getenv("E2V_WINO_WS_MB")) c->wino_ws_floats = (size_t)std::atol(e) * (1u << 18);
    });
    if (st != E2V_OK) { delete c; return st; }
    *out = c;
    return E2V_OK;
}

void e2v_destroy(e2v_ctx* c) {
    if (!c) return;
    if (c->device < 0) { delete c; return; }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)e2v_comm_destroy(c);
    for (auto& kv : c->raw) if (kv.second.d) (void)hipFree(kv.second.d);
    for (float* p : c->owned) (void)hipFree(p);
    for (auto& part : c->owned_part) for (float* p : part) (void)hipFree(p);
    if (c->stream_ev) (void)hipEventDestroy(c->stream_ev);
    if (c->gn_part) (void)hipFree(c->gn_part);
    if (c->gn_scale) (void)hipFree(c->gn_scale);
    if (c->d_timesteps) (void)hipFree(c->d_timesteps);
    delete c;
}

const char* e2v_last_error(const e2v_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int64_t e2v_num_expected_keys(const e2v_ctx* c) { return c ? (int64_t)c->keys.size() : 0; }

const char* e2v_expected_key(const e2v_ctx* c, int64_t i, int64_t* shape4, int* ndim) {
    if (!c || i < 0 || i >= (int64_t)c->keys.size()) return nullptr;
    const std::string& k = c->keys[(size_t)i];
    const WTensor& t = c->raw.at(k);
    if (ndim) *ndim = (int)t.shape.size();
    if (shape4) for (size_t d = 0; d < 4; ++d) shape4[d] = d < t.shape.size() ? t.shape[d] : 1;
    return k.c_str();
}

e2v_status e2v_load_tensor(e2v_ctx* c, const char* key, const void* host, e2v_dtype dtype, const int64_t* shape, int ndim) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(key && host && shape, E2V_EINVAL, "null argument");
        auto it = c->raw.find(key);
        E2V_REQUIRE(it != c->raw.end(), E2V_ENOWEIGHT, std::string("unexpected state-dict key: ") + key);
        WTensor& t = it->second;
        bool same = (int)t.shape.size() == ndim;
        for (int d = 0; same && d < ndim; ++d) same = t.shape[d] == shape[d];
        E2V_REQUIRE(same, E2V_ENOWEIGHT, std::string("shape mismatch for ") + key);
        if (!t.d) {
            E2V_HIP(hipMalloc((void**)&t.d, t.numel * sizeof(float)));
            c->weight_bytes += t.numel * sizeof(float);
        }
        if (dtype == E2V_F32) {
            E2V_HIP(hipMemcpy(t.d, host, t.numel * sizeof(float), hipMemcpyHostToDevice));
        } else if (dtype == E2V_F16) {
            std::vector<float> tmp(t.numel);
            const uint16_t* h = static_cast<const uint16_t*>(host);
            for (size_t i = 0; i < t.numel; ++i) tmp[i] = half_to_float(h[i]);
            E2V_HIP(hipMemcpy(t.d, tmp.data(), t.numel * sizeof(float), hipMemcpyHostToDevice));
        } else {
            throw Error(E2V_EINVAL, "unsupported dtype");
        }
        t.loaded = true;
    });
}

e2v_status e2v_finalize_weights(e2v_ctx* c, int which) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(which >= 1 && which <= 7, E2V_EINVAL, "which is a bit mask: 1 UNet, 2 VAE, 4 semantic predictor");
        c->finalize(which);
    });
}

e2v_status e2v_ddim_timesteps(const e2v_ctx* c, int n, int64_t* out) {
    if (!c || !out || n <= 0 || n > c->cfg.num_train_timesteps) return E2V_EINVAL;
    const int64_t ratio = c->cfg.num_train_timesteps / n;            // step_ratio = T // n
    for (int i = 0; i < n; ++i) out[i] = (int64_t)(n - 1 - i) * ratio + c->cfg.steps_offset;
    return E2V_OK;
}

e2v_status e2v_ddim_alphas_cumprod(const e2v_ctx* c, float* out) {
    if (!c || !out) return E2V_EINVAL;
    std::memcpy(out, c->alphas.data(), c->alphas.size() * sizeof(float));
    return E2V_OK;
}

e2v_status e2v_set_alphas_cumprod(e2v_ctx* c, const float* t, int n) {
    if (!c || !t || n != c->cfg.num_train_timesteps) return E2V_EINVAL;
    c->alphas.assign(t, t + n);
    return E2V_OK;
}

e2v_status e2v_set_ddim_schedule(e2v_ctx* c, const float* t, int n, int steps_offset) {
    if (!c || !t || n <= 0 || steps_offset < 0 || steps_offset >= n) return E2V_EINVAL;
    c->cfg.num_train_timesteps = n;
    c->cfg.steps_offset = steps_offset;
    c->alphas.assign(t, t + n);
    return E2V_OK;
}

// ---- SURVEY 8(f) rows ---------------------------------------------------------------------------------------
e2v_status e2v_semantic_predict(e2v_ctx* c, const float* eeg, int B, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(c->sem_ready, E2V_ESTATE, "semantic predictor weights are not finalized");
        E2V_REQUIRE(eeg && out && B > 0, E2V_EINVAL, "bad arguments");
        hipStream_t s = S(c, stream);
        const bool b16 = c->bf16_compute;                        // bf16-activation mode: bf16 rows between the layers, fp32 result
        static const bool gemv_on = [] { const char* e = std::getenv("E2V_SEM_GEMV"); return !e || std::atoi(e) != 0; }();
        if (gemv_on && gemv_rows_supported(B, c->cfg.sem_hidden, b16) && gemv_rows_supported(B, c->sem_in_pad, b16)) {
            // the reference's batch sizes (1 .. a few EEG segments): every layer is a stream over its weight matrix
            Act x(c->pool, B, c->sem_in_pad);
            pad_cols(eeg, c->cfg.sem_in_features, x.p, c->sem_in_pad, B, s);
            for (size_t i = 0; i < c->sem.size(); ++i) {
                const LinW& w = c->sem[i];
                const bool last = i + 1 == c->sem.size();
                Act y;
                if (!last) y = Act(c->pool, B, w.out);
                const void* w16 = c->h16_mode == H16_FP16 ? c->lin_f16(w, s) : w.w16;
                gemv_rows(x.p, x.C, b16 ? w16 : (const void*)w.w, b16 ? w.in16 : w.in, b16 ? c->h16_mode : 0, w.b, last ? out : y.p, w.out, B, w.out,
                          b16 ? w.in16 : w.in, last ? 0 : 1, s);
                if (!last) x = std::move(y);
            }
            E2V_HIP(hipGetLastError());
            return;
        }
        Act x(c->pool, B, c->sem_in_pad, b16);
        pad_cols(eeg, c->cfg.sem_in_features, x.p, c->sem_in_pad, B, s, b16 ? c->h16_mode : 0);
        for (size_t i = 0; i < c->sem.size(); ++i) {              // Linear -> ReLU ... -> Linear (train_semantic_predictor.py:14-28)
            const LinW& w = c->sem[i];
            const bool last = i + 1 == c->sem.size();
            Act y;
            if (!last) y = Act(c->pool, B, w.out, b16);
            IgemmArgs g;
            g.a0 = x.p; g.c0 = b16 ? w.in16 : w.in; g.lda0 = g.c0; g.w = w.w; g.ldw = w.in; g.bias = w.b;
            g.out = last ? out : y.p; g.ldc = w.out; g.M = B; g.N = w.out; g.taps = 1; g.relu = last ? 0 : 1; g.ldw16 = w.in16;
            if (b16) { g.w16 = c->h16_mode == H16_FP16 ? c->lin_f16(w, s) : w.w16; g.a_bf16 = c->h16_mode; g.out_f32 = last ? 1 : 0; }
            igemm(g, s);
            if (!last) x = std::move(y);
        }
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_dana_noise(e2v_ctx* c, const float* x0, const float* ed, const float* es, const int64_t* host_t, int steps,
                          float dyn_beta, int B, int F, int C, int H, int W, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(x0 && ed && es && host_t && out, E2V_EINVAL, "null argument");
        E2V_REQUIRE(B > 0 && F > 0 && C > 0 && H > 0 && W > 0 && steps > 1, E2V_ESHAPE, "non-positive dimension");
        E2V_REQUIRE(dyn_beta >= 0.f && dyn_beta <= 1.f, E2V_EINVAL, "dynamic_beta must be in [0, 1]");
        // linear betas 1e-4 .. 0.02 (DANA_module.py:42-52), alphas_cumprod in fp32 as torch.cumprod does
        std::vector<float> ac(steps);
        {
            const float b0 = 0.0001f, b1 = 0.02f, step = (b1 - b0) / (float)(steps - 1);
            float prod = 1.f;
            for (int i = 0; i < steps; ++i) {
                const float beta = (i < steps / 2) ? b0 + step * (float)i : b1 - step * (float)(steps - 1 - i);
                prod *= 1.f - beta;
                ac[i] = prod;
            }
        }
        std::vector<float> coef(2 * (size_t)B);
        for (int b = 0; b < B; ++b) {
            E2V_REQUIRE(host_t[b] >= 0 && host_t[b] < steps, E2V_EINVAL, "timestep out of range");
            coef[2 * b] = std::sqrt(ac[(size_t)host_t[b]]);
            coef[2 * b + 1] = std::sqrt(1.f - ac[(size_t)host_t[b]]);
        }
        hipStream_t s = S(c, stream);
        Act dc(c->pool, B, 2);
        E2V_HIP(hipMemcpyAsync(dc.p, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice, s));
        dana_noise(x0, ed, es, dc.p, std::sqrt(1.f - dyn_beta), std::sqrt(dyn_beta), out, B, F, C, H * W, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_frames_to_uint8(e2v_ctx* c, const float* videos, uint8_t* out, int64_t count, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(videos && out && count >= 0, E2V_EINVAL, "bad arguments");
        frames_to_u8(videos, out, count, S(c, stream));
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_profile_begin(e2v_ctx* c) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] { E2V_HIP(hipDeviceSynchronize()); profiler().begin(); });
}

int64_t e2v_profile_end(e2v_ctx* c, char* json, int64_t cap) {
    if (!c || !json || cap <= 0) return -1;
    const std::string s = profiler().end_json();
    const int64_t n = (int64_t)s.size() < cap - 1 ? (int64_t)s.size() : cap - 1;
    std::memcpy(json, s.data(), (size_t)n);
    json[n] = 0;
    return (int64_t)s.size();
}

e2v_status e2v_set_compute_dtype(e2v_ctx* c, int dtype) {
    if (!c || (dtype != E2V_F32 && dtype != E2V_BF16 && dtype != E2V_F16 && dtype != E2V_F32X3)) return E2V_EINVAL;
    if (dtype == E2V_F32X3 && (c->unet_ready || c->vae_ready || c->sem_ready)) {
        c->err = "E2V_F32X3 needs the split weights: select it before e2v_finalize_weights";
        return E2V_ESTATE;
    }
    c->set_h16_mode(dtype == E2V_BF16 ? H16_BF16 : dtype == E2V_F16 ? H16_FP16 : H16_NONE);
    c->x3_compute = dtype == E2V_F32X3;
    return E2V_OK;
}

e2v_status e2v_set_conv_algo(e2v_ctx* c, int algo) {
    if (!c || algo < E2V_CONV_AUTO || algo > E2V_CONV_WINOGRAD4) return E2V_EINVAL;
    c->conv_algo = algo;
    return E2V_OK;
}

int64_t e2v_device_bytes(const e2v_ctx* c) { return c ? (int64_t)(c->weight_bytes + c->pool.bytes()) : 0; }

// ---------------------------------------------------------------------------------------------------
e2v_status e2v_unet_forward(e2v_ctx* c, const float* sample, const int64_t* host_t, int n_t, const float* cond, int N,
                            int F, int H, int W, int T, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(sample && host_t && cond && out, E2V_EINVAL, "null argument");
        E2V_REQUIRE(N > 0 && F > 0 && H > 0 && W > 0 && T > 0, E2V_ESHAPE, "non-positive dimension");
        hipStream_t s = S(c, stream);
        const int Cin = c->cfg.in_channels, Cout = c->cfg.out_channels;
        const int FHW = F * H * W;
        Act x(c->pool, (int64_t)N * FHW, Cin);
        ncfhw_to_cl(sample, x.p, N, Cin, Cin, FHW, 1.0f, s);
        Act y = c->unet_forward_cl(x.p, host_t, n_t, cond, N, F, H, W, T, s);
        cl_to_ncfhw(y.p, Cout, out, N, Cout, FHW, 1.0f, 0.0f, 0, 0.f, 0.f, s);
        E2V_HIP(hipGetLastError());
    });
}

// Test aid (eeg2video_hip_ops.h): the forward above with the block outputs the oracle exposes copied out on the way.
e2v_status e2v_op_unet_forward_taps(e2v_ctx* c, const float* sample, const int64_t* host_t, int n_t, const float* cond, int N,
                                    int F, int H, int W, int T, float* out, float* taps, int64_t taps_cap, int64_t* shapes,
                                    int* n_taps, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(sample && host_t && cond && out && taps && shapes && n_taps, E2V_EINVAL, "null argument");
        E2V_REQUIRE(N > 0 && F > 0 && H > 0 && W > 0 && T > 0 && taps_cap > 0, E2V_ESHAPE, "non-positive dimension");
        hipStream_t s = S(c, stream);
        const int Cin = c->cfg.in_channels, Cout = c->cfg.out_channels;
        const int FHW = F * H * W;
        e2v_ctx::TapSink sink;
        sink.buf = taps; sink.cap = taps_cap;
        struct Guard { e2v_ctx* c; ~Guard() { c->tap_sink = nullptr; } } guard{c};
        c->tap_sink = &sink;
        Act x(c->pool, (int64_t)N * FHW, Cin);
        ncfhw_to_cl(sample, x.p, N, Cin, Cin, FHW, 1.0f, s);
        Act y = c->unet_forward_cl(x.p, host_t, n_t, cond, N, F, H, W, T, s);
        cl_to_ncfhw(y.p, Cout, out, N, Cout, FHW, 1.0f, 0.0f, 0, 0.f, 0.f, s);
        E2V_HIP(hipGetLastError());
        *n_taps = sink.count;
        for (int i = 0; i < sink.count; ++i)
            for (int k = 0; k < 5; ++k) shapes[5 * i + k] = sink.shapes[i][k];
    });
}

e2v_status e2v_unet_forward_ft(e2v_ctx* c, const float* sample, const float* host_t, int n_t, const float* cond, int N,
                               int F, int H, int W, int T, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(sample && host_t && cond && out, E2V_EINVAL, "null argument");
        E2V_REQUIRE(N > 0 && F > 0 && H > 0 && W > 0 && T > 0, E2V_ESHAPE, "non-positive dimension");
        hipStream_t s = S(c, stream);
        const int Cin = c->cfg.in_channels, Cout = c->cfg.out_channels;
        const int FHW = F * H * W;
        Act x(c->pool, (int64_t)N * FHW, Cin);
        ncfhw_to_cl(sample, x.p, N, Cin, Cin, FHW, 1.0f, s);
        Act y = c->unet_forward_cl(x.p, nullptr, n_t, cond, N, F, H, W, T, s, false, host_t);
        cl_to_ncfhw(y.p, Cout, out, N, Cout, FHW, 1.0f, 0.0f, 0, 0.f, 0.f, s);
        E2V_HIP(hipGetLastError());
    });
}

static void ddim_coeffs(const e2v_ctx* c, int64_t t, int64_t t_prev, float co[4]) {
    E2V_REQUIRE(t >= 0 && t < (int64_t)c->alphas.size() && t_prev < (int64_t)c->alphas.size(), E2V_EINVAL, "timestep out of range");
    const float a_t = c->alphas[(size_t)t];
    const float a_p = t_prev >= 0 ? c->alphas[(size_t)t_prev] : c->alphas[0];   // final_alpha_cumprod (set_alpha_to_one = False)
    co[0] = std::sqrt(a_t);
    co[1] = std::sqrt(1.0f - a_t);
    co[2] = std::sqrt(a_p);
    co[3] = std::sqrt(1.0f - a_p);
}

e2v_status e2v_ddim_cfg_step(e2v_ctx* c, const float* eu, const float* ec, const float* x, float* xo, int64_t count,
                             float g, int64_t t, int64_t t_prev, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(eu && x && xo && count >= 0, E2V_EINVAL, "null argument");
        float co[4];
        ddim_coeffs(c, t, t_prev, co);
        ddim_cfg_step(eu, ec, x, xo, count, g, co[0], co[1], co[2], co[3], S(c, stream));
        E2V_HIP(hipGetLastError());
    });
}

static void decode_to_video(e2v_ctx* c, const float* z_cl, int B, int F, int h, int w, int post, float* videos, hipStream_t s) {
    const int HW8 = 64 * h * w;
    const int C3 = c->cfg.vae_in_channels;
    Act frames(c->pool, (int64_t)B * F * HW8, C3);
    // frames are independent samples, so the grouping does not change a bit of the result.  fp32: one clip (F frames) per pass keeps
    // the workspace at ~1 clip.  bf16 mode: four -- the 36x64 levels of ONE clip (M = 13 824 rows) are 432 tiles for 512 slots; 0.6 % of
    // a B = 32 pass (same-box A/B; fp32: +-0), 0.9 GB per 288x512x128 bf16 tensor.  E2V_VAE_CLIPS_PER_PASS overrides both.
    static const int forced = [] { const char* e = std::getenv("E2V_VAE_CLIPS_PER_PASS"); return e ? std::atoi(e) : 0; }();
    const int group = forced > 0 ? forced : (c->bf16_compute ? 4 : 1);
    for (int b = 0; b < B; b += group) {
        const int nb = B - b < group ? B - b : group;
        c->vae_decode_frames(z_cl + (size_t)b * F * h * w * c->cfg.vae_latent_channels, nb * F, h, w,
                             frames.p + (size_t)b * F * HW8 * C3, s, B <= *knob("E2V_SMALL_FAMILY_CLIPS", 4));      // the dispatch family is the CALL's (model.h), not the pass's
    }
    nchw_frames_to_ncfhw(frames.p, C3, videos, B, F, C3, HW8, post ? 0.5f : 1.0f, post ? 0.5f : 0.0f, post ? 1 : 0, s);
}

e2v_status e2v_vae_decode(e2v_ctx* c, const float* latents, int B, int F, int h, int w, int post, float* videos,
                          e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(latents && videos, E2V_EINVAL, "null argument");
        E2V_REQUIRE(B > 0 && F > 0 && h > 0 && w > 0, E2V_ESHAPE, "non-positive dimension");
        hipStream_t s = S(c, stream);
        const int lat = c->cfg.vae_latent_channels;
        Act z(c->pool, (int64_t)B * F * h * w, lat);
        ncfhw_to_cl(latents, z.p, B, lat, lat, F * h * w, post ? (float)(1.0 / c->cfg.vae_scaling_factor) : 1.0f, s);   // :177
        decode_to_video(c, z.p, B, F, h, w, post, videos, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_vae_encode(e2v_ctx* c, const float* images, int n, int H, int W, float* mean, float* logvar, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(images && mean && logvar, E2V_EINVAL, "null argument");
        E2V_REQUIRE(n > 0 && H % 8 == 0 && W % 8 == 0 && H > 0 && W > 0, E2V_ESHAPE, "H and W must be positive multiples of 8");
        hipStream_t s = S(c, stream);
        const int Cimg = c->cfg.vae_in_channels, Cp = c->vae.enc_in.cin_pad, lat = c->cfg.vae_latent_channels;
        Act x(c->pool, (int64_t)n * H * W, Cp);
        ncfhw_to_cl(images, x.p, n, Cimg, Cp, H * W, 1.0f, s);
        const int hw = (H / 8) * (W / 8);
        Act m(c->pool, (int64_t)n * hw, 2 * lat);
        c->vae_encode_frames(x.p, n, H, W, m.p, s);
        cl_to_ncfhw(m.p, 2 * lat, mean, n, lat, hw, 1.f, 0.f, 0, 0.f, 0.f, s);
        cl_to_ncfhw(m.p + lat, 2 * lat, logvar, n, lat, hw, 1.f, 0.f, 1, -30.f, 20.f, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_generate(e2v_ctx* c, const float* latents, const float* cond, const float* uncond, int Bu, int B, int F,
                        int h, int w, int T, int steps, float guidance, float eta, float* videos, float* latents_out,
                        e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(latents && cond, E2V_EINVAL, "null argument");
        E2V_REQUIRE(B > 0 && F > 0 && h > 0 && w > 0 && T > 0, E2V_ESHAPE, "non-positive dimension");
        E2V_REQUIRE(steps > 0 && steps <= c->cfg.num_train_timesteps, E2V_EINVAL, "num_inference_steps out of range");
        E2V_REQUIRE(eta == 0.0f, E2V_EINVAL, "only the deterministic DDIM update (eta = 0) is implemented");
        const bool cfg_on = guidance > 1.0f;                                     // pipeline_tuneeeg2video.py:281
        E2V_REQUIRE(!cfg_on || (uncond && (Bu == 1 || Bu == B)), E2V_EINVAL, "uncond must have batch 1 or B");
        hipStream_t s = S(c, stream);
        const int Cl = c->cfg.in_channels, D = c->cfg.cross_attention_dim;
        const int P = F * h * w;
        const size_t per = (size_t)P * Cl;
        const int N = cfg_on ? 2 * B : B;

        Act x(c->pool, (int64_t)B * P, Cl);
        ncfhw_to_cl(latents, x.p, B, Cl, Cl, P, 1.0f, s);                        // init_noise_sigma = 1 (:244)
        Act emb;                                                                 // [uncond ; cond] (:162-172)
        const float* embp = cond;
        if (cfg_on) {
            emb = Act(c->pool, (int64_t)N * T, D);
            const size_t one = (size_t)T * D;
            if (Bu == B) E2V_HIP(hipMemcpyAsync(emb.p, uncond, one * B * sizeof(float), hipMemcpyDeviceToDevice, s));
            else for (int b = 0; b < B; ++b)
                E2V_HIP(hipMemcpyAsync(emb.p + b * one, uncond, one * sizeof(float), hipMemcpyDeviceToDevice, s));
            E2V_HIP(hipMemcpyAsync(emb.p + B * one, cond, one * B * sizeof(float), hipMemcpyDeviceToDevice, s));
            embp = emb.p;
        }
        std::vector<int64_t> ts(steps);
        e2v_ddim_timesteps(c, steps, ts.data());                                 // :287-288
        const int64_t ratio = c->cfg.num_train_timesteps / steps;
        // everything that depends only on (t_i, conditioning) is computed once for the whole loop
        struct CacheGuard {
            e2v_ctx* c;
            ~CacheGuard() { c->step_cache_on = false; c->temb_cache.clear(); c->kv_cache.clear(); }
        } guard{c};
        c->build_step_caches(ts.data(), steps, embp, N, T, s);
        c->step_cache_on = true;
        // latent_model_input = cat([latents] * 2) (:313) is not materialised: the UNet is told that its input stands for the
        // pair, and computes what the two copies share (up to the first cross-attention) once
        const bool pair = cfg_on && !c->unet.down[0].attn.empty();
        Act xin;
        if (cfg_on && !pair) xin = Act(c->pool, (int64_t)N * P, Cl);
        for (int i = 0; i < steps; ++i) {                                        // :311
            const float* in = x.p;
            if (cfg_on && !pair) {
                E2V_HIP(hipMemcpyAsync(xin.p, x.p, per * B * sizeof(float), hipMemcpyDeviceToDevice, s));
                E2V_HIP(hipMemcpyAsync(xin.p + per * B, x.p, per * B * sizeof(float), hipMemcpyDeviceToDevice, s));
                in = xin.p;
            }
            c->step_cache_step = i;
            Act eps = c->unet_forward_cl(in, &ts[i], 1, embp, N, F, h, w, T, s, pair);  // :317
            float co[4];
            ddim_coeffs(c, ts[i], ts[i] - ratio, co);
            ddim_cfg_step(eps.p, cfg_on ? eps.p + per * B : nullptr, x.p, x.p, (long long)(per * B), guidance, co[0], co[1],
                          co[2], co[3], s);                                      // :320-325
            // profiling aid: under rocprofv3 --pmc (ROCm 7.2) rocprofiler-sdk's queue-intercept callback faulted with a whole pass
            // (~34 000 launches, ~680 per DDIM step) queued behind the GPU (profiles/r02_pmc_async_abort_README.md names the frames);
            // E2V_SYNC_EACH_STEP=1 drains the stream after every DDIM step.  Read once per process; never set in a timed run
            static const bool sync_each = [] { const char* e = std::getenv("E2V_SYNC_EACH_STEP"); return e && std::atoi(e) != 0; }();
            if (sync_each) E2V_HIP(hipStreamSynchronize(s));
        }
        if (latents_out) cl_to_ncfhw(x.p, Cl, latents_out, B, Cl, P, 1.f, 0.f, 0, 0.f, 0.f, s);
        if (videos) {                                                            // decode_latents (:175-184)
            Act z(c->pool, (int64_t)B * P, Cl);
            const float inv = (float)(1.0 / c->cfg.vae_scaling_factor);
            ncfhw_to_cl(x.p, z.p, 1, 1, 1, (int)(per * B), inv, s);      // z = 1 / 0.18215 * latents (:177), flat scale
            decode_to_video(c, z.p, B, F, h, w, 1, videos, s);
        }
        E2V_HIP(hipGetLastError());
    });
}

// ---- which kernel and tile every launch of a configuration takes, without a GPU -----------------------------------------------------
// Runs e2v_generate (one DDIM step with guidance + decode) of B clips of [4, F, h, w] latents as a DRY RUN on a host-only ctx: the
// graph walker, the launch rules of every launcher and the per-layer decisions execute exactly as in a real call, no HIP call is
// made, and every launch leaves one record "class shape -> kernel tile".  Output: one line per distinct record in first-occurrence
// order, "<count>x <record>\n".  A rule change shows up as a diff of this text (tests/golden/dispatch_sd_v1_4.json).
e2v_status e2v_op_describe_dispatch(e2v_ctx* c, int dtype, int B, int F, int h, int w, int T, char* buf, int64_t cap, int64_t* needed) {
    if (!c || !needed || (dtype != E2V_F32 && dtype != E2V_BF16 && dtype != E2V_F16) || B <= 0 || F <= 0 || h <= 0 || w <= 0 || T <= 0 || (cap > 0 && !buf)) return E2V_EINVAL;
    if (c->device >= 0) { c->err = "e2v_op_describe_dispatch runs on a host-only context (e2v_create(..., device = -1, ...))"; return E2V_ESTATE; }
    struct Dry {
        Dry() { dry_run() = true; dry_log().clear(); }
        ~Dry() { dry_run() = false; dry_log().clear(); }
    } dry;
    e2v_status st = guarded(c, [&] {
        if (!c->unet_ready || !c->vae_ready) c->finalize(3);
    });
    if (st != E2V_OK) return st;
    const int was_mode = c->h16_mode;
    c->set_h16_mode(dtype == E2V_BF16 ? H16_BF16 : dtype == E2V_F16 ? H16_FP16 : H16_NONE);
    dry_log().clear();
    const float* fake = dry_fake_ptr(1 << 20);
    st = e2v_generate(c, fake, fake, fake, 1, B, F, h, w, T, 1, 12.5f, 0.0f, dry_fake_ptr(1 << 20), nullptr, nullptr);
    c->set_h16_mode(was_mode);
    if (st != E2V_OK) return st;
    std::vector<std::pair<std::string, long>> agg;
    std::unordered_map<std::string, size_t> at;
    for (const std::string& r : dry_log()) {
        auto it = at.find(r);
        if (it == at.end()) { at.emplace(r, agg.size()); agg.emplace_back(r, 1); }
        else agg[it->second].second += 1;
    }
    std::string out;
    for (const auto& kv : agg) out += std::to_string(kv.second) + "x " + kv.first + "\n";
    *needed = (int64_t)out.size() + 1;
    if (cap >= *needed) std::memcpy(buf, out.c_str(), out.size() + 1);
    return E2V_OK;
}

// ---- building blocks of the schedulers other than DDIM (pipeline_tuneeeg2video.py:48-55) -----------
e2v_status e2v_cfg_combine(e2v_ctx* c, const float* eu, const float* ec, float g, float* out, int64_t count, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(eu && ec && out && count >= 0, E2V_EINVAL, "null argument");
        cfg_combine(eu, ec, g, out, count, S(c, stream));
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_lincomb(e2v_ctx* c, int n, const float* const* xs, const float* coefs, float* out, int64_t count, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(n >= 1 && n <= 5 && xs && coefs && out && count >= 0, E2V_EINVAL, "lincomb takes 1..5 terms");
        for (int k = 0; k < n; ++k) E2V_REQUIRE(xs[k] != nullptr, E2V_EINVAL, "null term");
        lincomb(n, xs, coefs, out, count, S(c, stream));
        E2V_HIP(hipGetLastError());
    });
}

// ---- DDIM inversion (EEG2Video_New/Generation/tuneavideo/util.py:56-101) --------------------------
// next_step: "timestep" = t - T/n (alpha of the step being left; < 0 -> final_alpha_cumprod), "next_timestep" = t.
static void ddim_next_coeffs(const e2v_ctx* c, int64_t t, int steps, float co[4]) {
    const int64_t T = c->cfg.num_train_timesteps;
    E2V_REQUIRE(steps > 0 && steps <= T, E2V_EINVAL, "num_inference_steps out of range");
    E2V_REQUIRE(t >= 0 && t < (int64_t)c->alphas.size(), E2V_EINVAL, "timestep out of range");
    const int64_t cur = std::min<int64_t>(t - T / steps, 999);                     // util.py:58-59
    const float a_t = cur >= 0 ? c->alphas[(size_t)cur] : c->alphas[0];           // :60 (set_alpha_to_one = False)
    const float a_n = c->alphas[(size_t)t];                                        // :61
    co[0] = std::sqrt(a_t);
    co[1] = std::sqrt(1.0f - a_t);
    co[2] = std::sqrt(a_n);
    co[3] = std::sqrt(1.0f - a_n);
}

e2v_status e2v_ddim_next_step(e2v_ctx* c, const float* eps, const float* x, float* xo, int64_t count, int64_t t,
                              int num_inference_steps, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(eps && x && xo && count >= 0, E2V_EINVAL, "null argument");
        float co[4];
        ddim_next_coeffs(c, t, num_inference_steps, co);
        ddim_cfg_step(eps, nullptr, x, xo, count, 1.0f, co[0], co[1], co[2], co[3], S(c, stream));   // :63-65, same form as step()
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_ddim_invert(e2v_ctx* c, const float* latents, const float* cond, int B, int F, int h, int w, int T,
                           int num_inv_steps, float* all_latents, float* final_latent, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(latents && cond && (all_latents || final_latent), E2V_EINVAL, "null argument");
        E2V_REQUIRE(B > 0 && F > 0 && h > 0 && w > 0 && T > 0, E2V_ESHAPE, "non-positive dimension");
        E2V_REQUIRE(num_inv_steps > 0 && num_inv_steps <= c->cfg.num_train_timesteps, E2V_EINVAL, "num_inv_steps out of range");
        hipStream_t s = S(c, stream);
        const int Cl = c->cfg.in_channels;
        const int P = F * h * w;
        const size_t per = (size_t)B * P * Cl;
        Act x(c->pool, (int64_t)B * P, Cl);
        ncfhw_to_cl(latents, x.p, B, Cl, Cl, P, 1.0f, s);
        if (all_latents) E2V_HIP(hipMemcpyAsync(all_latents, latents, per * sizeof(float), hipMemcpyDeviceToDevice, s));   // all_latent = [latent] (:86)
        std::vector<int64_t> ts(num_inv_steps);
        e2v_ddim_timesteps(c, num_inv_steps, ts.data());
        for (int i = 0; i < num_inv_steps; ++i) {                                 // :88
            const int64_t t = ts[num_inv_steps - 1 - i];                          // timesteps[len - i - 1] (:89): ascending
            Act eps = c->unet_forward_cl(x.p, &t, 1, cond, B, F, h, w, T, s);     // get_noise_pred_single (:68-70), no guidance
            float co[4];
            ddim_next_coeffs(c, t, num_inv_steps, co);
            ddim_cfg_step(eps.p, nullptr, x.p, x.p, (long long)per, 1.0f, co[0], co[1], co[2], co[3], s);   // next_step (:91)
            if (all_latents) cl_to_ncfhw(x.p, Cl, all_latents + (size_t)(i + 1) * per, B, Cl, P, 1.f, 0.f, 0, 0.f, 0.f, s);
        }
        if (final_latent) cl_to_ncfhw(x.p, Cl, final_latent, B, Cl, P, 1.f, 0.f, 0, 0.f, 0.f, s);
        E2V_HIP(hipGetLastError());
    });
}

// ---------------------------------------------------------------------------------------------------
// kernel-level entry points (eeg2video_hip_ops.h)
// ---------------------------------------------------------------------------------------------------
e2v_status e2v_op_conv3x3(e2v_ctx* c, const float* x0, int c0, const float* x1, int c1, int n_img, int Hs, int Ws, int Hi,
                          int Wi, int Ho, int Wo, int stride, int pad_lo, const float* w_oihw, const float* bias, int cout,
                          const float* rowbias, int rows_per_sample, const float* resid, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(x0 && w_oihw && out && c0 % 4 == 0 && c1 % 4 == 0 && c0 > 0, E2V_EINVAL, "bad conv arguments");
        hipStream_t s = S(c, stream);
        const int cin = c0 + c1;
        E2V_REQUIRE(c1 == 0 || c0 % 32 == 0, E2V_ESHAPE, "conv: the concat seam must be a multiple of 32 channels");
        if (c->bf16_compute) {       // bf16-activation mode: operands rounded to bf16 once (channels zero-padded to 8), fp32 result
            E2V_REQUIRE(c1 == 0 || c0 % 64 == 0, E2V_ESHAPE, "bf16 conv: the concat seam must be a multiple of 64 channels");
            const int c0p = c1 > 0 ? c0 : (c0 + 7) / 8 * 8, c1p = (c1 + 7) / 8 * 8;
            const int64_t rows_in = (int64_t)n_img * Hs * Ws;
            Act a0(c->pool, rows_in, c0p, true), a1;
            cvt_rows(x0, c0, 0, a0.p, c0p, c->h16_mode, rows_in, c0, c0p, s);
            if (c1 > 0) { a1 = Act(c->pool, rows_in, c1p, true); cvt_rows(x1, c1, 0, a1.p, c1p, c->h16_mode, rows_in, c1, c1p, s); }
            const int ld64 = conv3x3_packed_ld(cin, 64);
            Act w64(c->pool, cout, ld64), w16(c->pool, cout, (ld64 + 1) / 2);
            pack_conv3x3(w_oihw, w64.p, cout, cin, 64, s);
            to_h16(w64.p, w16.p, (size_t)cout * ld64, c->h16_mode, s);
            IgemmArgs g;
            g.a0 = a0.p; g.c0 = c0p; g.lda0 = c0p; g.a1 = c1 > 0 ? a1.p : nullptr; g.c1 = c1 > 0 ? c1p : 0; g.lda1 = c1p;
            g.w16 = w16.p; g.ldw16 = ld64; g.ldw = ld64; g.out = out; g.ldc = cout; g.bias = bias;
            g.rowbias = rowbias; g.rb_ld = cout; g.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
            g.resid = resid; g.ldr = cout; g.M = n_img * Ho * Wo; g.N = cout; g.taps = 9;
            g.Ho = Ho; g.Wo = Wo; g.Hi = Hi; g.Wi = Wi; g.Hs = Hs; g.Ws = Ws; g.stride = stride; g.pad = pad_lo;
            if (Hi != Hs || Wi != Ws) { g.upsample = 1; g.ups_h = (float)Hs / (float)Hi; g.ups_w = (float)Ws / (float)Wi; }
            g.a_bf16 = c->h16_mode; g.out_f32 = 1;
            // test aid: E2V_SPLITK_FORCE = S runs the launch as split-K with S runs where it is eligible (the graph asks for it by itself
            // in the small-batch family, model.cpp Runner::sk_setup)
            static const int* const sk_force = knob("E2V_SPLITK_FORCE", 0);
            Act skws;
            if (*sk_force >= 2 && !g.upsample) {
                const int runs = splitk_plan(g, *sk_force);
                if (runs >= 2) { skws = Act(c->pool, (int64_t)runs * g.M, g.N); g.sk = *sk_force; g.sk_ws = skws.p; }
            }
            if (c1 == 0 && c0p == c0 && bgemm_up2x_applies(g)) {   // exact 2x resize: the sub-pixel form the graph runner takes
                const size_t n = 4 * (size_t)cout * conv_up2x_packed_ld(cin);
                Act u32(c->pool, (int64_t)((n + 1023) / 1024), 1024), u16(c->pool, (int64_t)((n + 2047) / 2048), 1024);
                pack_conv_up2x(w_oihw, u32.p, cout, cin, s);
                to_h16(u32.p, u16.p, n, c->h16_mode, s);
                bgemm_up2x_launch(g, u16.p, s);
                E2V_HIP(hipGetLastError());
                return;
            }
            igemm(g, s);
            E2V_HIP(hipGetLastError());
            return;
        }
        const bool wino_shape = stride == 1 && pad_lo == 1 && Hi == Ho && Wi == Wo && cout % 4 == 0;
        int wm = 0;                                            // same policy as the graph runner (model.cpp: Runner::winograd)
        if (wino_shape && c->conv_algo == E2V_CONV_WINOGRAD) wm = 2;
        if (wino_shape && c->conv_algo == E2V_CONV_WINOGRAD4) wm = 4;
        if (wino_shape && c->conv_algo == E2V_CONV_AUTO) {
            const int cmin = std::min(cin, cout);
            const double padded = (double)((Ho + 3) / 4 * 4) * ((Wo + 3) / 4 * 4);
            if (c->wino_f4 && cmin >= c->wino4_min_c && padded <= c->wino_f4_pad * Ho * Wo) wm = 4;
            else if (cmin >= c->wino_min_c) wm = 2;
        }
        if (wm) {
            Act u(c->pool, (int64_t)(wm + 2) * (wm + 2) * cout, cin);
            wino_pack_weights(w_oihw, u.p, cout, cin, wm, s);
            WinoArgs a;
            a.m = wm;
            a.x0 = x0; a.c0 = c0; a.ld0 = c0; a.x1 = x1; a.c1 = c1; a.ld1 = c1;
            a.nimg = n_img; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo;
            if (Hi != Hs || Wi != Ws) { a.upsample = 1; a.ups_h = (float)Hs / (float)Hi; a.ups_w = (float)Ws / (float)Wi; }
            a.U = u.p; a.N = cout; a.out = out; a.ldc = cout; a.bias = bias;
            Act u3;
            if (c->x3_compute) {
                const size_t n = (size_t)(wm + 2) * (wm + 2) * cout * cin;
                u3 = Act(c->pool, (int64_t)((3 * n + 1) / 2 + 1023) / 1024, 1024);
                split_bf16x3(u.p, u3.p, n, n, s);
                a.U3 = u3.p;
            }
            a.rowbias = rowbias; a.rb_ld = cout; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
            a.resid = resid; a.ldr = cout;
            const int chunk = wino_chunk_images(a, c->wino_ws_floats);
            const size_t need = wino_workspace_floats(a, chunk);
            Act ws(c->pool, (int64_t)((need + 1023) / 1024), 1024);
            wino_conv3x3(a, ws.p, chunk, s);
            E2V_HIP(hipGetLastError());
            return;
        }
        const int ld32 = conv3x3_packed_ld(cin, 32);
        Act wp(c->pool, cout, ld32);
        pack_conv3x3(w_oihw, wp.p, cout, cin, 32, s);
        IgemmArgs g;
        g.a0 = x0; g.c0 = c0; g.lda0 = c0; g.a1 = x1; g.c1 = c1; g.lda1 = c1;
        g.w = wp.p; g.ldw = ld32; g.out = out; g.ldc = cout; g.bias = bias;
        g.rowbias = rowbias; g.rb_ld = cout; g.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
        g.resid = resid; g.ldr = cout; g.M = n_img * Ho * Wo; g.N = cout; g.taps = 9;
        g.Ho = Ho; g.Wo = Wo; g.Hi = Hi; g.Wi = Wi; g.Hs = Hs; g.Ws = Ws; g.stride = stride; g.pad = pad_lo;
        if (Hi != Hs || Wi != Ws) { g.upsample = 1; g.ups_h = (float)Hs / (float)Hi; g.ups_w = (float)Ws / (float)Wi; }
        igemm(g, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_linear(e2v_ctx* c, const float* x, int ldx, int64_t M, int K, const float* w, const float* bias, int N,
                         const float* resid, int geglu, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        // fp32 arithmetic reads 16-byte row pieces of the caller's tensor; the bf16-activation mode re-lays its operands out (K padded to 8)
        E2V_REQUIRE(x && w && out && (c->bf16_compute || (K % 4 == 0 && ldx % 4 == 0)), E2V_EINVAL, "bad linear arguments");
        hipStream_t s = S(c, stream);
        IgemmArgs g;
        g.a0 = x; g.c0 = K; g.lda0 = ldx; g.ldw = K; g.out = out; g.M = (int)M; g.taps = 1; g.resid = resid;
        Act wp, bp;
        if (geglu) {
            E2V_REQUIRE(N % 32 == 0 && bias, E2V_EINVAL, "GEGLU width must be a multiple of 32 and have a bias");
            wp = Act(c->pool, 2 * N, K);
            bp = Act(c->pool, 1, 2 * N);
            for (int q = 0; q < N / 32; ++q) {
                copy_rows(w + (size_t)(q * 32) * K, K, wp.p + (size_t)(q * 64) * K, K, 32, K, s);
                copy_rows(w + (size_t)(N + q * 32) * K, K, wp.p + (size_t)(q * 64 + 32) * K, K, 32, K, s);
                copy_rows(bias + q * 32, 32, bp.p + q * 64, 32, 1, 32, s);
                copy_rows(bias + N + q * 32, 32, bp.p + q * 64 + 32, 32, 1, 32, s);
            }
            g.w = wp.p; g.bias = bp.p; g.N = 2 * N; g.ldc = N; g.geglu = 1;
        } else {
            g.w = w; g.bias = bias; g.N = N; g.ldc = N; g.ldr = N;
        }
        Act w16, wpad, a16;
        if (c->bf16_compute) {       // bf16-activation mode: A and W rounded to bf16 once (K zero-padded to 8), fp32 result
            const int K8 = (K + 7) / 8 * 8;
            a16 = Act(c->pool, M, K8, true);
            cvt_rows(x, ldx, 0, a16.p, K8, c->h16_mode, M, K, K8, s);
            w16 = Act(c->pool, g.N, K8, true);
            cvt_rows(g.w, K, 0, w16.p, K8, c->h16_mode, g.N, K, K8, s);
            g.a0 = a16.p; g.c0 = K8; g.lda0 = K8;
            g.a_bf16 = c->h16_mode; g.out_f32 = 1; g.w16 = w16.p; g.ldw16 = K8;
        }
        static const int* const sk_force = knob("E2V_SPLITK_FORCE", 0);      // test aid, as in e2v_op_conv3x3
        Act skws;
        if (c->bf16_compute && *sk_force >= 2) {
            const int runs = splitk_plan(g, *sk_force);
            if (runs >= 2) { skws = Act(c->pool, (int64_t)runs * g.M, g.N); g.sk = *sk_force; g.sk_ws = skws.p; }
        }
        Act w3;
        if (c->x3_compute) {
            const size_t n = (size_t)g.N * K;
            w3 = Act(c->pool, (int64_t)((3 * n + 1) / 2 + 1023) / 1024, 1024);
            split_bf16x3(g.w, w3.p, n, n, s);
            g.x3 = 1; g.w3 = w3.p; g.w3_plane = (long long)n;
        }
        igemm(g, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_groupnorm(e2v_ctx* c, const float* x0, int c0, const float* x1, int c1, int samples, int P, int groups,
                            float eps, const float* gamma, const float* beta, int act, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        const int C = c0 + c1;
        E2V_REQUIRE(x0 && gamma && beta && out && c0 % 4 == 0 && c1 % 4 == 0 && C % groups == 0, E2V_EINVAL, "bad groupnorm arguments");
        hipStream_t s = S(c, stream);
        Act part(c->pool, (int64_t)samples * groupnorm_chunks(P), C * 2);
        Act sc(c->pool, samples, C * 2);
        GroupNormArgs a;
        a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1; a.ld0 = c0; a.ld1 = c1; a.gamma = gamma; a.beta = beta;
        a.out = out; a.ldo = C; a.samples = samples; a.P = P; a.groups = groups; a.eps = eps; a.silu = act;
        a.ws_part = part.p; a.ws_scale = sc.p;
        if (c->bf16_compute) {       // bf16-activation mode: bf16 rows in and out (fp32 statistics), converted at this boundary
            const int64_t rows = (int64_t)samples * P;
            Act b0(c->pool, rows, c0, true), b1, bo(c->pool, rows, C, true);
            cvt_rows(x0, c0, 0, b0.p, c0, c->h16_mode, rows, c0, c0, s);
            if (c1 > 0) { b1 = Act(c->pool, rows, c1, true); cvt_rows(x1, c1, 0, b1.p, c1, c->h16_mode, rows, c1, c1, s); }
            a.bf16 = c->h16_mode; a.x0 = b0.p; a.x1 = c1 > 0 ? b1.p : nullptr; a.out = bo.p;
            static const int* const fused = knob("E2V_GN_FUSED_SMALL", 1);      // 2 (test aid): the one-kernel form of the small-batch family here too
            a.fused_small = *fused == 2 ? 1 : 0;
            static const int* const coop = E2V_AB_KNOB("E2V_GN_COOP", 0);        // `make ab` builds: 2 = the cooperative one-launch form here too
            a.small_chunks = *coop == 2 ? 1 : 0;
            groupnorm(a, s);
            cvt_rows(bo.p, C, c->h16_mode, out, C, 0, rows, C, C, s);
        } else {
            groupnorm(a, s);
        }
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_layernorm(e2v_ctx* c, const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                            float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(x && gamma && beta && out && C % 4 == 0 && C <= 1280, E2V_EINVAL, "bad layernorm arguments");
        hipStream_t s = S(c, stream);
        if (c->bf16_compute) {
            Act bi(c->pool, rows, C, true), bo(c->pool, rows, C, true);
            cvt_rows(x, C, 0, bi.p, C, c->h16_mode, rows, C, C, s);
            layernorm(bi.p, C, gamma, beta, bo.p, C, (int)rows, C, eps, s, c->h16_mode);
            cvt_rows(bo.p, C, c->h16_mode, out, C, 0, rows, C, C, s);
        } else {
            layernorm(x, C, gamma, beta, out, C, (int)rows, C, eps, s);
        }
        E2V_HIP(hipGetLastError());
    });
}

// Test aid: the canonical row-block sums a GroupNorm takes instead of its statistics pass (norm.hip rowblock_sums; what the staged
// epilogue of bgemm_t256_kernel leaves with a conv's output): x is rounded to bf16 first, out[rows / 64][C][2] = (sum, sum of squares).
e2v_status e2v_op_rowblock_sums(e2v_ctx* c, const float* x, int64_t rows, int C, float* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(x && out && rows > 0 && rows % 64 == 0 && C > 0 && C % 8 == 0, E2V_EINVAL, "rowblock_sums: rows must be a multiple of 64, C of 8");
        hipStream_t s = S(c, stream);
        Act b(c->pool, rows, C, true);
        cvt_rows(x, C, 0, b.p, C, H16_BF16, rows, C, C, s);
        rowblock_sums(b.p, C, C, rows, rbsum_rows_per_pass(C), out, s);
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_attention(e2v_ctx* c, const float* q, int ldq, const float* k, const float* v, int ldkv, float* o, int ldo,
                            int n, int F, int heads, int D, int Nq, int Nk, int mode, float scale, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(q && k && v && o, E2V_EINVAL, "null argument");
        E2V_REQUIRE(D == 8 || D == 16 || D == 32 || D == 40 || D == 64 || D == 80 || D == 160, E2V_EINVAL,
                    "head dim must be one of 8, 16, 32, 40, 64, 80, 160");
        E2V_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && Nq > 0 && Nk > 0, E2V_EINVAL, "bad strides / sizes");
        E2V_REQUIRE(mode == 1 || Nq == Nk, E2V_ESHAPE, "self-attention needs Nq == Nk");
        AttnArgs a;
        a.q = q; a.ldq = ldq; a.k = k; a.v = v; a.ldkv = ldkv; a.o = o; a.ldo = ldo; a.n = n; a.F = F; a.heads = heads; a.D = D;
        a.Nq = Nq; a.Nk = Nk; a.mode = mode; a.scale = scale; a.x3 = c->x3_compute ? 1 : 0;
        hipStream_t s = S(c, stream);
        if (c->bf16_compute) {       // bf16-activation mode: Q, K, V rounded to bf16 rows once; O comes back as bf16
            const int C = heads * D;
            const int64_t qrows = (int64_t)n * F * Nq, krows = mode == 0 ? (int64_t)n * F * Nk : (int64_t)n * Nk;
            Act bq(c->pool, qrows, C, true), bk(c->pool, krows, C, true), bv(c->pool, krows, C, true), bo(c->pool, qrows, C, true);
            cvt_rows(q, ldq, 0, bq.p, C, c->h16_mode, qrows, C, C, s);
            cvt_rows(k, ldkv, 0, bk.p, C, c->h16_mode, krows, C, C, s);
            cvt_rows(v, ldkv, 0, bv.p, C, c->h16_mode, krows, C, C, s);
            a.q = bq.p; a.ldq = C; a.k = bk.p; a.v = bv.p; a.ldkv = C; a.o = bo.p; a.ldo = C; a.io_bf16 = c->h16_mode;
            flash_attention(a, s);
            cvt_rows(bo.p, C, c->h16_mode, o, ldo, 0, qrows, C, C, s);
        } else {
            flash_attention(a, s);
        }
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_temporal_attention(e2v_ctx* c, const float* qkv, float* out, int n, int F, int HW, int heads, int D,
                                     float scale, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(qkv && out && F <= 8 && D % 4 == 0, E2V_EINVAL, "bad temporal attention arguments");
        const int C = heads * D;
        hipStream_t s = S(c, stream);
        if (c->bf16_compute) {
            const int64_t rows = (int64_t)n * F * HW;
            Act bi(c->pool, rows, 3 * C, true), bo(c->pool, rows, C, true);
            cvt_rows(qkv, 3 * C, 0, bi.p, 3 * C, c->h16_mode, rows, 3 * C, 3 * C, s);
            temporal_attention(bi.p, 3 * C, bo.p, C, n, F, HW, heads, D, scale, s, c->h16_mode);
            cvt_rows(bo.p, C, c->h16_mode, out, C, 0, rows, C, C, s);
        } else {
            temporal_attention(qkv, 3 * C, out, C, n, F, HW, heads, D, scale, s);
        }
        E2V_HIP(hipGetLastError());
    });
}

e2v_status e2v_op_to_channels_last(e2v_ctx* c, const float* in, float* out, int n, int C, int Cpad, int FHW, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] { ncfhw_to_cl(in, out, n, C, Cpad, FHW, 1.0f, S(c, stream)); E2V_HIP(hipGetLastError()); });
}

e2v_status e2v_op_from_channels_last(e2v_ctx* c, const float* in, int ld, float* out, int n, int C, int FHW, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] { cl_to_ncfhw(in, ld, out, n, C, FHW, 1.f, 0.f, 0, 0.f, 0.f, S(c, stream)); E2V_HIP(hipGetLastError()); });
}

e2v_status e2v_op_set_knob(const char* name, int value) {
    if (!name) return E2V_EINVAL;
    return set_knob(name, value) ? E2V_OK : E2V_EINVAL;
}

}  // extern "C"
