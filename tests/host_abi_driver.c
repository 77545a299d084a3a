/* Drives the host-only part of the C ABI (e2v_create(device = -1): key scheme, DDIM schedule, argument checks, error paths)
 * -- built against the AddressSanitizer build of the library by tests/test_host_asan.py.  Plain C: this is also the proof
 * that include/eeg2video_hip.h is consumable from C.  Exit code 0 = every check held and ASan reported nothing. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eeg2video_hip.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void) {
    e2v_config cfg;
    CHECK(e2v_config_size() == (int64_t)sizeof(e2v_config));   /* header and library agree on the struct */
    e2v_default_config(&cfg);
    CHECK(cfg.block_out_channels[0] == 320 && cfg.cross_attention_dim == 768 && cfg.num_train_timesteps == 1000);
    e2v_ctx* ctx = NULL;
    CHECK(e2v_create(&cfg, -1, &ctx) == E2V_OK && ctx != NULL);
    /* every expected key: name + shape round trip */
    int64_t n = e2v_num_expected_keys(ctx), total = 0;
    CHECK(n == 798 + 248 + 10);
    for (int64_t i = 0; i < n; ++i) {
        int64_t shape[4];
        int nd = 0;
        const char* k = e2v_expected_key(ctx, i, shape, &nd);
        CHECK(k != NULL && nd >= 1 && nd <= 4 && strlen(k) > 0);
        int64_t numel = 1;
        for (int d = 0; d < nd; ++d) numel *= shape[d];
        total += numel;
    }
    CHECK(total > 900000000);                                  /* UNet 3D + VAE + semantic predictor */
    CHECK(e2v_expected_key(ctx, n, NULL, NULL) == NULL && e2v_expected_key(ctx, -1, NULL, NULL) == NULL);
    /* schedules */
    int64_t ts[1000];
    CHECK(e2v_ddim_timesteps(ctx, 50, ts) == E2V_OK && ts[0] == 981 && ts[49] == 1);
    CHECK(e2v_ddim_timesteps(ctx, 4, ts) == E2V_OK && ts[0] == 751 && ts[3] == 1);
    CHECK(e2v_ddim_timesteps(ctx, 1000, ts) == E2V_OK && ts[999] == 1);
    CHECK(e2v_ddim_timesteps(ctx, 0, ts) == E2V_EINVAL && e2v_ddim_timesteps(ctx, 1001, ts) == E2V_EINVAL);
    float* ab = (float*)malloc(1000 * sizeof(float));
    CHECK(e2v_ddim_alphas_cumprod(ctx, ab) == E2V_OK && ab[0] > 0.999f && ab[999] < 0.005f && ab[999] > 0.004f);
    CHECK(e2v_set_alphas_cumprod(ctx, ab, 999) == E2V_EINVAL && e2v_set_alphas_cumprod(ctx, ab, 1000) == E2V_OK);
    CHECK(e2v_set_ddim_schedule(ctx, ab, 500, 0) == E2V_OK);
    CHECK(e2v_ddim_timesteps(ctx, 5, ts) == E2V_OK && ts[0] == 400 && ts[4] == 0);
    CHECK(e2v_ddim_timesteps(ctx, 501, ts) == E2V_EINVAL);
    free(ab);
    /* device entry points refuse the host-only context, with a message */
    float x[4] = {0, 0, 0, 0};
    int64_t shp[1] = {4};
    CHECK(e2v_load_tensor(ctx, "conv_in.bias", x, E2V_F32, shp, 1) == E2V_ESTATE);
    CHECK(strstr(e2v_last_error(ctx), "host-only") != NULL);
    CHECK(e2v_finalize_weights(ctx, 1) == E2V_ESTATE && e2v_finalize_weights(ctx, 0) == E2V_ESTATE);
    CHECK(e2v_generate(ctx, x, x, x, 1, 1, 1, 1, 1, 1, 4, 7.5f, 0.f, x, NULL, NULL) == E2V_ESTATE);
    CHECK(e2v_set_compute_dtype(ctx, E2V_BF16) == E2V_OK && e2v_set_compute_dtype(ctx, 17) == E2V_EINVAL);
    CHECK(e2v_set_conv_algo(ctx, E2V_CONV_DIRECT) == E2V_OK && e2v_set_conv_algo(ctx, 9) == E2V_EINVAL);
    CHECK(e2v_device_bytes(ctx) == 0);
    e2v_destroy(ctx);
    /* rejected configurations leave nothing behind */
    cfg.block_out_channels[0] = 384;                           /* head dim 48: no kernel instance */
    ctx = NULL;
    CHECK(e2v_create(&cfg, -1, &ctx) == E2V_EINVAL && ctx == NULL && strstr(e2v_last_error(NULL), "head dim") != NULL);
    cfg.block_out_channels[0] = 320;
    cfg.norm_num_groups = 0;
    CHECK(e2v_create(&cfg, -1, &ctx) == E2V_EINVAL);
    CHECK(e2v_create(NULL, -1, &ctx) == E2V_EINVAL);
    e2v_destroy(NULL);
    printf("host_abi_driver: ok (%lld keys, %lld parameters)\n", (long long)n, (long long)total);
    return 0;
}
