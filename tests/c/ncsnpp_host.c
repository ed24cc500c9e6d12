/* A NON-PYTHON host of the whole-network C ABI (include/storm_hip.h: storm_ncsnpp_*): builds an NCSN++ from raw state_dict
 * tensors on disk, evaluates it once, writes the output.  tests/test_abi.py compiles this file (gcc against the host
 * simulation library on CPU, hipcc -DUSE_HIP against libstorm_hip.so on the GPU), feeds it the reference's golden input
 * and compares its output with the reference's (fixture F2): no Python planning, packing or dispatch is involved.
 *
 * usage: ncsnpp_host <dir> <nf> <input_channels> <B> <F> <T> <dtype>
 *   <dir>/w<i>.bin   fp32 tensor i of the state_dict (order of storm_ncsnpp_tensor_info)
 *   <dir>/x<j>.bin   complex64 [B][F][T] input channel j;  <dir>/t.bin fp32 [B];  writes <dir>/out.bin complex64 [B][F][T] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "storm_hip.h"
#ifdef USE_HIP
#include <hip/hip_runtime_api.h>
static void* dev_alloc(size_t n) { void* p = NULL; if (hipMalloc(&p, n ? n : 1) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); } return p; }
static void to_dev(void* d, const void* h, size_t n) { if (hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) exit(2); }
static void to_host(void* h, const void* d, size_t n) { if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, n, hipMemcpyDeviceToHost) != hipSuccess) exit(2); }
#else
static void* dev_alloc(size_t n) { void* p = calloc(n ? n : 1, 1); if (!p) exit(2); return p; }
static void to_dev(void* d, const void* h, size_t n) { memcpy(d, h, n); }
static void to_host(void* h, const void* d, size_t n) { memcpy(h, d, n); }
#endif

static void* load(const char* dir, const char* name, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    void* h = malloc(bytes);
    if (fread(h, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read (%zu bytes expected)\n", path, bytes); exit(2); }
    fclose(f);
    void* d = dev_alloc(bytes);
    to_dev(d, h, bytes);
    free(h);
    return d;
}
#define CHECK(call) do { int rc_ = (call); if (rc_ != STORM_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, storm_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: %s dir nf input_channels B F T dtype\n", argv[0]); return 2; }
    const char* dir = argv[1];
    storm_ncsnpp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.nf = atoi(argv[2]); cfg.n_levels = 4; cfg.ch_mult[0] = 1; cfg.ch_mult[1] = 2; cfg.ch_mult[2] = 2; cfg.ch_mult[3] = 2;
    cfg.num_res_blocks = 1; cfg.n_attn = 1; cfg.attn_resolutions[0] = 0; cfg.image_size = 256;
    cfg.input_channels = atoi(argv[3]); cfg.discriminative = 0;
    const int B = atoi(argv[4]), F = atoi(argv[5]), T = atoi(argv[6]), dtype = atoi(argv[7]);
    const int n = storm_ncsnpp_num_tensors(&cfg);
    if (n <= 0) { fprintf(stderr, "bad config: %s\n", storm_last_error()); return 1; }
    const void** w = (const void**)malloc(sizeof(void*) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        char name[128], file[32]; int nd; long long sh[4];
        CHECK(storm_ncsnpp_tensor_info(&cfg, i, name, sizeof name, &nd, sh));
        size_t numel = 1;
        for (int d = 0; d < nd; ++d) numel *= (size_t)sh[d];
        snprintf(file, sizeof file, "w%d.bin", i);
        w[i] = load(dir, file, numel * 4);
    }
    storm_ncsnpp* net = NULL;
    CHECK(storm_ncsnpp_create(&cfg, w, n, dtype, NULL, NULL, &net));
    const long long ws_bytes = storm_ncsnpp_workspace_bytes(net, B, F, T);
    if (ws_bytes < 0) { fprintf(stderr, "workspace: %s\n", storm_last_error()); return 1; }
    void* ws = dev_alloc((size_t)ws_bytes);
    const size_t spec = (size_t)B * F * T * 8;
    const int n_parts = cfg.input_channels / 2;
    const void* parts[3];
    for (int j = 0; j < n_parts; ++j) { char file[32]; snprintf(file, sizeof file, "x%d.bin", j); parts[j] = load(dir, file, spec); }
    const float* t = (const float*)load(dir, "t.bin", (size_t)B * 4);
    void* out = dev_alloc(spec);
    CHECK(storm_ncsnpp_forward(net, parts, n_parts, t, out, ws, ws_bytes, B, F, T, 0, NULL));
    void* h = malloc(spec);
    to_host(h, out, spec);
    char path[1024];
    snprintf(path, sizeof path, "%s/out.bin", dir);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(h, 1, spec, f) != spec) { fprintf(stderr, "cannot write %s\n", path); return 1; }
    fclose(f);
    storm_ncsnpp_destroy(net);
    printf("ncsnpp_host: %d tensors, workspace %lld bytes, forward OK\n", n, ws_bytes);
    return 0;
}
