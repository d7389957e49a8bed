// Sanitizer driver of libradarml_hip.so (tools/sanitize/build.py): a plain program over the C ABI of include/radarml.h plus the
// host-only internals of csrc/ (model packing, Pillow table builder).  Built with -fsanitize=address,undefined against the
// host-only library it runs on a box without a GPU (phase A: every path that precedes the first HIP call, and the graceful failure
// of that call); linked against the real library on a GPU box it adds phase B: four host threads on two streams sharing ONE
// rml_ctx through the fused projection -> SVM pipeline, rml_project and rml_svm_decision, results compared bit for bit.
// Exit status 0 = every check passed (a sanitizer report aborts the process with its own status).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <thread>
#include <vector>
#include "../../radar-ml_amd/csrc/rml_internal.h"
#include "../../radar-ml_amd/csrc/resize_tables.h"

static int g_fail = 0;
#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) { fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

static uint64_t g_rng = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 32); }

// ---- phase A ------------------------------------------------------------------------------------------------------------------
static void phase_a_arguments() {
    CHECK(rml_feature_len(22, 31, 176, RML_MASK_ALL) == 10010);
    CHECK(rml_feature_len(64, 64, 128, RML_MASK_ALL) == 20480);
    CHECK(rml_feature_len(22, 31, 176, RML_MASK_XY) == 682);
    CHECK(rml_feature_len(22, 31, 176, 0) == 0);
    CHECK(rml_version() != nullptr && strlen(rml_version()) > 0);
    // NULL context / NULL pointers: a status and a message, never a dereference
    float f = 0; double d = 0; int32_t i3[3] = {0, 0, 0}; int64_t n = 0; rml_svm* m = nullptr; rml_linear* lm = nullptr;
    CHECK(rml_project(nullptr, &f, RML_VOL_F32, 1, 2, 2, 4, RML_MODE_MAX, nullptr, 0.f, RML_MASK_ALL, &f, 16, nullptr, 0, nullptr, nullptr, nullptr, nullptr) < 0);
    CHECK(strlen(rml_last_error()) > 0);
    CHECK(rml_derive_targets(nullptr, &f, RML_VOL_F32, 1, 2, 2, 4, 1, i3, nullptr, nullptr) < 0);
    CHECK(rml_svm_load(nullptr, &d, 1, 1, &d, &d, i3, 3, RML_KERNEL_RBF, 0.1, 255.0, nullptr, nullptr, &m) < 0 && m == nullptr);
    CHECK(rml_svm_decision(nullptr, nullptr, 0, &f, 1, nullptr, 0, nullptr, nullptr, nullptr, 1, &d, nullptr, nullptr, nullptr, nullptr, nullptr) < 0);
    CHECK(rml_project_svm(nullptr, nullptr, &f, RML_VOL_F32, 1, 2, 2, 4, RML_MODE_MAX, nullptr, 255.f, RML_MASK_ALL, &d, nullptr, nullptr, nullptr, nullptr, nullptr) < 0);
    CHECK(rml_linear_load(nullptr, &d, &d, 3, 4, nullptr, nullptr, &lm) < 0);
    CHECK(rml_profile_read(nullptr, &n, &d, &n) < 0);
    CHECK(rml_probe_stream(nullptr, &f, 1 << 20, 1, &d, nullptr) < 0);
    CHECK(rml_ctx_set_option(nullptr, RML_OPT_PROJECT_SHARE_CU, 1) < 0);
    {
        int v = 0;
        CHECK(rml_ctx_get_option(nullptr, RML_OPT_CHUNK, &v) < 0 && rml_ctx_reserve_workspace(nullptr, 1 << 20) < 0 && rml_ctx_workspace_bytes(nullptr) == 0);
        CHECK(rml_dnn_trunk_x3_supported(80, 80) == 1 && rml_dnn_trunk_x3_supported(128, 128) == 0 && rml_dnn_trunk_x3_supported(81, 80) == 0);
        CHECK(rml_dnn_trunk_x3(nullptr, &f, &f, &f, 1, 80, 80, &f, &f, &f, &f, 2, &f, nullptr) < 0);
        CHECK(rml_dnn_exact_features_scratch_bytes(RML_VOL_F32, 10, 22, 31, 176, 80, 80, 1) > 10 * 22 * 31 * 176 * 4);
        CHECK(rml_dnn_exact_features(nullptr, &f, RML_VOL_F32, nullptr, 1, 2, 2, 4, RML_MODE_MAX, 80, 80, &f, &f, &f, &f, 2, &f, 1, &f, nullptr) < 0);
        CHECK(rml_dnn_top2_gap(nullptr, &f, 3, 1, 3, &f, nullptr) < 0);
        CHECK(rml_dnn_guard_apply(nullptr, &f, 3, 3, nullptr, 1, &f, 1e-4f, nullptr, nullptr, nullptr, nullptr) < 0);
    }
    CHECK(rml_svm_free(nullptr, nullptr) == RML_OK && rml_linear_free(nullptr, nullptr) == RML_OK && rml_ctx_destroy(nullptr) == RML_OK);
    CHECK(rml_svm_is_exact(nullptr) == 0 && rml_svm_num_sv(nullptr) == 0 && rml_svm_dim(nullptr) == 0);
    CHECK(rml_ctx_device(nullptr) < 0);
    rml_ctx* bad = nullptr;
    CHECK(rml_ctx_create(-5, &bad) < 0 && bad == nullptr);
    CHECK(rml_ctx_create(4099, &bad) < 0 && bad == nullptr);
    // pure-host predicates
    for (int X = 1; X <= 70; X += 7)
        for (int Y = 1; Y <= 70; Y += 9)
            for (int Z = 4; Z <= 260; Z += 36)
                for (int o : {16, 80, 128}) {
                    (void)rml_dnn_preprocess_supported(X, Y, Z, o, o);
                    (void)rml_derive_slice_supported(nullptr, &f, RML_VOL_F32, X, Y, Z, 1);
                    (void)rml_derive_slice_supported(nullptr, &f, RML_VOL_U8, X, Y, Z, 3);
                    (void)rml_code_rmw_default(rml_feature_len(X, Y, Z, RML_MASK_ALL), (int64_t)X * Y * Z, X & 1, Y & 1);
                }
    CHECK(rml_dnn_preprocess_supported(22, 31, 176, 80, 80) == 1);
    CHECK(rml_adam_entry_bytes() > 0);
}

static void phase_a_pillow_tables() {
    for (int in : {1, 2, 3, 22, 31, 64, 176, 500})
        for (int out : {1, 2, 3, 80, 128, 257}) {
            rmlresize::AxisTable t;
            rmlresize::precompute(in, out, t);
            CHECK((int)t.bounds.size() == 2 * out && (int)t.kk.size() == out * t.ksize);
            for (int x = 0; x < out; ++x) {
                const int x0 = t.bounds[2 * x], nt = t.bounds[2 * x + 1];
                CHECK(x0 >= 0 && nt >= 1 && nt <= t.ksize && x0 + nt <= in);
                double s = 0;
                for (int k = 0; k < nt; ++k) s += t.kk[(size_t)x * t.ksize + k];
                CHECK(fabs(s - 1.0) < 1e-12);
            }
        }
}

static void phase_a_model_packing() {
    struct Case { int64_t M, D; int C; int kind; double scale; };      // kind 0: codes/255 on the grid, 1: off the grid, 2: one NaN, 3: unscaled codes
    const Case cases[] = {{5, 7, 2, 0, 255.0}, {300, 1234, 3, 0, 255.0}, {17, 130, 6, 1, 255.0}, {64, 129, 3, 2, 255.0},
                          {128, 128, 4, 3, 1.0}, {129, 127, 5, 0, 255.0}, {1, 1, 2, 0, 255.0}};
    for (const Case& c : cases) {
        std::vector<double> sv((size_t)c.M * c.D), dc((size_t)(c.C - 1) * c.M);
        for (double& v : sv) {
            const int code = (int)(rnd() % 256);
            v = c.kind == 3 ? (double)code : (double)((float)code / 255.0f);
            if (c.kind == 1) v += 1e-3 * (double)(rnd() % 100);
        }
        if (c.kind == 2) sv[sv.size() / 2] = NAN;
        for (double& v : dc) v = ((double)(rnd() % 2001) - 1000.0) / 100.0;
        std::vector<int32_t> ns(c.C, 0);
        for (int64_t r = 0; r < c.M; ++r) ns[r % c.C]++;
        if (c.M < c.C) { std::fill(ns.begin(), ns.end(), 0); ns[0] = (int32_t)c.M; }
        rml_svm m;
        rml_svm_pack pk;
        int rc = rml_svm_pack_host(sv.data(), c.M, c.D, dc.data(), ns.data(), c.C, RML_KERNEL_RBF, 0.01, c.scale, true, &m, &pk);
        CHECK(rc == RML_OK);
        CHECK(m.Mpad >= c.M && m.Mpad % 128 == 0 && m.Dq >= c.D && m.Df >= c.D && ((m.Dq / 128) & 1) == 1);
        CHECK((int64_t)pk.W.size() == m.PT * m.Mpad && (int64_t)pk.svf.size() == m.Mpad * m.Df && (int64_t)pk.svq.size() == m.Mpad * m.Dq);
        CHECK(m.exact == (c.kind == 0 || c.kind == 3));
        CHECK(m.dig_ok ? ((int64_t)pk.svd.size() == 4 * m.Mpad * m.Dq && (int64_t)pk.dnsq.size() == m.Mpad) : pk.svd.empty());
        if (c.kind == 2) CHECK(!m.dig_ok);
        // linear kernel, no calibrators
        rc = rml_svm_pack_host(sv.data(), c.M, c.D, dc.data(), ns.data(), c.C, RML_KERNEL_LINEAR, 0.0, c.scale, false, &m, &pk);
        CHECK(rc == RML_OK && !m.dig_ok);
        // validation
        ns[0] += 1;
        CHECK(rml_svm_pack_host(sv.data(), c.M, c.D, dc.data(), ns.data(), c.C, RML_KERNEL_RBF, 0.01, c.scale, true, &m, &pk) == RML_ERR_INVALID);
        CHECK(strstr(rml_last_error(), "n_support") != nullptr);
        ns[0] -= 1;
        CHECK(rml_svm_pack_host(sv.data(), c.M, c.D, dc.data(), ns.data(), 7, RML_KERNEL_RBF, 0.01, c.scale, true, &m, &pk) == RML_ERR_UNSUPPORTED);
        CHECK(rml_svm_pack_host(sv.data(), c.M, c.D, dc.data(), ns.data(), c.C, 9, 0.01, c.scale, true, &m, &pk) == RML_ERR_UNSUPPORTED);
    }
}

static void phase_a_thread_local_errors() {
    // rml_last_error is thread-local: four threads fail differently at the same time and each reads its own message
    std::vector<std::thread> ts;
    int ok[4] = {0, 0, 0, 0};
    for (int t = 0; t < 4; ++t)
        ts.emplace_back([t, &ok] {
            float f = 0; double d = 0; int32_t i3[3] = {1, 1, 1}; rml_svm* m = nullptr; rml_svm sm; rml_svm_pack pk;
            int good = 1;
            for (int it = 0; it < 200; ++it) {
                const char* want = nullptr;
                if (t == 0) { (void)rml_svm_pack_host(&d, 3, 1, &d, i3, 9, RML_KERNEL_RBF, 0.1, 255.0, false, &sm, &pk); want = "classes"; }
                else if (t == 1) { (void)rml_svm_pack_host(&d, 5, 1, &d, i3, 3, RML_KERNEL_RBF, 0.1, 255.0, false, &sm, &pk); want = "n_support"; }
                else if (t == 2) { (void)rml_svm_load(nullptr, &d, 1, 1, &d, &d, i3, 3, RML_KERNEL_RBF, 0.1, 255.0, nullptr, nullptr, &m); want = "NULL"; }
                else { (void)rml_probe_stream(nullptr, &f, 4, 1, &d, nullptr); want = "rml_probe_stream"; }
                good &= strstr(rml_last_error(), want) != nullptr;
            }
            ok[t] = good;
        });
    for (auto& th : ts) th.join();
    CHECK(ok[0] && ok[1] && ok[2] && ok[3]);
}

// ---- phase B (a device is present) -----------------------------------------------------------------------------------------
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); ++g_fail; return; } } while (0)

static void phase_b(rml_ctx* ctx) {
    const int X = 22, Y = 31, Z = 176, C = 3;
    const int64_t B = 9000, M = 300;                   // two pipeline chunks (8192 + 808)
    const int64_t D = rml_feature_len(X, Y, Z, RML_MASK_ALL), vox = (int64_t)X * Y * Z;
    std::vector<float> vol((size_t)B * vox, 0.f);
    for (int64_t b = 0; b < B; ++b)
        for (int k = 0; k < 40; ++k) vol[(size_t)b * vox + rnd() % vox] = (float)(13 + rnd() % 243);
    std::vector<double> sv((size_t)M * D), dc((size_t)(C - 1) * M), ic(3), ca(3, -1.5), cb(3, 0.1);
    for (double& v : sv) v = (rnd() % 8) ? 0.0 : (double)((float)(rnd() % 256) / 255.0f);
    for (double& v : dc) v = ((double)(rnd() % 2001) - 1000.0) / 100.0;
    for (double& v : ic) v = ((double)(rnd() % 2001) - 1000.0) / 1000.0;
    int32_t ns[3] = {100, 100, 100};
    rml_svm* m = nullptr;
    CHECK(rml_svm_load(ctx, sv.data(), M, D, dc.data(), ic.data(), ns, C, RML_KERNEL_RBF, 0.01, 255.0, ca.data(), cb.data(), &m) == RML_OK);
    if (!m) return;
    CHECK(rml_svm_is_exact(m) == 1);
    float* dV = nullptr;
    HIPCK(hipMalloc(&dV, vol.size() * sizeof(float)));
    HIPCK(hipMemcpy(dV, vol.data(), vol.size() * sizeof(float), hipMemcpyHostToDevice));
    const int NT = 4;
    std::vector<std::vector<double>> dec(NT, std::vector<double>((size_t)B * 3));
    std::vector<std::vector<int32_t>> lab(NT, std::vector<int32_t>((size_t)B));
    hipStream_t st[2];
    HIPCK(hipStreamCreate(&st[0]));
    HIPCK(hipStreamCreate(&st[1]));
    int bad[NT] = {0, 0, 0, 0};
    std::vector<std::thread> ts;
    for (int t = 0; t < NT; ++t)
        ts.emplace_back([&, t] {
            double* d_dec = nullptr; double* d_pr = nullptr; int32_t* d_lv = nullptr; int32_t* d_lc = nullptr;
            if (hipSetDevice(rml_ctx_device(ctx)) != hipSuccess || hipMalloc(&d_dec, B * 3 * 8) != hipSuccess || hipMalloc(&d_pr, B * 3 * 8) != hipSuccess ||
                hipMalloc(&d_lv, B * 4) != hipSuccess || hipMalloc(&d_lc, B * 4) != hipSuccess) { bad[t] = 1; return; }
            for (int it = 0; it < 6; ++it) {
                int rc = rml_project_svm(ctx, m, dV, RML_VOL_F32, B, X, Y, Z, RML_MODE_MAX, nullptr, 255.f, RML_MASK_ALL, d_dec, nullptr, d_pr, d_lv, d_lc, st[t & 1]);
                if (rc != RML_OK) { fprintf(stderr, "thread %d: %s\n", t, rml_last_error()); bad[t] = 1; break; }
            }
            if (hipStreamSynchronize(st[t & 1]) != hipSuccess) bad[t] = 1;
            if (hipMemcpy(dec[t].data(), d_dec, B * 3 * 8, hipMemcpyDeviceToHost) != hipSuccess) bad[t] = 1;
            if (hipMemcpy(lab[t].data(), d_lc, B * 4, hipMemcpyDeviceToHost) != hipSuccess) bad[t] = 1;
            (void)hipFree(d_dec); (void)hipFree(d_pr); (void)hipFree(d_lv); (void)hipFree(d_lc);
        });
    for (auto& th : ts) th.join();
    for (int t = 0; t < NT; ++t) {
        CHECK(!bad[t]);
        CHECK(memcmp(dec[t].data(), dec[0].data(), dec[0].size() * 8) == 0);
        CHECK(memcmp(lab[t].data(), lab[0].data(), lab[0].size() * 4) == 0);
    }
    int hist[3] = {0, 0, 0};
    for (int32_t l : lab[0]) { CHECK(l >= 0 && l < 3); if (l >= 0 && l < 3) hist[l]++; }
    printf("phase B: %lld frames x %d threads on 2 streams, one context: identical results (labels %d/%d/%d)\n", (long long)B, NT, hist[0], hist[1], hist[2]);
    (void)hipStreamDestroy(st[0]); (void)hipStreamDestroy(st[1]);
    (void)hipFree(dV);
    CHECK(rml_svm_free(ctx, m) == RML_OK);
}

int main(int argc, char** argv) {
    const bool need_device = argc > 1 && strcmp(argv[1], "--need-device") == 0;
    phase_a_arguments();
    phase_a_pillow_tables();
    phase_a_model_packing();
    phase_a_thread_local_errors();
    printf("phase A: %s\n", g_fail ? "FAILED" : "ok");
    rml_ctx* ctx = nullptr;
    const int rc = rml_ctx_create(0, &ctx);
    if (rc != RML_OK) {
        // no GPU (or a host-only library): the failure must be a status with a message, and leave nothing behind
        CHECK(ctx == nullptr && strlen(rml_last_error()) > 0);
        printf("phase B: skipped, rml_ctx_create(0) = %d (%s)\n", rc, rml_last_error());
        if (need_device) { fprintf(stderr, "a device was required\n"); ++g_fail; }
    } else {
        phase_b(ctx);
        CHECK(rml_ctx_destroy(ctx) == RML_OK);
    }
    printf("%s (%d failed checks)\n", g_fail ? "FAILED" : "ALL OK", g_fail);
    return g_fail ? 1 : 0;
}
