// llama_soak.cpp -- integration harness (TEST INFRASTRUCTURE): repeats "prompt batch + n_decode single-token steps" of llama_logits.cpp many times through the
// UNMODIFIED reference libllama (include/llama.h) inside ONE process and checks that every repetition reproduces the first one bit for bit (64-bit FNV-1a of every
// logits row).  It exists to bound rare, non-reproducible wrong results on the captured / fused decode path of the ggml-hip-cdna4 shim ("weak" 1):
//   llama_soak <model.gguf> <ngl> <n_tokens> <n_decode> <iters> <n_threads> <split_mode: none|layer|graph> <mode: fresh|reuse> [ref.bin]
// mode fresh: a new llama_context (= a new backend instance: empty graph cache, first sighting -> eager, second -> capture, third -> replay) per repetition;
// mode reuse: one context, the KV cache cleared between repetitions (from the second repetition on every graph, the prompt's included, is a replay).
// ref.bin (optional): the logits of another run (e.g. -ngl 0) as written by this tool's first repetition with LLAMA_SOAK_DUMP=<file>: every row's NMSE against it is reported.
// Output: one JSON line on stdout; exit code 0 iff every repetition reproduced the first.
#include "llama.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p; uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static double nmse(const float *a, const float *b, int n) {
    double num = 0, den = 0; for (int i = 0; i < n; ++i) { const double d = (double)a[i] - b[i]; num += d * d; den += (double)b[i] * b[i]; }
    return num / (den > 1e-300 ? den : 1e-300);
}

int main(int argc, char **argv) {
    if (argc < 9) { fprintf(stderr, "usage: %s model ngl n_tokens n_decode iters n_threads split_mode fresh|reuse [ref.bin]\n", argv[0]); return 2; }
    const char *path = argv[1]; const int ngl = atoi(argv[2]), n_tokens = atoi(argv[3]), n_decode = atoi(argv[4]), iters = atoi(argv[5]), n_threads = atoi(argv[6]);
    const char *sm = argv[7]; const bool fresh = !strcmp(argv[8], "fresh"); const char *ref_path = argc > 9 ? argv[9] : nullptr;
    llama_backend_init();
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = ngl;
    mp.split_mode = !strcmp(sm, "graph") ? LLAMA_SPLIT_MODE_GRAPH : !strcmp(sm, "layer") ? LLAMA_SPLIT_MODE_LAYER : LLAMA_SPLIT_MODE_NONE;
    llama_model *model = llama_model_load_from_file(path, mp);
    if (!model) { fprintf(stderr, "failed to load %s\n", path); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = n_tokens + n_decode + 8; cp.n_batch = n_tokens > 0 ? n_tokens : 1; cp.n_ubatch = cp.n_batch;
    cp.n_threads = n_threads; cp.n_threads_batch = n_threads;
    cp.offload_kqv = getenv("LLAMA_LOGITS_KV_OFFLOAD") != nullptr;
    const int n_vocab = llama_n_vocab(model), rows = 1 + n_decode;
    std::vector<llama_token> tok(n_tokens);
    for (int i = 0; i < n_tokens; ++i) tok[i] = (7 * i + 3) % n_vocab;
    std::vector<float> first((size_t)rows * n_vocab), cur((size_t)rows * n_vocab), ref;
    if (ref_path) {
        FILE *f = fopen(ref_path, "rb"); ref.resize((size_t)rows * n_vocab);
        if (!f || fread(ref.data(), sizeof(float), ref.size(), f) != ref.size()) { fprintf(stderr, "cannot read %s\n", ref_path); return 1; }
        fclose(f);
    }
    std::vector<uint64_t> h0(rows);
    const double tol = getenv("LLAMA_SOAK_TOL") ? atof(getenv("LLAMA_SOAK_TOL")) : 0.0;      // rows whose hash differs but whose NMSE vs the first repetition is <= tol count as `rounding_rows` (atomic split-K sums)
    long rounding_rows = 0;
    long mismatched_rows = 0, nonfinite = 0; double max_nmse = 0, max_ref_nmse = 0; std::string bad;
    llama_context *ctx = nullptr;
    for (int it = 0; it < iters; ++it) {
        if (fresh || !ctx) { if (ctx) llama_free(ctx); ctx = llama_init_from_model(model, cp); if (!ctx) { fprintf(stderr, "failed to create the context\n"); return 1; } }
        else llama_kv_cache_clear(ctx);
        if (llama_decode(ctx, llama_batch_get_one(tok.data(), n_tokens, 0, 0)) != 0) { fprintf(stderr, "llama_decode (prompt) failed\n"); return 1; }
        llama_synchronize(ctx);
        memcpy(cur.data(), llama_get_logits_ith(ctx, n_tokens - 1), sizeof(float) * n_vocab);
        for (int i = 0; i < n_decode; ++i) {
            llama_token t = (11 * i + 5) % n_vocab;
            if (llama_decode(ctx, llama_batch_get_one(&t, 1, n_tokens + i, 0)) != 0) { fprintf(stderr, "llama_decode (token %d) failed\n", i); return 1; }
            llama_synchronize(ctx);
            memcpy(cur.data() + (size_t)(i + 1) * n_vocab, llama_get_logits_ith(ctx, 0), sizeof(float) * n_vocab);
        }
        for (int r = 0; r < rows; ++r) {
            const float *p = cur.data() + (size_t)r * n_vocab;
            for (int i = 0; i < n_vocab; ++i) if (!std::isfinite(p[i])) { ++nonfinite; break; }
            const uint64_t h = fnv1a(p, sizeof(float) * n_vocab);
            if (it == 0) { h0[r] = h; continue; }
            if (h != h0[r]) {
                const double e = nmse(p, first.data() + (size_t)r * n_vocab, n_vocab);
                if (e <= tol) { ++rounding_rows; continue; }
                ++mismatched_rows; if (e > max_nmse || e != e) max_nmse = e;
                if (bad.size() < 1500) { char b[96]; snprintf(b, sizeof(b), "%s[%d,%d,%.3g]", bad.empty() ? "" : ",", it, r, e); bad += b; }
            }
        }
        if (it == 0) {
            first = cur;
            if (const char *dump = getenv("LLAMA_SOAK_DUMP")) { FILE *f = fopen(dump, "wb"); if (f) { fwrite(first.data(), sizeof(float), first.size(), f); fclose(f); } }
        }
        if (!ref.empty()) for (int r = 0; r < rows; ++r) { const double e = nmse(cur.data() + (size_t)r * n_vocab, ref.data() + (size_t)r * n_vocab, n_vocab); if (e > max_ref_nmse || e != e) max_ref_nmse = e; }
    }
    printf("{\"model\": \"%s\", \"ngl\": %d, \"mode\": \"%s\", \"split\": \"%s\", \"n_tokens\": %d, \"n_decode\": %d, \"iters\": %d, \"rows_per_iter\": %d, \"mismatched_rows\": %ld, \"rounding_rows\": %ld, \"nonfinite_rows\": %ld, "
           "\"max_nmse_vs_first\": %.3g, \"max_nmse_vs_ref\": %s, \"hash_row0\": \"%016llx\", \"hash_last\": \"%016llx\", \"bad\": [%s]}\n",
           strrchr(path, '/') ? strrchr(path, '/') + 1 : path, ngl, fresh ? "fresh" : "reuse", sm, n_tokens, n_decode, iters, rows, mismatched_rows, rounding_rows, nonfinite, max_nmse,
           ref.empty() ? "null" : std::to_string(max_ref_nmse).c_str(), (unsigned long long)h0[0], (unsigned long long)h0[rows - 1], bad.c_str());
    if (ctx) llama_free(ctx);
    llama_free_model(model); llama_backend_free();
    return mismatched_rows == 0 && nonfinite == 0 ? 0 : 3;
}
