// llama_logits.cpp -- integration harness (TEST INFRASTRUCTURE): drives the UNMODIFIED reference libllama through its public API
// (include/llama.h) on a GGUF file and dumps the logits, so that a run with every layer offloaded to the ggml-hip-cdna4 shim
// (-ngl 99) can be compared with the reference CPU backend (-ngl 0) on the same tokens (tests/test_gpu_llama.py).
//   llama_logits <model.gguf> <ngl> <n_tokens> <n_threads> <split_mode: none|layer|graph> <out.bin> [n_decode]
// Evaluates tokens t_i = (7 i + 3) mod n_vocab as ONE batch (prompt path), writes the logits of the last token; with n_decode > 0 it then
// feeds n_decode further tokens one by one (decode path) and appends each step's logits.
#include "llama.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s model ngl n_tokens n_threads split_mode out [n_decode]\n", argv[0]); return 2; }
    const char *path = argv[1]; const int ngl = atoi(argv[2]), n_tokens = atoi(argv[3]), n_threads = atoi(argv[4]);
    const char *sm = argv[5], *out = argv[6]; const int n_decode = argc > 7 ? atoi(argv[7]) : 0;
    llama_backend_init();
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = ngl;
    mp.split_mode = !strcmp(sm, "graph") ? LLAMA_SPLIT_MODE_GRAPH : !strcmp(sm, "layer") ? LLAMA_SPLIT_MODE_LAYER : LLAMA_SPLIT_MODE_NONE;
    mp.use_mmap = getenv("LLAMA_LOGITS_NO_MMAP") == nullptr;
    // LLAMA_LOGITS_TENSOR_SPLIT="3,1,2,2": uneven shares over the devices (llama-bench -ts), llama_model_params::tensor_split
    static float ts[128] = {0};
    if (const char *e = getenv("LLAMA_LOGITS_TENSOR_SPLIT")) {
        int n = 0; for (const char *q = e; *q && n < 128; ++n) { ts[n] = (float)atof(q); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
        mp.tensor_split = ts;
    }
    llama_model *model = llama_model_load_from_file(path, mp);
    if (!model) { fprintf(stderr, "failed to load %s\n", path); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = n_tokens + n_decode + 8; cp.n_batch = n_tokens > 0 ? n_tokens : 1; cp.n_ubatch = cp.n_batch;
    cp.n_threads = n_threads; cp.n_threads_batch = n_threads;
    // LLAMA_LOGITS_KV_OFFLOAD: KV cache in device memory (CPY + attention run on the device); otherwise in host memory (llama-bench -nkvo 1),
    // and the scheduler gives the ops that write and read it to the CPU backend.
    cp.offload_kqv = getenv("LLAMA_LOGITS_KV_OFFLOAD") != nullptr;
    // LLAMA_LOGITS_CACHE_TYPE=q8_0: quantized K / V cache (llama-bench -ctk q8_0 -ctv q8_0; needs flash attention, which is this build's default)
    if (const char *ct = getenv("LLAMA_LOGITS_CACHE_TYPE")) { if (!strcmp(ct, "q8_0")) { cp.type_k = GGML_TYPE_Q8_0; cp.type_v = GGML_TYPE_Q8_0; } }
    llama_context *ctx = llama_init_from_model(model, cp);
    if (!ctx) { fprintf(stderr, "failed to create the context\n"); return 1; }
    const int n_vocab = llama_n_vocab(model);
    std::vector<llama_token> tok(n_tokens);
    for (int i = 0; i < n_tokens; ++i) tok[i] = (7 * i + 3) % n_vocab;
    FILE *f = fopen(out, "wb"); if (!f) { perror(out); return 1; }
    if (llama_decode(ctx, llama_batch_get_one(tok.data(), n_tokens, 0, 0)) != 0) { fprintf(stderr, "llama_decode (prompt) failed\n"); return 1; }
    llama_synchronize(ctx);
    fwrite(llama_get_logits_ith(ctx, n_tokens - 1), sizeof(float), n_vocab, f);
    for (int i = 0; i < n_decode; ++i) {
        llama_token t = (11 * i + 5) % n_vocab;
        if (llama_decode(ctx, llama_batch_get_one(&t, 1, n_tokens + i, 0)) != 0) { fprintf(stderr, "llama_decode (token %d) failed\n", i); return 1; }
        llama_synchronize(ctx);
        fwrite(llama_get_logits_ith(ctx, 0), sizeof(float), n_vocab, f);
    }
    fclose(f);
    llama_free(ctx); llama_free_model(model); llama_backend_free();
    return 0;
}
