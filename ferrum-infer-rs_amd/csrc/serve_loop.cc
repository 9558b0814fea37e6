// Closed-loop continuous-batching driver over the C ABI — the host-side mirror, in C++, of the reference's engine
// iteration (ferrum-engine/src/continuous_engine/inner.rs:365 run_iteration: ask the scheduler for a mixed batch under a
// token budget, run ONE unified forward, retire finished requests) fed by a `ferrum bench-serve`-style client (closed loop:
// `concurrency` requests in flight, a new one is submitted the moment one completes; random prompt ids in [256, V), fixed
// output length, ignore_eos — ferrum-cli bench_serve, seed 9271).  It touches nothing but include/ferrum_hip.h:
//   * admission   : ferrum_hip_model_reserve_kv_slots for prompt + output tokens (ModelExecutor::reserve_kv_slots,
//                   model_executor.rs:484) — a request is admitted only if its whole KV footprint fits
//   * scheduling  : every iteration = one decode token for each running sequence + prompt chunks of the admitted ones, up
//                   to max_batched_tokens query tokens (BatchHint.max_tokens, scheduler.rs:108-179 tokens_to_process)
//   * forward     : ferrum_hip_model_unified_forward (mixed prefill + decode, device greedy sampling); iterations without
//                   any prompt chunk go through ferrum_hip_model_decode_steps (hipGraph replay) for as many steps as no
//                   request finishes or can be admitted
//   * retirement  : ferrum_hip_model_release
// Prints one JSON line: output tok/s over the whole run, TTFT p50/p99, iterations, and the KV pool state at exit (every
// block must be back).  Synthetic weights (no checkpoints offline); `--layers` shrinks the model for smoke runs.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/ferrum_hip.h"

namespace {

struct Request {
    uint64_t id = 0;
    std::vector<uint32_t> prompt;
    int prefilled = 0;        // prompt tokens already in the KV cache
    int generated = 0;        // output tokens sampled so far
    uint32_t last_token = 0;  // most recent output token (input of the next decode step)
    int out_len = 0;          // output tokens this request asks for
    double t_submit = 0, t_first = 0;
    std::vector<uint32_t> out;
};

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        if (int rc_ = (call)) {                                                                      \
            std::fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, ferrum_hip_last_error());    \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int arg_int(int argc, char** argv, const char* name, int dflt) {
    for (int i = 1; i + 1 < argc; i++)
        if (!std::strcmp(argv[i], name)) return std::atoi(argv[i + 1]);
    return dflt;
}
bool arg_flag(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; i++)
        if (!std::strcmp(argv[i], name)) return true;
    return false;
}

}  // namespace

int main(int argc, char** argv) {
    const int layers = arg_int(argc, argv, "--layers", arg_flag(argc, argv, "--dense") ? 32 : 48);   // Llama-3.1-8B: 32 layers, Qwen3-30B-A3B: 48
    const int num_requests = arg_int(argc, argv, "--requests", 96);
    const int conc = arg_int(argc, argv, "--concurrency", 32);
    const int PL = arg_int(argc, argv, "--prompt-len", 256);
    const int OL = arg_int(argc, argv, "--out-len", 128);
    const int budget = arg_int(argc, argv, "--max-batched-tokens", 8192);
    const int kv_blocks = arg_int(argc, argv, "--kv-blocks", 0);      // 0: room for concurrency + 2 whole requests; smaller pools make admission wait
    const int jitter = arg_int(argc, argv, "--out-len-jitter", 0);   // output lengths uniform in [OL − jitter, OL + jitter]: staggers retirements
    const int dense = arg_flag(argc, argv, "--dense");          // Llama-3.1-8B dims instead of Qwen3-30B-A3B
    const bool dump = arg_flag(argc, argv, "--dump-tokens");
    uint64_t seed = (uint64_t)arg_int(argc, argv, "--seed", 9271);
    if (conc < 1 || PL < 1 || OL < 1 || budget < conc || num_requests < 1 || jitter < 0 || jitter >= OL) {
        std::fprintf(stderr, "bad arguments\n");
        return 2;
    }

    FerrumHipModelConfig cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.num_layers = layers;
    if (dense) {   // Meta-Llama-3.1-8B (BASELINE configs[1])
        cfg.hidden = 4096; cfg.num_heads = 32; cfg.num_kv_heads = 8; cfg.head_dim = 128; cfg.intermediate = 14336;
        cfg.vocab = 128256; cfg.has_qk_norm = 0; cfg.rope_theta = 500000.0; cfg.rope_scaling_kind = 2;
        cfg.rope_p0 = 8.0; cfg.rope_p1 = 1.0; cfg.rope_p2 = 4.0; cfg.rope_p3 = 8192.0;
    } else {       // Qwen3-30B-A3B (BASELINE configs[2])
        cfg.hidden = 2048; cfg.num_heads = 32; cfg.num_kv_heads = 4; cfg.head_dim = 128; cfg.intermediate = 0;
        cfg.vocab = 151936; cfg.has_qk_norm = 1; cfg.num_experts = 128; cfg.top_k = 8; cfg.expert_inter = 768;
        cfg.norm_topk_prob = 1; cfg.rope_theta = 1000000.0;
    }
    const int seq_cap = ((PL + OL + jitter + 15) / 16) * 16;
    cfg.max_seq_len = seq_cap;
    cfg.group_size = 128;
    cfg.kv_num_blocks = kv_blocks > 0 ? kv_blocks : (conc + 2) * (seq_cap / 16);
    cfg.max_seqs = conc;
    cfg.max_tokens = budget;
    cfg.rms_eps = 1e-6f;
    cfg.tp_world = 1;

    FerrumHipModel* model = nullptr;
    CHECK(ferrum_hip_model_create(&model, &cfg));
    CHECK(ferrum_hip_model_init_synthetic(model, seed));
    CHECK(ferrum_hip_model_finalize(model));

    std::deque<Request> pending;
    for (int i = 0; i < num_requests; i++) {
        Request r;
        r.id = 1 + (uint64_t)i;
        r.prompt.resize(PL);
        for (int t = 0; t < PL; t++) r.prompt[t] = 256u + (uint32_t)(splitmix(seed) % (uint64_t)(cfg.vocab - 256));
        r.out_len = jitter ? OL - jitter + (int)(splitmix(seed) % (uint64_t)(2 * jitter + 1)) : OL;
        pending.push_back(std::move(r));
    }
    std::vector<Request> running, finished;
    std::vector<FerrumHipBatchItem> items;
    std::vector<uint32_t> sampled((size_t)conc * (size_t)(OL + jitter));
    std::vector<uint64_t> ids;
    std::vector<uint32_t> toks;
    long iterations = 0, graph_steps = 0, mixed_iterations = 0, out_tokens = 0;

    const double t0 = now_s();
    while (!pending.empty() || !running.empty()) {
        // admission: closed loop — a client submits as soon as a slot AND the KV footprint of a whole request are free
        while ((int)running.size() < conc && !pending.empty()) {
            FerrumHipKvSlotRequest rq{pending.front().id, PL + pending.front().out_len, 0};
            FerrumHipKvSlotReservation rs;
            if (ferrum_hip_model_reserve_kv_slots(model, &rq, 1, &rs) != 0) break;      // pool full: retry after a retirement
            pending.front().t_submit = now_s();
            running.push_back(std::move(pending.front()));
            pending.pop_front();
        }
        if (running.empty()) {
            std::fprintf(stderr, "no request fits the KV pool\n");
            return 1;
        }
        bool any_prefill = false;
        for (const Request& r : running) any_prefill |= r.prefilled < PL;

        if (!any_prefill) {
            // pure decode: hipGraph replay for as many steps as nothing can change the batch (no retirement before the
            // shortest remaining output is done; admission only follows a retirement)
            int steps = OL + jitter;
            for (const Request& r : running) steps = std::min(steps, r.out_len - r.generated);
            ids.clear(); toks.clear();
            for (const Request& r : running) { ids.push_back(r.id); toks.push_back(r.last_token); }
            const int n = (int)running.size();
            if ((size_t)steps * n > sampled.size()) sampled.resize((size_t)steps * n);
            CHECK(ferrum_hip_model_decode_steps(model, ids.data(), toks.data(), n, steps, sampled.data()));
            for (int s = 0; s < steps; s++)
                for (int i = 0; i < n; i++) {
                    Request& r = running[i];
                    r.last_token = sampled[(size_t)s * n + i];
                    r.generated++;
                    if (dump) r.out.push_back(r.last_token);
                }
            out_tokens += (long)steps * n;
            graph_steps += steps;
            iterations += steps;
        } else {
            // mixed batch under the token budget: decode tokens first (they are latency-sensitive), then prompt chunks
            items.clear();
            std::vector<int> owner, chunk;          // per item: index into `running`, prompt tokens carried (0 = decode token)
            int left = budget;
            for (size_t i = 0; i < running.size(); i++) {
                Request& r = running[i];
                if (r.prefilled == PL) {
                    items.push_back(FerrumHipBatchItem{r.id, &r.last_token, 1, PL + r.generated - 1, 1, 0});
                    owner.push_back((int)i);
                    chunk.push_back(0);
                    left--;
                }
            }
            for (size_t i = 0; i < running.size() && left > 0; i++) {
                Request& r = running[i];
                if (r.prefilled < PL) {
                    const int n = std::min(PL - r.prefilled, left);
                    items.push_back(FerrumHipBatchItem{r.id, r.prompt.data() + r.prefilled, n, r.prefilled,
                                                       r.prefilled + n == PL ? 1 : 0, 0});
                    owner.push_back((int)i);
                    chunk.push_back(n);
                    left -= n;
                }
            }
            CHECK(ferrum_hip_model_unified_forward(model, items.data(), (int)items.size(), 1, sampled.data(), nullptr));
            const double t = now_s();
            int j = 0;      // sampled tokens come back in item order, one per final chunk
            for (size_t k = 0; k < items.size(); k++) {
                Request& r = running[owner[k]];
                r.prefilled += chunk[k];
                if (!items[k].is_final_chunk) continue;
                r.last_token = sampled[j++];
                if (chunk[k]) { r.generated = 1; r.t_first = t; } else { r.generated++; }
                out_tokens++;
                if (dump) r.out.push_back(r.last_token);
            }
            mixed_iterations++;
            iterations++;
        }
        // retirement
        for (size_t i = 0; i < running.size();) {
            if (running[i].generated >= running[i].out_len) {
                CHECK(ferrum_hip_model_release(model, running[i].id));
                finished.push_back(std::move(running[i]));
                running.erase(running.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
    const double wall = now_s() - t0;

    std::vector<double> ttft;
    for (const Request& r : finished) ttft.push_back((r.t_first - r.t_submit) * 1e3);
    std::sort(ttft.begin(), ttft.end());
    FerrumHipKvSlotReservation cap;
    CHECK(ferrum_hip_model_kv_capacity_snapshot(model, &cap));
    std::printf("{\"driver\": \"ferrum_hip_serve (C++ over the C ABI)\", \"model\": \"%s\", \"layers\": %d, \"requests\": %d, "
                "\"concurrency\": %d, \"prompt_len\": %d, \"out_len\": %d, \"max_batched_tokens\": %d, \"out_len_jitter\": %d, \"output_tokens\": %ld, "
                "\"wall_s\": %.4f, \"output_tok_s\": %.1f, \"ttft_ms_p50\": %.2f, \"ttft_ms_p99\": %.2f, \"iterations\": %ld, "
                "\"mixed_iterations\": %ld, \"graph_decode_steps\": %ld, \"kv_blocks_total\": %d, \"kv_blocks_free_at_exit\": %d}\n",
                dense ? "llama31-8b" : "qwen3-30b-a3b", layers, num_requests, conc, PL, OL, budget, jitter, out_tokens, wall,
                (double)out_tokens / wall, ttft[ttft.size() / 2], ttft[std::min(ttft.size() - 1, ttft.size() * 99 / 100)],
                iterations, mixed_iterations, graph_steps, cap.total_blocks, cap.free_blocks_after);
    if (dump) {
        std::sort(finished.begin(), finished.end(), [](const Request& a, const Request& b) { return a.id < b.id; });
        for (const Request& r : finished) {
            std::printf("tokens %llu:", (unsigned long long)r.id);
            for (uint32_t t : r.out) std::printf(" %u", t);
            std::printf("\n");
        }
    }
    CHECK(ferrum_hip_model_destroy(model));
    return cap.free_blocks_after == cap.total_blocks ? 0 : 3;
}
