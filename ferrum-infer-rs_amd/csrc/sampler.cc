// Host-side sampling chain: the C++ mirror of the reference's logits processors and samplers
// (ferrum-interfaces/src/sampler.rs:186-467), for callers that take FullLogits back from the runner.
//
// Semantics kept exactly (they decide token ids):
//  * processors run by priority — repetition penalty (High), top-k and top-p (Normal, in that order), temperature (Low,
//    i.e. LAST: sampler.rs:208-210);
//  * top-k masks logits strictly below the k-th largest (ties at the threshold survive); the order comes from a STABLE
//    descending sort (Rust `sort_by`), comparisons with NaN count as equal;
//  * top-p works on f32 softmax probabilities summed in index order, sorts them stably descending, accumulates in that
//    order and keeps everything up to and including the first element that takes the running sum above p;
//  * the repetition penalty touches every distinct previous token once: v > 0 → v / p, else v · p;
//  * greedy = Iterator::max_by ⇒ the LAST maximum (the device argmax keeps the FIRST, traits.rs:1547 — both exist in
//    the reference and both are reproduced);
//  * multinomial: non-finite logits get probability 0, probabilities are normalised in f32, the threshold is
//    next_u32 / u32::MAX in f32, the first index whose running sum reaches it wins, the last index is the fallback.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <unordered_set>
#include <vector>

#include "../../include/ferrum_hip.h"
#include "common.h"

namespace {

void temperature(float* logits, int n, float t) {
    if (t > 0.0f && t != 1.0f)
        for (int i = 0; i < n; i++) logits[i] /= t;
}

std::vector<int> stable_desc_order(const float* v, int n) {
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return v[x] > v[y]; });   // NaN: neither side greater ⇒ equal
    return idx;
}

void top_k(float* logits, int n, int k) {
    if (k <= 0 || k >= n) return;
    const std::vector<int> idx = stable_desc_order(logits, n);
    const float threshold = logits[idx[k - 1]];
    for (int i = 0; i < n; i++)
        if (logits[i] < threshold) logits[i] = -INFINITY;
}

void top_p(float* logits, int n, float p) {
    if (!(p < 1.0f && p > 0.0f)) return;
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) mx = std::max(mx, logits[i]);          // f32::max semantics for the non-NaN case
    std::vector<float> probs(n);
    for (int i = 0; i < n; i++) probs[i] = expf(logits[i] - mx);
    float sum = 0.0f;
    for (int i = 0; i < n; i++) sum += probs[i];
    for (int i = 0; i < n; i++) probs[i] /= sum;
    const std::vector<int> idx = stable_desc_order(probs.data(), n);
    float cum = 0.0f;
    int cutoff = n;
    for (int i = 0; i < n; i++) {
        cum += probs[idx[i]];
        if (cum > p) { cutoff = i + 1; break; }
    }
    for (int i = cutoff; i < n; i++) logits[idx[i]] = -INFINITY;
}

void repetition_penalty(float* logits, int n, const uint32_t* prev, int n_prev, float penalty) {
    if (penalty == 1.0f) return;
    std::unordered_set<uint32_t> seen;
    for (int i = 0; i < n_prev; i++) {
        const uint32_t id = prev[i];
        if (!seen.insert(id).second || id >= (uint32_t)n) continue;
        const float v = logits[id];
        logits[id] = v > 0.0f ? v / penalty : v * penalty;
    }
}

int greedy_last_max(const float* logits, int n) {
    int best = 0;
    for (int i = 1; i < n; i++)
        if (!(logits[i] < logits[best])) best = i;     // max_by keeps the later element on Equal (and on NaN ⇒ Equal)
    return best;
}

int multinomial(const float* logits, int n, uint32_t random_u32, int* out) {
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) mx = std::max(mx, logits[i]);
    std::vector<float> probs(n);
    float sum = 0.0f;
    for (int i = 0; i < n; i++) {
        probs[i] = std::isfinite(logits[i]) ? expf(logits[i] - mx) : 0.0f;
        sum += probs[i];
    }
    if (!(sum > 0.0f)) { fh::set_error("sampler: no valid tokens for sampling"); return FERRUM_HIP_INVALID; }
    const float threshold = (float)random_u32 / (float)UINT32_MAX;
    float cum = 0.0f;
    for (int i = 0; i < n; i++) {
        cum += probs[i] / sum;
        if (cum >= threshold) { *out = i; return 0; }
    }
    *out = n - 1;
    return 0;
}

}  // namespace

extern "C" {

int ferrum_hip_sampler_apply_temperature(float* logits, int n, float temperature_) {
    FH_REQUIRE(logits && n > 0, "sampler: empty logits");
    temperature(logits, n, temperature_);
    return 0;
}
int ferrum_hip_sampler_apply_top_k(float* logits, int n, int k) {
    FH_REQUIRE(logits && n > 0, "sampler: empty logits");
    top_k(logits, n, k);
    return 0;
}
int ferrum_hip_sampler_apply_top_p(float* logits, int n, float p) {
    FH_REQUIRE(logits && n > 0, "sampler: empty logits");
    top_p(logits, n, p);
    return 0;
}
int ferrum_hip_sampler_apply_repetition_penalty(float* logits, int n, const uint32_t* previous_tokens, int num_previous,
                                                float penalty) {
    FH_REQUIRE(logits && n > 0 && (num_previous == 0 || previous_tokens), "sampler: bad argument");
    repetition_penalty(logits, n, previous_tokens, num_previous, penalty);
    return 0;
}
int ferrum_hip_sampler_greedy(const float* logits, int n, uint32_t* token) {
    FH_REQUIRE(logits && n > 0 && token, "sampler: empty logits for sampling");
    *token = (uint32_t)greedy_last_max(logits, n);
    return 0;
}
int ferrum_hip_sampler_multinomial(const float* logits, int n, uint32_t random_u32, uint32_t* token) {
    FH_REQUIRE(logits && n > 0 && token, "sampler: empty logits for sampling");
    int id = 0;
    if (int rc = multinomial(logits, n, random_u32, &id)) return rc;
    *token = (uint32_t)id;
    return 0;
}

int ferrum_hip_sampler_sample(float* logits, int n, const FerrumHipSamplingParams* p, uint32_t* token) {
    FH_REQUIRE(logits && n > 0 && p && token, "sampler: bad argument");
    // SamplingConfig::sample: processors sorted by priority (High → Low), then the sampler
    repetition_penalty(logits, n, p->previous_tokens, p->num_previous_tokens, p->repetition_penalty);
    if (p->top_k > 0) top_k(logits, n, p->top_k);
    top_p(logits, n, p->top_p);
    temperature(logits, n, p->temperature);
    if (p->greedy) { *token = (uint32_t)greedy_last_max(logits, n); return 0; }
    int id = 0;
    if (int rc = multinomial(logits, n, p->random_u32, &id)) return rc;
    *token = (uint32_t)id;
    return 0;
}

}  // extern "C"
