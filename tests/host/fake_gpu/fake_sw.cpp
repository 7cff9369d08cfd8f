// fake_sw.cpp — TEST INFRASTRUCTURE: the C ABI of include/cudasw4_amd.h on host memory, next to fake_hip.cpp.  It does NOT
// align anything: a subject's "score" is a cheap deterministic function of its letters and the query, the same on every
// device, so that a two-device run of the real C++ driver must reproduce the one-device run bit for bit if (and only
// if) sharding, per-device state, the overflow / re-score plumbing and the host merge are right.  Every pointer a call
// receives must belong to the context's device (fake_hip_owner_of): a buffer of the other device is a counted violation.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../../include/cudasw4_amd.h"
#include "../../../include/cudasw4_amd_engine.h"

extern "C" int fake_hip_owner_of(const void* p);
extern "C" int fake_hip_current_device();

namespace {
std::string g_err;
int g_bad_owner = 0;
long g_scans[2] = {0, 0}, g_rescored[2] = {0, 0}, g_dry[2] = {0, 0};
}

struct sw_ctx {
    int device = 0;
    std::vector<int8_t> query;
    bool have_matrix = false;
    uint32_t* start_signal = nullptr;
    uint32_t* dry_signal = nullptr;
    uint32_t dry_value = 0;
    int grid_reserve = 0;
};

namespace {
void owned(const sw_ctx* c, const void* p) {
    const int o = fake_hip_owner_of(p);
    if (p && o >= 0 && o != c->device) g_bad_owner++;
    if (fake_hip_current_device() != c->device) g_bad_owner++;  // the real library switches itself; the driver must not rely on a stale device either
}
int fake_score(const sw_ctx* c, const int8_t* s, int32_t len) {
    uint32_t h = 2166136261u;
    for (int32_t i = 0; i < len; i++) h = (h ^ uint32_t(uint8_t(s[i]))) * 16777619u;
    for (int8_t q : c->query) h = (h ^ uint32_t(uint8_t(q))) * 16777619u;
    return int(h % 500u);
}
constexpr int kFakeLimit = 450;  // "packed overflow" threshold of the fake
}

// what the real batch engine (cudasw4_amd/csrc/sw_batch.hip, compiled into this fake library as it is) asks of its library
#include "../../../cudasw4_amd/csrc/sw_internal.hpp"
namespace swi {
int fail(int code, const std::string& msg) { g_err = msg; return code; }
int32_t query_length(const sw_ctx* c) { return c ? int32_t(c->query.size()) : 0; }
int device_of(const sw_ctx* c) { return c ? c->device : -1; }
int num_cus(const sw_ctx*) { return 256; }
}  // namespace swi

extern "C" {
int fake_sw_bad_owner() { return g_bad_owner; }

long fake_sw_scans(int device) { return g_scans[device]; }
long fake_sw_rescored(int device) { return g_rescored[device]; }

const char* sw_version(void) { return "fake (CPU test stub, two devices)"; }
const char* sw_last_error(void) { return g_err.c_str(); }
int sw_device_count(void) { return 2; }
int sw_ctx_create(int device, sw_ctx** out) {
    if (device < 0 || device > 1) { g_err = "device index out of range"; return SW_ERR_INVALID; }
    *out = new sw_ctx;
    (*out)->device = device;
    return SW_OK;
}
int sw_ctx_destroy(sw_ctx* c) { delete c; return SW_OK; }
int sw_set_matrix(sw_ctx* c, const int8_t*, int) { c->have_matrix = true; return SW_OK; }
int sw_set_query(sw_ctx* c, const int8_t* q, int32_t qlen, void*) { (void)hipSetDevice(c->device); c->query.assign(q, q + qlen); return SW_OK; }
size_t sw_scan_temp_bytes(sw_ctx*, int, int, int32_t, int32_t) { return 0; }
int sw_set_start_signal(sw_ctx* c, uint32_t* s) { c->start_signal = s; return SW_OK; }
int sw_probe_handshake(sw_ctx*, void*, void*, uint32_t*) { return 1; }
int sw_launch_vgpr_slot(sw_ctx*, int, int, int32_t, int32_t) { return 168; }
int sw_set_rows_pipeline_slot(sw_ctx*, int) { return SW_OK; }
int sw_set_dry_signal(sw_ctx* c, uint32_t* s, uint32_t v) { c->dry_signal = s; c->dry_value = v; return SW_OK; }
int sw_set_dirty_counter(sw_ctx*, int32_t*) { return SW_OK; }
int sw_set_grid_reserve(sw_ctx* c, int32_t n) { c->grid_reserve = n; return SW_OK; }
long fake_sw_dry_signals(int device) { return g_dry[device]; }

int sw_scan_partition(sw_ctx* c, int kind, int, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                      int32_t first_pos, int32_t n, int32_t max_subject_len, int, int, float* scores, int32_t* ids,
                      int64_t id_offset, int32_t* ovf_pos, int32_t* ovf_count, int ovf_check, void*, size_t, void*) {
    (void)hipSetDevice(c->device);
    owned(c, chars); owned(c, offsets); owned(c, lengths); owned(c, scores); owned(c, ids); owned(c, ovf_pos); owned(c, ovf_count);
    uint32_t* sig = c->start_signal;
    c->start_signal = nullptr;
    uint32_t* dry = c->dry_signal;
    c->dry_signal = nullptr;
    if (n == 0) return SW_OK;
    if (sig) (*sig)++;
    if (dry) { owned(c, dry); *dry = c->dry_value; g_dry[c->device]++; }
    g_scans[c->device]++;
    const bool packed = kind == SW_KIND_F16X2 || kind == SW_KIND_I16X2;
    for (int32_t i = 0; i < n; i++) {
        const int32_t pos = first_pos + i;
        if (lengths[pos] > max_subject_len) { g_err = "max_subject_len under-reports"; return SW_ERR_INVALID; }
        const int sc = fake_score(c, chars + (offsets[pos] - offsets[0]), lengths[pos]);
        ids[pos] = int32_t(id_offset + pos);
        if (packed && ovf_check && sc >= kFakeLimit) ovf_pos[(*ovf_count)++] = pos;
        else scores[pos] = float(sc);
    }
    return SW_OK;
}
int sw_set_long16_min(sw_ctx*, int32_t) { return SW_OK; }
size_t sw_scan_rows_pipelined_temp_bytes(sw_ctx*, int32_t n, int32_t max_subject_len) {
    return n > 0 ? size_t(n) * size_t((max_subject_len + 511) / 512) * 64 : 0;
}
int sw_scan_rows_pipelined(sw_ctx* c, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t first_pos, int32_t n,
                           int32_t max_subject_len, int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, int32_t* fail_count,
                           int32_t* over1, int32_t* over2, int32_t packed_limit, void* temp, size_t temp_bytes, void* stream) {
    owned(c, fail_count); owned(c, over1); owned(c, over2);
    if (n > 0) { owned(c, temp); if (temp_bytes < sw_scan_rows_pipelined_temp_bytes(c, n, max_subject_len)) return SW_ERR_TEMP; }
    const int rc = sw_scan_partition(c, SW_KIND_I32, 35, chars, offsets, lengths, first_pos, n, max_subject_len, gop, gex, scores, ids,
                                     id_offset, nullptr, nullptr, 0, nullptr, 0, stream);
    // what a packed launch of the fake would have flagged (kFakeLimit stands for the packed kind's limit here)
    if (rc == SW_OK && packed_limit > 0)
        for (int32_t i = 0; i < n; i++)
            if (scores[first_pos + i] >= float(kFakeLimit)) { if (over1) (*over1)++; if (over2) (*over2)++; }
    return rc;
}
int sw_rescore_overflow_stat(sw_ctx* c, int, const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                             const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t, int, int,
                             float* scores, int32_t* ids, int64_t id_offset, void*, size_t, int32_t,
                             int32_t* true_overflow_count, void*) {
    (void)hipSetDevice(c->device);
    owned(c, ovf_pos); owned(c, ovf_count); owned(c, chars); owned(c, scores); owned(c, true_overflow_count);
    c->start_signal = nullptr;
    const int32_t cnt = std::min(*ovf_count, max_count);
    for (int32_t i = 0; i < cnt; i++) {
        const int32_t pos = ovf_pos[i];
        scores[pos] = float(fake_score(c, chars + (offsets[pos] - offsets[0]), lengths[pos]));
        ids[pos] = int32_t(id_offset + pos);
        if (true_overflow_count) (*true_overflow_count)++;
        g_rescored[c->device]++;
    }
    return SW_OK;
}
int sw_rescore_overflow(sw_ctx* c, int kind, const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                        const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t m, int gop, int gex,
                        float* scores, int32_t* ids, int64_t id_offset, void* t, size_t tb, void* s) {
    return sw_rescore_overflow_stat(c, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, m, gop, gex, scores, ids,
                                    id_offset, t, tb, 0, nullptr, s);
}
size_t sw_topk_temp_bytes(int64_t, int) { return 16; }
int sw_topk(sw_ctx* c, const float* scores, const int32_t* ids, int64_t n, int k, float* out_s, int32_t* out_i, void*, size_t, void*) {
    (void)hipSetDevice(c->device);
    owned(c, scores); owned(c, ids); owned(c, out_s); owned(c, out_i);
    std::vector<int64_t> order(size_t(n), 0);
    for (int64_t i = 0; i < n; i++) order[size_t(i)] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return scores[a] > scores[b]; });
    for (int i = 0; i < k; i++) {
        out_s[i] = i < n ? scores[order[size_t(i)]] : -1.0f;
        out_i[i] = i < n ? ids[order[size_t(i)]] : -1;
    }
    return SW_OK;
}
int sw_check_letter_codes(sw_ctx* c, const int8_t* chars, size_t n, int32_t* bad, void*) {
    owned(c, chars); owned(c, bad);
    for (size_t i = 0; i < n; i++) if (uint8_t(chars[i]) > 20) *bad = 1;
    return SW_OK;
}
int sw_plan_launch(sw_ctx*, int kind, int part_id, int32_t, int32_t, int32_t* ek, int32_t* r, int32_t* ns, int32_t* lanes) {
    if (ek) *ek = kind;
    if (r) *r = 8;
    if (ns) *ns = 1;
    if (lanes) *lanes = part_id >= 34 ? 64 : 16;
    return SW_OK;
}
size_t sw_rescore_service_temp_bytes(sw_ctx*, int, int32_t, int) { return 0; }
int sw_rescore_service(sw_ctx* c, int, int32_t*, const int32_t*, int32_t, const int8_t*, const uint64_t*, const int32_t*, int32_t, int, int,
                       float*, int32_t*, int64_t, void*, size_t, int32_t, int32_t*, const uint32_t*, uint32_t, int, void*) {
    if (c->start_signal) (*c->start_signal)++;   // its workgroups are "resident"
    c->start_signal = nullptr;
    return SW_OK;   // the fake executes at enqueue time: a service started before its producer finds an empty list and leaves
}
int sw_rescore_overflow_claim(sw_ctx* c, int kind, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count, const int8_t* chars,
                              const uint64_t* offsets, const int32_t* lengths, int32_t m, int gop, int gex, float* scores, int32_t* ids,
                              int64_t id_offset, void* t, size_t tb, int32_t lim, int32_t* cnt, void* s) {
    return sw_rescore_overflow_stat(c, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, m, gop, gex, scores, ids, id_offset, t, tb, lim, cnt, s);
}
size_t sw_rescore_overflow_pipelined_temp_bytes(sw_ctx*, int32_t) { return 64; }
int sw_rescore_overflow_pipelined(sw_ctx* c, int32_t* ovf_pos, const int32_t* ovf_count, int32_t, const int8_t*, const uint64_t*, const int32_t*, int32_t,
                                  int32_t, int, int, float*, int32_t*, int64_t, int32_t* fail_count, int32_t, int32_t* cnt, void* temp, size_t, void*) {
    owned(c, ovf_pos); owned(c, ovf_count); owned(c, fail_count); owned(c, cnt); owned(c, temp);
    return SW_OK;   // (the fake leaves every entry to the claim launch behind)
}
int sw_streams_run_concurrently(sw_ctx*, void*, void*) { return 1; }
int32_t sw_window_overlap(sw_ctx*, int, int) { return -1; }   // the fake's scores are no alignment scores: never cut
int sw_reduce_windows(sw_ctx*, const float*, const int32_t*, const int32_t*, int32_t, float*, int32_t*, int64_t, void*) { return SW_OK; }
int sw_plan_query(int, int32_t, int32_t* r, int32_t* ns) { if (r) *r = 8; if (ns) *ns = 1; return SW_OK; }
}
