// sw_api.hip — the C ABI declared in include/cudasw4_amd.h.
//
// Host-side glue only: context state (matrix, query, per-kind profile), query planning (rows per
// lane x stripes), launch-grid / scratch sizing, overflow re-score and top-K.  There is NO CPU
// fallback anywhere: without a HIP device every entry point fails with SW_ERR_NO_DEVICE/SW_ERR_HIP.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/cudasw4_amd.h"
#include "../../include/cudasw4_amd_engine.h"
#include "sw_internal.hpp"
#include "sw_launch.hpp"
#include "sw_rows_pipeline.hpp"

namespace {

thread_local std::string g_last_error = "";

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define SW_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(SW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));          \
    } while (0)

const swk::KindLaunch* kind_launch(int kind) {
    switch (kind) {
        case SW_KIND_F16X2: return &swk::launch_f16x2();
        case SW_KIND_I16X2: return &swk::launch_i16x2();
        case SW_KIND_I32: return &swk::launch_i32();
        case SW_KIND_F32: return &swk::launch_f32();
    }
    return nullptr;
}

bool kind_packed(int kind) { return kind == SW_KIND_F16X2 || kind == SW_KIND_I16X2; }

struct QueryPlan {
    int rows = 0;      // R: query rows per lane
    int nstripes = 0;  // stripes of lanes*R rows
    int lanes = 16;    // lanes per alignment group: 16 (DPP row), 64 (whole wave, long subjects) or 8 (half row, short queries)
};

// Pick (R, nstripes) with R a compiled value.  Replaces the reference's subject-length ->
// (group_size, numRegs) table (cudasw4.cuh:1764-1912); here the tile shape follows the QUERY length
// because the query is the register-resident dimension, and the subject length only selects the group
// width: 16 lanes for the bulk of a DB, the whole wave for the long partitions 34/35.
// Cost model: a step of a stripe with R rows per lane issues ~(R + 1.2) row-equivalents of VALU work
// (per-step DPP/address overhead) and every stripe pays its own pipeline fill; minimise ns * (R + 1.2).
QueryPlan plan_query(int kind, int32_t qlen, int lanes) {
    const swk::KindLaunch* kl = kind_launch(kind);
    QueryPlan pl;
    pl.lanes = lanes;
    if (!kl || qlen <= 0) return pl;
    const int maxrows = swk::max_rows(kind, lanes);
    // fp32 multi-stripe kernels above 32 rows per lane lose the third wave per SIMD (their single-stripe kernels keep it up to 36)
    const int maxrows_multi = (kind == SW_KIND_F32 && lanes <= 16) ? std::min(maxrows, swk::kMaxRowsScalarMulti) : maxrows;
    // packed kinds: a multi-stripe kernel above kWaves3MaxRowsPackedMulti rows per lane holds two waves per SIMD, which
    // fill 93.5 % of the issue slots at best (three: 97+ %).  Measured on the peak DB, the same queries with 32-row stripes
    // at three waves against 40..47-row stripes at two (tools/variant_bench2.sh): a step-row of the tall kernels costs
    // 5 % more (fitted together with the per-step overhead below on ten queries of 1500..5478 residues).  CUDASW4_AMD_TWO_WAVE_PENALTY overrides the factor (1: the planner of rounds 1-3; tests use it to reach
    // every compiled shape).
    double two_wave_penalty = 1.05;
    if (const char* e = getenv("CUDASW4_AMD_TWO_WAVE_PENALTY")) two_wave_penalty = std::max(1.0, atof(e));
    const bool has_three_wave_multi = kl->packed && lanes <= 16 && swk::kWaves3MaxRowsPackedMulti > maxrows / 2;
    double best = 1e300;
    auto consider = [&](int ns) {
        const int64_t per_lane = (qlen + (int64_t)lanes * ns - 1) / ((int64_t)lanes * ns);
        int r = (int)((per_lane + swk::kRowsGranule - 1) / swk::kRowsGranule * swk::kRowsGranule);
        r = std::max(r, swk::kRowsGranule);
        if (r > (ns > 1 ? maxrows_multi : maxrows)) return;
        if (ns > 1 && 2 * r <= maxrows) return;  // multi-stripe kernels exist for R > max/2 only
        // per-step overhead in row equivalents: ~10 VALU ops single-stripe; with several stripes ~20 with the stripe border,
        // plus the tile load and the two barriers of every stripe switch spread over the stripe's steps
        double cost = ns * (r + (ns > 1 ? 3.0 : 1.2));
        if (has_three_wave_multi && ns > 1 && r > swk::kWaves3MaxRowsPackedMulti) cost *= two_wave_penalty;
        if (cost < best - 1e-9) { best = cost; pl.rows = r; pl.nstripes = ns; }
    };
    if (qlen <= (int64_t)lanes * maxrows) consider(1);
    const int64_t stripe_multi = (int64_t)lanes * maxrows_multi;
    const int ns_m = std::max<int>(2, (int)((qlen + stripe_multi - 1) / stripe_multi));
    int ns_hi = ns_m + 2;
    if (has_three_wave_multi) {  // ... and the plans on three-wave stripes
        const int64_t stripe3 = (int64_t)lanes * swk::kWaves3MaxRowsPackedMulti;
        ns_hi = std::max<int>(ns_hi, (int)((qlen + stripe3 - 1) / stripe3) + 1);
    }
    for (int ns = ns_m; ns <= ns_hi; ns++) consider(ns);
    return pl;
}

struct Profile {
    unsigned char* dev = nullptr;
    size_t capacity = 0;
    bool valid = false;
    QueryPlan plan;
    int shift = 0;               // added to every substitution score (a = -gex for the column-offset kernels, else 0)
    hipEvent_t ready = nullptr;  // recorded after the build; scans on other streams wait on it
};

}  // namespace

constexpr uint32_t kWorkSlots = 4096;
// Longest query that runs on 8-lane groups.  Tuned on RAGGED subjects (tools/ragged_query_sweep.py, Swiss-Prot-like DB
// through the C++ driver): the packed kinds gain 2.4 ... 7.3 % up to 256 residues (R = 32: still three waves per SIMD)
// and nothing consistent above (272: -1 %, 288: +1.5 %, 304: -3 %); the 32-bit kinds gain 4 ... 14 % up to 256 (R = 32:
// the last height at which their single-stripe kernels keep three waves per SIMD without spills) and lose from 272 on,
// so both limits are 256 (the 32-bit limit stood at 240 until round 4, below what the sweep supports).  (Round 2 tuned on the peak DB, whose identical subjects hid the LDS bank conflicts of the second group
// of a DPP row — fixed since, Geometry / dp_step: laneStep — and chose 288 / 240: +14 % at 144 residues there.)
#ifndef SW_LANES8_MAX_QUERY_PACKED
#define SW_LANES8_MAX_QUERY_PACKED 256
#endif
#ifndef SW_LANES4_MAX_QUERY_PACKED
#define SW_LANES4_MAX_QUERY_PACKED 96
#endif
#ifndef SW_LANES4_MAX_QUERY_SCALAR
#define SW_LANES4_MAX_QUERY_SCALAR 128
#endif
#ifndef SW_LANES4_MAX_SUBJECT
#define SW_LANES4_MAX_SUBJECT 1280
#endif
#ifndef SW_LANES4_MIN_BATCHES_PER_CU
#define SW_LANES4_MIN_BATCHES_PER_CU 12
#endif
#ifndef SW_LANES8_MAX_SUBJECT
#define SW_LANES8_MAX_SUBJECT 192
#endif
#ifndef SW_LANES8_MAX_QUERY_SCALAR
#define SW_LANES8_MAX_QUERY_SCALAR 256
#endif

struct sw_ctx {
    int device = 0;
    int num_cus = 0;
    int8_t* d_matrix = nullptr;  // (dim + 1) x 21: one row per query letter + the padding row, 21 subject letters
    int dim = swk::kLetters;     // what sw_set_matrix was given: query codes are 0..dim-1
    uint32_t* d_zeros = nullptr; // per kind 64 bytes of its zero pattern (first-stripe border): [kind * 16 words]
    uint32_t* d_work = nullptr;  // kWorkSlots pairs (batch counter of the dynamic batch distribution, started workgroups), one per launch in flight
    uint32_t work_next = 0;
    uint32_t work_zeroed = 0;    // slots from work_next on that sw_set_query has zeroed already (one memset per query instead of one per launch)
    uint32_t* start_signal = nullptr;  // sw_set_start_signal: one-shot, consumed by the next scan / re-score launch
    uint32_t* dry_signal = nullptr;    // sw_set_dry_signal: one-shot as well
    int32_t* dirty_counter = nullptr;  // sw_set_dirty_counter (sticky)
    uint32_t dry_value = 0;
    int grid_reserve = 0;              // sw_set_grid_reserve: workgroup slots a scan launch leaves free (until changed)
    int grid_mult = 4;           // persistent workgroups per CU
    int grid_cap = 0;            // CUDASW4_AMD_GRID_CAP (tests): most workgroups of a scan launch — small DBs then give long claims (sw_stream_kernel.hpp)
    bool have_matrix = false;
    int8_t* d_query = nullptr;
    size_t query_capacity = 0;
    // sw_set_query stages the query through a small ring of pinned host buffers, so that the upload needs no host
    // synchronisation (the caller's buffer is free again when the call returns, the copy runs stream-ordered)
    static constexpr int kQuerySlots = 4;
    int8_t* h_query[kQuerySlots] = {};
    size_t h_query_capacity[kQuerySlots] = {};
    hipEvent_t query_copied[kQuerySlots] = {};
    bool query_slot_used[kQuerySlots] = {};
    int query_next = 0;
    int32_t qlen = 0;
    bool have_query = false;
    Profile profiles[4][4][2];  // [kind][shape: 0 = 16-lane groups, 1 = 64-lane groups, 2 = 8-lane groups, 3 = 4-lane groups][plain | column-offset recurrence]
    bool use_offs = true;        // CUDASW4_AMD_NO_OFFS=1: always the plain recurrence (A/B measurements)
    int64_t long16_min = -1;     // sw_set_long16_min: partition 34 gets 16-lane groups from this many subjects up (-1: 512)
    int64_t long16_min_default = -1;  // the built-in rule (sw_set_long16_min(ctx, -1) returns to it)
    int matrix_max = 1;          // largest substitution score of the installed matrix
    bool i32_native = false;     // CUDASW4_AMD_I32_NATIVE=1: never compute the int32 kind in fp32 lanes (tests of the int32 kernels)
    int32_t lanes4_max_subject = -1;  // CUDASW4_AMD_LANES4_MAX_SUBJECT: ... when no subject of the launch is longer (-1: 1280)
    int32_t lanes4_max_q = -1;   // CUDASW4_AMD_LANES4_MAX_Q: queries up to this length use 4-lane groups (0: never; -1: the built-in limits)
    int32_t lanes8_max_q = -1;   // CUDASW4_AMD_LANES8_MAX_Q: queries up to this length use 8-lane groups (0: never; -1: the built-in limits)
    bool check_bounds = false;   // CUDASW4_AMD_CHECK_BOUNDS=1 (debug): every scan first verifies the max_subject_len contract on the device (synchronises)
    uint32_t pipe_spin_limit = 1u << 20;  // CUDASW4_AMD_PIPE_SPIN_LIMIT: polls (~2 us each) before a pipeline stage gives up waiting
    int32_t pipe_drop_stage = -1;         // CUDASW4_AMD_PIPE_TEST_DROP_STAGE (tests): this stage of every subject is lost
    int32_t pipe_cpl = 0;                 // CUDASW4_AMD_PIPE_CPL=4|8|16: columns per lane of a stage (0: by the subjects' length)
    int32_t stream_slots = 4;   // CUDASW4_AMD_STREAM=0..16: most batches whose subjects stream through the lanes back to back (sw_stream_kernel.hpp; 0 / 1: sw_scan_kernel, one batch at a time)
    int32_t stream_jump = 0;              // what the zero levels rise by at a slot border (0: 512 for fp16, 2048 for int16)
    int32_t stream_cols_max = 4096;       // most columns of a round of several slots
    int32_t stream_multi_cols_max = 1536; // ... of a multi-stripe query (its border arrays grow with the round)
    int32_t stream_multi_max_subject = -1;    // multi-stripe queries stream subjects up to this length (-1: 320 for fp16, 192 for int16)
    int32_t pipe_quorum = 0;              // (0: all tickets)
    int32_t pipe_slot = 0;                // sw_set_rows_pipeline_slot: VGPRs a stage occupies (128 / 168 / 256; 0: what it needs)
};

namespace {

int max_grid(const sw_ctx* ctx) {
    const int g = std::max(1, ctx->num_cus) * ctx->grid_mult;
    return ctx->grid_cap > 0 ? std::min(g, ctx->grid_cap) : g;
}

// Reference partitions 34 (1281..8000) and 35 (> 8000) hold the long subjects.  When there are only a few
// of them (the tail of a real DB) they get the wave-wide group shape: 4x the lanes per alignment, so the
// giants finish 4x sooner.  When the partition alone can fill the GPU several times (e.g. the L=2048
// peak DB) the 16-lane shape is more efficient (more rows per lane, less per-step overhead, wide profile words).
// Partition 34 takes the 16-lane shape already from 512 subjects up: in a Swiss-Prot-like DB it holds 2.4 % of the
// sequences but 11 % of the residues, its launch runs next to the bulk launch on a side stream, and a subject of at
// most 8000 residues is no tail there (Swiss-Prot-like DB: 10.7 -> 11.05 TCUPS).
// overflowed subjects can have any length: long ones would dominate a 16-lane launch
int rescore_lanes(int32_t max_subject_len) { return max_subject_len > 1280 ? 64 : 16; }

int shape_index(int lanes) { return lanes == 64 ? 1 : lanes == 8 ? 2 : lanes == 4 ? 3 : 0; }

// The int32 kind in fp32 lanes.  On gfx950 the fp32 form of the recurrence runs 30 % faster than the int32 form (8.35
// against 6.44 TCUPS on the peak DB): v_add_f32 co-issues with v_max3_f32, v_add_u32 does not.  fp32
// arithmetic on integers is exact below 2^24, a score cannot exceed min(query, subject) * (largest substitution score),
// and the column-offset frame raises values by at most 2^22 (scan_common: `room`) — so whenever that bound stays below
// 2^24 the int32 launch is served by the fp32 kernels with bit-identical results (they are written as floats either way,
// like the reference's BatchResultList).  Anything beyond the bound, e.g. a 2-million-residue query against a subject
// of the same size, gets the true int32 kernels.
int effective_kind_of(const sw_ctx* ctx, int kind, int32_t max_subject_len) {
    if (kind != SW_KIND_I32 || ctx->i32_native || !ctx->have_query) return kind;
    const int64_t bound = (int64_t)std::min(ctx->qlen, max_subject_len) * std::max(1, ctx->matrix_max) + ((int64_t)1 << 22) + 4096;
    return bound < ((int64_t)1 << 24) ? SW_KIND_F32 : SW_KIND_I32;
}

// Short queries run on 8-lane groups (half DPP rows): twice the rows per lane for the same query, so the per-step
// overhead is spread over twice the cells, and 7 instead of 15 fill steps per subject (sw_dp_kernel.hpp: Shift).
int lanes_for_partition_any(const sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len);
int lanes_for_partition(const sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len) {
    const int lanes = lanes_for_partition_any(ctx, kind, part_id, n, max_subject_len);
    // the streamed packed kernels (16-lane groups) keep a slot's lengths in 16 bits: subjects beyond that — thousands of
    // them in one packed launch, or a caller that forces the shape — take the wave-wide groups
    return (lanes == 16 && kind_packed(kind) && max_subject_len >= 0xffff) ? 64 : lanes;
}
int lanes_for_partition_any(const sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len) {
    if (part_id < SW_NUM_LENGTH_PARTITIONS - 2) {
        if (!ctx->have_query) return 16;
        // very short queries: 4-lane groups (a DPP quad; single-stripe kernels only).  A batch is 128 (packed kinds) or 64
        // subjects then: only where the launch still hands every workgroup slot several batches
        const int32_t limit4 = ctx->lanes4_max_q >= 0 ? ctx->lanes4_max_q : kind_packed(kind) ? SW_LANES4_MAX_QUERY_PACKED : SW_LANES4_MAX_QUERY_SCALAR;
        const int64_t batches4 = (int64_t)n / (64 * (kind_packed(kind) ? 2 : 1));
        const bool enough4 = ctx->lanes4_max_q >= 0 || batches4 >= (int64_t)SW_LANES4_MIN_BATCHES_PER_CU * std::max(1, ctx->num_cus);   // (an explicit limit: tests reach the shape with a handful of subjects)
        // ... and holds no long subject: a column costs a quad ~(6.5 R + 19) instructions with R = a quarter of the query, and
        // the longest subject's walk bounds the launch (partition 34 merged into a 96-residue query's launch: 3.6 ms
        // against 1.8 ms of instructions); the host driver keeps partition 34 out of such a launch
        const int32_t lmax4 = ctx->lanes4_max_subject >= 0 ? ctx->lanes4_max_subject : SW_LANES4_MAX_SUBJECT;
        if (ctx->qlen <= 4 * swk::max_rows(kind, 4) && ctx->qlen <= limit4 && enough4 && max_subject_len <= lmax4) return 4;
        const bool fits8 = ctx->qlen <= 8 * swk::max_rows(kind, 8);  // one stripe of 8-lane groups
        const int32_t limit = ctx->lanes8_max_q >= 0 ? ctx->lanes8_max_q : kind_packed(kind) ? SW_LANES8_MAX_QUERY_PACKED : SW_LANES8_MAX_QUERY_SCALAR;
        if (fits8 && ctx->qlen <= limit) return 8;
        // (rounds 2-5 ran multi-stripe queries on SHORT subjects — up to 192 residues — on 8-lane groups: twice the stripes, each
        // filling its pipeline in 7 instead of 15 steps, +2.5 % at L = 128.  The streamed 16-lane kernels pay the fill once per
        // round of up to 16 subject pairs: 10 931 against 10 537 GCUPS for the 1000-residue query at L = 128.)
        return 16;
    }
    const int64_t fills_gpu_twice = (int64_t)2 * std::max(1, ctx->num_cus) * 4 * 32;
    if (part_id == SW_NUM_LENGTH_PARTITIONS - 2) return n >= (ctx->long16_min >= 0 ? ctx->long16_min : 512) ? 16 : 64;
    return n >= fills_gpu_twice ? 16 : 64;
}


int ensure_profile(sw_ctx* ctx, int kind, int lanes, bool offs, int shift, hipStream_t stream) {
    Profile& pr = ctx->profiles[kind][shape_index(lanes)][offs];
    if (pr.valid && pr.shift != shift) pr.valid = false;  // other gap-extension score than last time
    if (pr.valid) {
        // built on another stream earlier in this query: order this stream after the build
        SW_HIP(hipStreamWaitEvent(stream, pr.ready, 0));
        return SW_OK;
    }
    const swk::KindLaunch* kl = kind_launch(kind);
    const QueryPlan pl = plan_query(kind, ctx->qlen, lanes);
    const size_t bytes = kl->tile_bytes(pl.rows, lanes) * (size_t)pl.nstripes;
    if (bytes == 0) return fail(SW_ERR_INVALID, "no kernel compiled for this query plan");
    if (bytes > pr.capacity) {
        // hipFree/hipMalloc synchronise the device: they would serialise launches that are meant to overlap
        // on other streams, so capacity starts at 2 MiB (queries up to ~50 k residues) and doubles
        const size_t cap = std::max<size_t>(std::max(bytes, 2 * pr.capacity), size_t(2) << 20);
        if (pr.dev) SW_HIP(hipFree(pr.dev));
        pr.dev = nullptr;
        pr.capacity = 0;
        SW_HIP(hipMalloc(&pr.dev, cap));
        pr.capacity = cap;
    }
    SW_HIP(kl->profile(pl.rows, lanes, ctx->d_query, ctx->qlen, ctx->d_matrix, ctx->dim, pl.nstripes, pr.dev, shift, stream));
    pr.shift = shift;
    if (!pr.ready) SW_HIP(hipEventCreateWithFlags(&pr.ready, hipEventDisableTiming));
    SW_HIP(hipEventRecord(pr.ready, stream));
    pr.plan = pl;
    pr.valid = true;
    return SW_OK;
}

// scratch words per (workgroup, group) array for subjects up to max_len
int32_t border_capacity(int32_t max_len, int lanes) {
    const int64_t steps = ((int64_t)max_len + lanes - 1 + 3) / 4 * 4;
    return (int32_t)((steps + 15) / 16 * 16 + 16);  // + one prefetched quad past the end, rounded to 64 bytes
}
size_t border_bytes_per_wg(int32_t lcap, int lanes) {
    if (lanes == 4) return 0;   // single-stripe kernels only: no border
    const size_t region = lanes == 16 ? swk::border_region_words<16>(lcap) : lanes == 8 ? swk::border_region_words<8>(lcap) : swk::border_region_words<64>(lcap);
    return (size_t)(swk::kThreads / lanes) * region * sizeof(uint32_t);
}

// Debug check of the max_subject_len contract (include/cudasw4_amd.h): longest subject of the range / of the listed positions
__global__ void max_length_kernel(const int32_t* lengths, const int32_t* positions, const int32_t* count_ptr, int32_t first_pos,
                                  int32_t n, int32_t* out) {
    const int32_t cnt = count_ptr ? min(*count_ptr, n) : n;
    int32_t m = 0;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x)
        m = max(m, lengths[positions ? positions[i] : first_pos + i]);
    if (m > 0) atomicMax(out, m);
}

// Streamed subjects (sw_stream_kernel.hpp).  The levels of a round start at `base` and rise by a per column and by `jump` per
// slot border; a subject is exact while level + score stays inside the kind's exact range, so the round's a * columns + jumps
// may add up to `room`: what is left between the base and the highest level at which unrelated subjects (and moderate hits)
// still pass unflagged.
//   fp16: exact integers in [-2048, 2048]: base -2016, levels up to 1024 (a subject in the last columns of a round is flagged
//         from 2048 - 1024 on — round 5's frame flagged from 988 on everywhere), jump 128;
//   int16: biased patterns exact up to 31 743, limit 25 000: base 0, levels up to 12 400, jump 512.
// A slot whose predecessor scored jump - 4 or more is re-scored (its lanes may have kept values above the raised levels):
// the jump is what separates "unrelated" from "a hit" — 128 is far above the noise floor of a 35 000-residue subject (~90).
struct StreamPlan { int slots = 0, cols = 0, room = 0, base = 0, jump = 0; };
StreamPlan stream_plan(const sw_ctx* ctx, int kind, int lanes, int a, const QueryPlan& pl, int32_t max_subject_len) {
    StreamPlan sp;
    if (!kind_packed(kind) || lanes != 16 || a < 0) return sp;   // (a < 0: offs_possible sends the launch to the 32-bit kind)
    const bool multi = pl.nstripes > 1;
    const int P = swk::frame_classes(true, pl.rows, lanes, multi);
    sp.base = kind == SW_KIND_F16X2 ? -2016 : 0;
    const int top = kind == SW_KIND_F16X2 ? 1024 : 12400;
    sp.jump = ctx->stream_jump > 0 ? ctx->stream_jump : (kind == SW_KIND_F16X2 ? 512 : 2048);
    sp.room = top - sp.base - a * (2 * lanes + 4 + P);
    sp.cols = a > 0 ? std::min(sp.room / a, ctx->stream_cols_max) : ctx->stream_cols_max;   // (a = 0: the levels do not grow)
    sp.slots = std::max(1, std::min(ctx->stream_slots, swk::kStreamMaxSlots));
    // a gap-extension score so large that not even two short slots fit a round: one slot at a time
    if (sp.room < 4 * lanes * a || sp.cols < 4 * lanes) sp.slots = 1;
    // Several stripes: a slot border costs the streamed kernels about seven step equivalents per stripe (the separator column,
    // sixteen steps of a second copy of the loop body, two transitions between the copies' register allocations) against the
    // sixteen of a pipeline drain and the rounding to whole quads — worth it while the subjects are short (measured, peak DB,
    // 1000- / 5478-residue queries: L = 128 +3.7 / +1.1 % over round 5's 8-lane groups, L = 256 +0.7 / +0.6 %, L = 512
    // +0.1 / -1.5 %; int16, whose rounds are longer: L = 256 -0.1 / -3.8 %, L = 512 -0.9 / -5.5 %), and with rounds whose
    // border arrays stay small.  Longer subjects: sw_scan_kernel, one pair at a time (slots = 1 selects it, sw_launch.hpp).
    if (multi) {
        // (measured with rounds of up to 1 536 columns, against sw_scan_kernel on 16-lane groups: fp16 L = 128 +4.5 / +2.8 %,
        // L = 256 +3.0 / +0.7 %, L = 512 +0.7 / -0.1 %; int16 L = 128 +4.2 / +2.0 %, L = 256 -0.4 / -0.4 %, L = 512 -0.7 / -1.6 %:
        // profiles/r06_stream_ab.txt)
        const int32_t limit = ctx->stream_multi_max_subject >= 0 ? ctx->stream_multi_max_subject : (kind == SW_KIND_F16X2 ? 320 : 192);
        if (max_subject_len > limit) sp.slots = 1;
        sp.cols = std::min(sp.cols, ctx->stream_multi_cols_max);
    }
    return sp;
}

// The column-offset recurrence needs (i) a frame period K of at least 4 * lanes columns inside the kind's room — the
// longest run of columns in one frame plus the pipeline skew: a * (K + 3 * lanes + 16) <= room —, (ii) for a packed kind an
// overflow list to flag into (a subject whose bound score + a * columns reaches the limit is flagged early), (iii) gop - gex
// inside the 16-bit encodings.  *K_out: the period (0: the frame is never lowered).
bool offs_possible(const sw_ctx* ctx, int kind, int lanes, int32_t max_subject_len, int gop, int gex, int ovf_check, int64_t* K_out) {
    const int a = -gex;
    const int64_t room = kind == SW_KIND_F16X2 ? 1536 : kind == SW_KIND_I16X2 ? 12500 : (int64_t)1 << 22;
    int64_t K = 1 << 21;
    while (K >= 4 * lanes && (int64_t)a * (K + 3 * lanes + 16) > room) K >>= 1;
    // int16: the room would allow a period of 8192 columns, but the flag bound grows with the period (score + a * min(columns,
    // K) + ...): with 8192 a 4000-residue protein was re-scored from 21 000 up instead of 25 000.  2048 columns cost 0.3 % in
    // lowering steps and keep the early flags within 8 % of the limit.
    if (kind == SW_KIND_I16X2) K = std::min<int64_t>(K, std::max<int64_t>(2048, 4 * lanes));
    if (!kind_packed(kind) && max_subject_len + 3 * lanes + 16 > K) K = 0;  // the 32-bit kernels do not lower their frame
    if (K_out) *K_out = K;
    return ctx->use_offs && K >= 4 * lanes && (!kind_packed(kind) || (ovf_check && a >= 0)) && gop - gex >= -1000;
}
// A packed launch that cannot run the column-offset recurrence is served by its 32-bit kind (exact, nothing to flag): the
// plain form of the packed recurrence is no longer compiled (round 6: it was half of the packed code objects for gap scores
// nobody uses — |gex| > 12 with fp16 — and for callers that pass no overflow list).  CUDASW4_AMD_NO_OFFS=1 takes every
// packed launch there (tests of the fallback).
int packed_fallback_kind(int kind) { return kind == SW_KIND_F16X2 ? SW_KIND_F32 : SW_KIND_I32; }

// A launch's pair of control words (batch counter of the dynamic batch distribution, counted-in workgroups) from the
// context's ring.  sw_set_query zeroes the next kWorkZeroBlock pairs with ONE memset on the query's stream — every launch of
// the query is ordered behind that stream position anyway (it reads the query) —, so the launches of a query take their
// words without a memset of their own (round 5: 13 hipMemsetAsync per query on a small shard); past the block, or without
// sw_set_query in between, a launch zeroes its own as before.
constexpr uint32_t kWorkZeroBlock = 64;
hipError_t take_work_slot(sw_ctx* ctx, hipStream_t stream, uint32_t** out) {
    uint32_t* w = ctx->d_work + 2 * (ctx->work_next++ % kWorkSlots);
    *out = w;
    if (ctx->work_zeroed > 0) { ctx->work_zeroed--; return hipSuccess; }
    return hipMemsetAsync(w, 0, 2 * sizeof(uint32_t), stream);
}

// overflow lists that are re-scored while they are filled (sw_dp_kernel.hpp: ScanParams::claim / service)
struct ListMode {
    int32_t* claim = nullptr;          // the list, writable: entries are taken by compare-and-swap
    int service_workgroups = 0;        // > 0: a service launch of this many workgroups
    const uint32_t* done_flag = nullptr;
    uint32_t done_value = 0;
};

int scan_common(sw_ctx* ctx, int kind, int lanes, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths,
                const int32_t* positions, const int32_t* count_ptr, int32_t first_pos, int32_t n,
                int32_t max_subject_len, int gop, int gex, float* scores, int32_t* ids, int64_t id_offset,
                int32_t* ovf_pos, int32_t* ovf_count, int ovf_check, void* temp, size_t temp_bytes,
                hipStream_t stream, int32_t* stat_count = nullptr, int32_t stat_limit = 0, const ListMode& list = ListMode()) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    // one-shot: whatever happens to this call, the pending start signal belongs to it (an error or an empty launch
    // cancels it: nothing will fire, the caller must not wait)
    uint32_t* const start_signal = ctx->start_signal;
    ctx->start_signal = nullptr;
    uint32_t* const dry_signal = ctx->dry_signal;
    ctx->dry_signal = nullptr;
    const swk::KindLaunch* kl = kind_launch(kind);
    if (!kl) return fail(SW_ERR_INVALID, "unknown kind");
    if (n < 0 || max_subject_len < 0) return fail(SW_ERR_INVALID, "negative count or length");
    if (max_subject_len > SW_MAX_SUBJECT_LEN)
        return fail(SW_ERR_INVALID, "max_subject_len " + std::to_string(max_subject_len) + ": pass the longest subject of the range (lengths[last]), not the partition's nominal boundary");
    if (gop > 0 || gex > 0) return fail(SW_ERR_INVALID, "gap scores must be <= 0");
    if (kind_packed(kind) && (gop < -1000 || gex < -1000)) return fail(SW_ERR_INVALID, "gap score out of range for a 16-bit kind");
    if (!ctx->have_matrix) return fail(SW_ERR_NO_MATRIX, "sw_set_matrix has not been called");
    if (!ctx->have_query) return fail(SW_ERR_NO_QUERY, "sw_set_query has not been called");
    if (n == 0) return SW_OK;
    if (!chars || !offsets || !lengths || !scores || !ids) return fail(SW_ERR_INVALID, "null buffer");
    if (ovf_check && kind_packed(kind) && (!ovf_pos || !ovf_count)) return fail(SW_ERR_INVALID, "overflow check without overflow buffers");
    SW_HIP(hipSetDevice(ctx->device));
    if (ctx->check_bounds) {
        int32_t* slot = reinterpret_cast<int32_t*>(ctx->d_zeros + 64);
        int32_t longest = 0;
        SW_HIP(hipMemsetAsync(slot, 0, sizeof(int32_t), stream));
        hipLaunchKernelGGL(max_length_kernel, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, stream, lengths, positions, count_ptr, first_pos, n, slot);
        SW_HIP(hipGetLastError());
        SW_HIP(hipMemcpyAsync(&longest, slot, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        if (longest > max_subject_len)
            return fail(SW_ERR_INVALID, "max_subject_len " + std::to_string(max_subject_len) + " under-reports: a subject of the range has " +
                                            std::to_string(longest) + " residues");
    }
    // Column-offset recurrence (7.5 instead of 8.5 instructions per cell pair): values grow by a = -gex per column,
    // so it is used while a * columns leaves room below the kind's limit (offs_possible); a subject whose bound
    // score + a * columns reaches the limit is flagged and re-scored like an overflow.
    const int a = -gex;
    int64_t K = 0;
    const bool offs = offs_possible(ctx, kind, lanes, max_subject_len, gop, gex, ovf_check, &K);
    if (kind_packed(kind) && !offs) return fail(SW_ERR_INVALID, "internal: a packed launch without the column-offset recurrence (packed_fallback_kind)");
    int rc = ensure_profile(ctx, kind, lanes, offs, offs ? a : 0, stream);
    if (rc != SW_OK) return rc;
    const Profile& prof = ctx->profiles[kind][shape_index(lanes)][offs];
    const QueryPlan pl = prof.plan;
    const bool multi = pl.nstripes > 1;

    const int subj_per_batch = (swk::kThreads / lanes) * (kind_packed(kind) ? 2 : 1);
    const int nbatches = (n + subj_per_batch - 1) / subj_per_batch;
    int grid = std::min(nbatches, max_grid(ctx));
    if (list.service_workgroups > 0) grid = std::min(grid, list.service_workgroups);

    swk::ScanParams p{};
    p.claim = list.claim;
    p.service = list.service_workgroups > 0 ? 1 : 0;
    p.done_flag = list.done_flag;
    p.done_value = list.done_value;
    p.chars = chars; p.offsets = offsets; p.lengths = lengths;
    p.positions = positions; p.count_ptr = count_ptr;
    p.first_pos = first_pos; p.n = n;
    p.profile = prof.dev; p.nstripes = pl.nstripes;
    switch (kind) {
        case SW_KIND_F16X2: p.gop = swk::Arith<swk::F16X2>::encode_gap(gop); p.gex = swk::Arith<swk::F16X2>::encode_gap(gex); break;
        case SW_KIND_I16X2: p.gop = swk::Arith<swk::I16X2>::encode_gap(gop); p.gex = swk::Arith<swk::I16X2>::encode_gap(gex); break;
        case SW_KIND_I32: p.gop = swk::Arith<swk::I32>::encode_gap(gop); p.gex = swk::Arith<swk::I32>::encode_gap(gex); break;
        default: p.gop = swk::Arith<swk::F32>::encode_gap(gop); p.gex = swk::Arith<swk::F32>::encode_gap(gex); break;
    }
    if (offs) {  // gop slot: gop + a
        const int g = gop - gex;
        switch (kind) {
            case SW_KIND_F16X2: p.gop = swk::Arith<swk::F16X2>::encode_gap(g); break;
            case SW_KIND_I16X2: p.gop = swk::Arith<swk::I16X2>::encode_gap(g); break;
            case SW_KIND_I32: p.gop = swk::Arith<swk::I32>::encode_gap(g); break;
            default: p.gop = swk::Arith<swk::F32>::encode_gap(g); break;
        }
        p.gex_mag = a;
        {   // row classes of the kernel that plan (rows, lanes) selects: lowering words at the class wrap / the last row
            const int P = swk::frame_classes(kl->packed, pl.rows, lanes, multi);
            const int wrap = -a * P, wrap_last = -a * ((pl.rows - 1) % P + 1);
            switch (kind) {
                case SW_KIND_F16X2: p.wrap_class = swk::Arith<swk::F16X2>::encode_gap(wrap); p.wrap_last = swk::Arith<swk::F16X2>::encode_gap(wrap_last); break;
                case SW_KIND_I16X2: p.wrap_class = swk::Arith<swk::I16X2>::encode_gap(wrap); p.wrap_last = swk::Arith<swk::I16X2>::encode_gap(wrap_last); break;
                case SW_KIND_I32: p.wrap_class = swk::Arith<swk::I32>::encode_gap(wrap); p.wrap_last = swk::Arith<swk::I32>::encode_gap(wrap_last); break;
                default: p.wrap_class = swk::Arith<swk::F32>::encode_gap(wrap); p.wrap_last = swk::Arith<swk::F32>::encode_gap(wrap_last); break;
            }
        }
        p.renorm_quads = (int32_t)(K / 4);
        const int lower = -(int)((int64_t)a * K);
        switch (kind) {
            case SW_KIND_F16X2: p.renorm_word = swk::Arith<swk::F16X2>::encode_gap(lower); break;
            case SW_KIND_I16X2: p.renorm_word = swk::Arith<swk::I16X2>::encode_gap(lower); break;
            case SW_KIND_I32: p.renorm_word = swk::Arith<swk::I32>::encode_gap(lower); break;
            default: p.renorm_word = swk::Arith<swk::F32>::encode_gap(lower); break;
        }
    }
    // Streamed subjects (sw_stream_kernel.hpp): packed kinds on 16-lane groups, column-offset recurrence, plain ranges
    const StreamPlan sp = stream_plan(ctx, kind, lanes, a, pl, max_subject_len);
    if (sp.slots >= 1) {
        p.stream_slots = sp.slots; p.stream_cols = sp.cols; p.stream_room = sp.room; p.level_base = sp.base;
        p.jump = sp.jump; p.jump_limit = sp.jump - 4;
        p.jump_word = kind == SW_KIND_F16X2 ? swk::Arith<swk::F16X2>::pos_word(sp.jump) : swk::Arith<swk::I16X2>::pos_word(sp.jump);
    }
    p.scores = scores; p.ids = ids; p.id_offset = id_offset;
    p.ovf_pos = ovf_pos; p.ovf_count = ovf_count; p.ovf_check = (ovf_check && kind_packed(kind)) ? 1 : 0;
    p.scratch = nullptr; p.lcap = 0; p.zeros = ctx->d_zeros + kind * 16;
    p.stat_count = stat_count; p.stat_limit = stat_limit;
    if (multi) {
        p.lcap = border_capacity(max_subject_len, lanes);
        if (p.stream_slots > 1) {
            // the border arrays of a round hold all of its slots: as many columns as the caller's scratch gives the whole grid
            int cols = p.stream_cols;
            while (cols > max_subject_len && (size_t)grid * border_bytes_per_wg(border_capacity(cols, lanes), lanes) > temp_bytes) cols = cols * 3 / 4;
            p.stream_cols = cols;
            p.lcap = border_capacity(std::max(max_subject_len, cols), lanes);
        }
        const size_t per_wg = border_bytes_per_wg(p.lcap, lanes);
        if (!temp || temp_bytes < per_wg) return fail(SW_ERR_TEMP, "temp buffer too small for a multi-stripe query");
        grid = (int)std::min<size_t>((size_t)grid, temp_bytes / per_wg);
        p.scratch = static_cast<uint32_t*>(temp);
    }
    // batches are handed out through an atomic counter (longest subjects first): workgroups that start late
    // because another launch still holds the CUs simply take fewer batches
    SW_HIP(take_work_slot(ctx, stream, &p.work_counter));
    // start handshake (sw_set_start_signal): the signal fires once `quorum` workgroups are resident — all of a small
    // launch, the first 64 of a larger one (its remaining workgroups are next in its queue when the waiter is released)
    p.start_signal = start_signal;
    p.start_quorum = (uint32_t)std::min(grid, 64);
    p.dry_signal = dry_signal;
    // (streamed rounds: the subjects listed for their predecessor's score only are counted apart, sw_set_dirty_counter)
    if (!stat_count && !positions && p.stream_slots > 1 && lanes == 16 && kind_packed(kind)) { p.stat_count = ctx->dirty_counter; p.stat_limit = INT32_MAX; }
    p.dry_value = ctx->dry_value;
    // (the launcher caps the grid at the kernel's resident workgroups minus the reserve and the quorum with it)
    const int reserve = (list.service_workgroups > 0 || list.claim) ? 0 : ctx->grid_reserve;
    SW_HIP(kl->scan(pl.rows, lanes, multi, offs, grid, reserve, stream, p));
    return SW_OK;
}

}  // namespace

namespace {
// Issue-rate micro-runs (tools/ubench/valu_rate.hip, mix_rate.hip in-process): `iters` iterations of 32 independent
// instructions of the kind's inner-loop mix on every lane of 16 waves per CU; every wave also reports how many shader-clock
// ticks (s_memtime) went by per 100 MHz tick (s_memrealtime): the clock the chip actually held during the run.
#define SW_REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define SW_OP_PK_MAX3(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SW_OP_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define SW_OP_MAX3_F32(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SW_OP_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define SW_OP_MAX3_I32(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define SW_OP_PK_FMA(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
#define SW_OP_PK_ADD(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define SW_OP_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define SW_OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
template <int MIX>
__global__ void __launch_bounds__(256) valu_rate_kernel(unsigned* out, unsigned long long* clocks, unsigned seed, int iters) {
    unsigned a[8];
    unsigned b = seed + threadIdx.x, c = seed * 3u + 1u;
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    if constexpr (MIX == 4) {
        asm volatile("v_mov_b32 v0, %0\nv_mov_b32 v1, %0\nv_mov_b32 v2, %0\nv_mov_b32 v3, %0\nv_mov_b32 v4, %0\nv_mov_b32 v5, %0\nv_mov_b32 v6, %0\nv_mov_b32 v7, %0\n"
                     "v_mov_b32 v8, %1\nv_mov_b32 v9, %1\nv_mov_b32 v10, %1\nv_mov_b32 v11, %1\nv_mov_b32 v12, %2\nv_mov_b32 v13, %2\nv_mov_b32 v14, %2\nv_mov_b32 v15, %2\n"
                     :: "v"(a[0]), "v"(b), "v"(c) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
    }
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (MIX == 0) { SW_REP8(SW_OP_PK_MAX3) SW_REP8(SW_OP_PK_MAX3) SW_REP8(SW_OP_PK_MAX3) SW_REP8(SW_OP_PK_MAX3) }
        // fp32 kind: 4 v_add_f32 : 3.5 v_max3_f32 per cell = 16 : 14 + 2 to fill the 32
        if constexpr (MIX == 1) { SW_REP8(SW_OP_ADD_F32) SW_REP8(SW_OP_MAX3_F32) SW_REP8(SW_OP_ADD_F32) SW_OP_MAX3_F32(0) SW_OP_MAX3_F32(1) SW_OP_MAX3_F32(2) SW_OP_MAX3_F32(3) SW_OP_MAX3_F32(4) SW_OP_MAX3_F32(5) SW_OP_ADD_F32(6) SW_OP_ADD_F32(7) }
        // int32 kind: 2.25 v_add_u32 : 3.5 v_max3_i32 per cell ~ 12 : 20
        // the packed kernels' OWN mix (static histogram of the dominant loop body, sw_scan_kernel<f16x2, R = 32, 16 lanes, multi-stripe>,
        // per four steps: 128 v_pk_fma_f16, 454 v_pk_maximum3_f16, 164 v_pk_add_f16 and ~76 others — DPP moves, v_perm_b32, address
        // adds — of 822): in 64 slots 10 fma, 35 max3, 13 add, 3 DPP, 2 v_add_u32, 1 v_perm.  A pure v_pk_maximum3_f16 stream issues
        // FEWER lane-instructions per second than the kernel does (round 5: frac_of_measured_peak 1.06), this one cannot be beaten
        // by a loop of the same composition that also waits for LDS and neighbours
        if constexpr (MIX == 3) {
            SW_REP8(SW_OP_PK_MAX3) SW_OP_PK_FMA(0) SW_OP_PK_FMA(1) SW_OP_PK_ADD(2) SW_OP_PK_ADD(3) SW_OP_DPP(4) SW_OP_PK_FMA(5) SW_OP_PK_ADD(6) SW_OP_ADD_U32(7)
            SW_REP8(SW_OP_PK_MAX3) SW_OP_PK_FMA(0) SW_OP_PK_FMA(1) SW_OP_PK_ADD(2) SW_OP_PK_ADD(3) SW_OP_DPP(4) SW_OP_PK_FMA(5) SW_OP_PK_ADD(6) SW_OP_PERM(7)
            SW_REP8(SW_OP_PK_MAX3) SW_OP_PK_FMA(0) SW_OP_PK_FMA(1) SW_OP_PK_ADD(2) SW_OP_PK_ADD(3) SW_OP_DPP(4) SW_OP_PK_ADD(5) SW_OP_PK_ADD(6) SW_OP_ADD_U32(7)
            SW_REP8(SW_OP_PK_MAX3) SW_OP_PK_FMA(0) SW_OP_PK_FMA(1) SW_OP_PK_ADD(2) SW_OP_PK_ADD(3) SW_OP_PK_MAX3(4) SW_OP_PK_MAX3(5) SW_OP_PK_MAX3(6) SW_OP_PK_ADD(7)
        }
        // mix 4: mix 3's composition with every instruction's three sources in three different register banks (bank = register
        // number mod 4) — explicit registers: accumulators v0..v7, the second operand from v8..v11, the third from v12..v15.
        // With operands wherever the allocator puts them a VOP3(P) instruction whose sources meet in one bank takes an extra
        // cycle: mixes 0 and 3 sustain 58-59.5 lanes/clk/CU, less than the scan kernels themselves, whose registers hipcc assigns
        // with the banks in mind.
        if constexpr (MIX == 4) {
#define SW_B4_MAX3(i, j, k) "v_pk_maximum3_f16 v" #i ", v" #i ", v" #j ", v" #k "\n"
#define SW_B4_FMA(i, j, k) "v_pk_fma_f16 v" #i ", v" #j ", v" #k ", v" #i " op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
#define SW_B4_ADD(i, j) "v_pk_add_f16 v" #i ", v" #i ", v" #j "\n"
#define SW_B4_DPP(i, j) "v_mov_b32_dpp v" #i ", v" #j " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define SW_B4_ADDU(i, j) "v_add_u32 v" #i ", v" #i ", v" #j "\n"
#define SW_B4_ROW8 SW_B4_MAX3(0, 9, 14) SW_B4_MAX3(1, 10, 15) SW_B4_MAX3(2, 11, 12) SW_B4_MAX3(3, 8, 13) SW_B4_MAX3(4, 9, 14) SW_B4_MAX3(5, 10, 15) SW_B4_MAX3(6, 11, 12) SW_B4_MAX3(7, 8, 13)
            asm volatile(
                SW_B4_ROW8 SW_B4_FMA(0, 9, 14) SW_B4_FMA(1, 10, 15) SW_B4_ADD(2, 11) SW_B4_ADD(3, 8) SW_B4_DPP(4, 9) SW_B4_FMA(5, 10, 15) SW_B4_ADD(6, 11) SW_B4_ADDU(7, 8)
                SW_B4_ROW8 SW_B4_FMA(0, 9, 14) SW_B4_FMA(1, 10, 15) SW_B4_ADD(2, 11) SW_B4_ADD(3, 8) SW_B4_DPP(4, 9) SW_B4_FMA(5, 10, 15) SW_B4_ADD(6, 11) SW_B4_MAX3(7, 8, 13)
                SW_B4_ROW8 SW_B4_FMA(0, 9, 14) SW_B4_FMA(1, 10, 15) SW_B4_ADD(2, 11) SW_B4_ADD(3, 8) SW_B4_DPP(4, 9) SW_B4_ADD(5, 10) SW_B4_ADD(6, 11) SW_B4_ADDU(7, 8)
                SW_B4_ROW8 SW_B4_FMA(0, 9, 14) SW_B4_FMA(1, 10, 15) SW_B4_ADD(2, 11) SW_B4_ADD(3, 8) SW_B4_MAX3(4, 9, 14) SW_B4_MAX3(5, 10, 15) SW_B4_MAX3(6, 11, 12) SW_B4_ADD(7, 8)
                ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
        }
        if constexpr (MIX == 2) { SW_REP8(SW_OP_MAX3_I32) SW_REP8(SW_OP_ADD_U32) SW_REP8(SW_OP_MAX3_I32) SW_OP_ADD_U32(0) SW_OP_ADD_U32(1) SW_OP_ADD_U32(2) SW_OP_ADD_U32(3) SW_OP_MAX3_I32(4) SW_OP_MAX3_I32(5) SW_OP_MAX3_I32(6) SW_OP_MAX3_I32(7) }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if constexpr (MIX == 4) asm volatile("v_mov_b32 %0, v0\nv_xor_b32 %0, %0, v7" : "=v"(a[0]) :: "v0", "v7");
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
}
}  // namespace

namespace swi {
int fail(int code, const std::string& msg) { return ::fail(code, msg); }
int32_t query_length(const sw_ctx* ctx) { return ctx && ctx->have_query ? ctx->qlen : 0; }
int device_of(const sw_ctx* ctx) { return ctx ? ctx->device : -1; }
int num_cus(const sw_ctx* ctx) { return ctx ? ctx->num_cus : 0; }
}  // namespace swi

extern "C" {

const char* sw_version(void) { return "cudasw4_amd 0.2 (gfx950)"; }

const char* sw_last_error(void) { return g_last_error.c_str(); }

int sw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sw_ctx_create(int device, sw_ctx** out) {
    if (!out) return fail(SW_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SW_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(SW_ERR_INVALID, "device index out of range");
    SW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SW_HIP(hipGetDeviceProperties(&prop, device));
    sw_ctx* ctx = new sw_ctx;
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (const char* e = getenv("CUDASW4_AMD_GRID_CAP")) ctx->grid_cap = std::max(0, atoi(e));
    if (const char* e = getenv("CUDASW4_AMD_NO_OFFS")) ctx->use_offs = !(e[0] == '1');
    if (const char* e = getenv("CUDASW4_AMD_I32_NATIVE")) ctx->i32_native = e[0] == '1';
    if (const char* e = getenv("CUDASW4_AMD_LANES8_MAX_Q")) ctx->lanes8_max_q = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_LANES4_MAX_Q")) ctx->lanes4_max_q = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_LANES4_MAX_SUBJECT")) ctx->lanes4_max_subject = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_CHECK_BOUNDS")) ctx->check_bounds = e[0] == '1';
    if (const char* e = getenv("CUDASW4_AMD_PIPE_SPIN_LIMIT")) ctx->pipe_spin_limit = (uint32_t)std::max(1ll, atoll(e));
    if (const char* e = getenv("CUDASW4_AMD_PIPE_TEST_DROP_STAGE")) ctx->pipe_drop_stage = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_PIPE_CPL")) ctx->pipe_cpl = atoi(e);
    if (const char* e = getenv("CUDASW4_AMD_STREAM")) ctx->stream_slots = std::max(0, std::min(swk::kStreamMaxSlots, atoi(e)));
    hipError_t e = hipMalloc(&ctx->d_matrix, 26 * swk::kLetters);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_zeros, 256 + 64);  // + the word of the CUDASW4_AMD_CHECK_BOUNDS check (word 64) and the two of sw_streams_run_concurrently (72, 73)
    if (e == hipSuccess) e = hipMalloc(&ctx->d_work, 2 * kWorkSlots * sizeof(uint32_t));
    if (e == hipSuccess) {
        uint32_t z[64] = {};
        for (int i = 0; i < 16; i++) z[SW_KIND_I16X2 * 16 + i] = swk::Arith<swk::I16X2>::kZero;
        e = hipMemcpy(ctx->d_zeros, z, sizeof(z), hipMemcpyHostToDevice);
    }
    // load the four kinds' code objects now (an empty launch each) rather than inside the first query's scan
    for (int kind = 0; kind < 4 && e == hipSuccess; kind++) e = kind_launch(kind)->warm(nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { delete ctx; return fail(SW_ERR_HIP, std::string("context set-up: ") + hipGetErrorString(e)); }
    *out = ctx;
    return SW_OK;
}

int sw_ctx_destroy(sw_ctx* ctx) {
    if (!ctx) return SW_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_matrix) (void)hipFree(ctx->d_matrix);
    if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_query) (void)hipFree(ctx->d_query);
    for (int i = 0; i < sw_ctx::kQuerySlots; i++) {
        if (ctx->h_query[i]) (void)hipHostFree(ctx->h_query[i]);
        if (ctx->query_copied[i]) (void)hipEventDestroy(ctx->query_copied[i]);
    }
    for (auto& row : ctx->profiles)
        for (auto& shape : row)
            for (auto& pr : shape) {
                if (pr.dev) (void)hipFree(pr.dev);
                if (pr.ready) (void)hipEventDestroy(pr.ready);
            }
    delete ctx;
    return SW_OK;
}

int sw_set_matrix(sw_ctx* ctx, const int8_t* matrix_host, int dim) {
    if (!ctx || !matrix_host) return fail(SW_ERR_INVALID, "null argument");
    if (dim != 21 && dim != 25) return fail(SW_ERR_INVALID, "substitution matrices are 21 x 21 (20 amino acids + other) or 25 x 25 (types.hpp:205-396)");
    // internal form: one row per query letter (dim of them) + the row of the query's padding, 21 columns = the dbdata
    // alphabet of the subjects (codes 0..19, 20 = "other" and padding).  25-letter tables: subject code 20 is scored with
    // the table's X column (index 23), the padding row is the X row; both must be negative so that padding neutralises itself.
    constexpr int kX = 23;
    int8_t m[26 * swk::kLetters];
    auto subject_col = [&](int j) { return (dim == 25 && j == swk::kPadLetter) ? kX : j; };
    for (int i = 0; i <= dim; i++) {
        const int qi = i < dim ? i : (dim == 25 ? kX : swk::kPadLetter);
        for (int j = 0; j < swk::kLetters; j++) m[i * swk::kLetters + j] = matrix_host[qi * dim + subject_col(j)];
    }
    for (int i = 0; i <= dim; i++)
        if (m[i * swk::kLetters + swk::kPadLetter] >= 0) return fail(SW_ERR_INVALID, "scores against the padding letter must be negative");
    for (int j = 0; j < swk::kLetters; j++)
        if (m[dim * swk::kLetters + j] >= 0) return fail(SW_ERR_INVALID, "scores against the padding letter must be negative");
    SW_HIP(hipSetDevice(ctx->device));
    SW_HIP(hipMemcpy(ctx->d_matrix, m, (size_t)(dim + 1) * swk::kLetters, hipMemcpyHostToDevice));
    ctx->matrix_max = 1;
    for (int i = 0; i < dim * dim; i++) ctx->matrix_max = std::max(ctx->matrix_max, (int)matrix_host[i]);
    if (ctx->have_query && dim < ctx->dim) ctx->have_query = false;  // the installed query may hold codes of the larger alphabet
    ctx->dim = dim;
    ctx->have_matrix = true;
    for (auto& row : ctx->profiles)
        for (auto& shape : row)
            for (auto& pr : shape) pr.valid = false;
    return SW_OK;
}

int sw_set_query(sw_ctx* ctx, const int8_t* query_codes_host, int32_t qlen, void* stream) {
    if (!ctx || !query_codes_host) return fail(SW_ERR_INVALID, "null argument");
    if (qlen <= 0) return fail(SW_ERR_INVALID, "query length must be positive");
    for (int32_t i = 0; i < qlen; i++)
        if (query_codes_host[i] < 0 || query_codes_host[i] >= ctx->dim) return fail(SW_ERR_INVALID, "query code out of range for the installed matrix");
    SW_HIP(hipSetDevice(ctx->device));
    if ((size_t)qlen > ctx->query_capacity) {
        if (ctx->d_query) SW_HIP(hipFree(ctx->d_query));
        ctx->d_query = nullptr;
        ctx->query_capacity = 0;
        const size_t cap = std::max<size_t>(((size_t)qlen + 4095) / 4096 * 4096 * 2, size_t(1) << 16);
        SW_HIP(hipMalloc(&ctx->d_query, cap));
        ctx->query_capacity = cap;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Pinned staging ring: the caller's buffer is copied on the host (a few KB) and is free again on return; the upload
    // itself is stream-ordered — behind the scans of the previous query that still read d_query or its profiles on `s`
    // — and nothing here waits for the GPU unless the ring has wrapped around onto a copy that is still in flight.
    const int slot = ctx->query_next;
    ctx->query_next = (ctx->query_next + 1) % sw_ctx::kQuerySlots;
    if (ctx->query_slot_used[slot]) SW_HIP(hipEventSynchronize(ctx->query_copied[slot]));
    if ((size_t)qlen > ctx->h_query_capacity[slot]) {
        if (ctx->h_query[slot]) SW_HIP(hipHostFree(ctx->h_query[slot]));
        ctx->h_query[slot] = nullptr;
        ctx->h_query_capacity[slot] = 0;
        const size_t cap = std::max<size_t>(((size_t)qlen + 4095) / 4096 * 4096 * 2, size_t(1) << 16);
        SW_HIP(hipHostMalloc(&ctx->h_query[slot], cap));
        ctx->h_query_capacity[slot] = cap;
    }
    if (!ctx->query_copied[slot]) SW_HIP(hipEventCreateWithFlags(&ctx->query_copied[slot], hipEventDisableTiming));
    memcpy(ctx->h_query[slot], query_codes_host, (size_t)qlen);
    SW_HIP(hipMemcpyAsync(ctx->d_query, ctx->h_query[slot], qlen, hipMemcpyHostToDevice, s));
    // the control words of this query's launches, zeroed in one go (take_work_slot)
    ctx->work_next = (ctx->work_next + kWorkZeroBlock - 1) / kWorkZeroBlock * kWorkZeroBlock;
    SW_HIP(hipMemsetAsync(ctx->d_work + 2 * (ctx->work_next % kWorkSlots), 0, 2 * kWorkZeroBlock * sizeof(uint32_t), s));
    ctx->work_zeroed = kWorkZeroBlock;
    SW_HIP(hipEventRecord(ctx->query_copied[slot], s));
    ctx->query_slot_used[slot] = true;
    ctx->qlen = qlen;
    ctx->have_query = true;
    for (auto& row : ctx->profiles)
        for (auto& shape : row)
            for (auto& pr : shape) pr.valid = false;
    return SW_OK;
}

namespace {
__global__ void __launch_bounds__(256) check_codes_kernel(const int8_t* __restrict__ chars, size_t n, int32_t* bad) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    unsigned m = 0;
    // 16 bytes per lane where the pointer allows, the ragged ends byte by byte
    const size_t head = (16 - ((size_t)chars & 15)) & 15;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < head && tid < n) m |= (unsigned char)chars[tid] > 20 ? 1u : 0u;
    const size_t body = n > head ? (n - head) / 16 * 16 : 0;
    for (size_t i = tid * 16; i < body; i += stride) {
        const uint4 v = *reinterpret_cast<const uint4*>(chars + head + i);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // a byte is > 20 iff it has a bit of 0xE0 set, or it is 21..31: (b + 11) carries into bit 5
            m |= w[k] & 0xE0E0E0E0u;
            m |= ((w[k] & 0x1F1F1F1Fu) + 0x0B0B0B0Bu) & 0x20202020u;
        }
    }
    const size_t tail0 = head + body;
    if (tail0 + tid < n && tid < 16) m |= (unsigned char)chars[tail0 + tid] > 20 ? 1u : 0u;
    if (m) atomicOr(bad, 1);
}
}  // namespace

int sw_check_letter_codes(sw_ctx* ctx, const int8_t* chars, size_t n, int32_t* bad_flag, void* stream) {
    if (!ctx || !bad_flag) return fail(SW_ERR_INVALID, "null argument");
    if (n == 0) return SW_OK;
    if (!chars) return fail(SW_ERR_INVALID, "null buffer");
    SW_HIP(hipSetDevice(ctx->device));
    const size_t want = (n / 16 + 255) / 256 + 1;
    const int grid = (int)std::min<size_t>(want, (size_t)std::max(1, ctx->num_cus) * 16);
    hipLaunchKernelGGL(check_codes_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), chars, n, bad_flag);
    SW_HIP(hipGetLastError());
    return SW_OK;
}

int sw_set_start_signal(sw_ctx* ctx, uint32_t* signal) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    ctx->start_signal = signal;
    return SW_OK;
}


namespace {
// what both row-parallel entry points check before they touch the device
int rows_common_checks(sw_ctx* ctx, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t first_pos, int32_t n,
                       int32_t max_subject_len, int gop, int gex, float* scores, int32_t* ids, bool dry_armed, hipStream_t stream,
                       const char* who) {
    if (dry_armed) return fail(SW_ERR_INVALID, std::string(who) + " never runs dry: sw_set_dry_signal applies to sw_scan_partition launches (the armed signal was cancelled)");
    if (n < 0 || max_subject_len < 0) return fail(SW_ERR_INVALID, "negative count or length");
    if (gop > 0 || gex > 0) return fail(SW_ERR_INVALID, "gap scores must be <= 0");
    if (gop > gex) return fail(SW_ERR_INVALID, std::string(who) + " needs gop <= gex (the prefix form of the horizontal gap)");
    if (gex < -10000 || gop < -100000) return fail(SW_ERR_INVALID, std::string("gap score out of range for ") + who);
    if (!ctx->have_matrix) return fail(SW_ERR_NO_MATRIX, "sw_set_matrix has not been called");
    if (!ctx->have_query) return fail(SW_ERR_NO_QUERY, "sw_set_query has not been called");
    if (n == 0) return SW_OK;
    if (!chars || !offsets || !lengths || !scores || !ids) return fail(SW_ERR_INVALID, "null buffer");
    SW_HIP(hipSetDevice(ctx->device));
    if (ctx->check_bounds) {
        // the kernels never compare a subject's length with what the caller declared: columns beyond it would be dropped silently
        int32_t* slot = reinterpret_cast<int32_t*>(ctx->d_zeros + 64);
        int32_t longest = 0;
        SW_HIP(hipMemsetAsync(slot, 0, sizeof(int32_t), stream));
        hipLaunchKernelGGL(max_length_kernel, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, stream, lengths, nullptr, nullptr, first_pos, n, slot);
        SW_HIP(hipGetLastError());
        SW_HIP(hipMemcpyAsync(&longest, slot, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        SW_HIP(hipStreamSynchronize(stream));
        if (longest > max_subject_len)
            return fail(SW_ERR_INVALID, "max_subject_len " + std::to_string(max_subject_len) + " under-reports: a subject of the range has " +
                                            std::to_string(longest) + " residues");
    }
    return SW_OK;
}

// Columns per lane of a pipeline stage.  Narrow spans finish a row sooner, wide ones keep the number of stages — and of
// hand-offs the first row passes through before the last stage starts — down.  Measured on MI355X (tools/giants_bench.py,
// profiles/r05_giants_bench.txt): a row takes 0.27 / 0.38 / 0.57 us at 4 / 8 / 16 columns per lane, a hand-off 3.6 / 5.0 / 8 us
// (the consumer starts a batch of rows when the producer has finished it), so the launch takes about
// qlen * row + stages * hand-off: the width that minimises that for the current query and the longest subject.
int pipeline_cpl(const sw_ctx* ctx, int32_t n, int32_t max_subject_len) {
    if (ctx->pipe_cpl == 4 || ctx->pipe_cpl == 8 || ctx->pipe_cpl == 16) return ctx->pipe_cpl;
    // (sw_rows_pipeline.hpp: batches of 8 rows for queries up to kPipeShortBatchMaxQuery, of 16 above)
    static const double kRowUs16[3] = {0.27, 0.38, 0.57}, kHopUs16[3] = {3.6, 5.0, 8.0}, kRowUs8[3] = {0.32, 0.42, 0.61}, kHopUs8[3] = {2.4, 3.4, 5.3};
    const bool shortBatch = ctx->qlen <= swk::kPipeShortBatchMaxQuery;
    const double* kRowUs = shortBatch ? kRowUs8 : kRowUs16;
    const double* kHopUs = shortBatch ? kHopUs8 : kHopUs16;
    int best = 16;
    double bestT = 1e300;
    for (int k = 0; k < 3; k++) {
        const int cpl = 4 << k;
        const double stages = std::max(1.0, std::ceil((double)max_subject_len / (64.0 * cpl)));
        // every stage displaces a wave of the launch it runs beside for as long as it lives, and narrow spans spend more
        // instructions per cell: many subjects at once take wide spans (at most ~3 stages per 4 SIMDs of an MI355X)
        if (cpl < 16 && (double)std::max(n, 1) * stages > 768.0) continue;
        const double t = (double)std::max(ctx->qlen, 1) * kRowUs[k] + stages * kHopUs[k];
        if (t < bestT) { bestT = t; best = cpl; }
    }
    return best;
}
constexpr size_t kPipeCtrlBytes = 16;   // control words of a pipelined launch, in front of its hand-off words
int64_t pipeline_stages(int cpl, int32_t max_subject_len) { return std::max<int64_t>(1, ((int64_t)max_subject_len + 64 * cpl - 1) / (64 * cpl)); }
}  // namespace


size_t sw_scan_rows_pipelined_temp_bytes(sw_ctx* ctx, int32_t n, int32_t max_subject_len) {
    if (!ctx || !ctx->have_query || n <= 0 || max_subject_len < 0) return 0;
    const int64_t tickets = (int64_t)n * pipeline_stages(pipeline_cpl(ctx, n, max_subject_len), max_subject_len);
    return kPipeCtrlBytes + (size_t)tickets * ((size_t)ctx->qlen + 1) * sizeof(unsigned long long);
}

namespace {
// the launch both pipelined entry points share: `tickets` workgroups of one wave, hand-off words at `xfer`
int launch_pipeline(sw_ctx* ctx, swk::PipelineParams& p, int cpl, int64_t stages, int64_t tickets, uint32_t* start_signal, void* xfer,
                    hipStream_t stream) {
    const size_t need = kPipeCtrlBytes + (size_t)tickets * ((size_t)ctx->qlen + 1) * sizeof(unsigned long long);
    p.query = ctx->d_query; p.qlen = ctx->qlen; p.matrix = ctx->d_matrix; p.dim = ctx->dim;
    // the launch's control words sit in front of its hand-off words, in the caller's buffer: one memset for both, and no
    // shared ring that a launch in flight on another stream could still be using
    p.ctrl = static_cast<uint32_t*>(xfer);
    p.xfer = reinterpret_cast<unsigned long long*>(static_cast<char*>(xfer) + kPipeCtrlBytes);
    p.start_signal = start_signal;
    // every workgroup counts itself in, also those whose stage does not exist; the handshake fires once as many are
    // resident as the device holds of them beside a bulk grid that is about to take every other slot — one-wave workgroups
    // at the slot's register size, up to 16 per CU — or all of them, whichever is less (with more tickets than
    // resident slots the bulk launch was released only when almost the whole pipeline had run)
    const int slot_regs = ctx->pipe_slot > 0 ? ctx->pipe_slot : 128;
    const int64_t resident = (int64_t)std::max(1, ctx->num_cus) * 4 * std::max(1, std::min(8, 512 / slot_regs)) / 2;
    p.start_quorum = (uint32_t)std::max<int64_t>(1, std::min(tickets, ctx->pipe_quorum > 0 ? (int64_t)ctx->pipe_quorum : resident));
    p.max_stages = (int32_t)stages;
    p.spin_limit = ctx->pipe_spin_limit;
    p.test_drop_stage = ctx->pipe_drop_stage;
    SW_HIP(hipMemsetAsync(xfer, 0xFF, need, stream));   // "not written yet" / "minus one"
    const dim3 grid((unsigned)tickets), block(64);
    const int slot = ctx->pipe_slot;
#define SW_PIPE_LAUNCH_B(CPL, B)                                                                                            \
    do {                                                                                                                    \
        if (slot == 128) hipLaunchKernelGGL((swk::sw_rows_pipeline_kernel<CPL, 128, B>), grid, block, 0, stream, p);        \
        else if (slot == 168) hipLaunchKernelGGL((swk::sw_rows_pipeline_kernel<CPL, 168, B>), grid, block, 0, stream, p);   \
        else if (slot == 256) hipLaunchKernelGGL((swk::sw_rows_pipeline_kernel<CPL, 256, B>), grid, block, 0, stream, p);   \
        else hipLaunchKernelGGL((swk::sw_rows_pipeline_kernel<CPL, 0, B>), grid, block, 0, stream, p);                      \
    } while (0)
#define SW_PIPE_LAUNCH(CPL)                                                                                              \
    do {                                                                                                                 \
        if (ctx->qlen <= swk::kPipeShortBatchMaxQuery) SW_PIPE_LAUNCH_B(CPL, 8);                                         \
        else SW_PIPE_LAUNCH_B(CPL, 16);                                                                                  \
    } while (0)
    if (cpl == 4) SW_PIPE_LAUNCH(4);
    else if (cpl == 8) SW_PIPE_LAUNCH(8);
    else SW_PIPE_LAUNCH(16);
#undef SW_PIPE_LAUNCH_B
#undef SW_PIPE_LAUNCH
    SW_HIP(hipGetLastError());
    return SW_OK;
}

// Re-score lists: the entries whose subject is at least min_len residues long (at most `cap` of them) move to a list of
// their own, which the pipelined launch takes; in the original list they are marked taken (compare-and-swap: a re-score
// service may still be claiming entries), so that the ordinary re-score launch behind — in claim mode — skips them.
__global__ void __launch_bounds__(256) pipeline_pick_kernel(int32_t* list, const int32_t* count_ptr, int32_t max_count, const int32_t* lengths,
                                                            int32_t min_len, int32_t cap, int32_t* out_pos, int32_t* out_count) {
    __shared__ int k;
    if (threadIdx.x == 0) k = 0;
    __syncthreads();
    const int32_t cnt = min(*count_ptr, max_count);
    for (int32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int32_t v = __hip_atomic_load(list + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v < 0 || lengths[v] < min_len) continue;
        const int slot = atomicAdd(&k, 1);
        if (slot >= cap) continue;   // (the counter overshoots; what does not fit stays with the ordinary launch)
        if (atomicCAS(list + i, v, swk::kListTaken) == v) out_pos[slot] = v;
        else out_pos[slot] = -1;     // somebody else took it in between: an empty slot (the pipelined kernel skips it)
    }
    __syncthreads();
    if (threadIdx.x == 0) *out_count = min(k, cap);
}
constexpr int32_t kPipeRescoreCap = 64;   // entries of one re-score list that can run pipelined
size_t pipe_rescore_header_bytes() { return 256 + (size_t)kPipeRescoreCap * sizeof(int32_t); }
}  // namespace

int sw_scan_rows_pipelined(sw_ctx* ctx, const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t first_pos,
                           int32_t n, int32_t max_subject_len, int gop, int gex, float* scores, int32_t* ids, int64_t id_offset,
                           int32_t* fail_count, int32_t* over_limit_count, int32_t* over_limit_count2, int32_t packed_limit,
                           void* temp, size_t temp_bytes, void* stream_) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    uint32_t* const start_signal = ctx->start_signal;
    ctx->start_signal = nullptr;
    const bool dry_armed = ctx->dry_signal != nullptr;
    ctx->dry_signal = nullptr;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // the prefix runs in the frame H~ - k * gex: columns * |gex| must stay far inside 32 bits
    if ((int64_t)max_subject_len * (int64_t)(-gex) >= ((int64_t)1 << 28)) return fail(SW_ERR_INVALID, "subject length x gap extension out of range for sw_scan_rows_pipelined");
    const int rc = rows_common_checks(ctx, chars, offsets, lengths, first_pos, n, max_subject_len, gop, gex, scores, ids, dry_armed, stream, "sw_scan_rows_pipelined");
    if (rc != SW_OK || n == 0) return rc;
    const int cpl = pipeline_cpl(ctx, n, max_subject_len);
    const int64_t stages = pipeline_stages(cpl, max_subject_len);
    const int64_t tickets = (int64_t)n * stages;
    if (tickets > (int64_t)1 << 24) return fail(SW_ERR_INVALID, "too many pipeline stages for one sw_scan_rows_pipelined launch");
    const size_t need = kPipeCtrlBytes + (size_t)tickets * ((size_t)ctx->qlen + 1) * sizeof(unsigned long long);
    if (!temp || temp_bytes < need) return fail(SW_ERR_TEMP, "temp buffer too small for sw_scan_rows_pipelined (sw_scan_rows_pipelined_temp_bytes)");
    swk::PipelineParams p{};
    p.chars = chars; p.offsets = offsets; p.lengths = lengths; p.first_pos = first_pos; p.n = n;
    p.gop = gop; p.gex = gex; p.scores = scores; p.ids = ids; p.id_offset = id_offset;
    p.fail_count = fail_count;
    p.stat_count = over_limit_count; p.stat_count2 = over_limit_count2; p.stat_limit = packed_limit;
    return launch_pipeline(ctx, p, cpl, stages, tickets, start_signal, temp, stream);
}

size_t sw_rescore_overflow_pipelined_temp_bytes(sw_ctx* ctx, int32_t max_subject_len) {
    if (!ctx || !ctx->have_query || max_subject_len < 0) return 0;
    const int64_t tickets = (int64_t)kPipeRescoreCap * pipeline_stages(pipeline_cpl(ctx, kPipeRescoreCap, max_subject_len), max_subject_len);
    return pipe_rescore_header_bytes() + kPipeCtrlBytes + (size_t)tickets * ((size_t)ctx->qlen + 1) * sizeof(unsigned long long);
}

int sw_rescore_overflow_pipelined(sw_ctx* ctx, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count, const int8_t* chars,
                                  const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len, int32_t min_subject_len,
                                  int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, int32_t* fail_count,
                                  int32_t packed_limit, int32_t* true_overflow_count, void* temp, size_t temp_bytes, void* stream_) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    ctx->start_signal = nullptr;   // (these launches run behind the launch that filled their list: no handshake)
    const bool dry_armed = ctx->dry_signal != nullptr;
    ctx->dry_signal = nullptr;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!ovf_pos || !ovf_count) return fail(SW_ERR_INVALID, "null overflow buffers");
    if (max_count <= 0) return SW_OK;
    if ((int64_t)max_subject_len * (int64_t)(-gex) >= ((int64_t)1 << 28)) return fail(SW_ERR_INVALID, "subject length x gap extension out of range for sw_rescore_overflow_pipelined");
    const int rc = rows_common_checks(ctx, chars, offsets, lengths, 0, 1, max_subject_len, gop, gex, scores, ids, dry_armed, stream, "sw_rescore_overflow_pipelined");
    if (rc != SW_OK) return rc;
    const int cpl = pipeline_cpl(ctx, kPipeRescoreCap, max_subject_len);
    const int64_t stages = pipeline_stages(cpl, max_subject_len);
    const int64_t tickets = (int64_t)kPipeRescoreCap * stages;
    const size_t need = sw_rescore_overflow_pipelined_temp_bytes(ctx, max_subject_len);
    if (!temp || temp_bytes < need) return fail(SW_ERR_TEMP, "temp buffer too small for sw_rescore_overflow_pipelined");
    int32_t* out_count = static_cast<int32_t*>(temp);
    int32_t* out_pos = reinterpret_cast<int32_t*>(static_cast<char*>(temp) + 256);
    hipLaunchKernelGGL(pipeline_pick_kernel, dim3(1), dim3(256), 0, stream, ovf_pos, ovf_count, max_count, lengths, std::max(min_subject_len, 1),
                       kPipeRescoreCap, out_pos, out_count);
    SW_HIP(hipGetLastError());
    swk::PipelineParams p{};
    p.chars = chars; p.offsets = offsets; p.lengths = lengths; p.first_pos = 0; p.n = kPipeRescoreCap;
    p.positions = out_pos; p.count_ptr = out_count;
    p.gop = gop; p.gex = gex; p.scores = scores; p.ids = ids; p.id_offset = id_offset;
    p.fail_count = fail_count;
    p.stat_count = true_overflow_count; p.stat_count2 = nullptr; p.stat_limit = packed_limit;
    return launch_pipeline(ctx, p, cpl, stages, tickets, nullptr, static_cast<char*>(temp) + pipe_rescore_header_bytes(), stream);
}

int sw_set_dirty_counter(sw_ctx* ctx, int32_t* counter) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    ctx->dirty_counter = counter;
    return SW_OK;
}

int sw_set_dry_signal(sw_ctx* ctx, uint32_t* signal, uint32_t value) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    ctx->dry_signal = signal;
    ctx->dry_value = value;
    return SW_OK;
}

int sw_set_long16_min(sw_ctx* ctx, int32_t subjects) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    ctx->long16_min = subjects < 0 ? ctx->long16_min_default : subjects;
    return SW_OK;
}

int sw_set_grid_reserve(sw_ctx* ctx, int32_t workgroups) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    if (workgroups < 0) return fail(SW_ERR_INVALID, "negative reserve");
    ctx->grid_reserve = workgroups;
    return SW_OK;
}

namespace {
__global__ void reduce_windows_kernel(const float* __restrict__ win_scores, const int32_t* __restrict__ win_first,
                                      const int32_t* __restrict__ real_pos, int32_t n_real, float* __restrict__ scores,
                                      int32_t* __restrict__ ids, int64_t id_offset) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_real) return;
    float m = -1.0f;
    for (int32_t w = win_first[i]; w < win_first[i + 1]; w++) m = fmaxf(m, win_scores[w]);
    const int32_t pos = real_pos[i];
    scores[pos] = m;
    ids[pos] = (int32_t)(id_offset + pos);
}
}  // namespace

namespace {
__global__ void probe_wait_kernel(unsigned* flag, unsigned* saw, unsigned long long max_ticks) {
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(8);
    *saw = v;
}
__global__ void probe_set_kernel(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the side launch of the start handshake in miniature: count in (what releases the gated stream), then stay resident — for a
// bounded time — until the gated kernel has been seen to run
__global__ void probe_side_kernel(unsigned* signal, unsigned* flag, unsigned* saw, unsigned long long max_ticks) {
    __hip_atomic_fetch_add(signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && wall_clock64() - t0 < max_ticks)
        __builtin_amdgcn_s_sleep(8);
    *saw = v;
}
}  // namespace

int sw_probe_handshake(sw_ctx* ctx, void* side_stream, void* gated_stream, uint32_t* signal) {
    if (!ctx || !signal) return fail(SW_ERR_INVALID, "null argument");
    SW_HIP(hipSetDevice(ctx->device));
    hipStream_t side = static_cast<hipStream_t>(side_stream), gated = static_cast<hipStream_t>(gated_stream);
    unsigned* words = reinterpret_cast<unsigned*>(ctx->d_zeros + 72);  // (flag, saw): two spare words of the context's constant block
    SW_HIP(hipStreamSynchronize(side));
    SW_HIP(hipStreamSynchronize(gated));
    SW_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned), side));
    SW_HIP(hipStreamSynchronize(side));
    const uint32_t base = *reinterpret_cast<volatile uint32_t*>(signal);   // signal memory is host-visible
    SW_HIP(hipStreamWaitValue32(gated, signal, base + 1u, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, gated, words);
    hipLaunchKernelGGL(probe_side_kernel, dim3(1), dim3(1), 0, side, signal, words, words + 1, 1000000ull);  // 100 MHz ticks: 10 ms
    SW_HIP(hipGetLastError());
    // the gated stream must get through within a bounded time; if its wait is never released (a profiler that serialises
    // kernels, a runtime without working wait-value packets) the host releases it by hand
    bool released = true;
    for (int i = 0;; i++) {
        const hipError_t q = hipStreamQuery(gated);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(SW_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        if (i >= 2000) {   // ~2 s
            *reinterpret_cast<volatile uint32_t*>(signal) = base + 1u;
            released = false;
            break;
        }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, nullptr);
    }
    SW_HIP(hipStreamSynchronize(side));
    SW_HIP(hipStreamSynchronize(gated));
    unsigned saw = 0;
    SW_HIP(hipMemcpy(&saw, words + 1, sizeof(unsigned), hipMemcpyDeviceToHost));
    *reinterpret_cast<volatile uint32_t*>(signal) = base;
    return released && saw ? 1 : 0;
}

int sw_streams_run_concurrently(sw_ctx* ctx, void* stream_a, void* stream_b) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    SW_HIP(hipSetDevice(ctx->device));
    // a kernel on stream A waits (at most ~5 ms) for a word that a kernel on stream B sets: it sees the word only if the
    // second kernel could start while the first one was still running, i.e. if the runtime did not put both streams on
    // one hardware queue
    unsigned* words = reinterpret_cast<unsigned*>(ctx->d_zeros + 72);  // two spare words of the context's constant block
    SW_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned), static_cast<hipStream_t>(stream_a)));
    SW_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream_a)));
    hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_a), words, words + 1, 500000ull);  // 100 MHz ticks
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_b), words);
    SW_HIP(hipGetLastError());
    unsigned saw = 0;
    SW_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream_b)));
    SW_HIP(hipMemcpyAsync(&saw, words + 1, sizeof(unsigned), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream_a)));
    SW_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream_a)));
    return saw ? 1 : 0;
}

int sw_measure_valu_rate(sw_ctx* ctx, int mix, int millis, double* lane_instr_per_s, double* shader_hz) {
    if (!ctx || mix < 0 || mix > 4) return fail(SW_ERR_INVALID, "null context or unknown instruction mix");
    SW_HIP(hipSetDevice(ctx->device));
    const int grid = std::max(1, ctx->num_cus) * 4;   // 16 waves per CU: four per SIMD
    unsigned* out = nullptr;
    unsigned long long* clocks = nullptr;
    SW_HIP(hipMalloc(&out, (size_t)grid * 256 * sizeof(unsigned)));
    if (hipMalloc(&clocks, (size_t)grid * 2 * sizeof(unsigned long long)) != hipSuccess) { (void)hipFree(out); return fail(SW_ERR_HIP, "hipMalloc"); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    auto run = [&](int iters, float* ms) -> hipError_t {
        hipError_t e = hipEventRecord(e0, nullptr);
        if (e != hipSuccess) return e;
        if (mix == 0) hipLaunchKernelGGL(valu_rate_kernel<0>, dim3(grid), dim3(256), 0, nullptr, out, clocks, 7u, iters);
        else if (mix == 1) hipLaunchKernelGGL(valu_rate_kernel<1>, dim3(grid), dim3(256), 0, nullptr, out, clocks, 7u, iters);
        else if (mix == 2) hipLaunchKernelGGL(valu_rate_kernel<2>, dim3(grid), dim3(256), 0, nullptr, out, clocks, 7u, iters);
        else if (mix == 3) hipLaunchKernelGGL(valu_rate_kernel<3>, dim3(grid), dim3(256), 0, nullptr, out, clocks, 7u, iters);
        else hipLaunchKernelGGL(valu_rate_kernel<4>, dim3(grid), dim3(256), 0, nullptr, out, clocks, 7u, iters);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        return e;
    };
    float ms = 0.0f;
    int iters = 4096;
    if (err == hipSuccess) err = run(iters, &ms);          // warm-up and calibration
    if (err == hipSuccess && ms > 0.0f) {
        iters = (int)std::min(1.0e7, std::max(1024.0, iters * (double)std::max(millis, 1) / ms));
        err = run(iters, &ms);
    }
    double hz = 0.0;
    if (err == hipSuccess) {
        std::vector<unsigned long long> h((size_t)grid * 2);
        err = hipMemcpy(h.data(), clocks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double ticks = 0.0, wall = 0.0;
        for (int i = 0; i < grid; i++) { ticks += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
        if (wall > 0.0) hz = ticks / wall * 100.0e6;   // s_memrealtime counts at 100 MHz
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    (void)hipFree(clocks);
    if (err != hipSuccess) return fail(SW_ERR_HIP, std::string("sw_measure_valu_rate: ") + hipGetErrorString(err));
    if (lane_instr_per_s) *lane_instr_per_s = ms > 0.0f ? (double)grid * 256.0 * (mix >= 3 ? 64.0 : 32.0) * (double)iters / (ms * 1e-3) : 0.0;
    if (shader_hz) *shader_hz = hz;
    return SW_OK;
}

int32_t sw_window_overlap(sw_ctx* ctx, int gop, int gex) {
    if (!ctx || !ctx->have_query || !ctx->have_matrix) return -1;
    // every gap column costs at least min(|gop|, |gex|); an alignment with a positive score of a query of Q residues has at
    // most Q aligned columns worth at most max(M) each: fewer than Q * max(M) / cost gap columns, Q * (1 + max(M) / cost)
    // subject columns in all
    const int cost = std::min(-gop, -gex);
    if (cost <= 0) return -1;
    const int64_t span = (int64_t)ctx->qlen + (int64_t)ctx->qlen * std::max(1, ctx->matrix_max) / cost + 1;
    return span > 0x3fffffff ? -1 : (int32_t)span;
}

int sw_reduce_windows(sw_ctx* ctx, const float* win_scores, const int32_t* win_first, const int32_t* real_pos, int32_t n_real,
                      float* scores, int32_t* ids, int64_t id_offset, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    if (n_real <= 0) return SW_OK;
    if (!win_scores || !win_first || !real_pos || !scores || !ids) return fail(SW_ERR_INVALID, "null buffer");
    SW_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(reduce_windows_kernel, dim3((n_real + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), win_scores,
                       win_first, real_pos, n_real, scores, ids, id_offset);
    SW_HIP(hipGetLastError());
    return SW_OK;
}

int sw_plan_query(int kind, int32_t qlen, int32_t* rows_per_lane, int32_t* nstripes) {
    if (!kind_launch(kind) || qlen <= 0) return fail(SW_ERR_INVALID, "bad kind or query length");
    const QueryPlan pl = plan_query(kind, qlen, 16);
    if (rows_per_lane) *rows_per_lane = pl.rows;
    if (nstripes) *nstripes = pl.nstripes;
    return SW_OK;
}

int sw_plan_launch(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len, int32_t* effective_kind,
                   int32_t* rows_per_lane, int32_t* nstripes, int32_t* lanes) {
    if (!ctx || !kind_launch(kind)) return fail(SW_ERR_INVALID, "null context or unknown kind");
    if (!ctx->have_query) return fail(SW_ERR_NO_QUERY, "sw_set_query has not been called");
    if (part_id >= SW_NUM_LENGTH_PARTITIONS || max_subject_len < 0) return fail(SW_ERR_INVALID, "bad partition id or length");
    kind = effective_kind_of(ctx, kind, max_subject_len);
    const int ln = part_id < 0 ? rescore_lanes(max_subject_len) : lanes_for_partition(ctx, kind, part_id, n, max_subject_len);
    const QueryPlan pl = plan_query(kind, ctx->qlen, ln);
    if (effective_kind) *effective_kind = kind;
    if (rows_per_lane) *rows_per_lane = pl.rows;
    if (nstripes) *nstripes = pl.nstripes;
    if (lanes) *lanes = ln;
    return SW_OK;
}

int sw_launch_vgpr_slot(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len) {
    if (!ctx || !kind_launch(kind) || !ctx->have_query || part_id >= SW_NUM_LENGTH_PARTITIONS || max_subject_len < 0) return 0;
    kind = effective_kind_of(ctx, kind, max_subject_len);
    const int ln = part_id < 0 ? rescore_lanes(max_subject_len) : lanes_for_partition(ctx, kind, part_id, n, max_subject_len);
    const QueryPlan pl = plan_query(kind, ctx->qlen, ln);
    if (pl.rows <= 0) return 0;
    return swk::vgpr_slot_of(kind, pl.rows, ln, pl.nstripes > 1);
}

int sw_set_rows_pipeline_slot(sw_ctx* ctx, int vgprs) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    if (vgprs != 0 && vgprs != 128 && vgprs != 168 && vgprs != 256) return fail(SW_ERR_INVALID, "a register-file slot is 128, 168 or 256 VGPRs (0: as few as the stage needs)");
    ctx->pipe_slot = vgprs;
    return SW_OK;
}

static size_t scan_temp_bytes_of(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len);
size_t sw_scan_temp_bytes(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len) {
    if (!ctx || !ctx->have_query || !kind_launch(kind) || max_subject_len < 0 || n <= 0) return 0;
    if (kind_packed(kind) && part_id >= 0) {
        // (the gap scores are not known here: a packed launch that falls back to its 32-bit kind may need more)
        const int fb = packed_fallback_kind(kind);
        const size_t own = scan_temp_bytes_of(ctx, kind, part_id, n, max_subject_len);
        return std::max(own, scan_temp_bytes_of(ctx, effective_kind_of(ctx, fb, max_subject_len), part_id, n, max_subject_len));
    }
    return scan_temp_bytes_of(ctx, effective_kind_of(ctx, kind, max_subject_len), part_id, n, max_subject_len);
}

static size_t scan_temp_bytes_of(sw_ctx* ctx, int kind, int part_id, int32_t n, int32_t max_subject_len) {
    const int lanes = part_id < 0 ? rescore_lanes(max_subject_len) : lanes_for_partition(ctx, kind, part_id, n, max_subject_len);
    const QueryPlan pl = plan_query(kind, ctx->qlen, lanes);
    if (pl.nstripes <= 1) return 0;
    const int subj_per_batch = (swk::kThreads / lanes) * (kind_packed(kind) ? 2 : 1);
    const int64_t nbatches = ((int64_t)n + subj_per_batch - 1) / subj_per_batch;
    const int64_t grid = std::min<int64_t>(nbatches, max_grid(ctx));
    // streamed subjects: the border arrays hold a round of several slots (gap scores are not known here: -1 is what the
    // default and almost every caller use; a scratch sized for fewer columns only makes the rounds shorter)
    const StreamPlan sp = stream_plan(ctx, kind, lanes, 1, pl, max_subject_len);
    const int32_t cols = sp.slots > 1 ? std::max(max_subject_len, sp.cols) : max_subject_len;
    return (size_t)grid * border_bytes_per_wg(border_capacity(cols, lanes), lanes);
}

int sw_scan_partition(sw_ctx* ctx, int kind, int part_id, const int8_t* chars, const uint64_t* offsets,
                      const int32_t* lengths, int32_t first_pos, int32_t n, int32_t max_subject_len, int gop, int gex,
                      float* scores, int32_t* ids, int64_t id_offset, int32_t* ovf_pos, int32_t* ovf_count,
                      int ovf_check, void* temp, size_t temp_bytes, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    if (part_id < 0 || part_id >= SW_NUM_LENGTH_PARTITIONS || !kind_launch(kind)) {
        ctx->start_signal = nullptr;   // one-shot: a call that fails cancels what was armed for it
        ctx->dry_signal = nullptr;
        return fail(SW_ERR_INVALID, part_id < 0 || part_id >= SW_NUM_LENGTH_PARTITIONS ? "partition id out of range" : "unknown kind");
    }
    kind = effective_kind_of(ctx, kind, max_subject_len);
    int lanes = lanes_for_partition(ctx, kind, part_id, n, max_subject_len);
    if (kind_packed(kind) && !offs_possible(ctx, kind, lanes, max_subject_len, gop, gex, ovf_check, nullptr)) {
        kind = effective_kind_of(ctx, packed_fallback_kind(kind), max_subject_len);
        lanes = lanes_for_partition(ctx, kind, part_id, n, max_subject_len);
        ovf_pos = nullptr; ovf_count = nullptr; ovf_check = 0;
    }
    return scan_common(ctx, kind, lanes, chars, offsets, lengths, nullptr, nullptr, first_pos, n, max_subject_len, gop, gex,
                       scores, ids, id_offset, ovf_pos, ovf_count, ovf_check, temp, temp_bytes,
                       static_cast<hipStream_t>(stream));
}

int sw_rescore_overflow(sw_ctx* ctx, int kind, const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                        const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                        int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp,
                        size_t temp_bytes, void* stream) {
    return sw_rescore_overflow_stat(ctx, kind, ovf_pos, ovf_count, max_count, chars, offsets, lengths, max_subject_len, gop, gex,
                                    scores, ids, id_offset, temp, temp_bytes, 0, nullptr, stream);
}

int sw_rescore_overflow_stat(sw_ctx* ctx, int kind, const int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                             const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                             int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp,
                             size_t temp_bytes, int32_t packed_limit, int32_t* true_overflow_count, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    uint32_t* const start_signal = ctx->start_signal;   // (one-shot signals: as in sw_rescore_service)
    uint32_t* const dry_signal = ctx->dry_signal;
    ctx->start_signal = nullptr;
    ctx->dry_signal = nullptr;
    if (kind != SW_KIND_I32 && kind != SW_KIND_F32) return fail(SW_ERR_INVALID, "overflow re-score needs a 32-bit kind");
    if (!ovf_pos || !ovf_count) return fail(SW_ERR_INVALID, "null overflow buffers");
    if (max_count <= 0) return SW_OK;
    ctx->start_signal = start_signal;
    ctx->dry_signal = dry_signal;
    kind = effective_kind_of(ctx, kind, max_subject_len);
    // grid sized for max_count; the kernel reads the real count on the device (no host round trip,
    // no device-side launch — cf. float_kernels.cuh:1206-1258)
    const int lanes = rescore_lanes(max_subject_len);
    return scan_common(ctx, kind, lanes, chars, offsets, lengths, ovf_pos, ovf_count, 0, max_count, max_subject_len, gop, gex,
                       scores, ids, id_offset, nullptr, nullptr, 0, temp, temp_bytes, static_cast<hipStream_t>(stream),
                       true_overflow_count, packed_limit);
}

size_t sw_rescore_service_temp_bytes(sw_ctx* ctx, int kind, int32_t max_subject_len, int workgroups) {
    if (!ctx || !ctx->have_query || !kind_launch(kind) || max_subject_len < 0 || workgroups <= 0) return 0;
    kind = effective_kind_of(ctx, kind, max_subject_len);
    const int lanes = rescore_lanes(max_subject_len);
    const QueryPlan pl = plan_query(kind, ctx->qlen, lanes);
    if (pl.nstripes <= 1) return 0;
    return (size_t)workgroups * border_bytes_per_wg(border_capacity(max_subject_len, lanes), lanes);
}

int sw_rescore_service(sw_ctx* ctx, int kind, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                       const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                       int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp, size_t temp_bytes,
                       int32_t packed_limit, int32_t* true_overflow_count, const uint32_t* done_flag, uint32_t done_value,
                       int workgroups, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    // the one-shot signals belong to this call whatever happens to it: an early return must not leave them armed for an
    // unrelated later launch (scan_common takes them over again when the launch goes ahead)
    uint32_t* const start_signal = ctx->start_signal;
    uint32_t* const dry_signal = ctx->dry_signal;
    ctx->start_signal = nullptr;
    ctx->dry_signal = nullptr;
    if (kind != SW_KIND_I32 && kind != SW_KIND_F32) return fail(SW_ERR_INVALID, "overflow re-score needs a 32-bit kind");
    if (!ovf_pos || !ovf_count || !done_flag) return fail(SW_ERR_INVALID, "null overflow buffers or flag");
    if (max_count <= 0 || workgroups <= 0) return SW_OK;
    ctx->start_signal = start_signal;
    ctx->dry_signal = dry_signal;
    kind = effective_kind_of(ctx, kind, max_subject_len);
    ListMode lm;
    lm.claim = ovf_pos;
    lm.service_workgroups = workgroups;
    lm.done_flag = done_flag;
    lm.done_value = done_value;
    return scan_common(ctx, kind, rescore_lanes(max_subject_len), chars, offsets, lengths, ovf_pos, ovf_count, 0, max_count, max_subject_len,
                       gop, gex, scores, ids, id_offset, nullptr, nullptr, 0, temp, temp_bytes, static_cast<hipStream_t>(stream),
                       true_overflow_count, packed_limit, lm);
}

int sw_rescore_overflow_claim(sw_ctx* ctx, int kind, int32_t* ovf_pos, const int32_t* ovf_count, int32_t max_count,
                              const int8_t* chars, const uint64_t* offsets, const int32_t* lengths, int32_t max_subject_len,
                              int gop, int gex, float* scores, int32_t* ids, int64_t id_offset, void* temp, size_t temp_bytes,
                              int32_t packed_limit, int32_t* true_overflow_count, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    uint32_t* const start_signal = ctx->start_signal;   // (one-shot signals: as in sw_rescore_service)
    uint32_t* const dry_signal = ctx->dry_signal;
    ctx->start_signal = nullptr;
    ctx->dry_signal = nullptr;
    if (kind != SW_KIND_I32 && kind != SW_KIND_F32) return fail(SW_ERR_INVALID, "overflow re-score needs a 32-bit kind");
    if (!ovf_pos || !ovf_count) return fail(SW_ERR_INVALID, "null overflow buffers");
    if (max_count <= 0) return SW_OK;
    ctx->start_signal = start_signal;
    ctx->dry_signal = dry_signal;
    kind = effective_kind_of(ctx, kind, max_subject_len);
    ListMode lm;
    lm.claim = ovf_pos;
    return scan_common(ctx, kind, rescore_lanes(max_subject_len), chars, offsets, lengths, ovf_pos, ovf_count, 0, max_count, max_subject_len,
                       gop, gex, scores, ids, id_offset, nullptr, nullptr, 0, temp, temp_bytes, static_cast<hipStream_t>(stream),
                       true_overflow_count, packed_limit, lm);
}

// ------------------------------------------------------------------ top-K
// Replaces the reference's thrust sort_by_key / merge_by_key over ALL scores (cudasw4.cuh:1376-1401).
//   * small inputs, or k >= n: stable descending radix sort of (score, id) pairs, first k copied out;
//   * otherwise a radix SELECT: the composite key (orderable score bits, ~position) is unique, so exactly k
//     elements lie at or above the k-th largest key.  Six histogram passes of 11/11/10 bits (three over the
//     score, three over the position among the ties at the threshold score — skipped on the device when the
//     ties are all taken) find that key reading 4 bytes per element per pass; one more pass compacts the k
//     winners, which are then sorted by their 64-bit keys.  No host round trip: every decision is taken by a
//     one-workgroup kernel that updates a small state block.
// Both paths return the same list: descending score, equal scores in input (position) order — the order the
// driver's position-ordered result lists turn into ascending ids.

namespace {
struct TopkLayout {
    size_t keys_off, vals_off, cub_off, total, cub_bytes;
};
TopkLayout topk_layout(int64_t n) {
    TopkLayout L{};
    size_t cub = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const float*)nullptr, (float*)nullptr,
                                                       (const int32_t*)nullptr, (int32_t*)nullptr, (int)std::max<int64_t>(n, 1));
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    L.keys_off = 0;
    L.vals_off = al((size_t)n * sizeof(float));
    L.cub_off = L.vals_off + al((size_t)n * sizeof(int32_t));
    L.cub_bytes = cub;
    L.total = L.cub_off + al(cub);
    return L;
}
__global__ void topk_copy_kernel(const float* keys, const int32_t* vals, int64_t n, int k, float* out_s, int32_t* out_i) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) {
        out_s[i] = i < n ? keys[i] : -1.0f;
        out_i[i] = i < n ? vals[i] : -1;
    }
}

// ---- radix select
constexpr int kSelBins = 2048;
constexpr int kSelPasses = 6;                      // digits of the 64-bit composite key, most significant first
__constant__ const int kSelShift[kSelPasses] = {53, 42, 32, 21, 10, 0};
__constant__ const int kSelBits[kSelPasses] = {11, 11, 10, 11, 11, 10};

struct TopkState {
    unsigned long long prefix;      // digits of the k-th largest key decided so far (high bits)
    unsigned long long decided;     // mask of the decided bits
    unsigned long long remaining;   // how many elements are still to be taken from the current bucket
    unsigned int skip_rest;         // the whole current bucket is taken: no further passes needed
    unsigned int out_count;         // compaction cursor
    unsigned int blocks_done;       // workgroups of the current histogram pass that have added their part (the last one picks)
    unsigned int hist[kSelBins];
};

__device__ __forceinline__ unsigned long long topk_key(float score, unsigned int pos) {
    unsigned int u = __float_as_uint(score);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;  // monotonic: larger float -> larger unsigned
    return ((unsigned long long)u << 32) | (unsigned long long)(~pos);
}

__global__ void topk_init_kernel(TopkState* st, int k) {
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x) st->hist[i] = 0;
    if (threadIdx.x == 0) { st->prefix = 0; st->decided = 0; st->remaining = (unsigned long long)k; st->skip_rest = 0; st->out_count = 0; st->blocks_done = 0; }
}

// walk the histogram from the top bin down to the bin that holds the `remaining`-th element (one workgroup of 256 threads;
// h: kSelBins words, part: 256 double words of LDS)
__device__ void topk_pick(TopkState* st, int pass, unsigned int* h, unsigned long long* part) {
    const int nb = 1 << kSelBits[pass];
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x) { h[i] = i < nb ? __hip_atomic_load(&st->hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; }
    __syncthreads();
    // thread t owns bins [nb - 8(t+1), nb - 8t) (descending): sum, then a serial scan over 256 partial sums by thread 0
    constexpr int per = kSelBins / 256;
    unsigned long long mine = 0;
    for (int j = 0; j < per; j++) { const int b = kSelBins - 1 - (threadIdx.x * per + j); mine += h[b]; }
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long need = st->remaining, above = 0;
        int t = 0;
        while (t < 255 && above + part[t] < need) { above += part[t]; t++; }
        int b = kSelBins - 1 - t * per;
        for (int j = 0; j < per - 1 && above + h[b] < need; j++) { above += h[b]; b--; }
        // bin b holds the element: everything above it is taken
        const int shift = kSelShift[pass];
        st->prefix |= (unsigned long long)b << shift;
        st->decided |= (unsigned long long)((1u << kSelBits[pass]) - 1u) << shift;
        st->remaining = need - above;
        if (st->remaining == h[b]) st->skip_rest = 1;  // the whole bin is taken: its elements need no further ordering
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x) st->hist[i] = 0;
}

__global__ void __launch_bounds__(256) topk_hist_kernel(const float* __restrict__ scores, int64_t n, TopkState* st, int pass) {
    if (st->skip_rest) return;
    __shared__ unsigned int h[kSelBins];
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, decided = st->decided;
    const int shift = kSelShift[pass];
    const unsigned int mask = (1u << kSelBits[pass]) - 1u;
    // The scores of a scan crowd into a few bins of the leading digits (small integers: one exponent), and 64 lanes adding to
    // ONE LDS word serialise — 20 us per pass on 570 000 scores.  So the lanes that share the first active lane's bin add
    // once per wave; the others (the low digits, where few elements are left) add one by one.
    const int wlane = threadIdx.x & 63;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {   // wave-uniform trip count
        const int64_t i = i0 + threadIdx.x;
        bool act = false;
        unsigned int bin = 0;
        if (i < n) {
            const unsigned long long key = topk_key(scores[i], (unsigned int)i);
            act = (key & decided) == prefix;
            bin = (unsigned int)(key >> shift) & mask;
        }
        const unsigned long long m = __ballot(act);
        if (m == 0) continue;
        const unsigned int lb = (unsigned int)__shfl((int)bin, __ffsll((long long)m) - 1);
        const unsigned long long same = __ballot(act && bin == lb);
        if (act) {
            if (bin != lb) atomicAdd(&h[bin], 1u);
            else if (wlane == __ffsll((long long)same) - 1) atomicAdd(&h[lb], (unsigned int)__popcll(same));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += blockDim.x)
        if (h[i]) atomicAdd(&st->hist[i], h[i]);
    // the workgroup that adds its part last picks the bin (round 5: was a launch of its own after every pass — twelve small
    // launches per top-K instead of six weigh on a query that is scanned in a millisecond)
    __shared__ unsigned long long part[256];
    __shared__ bool last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&st->blocks_done, 1u) + 1u == gridDim.x;
    __syncthreads();
    if (!last) return;
    __threadfence();
    topk_pick(st, pass, h, part);
    if (threadIdx.x == 0) st->blocks_done = 0;
}

// winners: keys above the decided prefix, or inside it (all of them when skip_rest, else the prefix is a full key)
__global__ void __launch_bounds__(256) topk_compact_kernel(const float* __restrict__ scores, const int32_t* __restrict__ ids, int64_t n,
                                                           TopkState* st, int k, unsigned long long* out_keys, int32_t* out_ids) {
    const unsigned long long prefix = st->prefix, decided = st->decided;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long key = topk_key(scores[i], (unsigned int)i);
        if ((key & decided) >= prefix) {
            const unsigned int slot = atomicAdd(&st->out_count, 1u);
            if (slot < (unsigned int)k) { out_keys[slot] = key; out_ids[slot] = ids[i]; }
        }
    }
}

__global__ void topk_emit_kernel(const unsigned long long* keys, const int32_t* vals, int k, float* out_s, int32_t* out_i) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) {
        unsigned int u = (unsigned int)(keys[i] >> 32);
        u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
        out_s[i] = __uint_as_float(u);
        out_i[i] = vals[i];
    }
}

// the k winners in descending key order (keys are unique): every element's rank is the number of larger keys — one
// workgroup, k <= 1024 (hipCUB's radix sort of ten 64-bit keys took 15 ... 90 us)
__global__ void __launch_bounds__(256) topk_rank_emit_kernel(const unsigned long long* keys, const int32_t* vals, int k, float* out_s, int32_t* out_i) {
    __shared__ unsigned long long sk[1024];
    for (int i = threadIdx.x; i < k; i += blockDim.x) sk[i] = keys[i];
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const unsigned long long mine = sk[i];
        int rank = 0;
        for (int j = 0; j < k; j++) rank += sk[j] > mine ? 1 : 0;
        unsigned int u = (unsigned int)(mine >> 32);
        u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
        out_s[rank] = __uint_as_float(u);
        out_i[rank] = vals[i];
    }
}

// ---- small k (the reference's default: --top 10): two launches instead of nine.  Every workgroup walks chunks of 2 048 scores
// with the k best keys it has seen so far in LDS; a chunk that holds nothing above the k-th best is skipped after one
// block-wide test, otherwise its maxima are extracted one by one (keys are unique: the winner clears its own element).  The
// second kernel does the same over the workgroups' lists and emits the winners in order.  A query that is scanned in a
// millisecond and a half spent 3 % of it in the radix select's nine small launches.
constexpr int kSmallK = 32;
constexpr int kSmallChunk = 2048;   // 256 threads x 8 elements
constexpr int kSmallMaxGrid = 1024;

__device__ __forceinline__ unsigned long long block_max_u64(unsigned long long v, unsigned long long* wmax) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    __syncthreads();   // (wmax of the round before has been read)
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long m = wmax[0];
#pragma unroll
    for (int w = 1; w < 4; w++) m = wmax[w] > m ? wmax[w] : m;
    return m;
}

// merge the elements key[0..8) of the 256 threads into the descending list best[0..k)
__device__ __forceinline__ void small_merge(unsigned long long (&key)[8], unsigned long long* best, unsigned long long* wmax, int k) {
    unsigned long long mine = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) mine = key[e] > mine ? key[e] : mine;
    for (int round = 0; round < k; round++) {
        const unsigned long long m = block_max_u64(mine, wmax);
        if (m <= best[k - 1]) break;   // uniform: nothing of this chunk can still enter
        if (mine == m) {               // unique keys: exactly one thread; it drops the element and finds its next best
            mine = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (key[e] == m) key[e] = 0;
                mine = key[e] > mine ? key[e] : mine;
            }
        }
        if (threadIdx.x == 0) {        // sorted insertion
            int i = k - 1;
            while (i > 0 && best[i - 1] < m) { best[i] = best[i - 1]; i--; }
            best[i] = m;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) topk_small_partial_kernel(const float* __restrict__ scores, int64_t n, int k, unsigned long long* out) {
    __shared__ unsigned long long best[kSmallK];
    __shared__ unsigned long long wmax[4];
    if (threadIdx.x < kSmallK) best[threadIdx.x] = 0;
    __syncthreads();
    const int64_t nchunks = (n + kSmallChunk - 1) / kSmallChunk;
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        unsigned long long key[8];
        const unsigned long long thr = best[k - 1];
        bool above = false;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int64_t i = c * kSmallChunk + e * 256 + threadIdx.x;
            key[e] = i < n ? topk_key(scores[i], (unsigned int)i) : 0ull;
            above = above || key[e] > thr;
        }
        if (!__syncthreads_or(above)) continue;
        small_merge(key, best, wmax, k);
    }
    __syncthreads();
    if (threadIdx.x < k) out[(size_t)blockIdx.x * k + threadIdx.x] = best[threadIdx.x];
}

__global__ void __launch_bounds__(256) topk_small_final_kernel(const unsigned long long* __restrict__ cand, int ncand, const int32_t* __restrict__ ids,
                                                               int k, float* out_s, int32_t* out_i) {
    __shared__ unsigned long long best[kSmallK];
    __shared__ unsigned long long wmax[4];
    if (threadIdx.x < kSmallK) best[threadIdx.x] = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ncand; c0 += kSmallChunk) {
        unsigned long long key[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int i = c0 + e * 256 + threadIdx.x;
            key[e] = i < ncand ? cand[i] : 0ull;
        }
        small_merge(key, best, wmax, k);
    }
    __syncthreads();
    if (threadIdx.x < k) {
        const unsigned long long key = best[threadIdx.x];
        unsigned int u = (unsigned int)(key >> 32);
        u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
        out_s[threadIdx.x] = __uint_as_float(u);
        out_i[threadIdx.x] = ids[~(unsigned int)key];
    }
}

int small_grid(const sw_ctx* ctx, int64_t n) {
    const int64_t nchunks = (n + kSmallChunk - 1) / kSmallChunk;
    return (int)std::max<int64_t>(1, std::min<int64_t>(nchunks, std::min<int64_t>(kSmallMaxGrid, (int64_t)std::max(1, ctx->num_cus) * 4)));
}

struct SelectLayout {
    size_t state_off, keys_a, keys_b, vals_a, vals_b, cub_off, cub_bytes, total;
};
SelectLayout select_layout(int k) {
    SelectLayout L{};
    size_t cub = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                       (const int32_t*)nullptr, (int32_t*)nullptr, std::max(k, 1));
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    L.state_off = 0;
    L.keys_a = al(sizeof(TopkState));
    L.keys_b = L.keys_a + al((size_t)k * 8);
    L.vals_a = L.keys_b + al((size_t)k * 8);
    L.vals_b = L.vals_a + al((size_t)k * 4);
    L.cub_off = L.vals_b + al((size_t)k * 4);
    L.cub_bytes = cub;
    L.total = L.cub_off + al(cub);
    return L;
}

// the select pays off once the sort's passes over all n pairs dominate; CUDASW4_AMD_TOPK=sort|select forces a path (tests)
bool use_select(int64_t n, int k) {
    const char* force = getenv("CUDASW4_AMD_TOPK");
    if (k >= n) return false;
    if (force && force[0] == 's' && force[1] == 'o') return false;
    if (force && force[0] == 's' && force[1] == 'e') return true;
    return n >= (int64_t(1) << 17) && (int64_t)k * 8 <= n;
}
}  // namespace

size_t sw_topk_temp_bytes(int64_t n, int k) {
    if (n <= 0 || k <= 0) return 0;
    // sized for any path, so that a forced path (tests) never outgrows a buffer sized by this call
    return std::max(std::max(topk_layout(n).total, k < n ? select_layout(k).total : size_t(0)), (size_t)kSmallMaxGrid * kSmallK * sizeof(unsigned long long));
}

int sw_topk(sw_ctx* ctx, const float* scores, const int32_t* ids, int64_t n, int k, float* out_scores,
            int32_t* out_ids, void* temp, size_t temp_bytes, void* stream) {
    if (!ctx) return fail(SW_ERR_INVALID, "null context");
    if (k <= 0) return SW_OK;
    if (n < 0 || n > 0x7fffffffLL) return fail(SW_ERR_INVALID, "bad result count");
    if (!out_scores || !out_ids) return fail(SW_ERR_INVALID, "null output");
    SW_HIP(hipSetDevice(ctx->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n > 0 && (!scores || !ids)) return fail(SW_ERR_INVALID, "null input");
    const char* force = getenv("CUDASW4_AMD_TOPK");
    if (n > k && k <= kSmallK && (!force || (force[0] == 's' && force[1] == 'm'))) {   // CUDASW4_AMD_TOPK=small|select|sort forces a path (tests)
        const int grid = small_grid(ctx, n);
        if (!temp || temp_bytes < (size_t)grid * k * sizeof(unsigned long long)) return fail(SW_ERR_TEMP, "temp buffer too small for top-K");
        unsigned long long* cand = static_cast<unsigned long long*>(temp);
        hipLaunchKernelGGL(topk_small_partial_kernel, dim3(grid), dim3(256), 0, s, scores, n, k, cand);
        hipLaunchKernelGGL(topk_small_final_kernel, dim3(1), dim3(256), 0, s, cand, grid * k, ids, k, out_scores, out_ids);
        SW_HIP(hipGetLastError());
        return SW_OK;
    }
    if (n > 0 && use_select(n, k)) {
        const SelectLayout L = select_layout(k);
        if (!temp || temp_bytes < L.total) return fail(SW_ERR_TEMP, "temp buffer too small for top-K");
        char* base = static_cast<char*>(temp);
        TopkState* st = reinterpret_cast<TopkState*>(base + L.state_off);
        auto* keys_a = reinterpret_cast<unsigned long long*>(base + L.keys_a);
        auto* keys_b = reinterpret_cast<unsigned long long*>(base + L.keys_b);
        auto* vals_a = reinterpret_cast<int32_t*>(base + L.vals_a);
        auto* vals_b = reinterpret_cast<int32_t*>(base + L.vals_b);
        const int grid = (int)std::min<int64_t>((n + 256 * 16 - 1) / (256 * 16), (int64_t)std::max(1, ctx->num_cus) * 8);
        hipLaunchKernelGGL(topk_init_kernel, dim3(1), dim3(256), 0, s, st, k);
        for (int pass = 0; pass < kSelPasses; pass++) {
            hipLaunchKernelGGL(topk_hist_kernel, dim3(grid), dim3(256), 0, s, scores, n, st, pass);
        }
        hipLaunchKernelGGL(topk_compact_kernel, dim3(grid), dim3(256), 0, s, scores, ids, n, st, k, keys_a, vals_a);
        SW_HIP(hipGetLastError());
        if (k <= 1024) {
            hipLaunchKernelGGL(topk_rank_emit_kernel, dim3(1), dim3(256), 0, s, keys_a, vals_a, k, out_scores, out_ids);
        } else {
            size_t cub = L.cub_bytes;
            SW_HIP(hipcub::DeviceRadixSort::SortPairsDescending(base + L.cub_off, cub, keys_a, keys_b, vals_a, vals_b, k, 0, 64, s));
            hipLaunchKernelGGL(topk_emit_kernel, dim3((k + 255) / 256), dim3(256), 0, s, keys_b, vals_b, k, out_scores, out_ids);
        }
        SW_HIP(hipGetLastError());
        return SW_OK;
    }
    float* keys = nullptr;
    int32_t* vals = nullptr;
    if (n > 0) {
        const TopkLayout L = topk_layout(n);
        if (!temp || temp_bytes < L.total) return fail(SW_ERR_TEMP, "temp buffer too small for top-K");
        char* base = static_cast<char*>(temp);
        keys = reinterpret_cast<float*>(base + L.keys_off);
        vals = reinterpret_cast<int32_t*>(base + L.vals_off);
        size_t cub = L.cub_bytes;
        SW_HIP(hipcub::DeviceRadixSort::SortPairsDescending(base + L.cub_off, cub, scores, keys, ids, vals, (int)n, 0, 32, s));
    }
    hipLaunchKernelGGL(topk_copy_kernel, dim3((k + 255) / 256), dim3(256), 0, s, keys, vals, n, k, out_scores, out_ids);
    SW_HIP(hipGetLastError());
    return SW_OK;
}

}  // extern "C"
