// sw_rows_pipeline.hpp — the few very long subjects of a real DB on MANY compute units at once.
//
// One alignment group walks a subject column by column: 35 000 dependent steps per stripe of the query for Swiss-Prot's
// longest protein, 60 ms for a 5 478-residue query whatever else the GPU does.  Round 4 gave such a subject one workgroup
// of 1024 threads = one CU, walking the QUERY row by row with all columns at once (sw_rows_kernel.hpp, removed in round 6:
// 4.1 us per query row, 22.6 ms for that query, strictly serial — on a 1/8 shard of a Swiss-Prot-like DB the rank that
// holds that protein ran at 0.60 of the full-DB rate, profiles/r04_shard_proxy.txt).  The reference has no intra-subject
// parallelism either (one thread group per subject, cudasw4.cuh:2026-2103).
//
// Here the subject is cut into SPANS of 64 * CPL columns and every span is a STAGE of a pipeline: one wave (a workgroup
// of 64 threads, alone on its SIMD lane group, raised priority) that owns the span for all query rows and walks the
// query row by row, a few rows behind the stage to its left.  The algebra (tests/test_rows_algorithm_cpu.py restates it in numpy): F and the diagonal are
// local to a column, the horizontal gap is the max-plus prefix
//     E(i,j) = gop + (j-1) gex + max_{k<j} ( H~(i,k) - k gex ),      H~ = max(0, diagonal + score, F)
// (exact for gop <= gex).  What a stage needs from its left neighbour per row i is TWO numbers:
//     carry(i) = max over all columns k left of the span of  H~(i,k) - k gex      (the prefix so far, global frame)
//     hlast(i) = H(i, last column of the neighbour's span)                        (the diagonal input of row i + 1)
// and they travel as ONE 64-bit word per row through a hand-off array in device memory: written once by the producer's
// lane 63 with an agent-scope atomic store, read by the consumer a batch of rows at a time with agent-scope atomic loads
// (both bypass the XCD-local L2), "not written yet" = all ones (hlast >= 0 makes that pattern impossible).  No flags, no
// fences, no two-way waits: a stage only ever waits for data of the stage before it.
//
// Deadlock freedom does not rest on all stages being resident: a workgroup takes its (subject, stage) from a ticket
// counter when it STARTS, the stage before has the ticket before, so whoever a running stage waits for is itself
// running or done.  Every wait is bounded (spin_limit polls, ~2 s): a stage that gives up raises the launch's abort word,
// which ends everybody else's waits, marks its subject's score -2 and counts itself in *fail_count — the host driver
// checks that word with the scan's counters and fails the query loudly.
//
// A stage must not leave a hole behind.  The bulk launch that starts beside this one is a PERSISTENT grid: its waves are
// placed once and stay to the end of the scan, and both the vector registers of a SIMD and the LDS of a CU are allocated as
// contiguous ranges.  The first form of this kernel (2.3 KB of LDS for the substitution table, 56 VGPRs) ran 2 ms and cost
// the bulk launch 8 ... 35 % of its WHOLE duration (profiles/r05_pipeline_fragmentation.txt): behind the small allocations
// the bulk waves sat at odd offsets, and when the stages had left, the 56-register / 2.5 KB holes they left at the start of
// the register file / of the LDS were useless to the 168- or 256-register, 51 ... 63 KB workgroups still queued — three
// waves per SIMD became two (or two one) for the rest of the scan.  So a stage uses NO LDS (the substitution table lives in
// 26 VGPRs, one query letter per register, one subject letter per lane, and is read with ds_bpermute, which needs no
// allocation), and it occupies exactly one register-file SLOT of the launch it runs beside (SLOT = 128, 168 or 256
// VGPRs = what a wave of a four-, three- or two-waves-per-SIMD scan kernel takes): when it leaves, a queued wave of that
// launch fits the hole exactly.
//
// int32 arithmetic; ~12 VALU instructions per cell (the scan kernels: 6.5) but 69 SIMDs instead of one for a
// 35 000-residue subject: ~0.3 us per query row whatever the subject's length.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>


namespace swk {

constexpr int kRowsNeg = -(1 << 29);   // "no value": below every reachable score, far from the int32 range's end


struct PipelineParams {
    const int8_t* chars;       // subject letters; subject pos starts at chars + (offsets[pos] - offsets[0])
    const uint64_t* offsets;
    const int32_t* lengths;
    int32_t first_pos;         // subjects first_pos .. first_pos + n - 1, ascending length; the longest gets the first tickets
    int32_t n;
    const int32_t* positions;  // optional list of subject positions instead (a re-score list), n = its capacity ...
    const int32_t* count_ptr;  // ... and its length on the device
    const int8_t* query;       // letter codes 0 .. dim-1; readable up to the next multiple of 16 behind qlen
    int32_t qlen;
    const int8_t* matrix;      // (dim + 1) x 21 substitution scores, row = query letter
    int32_t dim;
    int32_t gop, gex;          // <= 0, gop <= gex
    float* scores;
    int32_t* ids;
    int64_t id_offset;
    unsigned long long* xfer;  // hand-off words: (qlen + 1) per ticket, all ones before the launch
    // the 16 bytes in front of the hand-off words, filled with 0xFF together with them by ONE memset ("not written yet" and
    // "minus one"): [0] tickets handed out, minus one; [1] workgroups counted in (start handshake), minus one; [2] abort (1:
    // raised).  (Round 5 kept these words in a ring of 64 blocks of the context that nothing guarded against reuse by a
    // launch still in flight on another stream: ADVICE r5.)
    uint32_t* ctrl;
    uint32_t* start_signal;    // start handshake (sw_set_start_signal), or nullptr
    uint32_t start_quorum;
    int32_t* fail_count;       // += 1 per stage that gave up waiting (nullptr: not counted)
    int32_t* stat_count;       // optional: += 1 (and stat_count2 likewise) per subject whose score is >= stat_limit — the
    int32_t* stat_count2;      // reference's overflow statistic for subjects of a packed partition (half2_kernels.cuh:1087-1109)
    int32_t stat_limit;
    int32_t max_stages;        // tickets per subject: the stages of the longest subject the caller declared
    uint32_t spin_limit;       // polls of one wait before the stage gives up
    int32_t test_drop_stage;   // >= 0 (tests of the failure path): this stage of every subject leaves without producing
};

// Rows whose hand-off words a stage fetches at once (BATCH) and how many batches ahead it requests them.  A stage starts a
// batch when its neighbour has finished it, so the first row reaches the last stage after stages x (BATCH rows + latency):
// 8 rows two batches ahead make a hand-off 2.4 / 3.4 / 5.3 us at 4 / 8 / 16 columns per lane instead of 3.6 / 5.0 / 8 us with
// 16 rows one batch ahead, for 10-15 % more time per row (0.32 / 0.42 / 0.61 against 0.27 / 0.38 / 0.57 us): short batches for
// queries up to kPipeShortBatchMaxQuery rows (35 213 x 567: 0.60 -> 0.46 ms, x 144: 0.39 -> 0.26 ms, x 5 478: 1.98 -> 2.10).
constexpr int kPipeShortBatchMaxQuery = 2048;
constexpr unsigned long long kPipeEmpty = ~0ull;
constexpr float kPipeFailedScore = -2.0f;

__device__ __forceinline__ unsigned long long pipe_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pipe_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pipe_pack(int lo, int hi) {
    return (unsigned long long)(uint32_t)lo | ((unsigned long long)(uint32_t)hi << 32);
}

constexpr int kPipeTableRows = 26;   // query letters (<= 25) + the padding row

template <int CPL, int SLOT, int BATCH>
__global__ void __launch_bounds__(64) sw_rows_pipeline_kernel(const PipelineParams p) {
    constexpr int kPipeBatch = BATCH, kPipeDepth = BATCH <= 8 ? 2 : 1;
    // the wave's register allocation is what the highest register it names says: claim the whole slot
    if constexpr (SLOT == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (SLOT == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if constexpr (SLOT == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);   // a stage is a dependent chain that everybody to its right waits for
    if (p.start_signal && lane == 0) {
        if (atomicAdd(p.ctrl + 1, 1u) + 2u == p.start_quorum)
            __hip_atomic_fetch_add(p.start_signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (subject, stage) from a ticket taken NOW: the stage this one waits for holds the ticket before, so it has started
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(p.ctrl, 1u) + 1u;
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    const int stage = (int)(t % (uint32_t)p.max_stages);
    int pos;
    if (p.positions) {
        const int subj = (int)(t / (uint32_t)p.max_stages);
        if (subj >= min(*p.count_ptr, p.n)) return;
        pos = p.positions[subj];
        if (pos < 0) return;   // (an entry somebody else took while the list was picked)
    } else {
        const int subj = p.n - 1 - (int)(t / (uint32_t)p.max_stages);
        if (subj < 0) return;
        pos = p.first_pos + subj;
    }
    const int len = p.lengths[pos];
    constexpr int kSpan = 64 * CPL;
    const int nstages = max(1, (len + kSpan - 1) / kSpan);
    if (stage >= nstages) return;
    if (stage == p.test_drop_stage) return;   // (tests: a stage that is lost without a trace — its successors must give up, loudly)
    const bool feeds = stage + 1 < nstages;   // somebody waits for this stage's words

    // substitution scores: register q holds query letter q's row, lane j the score against subject letter j; "letter" 21 =
    // a position behind the subject's end scores -30000 against everything, so nothing positive ever starts there
    int tbl[kPipeTableRows];
#pragma unroll
    for (int q = 0; q < kPipeTableRows; q++) tbl[q] = (q <= p.dim && lane < 21) ? (int)p.matrix[q * 21 + lane] : -30000;

    const int8_t* const s = p.chars + (p.offsets[pos] - p.offsets[0]);
    const int col0 = (stage * 64 + lane) * CPL;   // first owned column (0-based)
    int lofs[CPL];                                // 4 x the owned letters: the ds_bpermute address of their lane in a table register
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int col = col0 + c;
        int letter = 21;
        if (col < len) {
            letter = (int)s[col];
            if (letter < 0 || letter > 20) letter = 20;
        }
        lofs[c] = letter * 4;
    }
    const size_t region = (size_t)p.qlen + 1;
    const unsigned long long* const xin = p.xfer + (size_t)(t - (stage > 0 ? 1u : 0u)) * region;   // the stage before's words
    unsigned long long* const xout = p.xfer + (size_t)t * region;
    const int gop = p.gop, gex = p.gex;
    const int kg0 = (col0 + 1) * gex;   // k * gex of the first owned column (k counts from 1)

    int H[CPL], F[CPL], G[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) { H[c] = 0; F[c] = -10000; G[c] = 0; }
    int best = 0;
    int hleft = 0;   // H(i-1, col0 - 1): the diagonal input of the first owned column
    bool failed = false;
    const int sl = lane & (kPipeBatch - 1);

    // a wait for the stage before: poll until none of the wanted words is "not written yet"; bounded
    // (bounded by the CLOCK, not by a number of polls: a poll takes 2 us on an idle chip and many times that for a wave that is
    // descheduled or shares its SIMD with a busy grid — ADVICE r5.  spin_limit keeps its unit: polls of nominally 2 us = 200
    // ticks of the 100 MHz clock; the default 2^20 is ~2 s.)
    auto await = [&](unsigned long long v, size_t idx, bool want) -> unsigned long long {
        uint32_t spins = 0;
        unsigned long long t_first = 0;
        while (__builtin_amdgcn_ballot_w64(want && v == kPipeEmpty) != 0ull) {
            __builtin_amdgcn_s_sleep(2);
            v = pipe_load(xin + idx);
            spins++;
            if ((spins & 63u) == 0u) {
                const unsigned long long now = wall_clock64();
                if (t_first == 0) t_first = now;
                if (now - t_first > 200ull * (unsigned long long)p.spin_limit ||
                    __hip_atomic_load(p.ctrl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) {
                    failed = true;
                    break;
                }
            }
        }
        return v;
    };

    // the first stage has nothing to its left: "no prefix yet" and H(., -1) = 0 in every row
    const unsigned long long kNoLeft = pipe_pack(kRowsNeg, 0);
    unsigned long long pend[kPipeDepth];
#pragma unroll
    for (int d = 0; d < kPipeDepth; d++) pend[d] = stage > 0 ? pipe_load(xin + min(d * kPipeBatch + sl, p.qlen)) : kNoLeft;
    for (int i0 = 0; i0 < p.qlen; i0 += kPipeBatch) {
        const int nrows = __builtin_amdgcn_readfirstlane(min(kPipeBatch, p.qlen - i0));
        unsigned long long cur = pend[0];
        if (stage > 0) {
            cur = await(cur, (size_t)min(i0 + sl, p.qlen), sl < nrows);
            if (failed) break;
#pragma unroll
            for (int d = 0; d + 1 < kPipeDepth; d++) pend[d] = pend[d + 1];
            pend[kPipeDepth - 1] = pipe_load(xin + min(i0 + kPipeDepth * kPipeBatch + sl, p.qlen));   // kPipeDepth batches ahead
        }
        const int curLo = (int)(uint32_t)cur, curHi = (int)(uint32_t)(cur >> 32);
        // lane r holds query letter i0 + r (letters behind the query's end are never used)
        const int qvec = (int)p.query[i0 + sl];
        for (int r = 0; r < nrows; r++) {
            const int rowv = tbl[__builtin_amdgcn_readlane(qvec, r)];   // (a uniform index into registers: v_movrels)
            const int carryIn = __builtin_amdgcn_readlane(curLo, r);
            const int hlIn = __builtin_amdgcn_readlane(curHi, r);   // H(i, col0 - 1) of lane 0: next row's diagonal input

            // ---- pass 1: F and H~ of the owned columns, G = H~ - c * gex (lane-local frame), m = their maximum
            int sc[CPL];
#pragma unroll
            for (int c = 0; c < CPL; c++) sc[c] = __builtin_amdgcn_ds_bpermute(lofs[c], rowv);
            int prevUp = hleft, m = kRowsNeg;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int up = H[c];
                const int f = max(F[c] + gex, up + gop);
                F[c] = f;
                const int ht = max(max(prevUp + sc[c], f), 0);
                prevUp = up;
                H[c] = ht;
                G[c] = ht - c * gex;
                m = max(m, G[c]);
            }
            // ---- inclusive prefix maximum over the wave, global frame (G - kg0), then P = what lies left of the lane: the
            //      lanes before it (wave_shr:1) and the stages before (carryIn; lane 0 has only that).  v_max_i32_dpp in place:
            //      a lane without a source, or in a row outside the mask, keeps its value.  (From update_dpp + max hipcc makes
            //      four instructions per step; the wait states a DPP read needs behind the VALU write are spelled out here.)
            int inc = m - kg0;
            int P = carryIn;
            asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "s_nop 1\n\tv_max_i32_dpp %1, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1"
                : "+v"(inc), "+v"(P));
            // ---- pass 2: E and H.  mm = running maximum in the lane-local frame, E(c) = mm + gop + (c - 1) * gex
            int mm = P + kg0;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const int e = mm + (gop + (c - 1) * gex);
                const int h = max(H[c], e);
                H[c] = h;
                best = max(best, h);
                mm = max(mm, G[c]);
            }
            const int hlast = H[CPL - 1];
            hleft = __builtin_amdgcn_update_dpp(hlIn, hlast, 0x138, 0xf, 0xf, false);   // lane 0 keeps the stage before's
            if (feeds && lane == 63) pipe_store(xout + i0 + r, pipe_pack(max(carryIn, inc), hlast));
        }
    }
    // ---- the subject's score: the running maximum travels down the pipeline behind the last row
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d));
    if (stage > 0 && !failed) {
        const unsigned long long fin = await(pipe_load(xin + p.qlen), (size_t)p.qlen, true);
        best = max(best, (int)(uint32_t)fin);
    }
    if (failed) {
        if (lane == 0) {
            __hip_atomic_store(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.fail_count) atomicAdd(p.fail_count, 1);
            p.scores[pos] = kPipeFailedScore;
            p.ids[pos] = (int32_t)(p.id_offset + pos);
        }
        return;
    }
    if (lane == 0) {
        if (feeds) {
            pipe_store(xout + p.qlen, pipe_pack(best, 0));
        } else {
            p.scores[pos] = (float)best;
            p.ids[pos] = (int32_t)(p.id_offset + pos);
            if (p.stat_count && best >= p.stat_limit) atomicAdd(p.stat_count, 1);
            if (p.stat_count2 && best >= p.stat_limit) atomicAdd(p.stat_count2, 1);
        }
    }
}

}  // namespace swk
