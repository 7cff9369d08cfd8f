// sw_stream_kernel.hpp — the scan kernel for single-stripe queries with the subjects STREAMED through the lanes.
//
// sw_scan_kernel (sw_dp_kernel.hpp) gives a group one subject pair at a time: lane l works on column t - l, so every pair
// pays LANES - 1 steps in which part of the lanes sit on padding columns — the pipeline fills at its start and drains at its
// end: 15 / (L + 15) of the steps, 10.5 % at L = 128, 2.9 % at L = 512 (measured: 10 651 / 11 302 / 11 706 / 11 918 GCUPS at
// L = 128 / 256 / 512 / 2048 is R * L / (L + 16.4)).  The reference pays the same fill (half2_kernels.cuh:798-930: Q +
// thread_result steps per alignment).
//
// Here a workgroup claims up to four consecutive batches at once (a ROUND of up to four SLOTS per group) and every group
// streams its slots' subject pairs through the lanes back to back: when lane l has finished the last column of slot k - 1 it
// enters column 0 of slot k in the very next step, while the lanes behind it are still on slot k - 1.  What a lane carries
// over a slot border is nothing: at its own switch step it (i) converts the running maxima of the finished slot to a true
// score (`stash`), (ii) resets H, E, the diagonal input, the zero-level window and the maxima window to the state "column
// -1 of a fresh subject" — in the column-offset frame those are CONSTANTS of the launch (ScanParams::sw_levels: the frame of
// a column depends on the column alone, and every slot restarts it at column 0), written with v_mov from scalar registers.
// The switch steps are a third copy of the loop body (like the frame-lowering copy), LANES of them per slot; outside them
// nothing changes.  Cost per slot: LANES x (2R + 2P + 13) move-class instructions instead of LANES - 1 wasted steps of
// ~(6.5 R + 19): the bubble shrinks from 15 / (L + 15) to about 5 / (L + 5).
//
// All four groups of a wave (eight with 8-lane groups) switch together: a slot's width is the longest subject of the WAVE,
// rounded to whole quads (as sw_scan_kernel pads to the wave's longest subject: the DB is sorted by length, neighbours are
// equally long), at least 2 x LANES columns, and never ending inside the LANES columns behind a frame-lowering column.
// The letters of a round form one virtual stream; a lane's word of four letters comes from the slot its virtual column
// falls into (slot borders are wave-uniform).  Scores leave slot by slot: once the last lane has switched, the group reduces
// the stashed maxima and lane 0 writes / flags the slot's two subjects exactly like sw_scan_kernel does.
//
// Rounds are claimed by guided self-scheduling — four batches while plenty are left, fewer as the work counter runs out —
// so the tail of a launch stays what it is with single batches.
//
// Single-stripe queries, column-offset recurrence with windows (packed kinds, int32), 8- and 16-lane groups, plain
// [first_pos, first_pos + n) ranges.  Everything else — several stripes, position lists, the plain recurrence — stays with
// sw_scan_kernel.  CUDASW4_AMD_STREAM=0 turns it off (A/B measurements).
#pragma once
#include "sw_dp_kernel.hpp"

namespace swk {

constexpr int kStreamMaxSlots = 4;

struct StreamSlotMeta {       // one per (slot, group): 32 bytes in LDS
    unsigned long long s0, s1;  // first letter of the slot's subject(s)
    int32_t len0pad, len1pad;   // lengths rounded up to whole words of four letters (0: no such subject)
    int32_t pos0, pos1;         // positions (-1: none)
};

template <int KIND, int R, int LANES>
__global__ void __launch_bounds__(kThreads, (min_waves<KIND, R, LANES, false>())) sw_stream_kernel(const ScanParams p) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, LANES>;
    static_assert(A::kWindow && LANES <= 16, "streamed subjects: windowed column-offset kernels on 8- / 16-lane groups");
    constexpr int kGroups = kThreads / LANES;
    constexpr int kWaves = kThreads / 64;
    constexpr int SHL1 = Shift<LANES>::kShl1;
    constexpr int kQuadsPerLetterBlock = LANES;
    constexpr int kSwitchQuads = LANES / 4;
    constexpr int P = frame_classes(A::kPacked, R, LANES, false);
    constexpr int kLastClass = (R - 1) % P;
    constexpr int kSubjPerBatch = kGroups * A::kSubjects;
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 + G::kTileBytes];
    __shared__ __attribute__((aligned(16))) StreamSlotMeta meta[kStreamMaxSlots][kGroups];
    __shared__ int slotCols[kStreamMaxSlots][kWaves];
    __shared__ int next_batch, claimed;

    const int tid = threadIdx.x;
    const int lane = tid & (LANES - 1);
    const int group = tid / LANES;
    const int wave = tid >> 6;
    const bool head = lane == 0;
    const int slot16 = LANES == 8 ? (tid & 15) : lane;
    const u32 laneStep = (A::kPacked ? 0x00100010u : 16u) * ((LANES == 8 && (tid & 15) == 8) ? 9u : 1u);
    if (p.start_signal && tid == 0) {
        if (atomicAdd(p.work_counter + 1, 1u) + 1u == p.start_quorum)
            __hip_atomic_fetch_add(p.start_signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int n = p.n;
    const int nbatches = (n + kSubjPerBatch - 1) / kSubjPerBatch;
    if ((int)blockIdx.x >= nbatches) return;
    load_tile<G::kTileBytes>(lds, p.profile);

    const u32 apos = A::pos_word(p.gex_mag);
    const u32 apos4 = A::pos_word(4 * p.gex_mag);
    constexpr bool kLowers = A::kPacked;   // a 32-bit frame has room for any subject (sw_scan_kernel)
    const int rq = kLowers ? p.renorm_quads : 0;
    const u32 zstart = A::zero_at(p.gex_mag, LANES - lane);
    const u32 zbefore = A::zero_at(p.gex_mag, LANES - lane - 1);
    const int smax = min(max(p.stream_slots, 1), kStreamMaxSlots);
    // columns a workgroup walks in this launch if all of them get the same share: the chars of the range spread over the
    // groups of all workgroups; a claim takes at most a quarter of that (only thread 0 uses it)
    int colBudget = 1 << 30, peekLen = 1 << 30;
    if (tid == 0 && smax > 1) {
        const unsigned long long chars = p.offsets[p.first_pos + n] - p.offsets[p.first_pos];
        colBudget = (int)min((unsigned long long)(1 << 30), chars / ((unsigned long long)kSubjPerBatch * gridDim.x * 4ull));
        peekLen = p.lengths[p.first_pos + n - 1];   // the first claim sees the longest subject of the range
    }

    for (;;) {
        // a round: up to `smax` batches while plenty are left (guided self-scheduling), longest subjects first
        __syncthreads();
        if (tid == 0) {
            const int cur = (int)__hip_atomic_load(p.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // guided self-scheduling: a claim of about half of what is left per workgroup, so that the last claims are single
            // batches and the launch's tail is no longer than with one batch at a time — and never more COLUMNS than a
            // quarter of a workgroup's share of the launch: the subjects are sorted by length and the longest come first, a
            // round of four of the longest batches of a real DB is more than an average workgroup walks in the whole launch
            // (first form: 3.2 -> 7.6 ms for a 144-residue query on the Swiss-Prot-like DB)
            int want = max(1, min(smax, (nbatches - cur) / (2 * (int)gridDim.x)));
            want = max(1, min(want, colBudget / max(peekLen, 1)));
            // the longest subject of the batch the counter will roughly stand at when this workgroup claims again (loaded now,
            // used then: its latency hides behind the round)
            {
                const int ahead = min(nbatches - 1, cur + want + (int)gridDim.x);
                const int last = min(n - 1, (nbatches - 1 - ahead) * kSubjPerBatch + kSubjPerBatch - 1);
                peekLen = p.lengths[p.first_pos + max(last, 0)];
            }
            const int b = (int)atomicAdd(p.work_counter, (u32)want);
            // (the ONE claim whose range holds the value nbatches is the first that finds the counter dry: sw_set_dry_signal)
            if (b <= nbatches && b + want > nbatches && p.dry_signal)
                __hip_atomic_store(p.dry_signal, p.dry_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            next_batch = b;
            claimed = want;
        }
        __syncthreads();
        const int b0 = next_batch;
        if (b0 >= nbatches) break;
        const int S = min(claimed, nbatches - b0);

        // ---- the round's slots: subjects, widths
#pragma unroll
        for (int k = 0; k < kStreamMaxSlots; k++) {
            if (k < S) {
                const int batch = nbatches - 1 - (b0 + k);
                const int i0 = batch * kSubjPerBatch + group * A::kSubjects, i1 = i0 + 1;
                const bool valid0 = i0 < n, valid1 = A::kPacked && i1 < n;
                int pos0 = -1, pos1 = -1, len0 = 0, len1 = 0;
                const int8_t* s0 = p.chars;
                const int8_t* s1 = p.chars;
                if (valid0) { pos0 = p.first_pos + i0; len0 = p.lengths[pos0]; s0 = p.chars + (p.offsets[pos0] - p.offsets[0]); }
                if (valid1) { pos1 = p.first_pos + i1; len1 = p.lengths[pos1]; s1 = p.chars + (p.offsets[pos1] - p.offsets[0]); }
                int lmax = max(len0, len1);
                if constexpr (LANES == 8) lmax = max(lmax, __shfl_xor(lmax, 8));
                lmax = max(lmax, __shfl_xor(lmax, 16));
                lmax = max(lmax, __shfl_xor(lmax, 32));
                int cols = max((lmax + 3) & ~3, 2 * LANES);
                // a slot must not end inside the LANES columns behind a frame-lowering column: the lanes behind lane 0 would
                // switch before they have lowered
                if (rq > 0) {
                    const int K = 4 * rq, over = cols & (K - 1);
                    if (cols > K && over > 0 && over < LANES) cols += LANES - over;
                }
                if (lane == 0) {
                    StreamSlotMeta m;
                    m.s0 = (unsigned long long)s0; m.s1 = (unsigned long long)s1;
                    m.len0pad = (len0 + 3) & ~3; m.len1pad = (len1 + 3) & ~3;
                    m.pos0 = pos0; m.pos1 = pos1;
                    meta[k][group] = m;
                }
                if ((tid & 63) == 0) slotCols[k][wave] = cols;
            }
        }
        __syncthreads();
        // slot borders (virtual columns), uniform in the wave
        int bnd[kStreamMaxSlots + 1];
        bnd[0] = 0;
#pragma unroll
        for (int k = 0; k < kStreamMaxSlots; k++)
            bnd[k + 1] = __builtin_amdgcn_readfirstlane(bnd[k] + (k < S ? slotCols[k][wave] : 0));

        // ---- state of "column -lane of a fresh subject" (as at a stripe's start in sw_scan_kernel)
        StripeState<KIND, R, P> st;
        {
            u32 zc[P + 5];
            zc[0] = zbefore; zc[1] = zstart;
#pragma unroll
            for (int k = 2; k < P + 5; k++) zc[k] = A::add(zc[k - 1], apos);
#pragma unroll
            for (int r = 0; r < R; r++) { st.H[r] = zc[r % P]; st.E[r] = zc[r % P + 1]; }
            st.upH_prev = zc[(R - 1) % P]; st.Hlast = zc[(R - 1) % P]; st.Fout = zbefore;
#pragma unroll
            for (int k = 0; k < P + 4; k++) st.Zc[k] = zc[k + 1];
#pragma unroll
            for (int d = 0; d < P + 3; d++) st.maxv[d] = zc[d + 1];   // true score 0 in every frame
        }
        st.yA = ((u32)(kPadLetter * G::kLetterUnits) << G::kLetterShift) + 16u * (u32)(slot16 + 1);
        st.yB = st.yA;
        if constexpr (A::kPacked) st.yA |= st.yA << 16;

        // the round's letters as ONE stream: virtual column j lies in the slot k with bnd[k] <= j < bnd[k + 1]
        auto fetch2 = [&](int blk, u32& wa, u32& wb) {
            const int j = blk * (4 * LANES) + lane * 4;
            int k = 0;
#pragma unroll
            for (int i = 1; i < kStreamMaxSlots; i++) k += (i < S && j >= bnd[i]) ? 1 : 0;
            int base = 0;
#pragma unroll
            for (int i = 1; i < kStreamMaxSlots; i++) base = (i <= k) ? bnd[i] : base;
            const int c = j - base;
            const StreamSlotMeta& m = meta[k][group];
            wa = 0x14141414u;
            wb = 0x14141414u;
            if (j < bnd[kStreamMaxSlots]) {
                if (c < m.len0pad) wa = *reinterpret_cast<const u32*>(p.chars + ((long long)(m.s0 - (unsigned long long)p.chars) + c));
                if (A::kPacked && c < m.len1pad) wb = *reinterpret_cast<const u32*>(p.chars + ((long long)(m.s1 - (unsigned long long)p.chars) + c));
            }
            wa *= (u32)G::kLetterUnits;
            wb *= (u32)G::kLetterUnits;
        };
        u32 nextA, nextB, lettersA = 0, lettersB = 0;
        fetch2(0, nextA, nextB);
        u32 stash = 0;   // the finished slot's lane maximum, true scores

        // one lane's switch to the next slot, right before step BYTE of switch quad qs: lane 4 * qs + BYTE
        auto switch_lane = [&](int k, auto byte_tag) {
            constexpr int BYTE = decltype(byte_tag)::value;
            if (lane == k) {
                u32 m = A::true_of(st.maxv[0], st.Zc[0]);
#pragma unroll
                for (int d = 1; d < P + 3; d++) m = A::true_max(m, A::true_of(st.maxv[d], st.Zc[d]));
                stash = m;
                // the windows of a lane that is at column -BYTE in the quad's first step; sw_levels[j] = level a * (LANES - 4 + j)
#pragma unroll
                for (int i = 0; i < P + 4; i++) st.Zc[i] = p.sw_levels[i + 4 - BYTE];
#pragma unroll
                for (int d = 0; d < P + 3; d++) st.maxv[d] = p.sw_levels[d + 4 - BYTE];
#pragma unroll
                for (int r = 0; r < R; r++) { st.H[r] = p.sw_levels[r % P + 3]; st.E[r] = p.sw_levels[r % P + 4]; }
                st.upH_prev = p.sw_levels[kLastClass + 3];
            }
        };
        // q: quad of the round (letters); qrel: quad of the slot lane 0 is in (frame lowering, switch steps)
        auto quad = [&](int q, int qrel, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;   // 0: plain, 1: lanes lower their frame, 2: lanes switch slots
            if ((q & (kQuadsPerLetterBlock - 1)) == 0) {
                lettersA = nextA; lettersB = nextB;
                fetch2(q / kQuadsPerLetterBlock + 1, nextA, nextB);
            }
            const int lower_lane = 4 * (qrel & (rq - 1));
            auto lower_frame = [&](int k) {
                const u32 gw = lane == k ? p.renorm_word : 0u;
#pragma unroll
                for (int r = 0; r < R; r++) { st.H[r] = A::gap(st.H[r], gw); st.E[r] = A::gap(st.E[r], gw); }
                st.upH_prev = A::gap(st.upH_prev, gw);
#pragma unroll
                for (int k2 = 0; k2 < P + 4; k2++) st.Zc[k2] = A::gap(st.Zc[k2], gw);
#pragma unroll
                for (int d = 0; d < P + 3; d++) st.maxv[d] = A::gap(st.maxv[d], gw);
            };
            auto step = [&](auto byte_tag) {
                constexpr int BYTE = decltype(byte_tag)::value;
                if constexpr (MODE == 1) lower_frame(lower_lane + BYTE);
                if constexpr (MODE == 2) switch_lane(4 * qrel + BYTE, byte_tag);
                dp_step<KIND, R, LANES, BYTE, false, true, P>(st, lds, lettersA, lettersB, p.gop, p.gex, 0u, 0u, apos, false, p.wrap_class,
                                                              p.wrap_last, head, laneStep);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            lettersA = dpp<SHL1, true>(0u, lettersA);
            if constexpr (A::kPacked) lettersB = dpp<SHL1, true>(0u, lettersB);
#pragma unroll
            for (int k2 = 0; k2 < P + 4; k2++) st.Zc[k2] = A::add(st.Zc[k2], apos4);
#pragma unroll
            for (int d = 0; d < P + 3; d++) st.maxv[d] = A::add(st.maxv[d], apos4);
        };
        // the score of slot k from the lanes' stashed maxima: group maximum, overflow test, store (as sw_scan_kernel)
        auto finish_slot = [&](int k, int width, u32 lanemax) {
            u32 mv = lanemax;
            if constexpr (LANES == 8) {
                mv = half_row_max(mv, [](u32 a, u32 b) { return A::true_max(a, b); });
            } else {
                mv = A::true_max(mv, dpp<0x128, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x124, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x122, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x121, false>(mv, mv));
            }
            const int sc0 = A::true_lo(mv), sc1 = A::true_hi(mv);
            if (lane == 0) {
                const StreamSlotMeta& m = meta[k][group];
                const int cols = width + LANES + 4;
                const int guard = p.gex_mag * ((rq > 0 && cols > 4 * rq ? 4 * rq : cols) + 2 * LANES + 4 + P);
                if (m.pos0 >= 0) {
                    if (A::kPacked && p.ovf_check && sc0 >= A::kLimit - guard) {
                        __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), m.pos0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        p.scores[m.pos0] = (float)sc0;
                        if (p.stat_count && sc0 >= p.stat_limit) atomicAdd(p.stat_count, 1);
                    }
                    p.ids[m.pos0] = (int32_t)(p.id_offset + m.pos0);
                }
                if (A::kPacked && m.pos1 >= 0) {
                    if (p.ovf_check && sc1 >= A::kLimit - guard) {
                        __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), m.pos1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        p.scores[m.pos1] = (float)sc1;
                    }
                    p.ids[m.pos1] = (int32_t)(p.id_offset + m.pos1);
                }
            }
        };

        // ---- the round
        int q = 0, prevWidth = 0;
#pragma unroll 1
        for (int k = 0; k < S; k++) {
            int kbeg = 0, kend = 0;   // bnd[k], bnd[k + 1] without a dynamically indexed register array
#pragma unroll
            for (int i = 0; i < kStreamMaxSlots; i++) { kbeg = (i == k) ? bnd[i] : kbeg; kend = (i == k) ? bnd[i + 1] : kend; }
            // the last slot also takes the steps in which the lanes behind lane 0 finish it
            const int nq = __builtin_amdgcn_readfirstlane(((kend - kbeg) >> 2) + (k + 1 == S ? (LANES - 1 + 3) >> 2 : 0));
            int qrel = 0;
            if (k > 0) {
                for (; qrel < kSwitchQuads; qrel++, q++) quad(q, qrel, std::integral_constant<int, 2>{});
                // every lane has switched: slot k - 1 is complete
                finish_slot(k - 1, prevWidth, stash);
            }
            prevWidth = kend - kbeg;
            const int seg = (kLowers && rq > 0) ? rq : nq;
            while (qrel < nq) {
                const int segEnd = min(nq, (qrel / seg + 1) * seg);
                if constexpr (kLowers) {
                    if (qrel >= seg && (qrel & (seg - 1)) == 0) {
                        const int qlow = min(segEnd, qrel + kSwitchQuads);
                        for (; qrel < qlow; qrel++, q++) quad(q, qrel, std::integral_constant<int, 1>{});
                    }
                }
                for (; qrel < segEnd; qrel++, q++) quad(q, qrel, std::integral_constant<int, 0>{});
            }
        }
        {   // the last slot: every lane's maxima are final
            u32 m = A::true_of(st.maxv[0], st.Zc[0]);
#pragma unroll
            for (int d = 1; d < P + 3; d++) m = A::true_max(m, A::true_of(st.maxv[d], st.Zc[d]));
            finish_slot(S - 1, prevWidth, m);
        }
    }
}

}  // namespace swk
