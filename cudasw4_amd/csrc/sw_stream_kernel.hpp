// sw_stream_kernel.hpp — the packed scan kernels with the subjects STREAMED through the lanes: no pipeline drain between the
// subject pairs a group works on, and none between the stripes of a multi-stripe query either.
//
// sw_scan_kernel (sw_dp_kernel.hpp) gives a group one subject pair at a time: lane l works on column t - l, so every
// (pair, stripe) pays LANES - 1 steps in which part of the lanes sit on padding columns — 15 / (L + 15) of the steps: 10.5 % at
// L = 128, 2.9 % at L = 512, once per stripe.  The reference pays the same fill (half2_kernels.cuh:798-930: one alignment per
// thread group per pass).
//
// Here a workgroup claims a ROUND of up to 16 consecutive batches (SLOTS) and every group streams its slots' subject pairs
// through the lanes back to back, for every stripe of the query: the round is ONE long column sequence
//       [ slot 0 ] [sep][ slot 1 ] [sep][ slot 2 ] ...
// as far as the anti-diagonal wavefront, the letter words and — multi-stripe queries — the stripe-border rings and block
// transfers are concerned (all of them are indexed by the step, not by the subject), and the fill is paid once per round.
//
// What makes the switch cheap is the column-offset frame (dp_step<OFFS>): every value a lane holds is kept relative to a
// per-column ZERO LEVEL Z_j, E and F are clamped at their zero levels by the very max3 that computes them, and H never drops
// below E.  At its own switch step — the separator column in front of the next slot — a lane RAISES its zero levels by a
// constant `jump` that exceeds everything the finished subject left in its registers:
//       diag + s  <  Z',   E_old  <  Z',   F_in = Z' (exact, from the lane before / the border row / the lane-0 boundary)
//   =>  h = max3(diag + s, E_old, F) = Z' exactly, E' = max3(E_old, h + gop, Z'_next) = Z'_next exactly, F' likewise:
// after ONE ordinary step on a padding letter the lane is in the state "column -1 of a fresh subject" of the raised frame —
// no register is touched but the P + 4 zero levels of the window (P + 4 additions under a per-lane mask), plus the fold of the
// finished slot's running maxima into one stash register.  Round 5's streamed kernel re-initialised 2R + 2P + 13 registers per
// lane and switch and got 6.5 of the 15 fill steps back as switch steps; this form costs ~1.8 + the separator column.
// A subject whose score reaches `jump` may leave values above the raised level behind: its SUCCESSOR is flagged and
// re-scored by the 32-bit kind like an overflow (jump_limit; the flag is exact in the sense that it never misses: a
// reported score bounds every value the lane held).  The jumps use up range, so a round holds as many slots as fit
// a * columns + jumps <= stream_room (fp16 starts its levels at -2016, the bottom of its exact range, which doubles the room;
// int16 has room for thousands of columns), and the frame is never lowered inside a round of several slots; a round of ONE
// slot — a subject longer than the column budget — lowers its frame every K columns exactly like sw_scan_kernel.
//
// Slot widths are uniform per WAVE (the four groups of a wave run in lock-step: the DB is sorted by length, neighbours are
// equally long), not rounded to quads, at least 2 * LANES columns.  The letters of a round form one virtual stream; a lane's
// word of four letters is assembled from the slot(s) its four columns fall into (aligned loads + v_alignbyte_b32: the
// separator shifts every slot's letters by one more byte).  Scores leave slot by slot: once the last lane has switched, the
// group reduces the stashed maxima; multi-stripe queries keep slot k's running maximum in lane k of the group until the last
// stripe.  Rounds are claimed by guided self-scheduling — as many batches as the budgets allow while plenty are left,
// single batches at the end — so a launch's tail is no longer than with one batch at a time.
//
// Packed kinds, 16-lane groups, column-offset recurrence, plain [first_pos, first_pos + n) ranges.  Everything else — 32-bit
// kinds, position lists, the other group shapes — stays with sw_scan_kernel.
//
// What it buys, and what it does not (round 6, profiles/r06_stream_ab.txt, profiles/r06_results.md).  A switch is NOT free: the
// per-lane jump needs a per-step mask, which costs the plain path 5 % when it sits in the one loop body, so the event steps run
// in a second copy of the loop body (MODE 2; ~16 steps per slot border) and the transitions between the copies' register
// allocations cost ~120 scratch accesses — about 7 step equivalents per border against 15 fill steps + the quad rounding.
// Equal-length subjects (peak DB): +3 … +6.5 % at L = 128, +0 … +3.5 % at L = 256, ~+1 % from L = 384 up; rounds of FOUR slots
// do better than sixteen for the three-wave kernels (R <= 32) and about as well for the two-wave ones.  A real length
// distribution (Swiss-Prot-like, merged launch): nothing in the sum — short-subject batches gain, the 36 … 48-row kernels lose
// 2 … 5 %.  Hence the defaults (sw_api.hip: stream_plan): rounds of up to 4 slots; jumps of 512 (fp16) / 2048 (int16) so that a
// successor is flagged only behind a subject that scores that much (100 of 11 400 000 (query, subject) pairs on the
// Swiss-Prot-like DB; with jumps of 128 / 512 it was 1361, which kept the re-score service armed for every query); multi-stripe
// launches stream only while max_subject_len <= 320 (fp16) / 192 (int16) and use sw_scan_kernel beyond.  The flagged
// successors are counted apart (ScanParams::stat_count of a scan launch, sw_set_dirty_counter).
// Tried and dropped on the way: the jump at quad granularity (a lane still in the old frame takes its predecessor's raised
// values for a gap of `jump`: wrong scores — the separator would have to be >= 16 columns); a per-quad branch between the
// body copies (341 spilled registers); the inline per-step mask (-5 % everywhere).
// CUDASW4_AMD_STREAM=1 (test hook) gives every round one slot.
#pragma once
#include "sw_dp_kernel.hpp"

namespace swk {

constexpr int kStreamMaxSlots = 16;

template <int KIND, int R, int LANES, bool MULTI>
__global__ void __launch_bounds__(kThreads, (min_waves<KIND, R, LANES, MULTI>())) sw_scan_stream_kernel(const ScanParams p) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, LANES>;
    static_assert(A::kPacked && A::kWindow && LANES == 16, "streamed subjects: the packed windowed kernels on 16-lane groups");
    static_assert(kStreamMaxSlots <= LANES, "lane k of a group keeps slot k's running maximum");
    constexpr int kGroups = kThreads / LANES;
    constexpr int kWaves = kThreads / 64;
    using BD = Border<LANES>;
    constexpr int SHL1 = Shift<LANES>::kShl1;
    constexpr int kQuadsPerLetterBlock = LANES;  // a lane holds 4 letters: LANES quads per reload
    constexpr int P = frame_classes(true, R, LANES, MULTI);
    constexpr int kLastClass = (R - 1) % P;
    constexpr int kSubjPerBatch = kGroups * 2;
    constexpr int kMinWidth = 2 * LANES;        // a slot's columns: the last lane has left slot k - 1 before lane 0 leaves slot k
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 + G::kTileBytes];
    __shared__ __attribute__((aligned(16))) unsigned char rings[MULTI ? BD::ring_bytes(kGroups) : 16];
    __shared__ int slotBnd[kWaves][kStreamMaxSlots + 1];   // first column of every slot of the wave's stream (and its end)
    __shared__ int next_batch, claimed;

    const int tid = threadIdx.x;
    const int lane = tid & (LANES - 1);
    const int group = tid / LANES;
    const int wave = tid >> 6;
    const bool head = lane == 0;
    const u32 laneStep = 0x00100010u;
    if (p.start_signal && tid == 0) {
        if (atomicAdd(p.work_counter + 1, 1u) + 1u == p.start_quorum)
            __hip_atomic_fetch_add(p.start_signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int n = p.n;
    const int nbatches = (n + kSubjPerBatch - 1) / kSubjPerBatch;
    if ((int)blockIdx.x >= nbatches) return;
    if constexpr (!MULTI) load_tile<G::kTileBytes>(lds, p.profile);   // (the round loop starts with a barrier)

    unsigned char* const ringIn = rings + (MULTI ? group * BD::kGroupBytes : 0);
    unsigned char* const ringOut = ringIn + BD::kInBytes;
    u32* const gBorder = MULTI ? p.scratch + ((size_t)blockIdx.x * kGroups + group) * (size_t)border_region_words<LANES>(p.lcap) : nullptr;

    const int a = p.gex_mag;
    const u32 apos = A::pos_word(a);
    const u32 apos4 = A::pos_word(4 * a);
    // a lane starts every stripe "at column -lane": zero level base + a * (LANES - lane), + a per step
    const u32 zstart = A::level_word(p.level_base + a * (LANES - lane));
    const u32 zbefore = A::level_word(p.level_base + a * (LANES - lane - 1));
    const int smax = min(max(p.stream_slots, 1), kStreamMaxSlots);

    for (;;) {
        // ---- a round: as many batches as the budgets allow while plenty are left (guided self-scheduling), longest first
        __syncthreads();
        if (tid < 64) {
            // lane k of the first wave looks at the k-th batch the claim would start with: its longest subject is its last
            // one (the counter may have moved on by the time of the claim: then the batches are shorter ones, and the budgets
            // hold all the more).  A claim ends in front of the first batch with which the round's columns would exceed the
            // border scratch (stream_cols) or its levels the kind's range (a * columns + jumps > stream_room)
            int cur = 0;
            if (tid == 0) cur = (int)__hip_atomic_load(p.work_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cur = __shfl(cur, 0);
            const int want = max(1, min(smax, (nbatches - cur) / (int)gridDim.x));
            const int k = tid;
            const int batch = nbatches - 1 - (cur + k);
            const bool in = k < want && batch >= 0;
            int w = 0;
            if (in) w = max(p.lengths[p.first_pos + min(n - 1, batch * kSubjPerBatch + kSubjPerBatch - 1)] + (k > 0 ? 1 : 0), kMinWidth);
            int c = w;
#pragma unroll
            for (int d = 1; d < LANES; d <<= 1) {
                const int u = __shfl_up(c, d, LANES);
                if ((k & (LANES - 1)) >= d) c += u;
            }
            const bool ok = in && (k == 0 || (c <= p.stream_cols && a * c + k * p.jump <= p.stream_room));
            const unsigned long long m = __ballot(ok) & ((1ull << kStreamMaxSlots) - 1ull);
            const int take = max(1, __ffsll((long long)~m) - 1);   // batches in front of the first that does not fit
            if (tid == 0) {
                const int b = (int)atomicAdd(p.work_counter, (u32)take);
                // (the ONE claim whose range holds the value nbatches is the first that finds the counter dry: sw_set_dry_signal)
                if (b <= nbatches && b + take > nbatches && p.dry_signal)
                    __hip_atomic_store(p.dry_signal, p.dry_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                next_batch = b;
                claimed = take;
            }
        }
        __syncthreads();
        const int b0 = __builtin_amdgcn_readfirstlane(next_batch);
        if (b0 >= nbatches) break;
        const int S = __builtin_amdgcn_readfirstlane(min(claimed, nbatches - b0));   // uniform in the workgroup

        // ---- slot widths of this wave: lane k of every group looks at slot k — and keeps what the letter fetches of the round
        // need of it (where the group's two subjects start, how long they are): a fetch asks lane k of its group (ds_bpermute)
        // instead of walking lengths -> offsets -> chars in memory (three dependent loads in the middle of the quad loop
        // stalled the wave for microseconds at every letter block)
        u32 soff0 = 0, soff1 = 0;     // first letter of the slot's subjects, in words of four letters from p.chars
        u32 slens = 0;                // their lengths, 16 bits each (0: no such subject; launches with longer subjects are not streamed)
        {
            int slen0 = 0, slen1 = 0;
            if (lane < S) {
                const int batch = nbatches - 1 - (b0 + lane);
                const int i0 = batch * kSubjPerBatch + group * 2;
                if (i0 < n) { slen0 = p.lengths[p.first_pos + i0]; soff0 = (u32)((p.offsets[p.first_pos + i0] - p.offsets[0]) >> 2); }
                if (i0 + 1 < n) { slen1 = p.lengths[p.first_pos + i0 + 1]; soff1 = (u32)((p.offsets[p.first_pos + i0 + 1] - p.offsets[0]) >> 2); }
            }
            slens = (u32)slen0 | ((u32)slen1 << 16);
            int lmax = max(slen0, slen1);
            lmax = max(lmax, __shfl_xor(lmax, 16));
            lmax = max(lmax, __shfl_xor(lmax, 32));
            int w = lane < S ? max(lmax + (lane > 0 ? 1 : 0), kMinWidth) : 0;
            // inclusive prefix over the lanes of the row: end of slot `lane`
            int v = w;
#pragma unroll
            for (int d = 1; d < LANES; d <<= 1) {
                const int u = __shfl_up(v, d, LANES);
                if (lane >= d) v += u;
            }
            if ((tid & 63) < LANES) {
                if (lane == 0) slotBnd[wave][0] = 0;
                slotBnd[wave][lane + 1] = v;   // (slots >= S: the stream's end)
            }
        }
        __syncthreads();
        const int T = __builtin_amdgcn_readfirstlane(slotBnd[wave][S]);
        // (a DB that is not sorted by length can hand this wave longer subjects than the claim looked at: whatever the round's
        // budgets do not cover is flagged and re-scored instead of scored wrongly)
        const bool roundBad = S > 1 && (T > p.stream_cols || a * T + (S - 1) * p.jump > p.stream_room);
        // frame lowering: a round of one slot only (a subject longer than the budgets allow to combine)
        const int rq = S == 1 ? p.renorm_quads : 0;
        int nquads = (T + LANES - 1 + 3) >> 2;
        if constexpr (MULTI) nquads = min(nquads, (p.lcap - 4) >> 2);
        nquads = __builtin_amdgcn_readfirstlane(nquads);

        // the round's letters as ONE stream.  Word of subject `which` of slot k at the slot-local columns c0 .. c0 + 3
        // (c0 >= -1: the separator in front of slots > 0 shifts the letters by a byte): two aligned loads and a byte shift;
        // everything outside the subject's padded length reads as padding letters
        const int rowBase = (tid & 63) & ~(LANES - 1);   // lane 0 of this group within the wave
        auto slot_word = [&](int k, int c0, int which) -> u32 {
            u32 w = 0x14141414u;
            // (every lane of the wave takes part in the exchange; a slot beyond the round reads a lane with length 0)
            const u32 lens = (u32)__shfl((int)slens, rowBase + min(k, LANES - 1));
            const u32 off = (u32)__shfl((int)(which ? soff1 : soff0), rowBase + min(k, LANES - 1));
            const int len = (int)(which ? lens >> 16 : lens & 0xffffu);
            if (k < S && len > 0) {
                const int lenpad = (len + 3) & ~3;
                const int8_t* s = p.chars + ((size_t)off << 2);
                const int a0 = c0 & ~3, sh = c0 & 3;
                u32 lo = 0x14141414u, hi = 0x14141414u;
                if (a0 >= 0 && a0 < lenpad) lo = *reinterpret_cast<const u32*>(s + a0);
                if (sh != 0 && a0 + 4 < lenpad) hi = *reinterpret_cast<const u32*>(s + a0 + 4);
                w = __builtin_amdgcn_alignbyte(hi, lo, (u32)sh);
            }
            return w;
        };
        auto fetch2 = [&](int blk, u32& wa, u32& wb) {
            const int j0 = blk * (4 * LANES) + lane * 4;
            int k = 0;
            for (int i = 1; i < S; i++) k += (j0 >= slotBnd[wave][i]) ? 1 : 0;
            const int base = slotBnd[wave][k], nextb = slotBnd[wave][k + 1];
            const int c0 = j0 - base - (k > 0 ? 1 : 0);
            wa = slot_word(k, c0, 0);
            wb = slot_word(k, c0, 1);
            const int nv = nextb - j0;   // columns of this word that belong to slot k
            const bool straddles = nv < 4 && nv > 0 && k + 1 < S;
            if (__builtin_amdgcn_ballot_w64(straddles) != 0ull) {   // (wave-uniform: slot_word exchanges across lanes)
                // the word runs into the next slot: its separator, then its first letters
                const u32 na = slot_word(k + 1, -1, 0), nb = slot_word(k + 1, -1, 1);
                if (straddles) {
                    const u32 keep = (1u << (8 * nv)) - 1u;
                    wa = (wa & keep) | (na << (8 * nv));
                    wb = (wb & keep) | (nb << (8 * nv));
                }
            }
            wa *= (u32)G::kLetterUnits;
            wb *= (u32)G::kLetterUnits;
        };

        u32 slotAcc = 0u;          // lane k: running maximum (true scores) of slot k over the stripes done so far; its final value is
                                   // also what slot k + 1 needs to know (what this slot's lanes may have left behind)

        for (int stripe = 0; stripe < p.nstripes; stripe++) {
            const bool first = stripe == 0;
            const bool last = stripe + 1 == p.nstripes;
            if constexpr (MULTI) {
                __syncthreads();
                load_tile<G::kTileBytes>(lds, p.profile + (size_t)stripe * G::kTileBytes);
                __syncthreads();
            }
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
            st.yA = ((u32)(kPadLetter * G::kLetterUnits) << G::kLetterShift) + 16u * (u32)(lane + 1);
            st.yB = st.yA;
            st.yA |= st.yA << 16;

            u32 nextA, nextB, lettersA = 0, lettersB = 0;
            fetch2(0, nextA, nextB);
            u32 stashv = 0u;   // the lane's maximum (true scores) over the slot it left last

            // ---- stripe border (MULTI; Border<LANES>, as in sw_scan_kernel: everything is indexed by the step)
            const unsigned char* inPtr = ringIn;
            unsigned char* outPtr = (lane == LANES - 1) ? ringOut : rings + kGroups * BD::kGroupBytes + 8 * group;
            const u32 walkIn = 32u;   // bytes per quad; every lane reads along with lane 0 (one address per group: a broadcast)
            const u32 walkOut = (lane == LANES - 1) ? 32u : 0u;
            uint4 pend = make_uint4(0u, 0u, 0u, 0u);
            uint2 nxt = make_uint2(0u, 0u);
            const u32* const gIn = MULTI ? gBorder + 2 * (LANES - 1) + 4 * lane : nullptr;
            // What lane 0 takes in the FIRST stripe: columns j, j + 1 -> (H, F, H, F) with H the local-alignment boundary in the
            // column's frame — its zero level raised to the class of the lane's last row — and F the zero level itself (class 0:
            // "no vertical gap", and what a separator column needs).  The level of column j: base + a * (j [mod K] + LANES) + the
            // jumps of the slot borders up to j
            auto level_of = [&](int j) -> int {
                const int K = 4 * rq;
                int lv = p.level_base + a * ((K > 0 ? (j & (K - 1)) : j) + LANES);
                for (int i = 1; i < S; i++) lv += (j >= slotBnd[wave][i]) ? p.jump : 0;
                return lv;
            };
            auto first_stripe_pairs = [&](int j) -> uint4 {
                const int l0 = level_of(j), l1 = level_of(j + 1);
                return make_uint4(A::level_word(l0 + a * kLastClass), A::level_word(l0), A::level_word(l1 + a * kLastClass), A::level_word(l1));
            };
            if constexpr (MULTI) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (!first) {
                    const uint4 b0w = *reinterpret_cast<const uint4*>(gIn);
#if SWK_BORDER_PEND
                    pend = *reinterpret_cast<const uint4*>(gIn + BD::kBlockWords);
#endif
                    *reinterpret_cast<uint4*>(ringIn + 16 * lane) = b0w;
                } else {
                    *reinterpret_cast<uint4*>(ringIn + 16 * lane) = first_stripe_pairs(2 * lane);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                nxt = *reinterpret_cast<const uint2*>(inPtr);
            }
            auto block_end = [&](int blk) {
                int t = tid;
                asm volatile("" : "+v"(t));
                const int ln = t & (LANES - 1), grp = t / LANES;
                unsigned char* const rIn = rings + grp * BD::kGroupBytes + 16 * ln;
                u32* const gb = p.scratch + ((size_t)blockIdx.x * kGroups + grp) * (size_t)border_region_words<LANES>(p.lcap) + 4 * ln;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (!last) {
                    const uint4 v = *reinterpret_cast<const uint4*>(rIn + BD::kInBytes);
                    *reinterpret_cast<uint4*>(gb + (size_t)blk * BD::kBlockWords) = v;
                }
                outPtr -= walkOut * BD::kQuadsPerBlock;
                inPtr -= walkIn * BD::kQuadsPerBlock;
                if (first) {
                    *reinterpret_cast<uint4*>(rIn) = first_stripe_pairs((blk + 1) * BD::kBlockCols + 2 * ln);
                } else {
#if SWK_BORDER_PEND
                    *reinterpret_cast<uint4*>(rIn) = pend;
                    pend = *reinterpret_cast<const uint4*>(gb + 2 * (LANES - 1) + (size_t)(blk + 2) * BD::kBlockWords);
#else
                    *reinterpret_cast<uint4*>(rIn) = *reinterpret_cast<const uint4*>(gb + 2 * (LANES - 1) + (size_t)(blk + 1) * BD::kBlockWords);
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                nxt = *reinterpret_cast<const uint2*>(inPtr);
            };

            // ---- events of the stream: the LANES steps in which the lanes leave slot ksl - 1 one after the other (several
            // slots), or lower their frame (one slot: every K columns).  evq0 .. evq1: the quads that hold those steps
            int ksl = 1;
            int evb = S > 1 ? __builtin_amdgcn_readfirstlane(slotBnd[wave][1]) : (rq > 0 ? 4 * rq : 0x3fffffff);   // first step of the event
            int evq0 = evb >> 2, evq1 = (evb + LANES - 1) >> 2;

            // one lane's switch, right before the step in which it works on the separator column: lane `ksw`
            auto switch_lane = [&](int ksw) {
                u32 m = A::true_of(st.maxv[0], st.Zc[0]);
#pragma unroll
                for (int d = 1; d + 1 < P + 3; d += 2)
                    m = A::fold2(m, A::true_of(st.maxv[d], st.Zc[d]), A::true_of(st.maxv[d + 1], st.Zc[d + 1]));
                if constexpr ((P + 3) % 2 == 0) m = A::true_max(m, A::true_of(st.maxv[P + 2], st.Zc[P + 2]));
                const bool me = lane == ksw;
                stashv = me ? m : stashv;
                const u32 gw = me ? p.jump_word : 0u;
#pragma unroll
                for (int i = 0; i < P + 4; i++) st.Zc[i] = A::add(st.Zc[i], gw);
            };
            auto quad = [&](int q, auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;   // the copy of the loop body this is: 0 plain, 1 lanes lower their frame, 2 lanes switch slots
                if ((q & (kQuadsPerLetterBlock - 1)) == 0) {
                    lettersA = nextA; lettersB = nextB;
                    fetch2(q / kQuadsPerLetterBlock + 1, nextA, nextB);
                }
                const int lower_lane = 4 * (q & (rq - 1));
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
                auto border_step = [&](auto byte_tag) {
                    constexpr int BYTE = decltype(byte_tag)::value;
                    if constexpr (MODE == 2) switch_lane(4 * q + BYTE - evb);
                    if constexpr (MODE == 1) lower_frame(lower_lane + BYTE);
                    const uint2 in = nxt;
                    if constexpr (MULTI) {
                        if constexpr (BYTE == 3) {
                            inPtr += walkIn;
                            nxt = *reinterpret_cast<const uint2*>(inPtr);
                        } else {
                            nxt = *reinterpret_cast<const uint2*>(inPtr + 8 * (BYTE + 1));
                        }
                    }
                    dp_step<KIND, R, LANES, BYTE, MULTI, true, P, false>(st, lds, lettersA, lettersB, p.gop, p.gex, in.x, in.y, apos, first, p.wrap_class,
                                                                         p.wrap_last, head, laneStep);
                    if constexpr (MULTI) *reinterpret_cast<uint2*>(outPtr + 8 * BYTE) = make_uint2(st.Hlast, st.Fout);
                };
                border_step(std::integral_constant<int, 0>{});
                border_step(std::integral_constant<int, 1>{});
                border_step(std::integral_constant<int, 2>{});
                border_step(std::integral_constant<int, 3>{});
                if constexpr (MULTI) outPtr += walkOut;
                lettersA = dpp<SHL1, true>(0u, lettersA);
                lettersB = dpp<SHL1, true>(0u, lettersB);
#pragma unroll
                for (int k2 = 0; k2 < P + 4; k2++) st.Zc[k2] = A::add(st.Zc[k2], apos4);
#pragma unroll
                for (int d = 0; d < P + 3; d++) st.maxv[d] = A::add(st.maxv[d], apos4);
            };
            // slot k's maximum over this stripe from the lanes' maxima `lanemax` (true scores), kept in lane k of the group; the
            // scores leave after the round's last stripe (emit_slot below: nothing but this reduction sits between the loops)
            auto finish_slot = [&](int k, u32 lanemax) {
                u32 mv = lanemax;
                mv = A::true_max(mv, dpp<0x128, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x124, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x122, false>(mv, mv));
                mv = A::true_max(mv, dpp<0x121, false>(mv, mv));
                if (lane == k) slotAcc = (!MULTI || first) ? mv : A::true_max(slotAcc, mv);
            };
            // ---- the stream, block by block (MULTI: a block transfer behind every kQuadsPerBlock quads)
            constexpr int kBlockQuads = MULTI ? BD::kQuadsPerBlock : (1 << 28);
            for (int q0 = 0; q0 < nquads; q0 += kBlockQuads) {
                const int qend = MULTI ? min(nquads, q0 + kBlockQuads) : nquads;
                int q = q0;
                while (q < qend) {
                    // (a loop of its own per mode: the plain copy's quad is one basic block, which is what its speed rests on — the
                    // event code under a branch per STEP inside one body cost the plain path 5 %; a branch per quad between body
                    // copies inside one loop made the register allocator spill 300 registers)
                    const int e0 = min(qend, evq0);
                    for (; q < e0; q++) quad(q, std::integral_constant<int, 0>{});
                    if (q >= qend) break;
                    const int e1 = min(qend, evq1 + 1);
                    if (S > 1) {
                        for (; q < e1; q++) quad(q, std::integral_constant<int, 2>{});
                    } else {
                        for (; q < e1; q++) quad(q, std::integral_constant<int, 1>{});
                    }
                    if (q == evq1 + 1) {   // the event is over
                        if (S > 1) {
                            finish_slot(ksl - 1, stashv);
                            ksl++;
                            evb = ksl < S ? __builtin_amdgcn_readfirstlane(slotBnd[wave][ksl]) : 0x3fffffff;
                        } else {
                            evb += 4 * rq;
                        }
                        evq0 = evb >> 2; evq1 = (evb + LANES - 1) >> 2;
                    }
                }
                if constexpr (MULTI) {
                    if (qend == q0 + kBlockQuads) block_end(q0 / kBlockQuads);
                }
            }
            if constexpr (MULTI) {
                // LANES pairs of "no value" behind the last emitted position, the partial block leaves the OUT ring (as in
                // sw_scan_kernel; "no value": the kind's lowest pattern — nothing of it is ever used: the consumer's columns
                // behind the stream's end only feed padding cells of the last slot)
                if (!last) {
                    const int done = nquads & ~(BD::kQuadsPerBlock - 1);
                    const int slot = 4 * (nquads - done) + lane;
                    u32* const g = gBorder + (size_t)(done / BD::kQuadsPerBlock) * BD::kBlockWords;
                    const u32 none = A::level_word(p.level_base);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (slot < BD::kBlockCols) *reinterpret_cast<uint2*>(ringOut + 8 * slot) = make_uint2(none, none);
                    else *reinterpret_cast<uint2*>(g + 2 * slot) = make_uint2(none, none);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    *reinterpret_cast<uint4*>(g + 4 * lane) = *reinterpret_cast<const uint4*>(ringOut + 16 * lane);
                }
            }
            {   // the last slot: every lane's maxima are final
                u32 m = A::true_of(st.maxv[0], st.Zc[0]);
#pragma unroll
                for (int d = 1; d < P + 3; d++) m = A::true_max(m, A::true_of(st.maxv[d], st.Zc[d]));
                // (slots that the event loop did not get to finish — a stream cut short by lcap — are scored from what is there)
                for (; ksl < S; ksl++) finish_slot(ksl - 1, stashv);
                finish_slot(S - 1, m);
            }
            if constexpr (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        // ---- the round's scores: lane k of a group holds slot k's maximum over all stripes
        for (int k = 0; k < S; k++) {
            const u32 mv = (u32)__shfl((int)slotAcc, rowBase + k);
            const u32 pv = (u32)__shfl((int)slotAcc, rowBase + max(k - 1, 0));   // the slot before's reported scores
            const int sc0 = A::true_lo(mv), sc1 = A::true_hi(mv);
            const int prev0 = A::true_lo(pv), prev1 = A::true_hi(pv);
            // how high the frame of this slot's columns got: values stayed exact (and inside the kind's range) below the limit
            const int kend = __builtin_amdgcn_readfirstlane(slotBnd[wave][k + 1]);
            const int kcols = S > 1 ? kend : ((rq > 0 && 4 * nquads > 4 * rq) ? 4 * rq : 4 * nquads);
            const int ztop = p.level_base + a * (kcols + 2 * LANES + 4 + P) + k * p.jump;
            const int limit = min(A::kLimit, A::kLimit - ztop);
            // ... and what the slot before left behind stayed below the raised zero levels
            const bool dirty0 = roundBad || (k > 0 && prev0 >= p.jump_limit), dirty1 = roundBad || (k > 0 && prev1 >= p.jump_limit);
            if (lane == 0) {
                const int batch = nbatches - 1 - (b0 + k);
                const int i0 = batch * kSubjPerBatch + group * 2, i1 = i0 + 1;
                const int pos0 = p.first_pos + i0, pos1 = p.first_pos + i1;
                if (i0 < n) {
                    if (p.ovf_check && (sc0 >= limit || dirty0)) {
                        __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), pos0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.stat_count && sc0 < limit) atomicAdd(p.stat_count, 1);   // (listed for its predecessor's score only)
                    } else {
                        p.scores[pos0] = (float)sc0;
                    }
                    p.ids[pos0] = (int32_t)(p.id_offset + pos0);
                }
                if (i1 < n) {
                    if (p.ovf_check && (sc1 >= limit || dirty1)) {
                        __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), pos1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.stat_count && sc1 < limit) atomicAdd(p.stat_count, 1);
                    } else {
                        p.scores[pos1] = (float)sc1;
                    }
                    p.ids[pos1] = (int32_t)(p.id_offset + pos1);
                }
            }
        }
    }
}

}  // namespace swk
