// sw_stream_kernel.hpp — the single-stripe scan as a CONTINUOUS stream of subjects (16-lane groups).
//
// sw_scan_kernel pays LANES-1 idle steps per subject: the anti-diagonal pipeline of a group fills and drains for
// every alignment (15/(L+15) of the time — 10 % at L = 128).  Here a wave never drains: the next subject's
// letters follow the previous subject's immediately, separated by one quad of four special columns
//
//        [ SEP SEP pad pad ]
//
// A SEP column is letter 21 of the LDS tile: its substitution scores AND its gap scores are "minus infinity" for
// the arithmetic kind.  The gap scores therefore travel with the letter: every letter row of the tile carries
// two extra words (gap open, gap extend) that the lane reads along with its scores — no VALU cost.  With the
// clamped gap states of this design (E' = max(E,0), F' = max(F,0)) one SEP column forces E' = F' = 0 for every
// row and leaves H = E'_old (never above the old subject's maximum); the second forces H = 0.  After the pair the
// lane's whole DP state is the zero boundary of a fresh alignment, exactly what the reset at the top of
// sw_scan_kernel's subject loop produces, and the following pad columns keep it there.
//
// The running maximum is the only per-subject state that cannot be reset by a column: each lane hands its maximum
// over (and clears it) at a quad boundary after its own first SEP column and before its first column of the next
// subject — lanes 4m..4m+3 at the m-th boundary after the separator quad — and the subject's score is reduced and
// stored after the fourth.  Cost per subject: 4 steps + ~20 VALU instead of 15..18 idle steps and a state reset.
//
// Values that a SEP column cannot reset are the non-finite ones (fp16 +inf after a score beyond 65504, int16 NaN bit
// patterns after a score beyond 30719).  With the 21 x 21 BLOSUM tables they cannot occur here: a single-stripe
// query has at most 704 rows, so scores stay below 704 * 15.  With a custom matrix of large entries they can; they
// then persist into the following subjects of that group half, whose maxima also come out non-finite, i.e. above
// the overflow limit, so those subjects are flagged and re-scored in 32 bits like their predecessor — still exact,
// at the price of a few extra re-scores (tests/test_gpu_parity.py: test_stream_nonfinite_states).
//
// Waves are independent after the tile set-up: each wave pulls ROUNDS of 4 groups x (2|1) subjects from the
// launch's atomic counter (longest first), two rounds ahead of the one it computes, so that neither the counter
// nor the lengths/offsets/letter loads are ever waited for.
#pragma once

#include "sw_dp_kernel.hpp"

namespace swk {

template <int KIND> struct SepConst;
template <> struct SepConst<F16X2> { static constexpr u32 kScore = 0xFBFFFBFFu, kGap = 0xFBFFFBFFu; };  // -65504 (largest finite)
template <> struct SepConst<I16X2> { static constexpr u32 kScore = 0x80008000u, kGap = 0x80008000u; };  // +-32768: flips the sign bit of the pattern
template <> struct SepConst<I32>   { static constexpr u32 kScore = 0xC0000000u, kGap = 0xC0000000u; };  // -2^30
template <> struct SepConst<F32>   { static constexpr u32 kScore = 0xF1800000u, kGap = 0xF1800000u; };  // -2^100

// LDS tiles of the stream kernel, 22 letter rows each (21 = SEP), same chunk interleave as Geometry (chunk k of
// lane l at k*256 + l*16):
//   tile A (the group's first subject): a lane's row is [gap open, gap extend, NW score words] — the gap words
//          come with the first chunk, so the first rows of a step can start as soon as it has arrived;
//   tile B (packed kinds, second subject): the NW score words only.
template <int KIND, int R>
struct StreamGeometry {
    using G = Geometry<KIND, R, kGroup>;
    static constexpr bool kPacked = Arith<KIND>::kPacked;
    static constexpr int NW = G::NW;
    static constexpr int NWA = NW + 2;
    static constexpr int NCHA = (NWA + 3) / 4;
    static constexpr int NCHB = G::NCH;
    static constexpr int kChunkRowBytes = kGroup * 16;
    static constexpr int kRowBytesA = NCHA * kChunkRowBytes;
    static constexpr int kRowBytesB = NCHB * kChunkRowBytes;
    static constexpr int kSepLetter = kLetters;
    static constexpr int kTileBytesA = (kLetters + 1) * kRowBytesA;
    static constexpr int kTileBytesB = kPacked ? (kLetters + 1) * kRowBytesB : 0;
    static constexpr int kTileBytes = kTileBytesA + kTileBytesB;
    static constexpr int kUnitsA = kRowBytesA >> 8, kUnitsB = kRowBytesB >> 8;
    static_assert(kSepLetter * kUnitsA < 256, "letter offset must fit a byte");
    static constexpr u32 sep_word(int units) {  // [SEP SEP pad pad], premultiplied
        return (u32)(kSepLetter * units) * 0x0101u + (u32)(kPadLetter * units) * 0x01010000u;
    }
    static constexpr int kMinQuads = 4;  // a round is at least 16 columns long: the hand-over of the previous round's maxima takes 3 quads
};

// One anti-diagonal step; as dp_step (single stripe) with the gap scores taken from the letter's tile row.
template <int KIND, int R, int BYTE>
__device__ __forceinline__ void dp_step_stream(StripeState<KIND, R>& st, const unsigned char* tileA, const unsigned char* tileB,
                                               u32 lettersA, u32 lettersB) {
    using A = Arith<KIND>;
    using S = StreamGeometry<KIND, R>;
    constexpr int SHR1 = DPP_ROW_SHR1;
    constexpr u32 kSel = 0x0c0c000cu | ((u32)BYTE << 8);  // letter byte BYTE -> bits 15:8 (row offset = byte << 8)

    const u32 injA = __builtin_amdgcn_perm(0u, lettersA, kSel);
    st.yA = dpp<SHR1, false>(injA, st.yA) + 16u;
    // whole 16-byte chunks only: a ds_read_b32/_b64 at this 16-byte lane stride runs into 4-/2-way bank conflicts
    // and occupies the LDS as long as the conflict-free ds_read_b128
    u32 wa[4 * S::NCHA];
    lds_read_words<4 * S::NCHA, S::kChunkRowBytes>(wa, tileA + st.yA);
    u32 wb[4 * S::NCHB];
    if constexpr (A::kPacked) {
        const u32 injB = __builtin_amdgcn_perm(0u, lettersB, kSel);
        st.yB = dpp<SHR1, false>(injB, st.yB) + 16u;
        lds_read_words<4 * S::NCHB, S::kChunkRowBytes>(wb, tileB + st.yB);
    }
    const u32 gop = wa[0], gex = wa[1];

    u32 upH, F;
    if constexpr (A::kZero == 0u) {
        upH = dpp<SHR1, true>(0u, st.Hlast);
        F = dpp<SHR1, true>(0u, st.Fout);
    } else {
        upH = dpp<SHR1, false>(A::kZero, st.Hlast);
        F = dpp<SHR1, false>(A::kZero, st.Fout);
    }
    u32 diag = st.upH_prev;
    st.upH_prev = upH;

    u32 maxv = st.maxv;
#pragma unroll
    for (int r = 0; r < R; r++) {
        u32 s;
        if constexpr (A::kPacked) s = __builtin_amdgcn_perm(wb[r >> 1], wa[2 + (r >> 1)], (r & 1) ? 0x07060302u : 0x05040100u);
        else s = wa[2 + r];
        const u32 t = A::add(diag, s);
        diag = st.H[r];
        const u32 h = A::cell_h(t, st.E[r], F);
        const u32 hg = A::gap(h, gop);
        st.E[r] = A::gap_state(A::gap(st.E[r], gex), hg);
        F = A::gap_state(A::gap(F, gex), hg);
        st.H[r] = h;
        if (r & 1) maxv = A::fold2(maxv, st.H[r - 1], h);
        else if (r == R - 1) maxv = A::max2(maxv, h);
    }
    st.maxv = maxv;
    st.Hlast = st.H[R - 1];
    st.Fout = F;
    // keep the padding words of the last chunks "used": otherwise the compiler narrows those reads
#pragma unroll
    for (int w = S::NWA; w < 4 * S::NCHA; w++) asm volatile("" ::"v"(wa[w]));
    if constexpr (A::kPacked) {
#pragma unroll
        for (int w = S::NW; w < 4 * S::NCHB; w++) asm volatile("" ::"v"(wb[w]));
    }
}

template <int KIND, int R>
__global__ void __launch_bounds__(kThreads, (min_waves<KIND, R, kGroup, false>())) sw_stream_kernel(const ScanParams p) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, kGroup>;
    using S = StreamGeometry<KIND, R>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 + S::kTileBytes];

    const int tid = threadIdx.x;
    const int lane = tid & (kGroup - 1);
    const int wlane = tid & 63;
    const int n = p.n;
    constexpr int kSubjPerRound = 4 * A::kSubjects;
    const int nrounds = (n + kSubjPerRound - 1) / kSubjPerRound;

    // ---- tiles: built from the global profile tile (Geometry layout: NW score words per lane and letter)
    unsigned char* const tileA = lds;  // lane addresses carry a +16 bias, like sw_scan_kernel's
    unsigned char* const tileB = lds + S::kTileBytesA;
    {
        const u32* src = reinterpret_cast<const u32*>(p.profile);
        u32* a32 = reinterpret_cast<u32*>(tileA + 16);
        u32* b32 = reinterpret_cast<u32*>(tileB + 16);
        auto word_a = [&](int letter, int l, int w) -> u32& { return a32[(letter * S::kRowBytesA + (w >> 2) * S::kChunkRowBytes + l * 16 + (w & 3) * 4) >> 2]; };
        auto word_b = [&](int letter, int l, int w) -> u32& { return b32[(letter * S::kRowBytesB + (w >> 2) * S::kChunkRowBytes + l * 16 + (w & 3) * 4) >> 2]; };
        for (int i = tid; i < (kLetters + 1) * kGroup * S::NW; i += kThreads) {
            const int w = i % S::NW, l = (i / S::NW) % kGroup, letter = i / (S::NW * kGroup);
            const bool sep = letter == S::kSepLetter;
            const u32 v = sep ? SepConst<KIND>::kScore : src[(letter * G::kRowBytes + (w >> 2) * G::kChunkRowBytes + l * 16 + (w & 3) * 4) >> 2];
            word_a(letter, l, w + 2) = v;
            if constexpr (A::kPacked) word_b(letter, l, w) = v;
        }
        for (int i = tid; i < (kLetters + 1) * kGroup; i += kThreads) {
            const int letter = i / kGroup, l = i % kGroup;
            const bool sep = letter == S::kSepLetter;
            word_a(letter, l, 0) = sep ? SepConst<KIND>::kGap : p.gop;
            word_a(letter, l, 1) = sep ? SepConst<KIND>::kGap : p.gex;
        }
        __syncthreads();
    }

    // ---- per-wave round pipeline.  The metadata of the rounds in flight (current, next, the one after) lives in
    // a small LDS ring so that it costs no registers across the DP loop: lanes 0..7 of the wave load offset and
    // length of one subject each (3 VGPRs in flight) and publish them at the end of the round.
    __shared__ __attribute__((aligned(16))) u32 meta_ring[kThreads / 64][3][8][4];  // {offset lo, hi, length, -}
    u32 (*const ring)[8][4] = meta_ring[tid >> 6];
    const int my_subject = (wlane >> 4) * A::kSubjects;  // the group's first subject within a round
    const uint64_t offset0 = p.offsets[0];
    auto first_index = [&](int b) -> int { return b < nrounds ? (nrounds - 1 - b) * kSubjPerRound : n; };  // longest first
    struct MetaLoad { uint64_t off; int len; };
    auto issue_meta = [&](int b) -> MetaLoad {
        MetaLoad m;
        m.off = offset0; m.len = 0;
        const int i = first_index(b) + wlane;
        if (wlane < kSubjPerRound && i < n) {
            m.off = p.offsets[p.first_pos + i];
            m.len = p.lengths[p.first_pos + i];
        }
        return m;
    };
    auto commit_meta = [&](int slot, const MetaLoad& m) {
        if (wlane < kSubjPerRound)
            *reinterpret_cast<uint4*>(ring[slot][wlane]) = make_uint4((u32)m.off, (u32)(m.off >> 32), (u32)m.len, 0u);
        __builtin_amdgcn_wave_barrier();
    };
    auto quads_of = [&](int slot) -> int {  // wave-uniform number of data quads of a round
        int l = (int)ring[slot][my_subject][2];
        if constexpr (A::kPacked) l = max(l, (int)ring[slot][my_subject + 1][2]);
        l = max(l, __shfl_xor(l, 16));
        l = max(l, __shfl_xor(l, 32));
        return max(S::kMinQuads, __builtin_amdgcn_readfirstlane((l + 3) >> 2));
    };
    const int nwaves_ = (int)gridDim.x * (kThreads / 64);
    auto grab = [&]() -> u32 {  // next round of this wave; the value is read (readfirstlane) one round later
        u32 v = 0;
        if (wlane == 0) v = atomicAdd(p.work_counter, 1u) + 2u * (u32)nwaves_;
        return v;
    };
    // lane l holds letters 64*blk + 4l .. +3 of the round's column stream (subject, padding, separator quad),
    // premultiplied by kLetterUnits
    // A fetch only ISSUES the load (any use of the result is a wait for the memory latency); padding and the
    // separator quad are substituted when the word is consumed.  Branch-free on purpose (a load inside a divergent
    // branch is waited for on the spot): lanes past the end re-read the subject's first word (an empty subject:
    // the first word of the DB) and discard it.
    auto fetch = [&](int slot, int which, int blk) -> u32 {
        const uint4 m = *reinterpret_cast<const uint4*>(ring[slot][my_subject + which]);
        const int lenpad = ((int)m.z + 3) & ~3;
        const int j = blk * (4 * kGroup) + lane * 4;
        return *reinterpret_cast<const u32*>(p.chars + ((((uint64_t)m.y << 32) | m.x) - offset0) + (j < lenpad ? j : 0));
    };
    auto letters_of = [&](u32 raw, int slot, int which, int blk, int nq) -> u32 {
        const int lenpad = ((int)ring[slot][my_subject + which][2] + 3) & ~3;
        const int j = blk * (4 * kGroup) + lane * 4;
        const u32 units = which ? (u32)S::kUnitsB : (u32)S::kUnitsA;
        u32 w = j < lenpad ? raw * units : 0x14141414u * units;
        if (j == 4 * nq) w = which ? S::sep_word(S::kUnitsB) : S::sep_word(S::kUnitsA);
        return w;
    };

    // The first two rounds of a wave are assigned statically (wave w: rounds w and w + W) and the counter hands out
    // rounds from 2W on: 4096 waves starting with three dependent atomics each on one address cost ~120 us.
#ifdef SWK_TRACE  // per-wave timeline for tools/trace_stats.py (diagnostic builds only)
    const uint64_t t_start = wall_clock64();
    int rounds_done = 0;
#endif
    const int nwaves = (int)gridDim.x * (kThreads / 64);
    int b_cur = (int)blockIdx.x * (kThreads / 64) + (tid >> 6);
    if (b_cur >= nrounds) return;
    commit_meta(0, issue_meta(b_cur));
    int b_nxt = b_cur + nwaves;
    commit_meta(1, issue_meta(b_nxt));
    u32 ticket = grab();
    int slot_cur = 0;
    int nq_cur = quads_of(0);
    // letter words in flight: block 0 of the NEXT round (issued at the start of a round) and the next block
    // of the current round (issued one block, 64 columns, ahead)
    u32 headA = fetch(0, 0, 0);
    u32 headB = A::kPacked ? fetch(0, 1, 0) : 0u;
    u32 nextA = 0, nextB = 0;
    // vmcnt(0): with loads pending on entry, the loop's wait-count bookkeeping turns conservative (it then waits
    // for the freshly issued head loads in every round)
    __builtin_amdgcn_s_waitcnt(0x0F70);

    StripeState<KIND, R> st;
#pragma unroll
    for (int r = 0; r < R; r++) { st.H[r] = A::kZero; st.E[r] = A::kZero; }
    st.upH_prev = A::kZero; st.Hlast = A::kZero; st.Fout = A::kZero; st.maxv = A::kZero;
    st.yA = ((u32)(kPadLetter * S::kUnitsA) << 8) + 16u * (u32)(lane + 1);
    st.yB = ((u32)(kPadLetter * S::kUnitsB) << 8) + 16u * (u32)(lane + 1);
    u32 handed = A::kZero;  // the previous round's maximum of this lane, once handed over
    int pend_b = nrounds;   // the round whose scores are still to be stored
    int handovers = 4;      // hand-over boundaries passed since the last separator quad (4 = nothing pending)

    auto store_scores = [&]() {
        const u32 m = group_max<KIND, kGroup>(handed);
        const int pend_i0 = first_index(pend_b) + my_subject;
        if (lane == 0 && pend_i0 < n) {
            const int pos0 = p.first_pos + pend_i0;
            const int sc0 = A::score_lo(m);
            if (A::kPacked && p.ovf_check && sc0 >= A::kLimit) p.ovf_pos[atomicAdd(p.ovf_count, 1)] = pos0;
            else p.scores[pos0] = (float)sc0;
            p.ids[pos0] = (int32_t)(p.id_offset + pos0);
            if (A::kPacked && pend_i0 + 1 < n) {
                const int sc1 = A::score_hi(m);
                if (p.ovf_check && sc1 >= A::kLimit) p.ovf_pos[atomicAdd(p.ovf_count, 1)] = pos0 + 1;
                else p.scores[pos0 + 1] = (float)sc1;
                p.ids[pos0 + 1] = (int32_t)(p.id_offset + pos0 + 1);
            }
        }
    };

#ifdef SWK_TRACE
    const uint64_t t_loop = wall_clock64();
#endif
    for (;;) {
        const bool drain = b_cur >= nrounds;  // past the last round: three more quads complete the hand-over
        const int slot_nxt = slot_cur == 2 ? 0 : slot_cur + 1;
        const int slot_nxt2 = slot_nxt == 2 ? 0 : slot_nxt + 1;
        const int b_nxt2 = __builtin_amdgcn_readfirstlane((int)ticket);
        const int nq_nxt = quads_of(slot_nxt);
        const int total = drain ? 3 : nq_cur + 1;
        u32 lettersA = letters_of(headA, slot_cur, 0, 0, nq_cur);
        u32 lettersB = A::kPacked ? letters_of(headB, slot_cur, 1, 0, nq_cur) : 0u;
        headA = fetch(slot_nxt, 0, 0);
        if constexpr (A::kPacked) headB = fetch(slot_nxt, 1, 0);
        // unconditional (a block past the round's end just re-reads the subject's first word): a load assigned under
        // a condition reaches its loop-carried register through a copy, and the copy is a wait for the load
        nextA = fetch(slot_cur, 0, 1);
        if constexpr (A::kPacked) nextB = fetch(slot_cur, 1, 1);
        // the long-latency requests go out last: nothing is waited for until the first letter block is used up
        const MetaLoad inflight = issue_meta(b_nxt2);
        ticket = grab();
        for (int q = 0; q < total; q++) {
            if (q > 0 && (q & (kGroup - 1)) == 0) {
                const int blk = q >> 4;
                lettersA = letters_of(nextA, slot_cur, 0, blk, nq_cur);
                if constexpr (A::kPacked) lettersB = letters_of(nextB, slot_cur, 1, blk, nq_cur);
                nextA = fetch(slot_cur, 0, blk + 1);
                if constexpr (A::kPacked) nextB = fetch(slot_cur, 1, blk + 1);
            }
            if (q == nq_cur) handovers = 0;  // the separator quad
            dp_step_stream<KIND, R, 0>(st, tileA, tileB, lettersA, lettersB);
            dp_step_stream<KIND, R, 1>(st, tileA, tileB, lettersA, lettersB);
            dp_step_stream<KIND, R, 2>(st, tileA, tileB, lettersA, lettersB);
            dp_step_stream<KIND, R, 3>(st, tileA, tileB, lettersA, lettersB);
            lettersA = dpp<DPP_ROW_SHL1, true>(0u, lettersA);
            if constexpr (A::kPacked) lettersB = dpp<DPP_ROW_SHL1, true>(0u, lettersB);
            if (handovers < 4) {
                // lanes 4m..4m+3 are past their first SEP column and before their first column of the new round
                const bool mine = (lane >> 2) == handovers;
                handed = mine ? st.maxv : handed;
                st.maxv = mine ? A::kZero : st.maxv;
                if (++handovers == 4) store_scores();
            }
        }
        if (drain) break;
#ifdef SWK_TRACE
        rounds_done++;
#endif
        commit_meta(slot_nxt2, inflight);
        pend_b = b_cur;
        b_cur = b_nxt; nq_cur = nq_nxt; slot_cur = slot_nxt;
        b_nxt = b_nxt2;
    }
#ifdef SWK_TRACE
    if (wlane == 0 && p.scratch) {
        uint64_t* out = reinterpret_cast<uint64_t*>(p.scratch) + 4 * (size_t)(blockIdx.x * (kThreads / 64) + (tid >> 6));
        out[0] = t_start; out[1] = t_loop; out[2] = wall_clock64(); out[3] = (uint64_t)rounds_done | ((uint64_t)__smid() << 32);
    }
#endif
}

}  // namespace swk
