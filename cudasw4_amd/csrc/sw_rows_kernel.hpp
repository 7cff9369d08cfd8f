// sw_rows_kernel.hpp — ROW-PARALLEL Smith-Waterman for the few very long subjects of a real DB (partition 35: > 8000
// residues, up to 35 000 in Swiss-Prot).
//
// The scan kernels (sw_dp_kernel.hpp) give a subject to one alignment group — for these subjects one wave — which walks
// the subject column by column: 35 000 dependent steps per stripe of the query, 60 ms for a 5 478-residue query on one
// SIMD, whatever else the GPU does.  Beside the bulk launch of a whole DB that is hidden; on a SHARD of a real DB (what
// each of N GPUs gets) it is the floor of every query (tools/shard_proxy.sh), and for a short query it outlasts the bulk
// launch on one GPU already.  The reference has the same shape (one thread group per subject, cudasw4.cuh:2026-2103).
//
// Here one WORKGROUP of 1024 threads takes a subject and walks the QUERY row by row; every thread owns CPL consecutive
// subject columns, so a row is up to 40 960 cells computed at once by a whole CU.  The dependency inside a row — the
// horizontal gap E(i,j) = max(E(i,j-1) + gex, H(i,j-1) + gop) — is a max-plus prefix:
//     E(i,j) = gop + (j-1) gex + max_{k<j} ( H~(i,k) - k gex ),      H~ = max(0, diagonal + score, F)   (H without E)
// (exact for gop <= gex: opening a gap from a cell that was itself reached through E never beats extending that gap —
// the lazy-F argument of the striped CPU algorithms, turned to E), so a row is: H~ of the owned columns, ONE prefix maximum
// over the workgroup (wave scan + 16 wave totals through LDS, one barrier), then E and H.  int32 arithmetic throughout:
// the scores are the integers every other kind produces (bit-exact against the oracle; tests/test_gpu_rows.py).
// 14 VALU instructions per cell instead of 6.5, but a whole CU per subject instead of one SIMD lane group: the
// 35 000-residue subject takes 3.5 us per query row — 20 ms for the 5 478-residue query, 0.2 ms for a 48-residue one.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace swk {

struct RowsParams {
    const int8_t* chars;       // subject letters; subject pos starts at chars + (offsets[pos] - offsets[0])
    const uint64_t* offsets;
    const int32_t* lengths;
    int32_t first_pos;         // workgroup b takes subject first_pos + b
    const int8_t* query;       // letter codes 0 .. dim-1
    int32_t qlen;
    const int8_t* matrix;      // (dim + 1) x 21 substitution scores, row = query letter
    int32_t dim;
    int32_t gop, gex;          // <= 0, gop <= gex
    float* scores;
    int32_t* ids;
    int64_t id_offset;
    uint32_t* start_counter;   // start handshake (sw_set_start_signal): resident workgroups are counted here ...
    uint32_t* start_signal;    // ... and the one that completes start_quorum raises the signal
    uint32_t start_quorum;
};

constexpr int kRowsThreads = 1024;
constexpr int kRowsWaves = kRowsThreads / 64;
constexpr int kRowsMaxCpl = 40;
constexpr int kRowsMaxSubject = kRowsThreads * kRowsMaxCpl;   // 40 960 residues
constexpr int kRowsSubCols = 22;                              // 21 subject letters + the column of a position behind the subject's end
constexpr int kRowsQueryChunk = 4096;
constexpr int kRowsNeg = -(1 << 29);

// a DPP move whose lanes without a source (and rows outside ROW_MASK) receive kRowsNeg
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int rows_dpp(int v) {
    return __builtin_amdgcn_update_dpp(kRowsNeg, v, CTRL, ROW_MASK, 0xf, false);
}

template <int CPL>  // a multiple of 4
__global__ void __launch_bounds__(kRowsThreads) sw_rows_kernel(const RowsParams p) {
    __shared__ int sub[26 * kRowsSubCols];
    __shared__ int xT[2][kRowsWaves], xX[2][kRowsWaves], xM1[2][kRowsWaves], xH[2][kRowsWaves];
    __shared__ int wbest[kRowsWaves];
    __shared__ int8_t qbuf[kRowsQueryChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.start_signal && tid == 0) {
        if (atomicAdd(p.start_counter, 1u) + 1u == p.start_quorum)
            __hip_atomic_fetch_add(p.start_signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // substitution scores as LDS words; a position behind the subject's end scores -30000 against everything, so nothing
    // positive is ever created there (what flows in from the left or from above is smaller than where it came from)
    for (int t = tid; t < (p.dim + 1) * kRowsSubCols; t += kRowsThreads) {
        const int r = t / kRowsSubCols, c = t % kRowsSubCols;
        sub[t] = c < 21 ? (int)p.matrix[r * 21 + c] : -30000;
    }
    const int pos = p.first_pos + (int)blockIdx.x;
    const int len = p.lengths[pos];
    const int8_t* s = p.chars + (p.offsets[pos] - p.offsets[0]);
    const int col0 = tid * CPL;  // first owned column (0-based)
    // byte offsets of the owned letters' columns in a row of `sub`, four per word
    constexpr int NLW = (CPL + 3) / 4;
    uint32_t lw[NLW];
#pragma unroll
    for (int w = 0; w < NLW; w++) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c = 4 * w + b, col = col0 + c;
            int letter = 21;
            if (c < CPL && col < len) {
                letter = (int)s[col];
                if (letter < 0 || letter > 20) letter = 20;
            }
            word |= (uint32_t)(letter * 4) << (8 * b);
        }
        lw[w] = word;
    }
    int H[CPL], F[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) { H[c] = 0; F[c] = -10000; }
    int best = 0;
    int hleft = 0;  // H(i-1, col0 - 1): the diagonal input of the first owned column
    const int gop = p.gop, gex = p.gex;
    const int kg0 = (col0 + 1) * gex;  // k * gex of the first owned column (k counts from 1)

    for (int i = 0; i < p.qlen; i++) {
        if ((i & (kRowsQueryChunk - 1)) == 0) {  // the next 4096 query letters (uniform branch)
            __syncthreads();
            for (int t = tid; t < kRowsQueryChunk && i + t < p.qlen; t += kRowsThreads) qbuf[t] = p.query[i + t];
            __syncthreads();
        }
        const int qi = __builtin_amdgcn_readfirstlane((int)qbuf[i & (kRowsQueryChunk - 1)]);
        const char* const srow = reinterpret_cast<const char*>(sub) + qi * (kRowsSubCols * 4);

        // wide kernels: keep the letters packed (the compiler would otherwise hoist all CPL extracted offsets out of the row
        // loop, into registers the 128-VGPR budget of a 1024-thread workgroup does not have)
        if constexpr (CPL > 16) {
#pragma unroll
            for (int w = 0; w < NLW; w++) asm volatile("" : "+v"(lw[w]));
        }
        // ---- pass 1: F and H~ of the owned columns; m = max over them of G' = H~ - c * gex (lane-local frame)
        int m = kRowsNeg, m1 = kRowsNeg, prevUp = hleft;
        // eight (four) columns at a time: their substitution scores are requested together, and a scheduling fence between the
        // groups keeps the compiler from requesting all CPL of them at once (CPL more live registers)
        constexpr int kChunk = CPL % 8 == 0 ? 8 : 4;
#pragma unroll
        for (int c0 = 0; c0 < CPL; c0 += kChunk) {
            int sc[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                const int c = c0 + u;
                const int lofs = (int)((lw[c >> 2] >> (8 * (c & 3))) & 0xffu);
                sc[u] = *reinterpret_cast<const int*>(srow + lofs);
            }
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                const int c = c0 + u;
                const int up = H[c];
                const int f = max(F[c] + gex, up + gop);
                F[c] = f;
                const int ht = max(max(prevUp + sc[u], f), 0);
                prevUp = up;
                H[c] = ht;
                if (c == CPL - 1) m1 = m;
                m = max(m, ht - c * gex);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // true frame: G = G' - kg0
        const int M = m - kg0, M1 = m1 - kg0;
        // ---- prefix maximum over the workgroup: inclusive scan over the wave, wave totals through LDS
        // (DPP: shifts inside the rows of 16 lanes, then lane 15 / 31 of the rows before; lanes without a source get kRowsNeg)
        int inc = M;
        inc = max(inc, rows_dpp<0x111, 0xf>(inc));  // row_shr:1
        inc = max(inc, rows_dpp<0x112, 0xf>(inc));  // row_shr:2
        inc = max(inc, rows_dpp<0x114, 0xf>(inc));  // row_shr:4
        inc = max(inc, rows_dpp<0x118, 0xf>(inc));  // row_shr:8
        inc = max(inc, rows_dpp<0x142, 0xa>(inc));  // row_bcast:15 into rows 1 and 3
        inc = max(inc, rows_dpp<0x143, 0xc>(inc));  // row_bcast:31 into rows 2 and 3
        const int exc = rows_dpp<0x138, 0xf>(inc);  // wave_shr:1
        const int par = i & 1;
        const int htLast = H[CPL - 1];
        if (lane == 63) { xT[par][wave] = inc; xX[par][wave] = exc; xM1[par][wave] = M1; xH[par][wave] = htLast; }
        const int l_ht = rows_dpp<0x138, 0xf>(htLast), l_m1 = rows_dpp<0x138, 0xf>(M1), l_exc = rows_dpp<0x138, 0xf>(exc);
        __syncthreads();
        // the totals of the waves before this one (pw) and before the one before (pw1): lane l of every row of 16 lanes
        // reads wave l's total, the row reduces (one register instead of sixteen)
        const int tv = xT[par][lane & 15];
        int pw = (lane & 15) < wave ? tv : kRowsNeg, pw1 = (lane & 15) < wave - 1 ? tv : kRowsNeg;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            pw = max(pw, __shfl_xor(pw, d));
            pw1 = max(pw1, __shfl_xor(pw1, d));
        }
        const int P = max(pw, exc);
        // H(i, col0 - 1) for the next row: the left neighbour's last column, its E from what lies before that column
        int hleftNext = 0;
        if (tid > 0) {
            int q, hl;
            if (lane > 0) { q = max(pw, max(l_exc, l_m1)); hl = l_ht; }
            else { q = max(pw1, max(xX[par][wave - 1], xM1[par][wave - 1])); hl = xH[par][wave - 1]; }
            hleftNext = max(hl, gop + (col0 - 1) * gex + q);
        }
        // ---- pass 2: E and H.  mm = running maximum in the lane-local frame (true value + kg0), so that
        //      E(c) = mm + gop + (c - 1) * gex  with wave-uniform constants
        int mm = P + kg0;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            int ht = H[c];
            asm volatile("" : "+v"(ht));  // (recompute H~ - c * gex here instead of keeping pass 1's CPL values in registers)
            const int e = mm + (gop + (c - 1) * gex);
            const int h = max(ht, e);
            H[c] = h;
            best = max(best, h);
            mm = max(mm, ht - c * gex);
            if constexpr (CPL > 16) {
                if ((c & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (bounds the temporaries in flight, as in pass 1)
            }
        }
        hleft = hleftNext;
    }
    // ---- maximum over the workgroup
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d));
    if (lane == 0) wbest[wave] = best;
    __syncthreads();
    if (tid == 0) {
        int b = 0;
#pragma unroll
        for (int v = 0; v < kRowsWaves; v++) b = max(b, wbest[v]);
        p.scores[pos] = (float)b;
        p.ids[pos] = (int32_t)(p.id_offset + pos);
    }
}

}  // namespace swk
