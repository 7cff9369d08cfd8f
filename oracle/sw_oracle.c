/* oracle/sw_oracle.c — TEST INFRASTRUCTURE ONLY (checker + CPU baseline; never the product path).
 *
 * Plain-C restatement of the reference's CPU-side definition of the hot path:
 *
 *   swo_encode*            convert.cuh:6-34            (letter -> code)
 *   swo_blosum21           types.hpp:29-270            (tables generated from the reference, see gen_tables.py)
 *   swo_partition_*        length_partitions.hpp:11,75-113
 *   swo_score              cudasw4.cuh:2331-2392       (affine_local_DP_host_protein_blosum62_converted)
 *   swo_scan               cudasw4.cuh:767-796         (computeAllScoresCPU_blosum62: all subjects, OpenMP)
 *   swo_pseudodb_codes     dbdata.hpp:222-272          (PseudoDBdata: mt19937(seed) + uniform_int_distribution<>(0,19))
 *   swo_topk               cudasw4.cuh:1357-1401,1452-1458
 *
 * PARITY PINNING (tests/test_oracle.py, tests/golden/make_golden.py):
 *   - swo_score is checked against the reference's OWN scalar DP.  That function is a private
 *     member of a CUDA-only class, so oracle/Makefile slices it (by signature anchor, at build time,
 *     into the git-ignored oracle/_ref/) from /root/reference/src/cudasw4.cuh where it lies and
 *     compiles it with g++ against the reference's types.hpp; make_golden.py runs it on
 *     allqueries.fasta x {pseudo DBs, all-vs-all, random pairs} and commits the scores as
 *     tests/golden/*.json.  swo_score must reproduce every one of them.
 *   - tables, encoder, partition bounds, pseudo-DB residues and FASTA parsing are checked against
 *     oracle/_ref/libref_shim.so (reference headers compiled as they lie) and pinned as fixtures.
 *   - the reference repository itself ships no tests / known-answer vectors.
 */
#include "sw_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ encoding */

static const char SWO_LETTERS[21] = "ARNDCQEGHILKMFPSTWYV";

int8_t swo_encode_char(char c) {
    for (int i = 0; i < 20; i++)
        if (SWO_LETTERS[i] == c) return (int8_t)i;
    return 20;
}

void swo_encode(const char* in, int8_t* out, size_t n) {
    int8_t lut[256];
    for (int i = 0; i < 256; i++) lut[i] = 20;
    for (int i = 0; i < 20; i++) lut[(unsigned char)SWO_LETTERS[i]] = (int8_t)i;
    for (size_t i = 0; i < n; i++) out[i] = lut[(unsigned char)in[i]];
}

char swo_decode_char(int8_t code) { return (code >= 0 && code < 20) ? SWO_LETTERS[code] : '-'; }

/* ------------------------------------------------------------------ tables */

#define SW_BLOSUM_TABLE(which, low, ...) \
    static const int8_t swo_core_##which[400] = {__VA_ARGS__}; \
    static const int8_t swo_low_##which = (low);
#include "blosum_tables.inc"
#undef SW_BLOSUM_TABLE

static int8_t swo_tables[4][441];
static int swo_tables_ready = 0;

static void swo_expand(int8_t* dst, const int8_t* core, int8_t low) {
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 21; j++) dst[i * 21 + j] = (i < 20 && j < 20) ? core[i * 20 + j] : low;
}

const int8_t* swo_blosum21(int which) {
    if (!swo_tables_ready) {
#ifdef _OPENMP
#pragma omp critical(swo_tables_init)
#endif
        {
            swo_expand(swo_tables[0], swo_core_45, swo_low_45);
            swo_expand(swo_tables[1], swo_core_50, swo_low_50);
            swo_expand(swo_tables[2], swo_core_62, swo_low_62);
            swo_expand(swo_tables[3], swo_core_80, swo_low_80);
            swo_tables_ready = 1;
        }
    }
    switch (which) {
        case 45: return swo_tables[0];
        case 50: return swo_tables[1];
        case 62: return swo_tables[2];
        case 80: return swo_tables[3];
    }
    return NULL;
}

/* ------------------------------------------------------------------ length partitions */

int swo_partition_boundaries(int32_t* out, int cap) {
    int n = 0;
    int32_t b = 48;
#define SWO_PUSH(v) do { if (n < cap) out[n] = (v); n++; } while (0)
    SWO_PUSH(48);
    SWO_PUSH(64);
    for (b = 80; b <= 256; b += 16) SWO_PUSH(b);
    for (b = 288; b <= 512; b += 32) SWO_PUSH(b);
    for (b = 576; b <= 1280; b += 64) SWO_PUSH(b);
    SWO_PUSH(8000);
    SWO_PUSH(INT32_MAX - 1);
#undef SWO_PUSH
    return n;
}

int swo_partition_of(int32_t length) {
    int32_t b[64];
    int n = swo_partition_boundaries(b, 64);
    for (int i = 0; i < n; i++)
        if (length <= b[i]) return i;
    return n - 1;
}

/* ------------------------------------------------------------------ scalar Gotoh (the definition of "score") */

/* Same recurrence and boundary handling as cudasw4.cuh:2365-2390:
 *   E(i,j) = max(E(i,j-1)+gex, H(i,j-1)+gop)   reset to -10000 at the start of every row
 *   F(i,j) = max(F(i-1,j)+gex, H(i-1,j)+gop)   -10000 in row 0
 *   H(i,j) = max(0, H(i-1,j-1)+M[a][b], E, F)  0 on both borders
 *   score  = max H.
 * Rolling single-row storage instead of the reference's two-row arrays. */
int32_t swo_score(const int8_t* q, int32_t qlen, const int8_t* s, int32_t slen,
                  const int8_t* m21, int32_t gop, int32_t gex) {
    const int32_t NEG = -10000;
    if (qlen <= 0 || slen <= 0) return 0;
    int32_t* Hrow = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(slen + 1));
    int32_t* Frow = Hrow + (slen + 1);
    for (int32_t j = 0; j <= slen; j++) { Hrow[j] = 0; Frow[j] = NEG; }
    int32_t best = 0;
    for (int32_t i = 0; i < qlen; i++) {
        const int8_t* mrow = m21 + 21 * (int)q[i];
        int32_t diag = 0;      /* H(i-1, j-1) */
        int32_t left = 0;      /* H(i, j-1)   */
        int32_t E = NEG;
        for (int32_t j = 1; j <= slen; j++) {
            const int32_t up = Hrow[j];
            int32_t e = E + gex, e2 = left + gop;
            E = e > e2 ? e : e2;
            int32_t f = Frow[j] + gex, f2 = up + gop;
            const int32_t F = f > f2 ? f : f2;
            int32_t h = diag + mrow[(int)s[j - 1]];
            if (E > h) h = E;
            if (F > h) h = F;
            if (h < 0) h = 0;
            Frow[j] = F;
            Hrow[j] = h;
            diag = up;
            left = h;
            if (h > best) best = h;
        }
    }
    free(Hrow);
    return best;
}

void swo_scan(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
              const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
              int32_t* scores, int nthreads) {
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
#endif
    for (int64_t k = 0; k < n; k++)
        scores[k] = swo_score(q, qlen, chars + (offsets[k] - offsets[0]), lengths[k], m21, gop, gex);
    (void)nthreads;
}

/* ------------------------------------------------------------------ inter-sequence SIMD baseline */

/* W subjects advance in lock-step, one per int16 lane (SWIPE-style inter-sequence parallelism);
 * the compiler vectorises the fixed-width lane loops (AVX2/AVX-512 with -march=native).
 * Lanes whose score reaches SWO_I16_LIMIT are re-scored with swo_score (int32). */
#define SWO_W 32
#define SWO_I16_LIMIT 25000

typedef struct { int16_t v[SWO_W]; } swo_vec;

static void swo_scan_block(const int8_t* q, int32_t qlen, const int8_t* const* subj, const int32_t* slen,
                           int cnt, const int8_t* m21, int32_t gop, int32_t gex, int32_t* out,
                           swo_vec* Hcol, swo_vec* Ecol) {
    int32_t maxlen = 0;
    for (int l = 0; l < cnt; l++) if (slen[l] > maxlen) maxlen = slen[l];
    swo_vec best, colprof[21];
    for (int l = 0; l < SWO_W; l++) best.v[l] = 0;
    for (int32_t i = 0; i < qlen; i++)
        for (int l = 0; l < SWO_W; l++) { Hcol[i].v[l] = 0; Ecol[i].v[l] = -10000; }
    const int16_t g_o = (int16_t)gop, g_e = (int16_t)gex;
    for (int32_t j = 0; j < maxlen; j++) {
        int8_t letter[SWO_W];
        for (int l = 0; l < SWO_W; l++) letter[l] = (l < cnt && j < slen[l]) ? subj[l][j] : 20;
        for (int a = 0; a < 21; a++)
            for (int l = 0; l < SWO_W; l++) colprof[a].v[l] = m21[a * 21 + letter[l]];
        swo_vec diag, F, hup;
        for (int l = 0; l < SWO_W; l++) { diag.v[l] = 0; F.v[l] = -10000; hup.v[l] = 0; }
        for (int32_t i = 0; i < qlen; i++) {
            const int16_t* sc = colprof[(int)q[i]].v;
            int16_t* Hl = Hcol[i].v;   /* H(i, j-1) on entry, H(i, j) on exit */
            int16_t* El = Ecol[i].v;   /* E(i, j-1) on entry */
#pragma omp simd
            for (int l = 0; l < SWO_W; l++) {
                const int16_t hleft = Hl[l];
                int16_t e = (int16_t)(El[l] + g_e), e2 = (int16_t)(hleft + g_o);
                e = e > e2 ? e : e2;
                int16_t f = (int16_t)(F.v[l] + g_e), f2 = (int16_t)(hup.v[l] + g_o);
                f = f > f2 ? f : f2;
                int16_t h = (int16_t)(diag.v[l] + sc[l]);
                h = h > e ? h : e;
                h = h > f ? h : f;
                h = h > 0 ? h : 0;
                diag.v[l] = hleft;
                El[l] = e;
                F.v[l] = f;
                hup.v[l] = h;
                Hl[l] = h;
                best.v[l] = best.v[l] > h ? best.v[l] : h;
            }
        }
    }
    for (int l = 0; l < cnt; l++) {
        if (best.v[l] >= SWO_I16_LIMIT) out[l] = swo_score(q, qlen, subj[l], slen[l], m21, gop, gex);
        else out[l] = best.v[l];
    }
}

void swo_scan_simd(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
                   const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
                   int32_t* scores, int nthreads) {
    const int64_t nblocks = (n + SWO_W - 1) / SWO_W;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        swo_vec* Hcol = (swo_vec*)aligned_alloc(64, sizeof(swo_vec) * (size_t)(qlen > 0 ? qlen : 1));
        swo_vec* Ecol = (swo_vec*)aligned_alloc(64, sizeof(swo_vec) * (size_t)(qlen > 0 ? qlen : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t b = 0; b < nblocks; b++) {
            const int8_t* subj[SWO_W];
            int32_t slen[SWO_W];
            const int64_t first = b * SWO_W;
            const int cnt = (int)((n - first) < SWO_W ? (n - first) : SWO_W);
            for (int l = 0; l < cnt; l++) {
                subj[l] = chars + (offsets[first + l] - offsets[0]);
                slen[l] = lengths[first + l];
            }
            if (qlen <= 0) { for (int l = 0; l < cnt; l++) scores[first + l] = 0; continue; }
            swo_scan_block(q, qlen, subj, slen, cnt, m21, gop, gex, scores + first, Hcol, Ecol);
        }
        free(Hcol);
        free(Ecol);
    }
    (void)nthreads;
}

/* ------------------------------------------------------------------ Farrar striped baseline */

/* Striped layout (Farrar 2007): the query is cut into SWO_SW segments of seg = ceil(qlen/SWO_SW) rows; vector i
 * (0 <= i < seg) holds rows i, i+seg, i+2*seg, ... in its lanes.  One pass over the seg vectors per subject
 * letter computes H and E; the vertical gap F is propagated lane to lane afterwards by the lazy-F loop, which
 * (unlike the original SSE2 code) also refreshes E so that the result is exact for every gap setting. */
#if defined(__AVX512BW__)
#include <immintrin.h>
#define SWO_SW 32
typedef int16_t swo_v __attribute__((vector_size(64)));
static inline swo_v swo_vmax(swo_v a, swo_v b) { return (swo_v)_mm512_max_epi16((__m512i)a, (__m512i)b); }
static inline int swo_any_gt(swo_v a, swo_v b) { return _mm512_cmpgt_epi16_mask((__m512i)a, (__m512i)b) != 0; }
static inline swo_v swo_shift_in(swo_v v, int16_t first) {  /* lane l <- lane l-1, lane 0 <- first */
    const __m512i idx = _mm512_set_epi16(30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10,
                                         9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 0);
    swo_v r = (swo_v)_mm512_permutexvar_epi16(idx, (__m512i)v);
    r[0] = first;
    return r;
}
#elif defined(__AVX2__)
#include <immintrin.h>
#define SWO_SW 16
typedef int16_t swo_v __attribute__((vector_size(32)));
static inline swo_v swo_vmax(swo_v a, swo_v b) { return (swo_v)_mm256_max_epi16((__m256i)a, (__m256i)b); }
static inline int swo_any_gt(swo_v a, swo_v b) { return _mm256_movemask_epi8(_mm256_cmpgt_epi16((__m256i)a, (__m256i)b)) != 0; }
static inline swo_v swo_shift_in(swo_v v, int16_t first) {
    const __m256i lo = _mm256_permute2x128_si256((__m256i)v, (__m256i)v, 0x08);  /* [0, v.lo128] */
    swo_v r = (swo_v)_mm256_alignr_epi8((__m256i)v, lo, 14);
    r[0] = first;
    return r;
}
#else
#define SWO_SW 16
typedef int16_t swo_v __attribute__((vector_size(32)));
static inline swo_v swo_vmax(swo_v a, swo_v b) {
    const swo_v m = a > b;  /* all-ones lanes where a > b (C has no vector ?:) */
    return (a & m) | (b & ~m);
}
static inline swo_v swo_shift_in(swo_v v, int16_t first) {
    swo_v r;
    for (int l = SWO_SW - 1; l > 0; l--) r[l] = v[l - 1];
    r[0] = first;
    return r;
}
static inline int swo_any_gt(swo_v a, swo_v b) {
    swo_v m = a > b;
    for (int l = 0; l < SWO_SW; l++) if (m[l]) return 1;
    return 0;
}
#endif

static int32_t swo_striped_one(const swo_v* profile, int32_t seg, const int8_t* s, int32_t slen, int16_t go, int16_t ge,
                               swo_v* Hstore, swo_v* Hload, swo_v* E) {
    swo_v zero = {0}, vmaxv = {0}, vgo, vge;
    for (int l = 0; l < SWO_SW; l++) { vgo[l] = go; vge[l] = ge; }
    for (int32_t i = 0; i < seg; i++) { Hstore[i] = zero; E[i] = zero; }
    for (int32_t j = 0; j < slen; j++) {
        const swo_v* prow = profile + (size_t)s[j] * seg;
        swo_v vF = zero;
        swo_v vH = swo_shift_in(Hstore[seg - 1], 0);
        swo_v* t = Hload; Hload = Hstore; Hstore = t;
        for (int32_t i = 0; i < seg; i++) {
            vH = vH + prow[i];
            vH = swo_vmax(vH, E[i]);
            vH = swo_vmax(vH, vF);
            vH = swo_vmax(vH, zero);
            vmaxv = swo_vmax(vmaxv, vH);
            Hstore[i] = vH;
            const swo_v open = vH - vgo;
            E[i] = swo_vmax(swo_vmax(E[i] - vge, open), zero);   /* clamped like the GPU kernels: same H */
            vF = swo_vmax(swo_vmax(vF - vge, open), zero);
            vH = Hload[i];
        }
        /* lazy F: the F leaving a lane's last row enters the next lane's first row; propagate until the
         * chain equals what the first pass already computed (F - ge <= H - go in every lane) */
        for (int k = 0; k < SWO_SW; k++) {
            vF = swo_shift_in(vF, 0);
            int done = 0;
            for (int32_t i = 0; i < seg; i++) {
                swo_v h = Hstore[i];
                const int raised = swo_any_gt(vF, h);
                if (raised) {
                    h = swo_vmax(h, vF);
                    Hstore[i] = h;
                    vmaxv = swo_vmax(vmaxv, h);
                    E[i] = swo_vmax(E[i], swo_vmax(h - vgo, zero));
                }
                const swo_v open = swo_vmax(h - vgo, zero);
                const swo_v ext = swo_vmax(vF - vge, zero);
                /* H untouched here and the extended gap no better than a freshly opened one: identical to pass 1 */
                if (!raised && !swo_any_gt(ext, open)) { done = 1; break; }
                vF = swo_vmax(ext, open);
            }
            if (done) break;
        }
    }
    int16_t best = 0;
    for (int l = 0; l < SWO_SW; l++) if (vmaxv[l] > best) best = vmaxv[l];
    return best;
}

void swo_scan_striped(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
                      const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
                      int32_t* scores, int nthreads) {
    if (qlen <= 0) { for (int64_t k = 0; k < n; k++) scores[k] = 0; return; }
    const int32_t seg = (qlen + SWO_SW - 1) / SWO_SW;
    swo_v* profile = (swo_v*)aligned_alloc(64, sizeof(swo_v) * 21 * (size_t)seg);
    for (int c = 0; c < 21; c++)
        for (int32_t i = 0; i < seg; i++)
            for (int l = 0; l < SWO_SW; l++) {
                const int32_t row = i + l * seg;
                profile[(size_t)c * seg + i][l] = row < qlen ? m21[(int)q[row] * 21 + c] : m21[20 * 21 + c];
            }
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        swo_v* buf = (swo_v*)aligned_alloc(64, sizeof(swo_v) * 3 * (size_t)seg);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int64_t k = 0; k < n; k++) {
            const int8_t* s = chars + (offsets[k] - offsets[0]);
            int32_t sc = swo_striped_one(profile, seg, s, lengths[k], (int16_t)(-gop), (int16_t)(-gex), buf, buf + seg, buf + 2 * seg);
            if (sc >= SWO_I16_LIMIT) sc = swo_score(q, qlen, s, lengths[k], m21, gop, gex);
            scores[k] = sc;
        }
        free(buf);
    }
    free(profile);
    (void)nthreads;
}

/* ------------------------------------------------------------------ pseudo DB generator */

/* MT19937 (Matsumoto & Nishimura) == std::mt19937; seeding == std::mt19937(seed). */
typedef struct { uint32_t s[624]; int idx; } swo_mt;

static void swo_mt_seed(swo_mt* m, uint32_t seed) {
    m->s[0] = seed;
    for (int i = 1; i < 624; i++) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}

static uint32_t swo_mt_next(swo_mt* m) {
    if (m->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (m->s[i] & 0x80000000u) | (m->s[(i + 1) % 624] & 0x7fffffffu);
            uint32_t x = m->s[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) x ^= 0x9908b0dfu;
            m->s[i] = x;
        }
        m->idx = 0;
    }
    uint32_t y = m->s[m->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* libstdc++ (GCC >= 11) std::uniform_int_distribution<int>(0, range-1) on a 32-bit engine:
 * Lemire's nearly-divisionless method (bits/uniform_int_dist.h, _S_nd).  The reference's pseudo
 * DB content is therefore standard-library specific; this matches the g++ 11 of this image, the
 * same compiler oracle/_ref is built with. */
static uint32_t swo_uniform(swo_mt* m, uint32_t range) {
    uint64_t product = (uint64_t)swo_mt_next(m) * range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        const uint32_t threshold = (uint32_t)(-range) % range;
        while (low < threshold) {
            product = (uint64_t)swo_mt_next(m) * range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

void swo_pseudodb_codes(int32_t length, uint32_t seed, int8_t* out) {
    swo_mt m;
    swo_mt_seed(&m, seed);
    /* letters[dist(gen)] with letters == the encoder's alphabet, so the code IS the draw. */
    for (int32_t i = 0; i < length; i++) out[i] = (int8_t)swo_uniform(&m, 20);
}

/* ------------------------------------------------------------------ top-K */

typedef struct { int32_t score; int64_t id; } swo_hit;

static int swo_hit_cmp(const void* a, const void* b) {
    const swo_hit* x = (const swo_hit*)a;
    const swo_hit* y = (const swo_hit*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

void swo_topk(const int32_t* scores, int64_t n, int k, int32_t* out_scores, int64_t* out_ids) {
    swo_hit* h = (swo_hit*)malloc(sizeof(swo_hit) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) { h[i].score = scores[i]; h[i].id = i; }
    qsort(h, (size_t)n, sizeof(swo_hit), swo_hit_cmp);
    for (int i = 0; i < k; i++) {
        if (i < n) { out_scores[i] = h[i].score; out_ids[i] = h[i].id; }
        else { out_scores[i] = -1; out_ids[i] = -1; }
    }
    free(h);
}

int swo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
