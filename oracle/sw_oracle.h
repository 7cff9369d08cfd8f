/* oracle/sw_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Smith-Waterman scoring path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this; the product
 * (cudasw4_amd/, include/) never links, imports or executes anything from oracle/.
 *
 * Parity pinning: see the header of sw_oracle.c.
 */
#ifndef SW_ORACLE_H
#define SW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* convert.cuh:6-34 — letter -> code (ARNDCQEGHILKMFPSTWYV -> 0..19, anything else -> 20). */
int8_t swo_encode_char(char c);
void swo_encode(const char* in, int8_t* out, size_t n);
/* convert.cuh:36-64 */
char swo_decode_char(int8_t code);

/* types.hpp:29-270 — 21x21 tables (which = 45|50|62|80). NULL for anything else. */
const int8_t* swo_blosum21(int which);

/* length_partitions.hpp:75-113 — 36 upper bounds; length L is in partition i iff b[i-1] < L <= b[i]. */
int swo_partition_boundaries(int32_t* out, int cap);
int swo_partition_of(int32_t length);

/* cudasw4.cuh:2331-2392 — scalar affine-gap local alignment score of two encoded sequences. */
int32_t swo_score(const int8_t* q, int32_t qlen, const int8_t* s, int32_t slen,
                  const int8_t* m21, int32_t gop, int32_t gex);

/* cudasw4.cuh:767-796 — score every subject of a dbdata-layout DB against one query (OpenMP). */
void swo_scan(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
              const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
              int32_t* scores, int nthreads);

/* Inter-sequence SIMD (GCC vector extension, 16 x int16 lanes, int32 re-score on saturation):
 * the multi-core CPU baseline bench.py reports.  Same results as swo_scan. */
void swo_scan_simd(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
                   const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
                   int32_t* scores, int nthreads);

/* Farrar's striped Smith-Waterman (the SSW algorithm): intra-sequence SIMD over the query with a striped query
 * profile and the lazy-F correction loop, int16 lanes, int32 re-score on saturation.  Second CPU baseline and a
 * third independent implementation of the recurrence.  Same results as swo_scan. */
void swo_scan_striped(const int8_t* q, int32_t qlen, const int8_t* chars, const uint64_t* offsets,
                      const int32_t* lengths, int64_t n, const int8_t* m21, int32_t gop, int32_t gex,
                      int32_t* scores, int nthreads);

/* dbdata.hpp:222-272 — the pseudo-DB subject: `length` codes drawn with std::mt19937(seed) and
 * std::uniform_int_distribution<>(0,19) (libstdc++ >= 11 algorithm). */
void swo_pseudodb_codes(int32_t length, uint32_t seed, int8_t* out);

/* cudasw4.cuh:1357-1401,1452-1458 — K best (score desc; ties by ascending id). */
void swo_topk(const int32_t* scores, int64_t n, int k, int32_t* out_scores, int64_t* out_ids);

int swo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
