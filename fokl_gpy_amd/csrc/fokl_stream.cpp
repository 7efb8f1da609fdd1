// The numpy-legacy random stream of a fit, split into a BULK phase and a serial WALK (round 4).
//
// What the reference consumes (FoKLRoutines.py:1527, 1541, 1547: np.random.normal / np.random.gamma, three calls per Gibbs
// iteration on numpy's global legacy RandomState) is one MT19937 word stream read as 53-bit doubles, paired up by the
// polar method (one-value cache shared by normal and gamma) and by Marsaglia-Tsang's gamma (fokl_sampler.cpp restates the
// published algorithm and is pinned against numpy bit for bit).  Up to round 3 one thread produced all of it, draw by
// draw: 31 + 0.93 p1 ns per Gibbs iteration, 70 ms of an 80 ms fit.  Only a sliver of that is sequential:
//
//   * the WORDS are a linear recurrence, w[g] = w[g - 227] ^ twist(w[g - 624], w[g - 623]), whose shortest dependency is
//     227 words long: over a flat array it is a plain vector loop;
//   * tempering, the conversion to doubles, x = 2 d - 1, x^2 and the ACCEPT FLAG of the polar attempt that STARTS at each
//     double (0 < x_d^2 + x_{d+1}^2 < 1) are element-wise.  A uniform drawn by a gamma shifts the pairing of the doubles
//     by one, so the flags are kept for both alignments (attempts starting at even and at odd doubles), as two bit masks;
//   * what is left for the walk: "advance over ceil((p1 - cached) / 2) accepted attempts" -- popcounts over the mask
//     of the current alignment -- and the two gamma draws of the iteration (one normal, one uniform and a data-independent
//     accept test each).  The normals of an iteration are not touched by the walk at all: a tape row is (where the scan
//     began, the leading cached normal, the trailing half pair, the two gammas) -- 40 bytes instead of 12 p1.
//
// Bulk threads produce SEGMENTS of the stream (256 MT19937 blocks: 79 872 doubles) ahead of the walker: the recurrence of
// a segment runs under a token (it continues the previous segment's last block: 0.15 cycles per word), everything else
// of the segment -- tempering in place, flags -- runs outside it, so several threads overlap.  Consumers turn tape rows
// back into normals (fokl_stream_expand: any thread, any time while the tape's segments are held): the mask says which
// attempts were accepted, the tempered words give (x2, x1).
//
// Exactness: every double, every flag and every draw of the walk is computed with the operations fokl_sampler.cpp's
// LegacyRng uses (the conversion and x = 2 d - 1 are exact; x^2 and the sum are one rounding each, no contraction: this
// file is compiled with -ffp-contract=off), so positions, consumption and values are numpy's bit for bit
// (tests/test_stream_engine.py: against the one-thread recorder, against numpy itself, odd word positions, rewinds).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include <immintrin.h>
#include <sched.h>
#include <deque>

#include "../../include/fokl_hip_internal.h"
#include "fokl_spin.h"

extern void fokl_set_global_error(const std::string &msg);   // fokl_hip.hip
extern "C" __attribute__((visibility("hidden"))) void fokl_note_thread_cpu(int kind);   // fokl_hostpool.cpp

namespace {

constexpr int MT_N = 624, MT_SHIFT = 227;                   // 227 = 624 - 397
constexpr int kSegBlocks = FOKL_SEGMENT_BLOCKS;
constexpr int kSegWords = kSegBlocks * MT_N;                // 159 744 words
constexpr int kSegDoubles = kSegWords / 2;                  // 79 872 doubles (attempts may start at each of them)
static_assert(kSegDoubles == FOKL_SEGMENT_DOUBLES, "include/fokl_hip_internal.h and this file disagree on the segment size");
constexpr int kSegSlots = kSegDoubles / 2;                  // 39 936 attempts per alignment
constexpr int kSegMaskWords = kSegSlots / 64;               // 624 mask words per alignment
constexpr int kSegTail = 32;                                // words of the next segment kept behind this one's
constexpr int kTable = 8192;                                // segments that can be alive at a time (524 M doubles)
constexpr int kAhead = 24;                                  // segments produced ahead of the walker (a 585-column tape is 19)
// ... to begin with.  A walker that keeps arriving at segments not made yet is production bound (configs[3]: 1 490 doubles per
// Gibbs iteration, a segment per microsecond of walking) while the bulk threads sleep through every pause of the walk: the
// distance grows with the time the walker has waited -- 8 segments per half millisecond, up to kAheadMax -- so that pauses
// are used; a fit whose walker hardly waits (configs[2]: 0.3 ms) stays at kAhead and wastes nothing at its end (round 6).
constexpr int kAheadMax = 192;
constexpr uint64_t kLeadBit = FOKL_ROW_LEAD;                // fokl_tape_row.start: the row opens with the cached normal
constexpr uint64_t kCachedHalf = FOKL_SOURCE_X1_HALF;       // a normal's source: the x1 half of the attempt (else x2)
constexpr uint64_t kGivenGauss = FOKL_SOURCE_GIVEN;         // source position: the cached value of the state handed over
constexpr uint64_t kFinalValue = FOKL_GAMMA_FINAL_VALUE;    // fokl_tape_row.gamma[j]: the walker stored the variate itself

static_assert(kSegSlots % 64 == 0, "mask words must not straddle segments");

#define FOKL_WIDE_TARGET __attribute__((target("avx512f,avx512dq,avx512vl,avx512bw,bmi,bmi2,popcnt,lzcnt")))

inline bool cpu_is_wide()
{
    static const bool wide = [] {
        const char *isa = std::getenv("FOKL_SAMPLER_ISA");            // "base" forces the portable loops (tests)
        if (isa && std::strcmp(isa, "base") == 0) return false;
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
               __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") &&
               __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
    }();
    return wide;
}

// One segment: [previous block, raw | the segment's words | the first words of the next segment], contiguous because the
// recurrence reads up to 624 words back; the words are raw while the token holder generates them and tempered from
// `ready` on.  mask[a] bit q: the polar attempt on doubles (2 q + a, 2 q + a + 1) of this segment is accepted.
struct alignas(64) Segment {
    uint32_t buf[MT_N + kSegWords + kSegTail];
    alignas(64) uint64_t mask[2][kSegMaskWords];
    // Round 6, the walk in RANK space (walk_tape_ranked below).  Accepted attempts of alignment a are numbered 0, 1, .. in
    // stream order ("rank"); cum[a][w] = how many lie in mask words < w.  Indexed by rank i (bit i of a plane):
    //   g0[a]      the slot right behind attempt i (same alignment) is accepted;
    //   g1[.][a]   three bit planes of G1(i) = how far the rank moves over the two gamma draws that follow attempt i when its
    //              x1 half is cached: the accepted a-slots behind it up to the one under the second draw's uniform
    //              (0 .. 6; 7 = more than four rejected attempts in a row there: not tabulated, the position walk decides).
    // Entries of attempts closer than 24 slots to the segment's end look into the next segment and are not valid:
    // safe[a] = the first such rank.  Made by the AVX-512 / BMI2 build only (finish_segment_wide).
    alignas(64) uint64_t g0[2][kSegMaskWords + 1];
    alignas(64) uint64_t g1[3][2][kSegMaskWords + 1];
    uint16_t cum[2][kSegMaskWords + 2];
    int32_t safe[2];
    int64_t index;
    std::atomic<int> ready;

    uint32_t *words() { return buf + MT_N; }
    const uint32_t *words() const { return buf + MT_N; }
};

// ---------------------------------------------------------------------------------------------------------------
// the bulk phase
// ---------------------------------------------------------------------------------------------------------------

inline uint32_t twist(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

// buf[g] for g in [from, to): the MT19937 recurrence over the flat array (block b + 1 follows block b in memory)
void recurrence_portable(uint32_t *buf, int from, int to)
{
    for (int g = from; g < to; ++g) buf[g] = buf[g - MT_SHIFT] ^ twist(buf[g - MT_N], buf[g - MT_N + 1]);
}

// `from` must be a multiple of 16 words from a 64-byte aligned buf.  Every load is ALIGNED to the grid of the stores:
// the words 227 back were stored fourteen steps ago and may still sit in the store queue, where a load that spans two
// stores cannot be forwarded (Zen 5: the unaligned form of this loop ran at 2 cycles per word, four times this one).
FOKL_WIDE_TARGET void recurrence_wide(uint32_t *buf, int from, int to)
{
    const __m512i upper = _mm512_set1_epi32((int)0x80000000u), mag = _mm512_set1_epi32((int)0x9908b0dfu);
    const __m512i one = _mm512_set1_epi32(1);
    int g = from;
    if ((from & 15) == 0 && (reinterpret_cast<uintptr_t>(buf) & 63) == 0) {
        __m512i a = _mm512_load_si512(buf + g - MT_N);                          // 624 = 39 * 16
        __m512i c_lo = _mm512_load_si512(buf + g - 240);                        // 240 = 15 * 16 >= 227
        for (; g + 16 <= to; g += 16) {                                         // 16 <= 227: no lane reads what this step writes
            const __m512i a_next = _mm512_load_si512(buf + g - MT_N + 16);
            const __m512i c_hi = _mm512_load_si512(buf + g - 224);
            const __m512i b = _mm512_alignr_epi32(a_next, a, 1);                // words g - 623 ..
            const __m512i c = _mm512_alignr_epi32(c_hi, c_lo, 13);              // words g - 227 ..
            const __m512i y = _mm512_ternarylogic_epi32(upper, a, b, 0xca);     // upper ? a : b
            const __mmask16 odd = _mm512_test_epi32_mask(y, one);
            __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
            r = _mm512_mask_xor_epi32(r, odd, r, mag);
            _mm512_store_si512(buf + g, r);
            a = a_next;
            c_lo = c_hi;
        }
    }
    for (; g < to; ++g) buf[g] = buf[g - MT_SHIFT] ^ twist(buf[g - MT_N], buf[g - MT_N + 1]);
}

inline uint32_t temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// numpy's 53-bit double from two tempered words, and the polar coordinate 2 d - 1 (both exact)
inline double to_double(uint32_t wa, uint32_t wb)
{
    const int32_t a = (int32_t)(wa >> 5), b = (int32_t)(wb >> 6);
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);   // a power of two: the product is numpy's quotient
}

// Tempering in place and the accept flags of a segment; `o` = parity of the word position the doubles pair up from
// (double l of the segment <-> words o + 2 l, o + 2 l + 1).
void finish_segment_portable(Segment *seg, int o, const uint32_t *raw)
{
    uint32_t *w = seg->words();
    for (int i = 0; i < kSegWords + kSegTail; ++i) w[i] = temper(raw[i]);
    std::vector<double> sq((size_t)kSegDoubles + 1);
    for (int l = 0; l <= kSegDoubles; ++l) {
        const double x = 2.0 * to_double(w[o + 2 * l], w[o + 2 * l + 1]) - 1.0;
        sq[(size_t)l] = x * x;
    }
    std::memset(seg->mask, 0, sizeof(seg->mask));
    for (int l = 0; l < kSegDoubles; ++l) {
        const double r2 = sq[(size_t)l] + sq[(size_t)l + 1];
        const uint64_t ok = (uint64_t)((r2 < 1.0) & (r2 != 0.0));
        seg->mask[l & 1][(l >> 1) >> 6] |= ok << ((l >> 1) & 63);
    }
}

void build_rank_tables(Segment *seg);

// x^2 of sixteen doubles in single precision, from the doubles' upper words alone (q: the first of their 32 tempered words):
// x = (wa >> 5) / 2^26 - 1 to 2e-7
FOKL_WIDE_TARGET inline __m512 polar_squares16(const uint32_t *q)
{
    const __m512i even_words = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    const __m512i hi = _mm512_permutex2var_epi32(_mm512_loadu_si512(q), even_words, _mm512_loadu_si512(q + 16));
    const __m512 x = _mm512_fmadd_ps(_mm512_cvtepi32_ps(_mm512_srli_epi32(hi, 5)), _mm512_set1_ps(1.0f / 67108864.0f),
                                     _mm512_set1_ps(-1.0f));
    return _mm512_mul_ps(x, x);
}

FOKL_WIDE_TARGET inline __m512d polar_squares(const uint32_t *p)
{
    // eight doubles from sixteen tempered words: lane = wa | wb << 32; v = (wa >> 5) 2^26 + (wb >> 6) < 2^53;
    // x = 2 v / 2^53 - 1 = (v - 2^52) / 2^52 (exact either way); x^2 is the one rounding numpy's x * x makes
    const __m512i lanes = _mm512_loadu_si512(p);
    const __m512i a = _mm512_srli_epi64(_mm512_and_si512(lanes, _mm512_set1_epi64(0xffffffffll)), 5);
    const __m512i b = _mm512_srli_epi64(lanes, 38);
    const __m512i v = _mm512_add_epi64(_mm512_slli_epi64(a, 26), b);
    const __m512d x = _mm512_mul_pd(_mm512_cvtepi64_pd(_mm512_sub_epi64(v, _mm512_set1_epi64(1ll << 52))),
                                    _mm512_set1_pd(1.0 / 4503599627370496.0));
    return _mm512_mul_pd(x, x);
}

// FOKL_SEGMENT_STORES=cached: the tempered words are written to the segment with ordinary stores (round 5)
static const bool g_segment_stream = !(std::getenv("FOKL_SEGMENT_STORES") && std::strcmp(std::getenv("FOKL_SEGMENT_STORES"), "cached") == 0);

// `raw` (the calling thread's scratch) is tempered IN PLACE: the flags read it there -- it stays in that core's L2 from
// segment to segment -- and the segment, which last saw use hundreds of segments ago, receives the words through
// non-temporal stores: no line of it is fetched to be overwritten, none displaces the scratch (round 6).
// ... in pieces of FOKL_TEMPER_CHUNK words (a power of two): a non-temporal store per flag group -- two lines at a time between
// the flags' own loads and stores -- leaves the write-combining buffers half filled
static const int g_temper_chunk = [] {
    const char *v = std::getenv("FOKL_TEMPER_CHUNK");
    int n = v ? std::atoi(v) : 256;
    if (n < 16 || (n & (n - 1))) n = 256;
    return n;
}();
FOKL_WIDE_TARGET void finish_segment_wide(Segment *seg, int o, uint32_t *raw)
{
    uint32_t *dst = seg->words();
    const bool stream = g_segment_stream;
    const uint32_t *w = stream ? raw : dst;                 // what the flags below read
    const int chunk = g_temper_chunk - 1;
    const __m512i m7 = _mm512_set1_epi32((int)0x9d2c5680u), m15 = _mm512_set1_epi32((int)0xefc60000u);
    // tempering runs a few groups ahead of the flags that read its output (one pass over the segment: the flags find the
    // words in the first-level cache instead of fetching 640 KB a second time)
    int tempered = 0;
    auto temper_to = [&](int upto) FOKL_WIDE_TARGET {
        upto = std::min(upto, kSegWords + kSegTail);
        for (; tempered < upto; tempered += 16) {
            __m512i y = _mm512_loadu_si512(raw + tempered);
            y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 11));
            y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 7), m7, 0x78);      // y ^ (shifted & mask)
            y = _mm512_ternarylogic_epi32(y, _mm512_slli_epi32(y, 15), m15, 0x78);
            y = _mm512_xor_si512(y, _mm512_srli_epi32(y, 18));
            if (stream) {
                _mm512_storeu_si512(raw + tempered, y);
                _mm512_stream_si512(reinterpret_cast<__m512i *>(dst + tempered), y);
            } else {
                _mm512_storeu_si512(dst + tempered, y);
            }
        }
    };
    temper_to(128);
    // The accept flag of the attempt that starts at each double, 0 < x_l^2 + x_{l+1}^2 < 1, sixteen at a time in SINGLE precision
    // from the doubles' upper words (x to 2e-7, the sum to 1.2e-6): whatever lies further than 8e-6 from both bounds is decided
    // (round 6: a third of the double-precision form's instructions); the one attempt per segment that does not is decided
    // by the exact expressions -- conversion and 2 d - 1 exact, x * x and the sum one rounding each, as numpy forms them.
    const uint32_t *p = w + o;
    const __m512 below = _mm512_set1_ps(1.0f - 8e-6f), above = _mm512_set1_ps(1.0f + 8e-6f), tiny = _mm512_set1_ps(8e-6f);
    auto exact_flag = [&](int l) -> uint64_t {
        const double x1 = 2.0 * to_double(p[2 * l], p[2 * l + 1]) - 1.0, x2 = 2.0 * to_double(p[2 * l + 2], p[2 * l + 3]) - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        return (uint64_t)((r2 < 1.0) & (r2 != 0.0));
    };
    __m512 cur = polar_squares16(p);
    uint64_t *m0 = seg->mask[0], *m1 = seg->mask[1];
    for (int word = 0; word < kSegMaskWords; ++word) {      // 128 doubles -> one mask word per alignment
        uint64_t even = 0, odd = 0;
        for (int half = 0; half < 2; ++half) {
            uint64_t bits = 0;
            for (int j = 0; j < 4; ++j) {
                const int l0 = 128 * word + 64 * half + 16 * j;                  // first double of the group
                // (the last group's successor lies in the tail words behind the segment: 32 of them, 16 doubles)
                temper_to(stream ? (2 * (l0 + 32) + 16 + chunk) & ~chunk : 2 * (l0 + 32) + 16);                                   // (nxt reads words up to o + 2 l0 + 63)
                const __m512 nxt = polar_squares16(p + 2 * (l0 + 16));
                const __m512 r2 = _mm512_add_ps(cur, _mm512_castsi512_ps(_mm512_alignr_epi32(_mm512_castps_si512(nxt),
                                                                                              _mm512_castps_si512(cur), 1)));
                const unsigned yes = _mm512_cmp_ps_mask(r2, below, _CMP_LT_OQ) & _mm512_cmp_ps_mask(r2, tiny, _CMP_GT_OQ);
                const unsigned no = _mm512_cmp_ps_mask(r2, above, _CMP_GT_OQ);
                uint64_t group = yes;
                unsigned open = ~(yes | no) & 0xffffu;
                while (open) {
                    const int i = __builtin_ctz(open);
                    open &= open - 1;
                    group |= exact_flag(l0 + i) << i;
                }
                bits |= group << (16 * j);
                cur = nxt;
            }
            even |= _pext_u64(bits, 0x5555555555555555ull) << (32 * half);
            odd |= _pext_u64(bits, 0xaaaaaaaaaaaaaaaaull) << (32 * half);
        }
        m0[word] = even;
        m1[word] = odd;
    }
    temper_to(kSegWords + kSegTail);
    if (stream) _mm_sfence();                               // the words are in place before the segment is published
    build_rank_tables(seg);
}

// The rank tables of a finished segment (see Segment): word by word, bit-sliced.  For the attempt in a-slot j the second
// gamma draw of an iteration that ends there looks for the first accepted attempt of the OTHER alignment from slot
// j + 1 + a on; with m rejected ones in front of it the walk comes back to alignment a behind a-slot j + 2 + m, so
// G1 = acc(j+1) + acc(j+2) + sum_{m >= 1} [the first m rejected] acc(j+2+m): six one-bit planes added by full adders, then
// compressed from slot order to rank order (pext by the word's own mask) and appended at bit cum[w].
FOKL_WIDE_TARGET void build_rank_tables(Segment *seg)
{
    for (int a = 0; a < 2; ++a) {
        // mask words with one zero word behind them (windows of the last word look into it: those entries are not valid
        // anyway, see Segment::safe)
        alignas(64) uint64_t M[kSegMaskWords + 8], X[kSegMaskWords + 8];
        std::memcpy(M, seg->mask[a], sizeof(uint64_t) * kSegMaskWords);
        std::memcpy(X, seg->mask[a ^ 1], sizeof(uint64_t) * kSegMaskWords);
        for (int i = kSegMaskWords; i < kSegMaskWords + 8; ++i) M[i] = X[i] = 0;
        uint64_t *planes[4] = {seg->g0[a], seg->g1[0][a], seg->g1[1][a], seg->g1[2][a]};
        for (auto *pl : planes) std::memset(pl, 0, sizeof(uint64_t) * (kSegMaskWords + 1));
        uint32_t c = 0;
        for (int w0 = 0; w0 < kSegMaskWords; w0 += 8) {     // eight mask words at a time (624 = 78 * 8)
            const __m512i m = _mm512_load_si512(M + w0), mn = _mm512_loadu_si512(M + w0 + 1);
            const __m512i x = _mm512_load_si512(X + w0), xn = _mm512_loadu_si512(X + w0 + 1);
#define FOKL_WIN(lo, hi, sh) _mm512_or_si512(_mm512_srli_epi64(lo, sh), _mm512_slli_epi64(hi, 64 - (sh)))
            const __m512i A1 = FOKL_WIN(m, mn, 1), A2 = FOKL_WIN(m, mn, 2), A3 = FOKL_WIN(m, mn, 3), A4 = FOKL_WIN(m, mn, 4),
                          A5 = FOKL_WIN(m, mn, 5), A6 = FOKL_WIN(m, mn, 6);
            // (ternary logic 0x10: E & ~window -- "still rejected")
            const __m512i ones = _mm512_set1_epi64(-1);
            const __m512i E1 = _mm512_xor_si512(FOKL_WIN(x, xn, 1 + a), ones);
            const __m512i E2 = _mm512_andnot_si512(FOKL_WIN(x, xn, 2 + a), E1), E3 = _mm512_andnot_si512(FOKL_WIN(x, xn, 3 + a), E2),
                          E4 = _mm512_andnot_si512(FOKL_WIN(x, xn, 4 + a), E3), E5 = _mm512_andnot_si512(FOKL_WIN(x, xn, 5 + a), E4);
#undef FOKL_WIN
            const __m512i t3 = _mm512_and_si512(E1, A3), t4 = _mm512_and_si512(E2, A4), t5 = _mm512_and_si512(E3, A5),
                          t6 = _mm512_and_si512(E4, A6);
            // full adders: 0x96 = a ^ b ^ c, 0xe8 = majority
            const __m512i s1 = _mm512_ternarylogic_epi64(A1, A2, t3, 0x96), c1 = _mm512_ternarylogic_epi64(A1, A2, t3, 0xe8);
            const __m512i s2 = _mm512_ternarylogic_epi64(t4, t5, t6, 0x96), c2 = _mm512_ternarylogic_epi64(t4, t5, t6, 0xe8);
            const __m512i c3 = _mm512_and_si512(s1, s2);
            alignas(64) uint64_t value[4][8], own[8];
            _mm512_store_si512(own, m);
            _mm512_store_si512(value[0], A1);
            _mm512_store_si512(value[1], _mm512_or_si512(_mm512_xor_si512(s1, s2), E5));
            _mm512_store_si512(value[2], _mm512_or_si512(_mm512_ternarylogic_epi64(c1, c2, c3, 0x96), E5));
            _mm512_store_si512(value[3], _mm512_or_si512(_mm512_ternarylogic_epi64(c1, c2, c3, 0xe8), E5));
            for (int k = 0; k < 8; ++k) {
                seg->cum[a][w0 + k] = (uint16_t)c;
                const int at = (int)(c >> 6), sh = (int)(c & 63);
                for (int pl = 0; pl < 4; ++pl) {
                    const uint64_t bits = _pext_u64(value[pl][k], own[k]);
                    planes[pl][at] |= bits << sh;
                    if (sh) planes[pl][at + 1] |= bits >> (64 - sh);
                }
                c += (uint32_t)__builtin_popcountll(own[k]);
            }
        }
        seg->cum[a][kSegMaskWords] = seg->cum[a][kSegMaskWords + 1] = (uint16_t)c;
        seg->safe[a] = (int32_t)seg->cum[a][kSegMaskWords - 1] +
                       __builtin_popcountll(M[kSegMaskWords - 1] & ((1ull << 40) - 1));   // attempts in slots < 39936 - 24
    }
}

// process-wide spare segments: a fit uses a few dozen and the next fit wants them mapped and warm
std::mutex g_spare_m;
std::vector<Segment *> g_spares;
// (a configs[2] fit ends with ~330 segments alive -- tapes on the device hold theirs until their chains are confirmed; with
// 192 spares the other 140 were unmapped at the end of every fit, 2-4 ms, and page-faulted in again by the next)
constexpr size_t kSpareMax = 1024;                          // 720 MB (configs[3]: 450 segments held for the device + 192 ahead)

Segment *take_segment()
{
    {
        std::lock_guard<std::mutex> lock(g_spare_m);
        if (!g_spares.empty()) {
            Segment *s = g_spares.back();
            g_spares.pop_back();
            return s;
        }
    }
    void *mem = nullptr;
    if (posix_memalign(&mem, 64, sizeof(Segment)) != 0) return nullptr;
    return new (mem) Segment;                               // (not value-initialised: 700 KB of zeroes nobody reads)
}

void give_segment(Segment *s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lock(g_spare_m);
        if (g_spares.size() < kSpareMax) {
            g_spares.push_back(s);
            return;
        }
    }
    s->~Segment();
    std::free(s);
}

}  // namespace

namespace {
struct WalkCrew;
}

struct fokl_stream {
    WalkCrew *crew = nullptr;                               // helper threads of the walk in rank space (walk_tape_crew)
    // the stream as it was handed over (numpy's get_state()): block 0 is mt_key, the first unread word is pos0
    uint32_t key0[MT_N];
    int pos0 = 0;
    int o = 0;                                              // pos0 & 1: double D <-> words o + 2 D, o + 2 D + 1
    // producer side
    std::mutex token_m;                                     // holder runs the recurrence of segment next_raw
    // producers with nothing to do sleep here; the walker (raising `limit`) and whoever moves `low_water` wake them.  Not
    // token_m: the walker must never wait for a recurrence to finish
    std::mutex room_m;
    std::condition_variable room_cv;
    std::atomic<int> sleepers{0};                           // producers inside (or about to enter) room_cv.wait
    uint32_t carry[MT_N];                                   // last raw block generated so far
    int64_t next_raw = 0;                                   // next segment to be generated
    int64_t oldest_alive = 0;                               // segments below were given back
    std::atomic<int64_t> limit{kAhead};                     // segments < limit may be produced
    std::atomic<int> ahead{kAhead};                         // how far in front of the walker (kAhead .. kAheadMax)
    std::atomic<int64_t> low_water{0};                      // segments < low_water will not be read again
    // raw pre-states for a second consumer of the stream (the device regenerates segments from them): entry index %
    // pre_entries = [624 raw words of the block in front of segment `index` (segment 0: block 0 itself) | index as two
    // words | 1 if the words ARE block 0 | parity o | padding to FOKL_PRESTATE_WORDS]; written under the token, in order
    uint32_t *pre_ring = nullptr;
    int pre_entries = 0;
    std::atomic<int64_t> pre_published{0};
    bool stop = false;
    std::atomic<bool> stop_flag{false};
    std::vector<std::thread> threads;
    std::atomic<Segment *> table[kTable];
    std::atomic<int64_t> bulk_busy_ns{0}, segments_made{0};
    std::atomic<int64_t> token_ns{0}, token_recurrence_ns{0}, token_wait_ns{0};     // FOKL_WALK_PROFILE: the serial section
    // holds: positions (doubles) somebody may still read from -- open tapes, expansions in flight
    std::mutex hold_m;
    std::multiset<uint64_t> holds;
    uint64_t walker_floor = 0;
    // the walker (one thread at a time)
    uint64_t D = 0;
    int has_gauss = 0;
    uint64_t gauss_src = 0;                                 // the attempt whose x1 half is the cached normal, or kGivenGauss
    double gauss0 = 0.0;                                    // the cached normal the state handed over held (kGivenGauss)
    std::atomic<int64_t> walker_wait_ns{0};
    std::atomic<int64_t> exact_draws{0}, gamma_draws{0};    // gamma attempts that needed libm / all of them
    std::atomic<int64_t> rollbacks{0};                      // iterations the wide walk had to redo draw by draw
    std::atomic<int64_t> position_iterations{0};            // iterations the rank walk left to the position walk
    bool wide = false;
    std::string error;
};

namespace {

inline int64_t now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

void update_low_water(fokl_stream *e)                       // hold_m held
{
    uint64_t d = e->walker_floor;
    if (!e->holds.empty()) d = std::min(d, *e->holds.begin());
    e->low_water.store((int64_t)(d / kSegDoubles), std::memory_order_release);
}

// The token holder's part of segment `index`: its raw words (and the first words of the next segment behind them).
// The serial part of a segment, under the token: the recurrence from the block the previous segment ended with, into the
// calling thread's own scratch buffer (the layout of Segment::buf).  The scratch stays in that core's L2 from segment to
// segment: 6 us on Zen 5 -- run straight into a Segment that last saw use two dozen segments ago it cost 25 us of cache
// misses, and that, not the walk, was what a fit's random stream was limited by (walker waiting for bulk threads 9 ms per
// configs[2] fit).  Tempering reads the scratch and writes the Segment once, outside the token.
void generate_raw(fokl_stream *e, uint32_t *buf, int64_t index)
{
    int from = MT_N;
    if (index == 0) {
        std::memset(buf, 0, MT_N * sizeof(uint32_t));
        std::memcpy(buf + MT_N, e->key0, MT_N * sizeof(uint32_t));      // block 0 is given
        from = 2 * MT_N;
    } else {
        std::memcpy(buf, e->carry, MT_N * sizeof(uint32_t));
    }
    const int to = MT_N + kSegWords + kSegTail;
    if (e->wide)
        recurrence_wide(buf, from, to);
    else
        recurrence_portable(buf, from, to);
    std::memcpy(e->carry, buf + kSegWords, MT_N * sizeof(uint32_t));    // the segment's last block, raw
    if (e->pre_ring) {
        // the raw block in front of every run of FOKL_PRESTATE_BLOCKS blocks (the buffer is still raw here)
        constexpr int kRuns = kSegBlocks / FOKL_PRESTATE_BLOCKS;
        for (int run = 0; run < kRuns; ++run) {
            const int64_t entry_index = index * kRuns + run;
            uint32_t *entry = e->pre_ring + (size_t)(entry_index % e->pre_entries) * FOKL_PRESTATE_WORDS;
            const bool given = index == 0 && run == 0;
            std::memcpy(entry, given ? e->key0 : buf + (size_t)MT_N * FOKL_PRESTATE_BLOCKS * run, MT_N * sizeof(uint32_t));
            entry[MT_N] = (uint32_t)((uint64_t)entry_index & 0xffffffffu);
            entry[MT_N + 1] = (uint32_t)((uint64_t)entry_index >> 32);
            entry[MT_N + 2] = given ? 1u : 0u;
            entry[MT_N + 3] = (uint32_t)e->o;
        }
        e->pre_published.store(index + 1, std::memory_order_release);
    }
}

static const bool g_bulk_profile = std::getenv("FOKL_WALK_PROFILE") != nullptr;

void bulk_worker(fokl_stream *e)
{
    struct Note {
        ~Note() { fokl_note_thread_cpu(4); }
    } note;

    void *scratch_mem = nullptr;
    if (posix_memalign(&scratch_mem, 64, sizeof(uint32_t) * (size_t)(MT_N + kSegWords + kSegTail + 64)) != 0) {
        std::lock_guard<std::mutex> lock(e->token_m);
        e->error = "fokl_stream: out of memory";
        e->stop = true;
        e->stop_flag.store(true, std::memory_order_release);
        e->room_cv.notify_all();
        return;
    }
    struct Free {
        void *p;
        ~Free() { std::free(p); }
    } scratch_owner{scratch_mem};
    uint32_t *scratch = static_cast<uint32_t *>(scratch_mem);
    for (;;) {
        // the segment to fill is in hand BEFORE the token is taken: mapping a fresh one (a fit of wide models has hundreds alive)
        // took half of the serial section of configs[3]'s production
        Segment *seg = take_segment();
        if (!seg) {
            std::lock_guard<std::mutex> lock(e->token_m);
            e->error = "fokl_stream: out of memory";
            e->stop = true;
            e->stop_flag.store(true, std::memory_order_release);
            e->room_cv.notify_all();
            return;
        }
        int64_t index = 0;
        std::vector<Segment *> retired;
        {
            const int64_t t_want = g_bulk_profile ? now_ns() : 0;
            std::unique_lock<std::mutex> lock(e->token_m);
            int64_t t_got = g_bulk_profile ? now_ns() : 0;
            if (g_bulk_profile) e->token_wait_ns.fetch_add(t_got - t_want, std::memory_order_relaxed);
            for (;;) {
                if (e->stop) {
                    give_segment(seg);
                    return;
                }
                const int64_t low = e->low_water.load(std::memory_order_acquire);
                if (e->next_raw < e->limit.load(std::memory_order_acquire) && e->next_raw - low < kTable - 2) break;
                const int64_t next = e->next_raw;
                lock.unlock();
                auto room_now = [&] {
                    return e->stop_flag.load(std::memory_order_seq_cst) ||
                           (next < e->limit.load(std::memory_order_seq_cst) &&
                            next - e->low_water.load(std::memory_order_seq_cst) < kTable - 2);
                };
                // a walker in the middle of a tape comes for the next segments within microseconds: look a few times
                // before going to sleep (a sleep and a wake-up per segment cost more than the segment)
                bool ready = false;
                for (int spins = 0, lim = fokl_spin_budget(400); spins < lim && !ready; ++spins) {
                    ready = room_now();
                    if (!ready) _mm_pause();
                }
                if (!ready) {
                    std::unique_lock<std::mutex> room(e->room_m);
                    e->sleepers.fetch_add(1, std::memory_order_seq_cst);
                    e->room_cv.wait(room, room_now);
                    e->sleepers.fetch_sub(1, std::memory_order_seq_cst);
                }
                lock.lock();
                if (g_bulk_profile) t_got = now_ns();
            }
            // segments nobody will read again go back first (their table entries are about to be reused)
            const int64_t low = e->low_water.load(std::memory_order_acquire);
            while (e->oldest_alive < low && e->oldest_alive < e->next_raw) {
                retired.push_back(e->table[e->oldest_alive % kTable].exchange(nullptr, std::memory_order_acq_rel));
                ++e->oldest_alive;
            }
            index = e->next_raw;
            seg->index = index;
            seg->ready.store(0, std::memory_order_relaxed);
            const int64_t t0 = now_ns();
            generate_raw(e, scratch, index);
            e->bulk_busy_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
            ++e->next_raw;
            if (g_bulk_profile) {
                e->token_recurrence_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
                e->token_ns.fetch_add(now_ns() - t_got, std::memory_order_relaxed);
            }
        }
        for (Segment *s : retired) give_segment(s);
        const int64_t t0 = now_ns();
        std::memcpy(seg->buf, scratch, MT_N * sizeof(uint32_t));       // the raw block in front (fokl_stream_state)
        if (e->wide)
            finish_segment_wide(seg, e->o, scratch + MT_N);
        else
            finish_segment_portable(seg, e->o, scratch + MT_N);
        seg->ready.store(1, std::memory_order_release);
        e->table[index % kTable].store(seg, std::memory_order_release);
        e->segments_made.fetch_add(1, std::memory_order_relaxed);
        e->bulk_busy_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
    }
}

static const bool g_ahead_adapts = !(std::getenv("FOKL_STREAM_AHEAD") && std::strcmp(std::getenv("FOKL_STREAM_AHEAD"), "fixed") == 0);

// The segment that owns double D (blocks until a bulk thread has published it).  Callers hold a position <= D.
Segment *segment_of(fokl_stream *e, uint64_t D, bool walker)
{
    const int64_t index = (int64_t)(D / kSegDoubles);
    Segment *seg = e->table[index % kTable].load(std::memory_order_acquire);
    if (seg && seg->index == index) return seg;
    const int ahead = e->ahead.load(std::memory_order_relaxed);
    if (walker && index + ahead > e->limit.load(std::memory_order_relaxed)) {
        e->limit.store(index + ahead, std::memory_order_seq_cst);
        if (e->sleepers.load(std::memory_order_seq_cst) > 0) {
            { std::lock_guard<std::mutex> lock(e->room_m); }
            e->room_cv.notify_all();
        }
    }
    const int64_t t0 = now_ns();
    for (int spins = 0;; ++spins) {
        seg = e->table[index % kTable].load(std::memory_order_acquire);
        if (seg && seg->index == index) break;
        if (spins < fokl_spin_budget(4000))
            _mm_pause();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(5));
        if ((spins & 1023) == 1023 && e->stop_flag.load(std::memory_order_acquire)) return nullptr;
    }
    if (walker) {
        const int64_t waited = e->walker_wait_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed) + (now_ns() - t0);
        const int64_t want = kAhead + 8 * (waited / 500000);
        if (want > ahead && ahead < kAheadMax && g_ahead_adapts) e->ahead.store((int)std::min<int64_t>(want, kAheadMax), std::memory_order_relaxed);
    }
    return seg;
}

// A reader's view of the stream: the segment under the cursor is cached, crossing into the next one is the rare path.
struct Reader {
    fokl_stream *e;
    bool walker;
    int o;
    Segment *seg = nullptr;
    uint64_t lo = 1, hi = 0;                                // doubles [lo, hi) belong to seg
    bool failed = false;

    Reader(fokl_stream *engine, bool is_walker) : e(engine), walker(is_walker), o(engine->o) {}

    __attribute__((always_inline)) inline bool locate(uint64_t D)
    {
        if (D - lo < (uint64_t)kSegDoubles && seg) return true;
        return relocate(D);
    }

    __attribute__((noinline)) bool relocate(uint64_t D)
    {
        Segment *s = segment_of(e, D, walker);
        if (!s) {
            failed = true;
            return false;
        }
        seg = s;
        lo = (D / kSegDoubles) * kSegDoubles;
        hi = lo + kSegDoubles;
        if (walker) {
            // keep the producers `ahead` segments in front of this one
            const int64_t want = (int64_t)(D / kSegDoubles) + 1 + e->ahead.load(std::memory_order_relaxed);
            if (want > e->limit.load(std::memory_order_relaxed)) {
                e->limit.store(want, std::memory_order_seq_cst);
                if (e->sleepers.load(std::memory_order_seq_cst) > 0) {
                    { std::lock_guard<std::mutex> lock(e->room_m); }
                    e->room_cv.notify_all();
                }
            }
        }
        return true;
    }

    // the words of the doubles at D (a gamma draw's attempt and its uniform: 24 bytes), if they belong to the segment under
    // the cursor: they were written by another core a moment ago
    __attribute__((always_inline)) inline void prefetch_words(uint64_t D) const
    {
        if (D - lo >= (uint64_t)kSegDoubles) return;
        const char *p = reinterpret_cast<const char *>(seg->words() + o + 2 * (D - lo));
        __builtin_prefetch(p);
        __builtin_prefetch(p + 24);
    }

    // the flag words two cache lines ahead of double D (called when the scan enters a new line of flags)
    __attribute__((always_inline)) inline void prefetch_flags(uint64_t D) const
    {
        const uint64_t word = ((D - lo) >> 7) + 16;
        if (word < (uint64_t)kSegMaskWords) {
            __builtin_prefetch(&seg->mask[0][word]);
            __builtin_prefetch(&seg->mask[1][word]);
        }
    }

    // value of double D
    __attribute__((always_inline)) inline double dbl(uint64_t D)
    {
        if (!locate(D)) return 0.5;
        const uint32_t *w = seg->words() + o + 2 * (D - lo);
        return to_double(w[0], w[1]);
    }

    // polar coordinates of the attempt starting at double D (its second double may be the first of the next segment:
    // the tail words cover it)
    __attribute__((always_inline)) inline void pair(uint64_t D, double &x1, double &x2)
    {
        if (!locate(D)) {
            x1 = x2 = 0.5;
            return;
        }
        const uint32_t *w = seg->words() + o + 2 * (D - lo);
        x1 = 2.0 * to_double(w[0], w[1]) - 1.0;
        x2 = 2.0 * to_double(w[2], w[3]) - 1.0;
    }

    // start of the first accepted attempt at or after D, same alignment
    __attribute__((always_inline)) inline uint64_t next_accepted(uint64_t D)
    {
        const int a = (int)(D & 1);
        for (;;) {
            if (!locate(D)) return D;
            const uint64_t q = (D - lo) >> 1;
            int word = (int)(q >> 6);
            uint64_t m = seg->mask[a][word] & (~0ull << (q & 63));
            while (!m && ++word < kSegMaskWords) m = seg->mask[a][word];
            if (m) return lo + 2 * ((uint64_t)word * 64 + (uint64_t)__builtin_ctzll(m)) + (uint64_t)a;
            D = hi + (uint64_t)a;
        }
    }
};

inline uint64_t select_bit_portable(uint64_t m, int k)     // position of the k-th (0-based) set bit
{
    for (int i = 0; i < k; ++i) m &= m - 1;
    return (uint64_t)__builtin_ctzll(m);
}

FOKL_WIDE_TARGET inline uint64_t select_bit_wide(uint64_t m, int k)
{
    return (uint64_t)_tzcnt_u64(_pdep_u64(1ull << k, m));
}

// Position behind the k-th accepted attempt at or after D (same alignment), k >= 1.
template <bool WIDE>
static inline __attribute__((always_inline)) uint64_t skip_accepted(Reader &r, uint64_t D, int k)
{
    const int a = (int)(D & 1);
    for (;;) {
        if (!r.locate(D)) return D;
        const uint64_t q = (D - r.lo) >> 1;
        int word = (int)(q >> 6);
        const uint64_t *mask = r.seg->mask[a];
        uint64_t m = mask[word] & (~0ull << (q & 63));
        int c = __builtin_popcountll(m);
        while (c < k) {
            k -= c;
            if (++word == kSegMaskWords) break;
            m = mask[word];
            c = __builtin_popcountll(m);
        }
        if (word < kSegMaskWords) {
            const uint64_t bit = WIDE ? select_bit_wide(m, k - 1) : select_bit_portable(m, k - 1);
            return r.lo + 2 * ((uint64_t)word * 64 + bit + 1) + (uint64_t)a;
        }
        D = r.hi + (uint64_t)a;                             // the rest lies in the next segment
    }
}

// ln(y) for normal y > 0 to within 2e-6 (absolute): exponent + atanh series on [1/sqrt 2, sqrt 2).  The walker decides
// the squeeze test of a gamma draw from BOUNDS built on this (kLnSlack is far above its error) and evaluates the exact
// expressions -- libm's log, numpy's order of operations -- only where the bounds do not decide.
constexpr double kLnSlack = 1e-4;

inline double fast_ln(double y)
{
    uint64_t bits;
    std::memcpy(&bits, &y, sizeof(bits));
    int e = (int)(bits >> 52) - 1023;
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    std::memcpy(&m, &bits, sizeof(m));
    if (m > 1.4142135623730951) {
        m *= 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    return e * 0.6931471805599453 + 2.0 * s * (1.0 + s2 * (1.0 / 3.0 + s2 * (0.2 + s2 * (1.0 / 7.0))));
}

// The walk: numpy's legacy draws on top of the reader, positions only.  A normal is known by its SOURCE -- the accepted
// attempt it comes from and which half of it -- and never formed here unless a gamma draw's accept test needs its value.
struct Walk {
    Reader r;
    fokl_stream *e;
    uint64_t D;
    int has_gauss;
    uint64_t gauss_src;
    int64_t exact = 0, gammas = 0;

    Walk(fokl_stream *engine) : r(engine, true), e(engine), D(engine->D), has_gauss(engine->has_gauss),
                                gauss_src(engine->gauss_src) {}
    // for readers on other threads (fokl_stream_expand): the reader and value_of only -- the walker's cursor, which the
    // walking thread writes at the end of every tape, is not looked at
    struct ReaderOnly {};
    Walk(fokl_stream *engine, ReaderOnly) : r(engine, false), e(engine), D(0), has_gauss(0), gauss_src(0) {}

    __attribute__((always_inline)) inline double next_double()
    {
        const double d = r.dbl(D);
        D += 1;
        return d;
    }

    // source of the next gauss(): the cached half, or the x2 half of the next accepted attempt (whose x1 half is cached)
    __attribute__((always_inline)) inline uint64_t gauss_source()
    {
        if (has_gauss) {
            has_gauss = 0;
            return gauss_src | kCachedHalf;
        }
        const uint64_t at = r.next_accepted(D);
        gauss_src = at;
        has_gauss = 1;
        D = at + 2;
        return at;
    }

    // the normal itself: numpy's operations, libm's log
    inline double value_of(uint64_t source)
    {
        const uint64_t at = source & ~kCachedHalf;
        if (at == kGivenGauss) return e->gauss0;
        double x1, x2;
        r.pair(at, x1, x2);
        const double r2 = x1 * x1 + x2 * x2;
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        return (source & kCachedHalf) ? f * x1 : f * x2;
    }

    inline double std_exponential() { return -std::log(1.0 - next_double()); }

    // numpy's shape > 1 branch (b = shape - 1/3, c = 1 / sqrt(9 b)): -> the source of the X the accepted attempt used
    __attribute__((always_inline)) inline uint64_t marsaglia_tsang(const double b, const double c)
    {
        for (;;) {
            ++gammas;
            uint64_t source;
            double X = 0.0, V = 0.0, x2_hi = 0.0;
            bool formed = false, positive = true;
            for (;;) {                                      // do { X = gauss(); V = 1 + c X; } while (V <= 0)
                source = gauss_source();
                formed = false;
                const uint64_t at = source & ~kCachedHalf;
                if (at != kGivenGauss) {
                    double x1, x2;
                    r.pair(at, x1, x2);
                    const double xh = (source & kCachedHalf) ? x1 : x2;
                    const double r2 = x1 * x1 + x2 * x2;
                    // X^2 = -2 ln(r2) xh^2 / r2, from above
                    x2_hi = 2.0 * (kLnSlack - fast_ln(r2)) * (xh * xh / r2) * (1.0 + 1e-12);
                    positive = xh >= 0.0;
                    if (positive || x2_hi * (c * c) < 0.81) break;              // V = 1 + c X > 0 for sure
                }
                X = value_of(source);
                V = 1.0 + c * X;
                formed = true;
                if (V > 0.0) break;
            }
            const double U = next_double();
            if (!formed) {
                // U < 1 - 0.0331 X^4 for sure?
                if (U < 1.0 - 0.0331 * (x2_hi * x2_hi) - 1e-13) return source;
                // ln U < X^2 / 2 + b (1 - V + ln V) for sure?  With t = c X and 9 b c^2 = 1 the right-hand side is
                // b h(t), h(t) = 3 (ln(1 + t) - t + t^2 / 2 - t^3 / 3) = 3 sum_{n >= 4} (-1)^(n+1) t^n / n:
                // h >= -0.75 t^4 for t >= 0 (alternating, decreasing terms), h >= -0.75 t^4 / (1 - |t|) for -1 < t < 0
                // (all terms negative, geometric bound); ln U <= U - 1.  The slack covers what numpy's own evaluation
                // of the right-hand side loses to cancellation (b times a few ulp of 1 - V + ln V).
                const double t2 = x2_hi * (c * c);
                if (t2 < 0.81) {
                    const double bound = 0.75 * b * (t2 * t2) / (positive ? 1.0 : 1.0 - std::sqrt(t2));
                    if (1.0 - U > 1.001 * bound + 1e-12 + 1e-14 * b * (1.0 + x2_hi)) return source;
                }
                X = value_of(source);
                V = 1.0 + c * X;
            }
            ++exact;
            V = V * V * V;
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return source;
            if (std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V))) return source;
        }
    }

    double std_gamma_small(double shape)                    // numpy's legacy_standard_gamma for shape <= 1
    {
        if (shape == 1.0) return std_exponential();
        if (shape == 0.0) return 0.0;
        for (;;) {
            const double U = next_double();
            const double V = std_exponential();
            if (U <= 1.0 - shape) {
                const double X = std::pow(U, 1.0 / shape);
                if (X <= V) return X;
            } else {
                const double Y = -std::log((1 - U) / shape);
                const double X = std::pow(1.0 - shape + shape * Y, 1.0 / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
};

// The body is inlined into two entry points: one compiled for AVX-512 / BMI2 / POPCNT machines (the counting and
// selecting of flags become popcnt / pdep / tzcnt), one for any x86-64.
template <bool WIDE>
static inline __attribute__((always_inline)) int walk_body(fokl_stream *e, int p1, int draws, double astar,
                                                            double atau_star, fokl_tape_row *rows, double *gam_sig,
                                                            double *gam_tau, int32_t *progress)
{
    Walk w(e);
    const bool fast_sig = astar > 1.0, fast_tau = atau_star > 1.0;
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    for (int k0 = 0; k0 < draws; k0 += FOKL_TAPE_BLOCK) {
        const int k1 = std::min(draws, k0 + FOKL_TAPE_BLOCK);
        for (int k = k0; k < k1; ++k) {
            // np.random.normal(0, 1, (p1, 1)): p1 successive gauss draws -- the cached normal if there is one, then
            // accepted attempts two normals each; an odd one out leaves its other half in the cache
            fokl_tape_row row;
            const int lead = w.has_gauss ? 1 : 0;
            row.lead_source = lead ? w.gauss_src : 0;
            w.has_gauss = 0;
            const uint64_t begin = w.D;
            row.start = begin | (lead ? kLeadBit : 0);
            const int rest = p1 - lead, attempts = (rest + 1) >> 1;
            if (attempts > 0) w.D = skip_accepted<WIDE>(w.r, w.D, attempts);
            if (rest & 1) {
                w.gauss_src = w.D - 2;                      // the last of them: its x2 half closes the row
                w.has_gauss = 1;
            }
            const uint64_t site = w.D;                       // the gamma draws begin here
            if (fast_sig) {
                row.gamma[0] = w.marsaglia_tsang(b_sig, c_sig);
            } else {
                row.gamma[0] = kFinalValue;
                gam_sig[k] = w.std_gamma_small(astar);
            }
            if (fast_tau) {
                row.gamma[1] = w.marsaglia_tsang(b_tau, c_tau);
            } else {
                row.gamma[1] = kFinalValue;
                gam_tau[k] = w.std_gamma_small(atau_star);
            }
            rows[k] = row;
            if (((site - w.r.lo) >> 10) != ((begin - w.r.lo) >> 10)) w.r.prefetch_flags(site);
        }
        if (w.r.failed) break;
        if (progress) __atomic_store_n(progress, k1, __ATOMIC_RELEASE);
    }
    e->exact_draws.fetch_add(w.exact, std::memory_order_relaxed);
    e->gamma_draws.fetch_add(w.gammas, std::memory_order_relaxed);
    if (w.r.failed) {
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        fokl_set_global_error("fokl_stream_walk: the stream's producers stopped (" + e->error + ")");
        return FOKL_ERR_STATE;
    }
    e->D = w.D;
    e->has_gauss = w.has_gauss;
    e->gauss_src = w.gauss_src;
    return FOKL_OK;
}

// ---- the AVX-512 walk: positions first, accept tests eight at a time ----------------------------------------------
// A gamma draw nearly always accepts its first attempt, so where the walk goes hardly ever depends on the VALUES of the
// draws.  The wide walk therefore runs a block of FOKL_TAPE_BLOCK iterations on positions alone -- every gamma draw taken
// to be one normal + one uniform -- and then checks all of the block's 2 x 16 accept tests together: the bounds of
// Walk::marsaglia_tsang evaluated in vector registers.  A test the bounds do not decide is evaluated exactly (libm,
// numpy's order of operations); if it turns out REJECTED (or V <= 0), the block is rolled back to the start of that
// iteration, which is then walked by the scalar code above, and the positions-only pass resumes behind it.  Rows are
// those of the scalar walk bit for bit (tests/test_stream_engine.py compares the two and fokl_noise_tape).

struct IterationStart {
    uint64_t D, gauss_src;
    int has_gauss;
};

// x = 2 d - 1 of eight doubles given as lanes wa | wb << 32 of tempered words
FOKL_WIDE_TARGET inline __m512d lanes_to_x(__m512i lanes)
{
    const __m512i a = _mm512_srli_epi64(_mm512_and_si512(lanes, _mm512_set1_epi64(0xffffffffll)), 5);
    const __m512i b = _mm512_srli_epi64(lanes, 38);
    const __m512i v = _mm512_add_epi64(_mm512_slli_epi64(a, 26), b);
    return _mm512_mul_pd(_mm512_cvtepi64_pd(_mm512_sub_epi64(v, _mm512_set1_epi64(1ll << 52))),
                         _mm512_set1_pd(1.0 / 4503599627370496.0));
}

// ln(y), 0 < y < 1, to within 2e-5: exponent + atanh series of the mantissa in [1, 2)
FOKL_WIDE_TARGET inline __m512d fast_ln8(__m512d y)
{
    const __m512d m = _mm512_getmant_pd(y, _MM_MANT_NORM_1_2, _MM_MANT_SIGN_zero);
    const __m512d ex = _mm512_getexp_pd(y);
    const __m512d one = _mm512_set1_pd(1.0);
    const __m512d s = _mm512_div_pd(_mm512_sub_pd(m, one), _mm512_add_pd(m, one));
    const __m512d s2 = _mm512_mul_pd(s, s);
    __m512d poly = _mm512_set1_pd(1.0 / 9.0);
    poly = _mm512_add_pd(_mm512_mul_pd(poly, s2), _mm512_set1_pd(1.0 / 7.0));
    poly = _mm512_add_pd(_mm512_mul_pd(poly, s2), _mm512_set1_pd(0.2));
    poly = _mm512_add_pd(_mm512_mul_pd(poly, s2), _mm512_set1_pd(1.0 / 3.0));
    poly = _mm512_add_pd(_mm512_mul_pd(poly, s2), one);
    return _mm512_add_pd(_mm512_mul_pd(ex, _mm512_set1_pd(0.6931471805599453)),
                         _mm512_mul_pd(_mm512_add_pd(s, s), poly));
}

// bit i set: gamma draw i of the eight surely accepts its first attempt (Walk::marsaglia_tsang's two bounds)
FOKL_WIDE_TARGET inline unsigned sure_accepts8(__m512i first, __m512i second, __m512i uniform, __mmask8 cached,
                                               __m512d b, __m512d c)
{
    const __m512d x1 = lanes_to_x(first), x2 = lanes_to_x(second);
    const __m512d r2 = _mm512_add_pd(_mm512_mul_pd(x1, x1), _mm512_mul_pd(x2, x2));
    const __m512d xh = _mm512_mask_blend_pd(cached, x2, x1);
    const __m512d ln = fast_ln8(r2);
    const __m512d one = _mm512_set1_pd(1.0);
    __m512d x2_hi = _mm512_mul_pd(_mm512_sub_pd(_mm512_set1_pd(kLnSlack), ln), _mm512_div_pd(_mm512_mul_pd(xh, xh), r2));
    x2_hi = _mm512_mul_pd(x2_hi, _mm512_set1_pd(2.0 * (1.0 + 1e-12)));
    // U = v / 2^53
    const __m512i ua = _mm512_srli_epi64(_mm512_and_si512(uniform, _mm512_set1_epi64(0xffffffffll)), 5);
    const __m512i ub = _mm512_srli_epi64(uniform, 38);
    const __m512d U = _mm512_mul_pd(_mm512_cvtepi64_pd(_mm512_add_epi64(_mm512_slli_epi64(ua, 26), ub)),
                                    _mm512_set1_pd(1.0 / 9007199254740992.0));
    const __mmask8 positive = _mm512_cmp_pd_mask(xh, _mm512_setzero_pd(), _CMP_GE_OQ);
    const __m512d t2 = _mm512_mul_pd(x2_hi, _mm512_mul_pd(c, c));
    const __mmask8 small = _mm512_cmp_pd_mask(t2, _mm512_set1_pd(0.81), _CMP_LT_OQ);
    // U < 1 - 0.0331 X^4 for sure
    const __m512d thr = _mm512_sub_pd(_mm512_sub_pd(one, _mm512_mul_pd(_mm512_set1_pd(0.0331), _mm512_mul_pd(x2_hi, x2_hi))),
                                      _mm512_set1_pd(1e-13));
    const __mmask8 squeeze = _mm512_cmp_pd_mask(U, thr, _CMP_LT_OQ);
    // ln U < X^2 / 2 + b (1 - V + ln V) for sure (see Walk::marsaglia_tsang)
    const __m512d denom = _mm512_mask_blend_pd(positive, _mm512_sub_pd(one, _mm512_sqrt_pd(t2)), one);
    const __m512d bound = _mm512_div_pd(_mm512_mul_pd(_mm512_mul_pd(_mm512_set1_pd(0.75), b), _mm512_mul_pd(t2, t2)), denom);
    const __m512d need = _mm512_add_pd(_mm512_add_pd(_mm512_mul_pd(_mm512_set1_pd(1.001), bound), _mm512_set1_pd(1e-12)),
                                       _mm512_mul_pd(_mm512_mul_pd(_mm512_set1_pd(1e-14), b), _mm512_add_pd(one, x2_hi)));
    const __mmask8 second_test = small & _mm512_cmp_pd_mask(_mm512_sub_pd(one, U), need, _CMP_GT_OQ);
    return (unsigned)((positive | small) & (squeeze | second_test));
}

// one iteration of the scalar walk (the body of walk_body's loop for shapes > 1)
template <bool WIDE>
static inline __attribute__((always_inline)) void walk_iteration(Walk &w, int p1, double b_sig, double c_sig,
                                                                 double b_tau, double c_tau, fokl_tape_row &row)
{
    const int lead = w.has_gauss ? 1 : 0;
    row.lead_source = lead ? w.gauss_src : 0;
    w.has_gauss = 0;
    row.start = w.D | (lead ? kLeadBit : 0);
    const int rest = p1 - lead, attempts = (rest + 1) >> 1;
    if (attempts > 0) w.D = skip_accepted<WIDE>(w.r, w.D, attempts);
    if (rest & 1) {
        w.gauss_src = w.D - 2;
        w.has_gauss = 1;
    }
    row.gamma[0] = w.marsaglia_tsang(b_sig, c_sig);
    row.gamma[1] = w.marsaglia_tsang(b_tau, c_tau);
}

int walk_tape_ranked(fokl_stream *e, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                     int32_t *progress);

FOKL_WIDE_TARGET int walk_tape_wide(fokl_stream *e, int p1, int draws, double astar, double atau_star,
                                    fokl_tape_row *rows, double *gam_sig, double *gam_tau, int32_t *progress)
{
    static const bool scalar_walk = std::getenv("FOKL_STREAM_SCALAR_WALK") != nullptr;
    // FOKL_STREAM_WALK=positions: round 4/5's walk (positions first, tests afterwards) instead of the walk in rank space
    static const bool position_walk = [] {
        const char *v = std::getenv("FOKL_STREAM_WALK");
        return v && std::strcmp(v, "positions") == 0;
    }();
    if (!(astar > 1.0) || !(atau_star > 1.0) || scalar_walk)
        return walk_body<true>(e, p1, draws, astar, atau_star, rows, gam_sig, gam_tau, progress);
    if (p1 >= 2 && !position_walk) return walk_tape_ranked(e, p1, draws, astar, atau_star, rows, progress);
    constexpr int B = FOKL_TAPE_BLOCK;
    Walk w(e);
    Walk checker(e);                                        // its reader serves the values of the accept tests
    checker.r.walker = false;
    Reader &rv = checker.r;
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    const __m512d bvec = _mm512_setr_pd(b_sig, b_tau, b_sig, b_tau, b_sig, b_tau, b_sig, b_tau);
    const __m512d cvec = _mm512_setr_pd(c_sig, c_tau, c_sig, c_tau, c_sig, c_tau, c_sig, c_tau);
    const int o = e->o;
    int64_t exact = 0, rolled_back = 0;
    IterationStart at_start[B];
    uint64_t source[2 * B], upos[2 * B];
    alignas(64) uint64_t first[2 * B], second[2 * B], uniform[2 * B];
    for (int k0 = 0; k0 < draws; k0 += B) {
        const int k1 = std::min(draws, k0 + B);
        int k = k0;
        while (k < k1) {
            // positions of iterations k .. k1 - 1, every gamma draw taken to accept its first attempt
            const int count = k1 - k;
            for (int i = 0; i < count; ++i) {
                at_start[i] = {w.D, w.gauss_src, w.has_gauss};
                fokl_tape_row &row = rows[k + i];
                const int lead = w.has_gauss ? 1 : 0;
                row.lead_source = lead ? w.gauss_src : 0;
                w.has_gauss = 0;
                const uint64_t begin = w.D;
                row.start = begin | (lead ? kLeadBit : 0);
                const int rest = p1 - lead, attempts = (rest + 1) >> 1;
                if (attempts > 0) w.D = skip_accepted<true>(w.r, w.D, attempts);
                if (rest & 1) {
                    w.gauss_src = w.D - 2;
                    w.has_gauss = 1;
                }
                // the words the accept tests will read (after the block's positions are known) can be on their way
                const uint64_t site = w.D;
                for (int j = 0; j < 2; ++j) {
                    const uint64_t src = w.gauss_source();
                    source[2 * i + j] = row.gamma[j] = src;
                    w.r.prefetch_words(src & ~kCachedHalf);
                    upos[2 * i + j] = w.D;
                    w.D += 1;
                }
                if (((site - w.r.lo) >> 10) != ((begin - w.r.lo) >> 10)) w.r.prefetch_flags(site);
            }
            if (w.r.failed) break;
            // the words behind the 2 * count accept tests
            const int tests = 2 * count;
            unsigned given = 0;
            for (int n = 0; n < tests; ++n) {
                const uint64_t src = source[n] & ~kCachedHalf;
                if (src == kGivenGauss) {                   // the cached value handed over with the state: no words
                    given |= 1u << n;
                    first[n] = second[n] = uniform[n] = 0x0400000004000000ull;
                    continue;
                }
                if (!rv.locate(src)) break;
                const uint32_t *p = rv.seg->words() + o + 2 * (src - rv.lo);
                std::memcpy(&first[n], p, 8);
                std::memcpy(&second[n], p + 2, 8);
                if (!rv.locate(upos[n])) break;
                std::memcpy(&uniform[n], rv.seg->words() + o + 2 * (upos[n] - rv.lo), 8);
            }
            if (rv.failed) {
                w.r.failed = true;
                break;
            }
            for (int n = tests; n < ((tests + 7) & ~7); ++n) {
                first[n] = first[0];
                second[n] = second[0];
                uniform[n] = uniform[0];
                source[n] = source[0];
            }
            unsigned sure = 0;
            for (int v = 0; v < tests; v += 8) {
                unsigned cached = 0;
                for (int l = 0; l < 8; ++l) cached |= (unsigned)((source[v + l] >> 63) & 1) << l;
                sure |= sure_accepts8(_mm512_load_si512(first + v), _mm512_load_si512(second + v),
                                      _mm512_load_si512(uniform + v), (__mmask8)cached, bvec, cvec)
                        << v;
            }
            sure &= ~given;
            unsigned open = ~sure & (tests >= 32 ? 0xffffffffu : ((1u << tests) - 1u));
            int redo = -1;                                  // iteration (relative to k) whose draw did not accept
            while (open) {
                const int n = __builtin_ctz(open);
                open &= open - 1;
                ++exact;
                const double b = (n & 1) ? b_tau : b_sig, c = (n & 1) ? c_tau : c_sig;
                const double X = checker.value_of(source[n]);
                double V = 1.0 + c * X;
                bool accepted = false;
                if (V > 0.0) {
                    const double U = rv.dbl(upos[n]);
                    V = V * V * V;
                    accepted = U < 1.0 - 0.0331 * (X * X) * (X * X) ||
                               std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V));
                }
                if (!accepted) {
                    redo = n >> 1;
                    break;
                }
            }
            if (rv.failed) {
                w.r.failed = true;
                break;
            }
            if (redo < 0) {
                k = k1;
                continue;
            }
            // iterations k .. k + redo - 1 stand; iteration k + redo is walked draw by draw from where it began
            ++rolled_back;
            w.D = at_start[redo].D;
            w.gauss_src = at_start[redo].gauss_src;
            w.has_gauss = at_start[redo].has_gauss;
            walk_iteration<true>(w, p1, b_sig, c_sig, b_tau, c_tau, rows[k + redo]);
            k += redo + 1;
        }
        if (w.r.failed) break;
        if (progress) __atomic_store_n(progress, k1, __ATOMIC_RELEASE);
    }
    e->exact_draws.fetch_add(exact + w.exact, std::memory_order_relaxed);
    e->gamma_draws.fetch_add(2 * (int64_t)draws, std::memory_order_relaxed);
    e->rollbacks.fetch_add(rolled_back, std::memory_order_relaxed);
    if (w.r.failed) {
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        fokl_set_global_error("fokl_stream_walk: the stream's producers stopped (" + e->error + ")");
        return FOKL_ERR_STATE;
    }
    e->D = w.D;
    e->has_gauss = w.has_gauss;
    e->gauss_src = w.gauss_src;
    return FOKL_OK;
}

// ---- the walk in rank space (round 6) ---------------------------------------------------------------------------
// What made the walk serial was "advance over k accepted attempts": a popcount loop and a select per iteration, every one
// depending on the one before (21.7 ns per Gibbs iteration, 20 ms of a 29 ms configs[2] fit).  Counted in ACCEPTED ATTEMPTS
// of the current alignment instead of doubles, that step is `r += k`.  What is left of an iteration's dependence is how far
// the two gamma draws move the rank, and that is a function of the attempt the normals ended on alone -- tabulated per
// segment by the bulk threads (Segment::g0 / g1).  So a block of iterations is walked in three passes:
//   1. the CHASE: r -> r + k + table[r + k ..], one or three bit loads per iteration (~3 ns), nothing else;
//   2. POSITIONS: the ranks back to stream positions (a monotone cursor over cum[] + pdep / tzcnt), the tape rows and the
//      sources of the gamma draws -- iterations independent of each other: the core overlaps them;
//   3. the accept TESTS of the gamma draws, sixteen at a time in vector registers (check_tests, as in walk_tape_wide).
// Whatever the tables do not cover -- the last attempts of a segment, more than four rejected attempts in a row at a
// gamma site, a gamma draw that does not accept its first attempt -- is walked by the position walk (walk_iteration) from
// the exact state, which passes 2 and 3 always know.  Rows, positions and consumption are those of the scalar walk bit for
// bit (tests/test_stream_engine.py: the builds against each other, against the one-thread recorder and numpy).
constexpr int kChase = 128;                                 // iterations per block (a multiple of FOKL_TAPE_BLOCK)
// FOKL_WALK_PROFILE=1: nanoseconds per pass (chase, positions, tests, position walk), printed when a stream is destroyed
static const bool g_walk_profile = std::getenv("FOKL_WALK_PROFILE") != nullptr;
// FOKL_ROW_STORES=cached: ordinary stores for the tape rows (A/B; default: non-temporal)
static const bool g_row_stream = !(std::getenv("FOKL_ROW_STORES") && std::strcmp(std::getenv("FOKL_ROW_STORES"), "cached") == 0);
static std::atomic<int64_t> g_walk_ns[4];
static std::atomic<int64_t> g_crew_wall_ns{0}, g_crew_idle_ns{0}, g_crew_tapes{0};     // FOKL_WALK_PROFILE: the walking thread inside walk_tape_crew
static std::atomic<int64_t> g_walk_iterations{0}, g_walk_by_position{0};
#define FOKL_LAP(i)                                                          \
    do {                                                                     \
        if (g_walk_profile) {                                                \
            const int64_t t_ = now_ns();                                     \
            g_walk_ns[i].fetch_add(t_ - lap_, std::memory_order_relaxed);    \
            lap_ = t_;                                                       \
        }                                                                    \
    } while (0)

// What the chase leaves of a block of iterations, and what passes 2 and 3 make of it.
struct BlockIn {
    const Segment *seg;
    uint64_t lo;                                            // first double of seg
    int a;                                                  // alignment the block's normals are read in
    uint32_t wc0;                                           // a mask word at or before the first rank's
    uint64_t D0, gsrc0;                                     // the walker's exact state at the block's first iteration
    int count;
    double b_sig, c_sig, b_tau, c_tau;
    fokl_tape_row *rows;                                    // the block's rows
    int32_t e[kChase];                                      // rank of the last attempt of each iteration's normals
    uint8_t hs[kChase], h[kChase];                          // cached normal at the gamma site / at the iteration's start
};
struct BlockOut {
    uint64_t source[2 * kChase], upos[2 * kChase];
    IterationStart after[kChase];                           // the walker's state behind each iteration
};

// first iteration (0-based, within [0, count)) one of whose gamma draws does not accept its first attempt, or -1
FOKL_WIDE_TARGET static inline int check_tests(Walk &checker, int o, const uint64_t *source, const uint64_t *upos, int count,
                                               __m512d bvec, __m512d cvec, double b_sig, double c_sig, double b_tau,
                                               double c_tau, int64_t &exact)
{
    Reader &rv = checker.r;
    alignas(64) uint64_t first[2 * FOKL_TAPE_BLOCK], second[2 * FOKL_TAPE_BLOCK], uniform[2 * FOKL_TAPE_BLOCK];
    uint64_t src_pad[2 * FOKL_TAPE_BLOCK];
    for (int i0 = 0; i0 < count; i0 += FOKL_TAPE_BLOCK) {
        const int tests = 2 * std::min(FOKL_TAPE_BLOCK, count - i0);
        const uint64_t *src = source + 2 * i0, *up = upos + 2 * i0;
        for (int n = 0; n < tests; ++n) {
            const uint64_t at = src[n] & ~kCachedHalf;
            src_pad[n] = src[n];
            if (!rv.locate(at)) return -2;
            const uint32_t *p = rv.seg->words() + o + 2 * (at - rv.lo);
            std::memcpy(&first[n], p, 8);
            std::memcpy(&second[n], p + 2, 8);
            if (!rv.locate(up[n])) return -2;
            std::memcpy(&uniform[n], rv.seg->words() + o + 2 * (up[n] - rv.lo), 8);
        }
        for (int n = tests; n < ((tests + 7) & ~7); ++n) {
            first[n] = first[0];
            second[n] = second[0];
            uniform[n] = uniform[0];
            src_pad[n] = src_pad[0];
        }
        unsigned sure = 0;
        for (int v = 0; v < tests; v += 8) {
            unsigned cached = 0;
            for (int l = 0; l < 8; ++l) cached |= (unsigned)((src_pad[v + l] >> 63) & 1) << l;
            sure |= sure_accepts8(_mm512_load_si512(first + v), _mm512_load_si512(second + v), _mm512_load_si512(uniform + v),
                                  (__mmask8)cached, bvec, cvec)
                    << v;
        }
        unsigned open = ~sure & (tests >= 32 ? 0xffffffffu : ((1u << tests) - 1u));
        while (open) {
            const int n = __builtin_ctz(open);
            open &= open - 1;
            ++exact;
            const double b = (n & 1) ? b_tau : b_sig, c = (n & 1) ? c_tau : c_sig;
            const double X = checker.value_of(src[n]);
            double V = 1.0 + c * X;
            bool accepted = false;
            if (V > 0.0) {
                const double U = rv.dbl(up[n]);
                V = V * V * V;
                accepted = U < 1.0 - 0.0331 * (X * X) * (X * X) || std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V));
            }
            if (rv.failed) return -2;
            if (!accepted) return i0 + (n >> 1);
        }
    }
    return -1;
}

// pass 1.  From rank r / cached normal h: up to `want` iterations inside the segment.  -> count; r, h move on.
static inline int chase_block(const Segment *seg, int a, int p1, int want, int64_t &r, int &h, BlockIn &in)
{
    const int64_t limit = seg->safe[a];
    const uint64_t *G0 = seg->g0[a], *P0 = seg->g1[0][a], *P1 = seg->g1[1][a], *P2 = seg->g1[2][a];
    int count = 0;
    while (count < want) {
        const int rest = p1 - h, hs = rest & 1;
        const int64_t last = r + ((rest + 1) >> 1) - 1;
        if (last + 2 >= limit) break;
        int64_t next;
        if (hs) {
            const int sh = (int)(last & 63);
            const size_t at = (size_t)(last >> 6);
            const int g = (int)((P0[at] >> sh) & 1) | (int)(((P1[at] >> sh) & 1) << 1) | (int)(((P2[at] >> sh) & 1) << 2);
            if (g == 7) break;
            next = last + 1 + g;
        } else {
            next = last + 2 + (int64_t)((G0[(size_t)((last + 1) >> 6)] >> ((last + 1) & 63)) & 1);
        }
        in.e[count] = (int32_t)last;
        in.hs[count] = (uint8_t)hs;
        in.h[count] = (uint8_t)h;
        ++count;
        r = next;
        h = hs;
    }
    in.count = count;
    return count;
}

// Where iteration (e, hs) of alignment a ends: the two gamma draws' sources and uniforms, the state behind it.  wc: cursor
// over the mask words (ranks only grow within a block).
FOKL_WIDE_TARGET static inline __attribute__((always_inline)) void iteration_sites(const Segment *seg, uint64_t lo, int a,
                                                                                   uint32_t &wc, int32_t e, int hs,
                                                                                   uint64_t src[2], uint64_t up[2],
                                                                                   IterationStart &after)
{
    const uint16_t *cum = seg->cum[a];
    const uint32_t idx = (uint32_t)e + (hs ? 0u : 1u);
    while (cum[wc + 1] <= idx) ++wc;
    const uint64_t slot = 64ull * wc + (uint64_t)_tzcnt_u64(_pdep_u64(1ull << (idx - cum[wc]), seg->mask[a][wc]));
    const uint64_t A = lo + 2 * slot + (uint64_t)a;
    if (hs) {
        // the x1 half of the normals' last attempt is cached: the first draw uses it, the second takes the next accepted
        // attempt of the other alignment (at most four rejected ones in front of it: g < 7)
        const uint64_t *X = seg->mask[a ^ 1];
        uint64_t t = slot + 1 + (uint64_t)a;
        uint64_t mm = X[t >> 6] >> (t & 63);
        if (!mm) {
            t = (t | 63) + 1;
            mm = X[t >> 6];
        }
        t += (uint64_t)_tzcnt_u64(mm);
        const uint64_t Q = lo + 2 * t + (uint64_t)(a ^ 1);
        src[0] = A | kCachedHalf;
        up[0] = A + 2;
        src[1] = Q;
        up[1] = Q + 2;
        after = {Q + 3, Q, 1};
    } else {
        src[0] = A;
        up[0] = A + 2;
        src[1] = A | kCachedHalf;
        up[1] = A + 3;
        after = {A + 4, A, 0};
    }
}

static const bool g_prefetch_second = !(std::getenv("FOKL_WALK_PREFETCH") && std::strcmp(std::getenv("FOKL_WALK_PREFETCH"), "first") == 0);

// pass 2: the block's rows, the sources of its gamma draws, the state behind every iteration
FOKL_WIDE_TARGET static inline void block_positions(const BlockIn &in, BlockOut &out, int o)
{
    uint32_t wc = in.wc0;
    uint64_t D = in.D0, gsrc = in.gsrc0;
    const bool stream_rows = g_row_stream && (reinterpret_cast<uintptr_t>(in.rows) & 31) == 0;
    for (int i = 0; i < in.count; ++i) {
        const int lead = in.h[i];
        iteration_sites(in.seg, in.lo, in.a, wc, in.e[i], in.hs[i], out.source + 2 * i, out.upos + 2 * i, out.after[i]);
        // (a row is written once and read by another agent -- the device over the bus, a finish thread: streamed past the
        // cache, no line is fetched to be overwritten; callers fence before they publish progress)
        const __m256i rowv = _mm256_set_epi64x((long long)out.source[2 * i + 1], (long long)out.source[2 * i],
                                               (long long)(lead ? gsrc : 0), (long long)(D | (lead ? kLeadBit : 0)));
        if (stream_rows)
            _mm256_stream_si256(reinterpret_cast<__m256i *>(in.rows + i), rowv);
        else
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(in.rows + i), rowv);
        D = out.after[i].D;
        gsrc = out.after[i].gauss_src;
        // the words of this iteration's gamma site (both draws' attempts and uniforms lie within a few doubles of each other,
        // inside the block's own segment): written by another core a moment ago, read by the tests that follow
        const char *site = reinterpret_cast<const char *>(in.seg->words() + o + 2 * ((out.source[2 * i] & ~kCachedHalf) - in.lo));
        __builtin_prefetch(site);
        __builtin_prefetch(site + 24);
        if (g_prefetch_second) {
            // (the second draw's attempt may lie a few attempts further on: another line as often as not)
            const char *site2 = reinterpret_cast<const char *>(in.seg->words() + o + 2 * ((out.source[2 * i + 1] & ~kCachedHalf) - in.lo));
            __builtin_prefetch(site2 + 24);
        }
    }
}

// ---- the walk crew: passes 2 and 3 on other threads ------------------------------------------------------------------
// With the sub-stage loop native (csrc/fokl_run.cpp) a fit is bound by the walk: its thread is busy 18.6 of 23.3 ms.  The chase
// is 4 ns per iteration of its 19-20; positions and tests -- every gamma site a cache line another core wrote a moment ago --
// are the rest and do not depend on each other from block to block.  With helper threads (fokl_stream_set_helpers) the
// walking thread only chases: it publishes each block's ranks and exact first state (one select per block gives it the
// state behind the block) and goes on, taking the block to be good, which 899 in 900 iterations are; helpers form the rows,
// run the accept tests and report the first iteration to redo, if any.  Verdicts are read in order; a block that is not good
// takes everything issued behind it with it (the walker waits for those helpers, walks the one iteration by positions and
// chases on from there).  Rows, positions and consumption are the one-thread walk's bit for bit.
constexpr int kCrewRing = 16;                               // slots; blocks in flight at most g_crew_depth of them
static const int g_crew_depth = [] {
    const char *v = std::getenv("FOKL_CREW_DEPTH");
    const int n = v ? std::atoi(v) : 8;
    return n >= 2 && n <= kCrewRing ? n : 8;
}();
constexpr int kCrewMax = 4;

struct alignas(64) CrewSlot {
    BlockIn in;
    int redo = -1;                                          // the helper's verdict: -1 good, >= 0 first iteration to redo, -2 failure
    IterationStart redo_from{};                             // the exact state at that iteration's start
    int64_t exact = 0;
    alignas(64) std::atomic<uint64_t> ready{0};             // number of the block in `in` (published by the walker)
    alignas(64) std::atomic<uint64_t> done{0};              // number of the block whose verdict stands here
};

struct WalkCrew {
    fokl_stream *e = nullptr;
    int helpers = 0;
    CrewSlot slots[kCrewRing];
    uint64_t issued = 0;                                    // blocks are numbered 1, 2, ..: block n lives in slot n % kCrewRing
    alignas(64) std::atomic<uint64_t> claimed{0};           // blocks 1 .. claimed have a worker (helpers and, when it would
                                                            // only wait, the walking thread take the next published one)
    std::unique_ptr<BlockOut> walker_out;                   // the walking thread's own work area, when it lends a hand
    alignas(64) std::atomic<bool> stop{false};
    std::atomic<int> sleepers{0};
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::thread> threads;
    std::vector<int> cpus;                                  // helper i runs on cpus[i] (empty: wherever its creator may)
};

// passes 2 and 3 of block n (published, claimed by the caller)
FOKL_WIDE_TARGET static void crew_process(WalkCrew *crew, uint64_t n, BlockOut &out, Walk &checker)
{
    CrewSlot &slot = crew->slots[n % kCrewRing];
    const BlockIn &in = slot.in;
    const int o = crew->e->o;
    checker.r.failed = false;
    const int64_t t0 = g_walk_profile ? now_ns() : 0;
    block_positions(in, out, o);
    const int64_t t1 = g_walk_profile ? now_ns() : 0;
    const __m512d bvec = _mm512_setr_pd(in.b_sig, in.b_tau, in.b_sig, in.b_tau, in.b_sig, in.b_tau, in.b_sig, in.b_tau);
    const __m512d cvec = _mm512_setr_pd(in.c_sig, in.c_tau, in.c_sig, in.c_tau, in.c_sig, in.c_tau, in.c_sig, in.c_tau);
    int64_t exact = 0;
    const int redo = check_tests(checker, o, out.source, out.upos, in.count, bvec, cvec, in.b_sig, in.c_sig, in.b_tau, in.c_tau,
                                 exact);
    if (g_walk_profile) {
        g_walk_ns[1].fetch_add(t1 - t0, std::memory_order_relaxed);
        g_walk_ns[2].fetch_add(now_ns() - t1, std::memory_order_relaxed);
    }
    slot.redo = redo;
    slot.exact = exact;
    if (redo > 0)
        slot.redo_from = out.after[redo - 1];
    else if (redo == 0)
        slot.redo_from = {in.D0, in.gsrc0, (int)in.h[0]};
    _mm_sfence();                                           // the rows were streamed
    slot.done.store(n, std::memory_order_release);
}

// the next published block nobody has taken yet -> its number, or 0
static inline uint64_t crew_claim(WalkCrew *crew)
{
    uint64_t n = crew->claimed.load(std::memory_order_acquire);
    while (crew->slots[(n + 1) % kCrewRing].ready.load(std::memory_order_acquire) == n + 1)
        if (crew->claimed.compare_exchange_weak(n, n + 1, std::memory_order_acq_rel)) return n + 1;
    return 0;
}

FOKL_WIDE_TARGET void crew_helper(WalkCrew *crew, int which)
{
    struct Note {
        ~Note() { fokl_note_thread_cpu(0); }
    } note;
    if (which < (int)crew->cpus.size() && crew->cpus[(size_t)which] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(crew->cpus[(size_t)which], &set);
        (void)sched_setaffinity(0, sizeof(set), &set);
    }
    Walk checker(crew->e, Walk::ReaderOnly{});
    std::unique_ptr<BlockOut> out(new BlockOut());
    for (;;) {
        uint64_t n = 0;
        for (int spins = 0; (n = crew_claim(crew)) == 0; ++spins) {
            if (crew->stop.load(std::memory_order_acquire)) return;
            if (spins < fokl_spin_budget(20000)) {
                _mm_pause();
            } else {
                std::unique_lock<std::mutex> lock(crew->m);
                crew->sleepers.fetch_add(1, std::memory_order_seq_cst);
                crew->cv.wait(lock, [&] {
                    if (crew->stop.load(std::memory_order_seq_cst)) return true;
                    const uint64_t c = crew->claimed.load(std::memory_order_seq_cst);
                    return crew->slots[(c + 1) % kCrewRing].ready.load(std::memory_order_seq_cst) == c + 1;
                });
                crew->sleepers.fetch_sub(1, std::memory_order_seq_cst);
                spins = 0;
            }
        }
        crew_process(crew, n, *out, checker);
    }
}

void crew_stop(fokl_stream *e)
{
    WalkCrew *crew = e->crew;
    if (!crew) return;
    {
        std::lock_guard<std::mutex> lock(crew->m);
        crew->stop.store(true, std::memory_order_seq_cst);
    }
    crew->cv.notify_all();
    for (auto &t : crew->threads) t.join();
    delete crew;
    e->crew = nullptr;
}

// Before an iteration is walked by positions near a segment's end: the lines that walk is about to miss on one after the other
// -- the mask words under and behind w.D in both alignments, the next segment's first mask words, counts and table words (the
// chase resumes there), the words around the place the iteration will end on (its gamma site: attempts / 0.785 attempts on) --
// are requested together.  All of them were written by other cores; the walk's loads would fetch them in sequence, ~100 ns each.
FOKL_WIDE_TARGET static inline void prefetch_crossing(fokl_stream *e, Walk &w, int p1)
{
    if (!w.r.seg || w.D - w.r.lo >= (uint64_t)kSegDoubles) return;
    const Segment *seg = w.r.seg;
    const uint64_t q = (w.D - w.r.lo) >> 1;
    const size_t word = (size_t)(q >> 6);
    __builtin_prefetch(&seg->mask[0][word]);
    __builtin_prefetch(&seg->mask[1][word]);
    if (word + 8 < (size_t)kSegMaskWords) {
        __builtin_prefetch(&seg->mask[0][word + 8]);
        __builtin_prefetch(&seg->mask[1][word + 8]);
    }
    const uint64_t site = (w.D - w.r.lo) + (uint64_t)(1.27 * (p1 + 1));       // doubles from the segment's start
    const char *words = reinterpret_cast<const char *>(seg->words() + e->o);
    const Segment *next = e->table[(seg->index + 1) % kTable].load(std::memory_order_acquire);
    if (next && next->index != seg->index + 1) next = nullptr;
    for (int line = -2; line <= 3; ++line) {
        const int64_t d = (int64_t)site + 8 * line;
        if (d < 0) continue;
        if (d < kSegDoubles)
            __builtin_prefetch(words + 8 * d);
        else if (next)
            __builtin_prefetch(reinterpret_cast<const char *>(next->words() + e->o) + 8 * (d - kSegDoubles));
    }
    if (next && site + 512 >= (uint64_t)kSegDoubles) {
        for (int a = 0; a < 2; ++a) {
            __builtin_prefetch(&next->mask[a][0]);
            __builtin_prefetch(&next->cum[a][0]);
            __builtin_prefetch(&next->g0[a][0]);
            __builtin_prefetch(&next->g1[0][a][0]);
            __builtin_prefetch(&next->g1[1][a][0]);
            __builtin_prefetch(&next->g1[2][a][0]);
        }
        __builtin_prefetch(&next->safe[0]);
    }
}

static const bool g_crew_inline = !(std::getenv("FOKL_CREW_INLINE") && std::strcmp(std::getenv("FOKL_CREW_INLINE"), "0") == 0);

// the tape walked by the crew: the calling thread chases, the helpers do the rest
FOKL_WIDE_TARGET int walk_tape_crew(fokl_stream *e, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                                    int32_t *progress)
{
    WalkCrew *crew = e->crew;
    const int64_t t_enter = g_walk_profile ? now_ns() : 0;
    int64_t idle_ns = 0;
    Walk w(e);
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    int64_t exact = 0, rolled_back = 0, by_position = 0;
    int k = 0, k_issued = 0, published = 0;                 // iterations that stand / that were chased / that `progress` announced
    auto publish = [&](int upto) {
        const int whole = upto == draws ? draws : upto - upto % FOKL_TAPE_BLOCK;
        if (progress && whole > published) __atomic_store_n(progress, whole, __ATOMIC_RELEASE);    // (helpers fenced their rows)
        published = std::max(published, whole);
    };
    uint64_t first_open = crew->issued + 1;                 // the oldest block without a verdict
    // iterations this thread walked by positions from the PRESUMED state behind blocks still open (the chase stops at every
    // segment's end -- six times per tape of a hundred columns -- and waiting for every verdict there idled the whole crew):
    // entry = the number of the last block issued before the iteration; it stands once that block and all before it are good
    std::deque<uint64_t> inline_after;
    Walk lender(e, Walk::ReaderOnly{});                     // (reader of the blocks this thread processes itself)
    // a verdict this thread has to wait for: it takes published blocks nobody has claimed yet instead of only waiting
    auto wait_done = [&](uint64_t n) -> CrewSlot & {
        CrewSlot &slot = crew->slots[n % kCrewRing];
        int64_t idle_from = 0;
        for (int spins = 0; slot.done.load(std::memory_order_acquire) != n; ++spins) {
            if (const uint64_t mine = crew_claim(crew)) {
                if (idle_from) idle_ns += now_ns() - idle_from;
                idle_from = 0;
                crew_process(crew, mine, *crew->walker_out, lender);
                spins = 0;
            } else if (spins < 50000) {
                if (g_walk_profile && !idle_from) idle_from = now_ns();
                _mm_pause();
            } else {
                std::this_thread::yield();
            }
        }
        if (idle_from) idle_ns += now_ns() - idle_from;
        return slot;
    };
    // the oldest open block's verdict (waited for).  -> false: it was not good: the walker stands at the iteration to redo
    auto judge_oldest = [&]() -> bool {
        CrewSlot &slot = wait_done(first_open);
        exact += slot.exact;
        const int redo = slot.redo;
        if (redo == -1) {
            k += slot.in.count;
            ++first_open;
            while (!inline_after.empty() && inline_after.front() < first_open) {
                ++k;
                inline_after.pop_front();
            }
            publish(k);
            return true;
        }
        for (uint64_t n = first_open + 1; n <= crew->issued; ++n) (void)wait_done(n);      // what was issued behind it goes with it
        first_open = crew->issued + 1;
        by_position -= (int64_t)inline_after.size();
        inline_after.clear();
        if (redo == -2) {
            w.r.failed = true;
            return false;
        }
        k += redo;
        w.D = slot.redo_from.D;
        w.gauss_src = slot.redo_from.gauss_src;
        w.has_gauss = slot.redo_from.has_gauss;
        k_issued = k;
        ++rolled_back;
        publish(k);
        return false;
    };
    while (k < draws && !w.r.failed) {
        int64_t lap_ = g_walk_profile ? now_ns() : 0;
        bool exact_state = true;                            // w is the state behind the last iteration that stands
        if (k_issued < draws && w.r.locate(w.D) && w.D - w.r.lo < (uint64_t)kSegDoubles - 64) {
            if (crew->issued - first_open + 1 >= (uint64_t)g_crew_depth) {  // the ring is full: the oldest verdict first
                if (judge_oldest()) continue;
            } else {
                const Segment *seg = w.r.seg;
                const int a = (int)(w.D & 1);
                const uint64_t q = (w.D - w.r.lo) >> 1;
                int64_t r = (int64_t)seg->cum[a][q >> 6] + __builtin_popcountll(seg->mask[a][q >> 6] & ~(~0ull << (q & 63)));
                int h = w.has_gauss;
                const int want = std::min(kChase, draws - k_issued);
                BlockIn &in = crew->slots[(crew->issued + 1) % kCrewRing].in;
                const int count = chase_block(seg, a, p1, want, r, h, in);
                FOKL_LAP(0);
                if (count > 0) {
                    in.seg = seg;
                    in.lo = w.r.lo;
                    in.a = a;
                    in.wc0 = (uint32_t)(q >> 6);
                    in.D0 = w.D;
                    in.gsrc0 = w.gauss_src;
                    in.rows = rows + k_issued;
                    in.b_sig = b_sig;
                    in.c_sig = c_sig;
                    in.b_tau = b_tau;
                    in.c_tau = c_tau;
                    // the state behind the block (exact unless one of its draws turns out not to accept): one select
                    uint32_t wc = in.wc0;
                    uint64_t src[2], up[2];
                    IterationStart after;
                    iteration_sites(seg, in.lo, a, wc, in.e[count - 1], in.hs[count - 1], src, up, after);
                    const uint64_t n = ++crew->issued;
                    crew->slots[n % kCrewRing].ready.store(n, std::memory_order_seq_cst);
                    if (crew->sleepers.load(std::memory_order_seq_cst) > 0) {
                        { std::lock_guard<std::mutex> lock(crew->m); }
                        crew->cv.notify_all();
                    }
                    w.D = after.D;
                    w.gauss_src = after.gauss_src;
                    w.has_gauss = after.has_gauss;
                    k_issued += count;
                    // verdicts that have arrived meanwhile
                    bool good = true;
                    while (good && first_open <= crew->issued &&
                           crew->slots[first_open % kCrewRing].done.load(std::memory_order_acquire) == first_open)
                        good = judge_oldest();
                    if (good && count == want) continue;
                    exact_state = good;                     // (not good: w stands at the iteration to redo already)
                }
            }
        }
        // the chase stopped (the segment's end, a long run of rejected attempts, the tape's end) or a block was not good.
        // With blocks still open and iterations left, the one iteration the tables do not cover is walked by positions from the
        // state the open blocks will leave if they are good -- as the chase itself assumes; it stands when they have been judged
        if (exact_state && g_crew_inline && k_issued < draws && first_open <= crew->issued) {
            if (g_walk_profile) lap_ = now_ns();
            prefetch_crossing(e, w, p1);
            walk_iteration<true>(w, p1, b_sig, c_sig, b_tau, c_tau, rows[k_issued]);
            inline_after.push_back(crew->issued);
            ++k_issued;
            ++by_position;
            FOKL_LAP(3);
            continue;
        }
        // otherwise the position walk needs the exact state, i.e. every block judged
        while (exact_state && first_open <= crew->issued) exact_state = judge_oldest();
        if (w.r.failed) break;
        if (k < draws) {
            if (g_walk_profile) lap_ = now_ns();
            walk_iteration<true>(w, p1, b_sig, c_sig, b_tau, c_tau, rows[k]);
            ++k;
            k_issued = k;
            ++by_position;
            FOKL_LAP(3);
        }
        publish(k);
    }
    e->exact_draws.fetch_add(exact + w.exact, std::memory_order_relaxed);
    e->gamma_draws.fetch_add(2 * (int64_t)draws, std::memory_order_relaxed);
    e->rollbacks.fetch_add(rolled_back, std::memory_order_relaxed);
    e->position_iterations.fetch_add(by_position, std::memory_order_relaxed);
    if (g_walk_profile) {
        g_walk_iterations.fetch_add(k, std::memory_order_relaxed);
        g_walk_by_position.fetch_add(by_position, std::memory_order_relaxed);
        g_crew_wall_ns.fetch_add(now_ns() - t_enter, std::memory_order_relaxed);
        g_crew_idle_ns.fetch_add(idle_ns, std::memory_order_relaxed);
        g_crew_tapes.fetch_add(1, std::memory_order_relaxed);
    }
    if (w.r.failed) {
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        fokl_set_global_error("fokl_stream_walk: the stream's producers stopped (" + e->error + ")");
        return FOKL_ERR_STATE;
    }
    publish(draws);
    e->D = w.D;
    e->has_gauss = w.has_gauss;
    e->gauss_src = w.gauss_src;
    return FOKL_OK;
}

FOKL_WIDE_TARGET int walk_tape_ranked(fokl_stream *e, int p1, int draws, double astar, double atau_star,
                                      fokl_tape_row *rows, int32_t *progress)
{
    if (e->crew) return walk_tape_crew(e, p1, draws, astar, atau_star, rows, progress);
    Walk w(e);                                              // the exact state; its reader keeps the producers ahead
    Walk checker(e);                                        // its reader serves the values of the accept tests
    checker.r.walker = false;
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    const __m512d bvec = _mm512_setr_pd(b_sig, b_tau, b_sig, b_tau, b_sig, b_tau, b_sig, b_tau);
    const __m512d cvec = _mm512_setr_pd(c_sig, c_tau, c_sig, c_tau, c_sig, c_tau, c_sig, c_tau);
    const int o = e->o;
    int64_t exact = 0, rolled_back = 0, by_position = 0;
    int k = 0, published = 0;                               // iterations that stand / that `progress` has announced
    auto publish = [&](int upto) {
        const int whole = upto == draws ? draws : upto - upto % FOKL_TAPE_BLOCK;
        if (progress && whole > published) {
            _mm_sfence();                                   // the rows were streamed (block_positions)
            __atomic_store_n(progress, whole, __ATOMIC_RELEASE);
        }
        published = std::max(published, whole);
    };
    BlockIn in;
    std::unique_ptr<BlockOut> out(new BlockOut());
    in.b_sig = b_sig;
    in.c_sig = c_sig;
    in.b_tau = b_tau;
    in.c_tau = c_tau;
    while (k < draws && !w.r.failed) {
        int64_t lap_ = g_walk_profile ? now_ns() : 0;
        // ---- pass 1: the chase, inside the segment under the walker ----
        if (w.r.locate(w.D) && w.D - w.r.lo < (uint64_t)kSegDoubles - 64) {
            const Segment *seg = w.r.seg;
            const int a = (int)(w.D & 1);
            const uint64_t q = (w.D - w.r.lo) >> 1;         // attempts of alignment a from slot q on are unread
            int64_t r = (int64_t)seg->cum[a][q >> 6] + __builtin_popcountll(seg->mask[a][q >> 6] & ~(~0ull << (q & 63)));
            int h = w.has_gauss;
            const int want = std::min(kChase, draws - k);
            const int count = chase_block(seg, a, p1, want, r, h, in);
            FOKL_LAP(0);
            if (count > 0) {
                in.seg = seg;
                in.lo = w.r.lo;
                in.a = a;
                in.wc0 = (uint32_t)(q >> 6);
                in.D0 = w.D;
                in.gsrc0 = w.gauss_src;
                in.rows = rows + k;
                // ---- pass 2: positions, rows, the sources of the gamma draws ----
                block_positions(in, *out, o);
                FOKL_LAP(1);
                // ---- pass 3: the accept tests ----
                const int redo = check_tests(checker, o, out->source, out->upos, count, bvec, cvec, b_sig, c_sig, b_tau, c_tau,
                                             exact);
                FOKL_LAP(2);
                if (redo == -2) {
                    w.r.failed = true;
                    break;
                }
                const int good = redo < 0 ? count : redo;   // iterations k .. k + good - 1 stand
                if (good > 0) {
                    w.D = out->after[good - 1].D;
                    w.gauss_src = out->after[good - 1].gauss_src;
                    w.has_gauss = out->after[good - 1].has_gauss;
                    k += good;
                }
                if (redo >= 0) ++rolled_back;
                if (redo < 0 && count == want) {
                    publish(k);
                    continue;
                }
            }
        }
        if (w.r.failed) break;
        // ---- one iteration by the position walk: where the block stopped (a draw to redo, the segment's end, a long run of
        // rejected attempts), from the exact state ----
        if (k < draws) {
            if (g_walk_profile) lap_ = now_ns();
            prefetch_crossing(e, w, p1);
            walk_iteration<true>(w, p1, b_sig, c_sig, b_tau, c_tau, rows[k]);
            ++k;
            ++by_position;
            FOKL_LAP(3);
        }
        publish(k);
    }
    e->exact_draws.fetch_add(exact + w.exact, std::memory_order_relaxed);
    e->gamma_draws.fetch_add(2 * (int64_t)draws, std::memory_order_relaxed);
    e->rollbacks.fetch_add(rolled_back, std::memory_order_relaxed);
    e->position_iterations.fetch_add(by_position, std::memory_order_relaxed);
    if (g_walk_profile) {
        g_walk_iterations.fetch_add(k, std::memory_order_relaxed);
        g_walk_by_position.fetch_add(by_position, std::memory_order_relaxed);
    }
    if (w.r.failed || checker.r.failed) {
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        fokl_set_global_error("fokl_stream_walk: the stream's producers stopped (" + e->error + ")");
        return FOKL_ERR_STATE;
    }
    publish(draws);
    e->D = w.D;
    e->has_gauss = w.has_gauss;
    e->gauss_src = w.gauss_src;
    return FOKL_OK;
}

int walk_tape_base(fokl_stream *e, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                   double *gam_sig, double *gam_tau, int32_t *progress)
{
    return walk_body<false>(e, p1, draws, astar, atau_star, rows, gam_sig, gam_tau, progress);
}

// what the walker may still read: everything from its position on, and the attempt its cached normal comes from
inline uint64_t walker_reach(const fokl_stream *e)
{
    return e->has_gauss && e->gauss_src != kGivenGauss ? std::min(e->D, e->gauss_src) : e->D;
}

}  // namespace

extern "C" int fokl_stream_create(const uint32_t *mt_key, int32_t mt_pos, int32_t has_gauss, double gauss_cache,
                                  int bulk_threads, uint32_t *prestate_ring, int prestate_entries, fokl_stream **out)
{
    if (!out || !mt_key || mt_pos < 0 || mt_pos > MT_N || bulk_threads < 1 || bulk_threads > 16 ||
        (prestate_ring && prestate_entries < 4 * kAheadMax * (kSegBlocks / FOKL_PRESTATE_BLOCKS))) {
        fokl_set_global_error("fokl_stream_create: null pointer, invalid MT19937 position or thread count");
        return FOKL_ERR_ARG;
    }
    auto *e = new fokl_stream();
    std::memcpy(e->key0, mt_key, sizeof(e->key0));
    e->pos0 = mt_pos;
    e->o = mt_pos & 1;
    e->D = (uint64_t)(mt_pos >> 1);
    e->walker_floor = e->D;
    e->has_gauss = has_gauss ? 1 : 0;
    e->gauss_src = kGivenGauss;
    e->gauss0 = gauss_cache;
    e->pre_ring = prestate_ring;
    e->pre_entries = prestate_ring ? prestate_entries : 0;
    e->wide = cpu_is_wide();
    for (auto &t : e->table) t.store(nullptr, std::memory_order_relaxed);
    try {
        for (int i = 0; i < bulk_threads; ++i) e->threads.emplace_back(bulk_worker, e);
    } catch (const std::exception &ex) {
        {
            std::lock_guard<std::mutex> lock(e->token_m);
            e->stop = true;
        }
        {
            std::lock_guard<std::mutex> room(e->room_m);
            e->stop_flag.store(true, std::memory_order_release);
        }
        e->room_cv.notify_all();
        for (auto &t : e->threads) t.join();
        delete e;
        fokl_set_global_error(std::string("fokl_stream_create: ") + ex.what());
        return FOKL_ERR_STATE;
    }
    *out = e;
    return FOKL_OK;
}

extern "C" void fokl_stream_destroy(fokl_stream *e)
{
    if (!e) return;
    if (g_walk_profile && g_walk_iterations.load() > 0) {
        const double n = (double)g_walk_iterations.exchange(0);
        std::fprintf(stderr, "fokl_stream: rank walk, %.0f iterations: chase %.2f positions %.2f tests %.2f ns per iteration; "
                             "position walk %.0f iterations, %.1f ns each\n", n, g_walk_ns[0].exchange(0) / n,
                     g_walk_ns[1].exchange(0) / n, g_walk_ns[2].exchange(0) / n, (double)g_walk_by_position.load(),
                     g_walk_ns[3].exchange(0) / std::max(1.0, (double)g_walk_by_position.load()));
        g_walk_by_position.store(0);
        if (g_crew_tapes.load() > 0)
            std::fprintf(stderr, "fokl_stream: walk crew: %lld tapes, the walking thread inside them %.3f ms, of which waiting for a "
                                 "verdict with no block to take %.3f ms\n", (long long)g_crew_tapes.exchange(0),
                         g_crew_wall_ns.exchange(0) / 1e6, g_crew_idle_ns.exchange(0) / 1e6);
    }
    if (g_walk_profile && e->segments_made.load() > 0) {
        const double n = (double)e->segments_made.load();
        std::fprintf(stderr, "fokl_stream: %.0f segments; per segment: token held %.2f us (recurrence + pre-states %.2f), waited for "
                             "the token %.2f us, all of the bulk phase %.2f us; walker waited %.2f ms, ahead %d\n", n,
                     e->token_ns.load() / 1e3 / n, e->token_recurrence_ns.load() / 1e3 / n, e->token_wait_ns.load() / 1e3 / n,
                     e->bulk_busy_ns.load() / 1e3 / n, e->walker_wait_ns.load() / 1e6, e->ahead.load());
    }
    {
        std::lock_guard<std::mutex> lock(e->token_m);
        e->stop = true;
    }
    {
        std::lock_guard<std::mutex> room(e->room_m);
        e->stop_flag.store(true, std::memory_order_release);
    }
    e->room_cv.notify_all();
    crew_stop(e);
    for (auto &t : e->threads) t.join();
    for (auto &t : e->table) give_segment(t.exchange(nullptr));
    delete e;
}

// The bulk threads' CPUs: thread i may run on logical CPU cpus[i % count] only (round 6: with a segment's words written past
// the cache nothing ties a bulk thread to the walker's last-level cache; a physical core each, elsewhere, is worth a third
// of their time -- next to the driver's, chain and finish threads they ran two to a core).
extern "C" int fokl_stream_place_bulk(fokl_stream *e, const int32_t *cpus, int count)
{
    if (!e || !cpus || count < 1) {
        fokl_set_global_error("fokl_stream_place_bulk: null pointer or no CPUs");
        return FOKL_ERR_ARG;
    }
    int failed = 0;
    for (size_t i = 0; i < e->threads.size(); ++i) {
        cpu_set_t set;
        CPU_ZERO(&set);
        const int cpu = (int)cpus[i % (size_t)count];
        if (cpu < 0 || cpu >= CPU_SETSIZE) continue;
        CPU_SET(cpu, &set);
        failed += pthread_setaffinity_np(e->threads[i].native_handle(), sizeof(set), &set) != 0;
    }
    if (failed) {
        fokl_set_global_error("fokl_stream_place_bulk: pthread_setaffinity_np failed");
        return FOKL_ERR_STATE;
    }
    return FOKL_OK;
}

// Helper threads for the walk (walk_tape_crew): `count` of them (0: none; at most 4), helper i pinned to logical CPU cpus[i]
// (cpus NULL or an entry < 0: not pinned).  Before the first walk; the AVX-512 build only (elsewhere: no effect).
extern "C" int fokl_stream_set_helpers(fokl_stream *e, int count, const int32_t *cpus)
{
    if (!e || count < 0) {
        fokl_set_global_error("fokl_stream_set_helpers: null stream or negative count");
        return FOKL_ERR_ARG;
    }
    crew_stop(e);
    if (count == 0 || !e->wide) return FOKL_OK;
    auto *crew = new WalkCrew();
    crew->e = e;
    crew->helpers = std::min(count, kCrewMax);
    crew->walker_out.reset(new BlockOut());
    for (int i = 0; i < crew->helpers; ++i) crew->cpus.push_back(cpus ? (int)cpus[i] : -1);
    e->crew = crew;
    try {
        for (int i = 0; i < crew->helpers; ++i) crew->threads.emplace_back(crew_helper, crew, i);
    } catch (const std::exception &ex) {
        crew_stop(e);
        fokl_set_global_error(std::string("fokl_stream_set_helpers: ") + ex.what());
        return FOKL_ERR_STATE;
    }
    return FOKL_OK;
}

extern "C" int fokl_stream_walk(fokl_stream *e, int p1, int draws, double astar, double atau_star, fokl_tape_row *rows,
                                double *gam_sig, double *gam_tau, int32_t *progress)
{
    if (!e || p1 <= 0 || draws < 0 || !rows || !gam_sig || !gam_tau) {
        fokl_set_global_error("fokl_stream_walk: null pointer or empty model");
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        return FOKL_ERR_ARG;
    }
    if (!(astar >= 0.0) || !(atau_star >= 0.0)) {
        fokl_set_global_error("fokl_stream_walk: gamma shape parameter is negative or NaN");
        if (progress) __atomic_store_n(progress, -1, __ATOMIC_RELEASE);
        return FOKL_ERR_NUMERIC;
    }
    return e->wide ? walk_tape_wide(e, p1, draws, astar, atau_star, rows, gam_sig, gam_tau, progress)
                   : walk_tape_base(e, p1, draws, astar, atau_star, rows, gam_sig, gam_tau, progress);
}

// segments whose pre-states have been written to the ring so far (they are written in order)
extern "C" int64_t fokl_stream_prestates_published(const fokl_stream *e)
{
    return e ? e->pre_published.load(std::memory_order_acquire) : 0;
}

extern "C" double fokl_stream_given_gauss(const fokl_stream *e) { return e ? e->gauss0 : 0.0; }

extern "C" int fokl_stream_tell(const fokl_stream *e, fokl_stream_cursor *out)
{
    if (!e || !out) {
        fokl_set_global_error("fokl_stream_tell: null pointer");
        return FOKL_ERR_ARG;
    }
    out->position = e->D;
    out->has_gauss = e->has_gauss;
    out->gauss_source = e->gauss_src;
    return FOKL_OK;
}

extern "C" int fokl_stream_seek(fokl_stream *e, const fokl_stream_cursor *at)
{
    if (!e || !at) {
        fokl_set_global_error("fokl_stream_seek: null pointer");
        return FOKL_ERR_ARG;
    }
    {
        std::lock_guard<std::mutex> lock(e->hold_m);
        const uint64_t reach = at->has_gauss && at->gauss_source != kGivenGauss ? std::min(at->position, at->gauss_source)
                                                                                 : at->position;
        if ((int64_t)(reach / kSegDoubles) < e->low_water.load(std::memory_order_acquire)) {
            fokl_set_global_error("fokl_stream_seek: that part of the stream was not held and is gone");
            return FOKL_ERR_STATE;
        }
        e->walker_floor = std::min(e->walker_floor, reach);
        update_low_water(e);
    }
    e->D = at->position;
    e->has_gauss = at->has_gauss ? 1 : 0;
    e->gauss_src = at->gauss_source;
    return FOKL_OK;
}

// Keep the stream readable from the walker's present position on (a tape that begins here; the attempt its cached normal
// comes from included) until fokl_stream_release.
extern "C" int fokl_stream_hold(fokl_stream *e, uint64_t *position_out)
{
    if (!e || !position_out) {
        fokl_set_global_error("fokl_stream_hold: null pointer");
        return FOKL_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(e->hold_m);
    const uint64_t reach = walker_reach(e);
    e->holds.insert(reach);
    *position_out = reach;
    return FOKL_OK;
}

extern "C" int fokl_stream_release(fokl_stream *e, uint64_t position)
{
    if (!e) {
        fokl_set_global_error("fokl_stream_release: null stream");
        return FOKL_ERR_ARG;
    }
    bool moved = false;
    {
        std::lock_guard<std::mutex> lock(e->hold_m);
        auto it = e->holds.find(position);
        if (it == e->holds.end()) {
            fokl_set_global_error("fokl_stream_release: no such hold");
            return FOKL_ERR_ARG;
        }
        e->holds.erase(it);
        const int64_t before = e->low_water.load(std::memory_order_relaxed);
        update_low_water(e);
        moved = e->low_water.load(std::memory_order_relaxed) != before;
    }
    if (moved) {                                            // producers may have been waiting for table entries
        { std::lock_guard<std::mutex> room(e->room_m); }
        e->room_cv.notify_all();
    }
    return FOKL_OK;
}

// The walker lets go of everything behind its present position (between tapes: what is still needed is held).
extern "C" int fokl_stream_advance_floor(fokl_stream *e)
{
    if (!e) {
        fokl_set_global_error("fokl_stream_advance_floor: null stream");
        return FOKL_ERR_ARG;
    }
    bool moved = false;
    {
        std::lock_guard<std::mutex> lock(e->hold_m);
        e->walker_floor = walker_reach(e);
        const int64_t before = e->low_water.load(std::memory_order_relaxed);
        update_low_water(e);
        moved = e->low_water.load(std::memory_order_relaxed) != before;
    }
    if (moved) {
        { std::lock_guard<std::mutex> room(e->room_m); }
        e->room_cv.notify_all();
    }
    return FOKL_OK;
}

// numpy's state tuple at the walker's position: the block that holds the next unread word, raw, and its index there
// (pos = 624 with the block before it when the position is a block boundary: numpy refills lazily).
extern "C" int fokl_stream_state(fokl_stream *e, uint32_t *key_out, int32_t *pos_out, int32_t *has_gauss_out,
                                 double *gauss_out)
{
    if (!e || !key_out || !pos_out || !has_gauss_out || !gauss_out) {
        fokl_set_global_error("fokl_stream_state: null pointer");
        return FOKL_ERR_ARG;
    }
    const uint64_t g = (uint64_t)e->o + 2 * e->D;           // next unread word
    *has_gauss_out = e->has_gauss;
    *gauss_out = 0.0;
    if (e->has_gauss) {
        Walk w(e);
        *gauss_out = w.value_of(e->gauss_src | kCachedHalf);
        if (w.r.failed) {
            fokl_set_global_error("fokl_stream_state: the stream's producers stopped");
            return FOKL_ERR_STATE;
        }
    }
    if (g == (uint64_t)e->pos0) {                           // nothing was drawn
        std::memcpy(key_out, e->key0, sizeof(e->key0));
        *pos_out = e->pos0;
        return FOKL_OK;
    }
    // the raw block to hand back: the one that holds word g, or -- g on a block boundary -- the one before it
    const int64_t target = g % MT_N ? (int64_t)(g / MT_N) : (int64_t)(g / MT_N) - 1;
    *pos_out = g % MT_N ? (int)(g % MT_N) : MT_N;
    if (target == 0) {
        std::memcpy(key_out, e->key0, sizeof(e->key0));
        return FOKL_OK;
    }
    // The segment under the walker is alive (everything from its floor on is); the target block lies in it or is the raw
    // block kept in front of it.  Tempering overwrote the segment's raw words: regenerate from that block.
    const int64_t index = (int64_t)(e->D / kSegDoubles);
    Segment *seg = segment_of(e, e->D, true);
    if (!seg) {
        fokl_set_global_error("fokl_stream_state: the stream's producers stopped");
        return FOKL_ERR_STATE;
    }
    const int64_t local = target - index * kSegBlocks;      // -1: the block in front of the segment
    if (local < -1 || local >= kSegBlocks) {
        fokl_set_global_error("fokl_stream_state: internal error (block outside the walker's segment)");
        return FOKL_ERR_STATE;
    }
    if (local == -1) {
        std::memcpy(key_out, seg->buf, MT_N * sizeof(uint32_t));
        return FOKL_OK;
    }
    std::vector<uint32_t> buf((size_t)MT_N * (size_t)(local + 2));
    int from = MT_N;
    if (index == 0) {
        std::memset(buf.data(), 0, MT_N * sizeof(uint32_t));
        std::memcpy(buf.data() + MT_N, e->key0, MT_N * sizeof(uint32_t));
        from = 2 * MT_N;
    } else {
        std::memcpy(buf.data(), seg->buf, MT_N * sizeof(uint32_t));
    }
    recurrence_portable(buf.data(), from, MT_N * (int)(local + 2));
    std::memcpy(key_out, buf.data() + (size_t)MT_N * (size_t)(local + 1), MT_N * sizeof(uint32_t));
    return FOKL_OK;
}

// Tape rows k0 .. k1 - 1 back into the layout fokl_noise_tape records (include/fokl_hip.h): row k of normals_out
// [draws, p1] = lead_out[k] finished values, (p1 - lead) / 2 accepted pairs as (x2, x1) with r2 in row k of pair_r2_out
// [draws, p1 / 2 + 1], and one more finished value if p1 - lead is odd; gam_sig_out / gam_tau_out [draws] = the two
// standard gammas of the row (b V^3 of the normal the accepted attempt used; astar / atau_star as given to the walk).
// Finished values and gammas are formed with libm's log in numpy's order of operations: numpy's bits.  Any thread; the
// rows' part of the stream must be held.  pair_r2_out may be NULL (consumers that re-form r2 from the pair).
extern "C" int fokl_stream_expand(fokl_stream *e, int p1, double astar, double atau_star, const fokl_tape_row *rows,
                                  int k0, int k1, double *normals_out, double *pair_r2_out, int32_t *lead_out,
                                  double *gam_sig_out, double *gam_tau_out)
{
    if (!e || !rows || !normals_out || !lead_out || !gam_sig_out || !gam_tau_out || p1 <= 0 || k0 < 0 || k1 < k0) {
        fokl_set_global_error("fokl_stream_expand: null pointer, empty model or bad row range");
        return FOKL_ERR_ARG;
    }
    Walk w(e, Walk::ReaderOnly{});                          // only its reader and value_of are used
    Reader &r = w.r;
    const size_t half = (size_t)p1 / 2 + 1;
    const double b_sig = astar - 1.0 / 3.0, c_sig = 1.0 / std::sqrt(9 * b_sig);
    const double b_tau = atau_star - 1.0 / 3.0, c_tau = 1.0 / std::sqrt(9 * b_tau);
    for (int k = k0; k < k1; ++k) {
        const fokl_tape_row &row = rows[k];
        const int lead = (row.start & kLeadBit) ? 1 : 0;
        uint64_t D = row.start & ~kLeadBit;
        double *out = normals_out + (size_t)k * p1;
        double *r2 = pair_r2_out ? pair_r2_out + (size_t)k * half : nullptr;
        lead_out[k] = lead;
        if (lead) out[0] = w.value_of(row.lead_source | kCachedHalf);
        const int rest = p1 - lead, pairs = rest >> 1, attempts = (rest + 1) >> 1;
        const int a = (int)(D & 1);
        int have = 0;
        while (have < attempts) {
            if (!r.locate(D)) break;
            // accepted attempts of this alignment from D to the end of the segment, a mask word at a time
            const uint64_t q = (D - r.lo) >> 1;
            int word = (int)(q >> 6);
            uint64_t m = r.seg->mask[a][word] & (~0ull << (q & 63));
            const uint32_t *wd = r.seg->words() + r.o + a * 2;         // attempt s of the alignment starts at word 4 s
            for (;;) {
                while (m && have < pairs) {
                    const uint64_t s = (uint64_t)word * 64 + (uint64_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t *p = wd + 4 * s;
                    const double x1 = 2.0 * to_double(p[0], p[1]) - 1.0, x2 = 2.0 * to_double(p[2], p[3]) - 1.0;
                    out[lead + 2 * have] = x2;
                    out[lead + 2 * have + 1] = x1;
                    if (r2) r2[have] = x1 * x1 + x2 * x2;
                    ++have;
                }
                if (m && have == pairs && have < attempts) {
                    // the odd one out: its x2 half closes the row (final), its x1 half went to the cache
                    const uint64_t s = (uint64_t)word * 64 + (uint64_t)__builtin_ctzll(m);
                    out[p1 - 1] = w.value_of(r.lo + 2 * s + (uint64_t)a);
                    ++have;
                }
                if (have == attempts || ++word == kSegMaskWords) break;
                m = r.seg->mask[a][word];
            }
            D = r.hi + (uint64_t)a;
        }
        if (row.gamma[0] != kFinalValue) {
            const double X = w.value_of(row.gamma[0]);
            double V = 1.0 + c_sig * X;
            V = V * V * V;
            gam_sig_out[k] = b_sig * V;
        }
        if (row.gamma[1] != kFinalValue) {
            const double X = w.value_of(row.gamma[1]);
            double V = 1.0 + c_tau * X;
            V = V * V * V;
            gam_tau_out[k] = b_tau * V;
        }
        if (r.failed) {
            fokl_set_global_error("fokl_stream_expand: the stream's producers stopped");
            return FOKL_ERR_STATE;
        }
    }
    return FOKL_OK;
}

extern "C" int fokl_stream_stats(const fokl_stream *e, double *bulk_busy_s, double *walker_wait_s, int64_t *segments,
                                 int64_t *gamma_attempts, int64_t *gamma_attempts_exact)
{
    if (!e) {
        fokl_set_global_error("fokl_stream_stats: null stream");
        return FOKL_ERR_ARG;
    }
    if (bulk_busy_s) *bulk_busy_s = 1e-9 * (double)e->bulk_busy_ns.load();
    if (walker_wait_s) *walker_wait_s = 1e-9 * (double)e->walker_wait_ns.load();
    if (segments) *segments = e->segments_made.load();
    if (gamma_attempts) *gamma_attempts = e->gamma_draws.load();
    if (gamma_attempts_exact) *gamma_attempts_exact = e->exact_draws.load();
    return FOKL_OK;
}

// |fast_ln(y) - log(y)| over n values spread over (0, 1) (tests: the bound the walker's squeeze decisions rest on)
extern "C" double fokl_stream_fast_ln_error(int64_t n)
{
    double worst = 0.0;
    for (int64_t i = 1; i <= n; ++i) {
        const double u = (double)i / (double)(n + 1);
        for (const double y : {u, u * u * u * 1e-3, std::ldexp(u, -104 + (int)(i % 100))}) {
            if (!(y > 0.0) || y >= 1.0) continue;
            worst = std::max(worst, std::fabs(fast_ln(y) - std::log(y)));
        }
    }
    return worst;
}
