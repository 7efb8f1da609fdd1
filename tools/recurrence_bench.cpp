// The MT19937 recurrence over a flat array (csrc/fokl_stream.cpp recurrence_wide) in isolation: microseconds per segment
// of 256 blocks for the shipped loop and for variants (development aid; run on the GPU box's host CPU).
//   g++ -O3 -std=c++17 -mavx512f -mavx512dq -mavx512vl -mavx512bw -o /tmp/recurrence_bench tools/recurrence_bench.cpp
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int MT_N = 624, MT_SHIFT = 397 - 624 + 624;   // w[g] = w[g - 227] ^ twist(w[g - 624], w[g - 623])
static inline uint32_t twist(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

static void scalar(uint32_t *buf, int from, int to)
{
    for (int g = from; g < to; ++g) buf[g] = buf[g - 227] ^ twist(buf[g - MT_N], buf[g - MT_N + 1]);
}

static void shipped(uint32_t *buf, int from, int to)
{
    const __m512i upper = _mm512_set1_epi32((int)0x80000000u), mag = _mm512_set1_epi32((int)0x9908b0dfu);
    const __m512i one = _mm512_set1_epi32(1);
    int g = from;
    __m512i a = _mm512_load_si512(buf + g - MT_N);
    __m512i c_lo = _mm512_load_si512(buf + g - 240);
    for (; g + 16 <= to; g += 16) {
        const __m512i a_next = _mm512_load_si512(buf + g - MT_N + 16);
        const __m512i c_hi = _mm512_load_si512(buf + g - 224);
        const __m512i b = _mm512_alignr_epi32(a_next, a, 1);
        const __m512i c = _mm512_alignr_epi32(c_hi, c_lo, 13);
        const __m512i y = _mm512_ternarylogic_epi32(upper, a, b, 0xca);
        const __mmask16 odd = _mm512_test_epi32_mask(y, one);
        __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));
        r = _mm512_mask_xor_epi32(r, odd, r, mag);
        _mm512_store_si512(buf + g, r);
        a = a_next;
        c_lo = c_hi;
    }
    for (; g < to; ++g) buf[g] = buf[g - 227] ^ twist(buf[g - MT_N], buf[g - MT_N + 1]);
}

// the last 15 results stay in registers: the operand 227 words back never comes through memory
static void window(uint32_t *buf, int from, int to)
{
    const __m512i upper = _mm512_set1_epi32((int)0x80000000u), mag = _mm512_set1_epi32((int)0x9908b0dfu);
    const __m512i one = _mm512_set1_epi32(1);
    int g = from;
    __m512i w[15];                                  // w[k] = words g - 240 + 16 k ..
    for (int k = 0; k < 15; ++k) w[k] = _mm512_load_si512(buf + g - 240 + 16 * k);
    __m512i a = _mm512_load_si512(buf + g - MT_N);
#define STEP(K0, K1, KN)                                                                    \
    {                                                                                       \
        const __m512i a_next = _mm512_load_si512(buf + g - MT_N + 16);                      \
        const __m512i b = _mm512_alignr_epi32(a_next, a, 1);                                \
        const __m512i c = _mm512_alignr_epi32(w[K1], w[K0], 13);                            \
        const __m512i y = _mm512_ternarylogic_epi32(upper, a, b, 0xca);                     \
        const __mmask16 odd = _mm512_test_epi32_mask(y, one);                               \
        __m512i r = _mm512_xor_si512(c, _mm512_srli_epi32(y, 1));                           \
        r = _mm512_mask_xor_epi32(r, odd, r, mag);                                          \
        _mm512_store_si512(buf + g, r);                                                     \
        w[KN] = r;                                                                          \
        a = a_next;                                                                         \
        g += 16;                                                                            \
    }
    // ring of 15: at each step the oldest slot (K0) is consumed together with K1 and then overwritten by the new vector
    while (g + 16 * 15 <= to) {
        STEP(0, 1, 0) STEP(1, 2, 1) STEP(2, 3, 2) STEP(3, 4, 3) STEP(4, 5, 4) STEP(5, 6, 5) STEP(6, 7, 6) STEP(7, 8, 7)
        STEP(8, 9, 8) STEP(9, 10, 9) STEP(10, 11, 10) STEP(11, 12, 11) STEP(12, 13, 12) STEP(13, 14, 13) STEP(14, 0, 14)
    }
#undef STEP
    if (g < to) shipped(buf, g, to);
}

// 256-bit vectors
static void half_width(uint32_t *buf, int from, int to)
{
    const __m256i upper = _mm256_set1_epi32((int)0x80000000u), mag = _mm256_set1_epi32((int)0x9908b0dfu);
    const __m256i one = _mm256_set1_epi32(1);
    int g = from;
    for (; g + 8 <= to; g += 8) {
        const __m256i a = _mm256_load_si256((const __m256i *)(buf + g - MT_N));
        const __m256i b = _mm256_loadu_si256((const __m256i *)(buf + g - MT_N + 1));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(buf + g - 227));
        const __m256i y = _mm256_ternarylogic_epi32(upper, a, b, 0xca);
        const __mmask8 odd = _mm256_test_epi32_mask(y, one);
        __m256i r = _mm256_xor_si256(c, _mm256_srli_epi32(y, 1));
        r = _mm256_mask_xor_epi32(r, odd, r, mag);
        _mm256_store_si256((__m256i *)(buf + g), r);
    }
    for (; g < to; ++g) buf[g] = buf[g - 227] ^ twist(buf[g - MT_N], buf[g - MT_N + 1]);
}

int main()
{
    constexpr int kBlocks = 256, kWords = kBlocks * MT_N;
    uint32_t *buf = static_cast<uint32_t *>(aligned_alloc(64, sizeof(uint32_t) * (MT_N + kWords + 64)));
    std::vector<uint32_t> ref(MT_N + kWords);
    for (int i = 0; i < MT_N; ++i) ref[i] = 1812433253u * (uint32_t)(i + 7) + 12345u * (uint32_t)i;
    scalar(ref.data(), MT_N, MT_N + kWords);
    struct {
        const char *name;
        void (*fn)(uint32_t *, int, int);
    } variants[] = {{"scalar", scalar}, {"shipped (512-bit, operands from memory)", shipped},
                    {"register window of 15 vectors", window}, {"256-bit, unaligned loads", half_width}};
    for (auto &v : variants) {
        double best = 1e9;
        for (int rep = 0; rep < 40; ++rep) {
            std::memcpy(buf, ref.data(), sizeof(uint32_t) * MT_N);
            const auto t0 = std::chrono::steady_clock::now();
            v.fn(buf, MT_N, MT_N + kWords);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        const bool same = std::memcmp(buf, ref.data(), sizeof(uint32_t) * (MT_N + kWords)) == 0;
        std::printf("%-45s %7.2f us per segment of %d blocks  (%s)\n", v.name, best, kBlocks, same ? "bits ok" : "BITS DIFFER");
    }
    std::free(buf);
    return 0;
}
