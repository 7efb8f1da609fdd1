// scratch: cost of the bulk half of the stream per 624-word block (refill / convert / polar coordinates)
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <chrono>
constexpr int MT_N = 624, MT_M = 397;
static void next_state(uint32_t *k) {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
    for (int i = 0; i < MT_N - MT_M; ++i) { const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER); k[i] = k[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG); }
    for (int i = MT_N - MT_M; i < MT_N - 1; ++i) { const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER); k[i] = k[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG); }
    const uint32_t y = (k[MT_N - 1] & UPPER) | (k[0] & LOWER); k[MT_N - 1] = k[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MAG);
}
static void words_to_doubles(const uint32_t *__restrict__ k, int count, double *__restrict__ out) {
    for (int j = 0; j < count; ++j) { uint32_t wa = k[2 * j], wb = k[2 * j + 1];
        wa ^= (wa >> 11); wb ^= (wb >> 11); wa ^= (wa << 7) & 0x9d2c5680u; wb ^= (wb << 7) & 0x9d2c5680u; wa ^= (wa << 15) & 0xefc60000u; wb ^= (wb << 15) & 0xefc60000u; wa ^= (wa >> 18); wb ^= (wb >> 18);
        out[j] = ((double)(int32_t)(wa >> 5) * 67108864.0 + (double)(int32_t)(wb >> 6)) / 9007199254740992.0; } }
static void polar_coordinates(const double *__restrict__ d, int count, double *__restrict__ x, double *__restrict__ sq) {
    for (int j = 0; j < count; ++j) { const double v = 2.0 * d[j] - 1.0; x[j] = v; sq[j] = v * v; } }
int main() { uint32_t key[MT_N]; for (int i = 0; i < MT_N; i++) key[i] = i * 2654435761u + 1; double d[312], x[312], s[312]; double sink = 0; const int R = 200000;
    auto t0 = std::chrono::steady_clock::now(); for (int r = 0; r < R; r++) { next_state(key); sink += key[r % 624]; } auto t1 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; r++) { next_state(key); words_to_doubles(key, 312, d); sink += d[r % 312]; } auto t2 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; r++) { next_state(key); words_to_doubles(key, 312, d); polar_coordinates(d, 312, x, s); sink += x[r % 312] + s[r % 312]; } auto t3 = std::chrono::steady_clock::now();
    auto ns = [&](auto a, auto b) { return std::chrono::duration<double, std::nano>(b - a).count() / R; };
    printf("per block: refill %.0f ns, +convert %.0f ns, +polar %.0f ns (sink %g)\n", ns(t0, t1), ns(t1, t2), ns(t2, t3), sink); }
