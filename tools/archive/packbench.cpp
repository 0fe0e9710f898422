// packbench — how should the host lay a crowd's world matrices out in the pinned ring slot? memcpy of 64 B/bone against three ways of keeping the upper
// three rows (48 B/bone), 51 200 bones (C4), eight destination slots cycled like the ring.   g++ -O3 -std=c++17 tools/archive/packbench.cpp -o tools/archive/packbench
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__attribute__((target("avx512f"))) static bool pack_a(const float *world, size_t bones, float *out)
{
    const __m512i pick = _mm512_setr_epi32(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 3, 7, 11, 15);
    const __m512i bottom = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x3f800000);
    __mmask16 bad = 0;
    for (size_t b = 0; b < bones; ++b) {
        const __m512 m = _mm512_permutexvar_ps(pick, _mm512_loadu_ps(world + b * 16));
        _mm512_mask_storeu_ps(out + b * 12, (__mmask16)0x0fff, m);
        bad |= _mm512_mask_cmpneq_epi32_mask((__mmask16)0xf000, _mm512_castps_si512(m), bottom);
    }
    return bad == 0;
}
// four bones -> three full vectors
template <bool NT> __attribute__((target("avx512f"))) static bool pack_b(const float *world, size_t bones, float *out)
{
    // out0 = m0[0,1,2,4,5,6,8,9,10,12,13,14] m1[0,1,2,4]; out1 = m1[5,6,8,9,10,12,13,14] m2[0,1,2,4,5,6,8,9]; out2 = m2[10,12,13,14] m3[0,1,2,4,5,6,8,9,10,12,13,14]
    const __m512i i0 = _mm512_setr_epi32(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4);
    const __m512i i1 = _mm512_setr_epi32(5, 6, 8, 9, 10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4, 16 + 5, 16 + 6, 16 + 8, 16 + 9);
    const __m512i i2 = _mm512_setr_epi32(10, 12, 13, 14, 16 + 0, 16 + 1, 16 + 2, 16 + 4, 16 + 5, 16 + 6, 16 + 8, 16 + 9, 16 + 10, 16 + 12, 16 + 13, 16 + 14);
    const __m512i bottom = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x3f800000);
    __m512i acc = _mm512_setzero_si512();
    size_t b = 0;
    for (; b + 4 <= bones; b += 4) {
        const __m512 m0 = _mm512_loadu_ps(world + b * 16), m1 = _mm512_loadu_ps(world + b * 16 + 16), m2 = _mm512_loadu_ps(world + b * 16 + 32), m3 = _mm512_loadu_ps(world + b * 16 + 48);
        const __m512 o0 = _mm512_permutex2var_ps(m0, i0, m1), o1 = _mm512_permutex2var_ps(m1, i1, m2), o2 = _mm512_permutex2var_ps(m2, i2, m3);
        if (NT) { _mm512_stream_ps(out + b * 12, o0); _mm512_stream_ps(out + b * 12 + 16, o1); _mm512_stream_ps(out + b * 12 + 32, o2); }
        else { _mm512_storeu_ps(out + b * 12, o0); _mm512_storeu_ps(out + b * 12 + 16, o1); _mm512_storeu_ps(out + b * 12 + 32, o2); }
        // bottom rows: lanes 3, 7, 11, 15 of every matrix must be 0 0 0 1 (bit patterns)
        __m512i x = _mm512_xor_si512(_mm512_castps_si512(m0), bottom);
        x = _mm512_or_si512(x, _mm512_xor_si512(_mm512_castps_si512(m1), bottom));
        x = _mm512_or_si512(x, _mm512_xor_si512(_mm512_castps_si512(m2), bottom));
        x = _mm512_or_si512(x, _mm512_xor_si512(_mm512_castps_si512(m3), bottom));
        acc = _mm512_or_si512(acc, x);
    }
    __mmask16 bad = _mm512_mask_test_epi32_mask((__mmask16)0x8888, acc, acc);
    const __m512i pick = _mm512_setr_epi32(0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14, 3, 7, 11, 15);
    for (; b < bones; ++b) {
        const __m512 m = _mm512_loadu_ps(world + b * 16);
        _mm512_mask_storeu_ps(out + b * 12, (__mmask16)0x0fff, _mm512_permutexvar_ps(pick, m));
        bad |= _mm512_mask_cmpneq_epi32_mask((__mmask16)0x8888, _mm512_castps_si512(m), bottom);
    }
    if (NT) _mm_sfence();
    return bad == 0;
}
int main()
{
    const size_t bones = 51200;
    std::vector<float> w(bones * 16);
    for (size_t b = 0; b < bones; ++b) { for (int e = 0; e < 16; ++e) w[b * 16 + e] = (float)rand() / RAND_MAX; w[b*16+3]=w[b*16+7]=w[b*16+11]=0; w[b*16+15]=1; }
    const int slots = 8;
    std::vector<float *> out(slots);
    for (auto &p : out) p = (float *)aligned_alloc(4096, bones * 64);
    auto bench = [&](const char *name, auto &&fn) {
        for (int i = 0; i < 50; ++i) fn(out[i % slots]);
        auto t0 = std::chrono::steady_clock::now();
        const int n = 400;
        bool ok = true;
        for (int i = 0; i < n; ++i) ok &= fn(out[i % slots]);
        printf("%-28s %.1f us  ok=%d\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n, ok);
    };
    bench("memcpy 64 B/bone", [&](float *o) { memcpy(o, w.data(), bones * 64); return true; });
    bench("memcpy 48 B/bone worth", [&](float *o) { memcpy(o, w.data(), bones * 48); return true; });
    bench("masked store per bone", [&](float *o) { return pack_a(w.data(), bones, o); });
    bench("4 bones -> 3 vectors", [&](float *o) { return pack_b<false>(w.data(), bones, o); });
    bench("4 bones -> 3 vectors, NT", [&](float *o) { return pack_b<true>(w.data(), bones, o); });
    // verify b == a
    pack_a(w.data(), bones, out[0]); pack_b<false>(w.data(), bones, out[1]);
    printf("same: %d\n", memcmp(out[0], out[1], bones * 48) == 0);
    w[15] = 2.0f; printf("bad detected: %d %d\n", !pack_a(w.data(), bones, out[0]), !pack_b<false>(w.data(), bones, out[1]));
    w[15] = 1.0f; w[16*777+7] = -0.0f; printf("neg zero detected: %d %d\n", !pack_a(w.data(), bones, out[0]), !pack_b<false>(w.data(), bones, out[1]));
}
