// pack_host.hpp -- ASCII -> 2-bit query packing on the host (gdx_pack_queries / gdx_pack_queries_table, include/gdx.h
// "packed queries"; /root/reference ROADMAP.md:35-37 names reading and preparing the queries as a cost of its own).
// Symbol j of the buffer becomes bits 2 (j & 3) .. 2 (j & 3) + 1 of byte j >> 2 = (dense code - 1) of one of the dense
// symbols 1..4; any other byte is an exception (code 0, reported through `bad`).
//
// The byte-at-a-time loop through the alphabet's 256-entry table ran at 9.5 GB/s on 16 threads -- slower than shipping
// the ASCII reads over PCIe.  When the table has the shape of a nucleotide alphabet (which of A C G T a byte is follows
// from its low nibble, whether it is one at all from low and high nibble together: make_pack_plan checks exactly that,
// for any table), 32 symbols are translated by three nibble look-ups (pshufb), validated by one compare and packed by two
// multiply-adds.  Tables of another shape, and CPUs without AVX2, take the scalar loop.
//
// Header-only so that tests/host_checks compiles the same code under AddressSanitizer.
#pragma once

#include <cstdint>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define GDX_PACK_AVX2 1
#endif

namespace gdx {

struct PackPlan {
    bool fast = false;     // the nibble look-ups below reproduce the table exactly
    uint8_t lo_code[16];   // low nibble -> (dense code - 1) of the searchable bytes with that nibble
    uint8_t lo_class[16];  // low nibble -> one bit: the set of high nibbles that make such a byte searchable
    uint8_t hi_class[16];  // high nibble -> the bits of the sets it belongs to
};

// searchable here: dense code 1..4 (the four symbols 2 bits can name)
inline PackPlan make_pack_plan(const uint8_t *tab)
{
    PackPlan p;
    uint16_t sets[8];
    int n_sets = 0;
    for (int l = 0; l < 16; l++) {
        p.lo_code[l] = 0;
        p.lo_class[l] = 0;
    }
    for (int h = 0; h < 16; h++) p.hi_class[h] = 0;
    for (int l = 0; l < 16; l++) {
        uint16_t highs = 0;
        int code = -1;
        for (int h = 0; h < 16; h++) {
            const unsigned d = tab[(h << 4) | l];
            if (d - 1u < 4u) {
                if (code >= 0 && code != static_cast<int>(d - 1u)) return p;  // two symbols share a low nibble: scalar
                code = static_cast<int>(d - 1u);
                highs = static_cast<uint16_t>(highs | (1u << h));
            }
        }
        if (highs == 0) continue;
        int k = 0;
        while (k < n_sets && sets[k] != highs) k++;
        if (k == n_sets) {
            if (n_sets == 8) return p;  // more shapes than a byte has bits: scalar
            sets[n_sets++] = highs;
        }
        p.lo_code[l] = static_cast<uint8_t>(code);
        p.lo_class[l] = static_cast<uint8_t>(1u << k);
    }
    for (int k = 0; k < n_sets; k++)
        for (int h = 0; h < 16; h++)
            if (sets[k] & (1u << h)) p.hi_class[h] = static_cast<uint8_t>(p.hi_class[h] | (1u << k));
    p.fast = true;
    return p;
}

#ifdef GDX_PACK_AVX2
// 32 symbols -> their packed bytes, one in the low byte of each of the eight dwords; *bad = the mask of the symbols that are
// not searchable (their codes are 0)
__attribute__((target("avx2"))) inline __m256i pack32_avx2(const uint8_t *src, __m256i lo_code, __m256i lo_class, __m256i hi_class,
                                                           uint32_t *bad_mask)
{
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src));
    const __m256i nib = _mm256_set1_epi8(0x0f);
    const __m256i lo = _mm256_and_si256(v, nib), hi = _mm256_and_si256(_mm256_srli_epi16(v, 4), nib);
    const __m256i ok = _mm256_and_si256(_mm256_shuffle_epi8(lo_class, lo), _mm256_shuffle_epi8(hi_class, hi));
    const __m256i bad = _mm256_cmpeq_epi8(ok, _mm256_setzero_si256());
    const __m256i code = _mm256_andnot_si256(bad, _mm256_shuffle_epi8(lo_code, lo));
    // bytes c0 c1 c2 c3 -> c0 + 4 c1 + 16 c2 + 64 c3 in the low byte of their dword
    const __m256i pairs = _mm256_maddubs_epi16(code, _mm256_set1_epi16(0x0401));
    *bad_mask = static_cast<uint32_t>(_mm256_movemask_epi8(bad));
    return _mm256_madd_epi16(pairs, _mm256_set1_epi32(0x00100001));
}

template <class Bad>
__attribute__((target("avx2"))) inline void pack_span_avx2(const PackPlan &p, const uint8_t *qbuf, uint64_t j0, uint64_t j1,
                                                           uint8_t *out_packed, Bad &bad_symbol)
{
    // symbols [j0, j1), j0 and j1 multiples of 32: 128 symbols -> 32 output bytes per step, then 32 -> 8
    const __m128i lc = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p.lo_code));
    const __m128i ls = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p.lo_class));
    const __m128i hs = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p.hi_class));
    const __m256i lo_code = _mm256_broadcastsi128_si256(lc), lo_class = _mm256_broadcastsi128_si256(ls),
                  hi_class = _mm256_broadcastsi128_si256(hs);
    auto report = [&](uint64_t j, uint32_t bad) {
        while (bad != 0u) {
            const uint32_t k = static_cast<uint32_t>(__builtin_ctz(bad));
            bad &= bad - 1u;
            bad_symbol(j + k);
        }
    };
    uint64_t j = j0;
    const __m256i order = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    for (; j + 128 <= j1; j += 128) {
        uint32_t b0, b1, b2, b3;
        const __m256i q0 = pack32_avx2(qbuf + j, lo_code, lo_class, hi_class, &b0);
        const __m256i q1 = pack32_avx2(qbuf + j + 32, lo_code, lo_class, hi_class, &b1);
        const __m256i q2 = pack32_avx2(qbuf + j + 64, lo_code, lo_class, hi_class, &b2);
        const __m256i q3 = pack32_avx2(qbuf + j + 96, lo_code, lo_class, hi_class, &b3);
        // dwords -> bytes: two in-lane packs leave {q0 lo, q1 lo, q2 lo, q3 lo | q0 hi, ...}; one dword permutation sorts them
        const __m256i bytes = _mm256_packus_epi16(_mm256_packus_epi32(q0, q1), _mm256_packus_epi32(q2, q3));
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(out_packed + (j >> 2)), _mm256_permutevar8x32_epi32(bytes, order));
        if ((b0 | b1 | b2 | b3) != 0u) {
            report(j, b0);
            report(j + 32, b1);
            report(j + 64, b2);
            report(j + 96, b3);
        }
    }
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1,
                                          -1, -1, -1, -1, -1, -1, -1);
    for (; j < j1; j += 32) {
        uint32_t b;
        const __m256i bytes = _mm256_shuffle_epi8(pack32_avx2(qbuf + j, lo_code, lo_class, hi_class, &b), pick);
        const uint64_t out = static_cast<uint32_t>(_mm256_cvtsi256_si32(bytes)) |
                             (static_cast<uint64_t>(static_cast<uint32_t>(_mm256_extract_epi32(bytes, 4))) << 32);
        __builtin_memcpy(out_packed + (j >> 2), &out, 8);
        report(j, b);
    }
}
#endif

inline bool pack_have_avx2()
{
#ifdef GDX_PACK_AVX2
    static const bool have = __builtin_cpu_supports("avx2");
    return have;
#else
    return false;
#endif
}

// Packs the symbols of output bytes [lo_byte, hi_byte): symbol j (lo_byte * 4 <= j < hi_byte * 4) with first <= j < n_sym is
// translated, any other position gets code 0.  bad_symbol(j) is called, in ascending order of j, for every symbol that is not
// searchable; the caller turns positions into queries.
template <class Bad>
inline void pack_range(const PackPlan &plan, const uint8_t *tab, const uint8_t *qbuf, uint64_t first, uint64_t n_sym, uint64_t lo_byte,
                       uint64_t hi_byte, uint8_t *out_packed, Bad &&bad_symbol)
{
    auto scalar = [&](uint64_t b0, uint64_t b1) {
        for (uint64_t b = b0; b < b1; b++) {
            uint32_t out = 0;
            for (uint32_t k = 0; k < 4; k++) {
                const uint64_t j = 4 * b + k;
                if (j < first || j >= n_sym) continue;
                const uint32_t d = tab[qbuf[j]];
                if (d - 1u < 4u) {
                    out |= (d - 1u) << (2u * k);
                } else {
                    bad_symbol(j);
                }
            }
            out_packed[b] = static_cast<uint8_t>(out);
        }
    };
#ifdef GDX_PACK_AVX2
    if (plan.fast && pack_have_avx2()) {
        // whole 32-symbol groups that lie inside [first, n_sym): the vector loop; what is left at either end: scalar
        uint64_t v0 = (lo_byte * 4 > first ? lo_byte * 4 : first);
        v0 = (v0 + 31) / 32 * 32;
        uint64_t v1 = (hi_byte * 4 < n_sym ? hi_byte * 4 : n_sym) / 32 * 32;
        if (v1 > v0) {
            scalar(lo_byte, v0 / 4);
            pack_span_avx2(plan, qbuf, v0, v1, out_packed, bad_symbol);
            scalar(v1 / 4, hi_byte);
            return;
        }
    }
#else
    (void)plan;
#endif
    scalar(lo_byte, hi_byte);
}

}  // namespace gdx
