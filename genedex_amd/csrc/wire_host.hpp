// wire_host.hpp -- the host's end of the "found bitmap" wire (locate.hip wire_pack_kernel; gdx.h gdx_wire_pack_dev): a chunk's
// results cross PCIe as a bit per read, 4 bytes per found read and the exceptions with their hits -- 3.7 bytes per read where
// nine in ten are found (4.6 with text ids) -- and host threads expand them into what gdx_locate_many_alloc_layout32 returns: u32 hit offsets
// and 8-byte {text id, position} hits (lib.rs:187-197 locate: the hits of query i at [off[i], off[i + 1]) in suffix-array
// order).  Written by the device those cost 12.2 bytes per read of the link both directions share (4 offsets + 7.2 hits + 1
// status byte against 12.5 bytes of 2-bit reads going in); the link, not the kernels, bounds the call.
//
// Tiles of 2048 reads are independent: the device sends, per tile, the found reads and the hits in front of it.
// Header-only so that tests/host_checks compiles the same code under AddressSanitizer.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/gdx.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define GDX_WIRE_AVX2 1
#endif

namespace gdx {

constexpr uint64_t kHostWireTile = 2048;  // == kWireTileReads (kernels.hpp)

struct HostWire {
    const uint8_t *bitmap;       // a bit per read: exactly one hit, the next entry of found_pos
    const uint32_t *tile_found;  // found reads in front of every tile (n_tiles + 1)
    const uint32_t *tile_off;    // hits in front of every tile (n_tiles + 1)
    const uint32_t *found_pos;   // the found reads' positions in their text, read order; ONE MORE element than there are found
                                 // reads must be readable (of found_ids too; the values do not matter): a loop below loads before
                                 // it knows whether it stores
    const uint8_t *found_ids;    // their text ids; null: a collection of one text (the device does the lookup: it keeps the text
                                 // table in LDS, a host thread would spend more on it than on everything else)
    const uint32_t *exc_q, *exc_cnt;  // the other reads with hits (or a status): read number (ascending), number of hits
    const gdx_hit32_t *exc_hits;      // their hits back to back
    uint64_t n_exc;
};

// p[v][k] = the bits set among bits 0 .. k - 1 of v, n[v] = all of them
struct BitPrefixTable {
    uint8_t p[256][8];
    uint8_t n[256];
    BitPrefixTable()
    {
        for (int v = 0; v < 256; v++) {
            int c = 0;
            for (int k = 0; k < 8; k++) {
                p[v][k] = static_cast<uint8_t>(c);
                c += (v >> k) & 1;
            }
            n[v] = static_cast<uint8_t>(c);
        }
    }
};
inline const BitPrefixTable &bit_prefix_table()
{
    static const BitPrefixTable t;
    return t;
}

inline bool wire_have_avx2()
{
#ifdef GDX_WIRE_AVX2
    static const bool have = __builtin_cpu_supports("avx2");
    return have;
#else
    return false;
#endif
}

// The arrays written here are read next by the caller, not by these threads, and the link's DMA shares the memory bus: the
// vector paths store past the caches (no line is read in order to be overwritten), every call ends with a store fence.
// Two result forms: u32 offsets + gdx_hit32_t (gdx_locate_many_alloc_layout32) and u64 offsets + gdx_hit_t (gdx_locate_many*).
#ifdef GDX_WIRE_AVX2
__attribute__((target("avx2"))) inline void expand_offsets_avx2(const uint8_t *bitmap, uint64_t n_bytes, uint32_t o, uint32_t *out)
{
    const BitPrefixTable &t = bit_prefix_table();
    const bool aligned = (reinterpret_cast<uintptr_t>(out) & 31u) == 0;
    for (uint64_t j = 0; j < n_bytes; j++) {
        const uint8_t v = bitmap[j];
        const __m128i pre = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(t.p[v]));
        const __m256i x = _mm256_add_epi32(_mm256_cvtepu8_epi32(pre), _mm256_set1_epi32(static_cast<int>(o)));
        if (aligned) _mm256_stream_si256(reinterpret_cast<__m256i *>(out + 8 * j), x);
        else _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + 8 * j), x);
        o += t.n[v];
    }
}
__attribute__((target("avx2"))) inline void expand_offsets_avx2(const uint8_t *bitmap, uint64_t n_bytes, uint64_t o, uint64_t *out)
{
    const BitPrefixTable &t = bit_prefix_table();
    const bool aligned = (reinterpret_cast<uintptr_t>(out) & 31u) == 0;
    for (uint64_t j = 0; j < n_bytes; j++) {
        const uint8_t v = bitmap[j];
        const __m128i pre = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(t.p[v]));
        const __m256i base = _mm256_set1_epi64x(static_cast<long long>(o));
        const __m256i x0 = _mm256_add_epi64(_mm256_cvtepu8_epi64(pre), base), x1 = _mm256_add_epi64(_mm256_cvtepu8_epi64(_mm_srli_si128(pre, 4)), base);
        if (aligned) {
            _mm256_stream_si256(reinterpret_cast<__m256i *>(out + 8 * j), x0);
            _mm256_stream_si256(reinterpret_cast<__m256i *>(out + 8 * j + 4), x1);
        } else {
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + 8 * j), x0);
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + 8 * j + 4), x1);
        }
        o += t.n[v];
    }
}
// hits[k] = {ids[k] (0 when ids is null), pos[k]}
__attribute__((target("avx2"))) inline void expand_positions_avx2(const uint32_t *pos, const uint8_t *ids, uint64_t n, gdx_hit32_t *hits)
{
    uint64_t k = 0;
    for (; k < n && (reinterpret_cast<uintptr_t>(hits + k) & 31u) != 0; k++) {
        hits[k].text_id = ids ? ids[k] : 0u;
        hits[k].position = pos[k];
    }
    if (ids == nullptr) {
        for (; k + 4 <= n; k += 4) {  // {text 0, position}: the position in the upper half of a 64-bit word
            const __m256i wide = _mm256_cvtepu32_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i *>(pos + k)));
            _mm256_stream_si256(reinterpret_cast<__m256i *>(hits + k), _mm256_slli_epi64(wide, 32));
        }
    } else {
        for (; k + 8 <= n; k += 8) {  // eight {id, position} pairs: two in-lane interleaves, then the lanes sorted
            const __m256i id = _mm256_cvtepu8_epi32(_mm_loadl_epi64(reinterpret_cast<const __m128i *>(ids + k)));
            const __m256i p = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pos + k));
            const __m256i lo = _mm256_unpacklo_epi32(id, p), hi = _mm256_unpackhi_epi32(id, p);  // {0 1 | 4 5}, {2 3 | 6 7}
            _mm256_stream_si256(reinterpret_cast<__m256i *>(hits + k), _mm256_permute2x128_si256(lo, hi, 0x20));
            _mm256_stream_si256(reinterpret_cast<__m256i *>(hits + k + 4), _mm256_permute2x128_si256(lo, hi, 0x31));
        }
    }
    for (; k < n; k++) {
        hits[k].text_id = ids ? ids[k] : 0u;
        hits[k].position = pos[k];
    }
}
__attribute__((target("avx2"))) inline void expand_positions_avx2(const uint32_t *pos, const uint8_t *ids, uint64_t n, gdx_hit_t *hits)
{
    uint64_t k = 0;
    for (; k < n && (reinterpret_cast<uintptr_t>(hits + k) & 31u) != 0; k++) {
        hits[k].text_id = ids ? ids[k] : 0u;
        hits[k].position = pos[k];
    }
    for (; k + 4 <= n; k += 4) {  // four {id, position} pairs of 16 bytes
        uint32_t id4 = 0;
        if (ids != nullptr) std::memcpy(&id4, ids + k, 4);
        const __m256i id = _mm256_cvtepu8_epi64(_mm_cvtsi32_si128(static_cast<int>(id4)));
        const __m256i p = _mm256_cvtepu32_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i *>(pos + k)));
        const __m256i lo = _mm256_unpacklo_epi64(id, p), hi = _mm256_unpackhi_epi64(id, p);  // {0 | 2}, {1 | 3}
        _mm256_stream_si256(reinterpret_cast<__m256i *>(hits + k), _mm256_permute2x128_si256(lo, hi, 0x20));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(hits + k + 2), _mm256_permute2x128_si256(lo, hi, 0x31));
    }
    for (; k < n; k++) {
        hits[k].text_id = ids ? ids[k] : 0u;
        hits[k].position = pos[k];
    }
}
#endif

// out[8 j + k] = o + the bits set among the first 8 j + k bits of the bitmap (n_bytes whole bytes): the hit offset of every read
template <class OffT>
inline void expand_offsets(const uint8_t *bitmap, uint64_t n_bytes, OffT o, OffT *out)
{
#ifdef GDX_WIRE_AVX2
    if (wire_have_avx2()) {
        expand_offsets_avx2(bitmap, n_bytes, o, out);
        return;
    }
#endif
    const BitPrefixTable &t = bit_prefix_table();
    for (uint64_t j = 0; j < n_bytes; j++) {
        const uint8_t v = bitmap[j];
        for (int k = 0; k < 8; k++) out[8 * j + k] = o + t.p[v][k];
        o += t.n[v];
    }
}

template <class HitT>
inline void expand_positions(const uint32_t *pos, const uint8_t *ids, uint64_t n, HitT *hits)
{
#ifdef GDX_WIRE_AVX2
    if (wire_have_avx2()) {
        expand_positions_avx2(pos, ids, n, hits);
        return;
    }
#endif
    for (uint64_t k = 0; k < n; k++) {
        hits[k].text_id = ids ? ids[k] : 0u;
        hits[k].position = pos[k];
    }
}

// out[i] = in[i] widened (counts and interval borders leave the device as u32, the ABI takes u64), stored past the caches
#ifdef GDX_WIRE_AVX2
__attribute__((target("avx2"))) inline void widen_u32_avx2(const uint32_t *in, uint64_t n, uint64_t *out)
{
    uint64_t i = 0;
    for (; i < n && (reinterpret_cast<uintptr_t>(out + i) & 31u) != 0; i++) out[i] = in[i];
    for (; i + 4 <= n; i += 4)
        _mm256_stream_si256(reinterpret_cast<__m256i *>(out + i), _mm256_cvtepu32_epi64(_mm_loadu_si128(reinterpret_cast<const __m128i *>(in + i))));
    for (; i < n; i++) out[i] = in[i];
    _mm_sfence();
}
#endif
inline void widen_u32(const uint32_t *in, uint64_t n, uint64_t *out)
{
#ifdef GDX_WIRE_AVX2
    if (wire_have_avx2()) {
        widen_u32_avx2(in, n, out);
        return;
    }
#endif
    for (uint64_t i = 0; i < n; i++) out[i] = in[i];
}

// memcpy into a staging buffer the CPU does not read again (the link's DMA does): stores past the caches, so that no line of the
// destination is read in order to be overwritten -- a third of the memory traffic of the plain copy of a piece too small for the
// C library to do the same on its own
#ifdef GDX_WIRE_AVX2
__attribute__((target("avx2"))) inline void stream_copy_avx2(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    uint64_t i = 0;
    const uint64_t head = (32u - (reinterpret_cast<uintptr_t>(dst) & 31u)) & 31u;
    if (head != 0 && head <= n) {
        std::memcpy(dst, src, head);
        i = head;
    }
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    if (i < n) std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}
#endif
inline void stream_copy(uint8_t *dst, const uint8_t *src, uint64_t n)
{
#ifdef GDX_WIRE_AVX2
    if (n >= 4096 && wire_have_avx2()) {
        stream_copy_avx2(dst, src, n);
        return;
    }
#endif
    std::memcpy(dst, src, n);
}

inline void wire_expand_fence()
{
#ifdef GDX_WIRE_AVX2
    _mm_sfence();
#endif
}

// tiles [tile_lo, tile_hi) of a chunk of nq reads: offsets[q] = base + the hits of the reads in front of read q (offsets points
// at the chunk's first entry; the last tile writes offsets[nq] as well), hits[base + ...] = the hits.  offsets or hits may be
// null (the sizing pass of gdx_locate_many wants offsets only).  Every entry and hit slot has one writer.
template <class OffT, class HitT>
inline void wire_expand_tiles(const HostWire &w, uint64_t nq, uint64_t tile_lo, uint64_t tile_hi, OffT base, OffT *offsets, HitT *hits)
{
    if (tile_lo >= tile_hi) return;
    const uint64_t q_lo = tile_lo * kHostWireTile, q_hi = tile_hi * kHostWireTile < nq ? tile_hi * kHostWireTile : nq;
    uint64_t e = 0;  // the first exception at or behind q_lo
    {
        uint64_t hi = w.n_exc;
        while (e < hi) {
            const uint64_t mid = (e + hi) >> 1;
            if (w.exc_q[mid] < q_lo) e = mid + 1;
            else hi = mid;
        }
    }
    static const uint8_t zero_id = 0;
    const uint8_t *const ids = w.found_ids ? w.found_ids : &zero_id;
    const uint64_t id_step = w.found_ids ? 1 : 0;  // (one text: every read's id is the one zero)
    uint64_t f = w.tile_found[tile_lo];
    uint64_t eh = static_cast<uint64_t>(w.tile_off[tile_lo]) - f;  // hits in front of the tile = found reads + exception hits in front of it
    OffT o = base + w.tile_off[tile_lo];
    uint64_t next_exc = e < w.n_exc ? w.exc_q[e] : ~0ull;
    for (uint64_t q0 = q_lo; q0 < q_hi; q0 += 64) {
        if (q0 % kHostWireTile == 0 && q0 + kHostWireTile <= q_hi && next_exc >= q0 + kHostWireTile) {
            // a whole tile without exceptions (all of them where reads are unique): its found reads' hits are a straight run of
            // the positions, its offsets the running bit count -- eight entries per look-up of a bitmap byte
            const uint64_t tile = q0 / kHostWireTile;
            const uint64_t n_f = w.tile_found[tile + 1] - w.tile_found[tile];
            if (hits != nullptr) expand_positions(w.found_pos + f, w.found_ids ? w.found_ids + f : nullptr, n_f, hits + o);
            if (offsets != nullptr) expand_offsets(w.bitmap + q0 / 8, kHostWireTile / 8, o, offsets + q0);
            f += n_f;
            o += static_cast<OffT>(n_f);
            q0 += kHostWireTile - 64;
            continue;
        }
        uint64_t bits;
        std::memcpy(&bits, w.bitmap + q0 / 8, 8);  // (the bitmap is padded to whole tiles: 256 bytes each)
        const uint64_t n = q_hi - q0 < 64 ? q_hi - q0 : 64;
        if (next_exc >= q0 + n && hits != nullptr && offsets != nullptr) {  // no exception among these reads
            // Nine reads in ten are found, which ones is random: a branch on the bit is mispredicted every tenth read.  So every
            // read stores a hit -- into its slot when it has one, into a scratch slot otherwise -- and the slot and position
            // cursors advance by the bit.
            HitT scratch;
            OffT *const offs = offsets + q0;
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t bit = (bits >> i) & 1ull;
                HitT h;
                h.text_id = ids[f * id_step];
                h.position = w.found_pos[f];
                HitT *dst = bit ? hits + o : &scratch;
                *dst = h;
                offs[i] = o;
                o += static_cast<OffT>(bit);
                f += bit;
            }
            continue;
        }
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t q = q0 + i;
            if (offsets != nullptr) offsets[q] = o;
            if ((bits >> i) & 1ull) {
                if (hits != nullptr) {
                    hits[o].text_id = ids[f * id_step];
                    hits[o].position = w.found_pos[f];
                }
                o++;
                f++;
            } else if (q == next_exc) {
                const uint32_t cnt = w.exc_cnt[e];
                if (hits != nullptr)
                    for (uint32_t j = 0; j < cnt; j++) {
                        hits[o + j].text_id = w.exc_hits[eh + j].text_id;
                        hits[o + j].position = w.exc_hits[eh + j].position;
                    }
                o += cnt;
                eh += cnt;
                e++;
                next_exc = e < w.n_exc ? w.exc_q[e] : ~0ull;
            }
        }
    }
    if (q_hi == nq && offsets != nullptr) offsets[nq] = o;
    wire_expand_fence();
}

}  // namespace gdx
