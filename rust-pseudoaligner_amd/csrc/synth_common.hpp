// The read simulator as one function of (seed, read index), compiled for the host (synth.cpp) and for gfx950
// (kernels.hip) from the same text so that both produce identical reads.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "pa_common.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PA_SYN_HD __host__ __device__
#else
#define PA_SYN_HD
#endif

namespace pa {
namespace synth {

PA_SYN_HD static inline uint64_t mix(uint64_t x) {   // murmur3 fmix64
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

PA_SYN_HD static inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

PA_SYN_HD static inline uint64_t win32(const uint64_t* w, uint64_t pos) {
    const uint64_t i = pos >> 5;
    const uint32_t s = (uint32_t)(pos & 31) * 2;
    return s ? (w[i] >> s) | (w[i + 1] << (64 - s)) : w[i];
}

// cum[t] = number of valid read starts in transcripts < t (a transcript of length len has len-read_len+1 of them)
static inline void build_cum(const uint64_t* tx_start, uint32_t num_tx, uint32_t read_len, std::vector<uint64_t>& cum) {
    cum.assign((size_t)num_tx + 1, 0);
    // PA_SIM_TX_LIMIT=n (experiments only): reads are drawn from the first n transcripts, i.e. the index stays what it
    // is but the part of it the reads touch fits the caches
    uint32_t limit = num_tx;
    if (const char* v = knob_str("PA_SIM_TX_LIMIT")) { const long x = atol(v); if (x > 0 && (unsigned long)x < num_tx) limit = (uint32_t)x; }
    for (uint32_t t = 0; t < num_tx; ++t) {
        const uint64_t len = tx_start[t + 1] - tx_start[t];
        cum[t + 1] = cum[t] + (t < limit && len >= read_len ? len - read_len + 1 : 0);
    }
}

// words[0 .. ceil(read_len/32)) receive the packed read; bits past read_len are zero.
PA_SYN_HD static inline void simulate_read(const uint64_t* packed, const uint64_t* tx_start, const uint64_t* cum, uint32_t num_tx,
                                       uint64_t total, uint32_t read_len, uint64_t seed, uint32_t sub_rate_ppm,
                                       uint64_t read_index, uint64_t* words) {
    const uint64_t key = mix(seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull) ^ (read_index * 0xd1342543de82ef95ull);
    const uint64_t x = mulhi64(mix(key), total);   // uniform over all (transcript, start) pairs
    uint32_t lo = 0, hi = num_tx;                   // largest t with cum[t] <= x
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (cum[mid] <= x) lo = mid; else hi = mid;
    }
    const uint64_t start = tx_start[lo] + (x - cum[lo]);
    const uint32_t nw = (read_len + 31) / 32;
    for (uint32_t w = 0; w < nw; ++w) {
        uint64_t v = win32(packed, start + 32ull * w);
        const uint32_t rem = read_len - 32 * w;
        if (rem < 32) v &= (1ull << (2 * rem)) - 1;
        words[w] = v;
    }
    if (sub_rate_ppm) {
        const uint64_t ekey = mix(key ^ 0xa0761d6478bd642full);
        for (uint32_t j = 0; j < read_len; ++j) {
            const uint64_t r = mix(ekey + j);
            if ((uint32_t)r % 1000000u < sub_rate_ppm) {   // substitute with one of the three other bases
                const uint32_t sh = (j & 31) * 2;
                const uint32_t b = (uint32_t)(words[j >> 5] >> sh) & 3u;
                const uint32_t nb = (b + 1 + (uint32_t)((r >> 32) % 3u)) & 3u;
                words[j >> 5] = (words[j >> 5] & ~(3ull << sh)) | ((uint64_t)nb << sh);
            }
        }
    }
}

}  // namespace synth
}  // namespace pa
