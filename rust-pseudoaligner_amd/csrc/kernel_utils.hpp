// Device-side helpers shared by the kernels (wave64, gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "lane_steps.hpp"

namespace pa {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave64 inclusive prefix sum with DPP (no LDS traffic): row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into the upper half (gfx9 DPP controls; row/bank masks as in LLVM's buildScan)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// hash of a sorted id list; must equal list_hash_host (device_index.hip)
__device__ __forceinline__ uint64_t list_hash_dev(const uint32_t* v, uint32_t n) {
    uint64_t h = 0x243f6a8885a308d3ull ^ n;
    for (uint32_t i = 0; i < n; ++i) h = pa_mix64(h ^ v[i]) + 0x9e3779b97f4a7c15ull;
    return h;
}

// index class whose id list equals v[0..n), or 0xFFFFFFFF (content lookup in the class-list hash table)
__device__ __forceinline__ uint32_t class_of_list(const uint32_t* v, uint32_t n, const DevIndexView& ix,
                                                  const uint32_t* class_table, uint64_t class_table_size) {
    uint64_t j = list_hash_dev(v, n) % class_table_size;
    for (;;) {
        const uint32_t cand = class_table[j];
        if (cand == 0xFFFFFFFFu) return cand;
        if (ix.class_len[cand] == n) {
            const uint32_t* ids = class_ids(ix, ix.class_ref[cand]);
            bool eq = true;
            for (uint32_t t = 0; t < n && eq; ++t) eq = ids[t] == v[t];
            if (eq) return cand;
        }
        if (++j == class_table_size) j = 0;
    }
}

}  // namespace pa
