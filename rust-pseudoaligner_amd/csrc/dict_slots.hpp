// Builder and reader of the k <= 32 dictionary (layout: device_layout.hpp; the kernel's own probe is lane_steps.hpp, SEEK).
// One text for the CPU flattener (device_flatten.cpp: threads, __atomic builtins) and the GPU filler (index_fill.hip: atomicCAS /
// atomicAnd): the atomics come in as a policy type. Insertion is two passes over all keys with a barrier in between:
//   pass 1  every key tries its HOME slot only                    (so that as many keys as possible are found by the first load)
//   pass 2  keys that are not in their home slot take another slot of the bucket and leave a flag in the home slot, or go on to the
//           next bucket (overflow flag in the home slot) where the same rule applies
// Nothing is ever removed, so "the home slot names no other slot and no overflow" proves a key absent.
#pragma once
#include "lane_steps.hpp"

namespace pa {

template <class A>
PA_HD bool slot_claim(uint32_t* slot, uint64_t km, uint32_t handle, uint32_t off) {
    if (!A::cas(slot + 2, NO_HANDLE, handle)) return false;
    slot[0] = (uint32_t)km;
    slot[1] = (uint32_t)(km >> 32);
    A::and_(slot + 3, off | ~SLOT_OFF_MASK);   // (other threads may already be clearing flag bits of this word)
    return true;
}

template <class A>
PA_HD void dict_insert_home(uint32_t* table, uint32_t nbuckets, uint64_t km, uint32_t handle, uint32_t off) {
    uint32_t home;
    const uint32_t b = pa_bucket_home(km, nbuckets, home);
    slot_claim<A>(table + (uint64_t)b * BUCKET_WORDS + SLOT_WORDS * home, km, handle, off);
}

template <class A>
PA_HD void dict_insert_rest(uint32_t* table, uint32_t nbuckets, uint64_t km, uint32_t handle, uint32_t off) {
    uint32_t home;
    uint32_t b = pa_bucket_home(km, nbuckets, home);
    {
        const uint32_t* hs = table + (uint64_t)b * BUCKET_WORDS + SLOT_WORDS * home;   // written in pass 1 if at all
        if (hs[2] != NO_HANDLE && hs[0] == (uint32_t)km && hs[1] == (uint32_t)(km >> 32)) return;
    }
    for (bool first = true;; first = false) {   // load <= 1/2 (DICT_LOAD): a free slot exists
        uint32_t* line = table + (uint64_t)b * BUCKET_WORDS;
        uint32_t* hs = line + SLOT_WORDS * home;
        if (!first && slot_claim<A>(hs, km, handle, off)) return;   // (in a later bucket the home slot may be free)
        for (uint32_t i = 0; i < 3; ++i)
            if (slot_claim<A>(line + SLOT_WORDS * ((home + 1 + i) & 3u), km, handle, off)) {
                A::and_(hs + 3, ~(1u << (SLOT_FLAG_SHIFT + i)));
                return;
            }
        A::and_(hs + 3, ~(SLOT_FLAG_OVERFLOW << SLOT_FLAG_SHIFT));
        if (++b == nbuckets) b = 0;
    }
}

// the lookup the kernel performs, as a loop (builders' self-check, edge derivation); `probes` = buckets beyond the first
PA_HD bool dict_find64(const uint32_t* table, uint32_t nbuckets, uint64_t km, uint32_t& handle, uint32_t& off, uint32_t& probes) {
    uint32_t home;
    uint32_t b = pa_bucket_home(km, nbuckets, home);
    const uint32_t klo = (uint32_t)km, khi = (uint32_t)(km >> 32);
    for (probes = 0; probes < nbuckets; ++probes) {
        const uint32_t* line = table + (uint64_t)b * BUCKET_WORDS;
        const U4 v = *reinterpret_cast<const U4*>(line + SLOT_WORDS * home);
        if (slot_holds(v, klo, khi)) { handle = v.z; off = v.w & SLOT_OFF_MASK; return true; }
        const uint32_t fl = slot_flags(v);
        for (uint32_t c = fl & 7u; c; c &= c - 1) {
            const U4 v2 = *reinterpret_cast<const U4*>(line + SLOT_WORDS * ((home + 1 + pa_ctz32(c)) & 3u));
            if (slot_holds(v2, klo, khi)) { handle = v2.z; off = v2.w & SLOT_OFF_MASK; return true; }
        }
        if (!(fl & SLOT_FLAG_OVERFLOW)) return false;
        if (++b == nbuckets) b = 0;
    }
    return false;
}

}  // namespace pa
