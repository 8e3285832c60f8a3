// Per-read state machine of the mapping kernel: one lane works on one read at a time, one call = one memory-dependent step.
//
// This is the reference's map_read_to_nodes_with_mismatch (src/pseudoaligner.rs:64-319) re-expressed for a
// 64-wide wavefront: the three data-dependent loops of the reference (dictionary scan :91-114, left extension
// :131-203, forward extension :209-301) become the states SEEK / LEFT / FWD, each advancing by one dependent HBM
// fetch per call, so that a wave can run the same state for many reads at once (map_pool.hip schedules the states).
// The per-base comparison loops (:151-170, :236-255) are replaced by XOR + popcount on 32-base windows with a slow
// path (position of the (allowed+1)-th mismatch) that only runs when a node visit exceeds its mismatch budget; the
// nodes Vec (:219) is replaced by the running intersection of the classes seen as two 32-id windows (or, for classes
// that do not fit, by the set of DISTINCT classes), because nodes_to_eq_class (:323-356) only uses the id lists and
// intersection is idempotent/commutative.
//
// Written for instruction economy on gfx950: the read's state is 8 packed words + its id (two 16-byte LDS vectors in the
// pooled kernel), 64-bit funnel shifts are two v_alignbit_b32, the bucket index is one v_mul_hi_u32.
//
// Compiled for gfx950 by map_pool.hip and for the host by tests/emu (CPU parity tests of exactly this text).
//
// The lane state exists in two PACKINGS, and every step that reads or writes it is compiled once for each (lane_steps_body.hpp, sections
// PA_LS_LANE): pa::narrow — positions and counters of 14 / 16 bits, eight words, what the pooled kernel keeps in LDS for reads of up to 512 bases
// (and what a 16 383-base read fits) — and pa::wide — 28-bit positions, 32-bit counters, twelve words, for reads that stay in HBM (GREAD kernels:
// anything longer than 512 bases, up to PA_MAX_READ_LEN; the reference has no limit: its own test maps whole transcripts, src/build_index.rs:309).
// Code in namespace pa sees the narrow names (`using namespace narrow`); the wide kernels and the emulator's wide loop live in pa::wide.
#pragma once
#include "device_layout.hpp"

#define PA_LS_COMMON 1
#define PA_LS_LANE 0
namespace pa {
#include "lane_steps_body.hpp"
}  // namespace pa
#undef PA_LS_COMMON
#undef PA_LS_LANE
#define PA_LS_COMMON 0
#define PA_LS_LANE 1

#define PA_LANE_WIDE 0
namespace pa {
namespace narrow {
#include "lane_steps_body.hpp"
}  // namespace narrow
using namespace narrow;
}  // namespace pa
#undef PA_LANE_WIDE

#define PA_LANE_WIDE 1
namespace pa {
namespace wide {
#include "lane_steps_body.hpp"
}  // namespace wide
}  // namespace pa
#undef PA_LANE_WIDE
#undef PA_LS_COMMON
#undef PA_LS_LANE
