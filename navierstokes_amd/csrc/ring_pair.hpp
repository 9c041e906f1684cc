// ring_pair.hpp — one compile-time switch shared by the ring planner (host, ring_plan.hpp) and the ring kernels (spmv_ring.hpp,
// spmk_ring.hpp): which nonzeros of a row block a thread owns.
#pragma once

namespace mi355 {

// Configuration 4 loads its value stream 16 bytes per lane (round 3): thread t of a block owns the nonzero PAIRS t, t + T, ... —
// nonzeros 2(t + iT), 2(t + iT) + 1 — instead of the single nonzeros t + iT, so that one wave-wide load covers 1 KiB of
// contiguous values where the 8-byte form covers 512 B (an L1-bypassing 8-byte stream runs at 0.54-0.70 of the 16-byte rate on
// this part).  Which nonzero a thread owns is only visible in the ORDER of the 16-bit column stream (build_ring_slots) and in
// where the thread stages its operands; the row chains, and so every bit of y, are unchanged.  -DMI355_RING_PAIR=0 builds the
// 8-byte form (A/B of two library builds).
#ifndef MI355_RING_PAIR
#define MI355_RING_PAIR 1
#endif
constexpr bool kRingPair4 = MI355_RING_PAIR != 0;
constexpr bool ring_pairs(int threads) { return kRingPair4 && threads == 256; }
// the ring family stages {coef, x} pairs and reads a row chain's terms with ds_read_b128 (spmv_kernels.hpp: ring_stage_cx)
#ifndef MI355_RING_MERGED
#define MI355_RING_MERGED 1
#endif
constexpr bool kRingMergedStage = MI355_RING_MERGED != 0;
// (Measured on the stream and tile kernels too — the thread owning nonzero pairs, 16 + 8 bytes of values and columns per load: the
// 5 M-row mesh 184-186 -> 181-183 us through the stream kernel and no difference through the tile kernel; the 1 M-row mesh, which
// lives in the Infinity Cache, 28.75 -> 29.5 us — a wave's gather then spans 128 nonzeros instead of 64.  Not kept there.)

} // namespace mi355
