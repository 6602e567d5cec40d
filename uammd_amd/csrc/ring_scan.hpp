// Ring FIFO of candidate indices + half-precision superset scan of the force traversal (lj.hip: k_lj_ring, k_lj_ringh).
// See DESIGN.md 5.2 for the measurements behind both.
#pragma once
#include "device_common.hpp"

namespace uammd_hip {

#ifndef RING_CAP
#define RING_CAP 16
#endif
#ifndef RING_TAKE
#define RING_TAKE 4
#endif
constexpr int kRingCap = RING_CAP;   // entries per lane (power of two; at most kRingCap - 1 are ever queued)
constexpr int kRingTake = RING_TAKE;   // pairs a partial drain takes from each lane
#ifndef RING_LANES
#define RING_LANES 128
#endif
constexpr int kRingLanes = RING_LANES;                // threads per workgroup of the ring kernels
constexpr uint kRingStep = kRingLanes * 4u;           // byte stride between consecutive entries of a lane (lanes x uint)
constexpr uint kRingMask = kRingCap * kRingStep - 1;  // the ring array is aligned to its size: wrap = mask
using LdsU32 = __attribute__((address_space(3))) uint;

struct RingQ {
  uint base;  // LDS address of the ring array (multiple of its size)
  uint head;  // LDS address of the lane's oldest entry
  uint tail;  // LDS address of the lane's next free entry
  UH_D uint bytes() const { return (tail - head) & kRingMask; }  // queued entries x kRingStep
  UH_D uint wrap(uint a) const { return base | (a & kRingMask); }
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
constexpr float kHalfMargin = 0.012f;

// Scans the packed entries of [jb, je) eight candidates per step, appending the index of every candidate that passes the
// half-precision test to the lane's ring; makeRoom() is called (by the lanes still scanning) before every step and must
// leave at most kRingCap - 9 entries queued in every lane.
template <class MakeRoom>
UH_D void half_scan(RingQ &Q, const uint3 *__restrict__ PK, int jb, int je, half2_t px, half2_t py, half2_t pz, _Float16 thr,
                    MakeRoom &&makeRoom) {
  for (int j = jb; j < je; j += 8) {
    makeRoom();
    const uint3 *__restrict__ pk = PK + j;
    uint3 w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) w[m] = pk[2 * m];
    half2_t r2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const half2_t dx = __builtin_bit_cast(half2_t, w[m].x) - px;
      const half2_t dy = __builtin_bit_cast(half2_t, w[m].y) - py;
      const half2_t dz = __builtin_bit_cast(half2_t, w[m].z) - pz;
      half2_t t = dx * dx;
      t = __builtin_elementwise_fma(dy, dy, t);
      r2[m] = __builtin_elementwise_fma(dz, dz, t);
    }
    // range test per ENTRY (pair): the odd candidate past the end of a cell is +inf in the packed copy
    const int rem = je - j;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const _Float16 d = (u & 1) ? r2[u >> 1].y : r2[u >> 1].x;
      const bool hit = !(d >= thr) & ((u & ~1) < rem);  // keeps NaN, like the exact scan
      if (hit) {
        // store + advance + wrap in place (the compiler's version keeps the old address alive through a copy)
        const uint val = (uint)(j + u);
        asm volatile("ds_write_b32 %0, %1\n\tv_add_u32 %0, %2, %0\n\tv_and_or_b32 %0, %0, %3, %4"
                     : "+v"(Q.tail)
                     : "v"(val), "s"(kRingStep), "s"(kRingMask), "v"(Q.base)
                     : "memory");
      }
    }
  }
}

}  // namespace uammd_hip
