// Lennard-Jones traversal of the cell list, MI355X fast path: cell-pair tiles with the distance test on the matrix pipe.
//
// Replaces (same pairs, same per-pair arithmetic, another summation order -> rounding-level differences, SURVEY 8d tolerance):
//   transverseWithNeighbourContainer            Interactor/NeighbourList/common.cuh:10-34
//   CellList_ns::NeighbourContainer (27 cells)  Interactor/NeighbourList/CellList/NeighbourContainer.cuh:95-130
//   Radial<LJ>::Transverser::compute / set      Interactor/Potential/RadialPotential.cuh:107-127
//
// Why another kernel.  The thread-per-particle walks (lj.hip) sit on the VALU issue limit: a lane tests ~340 candidates at
// ~10 instructions each to find its ~52 neighbours, and every candidate costs the CU's one texture addresser a 64-lane load.
// Here one WAVE owns the particles of two x-adjacent cells (x0 even: one contiguous range of the Morton-sorted array, ~25
// particles at liquid density) and
//   1. the candidates — the cells around the pair, a few dozen contiguous ranges of the sorted array — are staged into LDS by
//      LDS-DMA loads (global_load_lds: no registers, nothing waits inside the loop), in the reference's visiting order;
//   2. the squared distances minus rc^2 between the wave's 32 owners and the 64 candidates of a "word" come from two
//      v_mfma_f32_32x32x16_f16 sharing ONE candidate operand (half-precision coordinates relative to the tile centre, |.|^2 carried as
//      hi + lo halves: the error is the coordinate rounding alone, covered by a margin — tile_margin): the matrix pipe is otherwise
//      idle and issues beside the VALU.  Each lane then holds 32 values of ITS owner (owner = lane & 31, the two half-waves take the
//      even and the odd candidates) whose SIGN BIT says "inside the cut-off + margin";
//   3. one v_alignbit_b32 per value shifts the sign bits into a per-lane 32-bit hit word, stored in LDS;
//   4. the drain walks each lane's hit words (find-first-bit, two pairs per iteration): the ~11 % of the tile's pairs that are hits
//      are re-evaluated EXACTLY as the reference does (r12 = rj - ri, minimum image where the tile touches a box face,
//      r2 >= rc2 -> 0, r2 == 0 -> 0, the same polynomial; the division is rcp + one Newton step) from the full-precision positions
//      in LDS.  The prefilter is a superset, so no pair is lost and none is added.
// Owners of the two half-waves are summed at the end; forces go to force[groupIndex[i]] as Transverser::set does.
//
// Two launch shapes share the scan / drain code:
//   k_lj_tile4  (AUTO) a 256-thread workgroup owns a 2 x 2 x 2 Morton brick of cells = four x-pairs; its 4 x 4 x 4-cell halo is
//               staged ONCE for the four waves (805 instead of 4 x 453 candidates at liquid density, and a quarter of the LDS per
//               wave: 5 workgroups = 20 waves per CU).  Measured on the single-wave shape: staging was 70 us of a 250 us launch
//               at C3 and the waves' phases (load latency, matrix chain, drain) did not overlap at ~3 waves per SIMD.
//   k_lj_tile   one wave per x-pair, 4 x 3 x 3-cell halo, candidates in chunks of 512: any density; also the in-kernel fallback of
//               a brick whose halo does not fit.
#include "celllist.hpp"
#include "lj_common.hpp"
#include "gj_step.hpp"
#include <hip/hip_ext.h>
#include <vector>
#include <algorithm>

namespace uammd_hip {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f4t __attribute__((ext_vector_type(4)));
typedef _Float16 h8t __attribute__((ext_vector_type(8)));
typedef _Float16 h2t __attribute__((ext_vector_type(2)));
typedef uint u4t __attribute__((ext_vector_type(4)));
using LdsF4 = __attribute__((address_space(3))) f4t;
using LdsU = __attribute__((address_space(3))) uint;
using GlobV = __attribute__((address_space(1))) void;
using LdsV = __attribute__((address_space(3))) void;

constexpr int kMaxW = 12;                  // hit words (of 64 candidates) one wave can hold per pass
constexpr int kSoloCap = 512;              // candidate slots of the single-wave kernel (8 words per chunk)
constexpr int kBrickCap = 1024;            // candidate slots of the four-wave kernel (a C3 brick stages 805 +- 30; with the tables: 31.2 KB, five workgroups per CU)
constexpr int kFallbackRegion = (kBrickCap + 64) / 4;  // slots per wave of the dense-brick fallback: 256 of work + guard
static_assert(kFallbackRegion >= 256 + 2 && kFallbackRegion - 256 <= 16, "fallback guard slots");
// per wave: hit words [kMaxW + 1][64] | LDS address of slot h of every word, for either half-wave h [2][kMaxW + 2]
constexpr int kTabWords = (kMaxW + 1) * 64 + 2 * (kMaxW + 2);

// count of leading zeros; 0xFFFFFFFF for 0 (v_ffbh_u32)
UH_D int __builtin_clz_or_neg1(uint v) { return v ? __builtin_clz(v) : -1; }

UH_D uint rdlane(uint v, int l) { return (uint)__builtin_amdgcn_readlane((int)v, l); }

UH_D void wrap_cell(int &c, int n, bool periodic, bool &ok, bool &wr) {
  if (n == 1) {
    ok = ok && c == 0;
  } else if (c < 0) {
    if (periodic) { c += n; wr = true; } else ok = false;
  } else if (c >= n) {
    if (periodic) { c -= n; wr = true; } else ok = false;
  }
}

// (a direction that is not periodic has mInvL = 0 AND L = 0 here — tile_scale, tile_eval's box copy —: floor(0.5) = 0 and fma(0, 0, d)
// = d, no select)
UH_D float min_image(float d, float L, float mInvL) { return fmaf(floorf(fmaf(d, mInvL, 0.5f)), L, d); }

// the box with L = 0 along the directions that are not periodic (what tile_eval's minimum image takes)
UH_D BoxT<float> pbc_box(BoxT<float> b) {
  if (!b.px()) b.boxSize.x = 0.0f;
  if (!b.py()) b.boxSize.y = 0.0f;
  if (!b.pz()) b.boxSize.z = 0.0f;
  return b;
}

// one pair in the drain: the reference's arithmetic up to the division, which is rcp + one Newton step (<= 1 ulp) here.
// UNIT: the (single) pair type has sigma^2 = 1 and epsilon / sigma^2 = 1 exactly — reduced units, what the reference's benchmark and
// test programs run — and the two products by 1.0f are left out: the same bits (x * 1.0f == x), two instructions per pair fewer.  The
// kernel is instantiated on it (a wave-uniform branch between two copies of the drain loop was measured 7 % SLOWER in round 2: code size).
template <bool PBC, bool WE, bool UNIT = false>
UH_D void tile_eval(const BoxT<float> &box, const LJParams &p, const float4 &ri, const f4t &rj, real3f &r12, float &fm, float &e) {
  r12 = real3f{rj.x - ri.x, rj.y - ri.y, rj.z - ri.z};
  if (PBC) {
    // Box::apply_pbc (utils/Box.cuh:51-58) without its three selects: `box` is the caller's copy with L = 0 where not periodic, so
    // the image count floor(0.5) = 0 adds 0 there.  Product and sum stay two roundings as in the reference (positions stored several
    // boxes away make counts whose product with L is not exact: an fma differs there, tests/test_gpu_lj_tile.py 'outside').
    r12.x = r12.x + floorf(fmaf(r12.x, box.minusInvBoxSize.x, 0.5f)) * box.boxSize.x;
    r12.y = r12.y + floorf(fmaf(r12.y, box.minusInvBoxSize.y, 0.5f)) * box.boxSize.y;
    r12.z = r12.z + floorf(fmaf(r12.z, box.minusInvBoxSize.z, 0.5f)) * box.boxSize.z;
  }
  const float r2 = dot3(r12, r12);
  const bool in = (r2 != 0.0f) & !(r2 >= p.cutOff2);
  float r = __builtin_amdgcn_rcpf(r2);
#ifndef UAMMD_TILE_NO_NEWTON   // (variant, tools/variants_tile.sh nonewton: the bare v_rcp_f32, 1 ulp — measured)
  r = fmaf(fmaf(-r2, r, 1.0f), r, r);
#endif
  const float invr2 = UNIT ? r : p.sigma2 * r;
  const float invr6 = invr2 * invr2 * invr2;
  const float f = UNIT ? fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2 : p.epsilonDivSigma2 * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
  fm = in ? f : 0.0f;
  if (WE) {
    const float E = fmaf(p.epsilonDivSigma2 * p.sigma2 * 4.0f * invr6, (invr6 - 1.0f), -p.shift);
    e = in ? 0.5f * E : 0.0f;
  } else
    e = 0.0f;
}

// Candidate <-> matrix row.  The 64 candidates of a word (64 consecutive LDS slots in the reference's visiting order) are fed to the
// word's two matrix steps so that the two lanes of an owner take the even and the odd candidates (with blocks of 32 per half the busiest
// lane of a model liquid holds 41 instead of 34 hits of a mean 26) and bit j FROM THE TOP of a lane's hit word is slot 2 j + h: the
// drain's address is one shift-add.  A lane of half-wave h holds, of step sp (0: the lower half-wave's candidates, 1: the upper's) and
// accumulator a, the value of hardware row (a & 3) + 8 (a >> 2) + 4 h; which BIT that value becomes is the sign collector's business
// (tile_bits32: value index v = 2 a + sp lands on bit tile_bit_of(v)), so feeding lane (row i, step sp) reads slot
// 2 (31 - tile_bit_of(2 a + sp)) + h with a = (i & 3) + 4 (i >> 3), h = (i >> 2) & 1.
#ifndef UAMMD_TILE_ALIGNBIT
// fp6 collector: value v sits at bits 6 v .. 6 v + 5 of the 192-bit result, its sign at 6 v + 5; the six dwords are merged without
// moving a bit except the upper three by one place (see tile_bits32): v < 16 -> (6 v + 5) mod 32 (the odd places), v >= 16 -> one below
UH_HD uint tile_bit_of(uint v) { return ((6u * (v & 15u) + 5u) & 31u) - (v >> 4); }
#else
// v_alignbit chains (round 2): step 0's accumulators fill bits 31..16 and step 1's bits 15..0, each in accumulator order
UH_HD uint tile_bit_of(uint v) { return 31u - (16u * (v & 1u) + (v >> 1)); }
#endif
UH_D uint row_slot(int i, int sp) {
  const uint a = (uint)(i & 3) + 4u * (uint)(i >> 3), h = (uint)((i >> 2) & 1);
  return 2u * (31u - tile_bit_of(2u * a + (uint)sp)) + h;
}

// |cand - owner|^2 - (rc^2 + margin) for 32 candidates (rows) x 32 owners (columns) from ONE v_mfma_f32_32x32x16_f16 (32 cycles of the
// matrix pipe; the exact-f32 form, three chained v_mfma_f32_32x32x2_f32 = 192 cycles per step, made the waves of a SIMD queue for the
// pipe, and a queued wave cannot issue its vector work either).  Coordinates are relative to the tile centre in units of the largest
// cell edge (|.| <= 2.5) and rounded to half precision (toward zero: v_cvt_pkrtz); the squares are formed from the ROUNDED values and
// carried as hi + lo halves, so the matrix returns |a^ - b^|^2 - rc2m to f32 accuracy and the only error is the coordinate rounding,
// covered by the margin (tile_margin).  K slots (candidate side A | owner side B):
//   k0 b^x | -2 a^x   k1 b^y | -2 a^y   k2 b^z | -2 a^z   k3 0 | 0   k4 |b^|^2 hi | 1   k5 |b^|^2 lo | 1   k6 1 | c hi   k7 1 | c lo
// with c = |a^|^2 - rc2m.  K 8..15 belong to the upper half-wave and hold the SAME eight entries for another candidate: the owner side
// is zero in one half-wave or the other (B0 / B1), which selects the candidates a product sees.  The SIGN BIT of a result is the hit flag.
UH_D uint pk_rtz(float a, float b) { return __builtin_bit_cast(uint, __builtin_amdgcn_cvt_pkrtz(a, b)); }
UH_D float sq3_h(uint pxy, uint pz0) {  // x^2 + y^2 + z^2 of packed halves, in f32
  // (through the builtin, not inline asm: a dot instruction's result needs wait states before another kind of instruction reads it, and
  // the compiler's hazard recognizer does not look inside asm — tried, wrong distances)
  const h2t hxy = __builtin_bit_cast(h2t, pxy), hz0 = __builtin_bit_cast(h2t, pz0);
  return __builtin_amdgcn_fdot2(hz0, hz0, __builtin_amdgcn_fdot2(hxy, hxy, 0.0f, false), false);
}
UH_D uint split_h(float v) {  // (hi, lo) halves with hi + lo = v to ~2^-21
  const float hi = (float)__builtin_bit_cast(h2t, pk_rtz(v, 0.0f)).x;
  return pk_rtz(v, v - hi);
}

struct TileFrame {  // tile centre and scale of the half-precision coordinates; the scaled box for tiles that touch a face
  float s, nox, noy, noz;     // b' = fma(c, s, no), no = -origin * s
  float Lx, Ly, Lz, mx, my, mz;  // box size * s and -1 / (box size * s) (0 when not periodic)
};

template <bool PBC> UH_D void tile_centre(const TileFrame &fr, float x, float y, float z, float &bx, float &by, float &bz) {
  bx = fmaf(x, fr.s, fr.nox);
  by = fmaf(y, fr.s, fr.noy);
  bz = fmaf(z, fr.s, fr.noz);
  if (PBC) {
    bx = min_image(bx, fr.Lx, fr.mx);
    by = min_image(by, fr.Ly, fr.my);
    bz = min_image(bz, fr.Lz, fr.mz);
  }
}

// candidate side of the matrix product for the slot at candAddr
template <bool PBC> UH_D h8t tile_operand(uint candAddr, const TileFrame &fr, uint ones) {
  const f4t c = *(const LdsF4 *)(uintptr_t)candAddr;
  float bx, by, bz;
  tile_centre<PBC>(fr, c.x, c.y, c.z, bx, by, bz);
  const uint pxy = pk_rtz(bx, by), pz0 = pk_rtz(bz, 0.0f);
  const u4t a = {pxy, pz0, split_h(sq3_h(pxy, pz0)), ones};
  return __builtin_bit_cast(h8t, a);
}
UH_D v16f tile_product(const h8t &A, const h8t &B) {
  const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);
}

// 2 x 16 values -> the lane's 32-bit hit word (bit tile_bit_of(2 a + sp) = sign of step sp's accumulator a).
#ifndef UAMMD_TILE_ALIGNBIT
// ONE v_cvt_scalef32_2xpk16_fp6_f32 packs the 32 values into 32 six-bit floats (a[n] -> value 2 n, b[n] -> value 2 n + 1; the sign is
// kept for zeros, denormals, huge and infinite inputs: tools/cvt_probe.hip), whose sign bits sit 6 apart: dwords 0, 1, 2 hold them at
// bits {5, 11, 17, 23, 29}, {3, 9, 15, 21, 27}, {1, 7, 13, 19, 25, 31} — together the 16 odd places — and dwords 3, 4, 5 the same for
// values 16..31.  Five bit selects and one shift merge them.  Measured (tools/cvt_probe.hip, 4 waves per SIMD): the conversion 28.3 ns
// per wave against 32 v_alignbit_b32 at 1.88 = 60 ns; 28.3 + 6 x 1.9 = 40 ns for the word.
typedef uint v6u __attribute__((ext_vector_type(6)));
// (a & mask) | (b & ~mask) as ONE v_bfi_b32, the mask in a scalar register: written in C the compiler, knowing the masks, turns the five
// selects into six ANDs, two three-way ORs and an AND-OR
UH_D uint bit_select(uint mask, uint a, uint b) {
  uint d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
  return d;
}
UH_D uint tile_bits32(const v16f &dA, const v16f &dB) {
  const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(dA, dB, 1.0f);
  constexpr uint M0 = 0x20820820u, M01 = M0 | 0x08208208u;  // bits {5, 11, 17, 23, 29} and {3, 9, 15, 21, 27}
  const uint lo = bit_select(M01, bit_select(M0, r[0], r[1]), r[2]);
  const uint hi = bit_select(M01, bit_select(M0, r[3], r[4]), r[5]);
  return bit_select(0xAAAAAAAAu, lo, hi >> 1);
}
#else
// one v_alignbit_b32 per value: m = (m << 1) | sign(d) (two independent chains of 8 per step: a dependent VALU instruction cannot issue
// back to back)
UH_D uint tile_bits16(const v16f &d) {
  uint m0 = 0, m1 = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(d[a]), 31);
    m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(d[a + 8]), 31);
  }
  return (m0 << 8) | m1;
}
UH_D uint tile_bits32(const v16f &dA, const v16f &dB) { return (tile_bits16(dA) << 16) | tile_bits16(dB); }
#endif

// diagnostic build (tools/tile_budget.py): section marks in the assembly, no effect on a normal build
#ifdef UAMMD_TILE_MARKERS
#define TILE_MARK(name, k) asm volatile("; MARK " name " %0" ::"n"(k))
#else
#define TILE_MARK(name, k) do {} while (0)
#endif
// The wave's 32 owners against nW words of staged candidates.  Word w = 64 consecutive LDS slots starting at byte candBase +
// wbase[w], of which the first wcnt[w] are this wave's candidates (the rest is staged data of other rows, or padding: their bits
// are cleared).  wbase / wcnt arrive in registers — lane w holds word w's, entries >= nW are zero or any staged slot — and the scan
// reads them with v_readlane; tab = this wave's table in LDS: hit words [kMaxW + 1][64] | slot address per half-wave [2][kMaxW + 2].
// NT: 0 = several types (a parameter lookup per pair), 1 = one type, 2 = one type in reduced units (tile_eval's UNIT)
template <bool PBC, int NT, bool WE, bool WV>
UH_D void tile_words(Acc &acc, uint nW, uint candBase, uint tab, uint wbaseV, uint wcntV, int lane, const TileFrame &fr, const h8t &B0, const h8t &B1, const float4 &pi, const BoxT<float> &box, const LJParams &p1, const LJParams *__restrict__ tbl, int ntypes) {
  const int hi = lane >> 5;
  // the candidate this lane feeds to the matrix: slot row_slot(lane & 31) + 32 hi of the word at + wbase[w]
  const uint rowAddr = candBase + 16u * row_slot(lane & 31, hi);
  const uint myMask = tab + 4u * (uint)lane;                  // hit word w of this lane at + 256 w
  // ---- scan: ONE operand per lane and word — the lower half-wave holds the 32 candidates of the word's first matrix step in K slots
  // 0..7, the upper half-wave those of the second step in K slots 8..15 — and two products: B0 carries the owners in K 0..7 and zeros
  // in K 8..15, B1 the reverse, so each product sees one half-wave's candidates.  (Operands are finite whatever a padding slot holds:
  // the conversions round toward zero and saturate.)  The next word's operand is built and its products are on the matrix pipe while
  // this word's values are turned into bits.
  // (the two 1.0 halves of the candidate operand, hidden from constant folding: as a literal the compiler assembles the operand from a
  // constant vector, five register moves per matrix step)
  uint ones = 0x3c003c00u;
  asm volatile("" : "+v"(ones));
  TILE_MARK("scan_setup", PBC);
  h8t A = tile_operand<PBC>(rowAddr + rdlane(wbaseV, 0), fr, ones);
  v16f dA = tile_product(A, B0), dB = tile_product(A, B1);
  for (uint w = 0; w < nW; ++w) {
    TILE_MARK("scan_body", PBC);
    const uint cnt = rdlane(wcntV, (int)w);
    // (unconditional: behind the last word this is entry nW of the table = slot 0, a wasted step — a conditional one makes the
    // compiler keep two register sets for the products and copy them every word)
    A = tile_operand<PBC>(rowAddr + rdlane(wbaseV, (int)w + 1), fr, ones);
    uint m = tile_bits32(dA, dB);
    dA = tile_product(A, B0);
    dB = tile_product(A, B1);
    if (cnt < 64u) {  // the word runs past the wave's candidates (the last word of a run): slots 2 j + h >= cnt are not its own
      asm volatile("");  // (keeps the wave-uniform branch: if-converted, these instructions run for every word)
      const uint mine = (cnt + 1u - (uint)hi) >> 1;  // <= 32
      m &= (uint)(0xFFFFFFFF00000000ull >> mine);
    }
    *(LdsU *)(uintptr_t)(myMask + 256u * w) = m;
  }
  TILE_MARK("drain_setup", PBC);
  __builtin_amdgcn_s_setprio(0);  // (see k_lj_tile4: the drain yields to waves that are loading or scanning)
  // ---- drain: every lane walks its own hit words (bit j from the top of word w = slot 2 j + h of the word) ----
  // cw / cb = the word being consumed and the LDS address of its slot h, nw / nb = the next one; word nW of every lane is zero and a
  // lane never moves past word nW - 1 (the look-ahead reads word nW: a row of the table that exists, never consumed).  A lane
  // without a set bit in cw takes a dead slot: ffbh(0) = -1 addresses slot -2 of the word — staged data of another row or the two
  // guard slots in front of the buffer, finite either way — with weight 0.
  *(LdsU *)(uintptr_t)(myMask + 256u * nW) = 0u;
  const uint wabsTab = tab + 4u * (uint)((kMaxW + 1) * 64 + (kMaxW + 2) * hi);  // candBase + 16 hi + wbase[w]
  uint cw = *(const LdsU *)(uintptr_t)myMask;
  uint nw = *(const LdsU *)(uintptr_t)(myMask + 256u);
  uint cb = *(const LdsU *)(uintptr_t)wabsTab;
  uint nb = *(const LdsU *)(uintptr_t)(wabsTab + 4u);
  uint wi1 = 1;  // 1 + index of cw, <= nW
  unsigned long long more;  // lanes with wi1 < nW
  asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(more) : "v"(wi1), "s"(nW));
  // Two pairs per iteration (instruction-level parallelism for the rcp / polynomial chains, two LDS reads in flight): advance to the
  // next word when this one is used up, then take the TWO highest set bits; a word with an odd number of hits leaves one dead slot:
  // 19 iterations per tile against 35 with one pair per iteration on a model liquid.
  auto pop2 = [&](bool &live0, bool &live1, uint &a0, uint &a1) {
    // advance: cw == 0 and words left.  (Lane masks by hand: from `wi1 += adv` the compiler makes a select and an add, from a ballot
    // of adv a select and a compare; the mask of the compare is the carry-in of one v_addc.)  The look-ahead holds ONE word: a lane that
    // advances takes it, and the word after its new index is requested at once (a lane that stays re-reads the same one) — the request
    // has the whole iteration to arrive.  (Round 6; until then the word after the next was requested at the top of the iteration and
    // selected a few instructions later, two more selects per iteration: 0.1344 -> 0.1328 ms per launch, tools/time_lj.py with REPS=2000,
    // four interleaved runs each, +- 0.0002.)
    unsigned long long adv, co;
    asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(adv) : "v"(cw));
    adv &= more;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cw) : "v"(cw), "v"(nw), "s"(adv));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cb) : "v"(cb), "v"(nb), "s"(adv));
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(wi1), "=s"(co) : "v"(wi1), "s"(adv));
    nw = *(const LdsU *)(uintptr_t)(myMask + 256u * wi1);
    nb = *(const LdsU *)(uintptr_t)(wabsTab + 4u * wi1);
    asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(more) : "v"(wi1), "s"(nW));
    live0 = cw != 0;
    const uint k0 = (uint)__builtin_clz_or_neg1(cw);
    cw &= ~(0x80000000u >> (k0 & 31u));
    live1 = cw != 0;
    const uint k1 = (uint)__builtin_clz_or_neg1(cw);
    cw &= ~(0x80000000u >> (k1 & 31u));
    a0 = cb + (k0 << 5);
    a1 = cb + (k1 << 5);
  };
#ifdef UAMMD_TILE_POP3
  // VARIANT (tools/variants_tile.sh pop3; measured and not kept): THREE pairs per iteration — fewer loop and word-advance
  // instructions per pair, more dead slots (a word's hit count is rarely a multiple of three).
  auto pop3 = [&](bool &live0, bool &live1, bool &live2, uint &a0, uint &a1, uint &a2) {
    const uint tw = *(const LdsU *)(uintptr_t)(myMask + 256u * wi1 + 256u);
    const uint tb = *(const LdsU *)(uintptr_t)(wabsTab + 4u * wi1 + 4u);
    unsigned long long adv, co;
    asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(adv) : "v"(cw));
    adv &= more;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cw) : "v"(cw), "v"(nw), "s"(adv));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cb) : "v"(cb), "v"(nb), "s"(adv));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nw) : "v"(nw), "v"(tw), "s"(adv));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nb) : "v"(nb), "v"(tb), "s"(adv));
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(wi1), "=s"(co) : "v"(wi1), "s"(adv));
    asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(more) : "v"(wi1), "s"(nW));
    live0 = cw != 0;
    const uint k0 = (uint)__builtin_clz_or_neg1(cw);
    cw &= ~(0x80000000u >> (k0 & 31u));
    live1 = cw != 0;
    const uint k1 = (uint)__builtin_clz_or_neg1(cw);
    cw &= ~(0x80000000u >> (k1 & 31u));
    live2 = cw != 0;
    const uint k2 = (uint)__builtin_clz_or_neg1(cw);
    cw &= ~(0x80000000u >> (k2 & 31u));
    a0 = cb + (k0 << 5);
    a1 = cb + (k1 << 5);
    a2 = cb + (k2 << 5);
  };
  bool lN0, lN1, lN2;
  uint a0, a1, a2;
  pop3(lN0, lN1, lN2, a0, a1, a2);
  f4t cN0 = *(const LdsF4 *)(uintptr_t)a0, cN1 = *(const LdsF4 *)(uintptr_t)a1, cN2 = *(const LdsF4 *)(uintptr_t)a2;
  while (__any(lN0) || more != 0) {
    TILE_MARK("drain_body", PBC);
    const f4t c0 = cN0, c1 = cN1, c2 = cN2;
    const bool l0 = lN0, l1 = lN1, l2 = lN2;
    pop3(lN0, lN1, lN2, a0, a1, a2);
    cN0 = *(const LdsF4 *)(uintptr_t)a0;
    cN1 = *(const LdsF4 *)(uintptr_t)a1;
    cN2 = *(const LdsF4 *)(uintptr_t)a2;
    real3f r0, r1, r2;
    float f0, f1, f2, e0, e1, e2;
    if (NT != 0) {
      tile_eval<PBC, WE, NT == 2>(box, p1, pi, c0, r0, f0, e0);
      tile_eval<PBC, WE, NT == 2>(box, p1, pi, c1, r1, f1, e1);
      tile_eval<PBC, WE, NT == 2>(box, p1, pi, c2, r2, f2, e2);
    } else {
      tile_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c0.w), pi, c0, r0, f0, e0);
      tile_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c1.w), pi, c1, r1, f1, e1);
      tile_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c2.w), pi, c2, r2, f2, e2);
    }
    lj_acc<WE, WV>(acc, r0, l0 ? f0 : 0.0f, l0 ? e0 : 0.0f);
    lj_acc<WE, WV>(acc, r1, l1 ? f1 : 0.0f, l1 ? e1 : 0.0f);
    lj_acc<WE, WV>(acc, r2, l2 ? f2 : 0.0f, l2 ? e2 : 0.0f);
  }
#else
  bool lN0, lN1;
  uint a0, a1;
  pop2(lN0, lN1, a0, a1);
  f4t cN0 = *(const LdsF4 *)(uintptr_t)a0, cN1 = *(const LdsF4 *)(uintptr_t)a1;
  // some lane still holds a pair or has words left (empty trailing words are walked through).  (The condition as ONE scalar word and a
  // for loop: written `while (__any(lN0) || more != 0)` the compiler split the loop into two blocks around the test and carried the three
  // force sums through register copies — four v_mov per iteration and non-destructive v_fma; this form is a single block with v_fmac
  // accumulators: 61 -> 57 vector instructions per iteration, 0.1328 -> 0.1314 ms per launch, tools/time_lj.py with REPS=2000.)
  for (unsigned long long busy = __builtin_amdgcn_ballot_w64(lN0) | more; busy != 0; busy = __builtin_amdgcn_ballot_w64(lN0) | more) {
    TILE_MARK("drain_body", PBC);
    const f4t c0 = cN0, c1 = cN1;
    const bool l0 = lN0, l1 = lN1;
    pop2(lN0, lN1, a0, a1);  // the candidates of the next two pairs are on their way from LDS while these two are evaluated
    cN0 = *(const LdsF4 *)(uintptr_t)a0;
    cN1 = *(const LdsF4 *)(uintptr_t)a1;
    real3f r0, r1;
    float f0, f1, e0, e1;
    if (NT != 0) {
      tile_eval<PBC, WE, NT == 2>(box, p1, pi, c0, r0, f0, e0);
      tile_eval<PBC, WE, NT == 2>(box, p1, pi, c1, r1, f1, e1);
    } else {
      tile_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c0.w), pi, c0, r0, f0, e0);
      tile_eval<PBC, WE>(box, lj_lookup(tbl, ntypes, (int)pi.w, (int)c1.w), pi, c1, r1, f1, e1);
    }
    lj_acc<WE, WV>(acc, r0, l0 ? f0 : 0.0f, l0 ? e0 : 0.0f);
    lj_acc<WE, WV>(acc, r1, l1 ? f1 : 0.0f, l1 ? e1 : 0.0f);
  }
#endif
  TILE_MARK("drain_end", PBC);
}

// owners' side of the matrix products for the 32 owners [o0, o0 + 32) of a wave, and the lane's owner position
struct OwnerSide { float4 pi; h8t B0, B1; bool valid; };
// (ownLds != 0: the LDS byte address of the owners' first staged slot — the brick kernel's halo holds its owners, and a global load here
// would be one more memory round trip between the staging barrier and the first matrix step)
UH_D OwnerSide tile_owner(const float4 *__restrict__ P, uint ownFirst, int nOwn, int o0, int lane, bool pbc, const TileFrame &fr, float ox,
                          float oy, float oz, float rc2ms, uint ownLds = 0u) {
  OwnerSide o;
  const int my = o0 + (lane & 31), hi = lane >> 5;
  o.valid = my < nOwn;
  if (ownLds) {
    const f4t q = *(const LdsF4 *)(uintptr_t)(ownLds + 16u * (uint)(o.valid ? my : 0));
    o.pi = o.valid ? make_float4(q.x, q.y, q.z, q.w) : make_float4(ox, oy, oz, 0.0f);
  } else
    o.pi = o.valid ? P[ownFirst + (uint)my] : make_float4(ox, oy, oz, 0.0f);
  float ax, ay, az;
  if (pbc) tile_centre<true>(fr, o.pi.x, o.pi.y, o.pi.z, ax, ay, az);
  else tile_centre<false>(fr, o.pi.x, o.pi.y, o.pi.z, ax, ay, az);
  const uint pxy = pk_rtz(ax, ay), pz0 = pk_rtz(az, 0.0f);
  // a row without an owner never hits (|b^|^2 + 6e4 > 0); the upper half-wave holds k = 8..15: zero
  const float c = o.valid ? sq3_h(pxy, pz0) - rc2ms : 6.0e4f;
  const uint m2xy = pk_rtz(-2.0f * (float)__builtin_bit_cast(h2t, pxy).x, -2.0f * (float)__builtin_bit_cast(h2t, pxy).y);
  const uint m2z0 = pk_rtz(-2.0f * (float)__builtin_bit_cast(h2t, pz0).x, 0.0f);
  // B0: the owners in the lower half-wave (K 0..7), zeros in the upper (K 8..15); B1 the reverse
  const u4t b = {m2xy, m2z0, 0x3c003c00u, split_h(c)}, zero = {0u, 0u, 0u, 0u};
  o.B0 = __builtin_bit_cast(h8t, hi ? zero : b);
  o.B1 = __builtin_bit_cast(h8t, hi ? b : zero);
  return o;
}

// the part of the frame that depends on the grid and the box only: k_lj_tile4 takes it from the host as a kernel argument (four IEEE
// divisions = ~48 vector instructions per wave otherwise; host and device round a division the same way, so the values are the same)
inline __host__ __device__ TileFrame tile_scale(const GridT<float> &grid, const BoxT<float> &box) {
  TileFrame fr;
  const float ey = grid.cellDim.y > 1 ? grid.cellSize.y : 0.f, ez = grid.cellDim.z > 1 ? grid.cellSize.z : 0.f;
  const float eyz = ey > ez ? ey : ez;
  const float e = grid.cellSize.x > eyz ? grid.cellSize.x : eyz;
  fr.s = 1.0f / e;
  fr.nox = fr.noy = fr.noz = 0.0f;
  fr.Lx = box.boxSize.x * fr.s; fr.Ly = box.boxSize.y * fr.s; fr.Lz = box.boxSize.z * fr.s;
  fr.mx = box.px() ? -1.0f / fr.Lx : 0.0f;
  fr.my = box.py() ? -1.0f / fr.Ly : 0.0f;
  fr.mz = box.pz() ? -1.0f / fr.Lz : 0.0f;
  if (!box.px()) fr.Lx = 0.0f;  // (min_image without a select: see there)
  if (!box.py()) fr.Ly = 0.0f;
  if (!box.pz()) fr.Lz = 0.0f;
  return fr;
}
UH_D TileFrame tile_centred(TileFrame fr, float ox, float oy, float oz) {
  fr.nox = -ox * fr.s; fr.noy = -oy * fr.s; fr.noz = -oz * fr.s;
  return fr;
}
UH_D TileFrame tile_frame(const GridT<float> &grid, const BoxT<float> &box, float ox, float oy, float oz) {
  return tile_centred(tile_scale(grid, box), ox, oy, oz);
}

// (giPre >= 0: the owner's input index, loaded by the caller BEFORE the scan and drain — at the end of a wave's life the index load and
// the velocity loads that depend on it are two exposed memory round trips that keep the workgroup's slot)
template <bool WE, bool WV>
UH_D void tile_finish(Acc &acc, const ListView &cl, const Outputs &out, uint ownFirst, int o0, int lane, bool valid, int giPre = -1) {
  acc.fx += __shfl_xor(acc.fx, 32);
  acc.fy += __shfl_xor(acc.fy, 32);
  acc.fz += __shfl_xor(acc.fz, 32);
  if (WE) acc.e += __shfl_xor(acc.e, 32);
  if (WV) acc.v += __shfl_xor(acc.v, 32);
  if ((lane >> 5) == 0 && valid) {
    const int gi = giPre >= 0 ? giPre : cl.groupIndex[ownFirst + (uint)(o0 + (lane & 31))];
    if (out.vel) {  // the fused step: half kick with the force that is still in registers (no group; ghosts of a slab's list are skipped)
      if (gi >= cl.numOwned) return;
      const float invMass = out.defaultMass > 0 ? out.invDefaultMass : 1.0f / out.mass[gi];
      float3 v = make_float3(out.vel[3 * (size_t)gi], out.vel[3 * (size_t)gi + 1], out.vel[3 * (size_t)gi + 2]);
      const float fx = 0.0f + acc.fx, fy = 0.0f + acc.fy, fz = 0.0f + acc.fz;  // what `force += f` leaves in a zeroed array
      gj_step2(v, fx, fy, fz, invMass, out.dt, out.is2D);
      out.vel[3 * (size_t)gi] = v.x; out.vel[3 * (size_t)gi + 1] = v.y; out.vel[3 * (size_t)gi + 2] = v.z;
      out.force[gi] = make_float4(fx, fy, fz, 0.0f);
    } else if (gi < cl.numOwned) write_out(out, out.globalIndex ? out.globalIndex[gi] : gi, acc);
  }
}

// ---- one wave, one x-pair of cells, its 4 x 3 x 3-cell halo in chunks of `cap` candidates (cap = 64 * words, <= 64 kMaxW) ----------
// candBase: cap + 64 slots of LDS private to the wave; tab: its table.  Uses only wave-level synchronisation.
template <bool NT1, bool WE, bool WV>
UH_D void tile_solo(const ListView &cl, const GridT<float> &grid, const BoxT<float> &box, const LJParams *__restrict__ tbl, int ntypes,
                    const Outputs &out, float margin, int x0, int ty, int tz, uint candBase, uint cap, uint tab, int lane) {
  const int cx = grid.cellDim.x, cy = grid.cellDim.y, cz = grid.cellDim.z;
  // the 4 x 3 x 3 candidate cells: lane = 4 * row + col, row = (dy + 1) + 3 (dz + 1), col -> x0 - 1 .. x0 + 2
  uint first = 0, len = 0;
  bool special = false;  // the range is a periodic image or holds particles stored outside the primary box
  if (lane < 36) {
    const int col = lane & 3, row = lane >> 2;
    int x = x0 - 1 + col, y = ty + (row % 3) - 1, z = tz + (row / 3) - 1;
    bool ok = true, wr = false;
    wrap_cell(x, cx, box.px(), ok, wr);
    wrap_cell(y, cy, box.py(), ok, wr);
    wrap_cell(z, cz, box.pz(), ok, wr);
    if (col == 3 && x0 + 1 >= cx) ok = false;  // x0 + 2 only neighbours the (missing) second owner cell
    if (ok) {
      const uint2 cr = cl.cellRange[x + cx * (y + cy * z)];
      first = cr.x;
      len = (cr.y & 0x7fffffffu) - cr.x;
      special = len != 0 && (wr || (cr.y >> 31) != 0);
    }
  }
  const bool pbcTile = __any(special);
  const bool own2 = x0 + 1 < cx;  // lane 17 = cell x0, lane 18 = cell x0 + 1 of the centre row (a periodic image when !own2)
  const uint len17 = rdlane(len, 17), len18 = own2 ? rdlane(len, 18) : 0u;
  const int nOwn = (int)(len17 + len18);
  if (nOwn == 0) return;
  const uint ownFirst = len17 ? rdlane(first, 17) : rdlane(first, 18);
  {  // x0 and x0 + 1 are neighbours in Morton order: one contiguous range whenever both hold particles
    const uint nf = __shfl_down(first, 1), nl = __shfl_down(len, 1);
    const bool mrg = lane < 36 && (lane & 3) == 1 && len != 0 && nl != 0 && nf == first + len;
    if (mrg) len += nl;
    if (__shfl_up((int)mrg, 1)) len = 0;
  }
  uint C = len;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) C += __shfl_xor(C, d);
  C = rdlane(C, 0);
  const unsigned long long ranges = __ballot(len != 0);
  const float ox = fmaf((float)(x0 + 1), grid.cellSize.x, -0.5f * box.boxSize.x);
  const float oy = fmaf((float)ty + 0.5f, grid.cellSize.y, -0.5f * box.boxSize.y);
  const float oz = fmaf((float)tz + 0.5f, grid.cellSize.z, -0.5f * box.boxSize.z);
  const float4 *__restrict__ P = cl.sortPos;
  const LJParams p1 = tbl[0];
  const TileFrame fr = tile_frame(grid, box, ox, oy, oz);
  const float maxCut2 = NT1 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  if (maxCut2 > cl.maxCut2Allowed) {  // wave-uniform; see ListView::maxCut2Allowed
    if (lane == 0 && cl.errFlag) cl.errFlag[0] = 2;
    return;
  }
  const float rc2ms = maxCut2 * fr.s * fr.s + margin;  // scaled units
  const f4t zero4 = {0.0f, 0.0f, 0.0f, 0.0f};  // padding behind the last candidate: finite; the word counts clear its bits
  LdsF4 *cand = (LdsF4 *)(uintptr_t)candBase;
  const uint wabsTab = tab + 4u * (uint)((kMaxW + 1) * 64);
  const uint wbaseV = lane < kMaxW + 2 ? 1024u * (uint)lane : 0u;  // words = consecutive blocks of 64 slots
  if (lane < kMaxW + 2) {
    *(LdsU *)(uintptr_t)(wabsTab + 4u * (uint)lane) = candBase + wbaseV;
    *(LdsU *)(uintptr_t)(wabsTab + 4u * (uint)(kMaxW + 2 + lane)) = candBase + 16u + wbaseV;
  }
  for (int o0 = 0; o0 < nOwn; o0 += 32) {
    const OwnerSide ow = tile_owner(P, ownFirst, nOwn, o0, lane, pbcTile, fr, ox, oy, oz, rc2ms);
    Acc acc;
    for (uint c0 = 0; c0 < C; c0 += cap) {
      const uint nC = min(C - c0, cap);
      if (o0 == 0 || C > cap) {
        __builtin_amdgcn_wave_barrier();
        // global -> LDS without registers (LDS DMA: lane i lands at the wave-uniform destination + 16 i): nothing waits in the loop
        unsigned long long m = ranges;
        uint sO = 0;  // flat index of the range's first candidate (the reference's visiting order)
        while (m) {
          const int r = __builtin_ctzll(m);
          m &= m - 1;
          const uint sF = rdlane(first, r), sL = rdlane(len, r);
          const uint lo = max(sO, c0), hiE = min(sO + sL, c0 + cap);
          for (uint k = lo; k < hiE; k += 64u) {
            const float4 *src = P + (sF + (k - sO));  // wave-uniform
            if ((uint)lane < hiE - k) __builtin_amdgcn_global_load_lds((const GlobV *)(src + lane), (LdsV *)(cand + (k - c0)), 16, 0, 0);
          }
          sO += sL;
        }
        for (uint k = nC + (uint)lane; k < ((nC + 63u) & ~63u); k += 64u) cand[k] = zero4;
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): the DMA and table writes have landed
        __builtin_amdgcn_wave_barrier();
      }
      const uint nW = (nC + 63u) >> 6;
      const uint wcntV = nC > 64u * (uint)lane ? min(nC - 64u * (uint)lane, 64u) : 0u;
      if (pbcTile)
        tile_words<true, NT1 ? 1 : 0, WE, WV>(acc, nW, candBase, tab, wbaseV, wcntV, lane, fr, ow.B0, ow.B1, ow.pi, pbc_box(box), p1, tbl, ntypes);
      else
        tile_words<false, NT1 ? 1 : 0, WE, WV>(acc, nW, candBase, tab, wbaseV, wcntV, lane, fr, ow.B0, ow.B1, ow.pi, box, p1, tbl, ntypes);
    }
    tile_finish<WE, WV>(acc, cl, out, ownFirst, o0, lane, ow.valid);
  }
}

template <bool NT1, bool WE, bool WV>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_lj_tile(ListView cl, GridT<float> grid, BoxT<float> box, const LJParams *__restrict__ tbl, int ntypes, Outputs out, float margin,
          int npx, uint nTiles) {
  __shared__ f4t candg[2 + kSoloCap + 64];  // two guard slots in front (dead slots of the drain read slot -2)
  __shared__ uint tab[kTabWords];
  if (threadIdx.x < 2) candg[threadIdx.x] = f4t{0.0f, 0.0f, 0.0f, 0.0f};
  LdsF4 *cand = (LdsF4 *)candg + 2;
  const uint t = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (t >= nTiles) return;
  const int cy = grid.cellDim.y;
  const int px = (int)(t % (uint)npx), ty = (int)((t / (uint)npx) % (uint)cy), tz = (int)(t / (uint)(npx * cy));
  tile_solo<NT1, WE, WV>(cl, grid, box, tbl, ntypes, out, margin, 2 * px, ty, tz, (uint)(uintptr_t)cand, (uint)kSoloCap,
                         (uint)(uintptr_t)(LdsU *)tab, (int)threadIdx.x);
}

// ---- four waves, one 2 x 2 x 2 brick of cells (four x-pairs), its 4 x 4 x 4-cell halo staged once ------------------------------------
// Halo rows: r = ry + 4 rz (ry, rz = 0..3 <-> y0 - 1 + ry, z0 - 1 + rz), three ranges per row (x0 - 1 | x0, x0 + 1 | x0 + 2), flat
// order (rz, ry, x) = the reference's visiting order restricted to any wave's 3 x 3 rows.  Wave k = wy + 2 wz owns the pair in row
// (1 + wy, 1 + wz); its candidates are the rows ry in [wy, wy + 2], rz in [wz, wz + 2]: three contiguous runs of the flat array.
// (five waves per SIMD: 96 VGPRs and 36 bytes of scratch; measured 0.2116 / 0.2003 / 0.2085 ms at 4 / 5 / 6 — six needs 80 VGPRs,
// 100 bytes of scratch and an LDS diet, kMaxW 10 / kBrickCap 896 / the range table inside the hit-word rows, that sends more bricks
// to the fallback)
template <int NT, bool WE, bool WV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8)))
k_lj_tile4(ListView cl, GridT<float> grid, BoxT<float> box, const LJParams *__restrict__ tbl, int ntypes, Outputs out, float margin,
           int nbx, int nby, uint nBricks, TileFrame scale) {
  // (one struct: the order matters — the drain's look-ahead of the last wave reads up to 32 bytes past its hit words, into rangeTab)
  __shared__ struct {
    f4t candg[2 + kBrickCap + 64];  // two guard slots in front (dead slots of the drain read slot -2)
    uint tabs[4 * kTabWords];
    uint4 rangeTab[48];             // {first, len, flat offset, special}
    uint total[2];                  // {number of candidates, some wave needs more than kMaxW words}
  } sh;
  f4t *candg = sh.candg;
  uint *tabs = sh.tabs;
  uint4 *rangeTab = sh.rangeTab;
  uint *total = sh.total;
  if (threadIdx.x < 2) candg[threadIdx.x] = f4t{0.0f, 0.0f, 0.0f, 0.0f};
  LdsF4 *cand = (LdsF4 *)candg + 2;
  // (dense-brick fallback: wave k works in slots [R k, R k + 256), R = kFallbackRegion; the slots in front of the next wave's region are
  // its guard)
  if (threadIdx.x < 64 && (threadIdx.x & 15u) < (uint)(kFallbackRegion - 256))
    cand[(uint)kFallbackRegion * (threadIdx.x >> 4) + 256u + (threadIdx.x & 15u)] = f4t{0.0f, 0.0f, 0.0f, 0.0f};
  // Wave priorities: the prologue (table reads, the staging loads) at 3, the scan at 2, the drain at 0.  A SIMD holds five waves of five
  // workgroups in different phases; by age a new workgroup's few prologue instructions queue behind the older waves' drain loops and its
  // memory latency starts late.  Letting the phases that WAIT go first took the launch from 0.1815 to 0.1626 ms (tools/time_lj.py; the
  // prologue alone: 0.170; scan at 1 or 2: the same; the final store raised as well: 0.166).
  __builtin_amdgcn_s_setprio(3);
  // (every XCD gets one contiguous eighth of the bricks, by COUNT.  Cutting by work instead — on an odd grid the last brick layer along y
  // and z holds one cell layer instead of two, and at C3 the XCD that owns the light z layer runs out of work early — was measured 5 %
  // slower with light bricks weighted 0.6 of a full one: a light brick holds its workgroup slot for as long as its two working waves run,
  // ~0.8 of a full brick; and a look-up table of brick indices in any order costs one more dependent memory round trip at the head of
  // every workgroup: 4 % — tools/variants_tile.sh, DESIGN 5.2 "Round 4")
  const uint t = xcd_contiguous_block(blockIdx.x, gridDim.x);
  if (t >= nBricks) return;
  if (cl.tileStats && threadIdx.x == 0) atomicAdd(&cl.tileStats[1], 1u);
#ifdef UAMMD_TILE_TIMELINE  // diagnostic build: where a workgroup's lifetime goes (100 MHz clock ticks summed over the launch's waves)
  const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#define TL_STAMP(k) do { if (cl.tileStats && (threadIdx.x & 63) == 0) atomicAdd(&cl.tileStats[k], (uint)(__builtin_amdgcn_s_memrealtime() - tl0)); } while (0)
#else
#define TL_STAMP(k) do {} while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TILE_MARK("k4_ranges", 0);
  const int wy = wave & 1, wz = wave >> 1;
  const int cx = grid.cellDim.x, cy = grid.cellDim.y, cz = grid.cellDim.z;
  const int bx = (int)(t % (uint)nbx), by = (int)((t / (uint)nbx) % (uint)nby), bz = (int)(t / (uint)(nbx * nby));
  const int x0 = 2 * bx, y0 = 2 * by, z0 = 2 * bz;
  const uint candBase = (uint)(uintptr_t)cand;
  const uint tab = (uint)(uintptr_t)(LdsU *)tabs + 4u * (uint)(wave * kTabWords);
  const float4 *__restrict__ P = cl.sortPos;
  // ---- wave 0: the 48 ranges and their flat offsets ----
  if (wave == 0) {
    uint first = 0, len = 0, special = 0;
    if (lane < 48) {
      const int row = lane / 3, c = lane - 3 * row, ry = row & 3, rz = row >> 2;
      int y = y0 - 1 + ry, z = z0 - 1 + rz;
      bool ok = true, wr = false;  // a collapsed direction has one cell: rows other than its own do not exist
      wrap_cell(y, cy, box.py(), ok, wr);
      wrap_cell(z, cz, box.pz(), ok, wr);
      const bool pairEndsGrid = x0 + 1 >= cx;  // odd grid: the last pair is the single cell x0
      // c = 0: x0 - 1;  c = 1: x0 (+ x0 + 1, contiguous in Morton order);  c = 2: x0 + 2, or the periodic image of x0 + 1 when the
      // grid ends at x0 (that image neighbours x0; x0 + 2 would only neighbour the missing second cell)
      int xa = c == 0 ? x0 - 1 : (c == 1 ? x0 : (pairEndsGrid ? x0 + 1 : x0 + 2));
      wrap_cell(xa, cx, box.px(), ok, wr);
      if (ok) {
        const int rowIdx = cx * (y + cy * z);
        const uint2 cr = cl.cellRange[xa + rowIdx];
        first = cr.x;
        len = (cr.y & 0x7fffffffu) - cr.x;
        uint outside = len ? (cr.y >> 31) : 0u;
        if (c == 1 && !pairEndsGrid) {
          const uint2 cb = cl.cellRange[x0 + 1 + rowIdx];
          const uint lb = (cb.y & 0x7fffffffu) - cb.x;
          if (len == 0) first = cb.x;
          len += lb;
          outside |= lb ? (cb.y >> 31) : 0u;
        }
        special = (len != 0 && (wr || outside != 0)) ? 1u : 0u;
      }
    }
    const uint incl = wave_inclusive_scan(len);
    if (lane < 48) rangeTab[lane] = make_uint4(first, len, incl - len, special);
    if (lane == 63) { total[0] = incl; total[1] = 0u; }
  }
  __syncthreads();
  TILE_MARK("k4_runs_staging", 0);
  TL_STAMP(4);  // ranges known
  const uint C = total[0];
  // every lane of every wave holds the range of its index (lanes >= 48: empty)
  uint4 rg = make_uint4(0u, 0u, 0u, 0u);
  if (lane < 48) rg = rangeTab[lane];
  // this wave's pair, runs and words
  const int ownRange = 3 * ((1 + wy) + 4 * (1 + wz)) + 1;
  const int nOwn = (y0 + wy < cy && z0 + wz < cz) ? (int)rdlane(rg.y, ownRange) : 0;
  const uint ownFirst = rdlane(rg.x, ownRange);
  const uint ownLds = candBase + 16u * rdlane(rg.z, ownRange);
  uint runStart[3], runLen[3], nW = 0;
  bool pbcWave = false;
  const unsigned long long sp = __ballot(rg.w != 0);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int ra = 3 * (wy + 4 * (wz + p)), rb = ra + 8;  // first and last range of the run: rows wy .. wy + 2 of plane wz + p
    runStart[p] = rdlane(rg.z, ra);
    runLen[p] = rdlane(rg.z, rb) + rdlane(rg.y, rb) - runStart[p];
    nW += (runLen[p] + 63u) >> 6;
    pbcWave = pbcWave || ((sp >> ra) & 0x1FFull) != 0;
  }
  if (nOwn != 0 && nW > (uint)kMaxW && lane == 0) total[1] = 1u;  // (benign race: every writer stores 1)
  // a list with ghosts (slab decomposition: particles with input index >= numOwned are neighbours only): a pair of cells that holds
  // ghosts alone — the halo planes — has nothing to compute.  The load is in flight during the staging.
  bool anyOwned = true;
  if (cl.numOwned != 0x7fffffff && nOwn <= 64) {
    const int gi = lane < nOwn ? cl.groupIndex[ownFirst + (uint)lane] : 0x7fffffff;
    anyOwned = __any(gi < cl.numOwned);
  }
  const bool fits = C <= (uint)kBrickCap;
  // ---- staging: the 48 ranges dealt to the four waves ----
  if (fits) {
    for (int r = wave; r < 48; r += 4) {
      const uint sF = rdlane(rg.x, r), sL = rdlane(rg.y, r), sO = rdlane(rg.z, r);
      for (uint k = 0; k < sL; k += 64u) {
        const float4 *src = P + (sF + k);  // wave-uniform
        if ((uint)lane < sL - k) __builtin_amdgcn_global_load_lds((const GlobV *)(src + lane), (LdsV *)(cand + (sO + k)), 16, 0, 0);
      }
    }
    if (wave == 3) {  // 64 slots behind the last candidate: the last word of the last run reads them (finite; their bits are cleared)
      const f4t far = {0.0f, 0.0f, 0.0f, 0.0f};
      cand[C + (uint)lane] = far;
    }
  }
  // this wave's words: run p contributes ceil(runLen / 64) words starting at its first slot.  Lane w keeps word w's byte offset and
  // count (the scan reads them with v_readlane) and writes the slot addresses the drain looks up
  uint wbaseV = 0, wcntV = 0;
  if (fits) {
    const uint w0 = (runLen[0] + 63u) >> 6, w1 = w0 + ((runLen[1] + 63u) >> 6);
    const uint w = (uint)lane;
    const int p = (w >= w0) + (w >= w1);
    const uint k = w - (p == 0 ? 0u : (p == 1 ? w0 : w1));
    const uint rs = p == 0 ? runStart[0] : (p == 1 ? runStart[1] : runStart[2]);
    const uint rl = p == 0 ? runLen[0] : (p == 1 ? runLen[1] : runLen[2]);
    const uint left = rl > 64u * k ? rl - 64u * k : 0u;
    wbaseV = w < nW ? 16u * (rs + 64u * k) : 0u;
    wcntV = w < nW ? min(left, 64u) : 0u;
    if (lane < kMaxW + 2) {
      const uint wabsTab = tab + 4u * (uint)((kMaxW + 1) * 64);
      *(LdsU *)(uintptr_t)(wabsTab + 4u * w) = candBase + wbaseV;
      *(LdsU *)(uintptr_t)(wabsTab + 4u * ((uint)(kMaxW + 2) + w)) = candBase + 16u + wbaseV;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  TL_STAMP(5);  // this wave's staging loads landed
  __syncthreads();
  TILE_MARK("k4_owner", 0);
  TL_STAMP(6);  // halo staged
  __builtin_amdgcn_s_setprio(2);
  if (!fits || total[1] != 0u) {
    // a dense brick: every wave runs the chunked single-pair algorithm on its quarter of the candidate buffer
    if (cl.tileStats && tid == 0) atomicAdd(&cl.tileStats[0], 1u);
    TILE_MARK("k4_fallback_dense_brick", 0);
    if (y0 + wy < cy && z0 + wz < cz)
      tile_solo<NT != 0, WE, WV>(cl, grid, box, tbl, ntypes, out, margin, x0, y0 + wy, z0 + wz,
                                 candBase + 16u * (uint)(wave * kFallbackRegion), 192u, tab, lane);
    return;
  }
  if (nOwn == 0 || !anyOwned) return;
  TILE_MARK("k4_frame_owner", 0);
  const float ox = fmaf((float)(x0 + 1), grid.cellSize.x, -0.5f * box.boxSize.x);
  const float oy = fmaf((float)(y0 + wy) + 0.5f, grid.cellSize.y, -0.5f * box.boxSize.y);
  const float oz = fmaf((float)(z0 + wz) + 0.5f, grid.cellSize.z, -0.5f * box.boxSize.z);
  const LJParams p1 = tbl[0];
  const TileFrame fr = tile_centred(scale, ox, oy, oz);
  const float maxCut2 = NT != 0 ? p1.cutOff2 : lj_max_cutoff2(tbl, ntypes);
  // (wave-uniform; see ListView::maxCut2Allowed.  The host picks the reduced-units instantiation from its cached copy of the table:
  // a table rewritten in place behind that cache is caught here, not evaluated with the wrong units)
  if (maxCut2 > cl.maxCut2Allowed || (NT == 2 && !(p1.sigma2 == 1.0f && p1.epsilonDivSigma2 == 1.0f))) {
    if (lane == 0 && cl.errFlag) cl.errFlag[0] = 2;
    return;
  }
  const float rc2ms = maxCut2 * fr.s * fr.s + margin;  // scaled units
  for (int o0 = 0; o0 < nOwn; o0 += 32) {
    const OwnerSide ow = tile_owner(P, ownFirst, nOwn, o0, lane, pbcWave, fr, ox, oy, oz, rc2ms, ownLds);
    const int giPre = ow.valid ? cl.groupIndex[ownFirst + (uint)(o0 + (lane & 31))] : 0;
    Acc acc;
    if (pbcWave)
      tile_words<true, NT, WE, WV>(acc, nW, candBase, tab, wbaseV, wcntV, lane, fr, ow.B0, ow.B1, ow.pi, pbc_box(box), p1, tbl, ntypes);
    else
      tile_words<false, NT, WE, WV>(acc, nW, candBase, tab, wbaseV, wcntV, lane, fr, ow.B0, ow.B1, ow.pi, box, p1, tbl, ntypes);
    TILE_MARK("k4_finish", 0);
    tile_finish<WE, WV>(acc, cl, out, ownFirst, o0, lane, ow.valid, giPre);
    TILE_MARK("k4_finish_end", 0);
  }
  TL_STAMP(7);  // wave done
#ifdef UAMMD_TILE_TIMELINE
  if (cl.tileStats && (threadIdx.x & 63) == 0) atomicAdd(&cl.tileStats[8], 1u);  // waves that reached the end with owners
#endif
}

// host side: can this list / box take the tile kernel, and with what margin
float lj_tile_max_cutoff2(const GridT<float> &g) {
  float e2 = 3.0e38f;
  const int n[3] = {g.cellDim.x, g.cellDim.y, g.cellDim.z};
  const float c[3] = {g.cellSize.x, g.cellSize.y, g.cellSize.z};
  for (int k = 0; k < 3; ++k)
    if (n[k] > 1 && c[k] * c[k] < e2) e2 = c[k] * c[k];
  return e2 * 1.0001f;  // (CellList::createUpdateGrid floors L / rc: the edge is >= rc up to the rounding of that division)
}

bool lj_tile_supported(const CellList *h, const BoxT<float> &box, float maxCutOff2) {
  const GridT<float> &g = h->grid;
  // every cell edge must hold the largest cut-off: the matrix prefilter's margin is derived for r <= rc <= e, and the 4-cell x halo
  // would otherwise pair owners with candidates two cells away, outside the reference's 27 cells
  if (!(maxCutOff2 <= lj_tile_max_cutoff2(g))) return false;
  if (!h->haveCellOutside || !h->cellRange.ptr) return false;
  const bool sameBox = box.boxSize.x == g.box.boxSize.x && box.boxSize.y == g.box.boxSize.y && box.boxSize.z == g.box.boxSize.z &&
                       box.px() == g.box.px() && box.py() == g.box.py() && box.pz() == g.box.pz();
  if (!sameBox) return false;
  const int n[3] = {g.cellDim.x, g.cellDim.y, g.cellDim.z};
  const bool per[3] = {g.box.px(), g.box.py(), g.box.pz()};
  for (int k = 0; k < 3; ++k) {
    // one cell along a periodic direction: the nearest image of a pair is not a function of the two centred coordinates, which is
    // what the matrix prefilter multiplies (thin periodic slabs take the thread-per-particle kernels)
    if (n[k] == 1 && per[k]) return false;
    if (n[k] == 1) continue;
    if (n[k] < 3) return false;
    if (k == 0 && per[k] && n[k] < 4) return false;  // x0 - 1 and x0 + 2 must be different cells
  }
  return true;
}

static float tile_margin(const GridT<float> &) {
  // In units of the largest cell edge e: owners lie within (1, 0.5, 0.5) and candidates within (2.5, 1.5, 1.5) of the tile centre.
  // Rounding toward zero to half precision moves a coordinate by < 2^-10 of its magnitude, so the difference vector of a pair moves
  // by |d| < 2^-10 |(3.5, 2, 2)| = 4.4e-3 and, for a pair inside the cut-off (r <= rc <= e, i.e. r' <= 1), the matrix result
  // |a^ - b^|^2 <= r'^2 + 2 r' |d| + |d|^2 < r'^2 + 8.9e-3.  The hi + lo halves of the squares lose < 2^-20 of values <= 9 and the f32
  // accumulation ~1e-6: the margin is 9.5e-3 (in e^2).  With e ~ rc that is a shell of 0.5 % of the cut-off: +1.4 % candidates reach
  // the drain, which re-tests every pair exactly as the reference does.
  return 9.5e-3f;
}

// shape 1 = one wave per x-pair (k_lj_tile), 4 = one workgroup of four waves per 2 x 2 x 2 brick (k_lj_tile4)
template <bool NT1, bool WE, bool WV>
int launch_lj_tile(CellList *h, const ListView &cl, const BoxT<float> &box, const LJParams *tbl, int ntypes, const Outputs &out,
                   int shape, hipStream_t st) {
  const GridT<float> &g = h->grid;
  const float margin = tile_margin(g);
  const int npx = (g.cellDim.x + 1) / 2;
  hipEvent_t e0 = nullptr, e1 = nullptr;  // null: a plain launch
  if (shape == 1) {
    if (h->prof.enabled) { if (int e = h->prof.next(&e0, &e1)) return e; }
    const uint nTiles = (uint)npx * (uint)g.cellDim.y * (uint)g.cellDim.z;
    hipExtLaunchKernelGGL((k_lj_tile<NT1, WE, WV>), dim3(nTiles), dim3(64), 0, st, e0, e1, 0, cl, g, box, tbl, ntypes, out, margin, npx, nTiles);
  } else {
    const int nby = (g.cellDim.y + 1) / 2, nbz = (g.cellDim.z + 1) / 2;
    const uint nBricks = (uint)npx * (uint)nby * (uint)nbz;
    if (h->prof.enabled) { if (int e = h->prof.next(&e0, &e1)) return e; }
    // (the reduced-units instantiation: one type, forces only, sigma^2 = epsilon / sigma^2 = 1 in the list's cached copy of the table;
    // the kernel checks the table it is given)
#ifdef UAMMD_TILE_NOUNIT  // (A/B timing)
    const bool unit = false;
#else
    const bool unit = NT1 && !WE && !WV && h->ljTable == (const void *)tbl && h->ljTableUnit;
#endif
    if (unit)
      hipExtLaunchKernelGGL((k_lj_tile4<2, false, false>), dim3(nBricks), dim3(256), 0, st, e0, e1, 0, cl, g, box, tbl, ntypes, out, margin, npx, nby, nBricks, tile_scale(g, box));
    else
      hipExtLaunchKernelGGL((k_lj_tile4<NT1 ? 1 : 0, WE, WV>), dim3(nBricks), dim3(256), 0, st, e0, e1, 0, cl, g, box, tbl, ntypes, out, margin, npx, nby, nBricks, tile_scale(g, box));
  }
  return 0;
}

template int launch_lj_tile<true, false, false>(CellList *, const ListView &, const BoxT<float> &, const LJParams *, int, const Outputs &, int, hipStream_t);
template int launch_lj_tile<true, true, true>(CellList *, const ListView &, const BoxT<float> &, const LJParams *, int, const Outputs &, int, hipStream_t);
template int launch_lj_tile<false, false, false>(CellList *, const ListView &, const BoxT<float> &, const LJParams *, int, const Outputs &, int, hipStream_t);
template int launch_lj_tile<false, true, true>(CellList *, const ListView &, const BoxT<float> &, const LJParams *, int, const Outputs &, int, hipStream_t);

}  // namespace uammd_hip
