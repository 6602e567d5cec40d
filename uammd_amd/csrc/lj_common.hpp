// Lennard-Jones pair arithmetic shared by the traversal kernels (cell list: lj.hip, Verlet list: verletlist.hip).
//   LJFunctor::force / energy               Interactor/Potential/Potential.cuh:37-65
//   BasicParameterHandler::Iterator          Interactor/Potential/ParameterHandler.cuh:41-60
//   Radial<LJ>::Transverser::compute / set   Interactor/Potential/RadialPotential.cuh:107-127
#pragma once
#include "device_common.hpp"

namespace uammd_hip {

struct LJParams { float cutOff2, sigma2, epsilonDivSigma2, shift; };

UH_D float lj_force(float r2, const LJParams &p) {  // |f|/r
  if (r2 >= p.cutOff2) return 0.0f;
  const float invr2 = p.sigma2 / r2;
  const float invr6 = invr2 * invr2 * invr2;
  return p.epsilonDivSigma2 * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
}
UH_D float lj_energy(float r2, const LJParams &p) {  // half the pair energy
  if (r2 >= p.cutOff2) return 0.0f;
  const float invr2 = p.sigma2 / r2;
  const float invr6 = invr2 * invr2 * invr2;
  const float E = fmaf(p.epsilonDivSigma2 * p.sigma2 * 4.0f * invr6, (invr6 - 1.0f), -p.shift);
  return 0.5f * E;
}
UH_D LJParams lj_lookup(const LJParams *tbl, int ntypes, int ti, int tj) {
  if (ti > tj) { const int t = ti; ti = tj; tj = t; }
  int typeIndex = ti + ntypes * tj;
  if (ti >= ntypes || tj >= ntypes) typeIndex = 0;
  return tbl[typeIndex];
}

struct Acc { float fx = 0.f, fy = 0.f, fz = 0.f, e = 0.f, v = 0.f; };

// One pair, branch free: returns |f|/r (0 outside the cut-off or at r = 0) and the pair energy.  A
// masked pair contributes fma(0, r12, acc) == acc, bit-identical to skipping it.
template <bool PBC, bool WE>
UH_D void lj_eval(const BoxT<float> &box, const LJParams &p, const float4 &ri, const float4 &rj, real3f &r12,
                  float &fm, float &e) {
  r12 = real3f{rj.x - ri.x, rj.y - ri.y, rj.z - ri.z};
  if (PBC) r12 = box.apply_pbc(r12);
  const float r2 = dot3(r12, r12);
  const bool in = (r2 != 0.0f) & !(r2 >= p.cutOff2);
  const float invr2 = p.sigma2 / r2;
  const float invr6 = invr2 * invr2 * invr2;
  const float f = p.epsilonDivSigma2 * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
  fm = in ? f : 0.0f;
  if (WE) {
    const float E = fmaf(p.epsilonDivSigma2 * p.sigma2 * 4.0f * invr6, (invr6 - 1.0f), -p.shift);
    e = in ? 0.5f * E : 0.0f;
  } else
    e = 0.0f;
}

// sigma2 / r2 exactly as the compiler's IEEE division computes it (v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup:
// SIISelLowering LowerFDIV32) when neither operand needs scaling and no special case applies — then v_div_scale returns its
// operand, v_div_fmas is a plain fma and v_div_fixup returns its first operand.  Preconditions (checked by the caller): both
// operands finite, positive and in [2^-40, 2^40].  Bit-identical to a / b there (tests compare the words over 1e6 particles).
UH_D float div_in_range(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float e = fmaf(-b, r, 1.0f);
  const float r1 = fmaf(e, r, r);
  const float q = a * r1;
  const float e2 = fmaf(-b, q, a);
  const float q1 = fmaf(e2, r1, q);
  const float e3 = fmaf(-b, q1, a);
  return fmaf(e3, r1, q1);
}
constexpr float kDivLo = 9.094947017729282e-13f;  // 2^-40
constexpr float kDivHi = 1.099511627776e12f;      // 2^40

// lj_eval with the division above; r12 and r2 are computed by the caller
template <bool WE> UH_D void lj_eval_r2_fastdiv(const LJParams &p, float r2, float &fm, float &e) {
  const bool in = (r2 != 0.0f) & !(r2 >= p.cutOff2);
  const float invr2 = div_in_range(p.sigma2, r2);
  const float invr6 = invr2 * invr2 * invr2;
  const float f = p.epsilonDivSigma2 * fmaf(-48.0f, invr6, 24.0f) * invr6 * invr2;
  fm = in ? f : 0.0f;
  if (WE) {
    const float E = fmaf(p.epsilonDivSigma2 * p.sigma2 * 4.0f * invr6, (invr6 - 1.0f), -p.shift);
    e = in ? 0.5f * E : 0.0f;
  } else
    e = 0.0f;
}

template <bool WE, bool WV>
UH_D void lj_acc(Acc &a, const real3f &r12, float fm, float e) {
  if (WE) a.e += e;
  a.fx = fmaf(fm, r12.x, a.fx);
  a.fy = fmaf(fm, r12.y, a.fy);
  a.fz = fmaf(fm, r12.z, a.fz);
  if (WV) a.v += dot3(real3f{fm * r12.x, fm * r12.y, fm * r12.z}, r12);
}

template <bool PBC> UH_D float lj_dist2(const BoxT<float> &box, const float4 &ri, const float4 &rj) {
  real3f r12{rj.x - ri.x, rj.y - ri.y, rj.z - ri.z};
  if (PBC) r12 = box.apply_pbc(r12);
  return dot3(r12, r12);
}

UH_D float lj_max_cutoff2(const LJParams *tbl, int ntypes) {
  float m = 0.0f;
  for (int t = 0; t < ntypes * ntypes; ++t) m = fmaxf(m, tbl[t].cutOff2);
  return m;
}

// POD view of a built cell list handed to the traversal kernels (CellListBase::CellListData, CellListBase.cuh:145-160, plus the
// tables the counting-sort build leaves behind)
struct ListView {
  const uint *cellStart;
  const int *cellEnd;
  const float4 *sortPos;
  const int *groupIndex;
  const uint *sortHash;
  const uint *keyStart;
  const unsigned char *cellOutside;  // per linear cell: some particle stored outside the primary box (nullable)
  const uint2 *cellRange;            // per linear cell {first, last | outside << 31}, entry ncells = {0, 0} (nullable, with cellOutside)
  const uint3 *packHalf;  // half-precision pairs of candidates (celllist.hip k_pack_half), null when not available
  float packScale;        // 1 / largest cell edge
  uint validCell;
  int N;
  int numOwned;  // particles whose input index is >= numOwned only act as neighbours (domain-decomposition ghosts)
  // tile kernels only: they need cut-off <= every multi-cell edge (27-cell reach of the 4-cell x halo, prefilter margin).  The host
  // routes other tables to the exact kernels from a cached copy of the table's largest cut-off; a launch that arrives here with a
  // larger one anyway (the caller rewrote its table in place) raises the list's host-mapped error flag instead of computing.
  float maxCut2Allowed;
  int *errFlag;
  // measurement hook (uammd_lj_tile_stats, null unless enabled): [0] bricks that took the dense-brick fallback, [1] bricks launched
  uint *tileStats;
};

struct Outputs {
  float4 *force;
  float *energy;
  float *virial;
  const int *globalIndex;
  // fused MD step (uammd_verletnvt_gj_lj_step), tile kernels only: vel != null -> the store is GronbechJensen's second half step
  // (GronbechJensen.cu:58-61) on the spot, v += dt / (2 m) f, and force = f (the first half step left the force array at zero)
  float *vel = nullptr;
  const float *mass = nullptr;
  float defaultMass = 0.f, dt = 0.f;
  float invDefaultMass = 0.f;  // 1 / defaultMass by the host's IEEE division (the same bits as the device's): not once per owner in the kernel
  int is2D = 0;
};

UH_D void write_out(const Outputs &o, int ori, const Acc &a) {
  if (o.force) {
    float4 f = o.force[ori];
    f.x += a.fx; f.y += a.fy; f.z += a.fz; f.w += 0.0f;
    o.force[ori] = f;
  }
  if (o.energy) o.energy[ori] += a.e;
  if (o.virial) o.virial[ori] += a.v;
}

}  // namespace uammd_hip
