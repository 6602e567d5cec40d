// Immersed-boundary stencil machinery shared by the generic IBM entry points and the FCM solver.
//
// Reference behaviour (misc/IBM.cu:10-65, misc/IBM.cuh:65-97, utils/Grid.cuh:124-131):
//   celli   = Grid::getCell(pos)
//   P       = support/2, minus 1 when the left-most node is farther than support*h/2 (even supports)
//   w_a[i]  = phi_a(distanceToCellCenter(pos, celli + i - P).a)      for i in [0, support.a)
//   node (ii,jj,kk) = pbc_cell(celli + (ii,jj,kk) - P), skipped when outside a non periodic grid
// One 64-lane wave handles one particle: lanes 0..3*support-1 evaluate the 1-D weights (one exp each),
// the rest of the wave reads them back with ds_bpermute while walking the support^3 nodes.
#pragma once
#include "device_common.hpp"
#include "../../include/uammd_hip.h"

namespace uammd_hip {

enum { kKernelGaussian = 0, kKernelPeskin3 = 1, kKernelPeskin4 = 2, kKernelConstant = 3, kKernelBarnettMagland = 4,
       kKernelSixPoint = 5, kKernelGauss2D = 6, kKernelGauss2DDriftX = 7, kKernelGauss2DDriftY = 8 };
constexpr int kMaxSupport = 42;  // the 3*support 1-D weights live in two registers of one wave (<= 128; the reference's largest test: 41)

struct IBMKernelDev {
  int kind;
  int3 support;
  float prefactor, tau, rmax;
  float invhx, invhy, invhz;
};

UH_D float phi_gaussian(const IBMKernelDev &k, float r) {
  // FCM_ns::Kernels::Gaussian::phi cuts only r >= rmax (BDHI/FCM/FCM_kernels.cuh:55-57)
  return (r >= k.rmax) ? 0.0f : k.prefactor * expf(k.tau * r * r);
}
UH_D float phi_peskin3(float invh, float rr) {  // misc/IBM_kernels.cuh:120-136
  const float r = fabsf(rr) * invh;
  if (r < 0.5f) return invh * (float)(1 / 3.0) * (1.0f + sqrtf(fmaf(-3.0f * r, r, 1.0f)));
  if (r < 1.5f) {
    const float omr = 1.0f - r;
    return invh * (float)(1 / 6.0) * (fmaf(-3.0f, r, 5.0f) - sqrtf(fmaf(-3.0f * omr, omr, 1.0f)));
  }
  return 0.0f;
}
UH_D float phi_peskin4(float invh, float rr) {  // misc/IBM_kernels.cuh:145-158
  const float r = fabsf(rr) * invh;
  if (r < 1.0f) return invh * 0.125f * (fmaf(-2.0f, r, 3.0f) + sqrtf(fmaf(4.0f * r, (1.0f - r), 1.0f)));
  if (r < 2.0f) return invh * 0.125f * (fmaf(-2.0f, r, 5.0f) - sqrtf(fmaf(-(4.0f * r), r, fmaf(12.0f, r, -7.0f))));
  return 0.0f;
}
// "Exponential of a semicircle" window, misc/IBM_kernels.cuh:82-112: prefactor = 1/norm, tau = beta, rmax = alpha.
// invhx holds the length unit a of the FCM wrapper, phi(r) = bm.phi(r/a)/a (BDHI/FCM/FCM_kernels.cuh:151-154);
// a = 1 gives the plain window exactly.
UH_D float phi_barnett_magland(const IBMKernelDev &k, float r) {
  const float z = (r / k.invhx) / k.rmax;
  const float dz2 = 1.0f - z * z;
  const float w = (dz2 < 0.0f) ? 0.0f : expf(k.tau * (sqrtf(dz2) - 1.0f));
  return w * k.prefactor / k.invhx;
}
// Six-point C3 window of Bao, Kaye & Peskin (GaussianFlexible::sixPoint, misc/IBM_kernels.cuh:162-236).
// The operation order follows the reference's single-precision expression, term by term.
UH_D float phi_sixpoint(float invh, float rr) {
  const float r = fabsf(rr) * invh;
  if (r >= 3.0f) return 0.0f;
  const float K = 0.714075092976608f;
  const float R = r - ceilf(r) + 1.0f;
  const float R2 = R * R;
  const float R3 = R2 * R;
  const float b = (float)(9.0 / 4.0) - 1.5f * (K + R2) + ((float)(22. / 3) - 7.0f * K) * R - (float)(7. / 3.) * R3;
  const float g = 0.25f * (0.5f * (161.0f / 36.0f - 59.0f / 6.0f * K + 5.0f * K * K) * R2 +
                           1.0f / 3.0f * (-109.0f / 24.0f + 5.0f * K) * R2 * R2 + 5.0f / 18.0f * R3 * R3);
  const float discr = b * b - 4.0f * 28.0f * g;
  const float pre = 1.0f / (2.0f * 28.0f) * (-b + sqrtf(discr));
  float v;
  if (r <= 0.0f) {
    const float t = r + 1.0f;
    v = 2.0f * pre + 0.25f + (float)(1. / 6) * (4.0f - 3.0f * K) * t - (float)(1. / 6) * t * t * t;
  } else if (r <= 1.0f) {
    v = 2.0f * pre + (float)(5. / 8) - 0.25f * (K + r * r);
  } else if (r <= 2.0f) {
    const float t = r + -1.0f;
    v = -3.0f * pre + 0.25f - (float)(1. / 6.) * (4.0f - 3.0f * K) * t + (float)(1. / 6) * t * t * t;
  } else {
    const float t = r + -2.0f;
    v = pre - (float)(1. / 16) + (float)(1. / 8) * (K + t * t) - (float)(1. / 12) * (3.0f * K - 1.0f) * t -
        (float)(1. / 12) * t * t * t;
  }
  return v * invh;
}
UH_D float phi_axis(const IBMKernelDev &k, int axis, float r) {
  switch (k.kind) {
    case kKernelGaussian: return phi_gaussian(k, r);
    case kKernelPeskin3: return phi_peskin3(axis == 0 ? k.invhx : (axis == 1 ? k.invhy : k.invhz), r);
    case kKernelPeskin4: return phi_peskin4(axis == 0 ? k.invhx : (axis == 1 ? k.invhy : k.invhz), r);
    case kKernelBarnettMagland: return phi_barnett_magland(k, r);
    case kKernelSixPoint: return phi_sixpoint(axis == 0 ? k.invhx : (axis == 1 ? k.invhy : k.invhz), r);
    // BDHI2D_ns::Gaussian / GaussianThermalDrift<dir> (Integrator/Hydro/BDHI_quasi2D.cuh:112-153); phiZ = 1 (make_stencil)
    case kKernelGauss2D: return k.prefactor * expf(k.tau * r * r);
    case kKernelGauss2DDriftX: return k.prefactor * expf(k.tau * r * r) * (axis == 0 ? r : 1.0f);
    case kKernelGauss2DDriftY: return k.prefactor * expf(k.tau * r * r) * (axis == 1 ? r : 1.0f);
    default: return 1.0f;
  }
}

// Per-particle stencil, identical in every lane of the wave except `w` (lane l < 3*support holds one weight).
struct Stencil {
  int3 celli, P, support;
  float w;   // weight t = lane:  [0,sx) x weights, [sx,sx+sy) y weights, [sx+sy, sx+sy+sz) z weights
  float w2;  // weight t = lane + 64 (supports above 21 nodes per axis)
};
// weight t of the stencil (t < 128); the second register is only consulted when the stencil has more than 64 weights (wave uniform)
UH_D float stencil_weight(const Stencil &s, int t) {
  float a = __shfl(s.w, t & 63, 64);
  if (s.support.x + s.support.y + s.support.z > 64) {
    const float b = __shfl(s.w2, t & 63, 64);
    a = t >= 64 ? b : a;
  }
  return a;
}

UH_D int3 compute_support_shift(const GridT<float> &g, real3f pos, int3 celli, int3 support) {  // IBM.cu:10-31
  int3 P = make_int3(support.x / 2, support.y / 2, support.z / 2);
  real3f d = g.distanceToCellCenter(pos, make_int3(celli.x - P.x, celli.y - P.y, celli.z - P.z));
  d.x = fabsf(d.x); d.y = fabsf(d.y); d.z = fabsf(d.z);
  if (g.cellSize.x > 0 && d.x > (float)support.x * g.cellSize.x / 2.0f) P.x -= 1;
  if (g.cellSize.y > 0 && d.y > (float)support.y * g.cellSize.y / 2.0f) P.y -= 1;
  if (g.cellSize.z > 0 && d.z > (float)support.z * g.cellSize.z / 2.0f) P.z -= 1;
  return P;
}

UH_D Stencil make_stencil(const GridT<float> &g, const IBMKernelDev &k, real3f pi, bool is2D, int lane) {
  Stencil s;
  s.celli = g.getCell(pi);
  s.support = k.support;
  s.P = compute_support_shift(g, pi, s.celli, s.support);
  if (is2D) { s.P.z = 0; s.support.z = 1; }
  const int sx = s.support.x, sy = s.support.y, sz = s.support.z;
  auto weight = [&](int t) -> float {
    float w = 0.0f;
    if (t < sx) {
      const int cx = g.pbc_x(s.celli.x + t - s.P.x);
      if (cx >= 0) w = phi_axis(k, 0, g.distanceToCellCenter(pi, make_int3(cx, s.celli.y, s.celli.z)).x);
    } else if (t < sx + sy) {
      const int cy = g.pbc_y(s.celli.y + (t - sx) - s.P.y);
      if (cy >= 0) w = phi_axis(k, 1, g.distanceToCellCenter(pi, make_int3(s.celli.x, cy, s.celli.z)).y);
    } else if (t < sx + sy + sz) {
      const int cz = g.pbc_z(s.celli.z + (t - sx - sy) - s.P.z);
      if (cz >= 0) w = phi_axis(k, 2, g.distanceToCellCenter(pi, make_int3(s.celli.x, s.celli.y, cz)).z);
      // 2D: the Peskin windows of the reference tests return phiZ = 1 (test/misc/ibm/test_ibm_regular.cu:83-85)
      if (is2D && (k.kind == kKernelPeskin3 || k.kind == kKernelPeskin4 || k.kind >= kKernelGauss2D)) w = 1.0f;
    }
    return w;
  };
  s.w = weight(lane);
  s.w2 = sx + sy + sz > 64 ? weight(lane + 64) : 0.0f;
  return s;
}

// exact i / d for 0 <= i < 2^32 / d with a host-precomputed M = ceil(2^32 / d) (d >= 2); d == 1 -> identity
struct FastDiv {
  uint M, d;
  UH_D uint div(uint i) const { return d == 1u ? i : __umulhi(i, M); }
};
inline FastDiv make_fastdiv(int d) {
  if (d <= 1) return FastDiv{0u, 1u};
  return FastDiv{(uint)((0x100000000ull + (unsigned long long)d - 1ull) / (unsigned long long)d), (uint)d};
}

IBMKernelDev to_dev(const uammd_ibm_kernel &k);

}  // namespace uammd_hip
