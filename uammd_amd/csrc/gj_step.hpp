// VerletNVT::GronbechJensen_ns::integrateGPU<step> (Integrator/VerletNVT/GronbechJensen.cu:28-62) as device functions, shared by the
// stand-alone integrator kernels (integrators.hip) and by the fused step, where step 1 rides in the cell list's hash kernel
// (celllist.hip) and step 2 in the tile traversal's store (lj_tile.hip).  One definition: the fused and the plain step are bit-identical.
#pragma once
#include "device_common.hpp"
#include "saru.hpp"

namespace uammd_hip {

// what the fused kernels need besides the particle arrays
struct GJFuse {
  float *vel;             // real3[N]
  float4 *force;          // real4[N]
  const float *mass;      // nullable
  float defaultMass, dt, friction, noiseAmplitude;
  int is2D;
  uint stepNum, seed;
  int keepForce;  // the hash kernel's fused half step leaves the force array alone (the caller's traversal overwrites every entry)
  // the domain-decomposed step (uammd_celllist_update_gj1): the thermostat's stream keyed by keys[row] (the global particle id) instead of
  // the row, rows with skip[row] != 0 left alone (the halo pack integrated them before it sent them), rows >= nRows are ghosts
  const int *keys;             // nullable
  const unsigned char *skip;   // nullable
  int nRows;                   // 0: every row
};

// step 1 for particle i, noise stream id: p, v updated in place; the caller stores them (and zeroes the force)
UH_D void gj_step1(float4 &p, float3 &v, const float4 &f4, float invMass, float dt, float friction, float noiseAmplitude, int is2D,
                   uint id, uint stepNum, uint seed) {
  Saru rng(id, stepNum, seed);
  noiseAmplitude *= 1.0f / sqrtf(invMass);
  const float2 n01 = rng.gf(0.0f, noiseAmplitude);
  float nz = 0.0f;
  if (!is2D) nz = rng.gf(0.0f, noiseAmplitude).x;
  const float gdthalfinvMass = friction * dt * 0.5f;
  const float b = 1.0f / (1.0f + gdthalfinvMass);
  const float a = (1.0f - gdthalfinvMass) * b;
  const float bdt = b * dt;
  const float c = 0.5f * invMass * dt * b;
  p.x = fmaf(c, fmaf(dt, f4.x, n01.x), fmaf(bdt, v.x, p.x));
  p.y = fmaf(c, fmaf(dt, f4.y, n01.y), fmaf(bdt, v.y, p.y));
  p.z = fmaf(c, fmaf(dt, f4.z, nz), fmaf(bdt, v.z, p.z));
  const float d = dt * 0.5f * invMass * a;
  const float e = b * invMass;
  v.x = fmaf(e, n01.x, fmaf(d, f4.x, a * v.x));
  v.y = fmaf(e, n01.y, fmaf(d, f4.y, a * v.y));
  v.z = fmaf(e, nz, fmaf(d, f4.z, a * v.z));
  if (is2D) v.z = 0.0f;
}
UH_D void gj_step2(float3 &v, float fx, float fy, float fz, float invMass, float dt, int is2D) {
  const float d = dt * 0.5f * invMass;
  v.x = fmaf(d, fx, v.x);
  v.y = fmaf(d, fy, v.y);
  v.z = fmaf(d, fz, v.z);
  if (is2D) v.z = 0.0f;
}

}  // namespace uammd_hip
