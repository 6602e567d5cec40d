// Streaming integrator kernels for gfx950 (one thread per particle, 256-thread workgroups,
// coalesced float4 / 3xfloat accesses; these are HBM-bound: ~132 B per particle per MD step).
//
// Reference behaviour:
//   VerletNVT::GronbechJensen_ns::integrateGPU<step>   Integrator/VerletNVT/GronbechJensen.cu:28-62
//   VerletNVT::Basic_ns::integrateGPU<step>            Integrator/VerletNVT/Basic.cu:86-114
//   VerletNVT::Basic_ns::initialVelocities             Integrator/VerletNVT/Basic.cu:12-29
//   BD::EulerMaruyama_ns::integrateGPU                 Integrator/BrownianDynamics.cu:119-144
//   BDHI::FCM_ns::integrateEulerMaruyamaD              Integrator/BDHI/BDHI_FCM.cu:67-92
// Noise streams are keyed exactly as in the reference: Saru(thread index in group, step, seed).
#include "celllist.hpp"
#include "saru.hpp"
#include "gj_step.hpp"

namespace uammd_hip {

constexpr int kIB = 256;

// noiseKey (nullable): the Saru stream of thread id is keyed by noiseKey[id] instead of id — a domain-decomposed run passes the GLOBAL
// particle ids, so that a particle draws the same kicks whichever rank and row holds it (the reference is single GPU: key = id)
template <int STEP>
__global__ void __launch_bounds__(kIB) k_verletnvt_gj(float4 *__restrict__ pos, float *__restrict__ vel,
                                                      float4 *__restrict__ force, const float *__restrict__ mass,
                                                      float defaultMass, const int *__restrict__ index, int N,
                                                      float dt, float friction, int is2D, float noiseAmplitude,
                                                      uint stepNum, uint seed, const int *__restrict__ noiseKey = nullptr) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  const float invMass = 1.0f / (defaultMass > 0 ? defaultMass : mass[i]);
  float3 v = make_float3(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]);
  const float4 f4 = force[i];
  if (STEP == 1) {
    float4 p = pos[i];
    gj_step1(p, v, f4, invMass, dt, friction, noiseAmplitude, is2D, noiseKey ? (uint)noiseKey[id] : (uint)id, stepNum, seed);
    pos[i] = p;
    force[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else
    gj_step2(v, f4.x, f4.y, f4.z, invMass, dt, is2D);
  if (is2D) v.z = 0.0f;
  vel[3 * i] = v.x; vel[3 * i + 1] = v.y; vel[3 * i + 2] = v.z;
}

template <int STEP>
__global__ void __launch_bounds__(kIB) k_verletnvt_basic(float4 *__restrict__ pos, float *__restrict__ vel,
                                                         float4 *__restrict__ force, const float *__restrict__ mass,
                                                         float defaultMass, const int *__restrict__ index, int N,
                                                         float dt, float friction, int is2D, float noiseAmplitude,
                                                         uint stepNum, uint seed) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  const float invMass = 1.0f / (defaultMass > 0 ? defaultMass : mass[i]);
  Saru rng((uint)(id + N * (STEP - 1)), stepNum, seed);
  noiseAmplitude *= sqrtf((float)(0.5 * (double)invMass));
  const float2 n01 = rng.gf(0.0f, noiseAmplitude);
  const float nz = rng.gf(0.0f, noiseAmplitude).x;
  float3 v = make_float3(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]);
  const float4 f4 = force[i];
  const float hdt = dt * 0.5f;
  v.x = v.x + fmaf(fmaf(f4.x, invMass, -(friction * v.x)), hdt, n01.x);
  v.y = v.y + fmaf(fmaf(f4.y, invMass, -(friction * v.y)), hdt, n01.y);
  v.z = v.z + fmaf(fmaf(f4.z, invMass, -(friction * v.z)), hdt, nz);
  if (is2D) v.z = 0.0f;
  vel[3 * i] = v.x; vel[3 * i + 1] = v.y; vel[3 * i + 2] = v.z;
  if (STEP == 1) {
    float4 p = pos[i];
    p.x = fmaf(v.x, dt, p.x);
    p.y = fmaf(v.y, dt, p.y);
    p.z = fmaf(v.z, dt, p.z);
    pos[i] = p;
    force[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Basic.cu:12-29: mass ignored; gd() = float
// (the reference indexes the group iterator TWICE, indexIterator[indexIterator[id]]: the identity for the "All" group and an
// out-of-bounds read for a proper sub-group — DESIGN.md deviations; one look-up here),
// uniforms + float libm scaled in double.
__global__ void __launch_bounds__(kIB) k_initial_velocities(float *__restrict__ vel, const int *__restrict__ index,
                                                            float vamp, int is2D, int N, uint seed) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  Saru rng((uint)id, seed);
  const int i = index ? index[id] : id;
  auto gd = [&](double mean, double std, double &o0, double &o1) {
    const double pi2 = 2.0 * 3.14159265358979323846;
    double u0;
    do { u0 = rng.f(); } while (u0 <= 2.2250738585072014e-308);
    const double u1 = rng.f();
    const double r = sqrtf((float)(-2.0 * logf((float)u0)));
    const double theta = pi2 * u1;
    o0 = r * sinf((float)theta) * std + mean;
    o1 = r * cosf((float)theta) * std + mean;
  };
  double nx, ny, nz = 0.0, tmp;
  gd(0.0, (double)vamp, nx, ny);
  if (!is2D) gd(0.0, (double)vamp, nz, tmp);
  const int idx = i;
  vel[3 * idx] = (float)nx; vel[3 * idx + 1] = (float)ny; vel[3 * idx + 2] = (float)nz;
}

// VerletNVT::Basic::sumKineticEnergy (VerletNVT/Basic.cu:173-207): energy[i] += 0.5 |v|^2 m
__global__ void __launch_bounds__(kIB) k_sum_kinetic_energy(const float *__restrict__ vel, float *__restrict__ energy,
                                                            const float *__restrict__ mass, float defaultMass,
                                                            const int *__restrict__ index, int N) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  const float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
  const float m = (defaultMass > 0.f || !mass) ? defaultMass : mass[i];
  const float v2 = __fmaf_rn(vz, vz, __fmaf_rn(vy, vy, vx * vx));
  energy[i] += 0.5f * v2 * m;
}

struct Shear { float3 Kx, Ky, Kz; };

__global__ void __launch_bounds__(kIB) k_bd_euler_maruyama(float4 *__restrict__ pos, const int *__restrict__ index,
                                                           const float4 *__restrict__ force, Shear K,
                                                           float selfMobility, const float *__restrict__ radius,
                                                           float dt, int is2D, float temperature, int N, uint stepNum,
                                                           uint seed) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  float4 p = pos[i];
  const float4 F = force[i];
  const real3f R{p.x, p.y, p.z};
  const float KRx = dot3(real3f{K.Kx.x, K.Kx.y, K.Kx.z}, R);
  const float KRy = dot3(real3f{K.Ky.x, K.Ky.y, K.Ky.z}, R);
  const float KRz = dot3(real3f{K.Kz.x, K.Kz.y, K.Kz.z}, R);
  const float M = selfMobility * (radius ? (1.0f / radius[i]) : 1.0f);
  float rx = fmaf(dt, fmaf(M, F.x, KRx), p.x);
  float ry = fmaf(dt, fmaf(M, F.y, KRy), p.y);
  float rz = fmaf(dt, fmaf(M, F.z, KRz), p.z);
  if (temperature > 0) {
    Saru rng((uint)i, stepNum, seed);
    const float B = sqrtf(2.0f * temperature * M * dt);
    const float2 d01 = rng.gf(0.0f, B);
    const float dz = rng.gf(0.0f, B).x;
    rx += d01.x; ry += d01.y; rz += dz;
  }
  p.x = rx;
  p.y = ry;
  if (!is2D) p.z = rz;
  pos[i] = p;
}

// The other three schemes of Integrator/BrownianDynamics.cu — one streaming kernel, the scheme a template parameter:
//   SCHEME 1, SUB 0 / 1  MidPoint (:178-214): half a step from the current forces (the starting point kept in aux[id]), then a whole step
//                        from the kept point with the midpoint's forces; variance T M dt per draw, the second sub-step repeats the first
//                        draw and adds another; the generator keyed by the GROUP index id
//   SCHEME 2             AdamsBashforth (:262-289): forces 3/2 F_n - 1/2 F_(n-1) (aux[id] = F_(n-1) in group order), sqrt(2 T M dt), keyed by id
//   SCHEME 3             Leimkuhler (:313-345): Euler drift, noise sqrt(T M dt / 2) (dW_n + dW_(n-1)), keyed by originalIndex[i]
template <int SCHEME, int SUB>
__global__ void __launch_bounds__(kIB) k_bd_scheme(float4 *__restrict__ pos, float4 *__restrict__ aux, const int *__restrict__ index,
                                                   const int *__restrict__ originalIndex, const float4 *__restrict__ force, Shear K,
                                                   float selfMobility, const float *__restrict__ radius, float dt, int is2D, float temperature,
                                                   int N, uint stepNum, uint seed) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  float4 p = pos[i];
  if (SCHEME == 1) {
    if (SUB == 0) aux[id] = p;
    else { const float4 q = aux[id]; p.x = q.x; p.y = q.y; p.z = q.z; }
  }
  const float4 F = force[i];
  const real3f R{p.x, p.y, p.z};
  float KRx = dot3(real3f{K.Kx.x, K.Kx.y, K.Kx.z}, R), KRy = dot3(real3f{K.Ky.x, K.Ky.y, K.Ky.z}, R), KRz = dot3(real3f{K.Kz.x, K.Kz.y, K.Kz.z}, R);
  const float M = selfMobility * (radius ? (1.0f / radius[i]) : 1.0f);
  float fx = F.x, fy = F.y, fz = F.z;
  if (SCHEME == 1 && SUB == 0) { fx *= 0.5f; fy *= 0.5f; fz *= 0.5f; KRx *= 0.5f; KRy *= 0.5f; KRz *= 0.5f; }
  if (SCHEME == 2) {
    const float4 Fp = aux[id];
    fx = fmaf(-0.5f, Fp.x, 1.5f * F.x); fy = fmaf(-0.5f, Fp.y, 1.5f * F.y); fz = fmaf(-0.5f, Fp.z, 1.5f * F.z);
  }
  float rx = fmaf(dt, fmaf(M, fx, KRx), p.x), ry = fmaf(dt, fmaf(M, fy, KRy), p.y), rz = fmaf(dt, fmaf(M, fz, KRz), p.z);
  if (temperature > 0.0f) {
    if (SCHEME == 3) {
      const uint ori = (uint)(originalIndex ? originalIndex[i] : i);
      const float B = sqrtf(0.5f * temperature * M * dt);
      Saru a(ori, stepNum, seed), b(ori, stepNum - 1u, seed);
      const float2 a01 = a.gf(0.0f, 1.0f);
      const float a2 = a.gf(0.0f, 1.0f).x;
      const float2 b01 = b.gf(0.0f, 1.0f);
      const float b2 = b.gf(0.0f, 1.0f).x;
      rx = fmaf(B, a01.x + b01.x, rx); ry = fmaf(B, a01.y + b01.y, ry); rz = fmaf(B, a2 + b2, rz);
    } else {
      const float B = SCHEME == 1 ? sqrtf(temperature * M * dt) : sqrtf(2.0f * temperature * M * dt);
      Saru rng((uint)id, stepNum, seed);
      const float2 d01 = rng.gf(0.0f, B);
      const float d2 = rng.gf(0.0f, B).x;
      rx += d01.x; ry += d01.y; rz += d2;
      if (SCHEME == 1 && SUB == 1) {
        const float2 e01 = rng.gf(0.0f, B);
        const float e2 = rng.gf(0.0f, B).x;
        rx += e01.x; ry += e01.y; rz += e2;
      }
    }
  }
  p.x = rx;
  p.y = ry;
  if (!is2D) p.z = rz;
  if (SCHEME == 1 && SUB == 1) {   // (only x, y, z come from the kept point: pos[i].w is the particle's own)
    const float w = pos[i].w;
    p.w = w;
  }
  pos[i] = p;
}

__global__ void __launch_bounds__(kIB) k_fcm_euler_maruyama(float4 *__restrict__ pos, const int *__restrict__ index,
                                                            const float *__restrict__ linearV, int N, float dt) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  float4 p = pos[i];
  p.x = fmaf(linearV[3 * id], dt, p.x);
  p.y = fmaf(linearV[3 * id + 1], dt, p.y);
  p.z = fmaf(linearV[3 * id + 2], dt, p.z);
  pos[i] = p;
}

// BDHI::EulerMaruyama_ns::integrateGPUD (Integrator/BDHI/BDHI_EulerMaruyama.cu:82-113): dR = dt (K R + M F) + sqrt(2 T dt) B dW
struct Shear9 { float k[9]; };
__global__ void __launch_bounds__(kIB) k_bdhi_euler_maruyama(float4 *__restrict__ pos, const int *__restrict__ index,
                                                             const float *__restrict__ MF, const float *__restrict__ BdW,
                                                             Shear9 K, bool haveK, int N, float sqrt2Tdt, float dt, bool is2D) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  float4 p = pos[i];
  if (haveK) {
    const float krx = fmaf(K.k[2], p.z, fmaf(K.k[1], p.y, K.k[0] * p.x));
    const float kry = fmaf(K.k[5], p.z, fmaf(K.k[4], p.y, K.k[3] * p.x));
    const float krz = is2D ? 0.0f : fmaf(K.k[8], p.z, fmaf(K.k[7], p.y, K.k[6] * p.x));
    p.x = fmaf(krx, dt, p.x);
    p.y = fmaf(kry, dt, p.y);
    p.z = fmaf(krz, dt, p.z);
  }
  p.x = fmaf(MF[3 * id], dt, p.x);
  p.y = fmaf(MF[3 * id + 1], dt, p.y);
  p.z = fmaf(MF[3 * id + 2], dt, p.z);
  if (BdW) {
    p.x = fmaf(sqrt2Tdt, BdW[3 * id], p.x);
    p.y = fmaf(sqrt2Tdt, BdW[3 * id + 1], p.y);
    p.z = fmaf(sqrt2Tdt, is2D ? 0.0f : BdW[3 * id + 2], p.z);
  }
  pos[i] = p;
}

// FCM_ns::integrateEulerMaruyamaD with orientations (BDHI_FCM.cu:67-92): dir = rotVec2Quaternion(angularV dt) * dir
__global__ void __launch_bounds__(kIB) k_fcm_euler_maruyama_dir(float4 *__restrict__ pos, float4 *__restrict__ dir,
                                                                const int *__restrict__ index, const float *__restrict__ linearV,
                                                                const float *__restrict__ angularV, int N, float dt) {
  const int id = blockIdx.x * kIB + threadIdx.x;
  if (id >= N) return;
  const int i = index ? index[id] : id;
  float4 p = pos[i];
  p.x = fmaf(linearV[3 * id], dt, p.x);
  p.y = fmaf(linearV[3 * id + 1], dt, p.y);
  p.z = fmaf(linearV[3 * id + 2], dt, p.z);
  pos[i] = p;
  if (dir) {
    float dx = angularV[3 * id] * dt, dy = angularV[3 * id + 1] * dt, dz = angularV[3 * id + 2] * dt;
    const float norm = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    float qn = 1.f, qx = 0.f, qy = 0.f, qz = 0.f;
    if (norm != 0.0f) {
      const float phi = sqrtf(norm);
      const float inv = 1.0f / phi;
      dx *= inv; dy *= inv; dz *= inv;
      float sn, cs;
      sincosf(phi * 0.5f, &sn, &cs);
      qn = cs; qx = sn * dx; qy = sn * dy; qz = sn * dz;
    }
    const float4 o = dir[i];  // (n, v)
    float4 r;
    r.x = qn * o.x - (qx * o.y + qy * o.z + qz * o.w);
    r.y = qn * o.y + o.x * qx + (qy * o.w - qz * o.z);
    r.z = qn * o.z + o.x * qy + (qz * o.y - qx * o.w);
    r.w = qn * o.w + o.x * qz + (qx * o.z - qy * o.y);
    dir[i] = r;
  }
}

static inline int nb(int n) { return (n + kIB - 1) / kIB; }

}  // namespace uammd_hip

using namespace uammd_hip;

extern "C" {

int uammd_verletnvt_gj(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                       const int *d_index, int N, float dt, float friction, int is2D, float noiseAmplitude,
                       unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_verletnvt_gj: no mass array and defaultMass <= 0"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  if (step == 1)
    hipLaunchKernelGGL(k_verletnvt_gj<1>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed);
  else
    hipLaunchKernelGGL(k_verletnvt_gj<2>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_verletnvt_gj_keyed(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass, float defaultMass,
                             const int *d_index, const int *d_noiseKey, int N, float dt, float friction, int is2D, float noiseAmplitude,
                             unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_verletnvt_gj_keyed: no mass array and defaultMass <= 0"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  if (step == 1)
    hipLaunchKernelGGL(k_verletnvt_gj<1>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed, d_noiseKey);
  else
    hipLaunchKernelGGL(k_verletnvt_gj<2>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed, d_noiseKey);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_verletnvt_basic(int step, float *d_pos, float *d_vel, float *d_force, const float *d_mass,
                          float defaultMass, const int *d_index, int N, float dt, float friction, int is2D,
                          float noiseAmplitude, unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  if (!d_mass && !(defaultMass > 0)) { set_last_error("uammd_verletnvt_basic: no mass array and defaultMass <= 0"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  if (step == 1)
    hipLaunchKernelGGL(k_verletnvt_basic<1>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed);
  else
    hipLaunchKernelGGL(k_verletnvt_basic<2>, dim3(nb(N)), dim3(kIB), 0, st, (float4 *)d_pos, d_vel, (float4 *)d_force,
                       d_mass, defaultMass, d_index, N, dt, friction, is2D, noiseAmplitude, stepNum, seed);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_verletnvt_initial_velocities(float *d_vel, const int *d_index, float velAmplitude, int is2D, int N,
                                       unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_initial_velocities, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, d_vel, d_index,
                     velAmplitude, is2D, N, seed);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_sum_kinetic_energy(const float *d_vel, float *d_energy, const float *d_mass, float defaultMass,
                             const int *d_index, int N, void *stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_sum_kinetic_energy, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, d_vel, d_energy, d_mass,
                     defaultMass, d_index, N);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_bd_euler_maruyama(float *d_pos, const int *d_index, const float *d_force, const float K[9],
                            float selfMobility, const float *d_radius, float dt, int is2D, float temperature, int N,
                            unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  Shear S{make_float3(0, 0, 0), make_float3(0, 0, 0), make_float3(0, 0, 0)};
  if (K) {
    S.Kx = make_float3(K[0], K[1], K[2]);
    S.Ky = make_float3(K[3], K[4], K[5]);
    S.Kz = make_float3(K[6], K[7], K[8]);
  }
  hipLaunchKernelGGL(k_bd_euler_maruyama, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, (float4 *)d_pos, d_index,
                     (const float4 *)d_force, S, selfMobility, d_radius, dt, is2D, temperature, N, stepNum, seed);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_bd_scheme_step(int scheme, int substep, float *d_pos, float *d_aux, const int *d_index, const int *d_originalIndex, const float *d_force,
                         const float K[9], float selfMobility, const float *d_radius, float dt, int is2D, float temperature, int N,
                         unsigned int stepNum, unsigned int seed, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_force || ((scheme == UAMMD_BD_MIDPOINT || scheme == UAMMD_BD_ADAMS_BASHFORTH) && !d_aux)) {
    set_last_error("uammd_bd_scheme_step: null argument");
    return -1;
  }
  Shear S{make_float3(0, 0, 0), make_float3(0, 0, 0), make_float3(0, 0, 0)};
  if (K) {
    S.Kx = make_float3(K[0], K[1], K[2]);
    S.Ky = make_float3(K[3], K[4], K[5]);
    S.Kz = make_float3(K[6], K[7], K[8]);
  }
  const dim3 g(nb(N)), b(kIB);
  hipStream_t st = (hipStream_t)stream;
#define UH_BD_LAUNCH(SC, SU) hipLaunchKernelGGL((k_bd_scheme<SC, SU>), g, b, 0, st, (float4 *)d_pos, (float4 *)d_aux, d_index, d_originalIndex, \
                                                (const float4 *)d_force, S, selfMobility, d_radius, dt, is2D, temperature, N, stepNum, seed)
  if (scheme == UAMMD_BD_MIDPOINT && substep == 0) UH_BD_LAUNCH(1, 0);
  else if (scheme == UAMMD_BD_MIDPOINT && substep == 1) UH_BD_LAUNCH(1, 1);
  else if (scheme == UAMMD_BD_ADAMS_BASHFORTH) UH_BD_LAUNCH(2, 0);
  else if (scheme == UAMMD_BD_LEIMKUHLER) UH_BD_LAUNCH(3, 0);
  else { set_last_error("uammd_bd_scheme_step: unknown scheme %d / sub-step %d", scheme, substep); return -1; }
#undef UH_BD_LAUNCH
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_fcm_euler_maruyama(float *d_pos, const int *d_index, const float *d_linearVelocity, int N, float dt,
                             void *stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_fcm_euler_maruyama, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, (float4 *)d_pos, d_index,
                     d_linearVelocity, N, dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_bdhi_euler_maruyama(float *d_pos, const int *d_index, const float *d_MF, const float *d_BdW, const float K[9],
                              int N, float sqrt2Tdt, float dt, int is2D, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_MF) { set_last_error("uammd_bdhi_euler_maruyama: null argument"); return -1; }
  Shear9 k{};
  if (K) for (int t = 0; t < 9; ++t) k.k[t] = K[t];
  hipLaunchKernelGGL(k_bdhi_euler_maruyama, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, (float4 *)d_pos, d_index, d_MF,
                     d_BdW, k, K != nullptr, N, sqrt2Tdt, dt, is2D != 0);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_fcm_euler_maruyama_dir(float *d_pos, float *d_dir, const int *d_index, const float *d_linearVelocity,
                                 const float *d_angularVelocity, int N, float dt, void *stream) {
  if (N <= 0) return 0;
  if (!d_pos || !d_linearVelocity || (d_dir && !d_angularVelocity)) { set_last_error("uammd_fcm_euler_maruyama_dir: null argument"); return -1; }
  hipLaunchKernelGGL(k_fcm_euler_maruyama_dir, dim3(nb(N)), dim3(kIB), 0, (hipStream_t)stream, (float4 *)d_pos, (float4 *)d_dir,
                     d_index, d_linearVelocity, d_angularVelocity, N, dt);
  UH_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
