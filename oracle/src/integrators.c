/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Integrator kernels on the hot path, restated from
 *   VerletNVT::GronbechJensen_ns::integrateGPU<step> (K11)   Integrator/VerletNVT/GronbechJensen.cu:28-62
 *   VerletNVT::Basic_ns::integrateGPU<step>          (K11)   Integrator/VerletNVT/Basic.cu:86-114
 *   VerletNVT::Basic_ns::initialVelocities                   Integrator/VerletNVT/Basic.cu:12-29
 *   BD::EulerMaruyama_ns::integrateGPU               (K12)   Integrator/BrownianDynamics.cu:119-144
 *   BDHI::FCM_ns::integrateEulerMaruyamaD            (K20)   Integrator/BDHI/BDHI_FCM.cu:67-92  (no orientations)
 *   BDHI::EulerMaruyama_ns::integrateGPUD            (K21)   Integrator/BDHI/BDHI_EulerMaruyama.cu:82-113
 *
 * Argument-evaluation order: `make_real3(rng.gf(..), rng.gf(..).x)` has unspecified order in C++;
 * the oracle pins left-to-right (first draw -> x,y; second draw -> z), which is what clang-based
 * device compilers (and the HIP side of this repo) do.
 * The Gaussians go through logf/sinf/cosf: compared with a tolerance, never bit-exact.
 */
#include "common.h"
#include "saru.h"

static inline void gf_real(Saru *s, real mean, real std, real *a, real *b) {
  float x, y;
  saru_gf(s, (float)mean, (float)std, &x, &y);
  *a = x; *b = y;
}

/* GronbechJensen.cu:28-62.  vel is real3 (stride 3).  mass nullable (defaultMass>0 wins). */
ORACLE_API void oracle_verletnvt_gj(int step, real4 *pos, real *vel3, real4 *force, const real *mass, real defaultMass,
                                    const int *indexIterator, int N, real dt, real friction, int is2D,
                                    real noiseAmplitude_in, uint stepNum, uint seed) {
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    real noiseAmplitude = noiseAmplitude_in;
    const real invMass = (real)1.0 / (defaultMass > 0 ? defaultMass : mass[i]);
    real3 v = mk3(vel3[3 * i], vel3[3 * i + 1], vel3[3 * i + 2]);
    real3 f = mk3(force[i].x, force[i].y, force[i].z);
    if (step == 1) {
      Saru rng = saru3((uint)id, stepNum, seed);
      noiseAmplitude *= (real)1.0 / SQRT(invMass); /* rsqrt(invMass) */
      real3 noisei;
      real tmp;
      gf_real(&rng, 0, noiseAmplitude, &noisei.x, &noisei.y);
      if (is2D) noisei.z = 0;
      else gf_real(&rng, 0, noiseAmplitude, &noisei.z, &tmp);
      const real gdthalfinvMass = friction * dt * (real)0.5;
      const real b = (real)1.0 / ((real)1.0 + gdthalfinvMass);
      const real a = ((real)1.0 - gdthalfinvMass) * b;
      real3 p = mk3(pos[i].x, pos[i].y, pos[i].z);
      /* p = p + b*dt*vel + 0.5*invMass*dt*b*(dt*force + noise) */
      const real bdt = b * dt;
      const real c = (real)0.5 * invMass * dt * b;
      p.x = FMA(c, FMA(dt, f.x, noisei.x), FMA(bdt, v.x, p.x));
      p.y = FMA(c, FMA(dt, f.y, noisei.y), FMA(bdt, v.y, p.y));
      p.z = FMA(c, FMA(dt, f.z, noisei.z), FMA(bdt, v.z, p.z));
      pos[i].x = p.x; pos[i].y = p.y; pos[i].z = p.z;
      /* vel = a*vel + dt*0.5*invMass*a*force + b*invMass*noise */
      const real d = dt * (real)0.5 * invMass * a;
      const real e = b * invMass;
      v.x = FMA(e, noisei.x, FMA(d, f.x, a * v.x));
      v.y = FMA(e, noisei.y, FMA(d, f.y, a * v.y));
      v.z = FMA(e, noisei.z, FMA(d, f.z, a * v.z));
      force[i].x = 0; force[i].y = 0; force[i].z = 0; force[i].w = 0;
    } else {
      const real d = dt * (real)0.5 * invMass;
      v.x = FMA(d, f.x, v.x);
      v.y = FMA(d, f.y, v.y);
      v.z = FMA(d, f.z, v.z);
    }
    if (is2D) v.z = (real)0.0;
    vel3[3 * i] = v.x; vel3[3 * i + 1] = v.y; vel3[3 * i + 2] = v.z;
  }
}

/* Basic.cu:86-114 */
ORACLE_API void oracle_verletnvt_basic(int step, real4 *pos, real *vel3, real4 *force, const real *mass,
                                       real defaultMass, const int *indexIterator, int N, real dt, real friction,
                                       int is2D, real noiseAmplitude_in, uint stepNum, uint seed) {
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    const real invMass = (real)1.0 / (defaultMass > 0 ? defaultMass : mass[i]);
    Saru rng = saru3((uint)(id + N * (step - 1)), stepNum, seed);
    real noiseAmplitude = noiseAmplitude_in * (real)sqrtf((float)(0.5 * (double)invMass));
    real3 noisei;
    real tmp;
    gf_real(&rng, 0, noiseAmplitude, &noisei.x, &noisei.y);
    gf_real(&rng, 0, noiseAmplitude, &noisei.z, &tmp);
    real3 v = mk3(vel3[3 * i], vel3[3 * i + 1], vel3[3 * i + 2]);
    real3 f = mk3(force[i].x, force[i].y, force[i].z);
    const real hdt = dt * (real)0.5;
    /* vel += (force*invMass - friction*vel)*(dt*0.5) + noise */
    v.x = v.x + FMA(FMA(f.x, invMass, -(friction * v.x)), hdt, noisei.x);
    v.y = v.y + FMA(FMA(f.y, invMass, -(friction * v.y)), hdt, noisei.y);
    v.z = v.z + FMA(FMA(f.z, invMass, -(friction * v.z)), hdt, noisei.z);
    if (is2D) v.z = (real)0.0;
    vel3[3 * i] = v.x; vel3[3 * i + 1] = v.y; vel3[3 * i + 2] = v.z;
    if (step == 1) {
      pos[i].x = FMA(v.x, dt, pos[i].x);
      pos[i].y = FMA(v.y, dt, pos[i].y);
      pos[i].z = FMA(v.z, dt, pos[i].z);
      force[i].x = 0; force[i].y = 0; force[i].z = 0; force[i].w = 0;
    }
  }
}

/* Basic.cu:12-29 — ignores mass, double-indexes the iterator (reproduced). */
ORACLE_API void oracle_verletnvt_initial_velocities(real *vel3, const int *indexIterator, real vamp, int is2D, int N,
                                                    uint seed) {
  for (int id = 0; id < N; id++) {
    Saru rng = saru2((uint)id, seed);
    int i = indexIterator ? indexIterator[id] : id;
    double mass_i = 1.0;
    double nx, ny, nz = 0.0, tmp;
    saru_gd(&rng, 0, (double)vamp / mass_i, &nx, &ny);
    if (!is2D) saru_gd(&rng, 0, (double)vamp / mass_i, &nz, &tmp);
    int index = i; /* the reference reads indexIterator[i] again: identity for "All", out of bounds for a sub-group */
    vel3[3 * index] = (real)nx; vel3[3 * index + 1] = (real)ny; vel3[3 * index + 2] = (real)nz;
  }
}

/* BrownianDynamics.cu:119-144.  K = shear matrix rows (9 reals, nullable = zero). */
ORACLE_API void oracle_bd_euler_maruyama(real4 *pos, const int *indexIterator, const real4 *force, const real *K9,
                                         real selfMobility, const real *radius, real dt, int is2D, real temperature,
                                         int N, uint stepNum, uint seed) {
  real3 Kx = mk3(0, 0, 0), Ky = Kx, Kz = Kx;
  if (K9) { Kx = mk3(K9[0], K9[1], K9[2]); Ky = mk3(K9[3], K9[4], K9[5]); Kz = mk3(K9[6], K9[7], K9[8]); }
  for (int id = 0; id < N; id++) {
    int i = indexIterator ? indexIterator[id] : id;
    real3 R = mk3(pos[i].x, pos[i].y, pos[i].z);
    real3 F = mk3(force[i].x, force[i].y, force[i].z);
    real3 KR = mk3(dot3(Kx, R), dot3(Ky, R), dot3(Kz, R));
    real M = selfMobility * (radius ? ((real)1.0 / radius[i]) : (real)1.0);
    /* R += dt*(KR + M*F) */
    R.x = FMA(dt, FMA(M, F.x, KR.x), R.x);
    R.y = FMA(dt, FMA(M, F.y, KR.y), R.y);
    R.z = FMA(dt, FMA(M, F.z, KR.z), R.z);
    if (temperature > 0) {
      Saru rng = saru3((uint)i, stepNum, seed);
      real B = SQRT((real)2.0 * temperature * M * dt);
      real3 dW;
      real tmp;
      gf_real(&rng, 0, B, &dW.x, &dW.y);
      gf_real(&rng, 0, B, &dW.z, &tmp);
      R.x += dW.x; R.y += dW.y; R.z += dW.z;
    }
    pos[i].x = R.x;
    pos[i].y = R.y;
    if (!is2D) pos[i].z = R.z;
  }
}

/* BD::MidPoint, MidPoint_ns::integrateGPU<step> (BrownianDynamics.cu:178-214).  step 0: half a step of drift from the CURRENT forces plus a
 * draw of variance T M dt, the starting position kept in initialPositions[id]; step 1: a whole step from the kept position with the forces
 * at the midpoint, the SAME first draw (the generator is keyed (id, stepNum, seed) both times) and a second one.  The generator is keyed
 * by the GROUP index id, not the particle index i (:203). */
ORACLE_API void oracle_bd_midpoint(int step, real4 *pos, real4 *initialPositions, const int *indexIterator, const real4 *force, const real *K9,
                                   real selfMobility, const real *radius, real dt, int is2D, real temperature, int N, uint stepNum, uint seed) {
  real3 Kx = mk3(0, 0, 0), Ky = Kx, Kz = Kx;
  if (K9) { Kx = mk3(K9[0], K9[1], K9[2]); Ky = mk3(K9[3], K9[4], K9[5]); Kz = mk3(K9[6], K9[7], K9[8]); }
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    real3 p;
    if (step == 0) {
      p = mk3(pos[i].x, pos[i].y, pos[i].z);
      initialPositions[id] = pos[i];
    } else {
      p = mk3(initialPositions[id].x, initialPositions[id].y, initialPositions[id].z);
    }
    const real M = selfMobility * (radius ? ((real)1.0 / radius[i]) : (real)1.0);
    real3 KR = mk3(dot3(Kx, p), dot3(Ky, p), dot3(Kz, p));
    real3 f = mk3(force[i].x, force[i].y, force[i].z);
    if (step == 0) {
      f.x *= (real)0.5; f.y *= (real)0.5; f.z *= (real)0.5;
      KR.x *= (real)0.5; KR.y *= (real)0.5; KR.z *= (real)0.5;
    }
    p.x = FMA(dt, FMA(M, f.x, KR.x), p.x);
    p.y = FMA(dt, FMA(M, f.y, KR.y), p.y);
    p.z = FMA(dt, FMA(M, f.z, KR.z), p.z);
    if (temperature > (real)0.0) {
      const real B = SQRT(temperature * M * dt);
      Saru rng = saru3((uint)id, stepNum, seed);
      real3 dW;
      real tmp;
      gf_real(&rng, 0, B, &dW.x, &dW.y);
      gf_real(&rng, 0, B, &dW.z, &tmp);
      p.x += dW.x; p.y += dW.y; p.z += dW.z;
      if (step == 1) {
        gf_real(&rng, 0, B, &dW.x, &dW.y);
        gf_real(&rng, 0, B, &dW.z, &tmp);
        p.x += dW.x; p.y += dW.y; p.z += dW.z;
      }
    }
    pos[i].x = p.x;
    pos[i].y = p.y;
    if (!is2D) pos[i].z = p.z;
  }
}

/* BD::AdamsBashforth, AdamsBashforth_ns::integrateGPU (BrownianDynamics.cu:262-289): x += dt (K x + M (3/2 F_n - 1/2 F_(n-1))) + sqrt(2 T M dt) dW,
 * previousForces in GROUP order, the generator keyed by the group index. */
ORACLE_API void oracle_bd_adams_bashforth(real4 *pos, const real4 *previousForces, const int *indexIterator, const real4 *force, const real *K9,
                                          real selfMobility, const real *radius, real dt, int is2D, real temperature, int N, uint stepNum,
                                          uint seed) {
  real3 Kx = mk3(0, 0, 0), Ky = Kx, Kz = Kx;
  if (K9) { Kx = mk3(K9[0], K9[1], K9[2]); Ky = mk3(K9[3], K9[4], K9[5]); Kz = mk3(K9[6], K9[7], K9[8]); }
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    real3 p = mk3(pos[i].x, pos[i].y, pos[i].z);
    const real M = selfMobility * (radius ? ((real)1.0 / radius[i]) : (real)1.0);
    const real3 KR = mk3(dot3(Kx, p), dot3(Ky, p), dot3(Kz, p));
    /* 1.5 fn - 0.5 fprev: one fused multiply-add on the product 1.5 fn */
    const real ax = FMA((real)-0.5, previousForces[id].x, (real)1.5 * force[i].x);
    const real ay = FMA((real)-0.5, previousForces[id].y, (real)1.5 * force[i].y);
    const real az = FMA((real)-0.5, previousForces[id].z, (real)1.5 * force[i].z);
    p.x = FMA(dt, FMA(M, ax, KR.x), p.x);
    p.y = FMA(dt, FMA(M, ay, KR.y), p.y);
    p.z = FMA(dt, FMA(M, az, KR.z), p.z);
    if (temperature > (real)0.0) {
      const real B = SQRT((real)2.0 * temperature * M * dt);
      Saru rng = saru3((uint)id, stepNum, seed);
      real3 dW;
      real tmp;
      gf_real(&rng, 0, B, &dW.x, &dW.y);
      gf_real(&rng, 0, B, &dW.z, &tmp);
      p.x += dW.x; p.y += dW.y; p.z += dW.z;
    }
    pos[i].x = p.x;
    pos[i].y = p.y;
    if (!is2D) pos[i].z = p.z;
  }
}

/* BD::Leimkuhler, Leimkuhler_ns::integrateGPU (BrownianDynamics.cu:313-345): Euler drift, noise sqrt(T M dt / 2) (dW_n + dW_(n-1)) with
 * dW_k = unit Gaussians keyed (originalIndex[i], k, seed): this step's second draw is the next step's first. */
ORACLE_API void oracle_bd_leimkuhler(real4 *pos, const int *indexIterator, const int *originalIndex, const real4 *force, const real *K9,
                                     real selfMobility, const real *radius, real dt, int is2D, real temperature, int N, uint stepNum, uint seed) {
  real3 Kx = mk3(0, 0, 0), Ky = Kx, Kz = Kx;
  if (K9) { Kx = mk3(K9[0], K9[1], K9[2]); Ky = mk3(K9[3], K9[4], K9[5]); Kz = mk3(K9[6], K9[7], K9[8]); }
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    real3 R = mk3(pos[i].x, pos[i].y, pos[i].z);
    const real3 F = mk3(force[i].x, force[i].y, force[i].z);
    const real3 KR = mk3(dot3(Kx, R), dot3(Ky, R), dot3(Kz, R));
    const real M = selfMobility * (radius ? ((real)1.0 / radius[i]) : (real)1.0);
    R.x = FMA(dt, FMA(M, F.x, KR.x), R.x);
    R.y = FMA(dt, FMA(M, F.y, KR.y), R.y);
    R.z = FMA(dt, FMA(M, F.z, KR.z), R.z);
    if (temperature > 0) {
      const int ori = originalIndex ? originalIndex[i] : i;
      const real B = SQRT((real)0.5 * temperature * M * dt);
      Saru a = saru3((uint)ori, stepNum, seed), b = saru3((uint)ori, stepNum - 1u, seed);
      real3 da, db;
      real tmp;
      gf_real(&a, 0, 1, &da.x, &da.y);
      gf_real(&a, 0, 1, &da.z, &tmp);
      gf_real(&b, 0, 1, &db.x, &db.y);
      gf_real(&b, 0, 1, &db.z, &tmp);
      R.x = FMA(B, da.x + db.x, R.x);
      R.y = FMA(B, da.y + db.y, R.y);
      R.z = FMA(B, da.z + db.z, R.z);
    }
    pos[i].x = R.x;
    pos[i].y = R.y;
    if (!is2D) pos[i].z = R.z;
  }
}

/* BDHI_FCM.cu:67-92 without orientations: pos[i] += linearV[id]*dt */
ORACLE_API void oracle_fcm_euler_maruyama(real4 *pos, const int *indexIterator, const real *linearV3, int N, real dt) {
  for (int id = 0; id < N; id++) {
    int i = indexIterator ? indexIterator[id] : id;
    pos[i].x = FMA(linearV3[3 * id], dt, pos[i].x);
    pos[i].y = FMA(linearV3[3 * id + 1], dt, pos[i].y);
    pos[i].z = FMA(linearV3[3 * id + 2], dt, pos[i].z);
  }
}

/* BDHI::EulerMaruyama_ns::integrateGPUD (Integrator/BDHI/BDHI_EulerMaruyama.cu:82-113).  K: 9 reals row major or NULL.
 * Contract: dot(K[r], p) = fma(Kz,pz, fma(Ky,py, Kx*px)); every `p += a*b` is one FMA. */
ORACLE_API void oracle_bdhi_euler_maruyama(real4 *pos, const int *indexIterator, const real *MF3, const real *BdW3, const real *K,
                                           int N, real sqrt2Tdt, real dt, int is2D) {
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    real4 p = pos[i];
    if (K) {
      const real krx = FMA(K[2], p.z, FMA(K[1], p.y, K[0] * p.x));
      const real kry = FMA(K[5], p.z, FMA(K[4], p.y, K[3] * p.x));
      const real krz = is2D ? 0 : FMA(K[8], p.z, FMA(K[7], p.y, K[6] * p.x));
      p.x = FMA(krx, dt, p.x); p.y = FMA(kry, dt, p.y); p.z = FMA(krz, dt, p.z);
    }
    p.x = FMA(MF3[3 * id], dt, p.x); p.y = FMA(MF3[3 * id + 1], dt, p.y); p.z = FMA(MF3[3 * id + 2], dt, p.z);
    if (BdW3) {
      p.x = FMA(sqrt2Tdt, BdW3[3 * id], p.x);
      p.y = FMA(sqrt2Tdt, BdW3[3 * id + 1], p.y);
      p.z = FMA(sqrt2Tdt, is2D ? 0 : BdW3[3 * id + 2], p.z);
    }
    pos[i] = p;
  }
}

/* Raw Saru streams for the tests: n draws of u32 from Saru(s1[,s2[,s3]]) */
ORACLE_API void oracle_saru_u32(int nseeds, uint s1, uint s2, uint s3, int n, uint *out) {
  Saru s = nseeds == 1 ? saru1(s1) : (nseeds == 2 ? saru2(s1, s2) : saru3(s1, s2, s3));
  for (int i = 0; i < n; i++) out[i] = saru_u32(&s);
}
ORACLE_API void oracle_saru_f_range(uint s1, float low, float high, int n, float *out) {
  Saru s = saru1(s1);
  for (int i = 0; i < n; i++) out[i] = saru_f_range(&s, low, high);
}
ORACLE_API void oracle_saru_gf(uint s1, uint s2, uint s3, float mean, float std, int npairs, float *out) {
  Saru s = saru3(s1, s2, s3);
  for (int i = 0; i < npairs; i++) saru_gf(&s, mean, std, &out[2 * i], &out[2 * i + 1]);
}

/* FCM_ns::integrateEulerMaruyamaD (BDHI_FCM.cu:67-92) with orientations: dir = rotVec2Quaternion(angularV*dt) * dir
 * (utils/quaternion.cuh:179-196; Quat product :86-96: (n1 n2 - v1.v2, n1 v2 + n2 v1 + v1 x v2)).  dir = (n, vx, vy, vz). */
ORACLE_API void oracle_fcm_euler_maruyama_dir(real4 *pos, real4 *dir, const int *indexIterator, const real *linearV3,
                                              const real *angularV3, int N, real dt) {
  for (int id = 0; id < N; id++) {
    const int i = indexIterator ? indexIterator[id] : id;
    pos[i].x = FMA(linearV3[3 * id], dt, pos[i].x);
    pos[i].y = FMA(linearV3[3 * id + 1], dt, pos[i].y);
    pos[i].z = FMA(linearV3[3 * id + 2], dt, pos[i].z);
    if (dir) {
      real3 d = mk3(angularV3[3 * id] * dt, angularV3[3 * id + 1] * dt, angularV3[3 * id + 2] * dt);
      const real phi = SQRT(dot3(d, d));
      const real norm = dot3(d, d);
      real qn = 1, qx = 0, qy = 0, qz = 0;
      if (norm != (real)0.0) {
        const real inv = (real)1.0 / SQRT(norm);
        d.x *= inv; d.y *= inv; d.z *= inv;
        const real c = COS(phi * (real)0.5), s = SIN(phi * (real)0.5);
        qn = c; qx = s * d.x; qy = s * d.y; qz = s * d.z;
      }
      const real4 o = dir[i]; /* (n, v) */
      const real n2 = o.x, vx = o.y, vy = o.z, vz = o.w;
      real4 r;
      r.x = qn * n2 - (qx * vx + qy * vy + qz * vz);
      r.y = qn * vx + n2 * qx + (qy * vz - qz * vy);
      r.z = qn * vy + n2 * qy + (qz * vx - qx * vz);
      r.w = qn * vz + n2 * qz + (qx * vy - qy * vx);
      dir[i] = r;
    }
  }
}
