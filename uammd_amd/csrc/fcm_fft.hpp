// Batched 3-D real FFT for the FCM solver, written for gfx950: five passes over the grid instead of the eight of rocFFT (three per
// transform) + the Fourier-space kernel + the interleaving copy, with the z transform, the Stokes / noise operator and the inverse z
// transform fused into one kernel that holds a tile of z-lines in LDS.
//
// Replaces, for axis lengths 2^a 3^b 5^c 7^d 11^e (x even, x and z <= 512, y <= 256) — the sizes Grid's nextFFTWiseSize3D hands out
// (anything else keeps rocFFT):
//   cufftExecR2C / cufftExecC2R (batched 3-D)                      Integrator/BDHI/FCM/FCM_impl.cuh:399-411, :544-581
//   forceFourier2Vel + fourierBrownianNoise between them           FCM_impl.cuh:375-397, :437-512
// Conventions are cuFFT's: forward exp(-i...), inverse exp(+i...), both unnormalised (1/N lives in the Stokes operator), the
// imaginary parts of the kx = 0 and kx = nx/2 self-conjugate inputs of the C2R are ignored.
//
//   k_fft_x_r2c        rows: nx reals -> nx/2 + 1 complex in place (one complex FFT of nx/2 points + untangling), 16 rows per workgroup
//   k_fft_xy_r2c_plane a whole (component, z) plane: rows, then columns, where the plane fits the LDS
//   k_fft_lines        y: strided lines of a z-plane, a tile of <= 16 consecutive kx per workgroup (contiguous 8 tl-byte segments)
//   k_fft_z_fused      z: a tile of consecutive (ky, kx) nodes x all nz x the three components in LDS: forward, operator, inverse
//   k_fft_x_c2r        rows back: three components of 16 rows -> the planar real grids or the gather's interleaved float4 grid
// All FFTs are Stockham autosort on LDS lines, mixed radix: the factor 2^a as radix-8 passes with one or two radix-4 (or one radix-2)
// passes in front, then radix-3, 5, 7 and 11 passes; a pass stages its butterflies in registers (read all, barrier, write all, barrier),
// twiddles from a table in LDS.  Index arithmetic is shifts for power-of-two lengths and an exact float reciprocal otherwise (IDiv).
#pragma once
// (included by fcm.hip INSIDE namespace uammd_hip, after the Fourier-space operator it fuses)

constexpr int kFftThreads = 256;
constexpr int kFftMax = 512;

// i / d and i % d for 0 <= i < 2^21 and a wave-uniform d: a shift and a mask when d is a power of two, otherwise
// (int)((i + 0.5) * (1 / d)) — exact: the product is off by less than i 2^-22 / d < 0.5 / d from (i + 0.5) / d
// P2: the caller knows d is a power of two (the kernels are instantiated for all-power-of-two grids and for the rest: with the choice at
// run time the power-of-two launches of C4 were 7 % slower)
template <bool P2> struct IDivT {
  int d, sh;
  float rcp;
  UH_D explicit IDivT(int dd) : d(dd), sh(31 - __builtin_clz((unsigned)dd)), rcp(1.0f / (float)dd) {}
  UH_D int div(int i) const { return P2 ? (i >> sh) : (int)(((float)i + 0.5f) * rcp); }
  UH_D int mod(int i) const { return P2 ? (i & (d - 1)) : i - __mul24(div(i), d); }
  UH_D int rem(int i, int q) const { return P2 ? (i & (d - 1)) : i - __mul24(q, d); }  // i % d given q = i / d
};

// a length the power-of-two instantiations can treat as one: written as a shift, so that the divisions and remainders of the copy loops
// compile to shifts and masks (with the plain kernel argument the fused z pass was 3 us slower at C4)
template <bool P2> UH_D int fft_len(int n) { return P2 ? (1 << (31 - __builtin_clz((unsigned)n))) : n; }

UH_D float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
UH_D float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * w (SIGN < 0) or a * conj(w) (SIGN > 0), w = exp(-2 pi i k / N) from the table
template <int SIGN> UH_D float2 ctw(float2 a, float2 w) {
  if (SIGN < 0) return make_float2(fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x));
  return make_float2(fmaf(a.x, w.x, a.y * w.y), fmaf(a.y, w.x, -(a.x * w.y)));
}

// exp(-2 pi i k / n), k < n, into LDS (n <= 512: sincospif is exact enough and runs once per workgroup)
template <int NT = kFftThreads> UH_D void fft_twiddles(float2 *tw, int n, int tid) {
  for (int k = tid; k < n; k += NT) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)n, &s, &c);
    tw[k] = make_float2(c, s);
  }
}

template <int R1, int R2, int SIGN> UH_D void fft_butterfly_ct(float2 (&v)[R1 * R2]);
// the R-point DFT of v (SIGN < 0: forward, exp(-2 pi i r m / R)), in place, natural order out
template <int R, int SIGN> UH_D void fft_butterfly(float2 (&v)[R]) {
  auto mul_mi = [](float2 d) { return SIGN < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x); };  // -i d (forward), +i d (inverse)
  if constexpr (R == 6 || R == 9 || R == 12) {
    fft_butterfly_ct<R / 3, 3, SIGN>(v);
  } else if constexpr (R == 2) {
    const float2 a = v[0], c = v[1];
    v[0] = cadd(a, c);
    v[1] = csub(a, c);
  } else if constexpr (R == 3) {
    // y0 = a + (b + c), y1,2 = a - (b + c) / 2 -+ i (sqrt(3) / 2) (b - c)  (forward; the inverse swaps y1 and y2)
    const float2 s = cadd(v[1], v[2]), d = csub(v[1], v[2]);
    const float2 m = make_float2(fmaf(-0.5f, s.x, v[0].x), fmaf(-0.5f, s.y, v[0].y));
    const float2 q = mul_mi(make_float2(0.86602540378443864676f * d.x, 0.86602540378443864676f * d.y));
    v[0] = cadd(v[0], s);
    v[1] = cadd(m, q);
    v[2] = csub(m, q);
  } else if constexpr (R == 4) {
    const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(a0, a2);
    v[1] = cadd(a1, a3);
    v[2] = csub(a0, a2);
    v[3] = csub(a1, a3);
  } else if constexpr (R == 5) {
    // with s1 = v1 + v4, d1 = v1 - v4, s2 = v2 + v3, d2 = v2 - v3, c1 = cos(2 pi / 5), c2 = cos(4 pi / 5), n1 = sin(2 pi / 5), n2 = sin(4 pi / 5):
    //   y0 = v0 + s1 + s2;  y1,4 = (v0 + c1 s1 + c2 s2) -+ i (n1 d1 + n2 d2);  y2,3 = (v0 + c2 s1 + c1 s2) -+ i (n2 d1 - n1 d2)   (forward)
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, n1 = 0.95105651629515357212f, n2 = 0.58778525229247312917f;
    const float2 s1 = cadd(v[1], v[4]), d1 = csub(v[1], v[4]), s2 = cadd(v[2], v[3]), d2 = csub(v[2], v[3]);
    const float2 a1 = make_float2(fmaf(c1, s1.x, fmaf(c2, s2.x, v[0].x)), fmaf(c1, s1.y, fmaf(c2, s2.y, v[0].y)));
    const float2 a2 = make_float2(fmaf(c2, s1.x, fmaf(c1, s2.x, v[0].x)), fmaf(c2, s1.y, fmaf(c1, s2.y, v[0].y)));
    const float2 b1 = mul_mi(make_float2(fmaf(n1, d1.x, n2 * d2.x), fmaf(n1, d1.y, n2 * d2.y)));
    const float2 b2 = mul_mi(make_float2(fmaf(n2, d1.x, -(n1 * d2.x)), fmaf(n2, d1.y, -(n1 * d2.y))));
    v[0] = cadd(v[0], cadd(s1, s2));
    v[1] = cadd(a1, b1);
    v[4] = csub(a1, b1);
    v[2] = cadd(a2, b2);
    v[3] = csub(a2, b2);
  } else if constexpr (R == 7 || R == 11) {
    // an odd prime: with s_r = v_r + v_(R-r), d_r = v_r - v_(R-r) (r = 1 .. H = (R - 1) / 2) and k = r m mod R,
    //   y_m, y_(R-m) = (v_0 + sum_r cos(2 pi k / R) s_r) -+ i sum_r sin(2 pi k / R) d_r   (forward; the inverse swaps the two)
    constexpr int H = (R - 1) / 2;
    constexpr float C7[3] = {0.62348980185873359439f, -0.22252093395631433737f, -0.90096886790241903498f};
    constexpr float S7[3] = {0.78183148246802980363f, 0.97492791218182361934f, 0.43388373911755823142f};
    constexpr float C11[5] = {0.84125353283118120551f, 0.41541501300188643508f, -0.14231483827328500480f, -0.65486073394528498959f,
                              -0.95949297361449736865f};
    constexpr float S11[5] = {0.54064081745559755543f, 0.90963199535451833011f, 0.98982144188093279524f, 0.75574957435425826890f,
                              0.28173255684142967104f};
    float2 s[H], d[H];
#pragma unroll
    for (int r = 1; r <= H; ++r) { s[r - 1] = cadd(v[r], v[R - r]); d[r - 1] = csub(v[r], v[R - r]); }
    float2 y0 = v[0];
#pragma unroll
    for (int r = 0; r < H; ++r) y0 = cadd(y0, s[r]);
    float2 out[R];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
      float2 a = v[0], b = make_float2(0.0f, 0.0f);
#pragma unroll
      for (int r = 1; r <= H; ++r) {
        const int k = (r * m) % R, kk = k <= H ? k : R - k;
        const float c = R == 7 ? C7[kk - 1] : C11[kk - 1];
        const float sn = (R == 7 ? S7[kk - 1] : S11[kk - 1]) * (k <= H ? 1.0f : -1.0f);
        a = make_float2(fmaf(c, s[r - 1].x, a.x), fmaf(c, s[r - 1].y, a.y));
        b = make_float2(fmaf(sn, d[r - 1].x, b.x), fmaf(sn, d[r - 1].y, b.y));
      }
      const float2 q = mul_mi(b);
      out[m] = cadd(a, q);
      out[R - m] = csub(a, q);
    }
    v[0] = y0;
#pragma unroll
    for (int m = 1; m < R; ++m) v[m] = out[m];
  } else {  // 8 = 2 x 4: sums and differences four apart, the differences turned by W8^r, then a 4-point DFT of either half
    float2 e[4], o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = cadd(v[r], v[r + 4]); o[r] = csub(v[r], v[r + 4]); }
    const float h = 0.70710678118654752440f;
    // W8 = exp(-i pi / 4) forward: (x + i y)(h - i h) = h (x + y) + i h (y - x); inverse the conjugate
    o[1] = SIGN < 0 ? make_float2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)) : make_float2(h * (o[1].x - o[1].y), h * (o[1].y + o[1].x));
    o[2] = mul_mi(o[2]);
    // W8^3 = exp(-3 i pi / 4) forward: (x + i y)(-h - i h) = h (y - x) - i h (x + y)
    o[3] = SIGN < 0 ? make_float2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)) : make_float2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y));
    {
      const float2 a0 = cadd(e[0], e[2]), a1 = csub(e[0], e[2]), a2 = cadd(e[1], e[3]), a3 = mul_mi(csub(e[1], e[3]));
      v[0] = cadd(a0, a2); v[2] = cadd(a1, a3); v[4] = csub(a0, a2); v[6] = csub(a1, a3);
    }
    {
      const float2 a0 = cadd(o[0], o[2]), a1 = csub(o[0], o[2]), a2 = cadd(o[1], o[3]), a3 = mul_mi(csub(o[1], o[3]));
      v[1] = cadd(a0, a2); v[3] = cadd(a1, a3); v[5] = csub(a0, a2); v[7] = csub(a1, a3);
    }
  }
}

// R = R1 R2 (6 = 2 x 3, 9 = 3 x 3, 12 = 4 x 3) by one Cooley-Tukey step in registers: with n = R2 n1 + n2 and k = k1 + R1 k2,
//   X[k1 + R1 k2] = sum_n2 W_R^(n2 k1) W_R2^(n2 k2) sum_n1 v[R2 n1 + n2] W_R1^(n1 k1)
// — R2 DFTs of R1 points, the turn by W_R^(n2 k1), R1 DFTs of R2 points.  Half the passes of a 2^a 3^b line (108 = 12 x 9 instead of
// 4 x 3 x 3 x 3): a pass is two workgroup barriers and a trip of every point through LDS whatever its radix.
template <int R1, int R2, int SIGN> UH_D void fft_butterfly_ct(float2 (&v)[R1 * R2]) {
  constexpr int R = R1 * R2;
  // cos, sin of 2 pi m / R for the exponents m = n2 k1 in use (n2 < R2 = 3, k1 < R1)
  constexpr float C6[3] = {1.0f, 0.5f, -0.5f}, S6[3] = {0.0f, 0.86602540378443864676f, 0.86602540378443864676f};
  constexpr float C9[5] = {1.0f, 0.76604444311897803520f, 0.17364817766693034885f, -0.5f, -0.93969262078590838405f};
  constexpr float S9[5] = {0.0f, 0.64278760968653932632f, 0.98480775301220805937f, 0.86602540378443864676f, 0.34202014332566873304f};
  constexpr float C12[7] = {1.0f, 0.86602540378443864676f, 0.5f, 0.0f, -0.5f, -0.86602540378443864676f, -1.0f};
  constexpr float S12[7] = {0.0f, 0.5f, 0.86602540378443864676f, 1.0f, 0.86602540378443864676f, 0.5f, 0.0f};
  float2 t[R2][R1];
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) {
    float2 a[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) a[n1] = v[R2 * n1 + n2];
    fft_butterfly<R1, SIGN>(a);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      const int m = n2 * k1;
      if (m == 0) t[n2][k1] = a[k1];
      else {
        const float c = R == 6 ? C6[m] : (R == 9 ? C9[m] : C12[m]), sn = R == 6 ? S6[m] : (R == 9 ? S9[m] : S12[m]);
        t[n2][k1] = ctw<SIGN>(a[k1], make_float2(c, -sn));
      }
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    float2 b[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) b[n2] = t[n2][k1];
    fft_butterfly<R2, SIGN>(b);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
  }
}

// One Stockham pass of radix R over `nlines` lines of N points; the sub-transforms entering the pass have Ns points (the product of the
// radices already done).  Butterfly j < N / R of a line: inputs at j + r N / R, turned by exp(-2 pi i k r / (Ns R)) with k = j mod Ns,
// outputs at (j - k) R + k + r Ns.  tw holds exp(-2 pi i t / (N twStride)) for t < N twStride.
// STRIDED = false: line l at buf[l * LS + j], consecutive threads take consecutive butterflies of a line;
// STRIDED = true:  element j of line l at buf[l * LS + j * ES] (columns of a row-major plane in LDS: LS = 1, ES = the row stride),
//                  consecutive threads take consecutive LINES (neighbouring LDS words).
template <int R, int SIGN, int MAXB, int NT, bool STRIDED, bool P2>
UH_D void fft_pass(float2 *buf, int LS, int ES, int N, int Ns, int nlines, const float2 *tw, int twStride, int tid) {
  const int per = N / R, total = nlines * per;
  const IDivT<P2> dPer(per), dNs(Ns);
  const IDivT<false> dLines(nlines);
  const int es = STRIDED ? ES : 1, pes = per * es, nes = Ns * es;  // (a compile-time 1 for contiguous lines)
  const int twk = (per / Ns) * twStride;  // index step of exp(-2 pi i k / (Ns R)) in the table: N / (Ns R) entries of the N-point table
  float2 v[MAXB][R];
  int dst[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = tid + q * NT;
    dst[q] = -1;
    if (b < total) {
      int line, j;
      if (STRIDED) { j = dLines.div(b); line = dLines.rem(b, j); }
      else { line = dPer.div(b); j = dPer.rem(b, line); }
      const int k = dNs.mod(j);
      const float2 *p = buf + __mul24(line, LS) + __mul24(j, es);
      const int t1 = __mul24(k, twk);
#pragma unroll
      for (int r = 0; r < R; ++r) v[q][r] = p[r * pes];
#pragma unroll
      for (int r = 1; r < R; ++r) v[q][r] = ctw<SIGN>(v[q][r], tw[t1 * r]);
      fft_butterfly<R, SIGN>(v[q]);
      dst[q] = __mul24(line, LS) + __mul24((j - k) * R + k, es);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXB; ++q)
    if (dst[q] >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) buf[dst[q] + r * nes] = v[q][r];
    }
  __syncthreads();
}

// ---- power-of-two lengths: the passes with shifts and masks (the all-power-of-two grids of C4 / C5 run these) ----
// One Stockham pass of radix R over `nlines` lines of N = 2^LOG2N points at buf[line * LS + j]; the sub-transforms entering the pass
// have 2^LOG2NS points.  tw holds exp(-2 pi i k / NT) for k < NT, NT = N << LOG2TWSHIFT... (twStride = NT / N).
template <int R, int SIGN, int MAXB, int NT>
UH_D void fft_pass_p2(float2 *buf, int LS, int log2N, int log2Ns, int nlines, const float2 *tw, int twStride, int tid) {
  constexpr int LR = R == 8 ? 3 : (R == 4 ? 2 : 1);
  const int N = 1 << log2N, per = N >> LR, Ns = 1 << log2Ns, total = nlines << (log2N - LR);
  float2 v[MAXB][R];
  int dst[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    const int b = tid + q * NT;
    dst[q] = -1;
    if (b < total) {
      const int line = b >> (log2N - LR), j = b & (per - 1), k = j & (Ns - 1);
      const float2 *p = buf + line * LS + j;
      const int t1 = (k << (log2N - log2Ns - LR)) * twStride;  // index of exp(-2 pi i k / (Ns R)) in the table
#pragma unroll
      for (int r = 0; r < R; ++r) v[q][r] = p[r * per];
#pragma unroll
      for (int r = 1; r < R; ++r) v[q][r] = ctw<SIGN>(v[q][r], tw[t1 * r]);
      fft_butterfly<R, SIGN>(v[q]);
      dst[q] = line * LS + ((j - k) << LR) + k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXB; ++q)
    if (dst[q] >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) buf[dst[q] + r * Ns] = v[q][r];
    }
  __syncthreads();
}

// the same pass over lines whose ELEMENTS are `ES` apart and whose first elements are `LS` apart (columns of a row-major plane in LDS:
// LS = 1, ES = the row stride)
template <int R, int SIGN, int MAXB, int NT>
UH_D void fft_pass_strided_p2(float2 *buf, int LS, int ES, int log2N, int log2Ns, int nlines, const float2 *tw, int twStride, int tid) {
  constexpr int LR = R == 8 ? 3 : (R == 4 ? 2 : 1);
  const int N = 1 << log2N, per = N >> LR, Ns = 1 << log2Ns, total = nlines << (log2N - LR);
  float2 v[MAXB][R];
  int dst[MAXB];
#pragma unroll
  for (int q = 0; q < MAXB; ++q) {
    // consecutive threads take consecutive LINES (neighbouring LDS words), not consecutive butterflies of a line
    const int b = tid + q * NT;
    dst[q] = -1;
    if (b < total) {
      const int line = b % nlines, j = b / nlines, k = j & (Ns - 1);
      const float2 *p = buf + line * LS + j * ES;
      const int t1 = (k << (log2N - log2Ns - LR)) * twStride;
#pragma unroll
      for (int r = 0; r < R; ++r) v[q][r] = p[r * per * ES];
#pragma unroll
      for (int r = 1; r < R; ++r) v[q][r] = ctw<SIGN>(v[q][r], tw[t1 * r]);
      fft_butterfly<R, SIGN>(v[q]);
      dst[q] = line * LS + (((j - k) << LR) + k) * ES;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < MAXB; ++q)
    if (dst[q] >= 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) buf[dst[q] + r * Ns * ES] = v[q][r];
    }
  __syncthreads();
}

// N = 2^a 3^b 5^c 7^d 11^e?  (host and device; e[] = the exponents of 2, 3, 5, 7, 11 — the factors of Grid's nextFFTWiseSize3D)
inline __host__ __device__ bool fft_factors(int n, int (&e)[5]) {
  const int p[5] = {2, 3, 5, 7, 11};
  for (int k = 0; k < 5; ++k) e[k] = 0;
  if (n < 1) return false;
  for (int k = 0; k < 5; ++k)
    while (n % p[k] == 0) { n /= p[k]; ++e[k]; }
  return n == 1;
}

// N-point FFTs of `nlines` LDS lines.  MAXB = radix-4 butterflies per thread the caller has room for (nlines N / 4 <= MAXB NT); the
// other radices scale from it (a radix-R pass has nlines N / R butterflies).  The caller has synchronised its writes.
// Pass plan for 2^a: radix 8 where it fits (a pass is two workgroup barriers and a trip of every point through LDS whatever its radix),
// a = 3 n8 + 2 n4 + n2 with the small passes first.
template <int SIGN, int MAXB, int NT, bool STRIDED, bool P2>
UH_D void fft_lds_any(float2 *buf, int LS, int ES, int N, int nlines, const float2 *tw, int twStride, int tid) {
  int e[5] = {0, 0, 0, 0, 0};
  if (P2) e[0] = 31 - __builtin_clz((unsigned)N);
  else fft_factors(N, e);
  const int e2 = e[0], e3 = e[1], e5 = e[2];
  int n8 = e2 / 3, n4 = 0, n2 = 0;
  if (e2 % 3 == 2) n4 = 1;
  else if (e2 % 3 == 1) { if (n8 > 0) { n8 -= 1; n4 = 2; } else n2 = 1; }
  if constexpr (P2) {  // (N >= 4: a two-point axis takes the general instantiation.  The plan loop is written as it was when only powers
    // of two were served — radix-4 passes until the rest is a multiple of three bits, then radix 8: counted loops over n4 and n8 compiled
    // to a fused z pass 3 us slower at C4)
    const int log2N = e2;
    const int m4 = (log2N % 3 == 0) ? 0 : (log2N % 3 == 2 ? 1 : 2);
    int ls = 0;
    for (int a = 0; a < m4; ++a, ls += 2) {
      if constexpr (STRIDED) fft_pass_strided_p2<4, SIGN, MAXB, NT>(buf, LS, ES, log2N, ls, nlines, tw, twStride, tid);
      else fft_pass_p2<4, SIGN, MAXB, NT>(buf, LS, log2N, ls, nlines, tw, twStride, tid);
    }
    for (; ls < log2N; ls += 3) {
      if constexpr (STRIDED) fft_pass_strided_p2<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, ES, log2N, ls, nlines, tw, twStride, tid);
      else fft_pass_p2<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, log2N, ls, nlines, tw, twStride, tid);
    }
  } else {
    // 3^b as radix-9 passes; an odd three left over joins a four (12) or a two (6) where that saves a pass of the 2^a part
    auto passes2 = [](int a) { return (a + 2) / 3; };
    const int n9 = e3 / 2, r3 = e3 & 1;
    int n12 = 0, n6 = 0, n3 = r3, a2 = e2;
    if (r3) {
      const int plain = passes2(e2) + 1;
      if (e2 >= 2 && passes2(e2 - 2) + 1 < plain) { n12 = 1; n3 = 0; a2 = e2 - 2; }
      else if (e2 >= 1 && passes2(e2 - 1) + 1 < plain) { n6 = 1; n3 = 0; a2 = e2 - 1; }
    }
    n8 = a2 / 3; n4 = 0; n2 = 0;
    if (a2 % 3 == 2) n4 = 1;
    else if (a2 % 3 == 1) { if (n8 > 0) { n8 -= 1; n4 = 2; } else n2 = 1; }
    int Ns = 1;
    for (int a = 0; a < n2; ++a, Ns *= 2) fft_pass<2, SIGN, 2 * MAXB, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n4; ++a, Ns *= 4) fft_pass<4, SIGN, MAXB, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n8; ++a, Ns *= 8) fft_pass<8, SIGN, (MAXB + 1) / 2, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n6; ++a, Ns *= 6) fft_pass<6, SIGN, (4 * MAXB + 5) / 6, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n12; ++a, Ns *= 12) fft_pass<12, SIGN, (4 * MAXB + 11) / 12, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n3; ++a, Ns *= 3) fft_pass<3, SIGN, (4 * MAXB + 2) / 3, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < n9; ++a, Ns *= 9) fft_pass<9, SIGN, (4 * MAXB + 8) / 9, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < e5; ++a, Ns *= 5) fft_pass<5, SIGN, (4 * MAXB + 4) / 5, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < e[3]; ++a, Ns *= 7) fft_pass<7, SIGN, (4 * MAXB + 6) / 7, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
    for (int a = 0; a < e[4]; ++a, Ns *= 11) fft_pass<11, SIGN, (4 * MAXB + 10) / 11, NT, STRIDED, false>(buf, LS, ES, N, Ns, nlines, tw, twStride, tid);
  }
}
// the power-of-two plan of fft_lds_any (radix-4 passes until the rest is a multiple of three bits, then radix 8) without its first and / or
// its last pass: a caller that has the lines in registers on the way in (or wants them there on the way out) does that pass itself —
// k_fft_z_fused's first forward pass on the values it loads, its last inverse pass into the values it stores
template <int SIGN, int MAXB, int NT, bool STRIDED = false>
UH_D void fft_lds_p2_inner(float2 *buf, int LS, int log2N, int nlines, const float2 *tw, int twStride, int tid, bool skipFirst, bool skipLast,
                           int ES = 1) {
  const int m4 = (log2N % 3 == 0) ? 0 : (log2N % 3 == 2 ? 1 : 2);
  int ls = 0;
  bool skip = skipFirst;
  for (int a = 0; a < m4; ++a, ls += 2) {
    if (!skip) {
      if constexpr (STRIDED) fft_pass_strided_p2<4, SIGN, MAXB, NT>(buf, LS, ES, log2N, ls, nlines, tw, twStride, tid);
      else fft_pass_p2<4, SIGN, MAXB, NT>(buf, LS, log2N, ls, nlines, tw, twStride, tid);
    }
    skip = false;
  }
  for (; ls + (skipLast ? 3 : 0) < log2N; ls += 3) {
    if (!skip) {
      if constexpr (STRIDED) fft_pass_strided_p2<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, ES, log2N, ls, nlines, tw, twStride, tid);
      else fft_pass_p2<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, log2N, ls, nlines, tw, twStride, tid);
    }
    skip = false;
  }
}
template <int SIGN, int MAXB, int NT, bool P2>
UH_D void fft_lds(float2 *buf, int LS, int N, int nlines, const float2 *tw, int twStride, int tid) {
  fft_lds_any<SIGN, MAXB, NT, false, P2>(buf, LS, 1, N, nlines, tw, twStride, tid);
}
template <int SIGN, int MAXB, int NT, bool P2>
UH_D void fft_lds_strided(float2 *buf, int LS, int ES, int N, int nlines, const float2 *tw, int twStride, int tid) {
  fft_lds_any<SIGN, MAXB, NT, true, P2>(buf, LS, ES, N, nlines, tw, twStride, tid);
}

// rows of nh = nx / 2 complex Z (the FFT of the rows' even / odd samples) -> the nh + 1 entries X_0 .. X_nh of the real rows' spectra:
// X_k = E_k + W^k O_k, E_k = (Z_k + conj Z_{nh-k}) / 2, O_k = (Z_k - conj Z_{nh-k}) / (2i), W = exp(-2 pi i / nx); the pair (k, nh - k)
// by one thread, k = 0 (with the middle entry nh / 2 when nh is even) by another.  tw[k twx] = W^k.
template <int NT, bool P2> UH_D void fft_untangle_r2c(float2 *buf, int LS, int nh, int nrows, const float2 *tw, int twx, int tid) {
  const int half = (nh + 1) / 2;  // k = 0 .. half - 1: pairs (k, nh - k), k >= 1
  const IDivT<P2> dHalf(half);
  for (int i = tid; i < nrows * half; i += NT) {
    const int r = dHalf.div(i), k = dHalf.rem(i, r);
    float2 *row = buf + r * LS;
    if (k == 0) {
      if ((nh & 1) == 0) {
        const float2 m = row[nh / 2];
        row[nh / 2] = make_float2(m.x, -m.y);  // X_{nh/2} = conj Z_{nh/2}
      }
      const float2 z = row[0];
      row[0] = make_float2(z.x + z.y, 0.0f);
      row[nh] = make_float2(z.x - z.y, 0.0f);
    } else {
      const float2 a = row[k], b = row[nh - k];
      const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));   // E_k
      const float2 o = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));  // O_k = (a - conj b) / (2i)
      const float2 wo = ctw<-1>(o, tw[k * twx]);
      row[k] = cadd(e, wo);
      // X_{nh-k} = conj(E_k - W^k O_k)
      row[nh - k] = make_float2(e.x - wo.x, -(e.y - wo.y));
    }
  }
}
// the reverse: X_0 .. X_nh -> Z_0 .. Z_{nh-1}, Z_k = (X_k + conj X_{nh-k}) + i W^{-k} (X_k - conj X_{nh-k}) (unnormalised: twice the above)
template <int NT, bool P2> UH_D void fft_tangle_c2r(float2 *buf, int LS, int nh, int nlines, const float2 *tw, int twx, int tid) {
  const int half = (nh + 1) / 2;
  const IDivT<P2> dHalf(half);
  for (int i = tid; i < nlines * half; i += NT) {
    const int line = dHalf.div(i), k = dHalf.rem(i, line);
    float2 *row = buf + line * LS;
    if (k == 0) {
      const float a = row[0].x, b = row[nh].x;
      row[0] = make_float2(a + b, a - b);
      if ((nh & 1) == 0) {
        const float2 m = row[nh / 2];
        row[nh / 2] = make_float2(2.0f * m.x, -2.0f * m.y);  // Z_{nh/2} = 2 conj X_{nh/2}
      }
    } else {
      const float2 a = row[k], b = row[nh - k];
      const float2 s = make_float2(a.x + b.x, a.y - b.y);   // X_k + conj X_{nh-k}
      const float2 d = make_float2(a.x - b.x, a.y + b.y);   // X_k - conj X_{nh-k}
      const float2 wd = ctw<1>(d, tw[k * twx]);             // W^{-k} d
      const float2 iwd = make_float2(-wd.y, wd.x);
      row[k] = cadd(s, iwd);
      row[nh - k] = make_float2(s.x - iwd.x, -(s.y - iwd.y));  // Z_{nh-k} = conj(s) + i W^{k} conj(d)
    }
  }
}

// ---- rows: R2C in place -------------------------------------------------------------------------------------------------------------
// g: rows of nxpad = nx + 2 floats, `nrows` of them back to back (the three planar component grids are contiguous).
// FOLD (slab decomposition): the first and the last `foldRows` rows also take what the neighbours spread into their halo planes,
// addLo / addHi (foldRows rows each, same row layout), added while the row is loaded — grid + halo, as the separate pass did.
template <bool FOLD, bool P2>
__global__ void __launch_bounds__(kFftThreads) k_fft_x_r2c(float *__restrict__ g, int nxArg, int nrows, int rowsPerBlock,
                                                           const float *__restrict__ addLo, const float *__restrict__ addHi, int foldRows) {
  extern __shared__ float2 lds[];
  const int nx = fft_len<P2>(nxArg);
  const int nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  const IDivT<P2> dNh(nh);
  float2 *tw = lds, *buf = lds + nx;
  const int tid = threadIdx.x, r0 = blockIdx.x * rowsPerBlock, nr = min(rowsPerBlock, nrows - r0);
  fft_twiddles(tw, nx, tid);
  // power-of-two rows without a halo to fold in: the first pass on the values as they arrive from memory (see k_fft_xy_r2c_plane)
  const int log2H = 31 - __builtin_clz((unsigned)nh);
  const bool first8 = log2H % 3 == 0;
  const bool edge = P2 && !FOLD && log2H >= 3 && nr * (nh >> (first8 ? 3 : 2)) <= (first8 ? 1 : 2) * kFftThreads;
  if (edge) {
    if (first8) {
      const int per = nh >> 3;
      if (tid < nr * per) {
        const int j = tid & (per - 1), r = tid >> (log2H - 3);
        const float2 *src = (const float2 *)(g + (size_t)(r0 + r) * nxpad) + j;
        float2 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = src[m * per];
        fft_butterfly<8, -1>(v);
        float2 *dst = buf + __mul24(r, LS) + 8 * j;
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[m] = v[m];
      }
    } else {
      const int per = nh >> 2;
      float2 v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = tid + q * kFftThreads;
        if (b < nr * per) {
          const float2 *src = (const float2 *)(g + (size_t)(r0 + (b >> (log2H - 2))) * nxpad) + (b & (per - 1));
#pragma unroll
          for (int m = 0; m < 4; ++m) v[q][m] = src[m * per];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = tid + q * kFftThreads;
        if (b < nr * per) {
          fft_butterfly<4, -1>(v[q]);
          float2 *dst = buf + __mul24(b >> (log2H - 2), LS) + 4 * (b & (per - 1));
#pragma unroll
          for (int m = 0; m < 4; ++m) dst[m] = v[q][m];
        }
      }
    }
    __syncthreads();
    fft_lds_p2_inner<-1, 2, kFftThreads>(buf, LS, log2H, nr, tw, 2, tid, true, false);
  } else {
  staged_copy<8, float2>(tid, nr * nh, kFftThreads,
      [&](int i) {
        const int r = dNh.div(i), j = dNh.rem(i, r);
        return *(const float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * j);
      },
      [&](int i, float2 v) { const int r = dNh.div(i); buf[__mul24(r, LS) + dNh.rem(i, r)] = v; });
  if (FOLD) {  // (the rows that take a neighbour's halo plane: a second pass over those rows only)
    __syncthreads();
    staged_copy<8, float2>(tid, nr * nh, kFftThreads,
        [&](int i) {
          const int r = dNh.div(i), j = dNh.rem(i, r), row = r0 + r;
          float2 a = make_float2(0.f, 0.f);
          if (row < foldRows) a = *(const float2 *)(addLo + (size_t)row * nxpad + 2 * j);
          else if (row >= nrows - foldRows) a = *(const float2 *)(addHi + (size_t)(row - (nrows - foldRows)) * nxpad + 2 * j);
          return a;
        },
        [&](int i, float2 a) {
          const int r = dNh.div(i);
          float2 &v = buf[__mul24(r, LS) + dNh.rem(i, r)];
          v = make_float2(v.x + a.x, v.y + a.y);
        });
  }
  __syncthreads();
  fft_lds<-1, 2, kFftThreads, P2>(buf, LS, nh, nr, tw, 2, tid);
  }
  fft_untangle_r2c<kFftThreads, P2>(buf, LS, nh, nr, tw, 1, tid);
  __syncthreads();
  for (int i = tid; i < nr * nh; i += kFftThreads) {
    const int r = dNh.div(i), k = dNh.rem(i, r);
    *(float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * k) = buf[r * LS + k];
    if (k == 0) *(float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * nh) = buf[r * LS + nh];
  }
}

// ---- a whole (component, z) plane: x then y, one pass over the grid instead of two -----------------------------------------------------------
// The rows' R2C and the columns' transform of one plane both fit a workgroup's LDS when ny (nx / 2 + 1) complex do (66.5 KB at 128 x 128 of
// gfx950's 160 KB per CU): rows in, ny FFTs of nx / 2 points + untangling, nx / 2 + 1 FFTs of ny points down the columns of the same array
// (element stride = the row stride), rows out.  3 nz workgroups of 1024 threads; a launch asks for its dynamic LDS beyond 64 KB through
// hipFuncSetAttribute.  Replaces k_fft_x_r2c + k_fft_lines<-1> where the plane fits (fcm_plane_fft_usable).
// LDS: exp(-2 pi i k / nx), k < nx | exp(-2 pi i k / ny), k < ny | the plane
constexpr int kPlaneThreads = 1024;
// DENSE: held to 64 vector registers so that TWO workgroups share a CU where the plane's LDS lets them (<= 80 KB): the mixed-radix
// instantiation wants 91 registers, which leaves one workgroup of 16 waves per CU and the 3 nz planes take two rounds of the chip — at
// 108^3 (the PSE far field: 324 planes, 47.5 KB each) 25 spilled registers cost less than the second round: solve 0.1848 -> 0.1785 ms.
// (The power-of-two instantiation needs 62 registers either way.)
template <bool P2, bool DENSE = false>
__global__ void __launch_bounds__(kPlaneThreads) __attribute__((amdgpu_waves_per_eu(DENSE ? 8 : 1, 8)))
k_fft_xy_r2c_plane(float *__restrict__ g, int nxArg, int nyArg) {
  extern __shared__ float2 lds[];
  const int nx = fft_len<P2>(nxArg), ny = fft_len<P2>(nyArg);
  const int nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  const IDivT<P2> dNh(nh);
  const IDivT<false> dLS(LS);
  float2 *twx = lds, *twy = lds + nx, *buf = lds + nx + ny;
  const int tid = threadIdx.x;
  float *plane = g + (size_t)blockIdx.x * ny * nxpad;   // (the three planar component grids are contiguous: plane index = c nz + z)
  fft_twiddles<kPlaneThreads>(twx, nx, tid);
  fft_twiddles<kPlaneThreads>(twy, ny, tid);
  // power-of-two planes: the rows' first pass on the values as they arrive from memory, the columns' last pass (radix 8) on the values as
  // they leave (see k_fft_z_fused): two trips of the plane through LDS and two barriers fewer, the same bits
  const int log2H = 31 - __builtin_clz((unsigned)nh), log2Y = 31 - __builtin_clz((unsigned)ny);
  // (the columns' plan must END with a radix-8 pass: every power of two from 8 up except 16 = 4 x 4)
  const bool edges = P2 && log2H >= 3 && log2Y >= 3 && log2Y != 4 && (ny >> 3) * LS <= 2 * kPlaneThreads &&
                     ny * (nh >> (log2H % 3 == 0 ? 3 : 2)) <= (log2H % 3 == 0 ? 1 : 2) * kPlaneThreads;
  if (edges) {
    if (log2H % 3 == 0) {   // the rows' plan starts with a radix-8 pass: one butterfly per thread
      const int per = nh >> 3, b = tid;
      if (b < ny * per) {
        const int j = b & (per - 1), r = b >> (log2H - 3);
        const float2 *src = (const float2 *)(plane + (size_t)r * nxpad) + j;
        float2 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = src[m * per];
        fft_butterfly<8, -1>(v);
        float2 *dst = buf + __mul24(r, LS) + 8 * j;
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[m] = v[m];
      }
    } else {                // with a radix-4 pass: up to two butterflies per thread
      const int per = nh >> 2;
      float2 v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = tid + q * kPlaneThreads;
        if (b < ny * per) {
          const int j = b & (per - 1), r = b >> (log2H - 2);
          const float2 *src = (const float2 *)(plane + (size_t)r * nxpad) + j;
#pragma unroll
          for (int m = 0; m < 4; ++m) v[q][m] = src[m * per];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = tid + q * kPlaneThreads;
        if (b < ny * per) {
          const int j = b & (per - 1), r = b >> (log2H - 2);
          fft_butterfly<4, -1>(v[q]);
          float2 *dst = buf + __mul24(r, LS) + 4 * j;
#pragma unroll
          for (int m = 0; m < 4; ++m) dst[m] = v[q][m];
        }
      }
    }
    __syncthreads();
    fft_lds_p2_inner<-1, 2, kPlaneThreads>(buf, LS, log2H, ny, twx, 2, tid, true, false);
    fft_untangle_r2c<kPlaneThreads, P2>(buf, LS, nh, ny, twx, 1, tid);
    __syncthreads();
    fft_lds_p2_inner<-1, 3, kPlaneThreads, true>(buf, 1, log2Y, LS, twy, 1, tid, false, true, LS);
    const int per = ny >> 3, total = per * LS;   // the columns' last pass: sub-transforms of ny / 8 points, outputs at rows j + m ny / 8
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int b = tid + q * kPlaneThreads;
      if (b < total) {
        const int j = dLS.div(b), k = dLS.rem(b, j);
        const float2 *p = buf + k + __mul24(j, LS);
        float2 v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = p[__mul24(m * per, LS)];
#pragma unroll
        for (int m = 1; m < 8; ++m) v[m] = ctw<-1>(v[m], twy[j * m]);
        fft_butterfly<8, -1>(v);
        float2 *dst = (float2 *)(plane + (size_t)j * nxpad) + k;
#pragma unroll
        for (int m = 0; m < 8; ++m) *(float2 *)((float *)dst + (size_t)(m * per) * nxpad) = v[m];
      }
    }
    return;
  }
  staged_copy<8, float2>(tid, ny * nh, kPlaneThreads,
      [&](int i) { const int r = dNh.div(i); return *(const float2 *)(plane + (size_t)r * nxpad + 2 * dNh.rem(i, r)); },
      [&](int i, float2 v) { const int r = dNh.div(i); buf[__mul24(r, LS) + dNh.rem(i, r)] = v; });
  __syncthreads();
  // rows: ny complex FFTs of nh points (exp(-2 pi i k / nh) = twx[2 k])
  fft_lds<-1, 2, kPlaneThreads, P2>(buf, LS, nh, ny, twx, 2, tid);
  fft_untangle_r2c<kPlaneThreads, P2>(buf, LS, nh, ny, twx, 1, tid);
  __syncthreads();
  // columns: nh + 1 FFTs of ny points, element stride LS
  fft_lds_strided<-1, 3, kPlaneThreads, P2>(buf, 1, LS, ny, nh + 1, twy, 1, tid);
  for (int i = tid; i < ny * LS; i += kPlaneThreads) {
    const int r = dLS.div(i), k = dLS.rem(i, r);
    *(float2 *)(plane + (size_t)r * nxpad + 2 * k) = buf[r * LS + k];
  }
}

// ---- strided lines (the y transform) ---------------------------------------------------------------------------------------------------
// group = one (component, z) plane of ny x nkx complex; tile = 16 consecutive kx (the last tile of a plane is narrower); element j of
// line l at group * ny * nkx + j * nkx + kx0 + l.  Thread (l = tid & 15, jg = tid >> 4) moves elements j = jg + 16 it of line l: 128-byte
// segments, no integer division.
template <int SIGN, int MAXB, int NT, bool P2>
__global__ void __launch_bounds__(NT) k_fft_lines(float2 *__restrict__ g, int nArg, int nkx, int tilesPerGroup) {
  extern __shared__ float2 lds[];
  const int n = fft_len<P2>(nArg);
  const int LS = n + 1;
  float2 *tw = lds, *buf = lds + n;
  // neighbouring tiles of a plane share the 128-byte lines their 16-complex segments straddle (a row is 8 (nx / 2 + 1) bytes: never a
  // multiple of 128): dealt round-robin to the XCDs every such line was fetched from HBM twice (55 MB per pass of a 25.6 MB grid at C4)
  const int blk = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, group = blk / tilesPerGroup, tile = blk - group * tilesPerGroup;
  const int kx0 = tile * 16, nl = min(16, nkx - kx0), l = tid & 15, jg = tid >> 4;
  float2 *base = g + (size_t)group * n * nkx + kx0 + l;
  fft_twiddles<NT>(tw, n, tid);
  // full tiles of power-of-two lines whose plan is 4 .. 8 (128, 256, 32): the first pass on the values as they arrive from memory, the last
  // on the values as they leave (see k_fft_z_fused: two trips through LDS and two barriers fewer, the same bits)
  constexpr int QF = MAXB, QL = (MAXB + 1) / 2;   // (MAXB = radix-4 butterflies per thread the launch has room for)
  const int log2N = 31 - __builtin_clz((unsigned)n);
  const bool edges = P2 && nl == 16 && log2N % 3 != 0 && log2N >= 5 && 16 * (n >> 2) <= QF * NT && 16 * (n >> 3) <= QL * NT;
  if (edges) {
    float2 *tile = g + (size_t)group * n * nkx + kx0;
    {
      const int per = n >> 2, total = 16 * per;
      float2 v[QF][4];
#pragma unroll
      for (int q = 0; q < QF; ++q) {
        const int b = tid + q * NT;
        if (b < total) {
          const float2 *src = tile + (b & 15) + (size_t)(b >> 4) * nkx;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[q][r] = src[(size_t)(r * per) * nkx];
        }
      }
#pragma unroll
      for (int q = 0; q < QF; ++q) {
        const int b = tid + q * NT;
        if (b < total) {
          fft_butterfly<4, SIGN>(v[q]);
          float2 *dst = buf + (b & 15) * LS + 4 * (b >> 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[r] = v[q][r];
        }
      }
    }
    __syncthreads();
    fft_lds_p2_inner<SIGN, MAXB, NT>(buf, LS, log2N, 16, tw, 1, tid, true, true);
    const int per = n >> 3, total = 16 * per;
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const int b = tid + q * NT;
      if (b < total) {
        const int line = b & 15, j = b >> 4;
        const float2 *p = buf + line * LS + j;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = p[r * per];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = ctw<SIGN>(v[r], tw[j * r]);
        fft_butterfly<8, SIGN>(v);
        float2 *dst = tile + line + (size_t)j * nkx;
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[(size_t)(r * per) * nkx] = v[r];
      }
    }
    return;
  }
  if (l < nl)
    staged_copy<16, float2>(jg, n, NT / 16, [&](int j) { return base[(size_t)j * nkx]; }, [&](int j, float2 v) { buf[l * LS + j] = v; });
  __syncthreads();
  fft_lds<SIGN, MAXB, NT, P2>(buf, LS, n, nl, tw, 1, tid);
  if (l < nl)
    for (int j = jg; j < n; j += NT / 16) base[(size_t)j * nkx] = buf[l * LS + j];
}

// ---- rows back: C2R of the three components ---------------------------------------------------------------------------------------------
// gc: planar complex grids (component stride planeC); output either the planar real grids in place (inter == nullptr) or the gather's
// interleaved float4 grid inter[(z ny + y) nx + x] = (vx, vy, vz, 0).
// Row r = (z, y) of component c starts at c compStride + z zStride + y nxpad floats (single GPU: compStride = one component grid,
// zStride = ny nxpad; slab window [z][c][y][x]: compStride = ny nxpad, zStride = 3 ny nxpad).
// wrapRows > 0 (slab window of a rank that is its own neighbour, world size 1): the first wrapRows rows are also stored wrapShift float4
// further on and the last wrapRows rows wrapShift earlier — the halo planes the gather reads, which with neighbours arrive by message.
// (Measured and not kept: the tangling on the values as they arrive from memory — a pair's two entries loaded together, tangled in registers,
// stored once, one trip through LDS and one barrier fewer: 14.8 -> 14.1 us at C4, 96.4 against 96.5 us at C5: 0.4 % of the C4 step for
// a second copy of the tangling arithmetic.)
template <bool P2>
__global__ void __launch_bounds__(kFftThreads) k_fft_x_c2r(float *__restrict__ g, size_t compStride, size_t zStride, int nyArg, int nxArg,
                                                           int nrows, int rowsPerBlock, float4 *__restrict__ inter, int wrapRows,
                                                           size_t wrapShift) {
  extern __shared__ float2 lds[];
  const int nx = fft_len<P2>(nxArg), ny = fft_len<P2>(nyArg);
  const int nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  const IDivT<P2> dNh(nh), dNy(ny);
  float2 *tw = lds, *buf = lds + nx;
  const int tid = threadIdx.x, r0 = blockIdx.x * rowsPerBlock, nr = min(rowsPerBlock, nrows - r0);  // rows x 3 components
  fft_twiddles(tw, nx, tid);
  auto rowOf = [&](int c, int r) {
    const int gr = r0 + r, z = dNy.div(gr);
    return g + (size_t)c * compStride + (size_t)z * zStride + (size_t)dNy.rem(gr, z) * nxpad;
  };
  {  // element e = c (nr nh) + r nh + k of the 3 nr rows; the rows' last (Nyquist) entries in a second, short round
    const int per = nr * nh;
    const IDivT<false> dPer(per);
    staged_copy<12, float2>(tid, 3 * per, kFftThreads,
        [&](int e) {
          const int c = dPer.div(e), i = dPer.rem(e, c), r = dNh.div(i);
          return *(const float2 *)(rowOf(c, r) + 2 * dNh.rem(i, r));
        },
        [&](int e, float2 v) {
          const int c = dPer.div(e), i = dPer.rem(e, c), r = dNh.div(i);
          buf[__mul24(c * nr + r, LS) + dNh.rem(i, r)] = v;
        });
    if (tid < 3 * nr) {
      const int c = tid / nr, r = tid - c * nr;
      buf[(c * nr + r) * LS + nh] = *(const float2 *)(rowOf(c, r) + 2 * nh);
    }
  }
  __syncthreads();
  fft_tangle_c2r<kFftThreads, P2>(buf, LS, nh, 3 * nr, tw, 1, tid);
  __syncthreads();
  fft_lds<1, 2, kFftThreads, P2>(buf, LS, nh, 3 * nr, tw, 2, tid);
  if (inter && wrapRows < 0) {  // (wrapRows = -1: the gather's grid with 12 bytes per node — grids the Infinity Cache does not hold, fcm.hip)
    for (int i = tid; i < nr * nh; i += kFftThreads) {
      const int r = dNh.div(i), j = dNh.rem(i, r);
      const float2 vx = buf[r * LS + j], vy = buf[(nr + r) * LS + j], vz = buf[(2 * nr + r) * LS + j];
      PackedNode *o = (PackedNode *)inter + ((size_t)(r0 + r) * nx + 2 * j);
      o[0] = PackedNode{vx.x, vy.x, vz.x};
      o[1] = PackedNode{vx.y, vy.y, vz.y};
    }
  } else if (inter) {
    for (int i = tid; i < nr * nh; i += kFftThreads) {
      const int r = dNh.div(i), j = dNh.rem(i, r);
      const float2 vx = buf[r * LS + j], vy = buf[(nr + r) * LS + j], vz = buf[(2 * nr + r) * LS + j];
      float4 *o = inter + (size_t)(r0 + r) * nx + 2 * j;
      const float4 a = make_float4(vx.x, vy.x, vz.x, 0.0f), b = make_float4(vx.y, vy.y, vz.y, 0.0f);
      o[0] = a;
      o[1] = b;
      if (r0 + r < wrapRows) { o[wrapShift] = a; o[wrapShift + 1] = b; }
      if (r0 + r >= nrows - wrapRows) { (o - wrapShift)[0] = a; (o - wrapShift)[1] = b; }
    }
  } else {
    for (int c = 0; c < 3; ++c)
      for (int i = tid; i < nr * nh; i += kFftThreads) {
        const int r = dNh.div(i), j = dNh.rem(i, r);
        *(float2 *)(rowOf(c, r) + 2 * j) = buf[(c * nr + r) * LS + j];
      }
  }
}

