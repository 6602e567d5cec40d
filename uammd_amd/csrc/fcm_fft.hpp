// Batched 3-D real FFT for the FCM solver on power-of-two grids, written for gfx950: five passes over the grid instead of the eight
// of rocFFT (three per transform) + the Fourier-space kernel + the interleaving copy, with the z transform, the Stokes / noise
// operator and the inverse z transform fused into one kernel that holds a tile of z-lines in LDS.
//
// Replaces, for nx, ny, nz in {16 ... 512} powers of two (anything else keeps rocFFT):
//   cufftExecR2C / cufftExecC2R (batched 3-D)                      Integrator/BDHI/FCM/FCM_impl.cuh:399-411, :544-581
//   forceFourier2Vel + fourierBrownianNoise between them           FCM_impl.cuh:375-397, :437-512
// Conventions are cuFFT's: forward exp(-i...), inverse exp(+i...), both unnormalised (1/N lives in the Stokes operator), the
// imaginary parts of the kx = 0 and kx = nx/2 self-conjugate inputs of the C2R are ignored.
//
//   k_fft_x_r2c        rows: nx reals -> nx/2 + 1 complex in place (one complex FFT of nx/2 points + untangling), 16 rows per workgroup
//   k_fft_lines        y: strided lines of a z-plane, a tile of <= 16 consecutive kx per workgroup (contiguous 8 tl-byte segments)
//   k_fft_z_fused      z: a tile of consecutive (ky, kx) nodes x all nz x the three components in LDS: forward, operator, inverse
//   k_fft_x_c2r        rows back: three components of 16 rows -> the planar real grids or the gather's interleaved float4 grid
// All FFTs are Stockham autosort (radix 4, one radix-2 pass for odd log2) on LDS lines; a pass stages its butterflies in registers
// (read all, barrier, write all, barrier), twiddles from a table in LDS.
#pragma once
// (included by fcm.hip INSIDE namespace uammd_hip, after the Fourier-space operator it fuses)

constexpr int kFftThreads = 256;
constexpr int kFftMaxLog2 = 9;  // 512

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

// the R-point DFT of v (SIGN < 0: forward, exp(-2 pi i r m / R)), in place, natural order out
template <int R, int SIGN> UH_D void fft_butterfly(float2 (&v)[R]) {
  auto mul_mi = [](float2 d) { return SIGN < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x); };  // -i d (forward), +i d (inverse)
  if (R == 2) {
    const float2 a = v[0], c = v[1];
    v[0] = cadd(a, c);
    v[1] = csub(a, c);
  } else if (R == 4) {
    const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(a0, a2);
    v[1] = cadd(a1, a3);
    v[2] = csub(a0, a2);
    v[3] = csub(a1, a3);
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

// One Stockham pass of radix R over `nlines` lines of N = 2^LOG2N points at buf[line * LS + j]; the sub-transforms entering the pass
// have 2^LOG2NS points.  tw holds exp(-2 pi i k / NT) for k < NT, NT = N << LOG2TWSHIFT... (twStride = NT / N).
template <int R, int SIGN, int MAXB, int NT>
UH_D void fft_pass(float2 *buf, int LS, int log2N, int log2Ns, int nlines, const float2 *tw, int twStride, int tid) {
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
UH_D void fft_pass_strided(float2 *buf, int LS, int ES, int log2N, int log2Ns, int nlines, const float2 *tw, int twStride, int tid) {
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
// Pass plan: radix 8 where it fits (a pass is two workgroup barriers and a trip of every point through LDS whatever its radix: 64 points
// are two passes instead of three, 128 and 256 three instead of four): log2 N = 3 a + 2 b with b <= 2, the radix-4 passes first.
// MAXB = radix-4 butterflies per thread; a radix-8 pass has half as many.
template <int SIGN, int MAXB, int NT>
UH_D void fft_lds_strided(float2 *buf, int LS, int ES, int log2N, int nlines, const float2 *tw, int twStride, int tid) {
  const int n4 = (log2N % 3 == 0) ? 0 : (log2N % 3 == 2 ? 1 : 2);  // 4 = 2 + 2, 5 = 2 + 3, 7 = 2 + 2 + 3, 8 = 2 + 3 + 3
  int s = 0;
  for (int a = 0; a < n4; ++a, s += 2) fft_pass_strided<4, SIGN, MAXB, NT>(buf, LS, ES, log2N, s, nlines, tw, twStride, tid);
  for (; s < log2N; s += 3) fft_pass_strided<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, ES, log2N, s, nlines, tw, twStride, tid);
}

// N-point FFTs of `nlines` LDS lines (N = 2^log2N <= 512, nlines * N / 4 <= MAXB * 256).  The caller has synchronised its writes.
template <int SIGN, int MAXB, int NT = kFftThreads>
UH_D void fft_lds(float2 *buf, int LS, int log2N, int nlines, const float2 *tw, int twStride, int tid) {
  const int n4 = (log2N % 3 == 0) ? 0 : (log2N % 3 == 2 ? 1 : 2);
  int s = 0;
  for (int a = 0; a < n4; ++a, s += 2) fft_pass<4, SIGN, MAXB, NT>(buf, LS, log2N, s, nlines, tw, twStride, tid);
  for (; s < log2N; s += 3) fft_pass<8, SIGN, (MAXB + 1) / 2, NT>(buf, LS, log2N, s, nlines, tw, twStride, tid);
}

// ---- rows: R2C in place -------------------------------------------------------------------------------------------------------------
// g: rows of nxpad = nx + 2 floats, `nrows` of them back to back (the three planar component grids are contiguous).
// FOLD (slab decomposition): the first and the last `foldRows` rows also take what the neighbours spread into their halo planes,
// addLo / addHi (foldRows rows each, same row layout), added while the row is loaded — grid + halo, as the separate pass did.
template <bool FOLD>
__global__ void __launch_bounds__(kFftThreads) k_fft_x_r2c(float *__restrict__ g, int log2nx, int nrows, int rowsPerBlock,
                                                           const float *__restrict__ addLo, const float *__restrict__ addHi, int foldRows) {
  extern __shared__ float2 lds[];
  const int nx = 1 << log2nx, nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  float2 *tw = lds, *buf = lds + nx;
  const int tid = threadIdx.x, r0 = blockIdx.x * rowsPerBlock, nr = min(rowsPerBlock, nrows - r0);
  fft_twiddles(tw, nx, tid);
  staged_copy<8, float2>(tid, nr * nh, kFftThreads,
      [&](int i) {
        const int r = i >> (log2nx - 1), j = i & (nh - 1);
        return *(const float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * j);
      },
      [&](int i, float2 v) { buf[(i >> (log2nx - 1)) * LS + (i & (nh - 1))] = v; });
  if (FOLD) {  // (the rows that take a neighbour's halo plane: a second pass over those rows only)
    __syncthreads();
    staged_copy<8, float2>(tid, nr * nh, kFftThreads,
        [&](int i) {
          const int r = i >> (log2nx - 1), j = i & (nh - 1), row = r0 + r;
          float2 a = make_float2(0.f, 0.f);
          if (row < foldRows) a = *(const float2 *)(addLo + (size_t)row * nxpad + 2 * j);
          else if (row >= nrows - foldRows) a = *(const float2 *)(addHi + (size_t)(row - (nrows - foldRows)) * nxpad + 2 * j);
          return a;
        },
        [&](int i, float2 a) {
          float2 &v = buf[(i >> (log2nx - 1)) * LS + (i & (nh - 1))];
          v = make_float2(v.x + a.x, v.y + a.y);
        });
  }
  __syncthreads();
  fft_lds<-1, 2>(buf, LS, log2nx - 1, nr, tw, 2, tid);
  // untangle: X_k = E_k + W^k O_k, E_k = (Z_k + conj Z_{nh-k}) / 2, O_k = (Z_k - conj Z_{nh-k}) / (2i), k = 0 .. nh (Z_nh = Z_0)
  // (k runs over 0 .. nh/2: nh/2 values through the bit mask + the middle one, k = nh/2, done by the threads that draw k = 0)
  for (int i = tid; i < nr * (nh / 2); i += kFftThreads) {
    const int r = i >> (log2nx - 2), k = i & (nh / 2 - 1);
    float2 *row = buf + r * LS;
    if (k == 0) {
      const float2 m = row[nh / 2];
      row[nh / 2] = make_float2(m.x, -m.y);  // X_{nh/2} = conj Z_{nh/2}
      const float2 z = row[0];
      row[0] = make_float2(z.x + z.y, 0.0f);
      row[nh] = make_float2(z.x - z.y, 0.0f);
    } else {
      const float2 a = row[k], b = row[nh - k];
      const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));   // E_k
      const float2 o = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));  // O_k = (a - conj b) / (2i)
      const float2 wo = ctw<-1>(o, tw[k]);
      row[k] = cadd(e, wo);
      // X_{nh-k} = conj(E_k - W^k O_k)
      row[nh - k] = make_float2(e.x - wo.x, -(e.y - wo.y));
    }
  }
  __syncthreads();
  for (int i = tid; i < nr * nh; i += kFftThreads) {
    const int r = i >> (log2nx - 1), k = i & (nh - 1);
    *(float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * k) = buf[r * LS + k];
    if (k == 0) *(float2 *)(g + (size_t)(r0 + r) * nxpad + 2 * nh) = buf[r * LS + nh];
  }
}

// ---- a whole (component, z) plane: x then y, one pass over the grid instead of two -----------------------------------------------------------
// The rows' R2C and the columns' transform of one plane both fit a workgroup's LDS when ny (nx / 2 + 1) complex do (66.5 KB at 128 x 128 of
// gfx950's 160 KB per CU): rows in, ny FFTs of nx / 2 points + untangling, nx / 2 + 1 FFTs of ny points down the columns of the same array
// (element stride = the row stride), rows out.  3 nz workgroups of 1024 threads; a launch asks for its dynamic LDS beyond 64 KB through
// hipFuncSetAttribute.  Replaces k_fft_x_r2c + k_fft_lines<-1> where the plane fits (fcm_plane_fft_usable).
constexpr int kPlaneThreads = 1024;
__global__ void __launch_bounds__(kPlaneThreads) k_fft_xy_r2c_plane(float *__restrict__ g, int log2nx, int log2ny) {
  extern __shared__ float2 lds[];
  const int nx = 1 << log2nx, ny = 1 << log2ny, nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  const int ntw = nx > ny ? nx : ny;           // one table exp(-2 pi i k / ntw) serves both transforms (both sizes divide it)
  float2 *tw = lds, *buf = lds + ntw;
  const int tid = threadIdx.x;
  float *plane = g + (size_t)blockIdx.x * ny * nxpad;   // (the three planar component grids are contiguous: plane index = c nz + z)
  fft_twiddles<kPlaneThreads>(tw, ntw, tid);
  staged_copy<8, float2>(tid, ny * nh, kPlaneThreads,
      [&](int i) { return *(const float2 *)(plane + (size_t)(i >> (log2nx - 1)) * nxpad + 2 * (i & (nh - 1))); },
      [&](int i, float2 v) { buf[(i >> (log2nx - 1)) * LS + (i & (nh - 1))] = v; });
  __syncthreads();
  // rows: ny complex FFTs of nh points (table stride: exp(-2 pi i k / nh) = tw[k ntw / nh])
  fft_lds<-1, 2, kPlaneThreads>(buf, LS, log2nx - 1, ny, tw, ntw / nh, tid);
  const int twx = ntw / nx;
  for (int i = tid; i < ny * (nh / 2); i += kPlaneThreads) {   // untangle (as k_fft_x_r2c)
    const int r = i >> (log2nx - 2), k = i & (nh / 2 - 1);
    float2 *row = buf + r * LS;
    if (k == 0) {
      const float2 m = row[nh / 2];
      row[nh / 2] = make_float2(m.x, -m.y);
      const float2 z = row[0];
      row[0] = make_float2(z.x + z.y, 0.0f);
      row[nh] = make_float2(z.x - z.y, 0.0f);
    } else {
      const float2 a = row[k], b = row[nh - k];
      const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
      const float2 o = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));
      const float2 wo = ctw<-1>(o, tw[k * twx]);
      row[k] = cadd(e, wo);
      row[nh - k] = make_float2(e.x - wo.x, -(e.y - wo.y));
    }
  }
  __syncthreads();
  // columns: nh + 1 FFTs of ny points, element stride LS
  fft_lds_strided<-1, 3, kPlaneThreads>(buf, 1, LS, log2ny, nh + 1, tw, ntw / ny, tid);
  for (int i = tid; i < ny * LS; i += kPlaneThreads) {
    const int r = i / LS, k = i - r * LS;
    *(float2 *)(plane + (size_t)r * nxpad + 2 * k) = buf[r * LS + k];
  }
}

// ---- strided lines (the y transform) ---------------------------------------------------------------------------------------------------
// group = one (component, z) plane of ny x nkx complex; tile = 16 consecutive kx (the last tile of a plane is narrower); element j of
// line l at group * ny * nkx + j * nkx + kx0 + l.  Thread (l = tid & 15, jg = tid >> 4) moves elements j = jg + 16 it of line l: 128-byte
// segments, no integer division.
template <int SIGN, int MAXB, int NT>
__global__ void __launch_bounds__(NT) k_fft_lines(float2 *__restrict__ g, int log2n, int nkx, int tilesPerGroup) {
  extern __shared__ float2 lds[];
  const int n = 1 << log2n, LS = n + 1;
  float2 *tw = lds, *buf = lds + n;
  // neighbouring tiles of a plane share the 128-byte lines their 16-complex segments straddle (a row is 8 (nx / 2 + 1) bytes: never a
  // multiple of 128): dealt round-robin to the XCDs every such line was fetched from HBM twice (55 MB per pass of a 25.6 MB grid at C4)
  const int blk = (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, group = blk / tilesPerGroup, tile = blk - group * tilesPerGroup;
  const int kx0 = tile * 16, nl = min(16, nkx - kx0), l = tid & 15, jg = tid >> 4;
  float2 *base = g + (size_t)group * n * nkx + kx0 + l;
  fft_twiddles<NT>(tw, n, tid);
  if (l < nl)
    staged_copy<16, float2>(jg, n, NT / 16, [&](int j) { return base[(size_t)j * nkx]; }, [&](int j, float2 v) { buf[l * LS + j] = v; });
  __syncthreads();
  fft_lds<SIGN, MAXB, NT>(buf, LS, log2n, nl, tw, 1, tid);
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
__global__ void __launch_bounds__(kFftThreads) k_fft_x_c2r(float *__restrict__ g, size_t compStride, size_t zStride, int log2ny, int log2nx,
                                                           int nrows, int rowsPerBlock, float4 *__restrict__ inter, int wrapRows,
                                                           size_t wrapShift) {
  extern __shared__ float2 lds[];
  const int nx = 1 << log2nx, nh = nx >> 1, LS = nh + 1, nxpad = nx + 2;
  float2 *tw = lds, *buf = lds + nx;
  const int tid = threadIdx.x, r0 = blockIdx.x * rowsPerBlock, nr = min(rowsPerBlock, nrows - r0);  // rows x 3 components
  fft_twiddles(tw, nx, tid);
  {  // element e = c (nr nh) + r nh + k of the 3 nr rows; the rows' last (Nyquist) entries in a second, short round
    const int per = nr * nh;
    auto rowOf = [&](int c, int r) {
      const int gr = r0 + r;
      return g + (size_t)c * compStride + (size_t)(gr >> log2ny) * zStride + (size_t)(gr & ((1 << log2ny) - 1)) * nxpad;
    };
    staged_copy<12, float2>(tid, 3 * per, kFftThreads,
        [&](int e) {
          const int c = e / per, i = e - c * per;
          return *(const float2 *)(rowOf(c, i >> (log2nx - 1)) + 2 * (i & (nh - 1)));
        },
        [&](int e, float2 v) {
          const int c = e / per, i = e - c * per;
          buf[(c * nr + (i >> (log2nx - 1))) * LS + (i & (nh - 1))] = v;
        });
    if (tid < 3 * nr) {
      const int c = tid / nr, r = tid - c * nr;
      buf[(c * nr + r) * LS + nh] = *(const float2 *)(rowOf(c, r) + 2 * nh);
    }
  }
  __syncthreads();
  // Z_k = (X_k + conj X_{nh-k}) + i W^{-k} (X_k - conj X_{nh-k}), k = 0 .. nh - 1
  for (int i = tid; i < 3 * nr * (nh / 2); i += kFftThreads) {
    const int line = i >> (log2nx - 2), k = i & (nh / 2 - 1);
    float2 *row = buf + line * LS;
    if (k == 0) {
      const float a = row[0].x, b = row[nh].x;
      row[0] = make_float2(a + b, a - b);
      const float2 m = row[nh / 2];
      row[nh / 2] = make_float2(2.0f * m.x, -2.0f * m.y);  // Z_{nh/2} = 2 conj X_{nh/2}
    } else {
      const float2 a = row[k], b = row[nh - k];
      const float2 s = make_float2(a.x + b.x, a.y - b.y);   // X_k + conj X_{nh-k}
      const float2 d = make_float2(a.x - b.x, a.y + b.y);   // X_k - conj X_{nh-k}
      const float2 wd = ctw<1>(d, tw[k]);                   // W^{-k} d
      const float2 iwd = make_float2(-wd.y, wd.x);
      row[k] = cadd(s, iwd);
      row[nh - k] = make_float2(s.x - iwd.x, -(s.y - iwd.y));  // Z_{nh-k} = conj(s) + i W^{k} conj(d)
    }
  }
  __syncthreads();
  fft_lds<1, 2>(buf, LS, log2nx - 1, 3 * nr, tw, 2, tid);
  if (inter) {
    for (int i = tid; i < nr * nh; i += kFftThreads) {
      const int r = i >> (log2nx - 1), j = i & (nh - 1);
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
        const int r = i >> (log2nx - 1), j = i & (nh - 1);
        const int gr = r0 + r;
        *(float2 *)(g + (size_t)c * compStride + (size_t)(gr >> log2ny) * zStride + (size_t)(gr & ((1 << log2ny) - 1)) * nxpad + 2 * j) =
            buf[(c * nr + r) * LS + j];
      }
  }
}

