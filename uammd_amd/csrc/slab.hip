// z-slab domain decomposition of path A: the per-rank bookkeeping of a membership refresh as a handful of kernels.
//
// New design (the reference is single GPU).  A rank owns the particles of its slab in persistent arrays (positions,
// velocities, ids; rows beyond the owned count are scratch / the ghost tail).  Every few steps (uammd_amd/parallel.py,
// DistributedLJ) it must (1) find who left through the two faces, (2) pack them for the neighbours, (3) put the arrivals into the
// holes they leave and keep the owned rows dense, (4) find who is within the halo distance of a face.  Done with torch element-wise /
// nonzero / cat / index_copy launches this cost 630 us per refresh at 1e6 particles; here it is
//   uammd_slab_select          ordered selection of the rows beyond two planes (three small launches, counts left on the device)
//   uammd_slab_pack_rows       leavers -> 8-float rows (position shifted into the receiver's frame, velocity, id)
//   uammd_slab_unpack_rows     arrivals into the holes, the tail compacted into the holes that stay open (one launch + a merge)
//   uammd_slab_max_displacement  the skin check's largest displacement since the last refresh
// The index lists are ASCENDING, as torch.nonzero gives them: both code paths (this one on the GPU, the torch one under gloo on the
// CPU) produce the same order of owned rows and ghosts, hence the same forces bit for bit.
#include "celllist.hpp"

namespace uammd_hip {

constexpr int kSlabBlock = 256;
constexpr int kSlabPerThread = 4;
constexpr int kSlabTile = kSlabBlock * kSlabPerThread;

// pass 1: per-tile counts of the rows with z >= zUp (list 0) and z < zDown (list 1)
__global__ void __launch_bounds__(kSlabBlock) k_slab_count(const float4 *__restrict__ pos, int n, float zUp, float zDown,
                                                           uint2 *__restrict__ tileCount) {
  __shared__ uint wsum[2 * (kSlabBlock / 64)];
  uint cu = 0, cd = 0;
#pragma unroll
  for (int u = 0; u < kSlabPerThread; ++u) {
    const int i = blockIdx.x * kSlabTile + u * kSlabBlock + threadIdx.x;
    if (i < n) {
      const float z = pos[i].z;
      cu += z >= zUp;
      cd += z < zDown;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cu += __shfl_xor(cu, o, 64);
    cd += __shfl_xor(cd, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = cu; wsum[kSlabBlock / 64 + (threadIdx.x >> 6)] = cd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint a = 0, b = 0;
    for (int w = 0; w < kSlabBlock / 64; ++w) { a += wsum[w]; b += wsum[kSlabBlock / 64 + w]; }
    tileCount[blockIdx.x] = make_uint2(a, b);
  }
}

// pass 2: exclusive scan of the tile counts by one workgroup (<= a few thousand tiles); totals to counts[0..1]
__global__ void __launch_bounds__(1024) k_slab_scan(const uint2 *__restrict__ tileCount, int ntiles, uint2 *__restrict__ tileStart,
                                                    int *__restrict__ counts) {
  __shared__ uint2 waveTotal[16];
  const int per = (ntiles + 1023) / 1024;
  const int lo = min((int)threadIdx.x * per, ntiles), hi = min(lo + per, ntiles);
  uint2 mine = make_uint2(0u, 0u);
  for (int i = lo; i < hi; ++i) { mine.x += tileCount[i].x; mine.y += tileCount[i].y; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint tx = __shfl_up(incl.x, o, 64), ty = __shfl_up(incl.y, o, 64);
    if (lane >= o) { incl.x += tx; incl.y += ty; }
  }
  if (lane == 63) waveTotal[wave] = incl;
  __syncthreads();
  uint2 before = make_uint2(0u, 0u);
  for (int w = 0; w < wave; ++w) { before.x += waveTotal[w].x; before.y += waveTotal[w].y; }
  uint2 run = make_uint2(before.x + incl.x - mine.x, before.y + incl.y - mine.y);
  for (int i = lo; i < hi; ++i) {
    tileStart[i] = run;
    run.x += tileCount[i].x;
    run.y += tileCount[i].y;
  }
  if (threadIdx.x == 1023) { counts[0] = (int)(before.x + incl.x); counts[1] = (int)(before.y + incl.y); }
}

// pass 3: the indices, ascending.  Inside a tile the order is (u, thread): position = tile start + rows of earlier u + earlier threads
__global__ void __launch_bounds__(kSlabBlock) k_slab_write(const float4 *__restrict__ pos, int n, float zUp, float zDown,
                                                           const uint2 *__restrict__ tileStart, int *__restrict__ idxUp,
                                                           int *__restrict__ idxDown) {
  __shared__ uint wcount[2 * (kSlabBlock / 64)];
  uint2 run = tileStart[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int u = 0; u < kSlabPerThread; ++u) {
    const int i = blockIdx.x * kSlabTile + u * kSlabBlock + threadIdx.x;
    bool up = false, down = false;
    if (i < n) {
      const float z = pos[i].z;
      up = z >= zUp;
      down = z < zDown;
    }
    const unsigned long long mu = __ballot(up), md = __ballot(down);
    if (lane == 0) { wcount[wave] = (uint)__popcll(mu); wcount[kSlabBlock / 64 + wave] = (uint)__popcll(md); }
    __syncthreads();
    uint bu = 0, bd = 0, tu = 0, td = 0;
#pragma unroll
    for (int w = 0; w < kSlabBlock / 64; ++w) {
      const uint a = wcount[w], b = wcount[kSlabBlock / 64 + w];
      if (w < wave) { bu += a; bd += b; }
      tu += a;
      td += b;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (up) idxUp[run.x + bu + (uint)__popcll(mu & below)] = i;
    if (down) idxDown[run.y + bd + (uint)__popcll(md & below)] = i;
    run.x += tu;
    run.y += td;
    __syncthreads();
  }
}

// leavers -> rows of 8 floats {x, y, z + dz, w, vx, vy, vz, id}
__global__ void __launch_bounds__(kSlabBlock) k_slab_pack_rows(const float4 *__restrict__ pos, const float *__restrict__ vel,
                                                               const int *__restrict__ ids, const int *__restrict__ idxUp, int nUp,
                                                               const int *__restrict__ idxDown, int nDown, float dzUp, float dzDown,
                                                               float4 *__restrict__ outUp, float4 *__restrict__ outDown) {
  const int t = blockIdx.x * kSlabBlock + threadIdx.x;
  if (t >= nUp + nDown) return;
  const bool up = t < nUp;
  const int k = up ? t : t - nUp;
  const int i = up ? idxUp[k] : idxDown[k];
  float4 p = pos[i];
  p.z += up ? dzUp : dzDown;
  const float4 q = make_float4(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], __int_as_float(ids[i]));
  float4 *o = (up ? outUp : outDown) + 2 * (size_t)k;
  o[0] = p;
  o[1] = q;
}

UH_D int lower_bound(const int *__restrict__ a, int n, int v) {  // first k with a[k] >= v
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// the two ascending lists of leavers merged into one ascending list of holes (each element finds its rank in the other list)
__global__ void __launch_bounds__(kSlabBlock) k_slab_merge(const int *__restrict__ idxUp, int nUp, const int *__restrict__ idxDown, int nDown,
                                                           int *__restrict__ holes) {
  const int t = blockIdx.x * kSlabBlock + threadIdx.x;
  if (t < nUp) {
    const int v = idxUp[t];
    holes[t + lower_bound(idxDown, nDown, v)] = v;
  } else if (t < nUp + nDown) {
    const int v = idxDown[t - nUp];
    holes[(t - nUp) + lower_bound(idxUp, nUp, v)] = v;
  }
}

UH_D void slab_store_row(float4 *pos, float *vel, int *ids, int dst, float4 p, float4 q) {
  pos[dst] = p;
  vel[3 * dst] = q.x; vel[3 * dst + 1] = q.y; vel[3 * dst + 2] = q.z;
  ids[dst] = __float_as_int(q.w);
}

// Arrival k goes to holes[k] (k < nLeave) or is appended at n + (k - nLeave).  If fewer arrive than leave, the holes that stay open below
// the new count newN are filled by the rows of the tail [newN, n) that stay, both in ascending order: row r of the tail is the
// (r - newN - #holes in [newN, r))-th stayer and takes the open hole of that rank.
__global__ void __launch_bounds__(kSlabBlock) k_slab_unpack_rows(float4 *__restrict__ pos, float *__restrict__ vel, int *__restrict__ ids,
                                                                 int n, const int *__restrict__ holes, int nLeave,
                                                                 const float4 *__restrict__ arrivals, int nArrive) {
  const int t = blockIdx.x * kSlabBlock + threadIdx.x;
  if (t < nArrive) {
    const int dst = t < nLeave ? holes[t] : n + (t - nLeave);
    slab_store_row(pos, vel, ids, dst, arrivals[2 * (size_t)t], arrivals[2 * (size_t)t + 1]);
    return;
  }
  const int k = t - nArrive;  // tail row index
  const int newN = n - nLeave + nArrive;
  if (nArrive >= nLeave || k >= n - newN) return;
  const int r = newN + k;
  const int hr = lower_bound(holes, nLeave, r);
  if (hr < nLeave && holes[hr] == r) return;              // r itself leaves
  const int h0 = lower_bound(holes, nLeave, newN);        // holes at or beyond newN start here
  const int rank = k - (hr - h0);
  const int dst = holes[nArrive + rank];                  // open holes: holes[nArrive ...), those below newN come first
  slab_store_row(pos, vel, ids, dst, pos[r], make_float4(vel[3 * r], vel[3 * r + 1], vel[3 * r + 2], __int_as_float(ids[r])));
}

__global__ void __launch_bounds__(kSlabBlock) k_slab_max_disp(const float4 *__restrict__ pos, const float4 *__restrict__ ref, int n,
                                                              uint *__restrict__ maxBits) {
  const int i = blockIdx.x * kSlabBlock + threadIdx.x;
  float d = 0.0f;
  if (i < n) {
    const float4 a = pos[i], b = ref[i];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    d = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d = fmaxf(d, __shfl_xor(d, o, 64));
  __shared__ float wmax[kSlabBlock / 64];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kSlabBlock / 64; ++w) d = fmaxf(d, wmax[w]);
    // one atomic per workgroup, and only from those that can raise the maximum (16 k waves hammering one address serialised into
    // 180 us per call); non-negative floats order as their bits
    const uint bits = __float_as_uint(d);
    if (bits > __hip_atomic_load(maxBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxBits, bits);
  }
}

}  // namespace uammd_hip

using namespace uammd_hip;

// Two segments in one launch (blockIdx.y picks the segment): the halo planes a slab folds into its owned planes after the spreading
// (ADD) and the planes it receives before the gather (copy).  As torch expressions these were four launches per FCM step.
template <bool ADD>
__global__ void __launch_bounds__(kSlabBlock) k_slab_pair(float *__restrict__ d0, const float *__restrict__ s0, float *__restrict__ d1,
                                                          const float *__restrict__ s1, size_t count, int vec) {
  float *d = blockIdx.y ? d1 : d0;
  const float *s = blockIdx.y ? s1 : s0;
  const size_t i = (size_t)blockIdx.x * kSlabBlock + threadIdx.x;
  if (vec) {
    if (4 * i >= count) return;
    float4 v = reinterpret_cast<const float4 *>(s)[i];
    if (ADD) {
      const float4 o = reinterpret_cast<const float4 *>(d)[i];
      v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
    }
    reinterpret_cast<float4 *>(d)[i] = v;
  } else {
    if (i >= count) return;
    d[i] = ADD ? d[i] + s[i] : s[i];
  }
}
static int slab_pair(bool add, float *d0, const float *s0, float *d1, const float *s1, size_t count, void *stream) {
  if (!d0 || !s0 || !d1 || !s1) { set_last_error("uammd_slab_add2 / copy2: null pointer"); return -1; }
  if (count == 0) return 0;
  const bool vec = count % 4 == 0 && (((uintptr_t)d0 | (uintptr_t)s0 | (uintptr_t)d1 | (uintptr_t)s1) & 15u) == 0;
  const size_t threads = vec ? count / 4 : count;
  const dim3 grid((unsigned)((threads + kSlabBlock - 1) / kSlabBlock), 2);
  if (add) hipLaunchKernelGGL(k_slab_pair<true>, grid, dim3(kSlabBlock), 0, (hipStream_t)stream, d0, s0, d1, s1, count, vec ? 1 : 0);
  else hipLaunchKernelGGL(k_slab_pair<false>, grid, dim3(kSlabBlock), 0, (hipStream_t)stream, d0, s0, d1, s1, count, vec ? 1 : 0);
  UH_CHECK(hipGetLastError());
  return 0;
}

extern "C" {

int uammd_slab_select_workspace(int n, size_t *bytes) {
  if (!bytes || n < 0) { set_last_error("uammd_slab_select_workspace: bad arguments"); return -1; }
  const size_t ntiles = ((size_t)n + kSlabTile - 1) / kSlabTile;
  *bytes = 2 * sizeof(uint2) * (ntiles + 1);
  return 0;
}

int uammd_slab_select(const float *d_pos, int n, float zUp, float zDown, int *d_idxUp, int *d_idxDown, int *d_counts, void *d_workspace,
                      void *stream) {
  if (n < 0 || !d_counts || (n > 0 && (!d_pos || !d_idxUp || !d_idxDown || !d_workspace))) {
    set_last_error("uammd_slab_select: bad arguments");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { UH_CHECK(hipMemsetAsync(d_counts, 0, 2 * sizeof(int), st)); return 0; }
  const int ntiles = (n + kSlabTile - 1) / kSlabTile;
  if (ntiles > 1024 * 64) { set_last_error("uammd_slab_select: too many particles (%d)", n); return -2; }
  uint2 *tileCount = (uint2 *)d_workspace, *tileStart = tileCount + (ntiles + 1);
  hipLaunchKernelGGL(k_slab_count, dim3(ntiles), dim3(kSlabBlock), 0, st, (const float4 *)d_pos, n, zUp, zDown, tileCount);
  hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(1024), 0, st, (const uint2 *)tileCount, ntiles, tileStart, d_counts);
  hipLaunchKernelGGL(k_slab_write, dim3(ntiles), dim3(kSlabBlock), 0, st, (const float4 *)d_pos, n, zUp, zDown, (const uint2 *)tileStart,
                     d_idxUp, d_idxDown);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_slab_pack_rows(const float *d_pos, const float *d_vel, const int *d_ids, const int *d_idxUp, int nUp, const int *d_idxDown,
                         int nDown, float dzUp, float dzDown, float *d_outUp, float *d_outDown, void *stream) {
  if (nUp < 0 || nDown < 0) { set_last_error("uammd_slab_pack_rows: negative count"); return -1; }
  if (nUp + nDown == 0) return 0;
  hipLaunchKernelGGL(k_slab_pack_rows, dim3((nUp + nDown + kSlabBlock - 1) / kSlabBlock), dim3(kSlabBlock), 0, (hipStream_t)stream,
                     (const float4 *)d_pos, d_vel, d_ids, d_idxUp, nUp, d_idxDown, nDown, dzUp, dzDown, (float4 *)d_outUp, (float4 *)d_outDown);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_slab_unpack_rows(float *d_pos, float *d_vel, int *d_ids, int n, const int *d_idxUp, int nUp, const int *d_idxDown, int nDown,
                           const float *d_arrivals, int nArrive, int *d_holes, void *stream) {
  if (n < 0 || nUp < 0 || nDown < 0 || nArrive < 0) { set_last_error("uammd_slab_unpack_rows: negative count"); return -1; }
  const int nLeave = nUp + nDown;
  if (nLeave == 0 && nArrive == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (nLeave > 0)
    hipLaunchKernelGGL(k_slab_merge, dim3((nLeave + kSlabBlock - 1) / kSlabBlock), dim3(kSlabBlock), 0, st, d_idxUp, nUp, d_idxDown, nDown, d_holes);
  const int tail = nArrive < nLeave ? nLeave - nArrive : 0;  // rows [newN, n)
  const int threads = nArrive + tail;
  if (threads > 0)
    hipLaunchKernelGGL(k_slab_unpack_rows, dim3((threads + kSlabBlock - 1) / kSlabBlock), dim3(kSlabBlock), 0, st, (float4 *)d_pos, d_vel,
                       d_ids, n, (const int *)d_holes, nLeave, (const float4 *)d_arrivals, nArrive);
  UH_CHECK(hipGetLastError());
  return 0;
}

int uammd_slab_max_displacement(const float *d_pos, const float *d_ref, int n, float *d_max, void *stream) {
  if (n < 0 || !d_max) { set_last_error("uammd_slab_max_displacement: bad arguments"); return -1; }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_slab_max_disp, dim3((n + kSlabBlock - 1) / kSlabBlock), dim3(kSlabBlock), 0, (hipStream_t)stream, (const float4 *)d_pos,
                     (const float4 *)d_ref, n, (uint *)d_max);
  UH_CHECK(hipGetLastError());
  return 0;
}

// The whole membership refresh of a slab step as ONE library call (the kernels above + the communicator's entry points in the order
// DistributedLJ runs them): who leaves, the migration rows there and back, who is in the halo, the ghosts into the tail of the position
// array, the skin check's displacement and its new reference.  Two host reads (the message sizes: they size the launches that follow)
// and nothing between the launches but this function — from Python the same sequence was ~270 us of host per refresh, most of it the
// interpreter between twelve launches.  comm == NULL: a world of one in process (what goes up arrives from below).
}  // extern "C"

__global__ void __launch_bounds__(256) k_mark_listed(const int *__restrict__ up, int nUp, const int *__restrict__ down, int nDown,
                                                     unsigned char *__restrict__ mask) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < nUp + nDown) mask[t < nUp ? up[t] : down[t - nUp]] = 1;
}

extern "C" {

int uammd_slab_refresh_lj(uammd_comm *comm, float *d_pos, float *d_vel, int *d_ids, float *d_force, int n, int capRows, float width, float reach,
                          int *d_idx, int *d_holes, int *d_counts, void *d_selectWorkspace, float *d_rows, float *d_arrivals, float *d_send,
                          float *d_ref, int refRows, float *d_maxDisplacement, unsigned char *d_listedMask, int out[10], void *stream) {
  if (!d_pos || !d_vel || !d_ids || !d_force || !d_idx || !d_holes || !d_counts || !d_selectWorkspace || !d_rows || !d_arrivals || !d_send ||
      !out || n < 0 || capRows < n) {
    set_last_error("uammd_slab_refresh_lj: bad arguments");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  int *idxUp = d_idx, *idxDown = d_idx + capRows, *haloUp = d_idx + 2 * (size_t)capRows, *haloDown = d_idx + 3 * (size_t)capRows;
  const float half = 0.5f * width;
  if (d_ref && d_maxDisplacement && refRows == n && n > 0)
    if (int e = uammd_slab_max_displacement(d_pos, d_ref, n, d_maxDisplacement, stream)) return e;
  // ---- who leaves ----
  if (int e = uammd_slab_select(d_pos, n, half, -half, idxUp, idxDown, d_counts, d_selectWorkspace, stream)) return e;
  int c4[4] = {0, 0, 0, 0};   // {to up, to down, from down, from up}
  auto sizes = [&](const int *d_two) -> int {
    if (comm) return uammd_comm_exchange_counts_device(comm, d_two, c4, stream);
    UH_CHECK(hipMemcpyAsync(c4, d_two, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    UH_CHECK(hipStreamSynchronize(st));
    c4[2] = c4[0]; c4[3] = c4[1];
    return 0;
  };
  if (int e = sizes(d_counts)) return e;
  const int nUp = c4[0], nDown = c4[1], nFromDown = c4[2], nFromUp = c4[3];
  const int nLeave = nUp + nDown, nArrive = nFromDown + nFromUp;
  if (nLeave > capRows || nArrive > capRows) { set_last_error("uammd_slab_refresh_lj: migration overflows the exchange buffer"); return -2; }
  if (nLeave || nArrive) {
    float *sendUp = d_rows, *sendDown = d_rows + 8 * (size_t)nUp;
    if (int e = uammd_slab_pack_rows(d_pos, d_vel, d_ids, idxUp, nUp, idxDown, nDown, -width, width, nUp ? sendUp : nullptr,
                                     nDown ? sendDown : nullptr, stream)) return e;
    const float *arrivals = d_rows;   // in process: [from down | from up] = [send up | send down]
    if (comm) {
      if (int e = uammd_comm_halo_exchange(comm, sendUp, nUp, sendDown, nDown, d_arrivals, nFromDown, d_arrivals + 8 * (size_t)nFromDown, nFromUp, 8,
                                           stream)) return e;
      arrivals = d_arrivals;
    }
    if (n - nLeave + nArrive > capRows) { set_last_error("uammd_slab_refresh_lj: migration overflows the particle buffers"); return -2; }
    if (int e = uammd_slab_unpack_rows(d_pos, d_vel, d_ids, n, idxUp, nUp, idxDown, nDown, arrivals, nArrive, d_holes, stream)) return e;
    n = n - nLeave + nArrive;
  }
  if (n > 0) UH_CHECK(hipMemsetAsync(d_force, 0, sizeof(float) * 4 * (size_t)n, st));   // (arrivals and moved rows: the half step zeroed the old layout)
  // ---- who is in the halo ----
  if (int e = uammd_slab_select(d_pos, n, half - reach, -half + reach, haloUp, haloDown, d_counts + 2, d_selectWorkspace, stream)) return e;
  if (int e = sizes(d_counts + 2)) return e;
  const int hUp = c4[0], hDown = c4[1], gFromDown = c4[2], gFromUp = c4[3];
  if (n + gFromDown + gFromUp > capRows) { set_last_error("uammd_slab_refresh_lj: the halo overflows the position buffer"); return -2; }
  // both halo lists share d_send (float[capRows][4]): a slab narrower than two reaches lists a particle twice
  if (comm && hUp + hDown > capRows) { set_last_error("uammd_slab_refresh_lj: the halo overflows the send buffer"); return -2; }
  float *tailDown = d_pos + 4 * (size_t)n, *tailUp = d_pos + 4 * (size_t)(n + gFromDown);
  if (comm) {
    float *outUp = d_send, *outDown = d_send + 4 * (size_t)hUp;
    if (int e = uammd_halo_pack(d_pos, haloUp, hUp, haloDown, hDown, -width, width, outUp, outDown, stream)) return e;
    if (int e = uammd_comm_halo_exchange(comm, outUp, hUp, outDown, hDown, tailDown, gFromDown, tailUp, gFromUp, 4, stream)) return e;
  } else {
    if (int e = uammd_halo_pack(d_pos, haloUp, hUp, haloDown, hDown, -width, width, tailDown, tailUp, stream)) return e;
  }
  if (d_ref && n > 0) UH_CHECK(hipMemcpyAsync(d_ref, d_pos, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToDevice, st));
  if (d_listedMask && n > 0) {   // one byte per owned row: is it in a membership list (uammd_halo_pack_gj1 / uammd_celllist_update_gj1's skip)
    UH_CHECK(hipMemsetAsync(d_listedMask, 0, (size_t)n, st));
    if (hUp + hDown > 0) {
      hipLaunchKernelGGL(k_mark_listed, dim3((hUp + hDown + 255) / 256), dim3(256), 0, st, haloUp, hUp, haloDown, hDown, d_listedMask);
      UH_CHECK(hipGetLastError());
    }
  }
  out[0] = n; out[1] = n + gFromDown + gFromUp; out[2] = hUp; out[3] = hDown; out[4] = gFromDown; out[5] = gFromUp;
  out[6] = nUp; out[7] = nDown; out[8] = nFromDown; out[9] = nFromUp;
  return 0;
}

int uammd_slab_add2(float *d_dst0, const float *d_src0, float *d_dst1, const float *d_src1, size_t count, void *stream) {
  return slab_pair(true, d_dst0, d_src0, d_dst1, d_src1, count, stream);
}
int uammd_slab_copy2(float *d_dst0, const float *d_src0, float *d_dst1, const float *d_src1, size_t count, void *stream) {
  return slab_pair(false, d_dst0, d_src0, d_dst1, d_src1, count, stream);
}

}  // extern "C"
