// Host-side state of one cell list (owned through the opaque uammd_celllist handle).
#pragma once
#include "device_common.hpp"
#include "../../include/uammd_hip.h"

#include <string>

namespace uammd_hip {

struct DeviceBuffer {
  void *ptr = nullptr;
  size_t cap = 0;
  bool owned = true;
  int reserve(size_t bytes);  // grows (never shrinks); contents are NOT preserved
  void alias(void *p, size_t bytes) { ptr = p; cap = bytes; owned = false; }  // a view into another buffer
  ~DeviceBuffer();
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

struct CellList {
  // outputs (device)
  DeviceBuffer hash, sortHash, index, indexAlt, sortPos, cellStart, cellEnd, errorFlag;
  // counting-sort build state
  DeviceBuffer keyCount, keyStart, provRank, members, scratch, keyOutside, cellOutside, cellRange;
  DeviceBuffer scanFlags;        // k_key_scan's per-workgroup totals, tagged with scanGeneration
  uint scanGeneration = 0;
  bool lastFusedTile = false;  // uammd_verletnvt_gj_lj_step: the previous fused step ended in the tile traversal
  bool gjInHash = false;  // the last update applied the half step inside its hash kernel (gjDone: applied at all)
  DeviceBuffer packHalf;  // half-precision copy of sortPos for the traversals' prefilter, built on demand (ensure_pack)
  bool packValid = false;
  float packScale = 0.f;
  DeviceBuffer zeroBlock;  // errorFlag | keyOutside | keyCount live here: ONE memset per build instead of three — and none when the
                           // previous counting build of the same grid left it zeroed (zeroBlockClean)
  bool zeroBlockClean = false;
  hipStream_t buildStream = nullptr;  // stream of the last build: its tail zeroes the block, which orders nothing for ANOTHER stream
  bool buildStreamSet = false;
  char *zeroBase = nullptr;
  size_t zeroLayout[2] = {0, 0};
  GridT<float> grid{};
  float boxL[3] = {0, 0, 0};
  int boxPeriodic[3] = {0, 0, 0};
  uint validCell = 0;
  long long validCounter = -1;
  int lastN = -1;
  int nCellsAlloc = -1;
  int numberParticlesBuilt = 0;
  int endBit = 0;
  uint nKeys = 0;          // 2^endBit (0 if the key space is not tabulated)
  bool haveKeyStart = false;
  bool haveCellOutside = false;
  bool usedCounting = false;
  bool forceRadix = false;  // test hook: always take the rocPRIM radix path
  bool aggregateHash = true;  // k_hash_agg (LDS-aggregated histogram) instead of one global atomic per particle
  // NaN positions / particles outside a non-periodic box (CellListBase.cuh:82-85 raises a flag; :258-264 synchronises and throws
  // in UAMMD_DEBUG builds only, release builds carry on with those particles missing from the tables).
  // The kernels raise the flag in HOST-mapped memory (written only in the error case, so it costs nothing per step); the host
  // looks at it at the start of the next update / get / traversal and fails that call, or at once with strictErrors (sync).
  int *hostErr = nullptr, *devErr = nullptr;
  bool strictErrors = false, reportErrors = true;
  int check_errors(hipStream_t st, bool sync);
  // largest cutOff2 of an LJ parameter table in device memory, read back once per (pointer, ntypes) (lj.hip: is the tile kernel allowed?)
  const void *ljTable = nullptr;
  int ljTableTypes = 0;
  float ljTableMaxCut2 = 0.f;
  unsigned ljTableEpoch = 0;  // uammd_lj_table_changed() count at the time the table was read
  bool ljTableUnit = false;  // one type with sigma^2 = epsilon / sigma^2 = 1 (the tile kernel's reduced-units instantiation)
  int lj_max_cutoff2(const void *d_table, int ntypes, hipStream_t st, float *out);
  ~CellList();
  // Traversal-kernel timing for bench.py's roofline line: when enabled the LJ traversal is launched with hipExtLaunchKernel and a
  // start / stop event pair from a ring (the events ride on the kernel's own dispatch packet: no extra barrier packets in the
  // stream, unlike an hipEventRecord on either side of every launch).  Completed pairs are summed lazily.
  struct Profile {
    static constexpr int kRing = 128;
    bool enabled = false;
    hipEvent_t ev[kRing][2] = {};
    int head = 0, live = 0;   // live pairs are [head - live, head)
    double totalMs = 0.0;
    long long launches = 0;
    int next(hipEvent_t *start, hipEvent_t *stop);  // a pair for the next launch (collects the oldest one when the ring is full)
    int collect(int upTo);                           // sum and retire all but the newest `upTo` live pairs
    ~Profile();
  } prof;
  DeviceBuffer tileStats;  // uammd_lj_tile_stats: counters the tile kernel bumps while enabled
  bool tileStatsOn = false;
  int numOwned = 0x7fffffff;  // traversal option: particles with input index >= numOwned are ghosts (neighbours only, no output)

  int next_valid_cell(int numberParticles, bool *needsClear);
  int update(const float4 *d_pos, int numberParticles, const float L[3], const int periodic[3], const int cellDim[3],
             hipStream_t st, const struct GJFuse *gj = nullptr);
  bool gjDone = false;  // the last update applied the fused half step (see update)
  int ensure_pack(hipStream_t st);  // 0 = packHalf is valid for the current list; 1 = this grid has no packed copy
};

extern thread_local char g_last_error[1024];

}  // namespace uammd_hip
