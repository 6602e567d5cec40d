/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (uammd_amd/, include/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker / reported baseline.
 *
 * CPU restatement of the arithmetic of UAMMD's value types.  Citations are relative to
 * /root/reference/src.
 *
 *   real / real3 / real4     global/defines.h:33-44   (float unless -DDOUBLE_PRECISION)
 *   Box                      utils/Box.cuh:16-58
 *   Grid                     utils/Grid.cuh:21-131
 *
 * FLOATING-POINT CONTRACT (shared, by construction, with the HIP kernels in uammd_amd/csrc):
 * the reference is compiled by nvcc, whose default (--fmad=true) fuses a*b+c into one FMA in
 * device code.  Which products get fused is a compiler decision, so this file pins one choice and
 * spells it with explicit FMA() calls; everything else is compiled with -ffp-contract=off:
 *   - a*b + c (c not itself a product)            -> FMA(a,b,c)
 *   - a.x*b.x + a.y*b.y + a.z*b.z  (dot)          -> FMA(a.z,b.z, FMA(a.y,b.y, a.x*b.x))
 *   - r + (cond ? off*L : 0)   (Box::apply_pbc)   -> unfused mul, select, add (a select sits
 *                                                     between the product and the sum)
 * Division and sqrt are IEEE correctly rounded on both sides.  exp/log/sin/cos are NOT bit
 * reproducible between libm and device intrinsics: anything downstream of them is compared with
 * a stated tolerance.
 */
#ifndef ORACLE_COMMON_H
#define ORACLE_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef DOUBLE_PRECISION
typedef double real;
#define FMA(a, b, c) fma((a), (b), (c))
#define FLOOR(a) floor(a)
#define CEIL(a) ceil(a)
#define SQRT(a) sqrt(a)
#define EXP(a) exp(a)
#define FABS(a) fabs(a)
#define ROUND(a) round(a)
#define SIN(a) sin(a)
#define COS(a) cos(a)
#else
typedef float real;
#define FMA(a, b, c) fmaf((a), (b), (c))
#define FLOOR(a) floorf(a)
#define CEIL(a) ceilf(a)
#define SQRT(a) sqrtf(a)
#define EXP(a) expf(a)
#define FABS(a) fabsf(a)
#define ROUND(a) roundf(a)
#define SIN(a) sinf(a)
#define COS(a) cosf(a)
#endif

typedef unsigned int uint;
typedef struct { real x, y, z; } real3;
typedef struct { real x, y, z, w; } real4;
typedef struct { real x, y; } real2;
typedef struct { int x, y, z; } int3;

static inline real3 mk3(real x, real y, real z) { real3 r = {x, y, z}; return r; }
static inline int3 mki3(int x, int y, int z) { int3 r = {x, y, z}; return r; }
/* utils/vector.cuh:737-739 */
static inline real dot3(real3 a, real3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }

/* ---- Box: utils/Box.cuh:16-58 -------------------------------------------------------------- */
typedef struct { real3 boxSize, minusInvBoxSize; } Box;

static inline Box box_make(real3 L) { /* Box.cuh:22-32 */
  Box b;
  b.boxSize = L;
  b.minusInvBoxSize = mk3((real)(-1.0) / L.x, (real)(-1.0) / L.y, (real)(-1.0) / L.z);
  if (L.x == (real)0.0 || isinf(L.x)) b.minusInvBoxSize.x = (real)0.0;
  if (L.y == (real)0.0 || isinf(L.y)) b.minusInvBoxSize.y = (real)0.0;
  if (L.z == (real)0.0 || isinf(L.z)) b.minusInvBoxSize.z = (real)0.0;
  return b;
}
static inline void box_set_periodicity(Box *b, int x, int y, int z) { /* Box.cuh:34-41 */
  if (!x) b->minusInvBoxSize.x = 0;
  if (!y) b->minusInvBoxSize.y = 0;
  if (!z) b->minusInvBoxSize.z = 0;
}
static inline int box_px(const Box *b) { return b->minusInvBoxSize.x != 0; }
static inline int box_py(const Box *b) { return b->minusInvBoxSize.y != 0; }
static inline int box_pz(const Box *b) { return b->minusInvBoxSize.z != 0; }
/* C-ABI helper: L[3] + periodic[3] -> Box (Box(real3) then setPeriodicity) */
static inline Box box_from(const real *L, const int *periodic) {
  Box b = box_make(mk3(L[0], L[1], L[2]));
  box_set_periodicity(&b, periodic[0], periodic[1], periodic[2]);
  return b;
}

static inline real3 box_apply_pbc(const Box *b, real3 r) { /* Box.cuh:51-58 */
  real3 off;
  off.x = FLOOR(FMA(r.x, b->minusInvBoxSize.x, (real)0.5));
  off.y = FLOOR(FMA(r.y, b->minusInvBoxSize.y, (real)0.5));
  off.z = FLOOR(FMA(r.z, b->minusInvBoxSize.z, (real)0.5));
  real tx = off.x * b->boxSize.x, ty = off.y * b->boxSize.y, tz = off.z * b->boxSize.z;
  r.x += box_px(b) ? tx : 0;
  r.y += box_py(b) ? ty : 0;
  r.z += box_pz(b) ? tz : 0;
  return r;
}

/* ---- Grid: utils/Grid.cuh:21-131 ----------------------------------------------------------- */
typedef struct {
  int3 gridPos2CellIndex;
  int3 cellDim;
  real3 cellSize, invCellSize;
  Box box;
  real cellVolume;
} Grid;

static inline Grid grid_make(Box box, int3 cellDim) { /* Grid.cuh:35-48 */
  Grid g;
  g.box = box;
  if (cellDim.z == 0) cellDim.z = 1;
  g.cellDim = cellDim;
  g.cellSize = mk3(box.boxSize.x / (real)cellDim.x, box.boxSize.y / (real)cellDim.y,
                   box.boxSize.z / (real)cellDim.z);
  /* `1.0 / cellSize` resolves to operator/(const float&, float3): b / a.x in `real` */
  g.invCellSize = mk3((real)1.0 / g.cellSize.x, (real)1.0 / g.cellSize.y, (real)1.0 / g.cellSize.z);
  if (box.boxSize.z == (real)0.0) g.invCellSize.z = 0;
  g.gridPos2CellIndex = mki3(1, cellDim.x, cellDim.x * cellDim.y);
  g.cellVolume = g.cellSize.x * g.cellSize.y;
  if (cellDim.z > 1) g.cellVolume *= g.cellSize.z;
  return g;
}
/* Grid(Box, real3 minCellSize): cellDim = make_int3(boxSize/minCellSize), C truncation (Grid.cuh:31) */
static inline Grid grid_make_mincell(Box box, real3 minCellSize) {
  int3 cd = mki3((int)(box.boxSize.x / minCellSize.x), (int)(box.boxSize.y / minCellSize.y),
                 (int)(box.boxSize.z / minCellSize.z));
  return grid_make(box, cd);
}
static inline int3 grid_get_cell(const Grid *g, real3 r) { /* Grid.cuh:49-72 */
  real3 p = box_apply_pbc(&g->box, r);
  /* (pbc(r) + 0.5*L) * invCellSize: the sum is the multiplicand, nothing to fuse */
  real fx = (p.x + (real)0.5 * g->box.boxSize.x) * g->invCellSize.x;
  real fy = (p.y + (real)0.5 * g->box.boxSize.y) * g->invCellSize.y;
  real fz = (p.z + (real)0.5 * g->box.boxSize.z) * g->invCellSize.z;
  int3 c = mki3((int)fx, (int)fy, (int)fz);
  if (c.x == g->cellDim.x) c.x = 0;
  if (c.y == g->cellDim.y) c.y = 0;
  if (c.z == g->cellDim.z) c.z = 0;
  return c;
}
static inline int grid_cell_index(const Grid *g, int3 c) { /* Grid.cuh:74-76 */
  return c.x * g->gridPos2CellIndex.x + c.y * g->gridPos2CellIndex.y + c.z * g->gridPos2CellIndex.z;
}
static inline int grid_pbc_coord(const Grid *g, int coord, int cell) { /* Grid.cuh:90-106 */
  int ncells = 0;
  if (coord == 0 && box_px(&g->box)) ncells = g->cellDim.x;
  if (coord == 1 && box_py(&g->box)) ncells = g->cellDim.y;
  if (coord == 2 && box_pz(&g->box)) ncells = g->cellDim.z;
  if (cell <= -1) cell += ncells;
  else if (cell >= ncells) cell -= ncells;
  return cell;
}
static inline int3 grid_pbc_cell(const Grid *g, int3 c) { /* Grid.cuh:82-88 */
  return mki3(grid_pbc_coord(g, 0, c.x), grid_pbc_coord(g, 1, c.y), grid_pbc_coord(g, 2, c.z));
}
static inline int grid_ncells(const Grid *g) { return g->cellDim.x * g->cellDim.y * g->cellDim.z; }
/* Grid.cuh:124-131: cell centres at (c+0.5)*h measured from the lower box corner */
static inline real3 grid_cell_center(const Grid *g, int3 c) {
  return mk3(g->cellSize.x * ((real)c.x + (real)0.5), g->cellSize.y * ((real)c.y + (real)0.5),
             g->cellSize.z * ((real)c.z + (real)0.5));
}
static inline real3 grid_distance_to_cell_center(const Grid *g, real3 pos, int3 c) {
  /* (pos + L*0.5) - cellSize*(c+0.5): L*0.5 is exact, so the first sum is the same fused or not;
   * the subtraction of the product is the a*b+c pattern -> one FMA (contract in the header). */
  real3 d = mk3(FMA(-g->cellSize.x, (real)c.x + (real)0.5, pos.x + g->box.boxSize.x * (real)0.5),
                FMA(-g->cellSize.y, (real)c.y + (real)0.5, pos.y + g->box.boxSize.y * (real)0.5),
                FMA(-g->cellSize.z, (real)c.z + (real)0.5, pos.z + g->box.boxSize.z * (real)0.5));
  return box_apply_pbc(&g->box, d);
}

#define ORACLE_API __attribute__((visibility("default")))
/* celllist.c: all-cores mode of the build and the IBM spreading for bench.py's cpu_baseline (results: see there) */
ORACLE_API void oracle_set_parallel(int on);
ORACLE_API int oracle_get_parallel(void);
#endif
