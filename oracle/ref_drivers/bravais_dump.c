/* TEST INFRASTRUCTURE ONLY (oracle/ref.mk).  Runs the reference's own lattice generator — third_party/bravais/bravais.h, compiled from where
 * it lies under /root/reference (include path given by ref.mk; nothing is copied) — and prints the positions as raw float32, for
 * tests/golden/make_bravais_golden.py.  usage: bravais_dump <type 0..6> <N> <Lx> <Ly> <Lz>  (binary float4 array on stdout) */
#include "bravais.h"
#include <string.h>

int main(int argc, char **argv) {
  if (argc != 6) return 2;
  const int type = atoi(argv[1]), n = atoi(argv[2]);
  const float lx = (float)atof(argv[3]), ly = (float)atof(argv[4]), lz = (float)atof(argv[5]);
  float *pos = (float *)calloc((size_t)4 * n, sizeof(float));
  Bravais(pos, (BRAVAISLAT)type, n, lx, ly, lz, 0.0f, NULL, NULL, false);
  fwrite(pos, sizeof(float), (size_t)4 * n, stdout);
  free(pos);
  return 0;
}
