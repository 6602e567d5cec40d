// The reference's examples/misc/checkpoint.cu on this build: fill a ParticleData, save it in UAMMD's text checkpoint format,
// restore it, and check both the values and the file itself (the text a real UAMMD run writes for the same state).
#include "utils/checkpoint.h"
#include <cstdio>
#include <sstream>
using namespace uammd;

int main(int argc, char *argv[]) {
  auto sys = std::make_shared<System>(argc, argv);
  const int N = 10;
  const char *file = argc > 1 ? argv[1] : "/tmp/uammd_pd.dat";
  {
    auto pd = std::make_shared<ParticleData>(N, sys);
    {
      auto pos = pd->getPos(access::cpu, access::write);
      for (auto &p : pos) p = make_real4(1, 1, 1, 1);
      auto charge = pd->getCharge(access::cpu, access::write);
      for (auto &c : charge) c = 2;
      pos[3] = make_real4(0.125, -2.5, 1e-7, 3);
    }  // the handles are released before the properties are requested again
    saveParticleData(file, pd);
  }
  auto pd = restoreParticleData(file, sys);
  int bad = pd->getNumParticles() != N;
  {
    auto pos = pd->getPos(access::cpu, access::read);
    auto charge = pd->getCharge(access::cpu, access::read);
    for (int i = 0; i < N; ++i) {
      const real4 e = i == 3 ? make_real4(0.125, -2.5, 1e-7, 3) : make_real4(1, 1, 1, 1);
      bad += !(pos[i].x == e.x && pos[i].y == e.y && pos[i].z == e.z && pos[i].w == e.w && charge[i] == 2);
    }
  }
  bad += pd->isVelAllocated();  // only the blocks present in the file are allocated
  std::ifstream in(file);
  std::stringstream ss;
  ss << in.rdbuf();
  std::string expect = "# version 3.0.0\n# 10\n# Pos\n";
  for (int i = 0; i < N; ++i) expect += i == 3 ? "0.125 -2.5 1e-07 3\n" : "1 1 1 1\n";
  expect += "# Charge\n";
  for (int i = 0; i < N; ++i) expect += "2\n";
  bad += ss.str() != expect;
  std::printf("checkpoint %s: %s\n", file, bad ? "MISMATCH" : "ok");
  sys->finish();
  return bad;
}
