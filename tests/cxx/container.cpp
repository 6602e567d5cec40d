// utils/container.h in a PLAIN C++ translation unit (g++, no thrust): uninitialized_cached_vector<T> on the runtime-API handles
// (detail::DevicePtr / DeviceRef) — the operations the reference's programs perform on it (test/BDHI/FCM/fcm_test.cu:105-136,
// test/BDHI/PSE/pse_test.cu:77-111): construct with a size, element writes and reads from the host, copy (device to device), growing
// resize that keeps the contents, data().get() into a C-ABI call, conversion to a host vector, the pool getting every block back.
#include "utils/container.h"
#include <cassert>
#include <cstdio>
#include <vector>
using namespace uammd;

int main() {
  auto &pool = detail::DevicePool::instance();
  const size_t live0 = pool.blocksLive();
  {
    uninitialized_cached_vector<real4> pos(5);
    assert(pos.size() == 5 && !pos.empty());
    for (int i = 0; i < 5; ++i) pos[i] = make_real4(real(i), real(2 * i), real(-i), real(7));
    const real4 p3 = pos[3];
    assert(p3.x == 3 && p3.y == 6 && p3.z == -3 && p3.w == 7);
    auto force = pos;                       // a copy: its own block, the same values
    assert(force.data().get() != pos.data().get());
    force[3] = make_real4(1, 0, 0, 0);
    assert(real4(pos[3]).y == 6 && real4(force[3]).x == 1 && real4(force[2]).y == 4);
    pos.resize(9);                          // grows: the first five stay
    assert(pos.size() == 9 && real4(pos[4]).x == 4);
    pos.resize(2);                          // shrinks in place
    assert(pos.size() == 2 && real4(pos[1]).y == 2);
    pos[0] = force[3];                      // element to element on the device
    assert(real4(pos[0]).x == 1);
    std::vector<real4> host = force;        // conversion to a host vector
    assert(host.size() == 5 && host[2].y == 4);
    uninitialized_cached_vector<real> fromHost(std::vector<real>{1, 2, 3});
    assert(real(fromHost[2]) == 3);
    // data().get() / raw() into the C ABI: zero the block, read it back through the handle
    detail::check(uammd_fill_zero(force.data().get(), sizeof(real4) * force.size(), nullptr));
    detail::hipCheck(hipDeviceSynchronize(), "sync");
    assert(real4(force[4]).z == 0 && force.raw() == force.data().get());
    // iterators: begin() + i addresses element i
    auto it = force.begin() + 2;
    *it = make_real4(9, 9, 9, 9);
    assert(real4(force[2]).w == 9 && force.end() - force.begin() == 5);
    uninitialized_cached_vector<real4> moved = std::move(force);
    assert(force.size() == 0 && moved.size() == 5);
    moved.clear();
    assert(moved.empty());
    uninitialized_cached_vector<int> none;
    assert(none.size() == 0 && none.begin() == none.end());
  }
  assert(pool.blocksLive() == live0);       // every block went back to the pool
  {  // System::getSystemParameters (System.h:37-46,299) and the Integrator's set of updatables (Integrator.cuh:90-124)
    auto sys = std::make_shared<System>();
    const SystemParameters sp = sys->getSystemParameters();
    std::printf("device %d, architecture %d, managed memory %d\n", sp.device, sp.cuda_arch, (int)sp.managedMemoryAvailable);
    assert(sp.device >= 0 && sp.cuda_arch == 950);
    struct Heard : public Interactor {
      using Interactor::Interactor;
      int times = 0;
      void sum(Computables, hipStream_t) override {}
      void updateSimulationTime(real) override { ++times; }
    };
    auto pd = std::make_shared<ParticleData>(8, sys);
    BD::EulerMaruyama::Parameters par;
    par.dt = real(0.1); par.hydrodynamicRadius = 1; par.viscosity = 1;
    BD::EulerMaruyama bd(pd, par);
    auto in = std::make_shared<Heard>(pd, "heard");
    bd.addInteractor(in);
    bd.addUpdatable(in);                    // the same object again: a set, it hears each update once
    assert(bd.getUpdatables().size() == 1 && bd.getInteractors().size() == 1);
    bd.forwardTime();
    bd.forwardTime();
    assert(in->times == 2);
  }
  std::printf("container: ok\n");
  return 0;
}
