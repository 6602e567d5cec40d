// signal / connection / scoped_connection of include/uammd/uammd.h (the reference's nod::unsafe_signal / nod::connection,
// ParticleData/ParticleData.cuh:110-125) and the pool behind System::allocator — host-only: no device call is made, runs without a GPU.
#include <cassert>
#include <cstdio>
#include <memory>
#include <vector>

#include "uammd.cuh"
using namespace uammd;

struct Listener {  // what CellList / VerletList / FCMIntegrator do: a [this] slot held through a connection member
  int calls = 0;
  connection c;
  explicit Listener(signal<void(void)> &s) { c = s.connect([this]() { ++calls; }); }
  ~Listener() { c.disconnect(); }
};

int main() {
  signal<void(void)> sig;
  assert(sig.slot_count() == 0);
  int a = 0, b = 0;
  connection ca = sig.connect([&]() { ++a; });
  connection cb = sig.connect([&]() { ++b; });
  assert(ca.connected() && cb.connected() && sig.slot_count() == 2);
  sig();
  assert(a == 1 && b == 1);
  ca.disconnect();
  assert(!ca.connected() && sig.slot_count() == 1);
  sig();
  assert(a == 1 && b == 2);
  ca.disconnect();  // twice: harmless
  // the use-after-free of round 4: a listener destroyed before the signal's owner
  {
    auto l = std::make_unique<Listener>(sig);
    sig();
    assert(l->calls == 1 && sig.slot_count() == 2);
  }
  assert(sig.slot_count() == 1);
  sig();  // would have called into freed memory with append-only callbacks
  assert(b == 4);
  // a slot that disconnects itself, and one that connects another slot, while the signal is being emitted
  int selfCalls = 0, lateCalls = 0;
  connection self, late;
  self = sig.connect([&]() { ++selfCalls; self.disconnect(); });
  connection adder = sig.connect([&]() { if (!late.connected()) late = sig.connect([&]() { ++lateCalls; }); });
  sig();
  assert(selfCalls == 1 && lateCalls == 0);  // connected during the emission: not called by it
  sig();
  assert(selfCalls == 1 && lateCalls == 1);
  // scoped_connection drops its slot at the end of the scope; moving a connection moves the handle
  {
    scoped_connection sc = sig.connect([&]() { ++a; });
    assert(sc.connected());
    const int before = sig.slot_count();
    connection moved = sig.connect([&]() {});
    connection target = std::move(moved);
    assert(!moved.connected() && target.connected() && sig.slot_count() == before + 1);
    target.disconnect();
  }
  const int n = sig.slot_count();
  sig();
  assert(sig.slot_count() == n);
  // a connection that outlives its signal
  connection orphan;
  {
    signal<void(int)> s2;
    int got = 0;
    orphan = s2.connect([&](int v) { got = v; });
    s2(7);
    assert(got == 7 && orphan.connected());
  }
  assert(!orphan.connected());
  orphan.disconnect();
  // a slot that destroys the object owning the signal it is called from
  {
    struct Owner { std::shared_ptr<signal<void(void)>> s = std::make_shared<signal<void(void)>>(); };
    auto owner = std::make_unique<Owner>();
    auto keep = owner->s;
    connection c1 = keep->connect([&]() { owner.reset(); });
    int after = 0;
    connection c2 = keep->connect([&]() { ++after; });
    (*keep)();
    assert(!owner && after == 1);
  }
  sig.disconnect_all_slots();
  assert(sig.empty() && !cb.connected());
  std::printf("signal semantics ok\n");
  return 0;
}
