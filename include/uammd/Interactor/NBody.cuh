// Interactor/NBody.cuh — the include path and interface of the reference's src/Interactor/NBody.cuh:33-58: a Transverser applied to every
// pair of a ParticleGroup's members, `NBody(pg).transverse(tr, stream)`; made where it is needed, holds nothing.  hipcc (device code of
// the user's Transverser).
#pragma once
#include "NBodyBase.cuh"
#if defined(__HIPCC__)
namespace uammd {
class NBody {
  shared_ptr<ParticleGroup> pg;
public:
  NBody(shared_ptr<ParticleGroup> pg) : pg(pg) { System::log<System::DEBUG>("[NBody] Created"); }
  NBody(shared_ptr<ParticleData> pd) : NBody(std::make_shared<ParticleGroup>(pd, "All")) {}
  template <class Transverser> inline void transverse(Transverser &a_tr, hipStream_t st = 0) {
    const int N = pg->getNumberParticles();
    auto pd = pg->getParticleData();
    auto pos = pd->getPos(access::location::gpu, access::mode::read);
    device::detail::prepare(a_tr, pd);   // TransverserAdaptor::prepare (NBody.cuh:52): tr.prepare(pd) when the functor has it
    const int *members = pg->getIndicesRawPtr(access::location::gpu);
    if (members) NBodyBase::transverse(pos.raw(), members, a_tr, N, st);
    else NBodyBase::transverse(pos.raw(), a_tr, N, st);
  }
};
}  // namespace uammd
#endif
