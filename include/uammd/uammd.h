// uammd.h — UAMMD's host-side C++14 interface on top of libuammd_hip (MI355X / gfx950).
//
// Same class names, member functions, parameter structs and error behaviour as the reference headers for the two
// hot paths (citations relative to the reference's src/):
//   System                        System/System.h:63-316            (log levels, rng(), finish())
//   Box, Grid                     utils/Box.cuh:16-92, utils/Grid.cuh:21-139
//   ParticleData, property_ptr    ParticleData/ParticleData.cuh:161-465, ParticleData/Property.cuh:49-147
//   Interactor / Integrator       Interactor/Interactor.cuh:56-119, Integrator/Integrator.cuh:33-125
//   CellList                      Interactor/NeighbourList/CellList.cuh:83-205
//   VerletList                    Interactor/NeighbourList/VerletList.cuh:83-201
//   Potential::LJ                 Interactor/Potential/Potential.cuh:25-85, RadialPotential.cuh:49-154
//   PairForces<Potential, NL>     Interactor/PairForces.cuh:23-64, PairForces.cu:43-78
//   VerletNVT::{Basic,GronbechJensen}   Integrator/VerletNVT.cuh:55-115
//   BD::EulerMaruyama             Integrator/BrownianDynamics.cuh
//   BDHI::FCM, BDHI::FCMIntegrator, BDHI::EulerMaruyama<Method>   Integrator/BDHI/BDHI_FCM.cuh:84-198, BDHI_EulerMaruyama.cuh
//   IBM                           misc/IBM.cuh:99-203   (windows UAMMD ships; see uammd_hip.h)
//   BDHI::PSE, BDHI::EulerMaruyama<Method>   Integrator/BDHI/BDHI_PSE.cuh:79-176, BDHI_EulerMaruyama.cu:125-166
//   lanczos::Solver, MatrixDot    misc/LanczosAlgorithm.cuh:32-83, LanczosAlgorithm/MatrixDot.h:7-25
//
// PRECISION.  `real` is float unless the program is compiled with -DDOUBLE_PRECISION (global/defines.h), as the reference's unit tests are
// (test/CMakeLists.txt:9).  The library's tuned hot path — cell list, LJ traversal, integrators, the tile-owned spreading and in-LDS FFT of
// FCM, the pair-record PSE near field — is single precision by construction; its DOUBLE_PRECISION build is the `_f64` part of the C ABI
// (layout-generic kernels, rocFFT in double).  A DOUBLE_PRECISION program therefore gets: System, Box, Grid, ParticleData (+ sortParticles),
// ParticleGroup, ParticleSorter, uninitialized_cached_vector, IBM<Kernel> (any kernel, device template), the FCM kernels, FCM_impl,
// BDHI::FCM, BDHI::FCMIntegrator (no rotation), BDHI::PSE, BDHI::Lanczos, BDHI::Cholesky, BDHI::EulerMaruyama<Method>, BDHI::True2D / Quasi2D, Poisson, lanczos::Solver, the
// four BD schemes — every class the reference's unit tests (test/CMakeLists.txt:19-28, minus the doubly periodic and Chebyshev ones) and its
// double-precision acceptance programs of path B (test/BDHI/{FCM, quasi2D, Lanczos_Cholesky}) construct.  The classes whose backend exists in
// single precision only (CellList, VerletList, PairForces, VerletNVT, FIB, ICM, Comm) are not declared in that build: their
// forwarding headers stop the compilation with a message instead of silently computing in float.
//
// This header is plain host C++14: compile with any C++ compiler,
//     g++ -std=c++14 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude main.cpp \
//         -Luammd_amd/lib -luammd_hip -L/opt/rocm/lib -lamdhip64
// Every device computation goes through the C ABI of uammd_hip.h; there is no CPU fallback.  User-defined DEVICE
// functors (custom Transversers / IBM windows) are the one thing that cannot cross a C ABI: they need hipcc, as they
// need nvcc in the reference.  Streams are hipStream_t where the reference says cudaStream_t.
#ifndef UAMMD_MI355X_UAMMD_H
#define UAMMD_MI355X_UAMMD_H

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../uammd_hip.h"
#include "utils/vector.cuh"
#include "third_party/saruprng.cuh"   // (uammd.cuh makes Saru visible to every program, src/uammd.cuh:14)
// A TU compiled by hipcc also gets the DEVICE side of the neighbour lists (NeighbourContainer and its iterators, the list kernels): the
// host classes below then hand out getNeighbourContainer() and take device iterators, as the reference's do under nvcc.  A plain C++
// compiler sees the host interface alone.
#if defined(__HIPCC__)
#include "device/Transverser.hip.hpp"
#include <thrust/device_malloc_allocator.h>
#include <thrust/copy.h>
#include <thrust/device_ptr.h>
#include <thrust/device_reference.h>
#include <thrust/execution_policy.h>
#include <thrust/host_vector.h>
#include <thrust/iterator/detail/normal_iterator.h>
#include <thrust/iterator/iterator_traits.h>
#include <thrust/iterator/reverse_iterator.h>
#include <thrust/transform.h>
#endif

namespace uammd {

using std::make_shared;
using std::shared_ptr;

// ---- global/defines.h:33-44, utils/vector.cuh: real, real2/3/4, int2/3 and their arithmetic (own headers at the reference's paths) ----
// ---- errors: utils/debugTools.h:20-64, utils/exception.h ---------------------------------------------------------
struct cuda_generic_error : public std::runtime_error {
  cuda_generic_error(const std::string &m, int code) : std::runtime_error(m), code(code) {}
  int code;
};
struct illegal_property_access : public std::runtime_error {
  using std::runtime_error::runtime_error;
};
using exception = std::exception;  // utils/exception.h:11
// utils/exception.h:13-27: what() of an exception and of every exception nested in it, one log line per level
inline void backtrace_nested_exception(const uammd::exception &e, int level = 0);
namespace detail {
inline void check(int rc) {
  if (rc != 0) throw cuda_generic_error(std::string(uammd_hip_last_error()), rc);
}
inline void hipCheck(hipError_t e, const char *what) {
  if (e != hipSuccess) throw cuda_generic_error(std::string(what) + ": " + hipGetErrorString(e), (int)e);
}
inline int nextFFTWiseSize(int v) {  // utils/Grid.cuh:142-213, one dimension
  for (;; ++v) {
    int m = v;
    if (m % 2) continue;
    for (int p : {2, 3, 5, 7, 11}) while (m % p == 0) m /= p;
    if (m == 1) return v;
  }
}
// The pool of temporary device memory (System.h:65-78, misc/allocator.h: pool_memory_resource_adaptor): blocks that are given back are
// kept, by size, and handed out again instead of going through hipFree / hipMalloc (a hipFree waits for the device).  One pool per
// process, single host thread (the reference's assumption too); System::finish() and the end of the process return the kept blocks.
// STREAM ASSUMPTION: a block given back may be handed out again at once, and DeviceArray::resize clears it on the NULL stream.  That is
// ordered after earlier work only for the null stream and for streams created with the default (blocking) flag, which the null stream
// waits for — thrust's pool resource in the reference makes the same assumption.  A program that runs the modules on a stream created
// with hipStreamNonBlocking must synchronise that stream before it destroys (or lets the library resize) a temporary the stream uses.
class DevicePool {
  std::multimap<size_t, void *> kept;  // free blocks by capacity
  std::map<void *, size_t> live;       // blocks handed out -> capacity
  size_t keptBytes = 0;
  static size_t capacityFor(size_t bytes) {  // 256-byte granules up to 1 MiB, then 1/8-octave steps: bounded slack, few distinct sizes
    if (bytes <= (size_t(1) << 20)) return (bytes + 255) & ~size_t(255);
    size_t step = size_t(1) << 17;
    while ((step << 4) < bytes) step <<= 1;
    return (bytes + step - 1) / step * step;
  }
public:
  static DevicePool &instance() { static DevicePool *p = new DevicePool; return *p; }  // (never destroyed: no runtime calls at exit)
  void *allocate(size_t bytes) {
    if (bytes == 0) return nullptr;
    const size_t cap = capacityFor(bytes);
    auto it = kept.find(cap);
    void *ptr = nullptr;
    if (it != kept.end()) { ptr = it->second; kept.erase(it); keptBytes -= cap; }
    else {
      hipError_t e = hipMalloc(&ptr, cap);
      if (e != hipSuccess) {  // give the kept blocks back to the runtime and try once more
        (void)hipGetLastError();
        release();
        hipCheck(hipMalloc(&ptr, cap), "hipMalloc");
      }
    }
    live[ptr] = cap;
    return ptr;
  }
  void deallocate(void *ptr) {
    if (!ptr) return;
    auto it = live.find(ptr);
    if (it == live.end()) { (void)hipFree(ptr); return; }  // not ours
    kept.emplace(it->second, ptr);
    keptBytes += it->second;
    live.erase(it);
  }
  void release() {
    for (auto &b : kept) (void)hipFree(b.second);
    kept.clear();
    keptBytes = 0;
  }
  size_t bytesKept() const { return keptBytes; }
  size_t blocksLive() const { return live.size(); }
};
template <class T> struct DeviceArray {  // owning device buffer (thrust::device_vector stand-in for host code), memory from the pool
  T *d = nullptr;
  size_t n = 0;
  DeviceArray() = default;
  explicit DeviceArray(size_t n_) { resize(n_); }
  DeviceArray(const DeviceArray &) = delete;
  DeviceArray &operator=(const DeviceArray &) = delete;
  DeviceArray(DeviceArray &&o) noexcept : d(o.d), n(o.n) { o.d = nullptr; o.n = 0; }
  DeviceArray &operator=(DeviceArray &&o) noexcept { swap(o); return *this; }
  ~DeviceArray() { DevicePool::instance().deallocate(d); }
  T *data() { return d; }
  const T *data() const { return d; }
  T *begin() { return d; }
  T *end() { return d + n; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void resize(size_t m) {
    if (m == n) return;
    DevicePool::instance().deallocate(d);
    d = nullptr;
    n = m;
    if (m) {
      d = static_cast<T *>(DevicePool::instance().allocate(sizeof(T) * m));
      hipCheck(hipMemset(d, 0, sizeof(T) * m), "hipMemset");
    }
  }
  void swap(DeviceArray &o) { std::swap(d, o.d); std::swap(n, o.n); }
};
// ---- utils/container.h:16-117: uninitialized_cached_vector<T> — a device vector over the pool that never runs a fill kernel ----------
// What user code does with it (test/utils/ParticleSorter.cu:26-32, test/BDHI/FCM/fcm_test.cu:105-136, test/BDHI/PSE/pse_test.cu:77-111):
// construct with a size, copy (device to device), resize keeping the contents, v[i] = x / x = v[i] from the host, begin() / end() /
// rbegin() / rend() into thrust algorithms, v.data().get() into kernels and C-ABI calls, conversion to a host vector.  In a TU compiled by
// hipcc the handles are thrust's (device_ptr, device_reference, device_vector's iterator type), so thrust dispatches to the device; a
// plain C++ TU gets handles of the same surface written on the runtime API (DevicePtr / DeviceRef below).
#if !defined(__HIPCC__)
template <class T> class DeviceRef {   // v[i] on the host: reads and writes one element through the runtime (thrust::device_reference's role)
  T *p;
public:
  explicit DeviceRef(T *p_) : p(p_) {}
  operator T() const { T h; hipCheck(hipMemcpy(&h, p, sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H"); return h; }
  DeviceRef &operator=(const T &v) { hipCheck(hipMemcpy(p, &v, sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D"); return *this; }
  DeviceRef &operator=(const DeviceRef &o) { if (p != o.p) hipCheck(hipMemcpy(p, o.p, sizeof(T), hipMemcpyDeviceToDevice), "hipMemcpy D2D"); return *this; }
  T *operator&() const { return p; }
};
template <class T> class DevicePtr {   // thrust::device_ptr's role: a tagged raw pointer; get() is the raw pointer
  T *p;
public:
  using difference_type = std::ptrdiff_t;
  using value_type = T;
  explicit DevicePtr(T *p_ = nullptr) : p(p_) {}
  T *get() const { return p; }
  operator T *() const { return p; }   // (C-ABI calls and runtime copies take the handle as it is)
  DeviceRef<T> operator*() const { return DeviceRef<T>(p); }
  DeviceRef<T> operator[](std::ptrdiff_t i) const { return DeviceRef<T>(p + i); }
  DevicePtr operator+(std::ptrdiff_t i) const { return DevicePtr(p + i); }
  DevicePtr operator-(std::ptrdiff_t i) const { return DevicePtr(p - i); }
  std::ptrdiff_t operator-(const DevicePtr &o) const { return p - o.p; }
  DevicePtr &operator++() { ++p; return *this; }
  DevicePtr &operator+=(std::ptrdiff_t i) { p += i; return *this; }
  bool operator==(const DevicePtr &o) const { return p == o.p; }
  bool operator!=(const DevicePtr &o) const { return p != o.p; }
};
#endif
template <class T> class PooledVector {
  T *d = nullptr;
  size_t n = 0, cap = 0;
  void copyFrom(const T *src, size_t count, hipMemcpyKind kind) {
    if (count) hipCheck(hipMemcpy(d, src, sizeof(T) * count, kind), "hipMemcpy");
  }
public:
  using value_type = T;
#if defined(__HIPCC__)
  using pointer = thrust::device_ptr<T>;
  using reference = thrust::device_reference<T>;
  using iterator = thrust::detail::normal_iterator<thrust::device_ptr<T>>;   // (thrust::device_vector<T>::iterator, container.h:42)
#else
  using pointer = DevicePtr<T>;
  using reference = DeviceRef<T>;
  using iterator = DevicePtr<T>;
#endif
  PooledVector() = default;
  explicit PooledVector(size_t size) { resize(size); }
  PooledVector(const std::vector<T> &host) : PooledVector(host.size()) { copyFrom(host.data(), n, hipMemcpyHostToDevice); }
  PooledVector(const PooledVector &o) : PooledVector(o.n) { copyFrom(o.d, n, hipMemcpyDeviceToDevice); }
  PooledVector(PooledVector &&o) noexcept : d(o.d), n(o.n), cap(o.cap) { o.d = nullptr; o.n = o.cap = 0; }
  PooledVector &operator=(PooledVector o) noexcept { swap(o); return *this; }
  ~PooledVector() { DevicePool::instance().deallocate(d); }
#if defined(__HIPCC__)
  PooledVector(const thrust::host_vector<T> &host) : PooledVector(host.size()) { copyFrom(thrust::raw_pointer_cast(host.data()), n, hipMemcpyHostToDevice); }
  operator thrust::host_vector<T>() const {
    thrust::host_vector<T> h(n);
    if (n) hipCheck(hipMemcpy(thrust::raw_pointer_cast(h.data()), d, sizeof(T) * n, hipMemcpyDeviceToHost), "hipMemcpy");
    return h;
  }
  thrust::reverse_iterator<iterator> rbegin() const { return thrust::make_reverse_iterator(end()); }
  thrust::reverse_iterator<iterator> rend() const { return thrust::make_reverse_iterator(begin()); }
#endif
  operator std::vector<T>() const {
    std::vector<T> h(n);
    if (n) hipCheck(hipMemcpy(h.data(), d, sizeof(T) * n, hipMemcpyDeviceToHost), "hipMemcpy");
    return h;
  }
  pointer data() const { return pointer(d); }
  T *raw() const { return d; }                 // (not in the reference: data().get() without the handle)
  iterator begin() const { return iterator(pointer(d)); }
  iterator end() const { return iterator(pointer(d + n)); }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  reference operator[](size_t i) const { return reference(pointer(d + i)); }
  // growing keeps the elements and leaves the new tail as the pool hands it out; shrinking keeps the block (container.h:74-86)
  void resize(size_t newSize) {
    if (newSize > cap) {
      T *grown = static_cast<T *>(DevicePool::instance().allocate(sizeof(T) * newSize));
      if (n) hipCheck(hipMemcpy(grown, d, sizeof(T) * n, hipMemcpyDeviceToDevice), "hipMemcpy");
      DevicePool::instance().deallocate(d);
      d = grown;
      cap = newSize;
    }
    n = newSize;
  }
  void clear() { DevicePool::instance().deallocate(d); d = nullptr; n = cap = 0; }
  void swap(PooledVector &o) noexcept { std::swap(d, o.d); std::swap(n, o.n); std::swap(cap, o.cap); }
};
}  // namespace detail
template <class T> using uninitialized_cached_vector = detail::PooledVector<T>;

// ---- utils/utils.h:21-32: wall-clock stopwatch, seconds ----
class Timer {
  std::chrono::time_point<std::chrono::system_clock> t0;
public:
  void tic() { t0 = std::chrono::system_clock::now(); }
  float toc() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now() - t0).count() * 1e-9; }
};

// ---- utils/utils.h:38-115 ----------------------------------------------------------------------------------------
class Xorshift128plus {
  uint64_t s[2];
public:
  Xorshift128plus() { s[0] = 12679825035178159220ULL; s[1] = 15438657923749336752ULL; }
  explicit Xorshift128plus(uint64_t s0) { setSeed(s0); }
  uint64_t next() {
    uint64_t x = s[0];
    const uint64_t y = s[1];
    s[0] = y;
    x ^= x << 23;
    x ^= x >> 17;
    x ^= y ^ (y >> 26);
    s[1] = x;
    return x + y;
  }
  uint32_t next32() { return next() % std::numeric_limits<uint32_t>::max(); }
  double uniform(double min, double max) { return min + (next() / ((double)0xFFffFFffFFffFFffULL)) * (max - min); }
  ::double3 uniform3(double min, double max) {   // (double3, as utils/utils.h:70)
    const double a = uniform(min, max), b = uniform(min, max), c = uniform(min, max);
    return ::double3(a, b, c);
  }
  ::double2 uniform2(double min, double max) { const double a = uniform(min, max), b = uniform(min, max); return ::double2(a, b); }
  // Box-Muller, two numbers per pair of uniforms: the second is handed out by the next call (utils/utils.h:77-104; the spare is shared by
  // all generators of the process there, and here)
  double gaussian(double mean, double std) {
    static double spare;
    static bool haveSpare = false;
    haveSpare = !haveSpare;
    if (!haveSpare) return spare * std + mean;
    double u1, u2;
    do { u1 = uniform(0, 1); u2 = uniform(0, 1); } while (u1 <= std::numeric_limits<double>::min());
    const double r = std::sqrt(-2.0 * std::log(u1)), phi = 2.0 * M_PI * u2;
    spare = r * std::sin(phi);
    return r * std::cos(phi) * std + mean;
  }
  ::double3 gaussian3(double mean, double std) { const double a = gaussian(mean, std), b = gaussian(mean, std), c = gaussian(mean, std); return ::double3(a, b, c); }
  ::double2 gaussian2(double mean, double std) { const double a = gaussian(mean, std), b = gaussian(mean, std); return ::double2(a, b); }
  void setSeed(uint64_t s0, uint64_t s1) { s[0] = s0; s[1] = s1; }
  void setSeed(uint64_t s0) { s[0] = s0; s[1] = (s0 + 15438657923749336752ULL) % 0xFFffFFffFFffFFffULL; }
};

// ---- System/System.h -----------------------------------------------------------------------------------------------
// System.h:37-46.  `device` is the device the System found current (or set: --device); `cuda_arch` is the reference's name for the
// device's architecture as a number — 950 for gfx950 (the digits of hipDeviceProp_t::gcnArchName), 0 when there is no device.
struct SystemParameters {
  int device = -1;
  int cuda_arch = 0;
  int minimumCudaArch = 200;
  bool managedMemoryAvailable = false;
};
class System {
  Xorshift128plus m_rng;
  int m_argc = 0;
  char **m_argv = nullptr;
  SystemParameters sysPar;
public:
  enum LogLevel { CRITICAL = 0, ERROR, EXCEPTION, WARNING, MESSAGE, STDERR, STDOUT, DEBUG, DEBUG1, DEBUG2, DEBUG3, DEBUG4, DEBUG5, DEBUG6, DEBUG7 };
  System() : System(0, nullptr) {}
  System(int argc, char **argv) : m_argc(argc), m_argv(argv) {
    const auto now = std::chrono::steady_clock::now().time_since_epoch();
    m_rng.setSeed(0xf31337Bada55D00dULL ^ (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now).count());
    int dev = -1;
    for (int i = 1; argv && i < argc; ++i)
      if (std::string(argv[i]) == "--device" && i + 1 < argc) dev = std::atoi(argv[i + 1]);  // System.h:128-139
    if (dev >= 0) detail::check(uammd_hip_set_device(dev));
    int current = -1;
    if (hipGetDevice(&current) == hipSuccess && current >= 0) {
      sysPar.device = current;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, current) == hipSuccess) {
        for (const char *c = prop.gcnArchName; *c && *c != ':'; ++c) if (*c >= '0' && *c <= '9') sysPar.cuda_arch = 10 * sysPar.cuda_arch + (*c - '0');
        sysPar.managedMemoryAvailable = prop.managedMemory != 0;
      }
    } else (void)hipGetLastError();
  }
  const SystemParameters getSystemParameters() const { return sysPar; }   // System.h:299
  Xorshift128plus &rng() { return m_rng; }
  int getargc() const { return m_argc; }                          // System.h:282-290
  const char **getargv() const { return (const char **)m_argv; }
  // System/Log.h:30-80: levels up to MAXLOGLEVEL (6 unless the program is compiled with another) are printed — CRITICAL .. STDERR and
  // the DEBUG levels on stderr, STDOUT on stdout — behind the level's tag; CRITICAL also throws (System.h:251-257)
  template <LogLevel level> static void log(const char *fmt, ...) {
    if (level > maxLogLevel()) return;
    va_list ap;
    va_start(ap, fmt);
    char buf[2048];
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    static const char *tag[] = {"\033[101m[CRITICAL] ", "\033[91m[ERROR] \033[0m", "\033[1m\033[91m[EXCEPTION] \033[0m", "\033[34m[WARNING] \033[0m",
                                "\033[92m[MESSAGE] \033[0m", " ", " "};
    std::fprintf(level == STDOUT ? stdout : stderr, "%s%s%s\n", level <= STDOUT ? tag[level] : "\033[96m[ DEBUG ] \033[0m", buf,
                 level == CRITICAL ? "\033[0m" : "");
    if (level == CRITICAL) throw std::runtime_error(std::string("[CRITICAL] ") + buf);
  }
  template <LogLevel level> static void log(const std::string &msg) { log<level>("%s", msg.c_str()); }
#ifdef MAXLOGLEVEL
  static int &maxLogLevel() { static int l = MAXLOGLEVEL; return l; }
#else
  static int &maxLogLevel() { static int l = STDOUT; return l; }
#endif
  void finish() { (void)hipDeviceSynchronize(); detail::DevicePool::instance().release(); }
  // System.h:65-78: allocators over the pool of temporary device memory.  allocator<T> hands out raw device pointers;
  // allocator_thrust<T> is the same pool behind thrust's allocator interface (thrust::device_vector<T, System::allocator_thrust<T>>,
  // thrust::device(System::allocator_thrust<char>()) as an execution policy) and needs a TU compiled by hipcc, as thrust does.
  template <class T> struct allocator {
    using value_type = T;
    allocator() = default;
    template <class U> allocator(const allocator<U> &) {}
    T *allocate(size_t n) const { return static_cast<T *>(detail::DevicePool::instance().allocate(n * sizeof(T))); }
    void deallocate(T *p, size_t = 0) const { detail::DevicePool::instance().deallocate(p); }
    template <class U> bool operator==(const allocator<U> &) const { return true; }
    template <class U> bool operator!=(const allocator<U> &) const { return false; }
  };
  template <class T> allocator<T> getTemporaryDeviceAllocator() { return allocator<T>(); }  // System.h:292-296
#if defined(__HIPCC__)
  template <class T> struct allocator_thrust : thrust::device_malloc_allocator<T> {
    using super = thrust::device_malloc_allocator<T>;
    using pointer = typename super::pointer;
    using size_type = typename super::size_type;
    template <class U> struct rebind { using other = allocator_thrust<U>; };
    __host__ allocator_thrust() {}
    __host__ allocator_thrust(const allocator_thrust &) = default;
    template <class U> __host__ allocator_thrust(const allocator_thrust<U> &) {}
    __host__ pointer allocate(size_type n) { return pointer(static_cast<T *>(detail::DevicePool::instance().allocate(n * sizeof(T)))); }
    __host__ void deallocate(pointer p, size_type = 0) { detail::DevicePool::instance().deallocate(thrust::raw_pointer_cast(p)); }
  };
#endif
};
inline void backtrace_nested_exception(const uammd::exception &e, int level) {
  System::log<System::EXCEPTION>(std::string(level, ' ') + "level " + std::to_string(level) + " exception: " + e.what());
  try { std::rethrow_if_nested(e); }
  catch (const std::exception &nested) { backtrace_nested_exception(nested, level + 1); }
  catch (...) {}
}

// ---- utils/Box.cuh ----------------------------------------------------------------------------------------------------
struct Box {  // a POD a kernel takes by value: every member is callable from device code in a TU compiled by hipcc (utils/Box.cuh:16-92)
  real3 boxSize, minusInvBoxSize;
  UAMMD_HOSTDEV Box() : Box(real(0)) {}
  UAMMD_HOSTDEV Box(real L) : Box(make_real3(L)) {}
  UAMMD_HOSTDEV Box(real3 L) : boxSize(L), minusInvBoxSize{real(-1.0) / L.x, real(-1.0) / L.y, real(-1.0) / L.z} {
    if (boxSize.x == real(0.0) || __builtin_isinf(boxSize.x)) minusInvBoxSize.x = real(0.0);
    if (boxSize.y == real(0.0) || __builtin_isinf(boxSize.y)) minusInvBoxSize.y = real(0.0);
    if (boxSize.z == real(0.0) || __builtin_isinf(boxSize.z)) minusInvBoxSize.z = real(0.0);
  }
  UAMMD_HOSTDEV void setPeriodicity(bool x, bool y, bool z) {
    if (!x) minusInvBoxSize.x = 0;
    if (!y) minusInvBoxSize.y = 0;
    if (!z) minusInvBoxSize.z = 0;
  }
  UAMMD_HOSTDEV bool isPeriodicX() const { return minusInvBoxSize.x != 0; }
  UAMMD_HOSTDEV bool isPeriodicY() const { return minusInvBoxSize.y != 0; }
  UAMMD_HOSTDEV bool isPeriodicZ() const { return minusInvBoxSize.z != 0; }
  // utils/Box.cuh:51-58 under the library's floating-point contract: the offset from ONE fused multiply-add, the shift
  // r + offset * L unfused — what the cell-list and traversal kernels compute, so a user kernel takes the same decision for a pair an
  // ulp from the cut-off
  UAMMD_HOSTDEV real3 apply_pbc(real3 r) const {
    const real ox = imageOffset(r.x, minusInvBoxSize.x), oy = imageOffset(r.y, minusInvBoxSize.y), oz = imageOffset(r.z, minusInvBoxSize.z);
    const real sx = ox * boxSize.x, sy = oy * boxSize.y, sz = oz * boxSize.z;
    r.x += isPeriodicX() ? sx : 0;
    r.y += isPeriodicY() ? sy : 0;
    r.z += isPeriodicZ() ? sz : 0;
    return r;
  }
  template <class VecType> UAMMD_HOSTDEV real3 apply_pbc(const VecType &r) const { return apply_pbc(make_real3(r)); }
  UAMMD_HOSTDEV bool isInside(const real3 &r) const {  // utils/Box.cuh:60-70
    const real3 h = boxSize * real(0.5);
    return !(r.x <= -h.x || r.x > h.x) && !(r.y <= -h.y || r.y > h.y) && !(r.z <= -h.z || r.z > h.z);
  }
  UAMMD_HOSTDEV real getVolume() const { return boxSize.z != real(0.0) ? boxSize.x * boxSize.y * boxSize.z : boxSize.x * boxSize.y; }
  UAMMD_HOSTDEV bool operator==(const Box &o) const {
    return boxSize.x == o.boxSize.x && boxSize.y == o.boxSize.y && boxSize.z == o.boxSize.z &&
           isPeriodicX() == o.isPeriodicX() && isPeriodicY() == o.isPeriodicY() && isPeriodicZ() == o.isPeriodicZ();
  }
  UAMMD_HOSTDEV bool operator!=(const Box &o) const { return !(*this == o); }
  // helpers for the C ABI (its single-precision entry points take float[3], the _f64 ones double[3])
  template <class S> void toArrays(S L[3], int per[3]) const {
    L[0] = S(boxSize.x); L[1] = S(boxSize.y); L[2] = S(boxSize.z);
    per[0] = isPeriodicX(); per[1] = isPeriodicY(); per[2] = isPeriodicZ();
  }
private:
  // floor(r * (-1 / L) + 1 / 2) with the product and the sum fused, in the working precision
  UAMMD_HOSTDEV static float imageOffset(float r, float minusInvL) { return ::floorf(::fmaf(r, minusInvL, 0.5f)); }
  UAMMD_HOSTDEV static double imageOffset(double r, double minusInvL) { return ::floor(::fma(r, minusInvL, 0.5)); }
};

// ---- utils/Grid.cuh:21-139: a box cut into cellDim cells — which cell holds a position, linear cell indices, periodic wrapping of cell
// coordinates.  Host twin of the arithmetic the cell-list and spreading kernels do on the device (csrc/device_common.hpp). ----
struct Grid {
  int3 gridPos2CellIndex;   // strides of the linear index: (1, nx, nx ny)
  int3 cellDim;
  real3 cellSize, invCellSize;
  Box box;
  real cellVolume;
  UAMMD_HOSTDEV Grid() : Grid(Box(), make_int3(0, 0, 0)) {}
  UAMMD_HOSTDEV Grid(Box box, real3 minCellSize) : Grid(box, make_int3(box.boxSize / minCellSize)) {}
  UAMMD_HOSTDEV Grid(Box box, real minCellSize) : Grid(box, make_real3(minCellSize)) {}
  UAMMD_HOSTDEV Grid(Box box_, int3 cells) : cellDim(cells), box(box_) {
    if (cellDim.z == 0) cellDim.z = 1;
    cellSize = box.boxSize / make_real3(cellDim);
    invCellSize = 1.0 / cellSize;
    if (box.boxSize.z == real(0.0)) invCellSize.z = 0;
    gridPos2CellIndex = make_int3(1, cellDim.x, cellDim.x * cellDim.y);
    cellVolume = cellSize.x * cellSize.y * (cellDim.z > 1 ? cellSize.z : real(1.0));
  }
  template <class VecType> UAMMD_HOSTDEV int3 getCell(const VecType &r) const {
    int3 cell = make_int3((box.apply_pbc(make_real3(r)) + real(0.5) * box.boxSize) * invCellSize);
    // (a position exactly on the upper face rounds to cell cellDim: it belongs to cell 0)
    if (cell.x == cellDim.x) cell.x = 0;
    if (cell.y == cellDim.y) cell.y = 0;
    if (cell.z == cellDim.z) cell.z = 0;
    return cell;
  }
  UAMMD_HOSTDEV int getCellIndex(const int3 &cell) const { return dot(cell, gridPos2CellIndex); }
  UAMMD_HOSTDEV int getCellIndex(const int2 &cell) const { return dot(cell, make_int2(gridPos2CellIndex)); }
  template <int coordinate> UAMMD_HOSTDEV int pbc_cell_coord(int cell) const {
    const int ncells = coordinate == 0 ? (box.isPeriodicX() ? cellDim.x : 0)
                                       : (coordinate == 1 ? (box.isPeriodicY() ? cellDim.y : 0) : (box.isPeriodicZ() ? cellDim.z : 0));
    if (cell <= -1) cell += ncells;
    else if (cell >= ncells) cell -= ncells;
    return cell;
  }
  UAMMD_HOSTDEV int3 pbc_cell(const int3 &cell) const { return make_int3(pbc_cell_coord<0>(cell.x), pbc_cell_coord<1>(cell.y), pbc_cell_coord<2>(cell.z)); }
  UAMMD_HOSTDEV int getNumberCells() const { return cellDim.x * cellDim.y * cellDim.z; }
  UAMMD_HOSTDEV real getCellVolume() const { return cellVolume; }
  UAMMD_HOSTDEV real getCellVolume(int3) const { return cellVolume; }
  UAMMD_HOSTDEV real3 getCellSize() const { return cellSize; }
  UAMMD_HOSTDEV real3 getCellSize(int3) const { return cellSize; }
  UAMMD_HOSTDEV real3 getCellCenter(int3 cell) const { return cellSize * (make_real3(cell) + real(0.5)); }
  UAMMD_HOSTDEV real3 distanceToCellCenter(real3 pos, int3 cell) const { return box.apply_pbc(pos + box.boxSize * real(0.5) - getCellCenter(cell)); }
  UAMMD_HOSTDEV real3 distanceToCellUpperLeftCorner(real3 pos, int3 cell) const { return box.apply_pbc(pos + box.boxSize * real(0.5) - cellSize * make_real3(cell)); }
};
// the next grid size, per axis, that is even and has only the factors 2, 3, 5, 7, 11 (utils/Grid.cuh:142-213)
inline int3 nextFFTWiseSize3D(int3 size) {
  return make_int3(detail::nextFFTWiseSize(size.x), detail::nextFFTWiseSize(size.y), size.z > 1 ? detail::nextFFTWiseSize(size.z) : size.z);
}

// ---- signal / connection (ParticleData.cuh:110-125: nod::unsafe_signal / nod::connection; single host thread, as there) ------------
// signal<void(Args...)>::connect(slot) hands out a connection; connection::disconnect() removes the slot; a connection may outlive its
// signal (it then reports !connected()).  A slot may disconnect itself or others while the signal is being emitted.
namespace detail {
struct signal_state_base {
  virtual ~signal_state_base() = default;
  virtual void remove(size_t id) = 0;
  virtual bool has(size_t id) const = 0;
};
}  // namespace detail
class connection {
  std::weak_ptr<detail::signal_state_base> state;
  size_t id = 0;
public:
  connection() = default;
  connection(std::weak_ptr<detail::signal_state_base> s, size_t id) : state(std::move(s)), id(id) {}
  connection(connection &&o) noexcept : state(std::move(o.state)), id(o.id) { o.state.reset(); o.id = 0; }
  connection &operator=(connection &&o) noexcept { state = std::move(o.state); id = o.id; o.state.reset(); o.id = 0; return *this; }
  connection(const connection &) = delete;
  connection &operator=(const connection &) = delete;
  bool connected() const { auto s = state.lock(); return s && s->has(id); }
  void disconnect() {
    if (auto s = state.lock()) s->remove(id);
    state.reset();
    id = 0;
  }
};
// a connection that disconnects when it goes out of scope (nod::scoped_connection): what the modules of this header hold
class scoped_connection {
  connection c;
public:
  scoped_connection() = default;
  scoped_connection(connection &&o) : c(std::move(o)) {}
  scoped_connection(scoped_connection &&) = default;
  scoped_connection &operator=(connection &&o) { c.disconnect(); c = std::move(o); return *this; }
  scoped_connection &operator=(scoped_connection &&o) { c.disconnect(); c = std::move(o.c); return *this; }
  ~scoped_connection() { c.disconnect(); }
  bool connected() const { return c.connected(); }
  void disconnect() { c.disconnect(); }
  void reset() { c.disconnect(); }
  connection release() { return std::move(c); }
};
template <class Signature> class signal;
template <class... Args> class signal<void(Args...)> {
  struct State : detail::signal_state_base {
    std::deque<std::pair<size_t, std::function<void(Args...)>>> slots;  // (a deque: connecting from inside a slot does not move the running one)
    size_t next = 1;
    int emitting = 0;
    bool holes = false;
    void remove(size_t id) override {
      for (auto &s : slots)
        if (s.first == id) { s.first = 0; holes = true; }  // (the slot itself is destroyed outside any emission: it may be the caller)
      if (!emitting) compact();
    }
    bool has(size_t id) const override {
      for (auto &s : slots) if (s.first == id && id) return true;
      return false;
    }
    void compact() {
      if (!holes) return;
      slots.erase(std::remove_if(slots.begin(), slots.end(), [](const std::pair<size_t, std::function<void(Args...)>> &s) { return s.first == 0; }), slots.end());
      holes = false;
    }
  };
  std::shared_ptr<State> st = std::make_shared<State>();
public:
  signal() = default;
  signal(const signal &) = delete;
  signal &operator=(const signal &) = delete;
  template <class Slot> connection connect(Slot &&slot) {
    const size_t id = st->next++;
    st->slots.emplace_back(id, std::function<void(Args...)>(std::forward<Slot>(slot)));
    return connection(std::weak_ptr<detail::signal_state_base>(st), id);
  }
  void operator()(Args... a) const {
    const std::shared_ptr<State> keep = st;  // (a slot may destroy the object that owns this signal)
    State &s = *keep;
    ++s.emitting;
    const size_t n = s.slots.size();  // slots connected during the emission are not called by it
    for (size_t i = 0; i < n; ++i)
      if (s.slots[i].first) s.slots[i].second(a...);
    if (--s.emitting == 0) s.compact();
  }
  int slot_count() const { int c = 0; for (auto &s : st->slots) c += s.first != 0; return c; }
  bool empty() const { return slot_count() == 0; }
  void disconnect_all_slots() { for (auto &s : st->slots) s.first = 0; st->holes = true; if (!st->emitting) st->compact(); }
};

// ---- access, property_ptr, ParticleData ---------------------------------------------------------------------------------
struct access {  // a struct, as in the reference (ParticleData/Property.cuh:30-39): `using uammd::access;` works
  enum location { cpu, gpu, managed, nodevice };
  enum mode { read, write, readwrite, nomode };
};

template <class T> class Property;
// RAII handle on a property (Property.cuh:49-147): releases the lock (and uploads host writes) when destroyed
template <class T> class property_ptr {
  T *ptr = nullptr;
  size_t m_size = 0;
  Property<T> *owner = nullptr;
  access::location loc = access::nodevice;
  access::mode mod = access::nomode;
public:
  property_ptr() = default;
  property_ptr(T *p, size_t n, Property<T> *o, access::location l, access::mode m) : ptr(p), m_size(n), owner(o), loc(l), mod(m) {}
  property_ptr(property_ptr &&o) noexcept { *this = std::move(o); }
  property_ptr &operator=(property_ptr &&o) noexcept {
    release();
    ptr = o.ptr; m_size = o.m_size; owner = o.owner; loc = o.loc; mod = o.mod;
    o.owner = nullptr; o.ptr = nullptr;
    return *this;
  }
  property_ptr(const property_ptr &) = delete;
  property_ptr &operator=(const property_ptr &) = delete;
  ~property_ptr() { release(); }
  void release();
  T *raw() const { return ptr; }
  T *get() const { return ptr; }
  T *begin() const { return ptr; }
  T *end() const { return ptr + m_size; }
  size_t size() const { return m_size; }
  T &operator[](size_t i) const { return ptr[i]; }
  access::location location() const { return loc; }
};

template <class T> class Property {
  std::string name;
  size_t n = 0;
  detail::DeviceArray<T> dev;
  std::vector<T> host;
  bool allocated = false, hostStale = true, deviceStale = false;
  bool isBeingWritten = false;
  int readers = 0;
  friend class property_ptr<T>;
public:
  explicit Property(std::string nm) : name(std::move(nm)) {}
  void resize(size_t m) { n = m; }
  bool isAllocated() const { return allocated; }
  size_t size() const { return n; }
  property_ptr<T> data(access::location loc, access::mode mode) {
    if (!allocated) { dev.resize(n); allocated = true; hostStale = true; }
    // Property.cuh:310-338: one writer XOR many readers
    if (isBeingWritten || (mode != access::read && readers > 0))
      throw illegal_property_access("[Property] You cant request " + name + " while it is locked");
    if (mode == access::read) ++readers; else isBeingWritten = true;
    if (loc == access::cpu) {
      if (host.size() != n) { host.resize(n); hostStale = true; }
      if (hostStale && mode != access::write) {
        detail::hipCheck(hipMemcpy(host.data(), dev.d, sizeof(T) * n, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      }
      hostStale = false;
      return property_ptr<T>(host.data(), n, this, loc, mode);
    }
    if (deviceStale) {
      detail::hipCheck(hipMemcpy(dev.d, host.data(), sizeof(T) * n, hipMemcpyHostToDevice), "hipMemcpy H2D");
      deviceStale = false;
    }
    if (mode != access::read) hostStale = true;
    return property_ptr<T>(dev.d, n, this, loc, mode);
  }
  void unlock(access::location loc, access::mode mode) {
    if (mode == access::read) --readers; else isBeingWritten = false;
    if (loc == access::cpu && mode != access::read) {  // host writes go back to the device when the handle dies
      detail::hipCheck(hipMemcpy(dev.d, host.data(), sizeof(T) * n, hipMemcpyHostToDevice), "hipMemcpy H2D");
    }
  }
  T *deviceRaw() { return dev.d; }
  void swapDeviceBuffer(detail::DeviceArray<T> &alt) { dev.swap(alt); hostStale = true; }
};
template <class T> void property_ptr<T>::release() {
  if (owner) owner->unlock(loc, mod);
  owner = nullptr;
}

class ParticleData {
  int numberParticles;
  shared_ptr<System> sys;
  Property<real4> pos{"pos"}, force{"force"}, torque{"torque"}, dir{"dir"};
  Property<real3> vel{"vel"};
  Property<real> energy{"energy"}, virial{"virial"}, mass{"mass"}, radius{"radius"}, charge{"charge"};
  Property<int> id{"id"};
  // one write-requested signal per property + reorder + number-of-particles (ParticleData.cuh:182-194): modules hold the connections
  using VoidSignal = signal<void(void)>;
  shared_ptr<VoidSignal> reorderSignal = make_shared<VoidSignal>();
  shared_ptr<signal<void(int)>> numParticlesChangedSignal = make_shared<signal<void(int)>>();
  shared_ptr<VoidSignal> posWriteRequestedSignal = make_shared<VoidSignal>(), forceWriteRequestedSignal = make_shared<VoidSignal>(),
                         torqueWriteRequestedSignal = make_shared<VoidSignal>(), dirWriteRequestedSignal = make_shared<VoidSignal>(),
                         velWriteRequestedSignal = make_shared<VoidSignal>(), energyWriteRequestedSignal = make_shared<VoidSignal>(),
                         virialWriteRequestedSignal = make_shared<VoidSignal>(), massWriteRequestedSignal = make_shared<VoidSignal>(),
                         radiusWriteRequestedSignal = make_shared<VoidSignal>(), chargeWriteRequestedSignal = make_shared<VoidSignal>(),
                         idWriteRequestedSignal = make_shared<VoidSignal>();
  static void announce(const shared_ptr<VoidSignal> &sig, access::mode m) {  // ParticleData.cuh:232-239
    if (m == access::write || m == access::readwrite) (*sig)();
  }
  struct Hints { Box hash_box = Box(real(128)); real3 hash_cutOff = make_real3(10.0); } hints;  // ParticleData.cuh:164-169
  template <class T> property_ptr<T> get(Property<T> &p, access::location l, access::mode m) { return p.data(l, m); }
public:
  ParticleData(int N, shared_ptr<System> s = nullptr) : numberParticles(N), sys(s ? s : make_shared<System>()) {
    for (auto *p : {&pos, &force}) p->resize(N);
    vel.resize(N); energy.resize(N); virial.resize(N); mass.resize(N); radius.resize(N); id.resize(N);
    auto ids = id.data(access::cpu, access::write);  // ParticleData.cuh:471-490
    for (int i = 0; i < N; ++i) ids[i] = i;
  }
  ParticleData(shared_ptr<System> s, int N) : ParticleData(N, s) {}  // ParticleData.cuh:215-216
  ~ParticleData() { if (sorter) uammd_celllist_destroy(sorter); }
  ParticleData(const ParticleData &) = delete;
  ParticleData &operator=(const ParticleData &) = delete;
  shared_ptr<System> getSystem() { return sys; }
  int getNumParticles() const { return numberParticles; }
  property_ptr<real4> getPos(access::location l, access::mode m) { announce(posWriteRequestedSignal, m); return pos.data(l, m); }
  property_ptr<real4> getForce(access::location l, access::mode m) { announce(forceWriteRequestedSignal, m); return force.data(l, m); }
  property_ptr<real3> getVel(access::location l, access::mode m) { announce(velWriteRequestedSignal, m); return vel.data(l, m); }
  // torques (real4) and orientation quaternions (n, vx, vy, vz) are allocated on first request, as every UAMMD property
  property_ptr<real4> getTorque(access::location l, access::mode m) {
    if (!torque.isAllocated()) torque.resize(numberParticles);
    announce(torqueWriteRequestedSignal, m);
    return torque.data(l, m);
  }
  property_ptr<real4> getDir(access::location l, access::mode m) {
    if (!dir.isAllocated()) {
      dir.resize(numberParticles);
      auto d = dir.data(access::cpu, access::write);
      for (int i = 0; i < numberParticles; ++i) d[i] = make_real4(1, 0, 0, 0);
    }
    announce(dirWriteRequestedSignal, m);
    return dir.data(l, m);
  }
  property_ptr<real4> getTorqueIfAllocated(access::location l, access::mode m) { return torque.isAllocated() ? getTorque(l, m) : property_ptr<real4>(); }
  property_ptr<real4> getDirIfAllocated(access::location l, access::mode m) { return dir.isAllocated() ? getDir(l, m) : property_ptr<real4>(); }
  bool isTorqueAllocated() const { return torque.isAllocated(); }
  bool isDirAllocated() const { return dir.isAllocated(); }
  property_ptr<real> getEnergy(access::location l, access::mode m) { announce(energyWriteRequestedSignal, m); return energy.data(l, m); }
  property_ptr<real> getVirial(access::location l, access::mode m) { announce(virialWriteRequestedSignal, m); return virial.data(l, m); }
  property_ptr<real> getMass(access::location l, access::mode m) { announce(massWriteRequestedSignal, m); return mass.data(l, m); }
  property_ptr<real> getRadius(access::location l, access::mode m) { announce(radiusWriteRequestedSignal, m); return radius.data(l, m); }
  property_ptr<real> getCharge(access::location l, access::mode m) {
    if (!charge.isAllocated()) charge.resize(numberParticles);
    announce(chargeWriteRequestedSignal, m);
    return charge.data(l, m);
  }
  property_ptr<int> getId(access::location l, access::mode m) {
    if (m != access::read) idOrderValid = false;
    announce(idWriteRequestedSignal, m);
    return id.data(l, m);
  }
  // ParticleData::getIdOrderedIndices (ParticleData.cuh:298-323): the particle with id i sits at row getIdOrderedIndices()[i]; valid
  // until the next sortParticles
  const int *getIdOrderedIndices(access::location dev) {
    if (!idOrderValid) {
      id2indexHost.assign(numberParticles, 0);
      {
        auto ids = id.data(access::cpu, access::read);
        for (int i = 0; i < numberParticles; ++i) id2indexHost[ids[i]] = i;
      }
      id2indexDevice.resize(numberParticles);
      if (numberParticles)
        detail::hipCheck(hipMemcpy(id2indexDevice.d, id2indexHost.data(), sizeof(int) * numberParticles, hipMemcpyHostToDevice), "hipMemcpy");
      idOrderValid = true;
    }
    return dev == access::gpu ? id2indexDevice.d : id2indexHost.data();
  }
  // get<Name>IfAllocated (ParticleData.cuh:247-259): the getter (signal included) when the property exists, an empty handle otherwise
  property_ptr<real4> getPosIfAllocated(access::location l, access::mode m) { return pos.isAllocated() ? getPos(l, m) : property_ptr<real4>(); }
  property_ptr<real4> getForceIfAllocated(access::location l, access::mode m) { return force.isAllocated() ? getForce(l, m) : property_ptr<real4>(); }
  property_ptr<real3> getVelIfAllocated(access::location l, access::mode m) { return vel.isAllocated() ? getVel(l, m) : property_ptr<real3>(); }
  property_ptr<real> getEnergyIfAllocated(access::location l, access::mode m) { return energy.isAllocated() ? getEnergy(l, m) : property_ptr<real>(); }
  property_ptr<real> getVirialIfAllocated(access::location l, access::mode m) { return virial.isAllocated() ? getVirial(l, m) : property_ptr<real>(); }
  property_ptr<real> getChargeIfAllocated(access::location l, access::mode m) { return charge.isAllocated() ? getCharge(l, m) : property_ptr<real>(); }
  property_ptr<real> getMassIfAllocated(access::location l, access::mode m) { return mass.isAllocated() ? getMass(l, m) : property_ptr<real>(); }
  property_ptr<real> getRadiusIfAllocated(access::location l, access::mode m) { return radius.isAllocated() ? getRadius(l, m) : property_ptr<real>(); }
  bool isPosAllocated() const { return pos.isAllocated(); }
  bool isVelAllocated() const { return vel.isAllocated(); }
  bool isForceAllocated() const { return force.isAllocated(); }
  bool isMassAllocated() const { return mass.isAllocated(); }
  bool isRadiusAllocated() const { return radius.isAllocated(); }
  // ParticleData.cuh:364-381: the signals themselves; `auto c = pd->getReorderSignal()->connect(slot); ... c.disconnect();`
  shared_ptr<signal<void(void)>> getReorderSignal() { return reorderSignal; }
  shared_ptr<signal<void(int)>> getNumParticlesChangedSignal() { return numParticlesChangedSignal; }
  shared_ptr<signal<void(void)>> getPosWriteRequestedSignal() { return posWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getForceWriteRequestedSignal() { return forceWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getTorqueWriteRequestedSignal() { return torqueWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getDirWriteRequestedSignal() { return dirWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getVelWriteRequestedSignal() { return velWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getEnergyWriteRequestedSignal() { return energyWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getVirialWriteRequestedSignal() { return virialWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getMassWriteRequestedSignal() { return massWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getRadiusWriteRequestedSignal() { return radiusWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getChargeWriteRequestedSignal() { return chargeWriteRequestedSignal; }
  shared_ptr<signal<void(void)>> getIdWriteRequestedSignal() { return idWriteRequestedSignal; }
  void hintSortByHash(Box hash_box, real3 hash_cutOff) { hints.hash_box = hash_box; hints.hash_cutOff = hash_cutOff; }
  // ParticleData::sortParticles (ParticleData.cuh:492-522): Morton order on the hint grid, every allocated property
  void sortParticles(hipStream_t st = 0) {
    // the sorter's cell list is kept between calls: creating one per sort costs ~10 ms of device allocations
    if (!sorter) detail::check(uammd_celllist_create(&sorter));
    uammd_celllist *cl = sorter;
    float L[3]; int per[3];
    hints.hash_box.toArrays(L, per);
    int cd[3] = {(int)(L[0] / hints.hash_cutOff.x), (int)(L[1] / hints.hash_cutOff.y), (int)(L[2] / hints.hash_cutOff.z)};
    if (cd[2] == 0) cd[2] = 1;
    {
      auto p = pos.data(access::gpu, access::read);
#if defined(DOUBLE_PRECISION)
      // the order is a memory-locality order, not a result: the keys come from the positions rounded to single precision
      detail::DeviceArray<float> p32(4 * (size_t)numberParticles);
      detail::check(uammd_convert_f64_to_f32((const double *)p.raw(), p32.d, 4 * (size_t)numberParticles, (void *)st));
      detail::check(uammd_celllist_update(cl, p32.d, numberParticles, L, per, cd, (void *)st));
      detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");   // (p32 goes back to the pool here)
#else
      detail::check(uammd_celllist_update(cl, (const float *)p.raw(), numberParticles, L, per, cd, (void *)st));
#endif
    }
    uammd_celllist_data d;
    detail::check(uammd_celllist_get(cl, &d));
    reorder(pos, d.d_groupIndex, st); reorder(force, d.d_groupIndex, st); reorder(vel, d.d_groupIndex, st);
    reorder(torque, d.d_groupIndex, st); reorder(dir, d.d_groupIndex, st);
    reorder(energy, d.d_groupIndex, st); reorder(virial, d.d_groupIndex, st); reorder(mass, d.d_groupIndex, st);
    reorder(radius, d.d_groupIndex, st); reorder(charge, d.d_groupIndex, st); reorder(id, d.d_groupIndex, st);
    detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");
    idOrderValid = false;
    (*reorderSignal)();  // emitReorder, ParticleData.cu:519 (the positions moved in memory: every listener of this header's modules that
                         // caches something per row listens to this signal as well as to the position writes)
  }
private:
  uammd_celllist *sorter = nullptr;
  std::vector<int> id2indexHost;
  detail::DeviceArray<int> id2indexDevice;
  bool idOrderValid = false;
  template <class T> void reorder(Property<T> &p, const int *d_index, hipStream_t st) {
    if (!p.isAllocated()) return;
    auto h = p.data(access::gpu, access::readwrite);
    detail::DeviceArray<T> alt(numberParticles);
    detail::check(uammd_gather(h.raw(), d_index, alt.d, numberParticles, (int)sizeof(T), (void *)st));
    detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");
    h.release();
    p.swapDeviceBuffer(alt);
  }
};

// ---- utils/checkpoint.h:29-76: the text checkpoint of a real UAMMD run -----------------------------------------------------------
// "# version V" / "# N" / one "# Name" block per allocated property in the order of ParticleData's property list, N lines each in
// particle-ID order, default stream formatting.  Id is not written.
namespace detail {
inline std::ostream &put(std::ostream &o, const real &v) { return o << v; }
inline std::ostream &put(std::ostream &o, const real3 &v) { return o << v.x << " " << v.y << " " << v.z; }
inline std::ostream &put(std::ostream &o, const real4 &v) { return o << v.x << " " << v.y << " " << v.z << " " << v.w; }
inline std::istream &get(std::istream &i, real &v) { return i >> v; }
inline std::istream &get(std::istream &i, real3 &v) { return i >> v.x >> v.y >> v.z; }
inline std::istream &get(std::istream &i, real4 &v) { return i >> v.x >> v.y >> v.z >> v.w; }
template <class T> void saveBlock(property_ptr<T> prop, const std::vector<int> &id2index, const char *name, std::ostream &out) {
  if (!prop.raw()) return;
  out << "# " << name << std::endl;
  for (size_t i = 0; i < prop.size(); ++i) { put(out, prop[id2index[i]]); out << "\n"; }
}
template <class T> void readBlock(property_ptr<T> prop, std::istream &in) {
  for (size_t i = 0; i < prop.size(); ++i) get(in, prop[i]);
}
}  // namespace detail
inline void saveParticleData(const std::string &fileName, shared_ptr<ParticleData> pd) {
  std::ofstream out(fileName);
  out << "# version " << "3.0.0" << std::endl;  // UAMMD_VERSION, global/defines.h:7
  const int N = pd->getNumParticles();
  out << "# " << N << std::endl;
  std::vector<int> id2index(N);
  {
    auto id = pd->getId(access::cpu, access::read);
    for (int i = 0; i < N; ++i) id2index[id[i]] = i;  // ParticleData::getIdOrderedIndices
  }
  detail::saveBlock(pd->getPosIfAllocated(access::cpu, access::read), id2index, "Pos", out);
  detail::saveBlock(pd->getMassIfAllocated(access::cpu, access::read), id2index, "Mass", out);
  detail::saveBlock(pd->getForceIfAllocated(access::cpu, access::read), id2index, "Force", out);
  detail::saveBlock(pd->getVirialIfAllocated(access::cpu, access::read), id2index, "Virial", out);
  detail::saveBlock(pd->getEnergyIfAllocated(access::cpu, access::read), id2index, "Energy", out);
  detail::saveBlock(pd->getVelIfAllocated(access::cpu, access::read), id2index, "Vel", out);
  detail::saveBlock(pd->getRadiusIfAllocated(access::cpu, access::read), id2index, "Radius", out);
  detail::saveBlock(pd->getChargeIfAllocated(access::cpu, access::read), id2index, "Charge", out);
  detail::saveBlock(pd->getTorqueIfAllocated(access::cpu, access::read), id2index, "Torque", out);
  detail::saveBlock(pd->getDirIfAllocated(access::cpu, access::read), id2index, "Dir", out);
}
inline shared_ptr<ParticleData> restoreParticleData(const std::string &fileName, shared_ptr<System> sys) {
  std::ifstream in(fileName);
  std::string str;
  in >> str >> str >> str;
  if (str != "3.0.0") System::log<System::WARNING>("This restore file was saved with a different UAMMD version (%s)", str.c_str());
  int N = 0;
  in >> str >> N;
  auto pd = make_shared<ParticleData>(N, sys);
  while (in >> str) {
    std::string name;
    in >> name;
    if (name == "Pos") detail::readBlock(pd->getPos(access::cpu, access::write), in);
    else if (name == "Mass") detail::readBlock(pd->getMass(access::cpu, access::write), in);
    else if (name == "Force") detail::readBlock(pd->getForce(access::cpu, access::write), in);
    else if (name == "Virial") detail::readBlock(pd->getVirial(access::cpu, access::write), in);
    else if (name == "Energy") detail::readBlock(pd->getEnergy(access::cpu, access::write), in);
    else if (name == "Vel") detail::readBlock(pd->getVel(access::cpu, access::write), in);
    else if (name == "Radius") detail::readBlock(pd->getRadius(access::cpu, access::write), in);
    else if (name == "Charge") detail::readBlock(pd->getCharge(access::cpu, access::write), in);
    else if (name == "Torque") detail::readBlock(pd->getTorque(access::cpu, access::write), in);
    else if (name == "Dir") detail::readBlock(pd->getDir(access::cpu, access::write), in);
    else if (name == "AngVel") { real4 skip; for (int i = 0; i < N; ++i) detail::get(in, skip); }  // not a property of this build
  }
  return pd;
}

// ---- ParticleGroup (ParticleData/ParticleGroup.cuh:60-135 selectors, :170-379 group): a subset tracked by particle ID ------------
namespace particle_selector {
struct All { bool isSelected(int, shared_ptr<ParticleData> &) { return true; } };
struct None { bool isSelected(int, shared_ptr<ParticleData> &) { return false; } };
class IDRange {  // ids in [firstID, lastID]
  int firstID, lastID;
public:
  IDRange(int first, int last) : firstID(first), lastID(last) {}
  bool isSelected(int particleIndex, shared_ptr<ParticleData> &pd) {
    const int id = pd->getId(access::cpu, access::read)[particleIndex];
    return id >= firstID && id <= lastID;
  }
};
class Type {  // pos.w
  std::vector<int> typesToSelect;
public:
  Type(int type) : typesToSelect({type}) {}
  Type(std::vector<int> types) : typesToSelect(std::move(types)) {}
  bool isSelected(int particleIndex, shared_ptr<ParticleData> &pd) {
    const int type_i = (int)pd->getPos(access::cpu, access::read)[particleIndex].w;
    for (int t : typesToSelect) if (t == type_i) return true;
    return false;
  }
};
}  // namespace particle_selector

class ParticleGroup {
  shared_ptr<ParticleData> pd;
  std::string name;
  bool allParticlesInGroup = false;
  std::vector<int> myParticlesIds;      // sorted by id (the reference keeps the members id-ordered)
  std::vector<int> h_index;             // current ParticleData indices of the members
  detail::DeviceArray<int> d_index;
  bool needsIndexUpdate = true;
  void updateIndices() {  // ParticleGroup_ns::updateGroupIndices, ParticleGroup.cuh:140-153: index = id2index[id]
    if (!needsIndexUpdate || allParticlesInGroup) return;
    const int N = pd->getNumParticles();
    std::vector<int> id2index(N);
    {
      auto id = pd->getId(access::cpu, access::read);
      for (int i = 0; i < N; ++i) id2index[id[i]] = i;
    }
    h_index.resize(myParticlesIds.size());
    for (size_t k = 0; k < myParticlesIds.size(); ++k) h_index[k] = id2index[myParticlesIds[k]];
    d_index.resize(h_index.size());
    if (!h_index.empty()) detail::hipCheck(hipMemcpy(d_index.d, h_index.data(), sizeof(int) * h_index.size(), hipMemcpyHostToDevice), "hipMemcpy");
    needsIndexUpdate = false;
  }
  scoped_connection reorderConnection;  // ParticleGroup.cuh:188,212: dropped with the group
  void init() { reorderConnection = pd->getReorderSignal()->connect([this]() { needsIndexUpdate = true; }); }
public:
  // index[i] of member i; the "All" group is the identity and raw() is null, which is what the C ABI takes for "all particles"
  struct IndexIterator {
    const int *ptr;
    int operator[](int i) const { return ptr ? ptr[i] : i; }
    const int *raw() const { return ptr; }
  };
  template <class T> struct PropertyIterator {  // property[index[i]] (host side)
    T *base;
    IndexIterator index;
    T &operator[](int i) const { return base[index[i]]; }
  };
  ParticleGroup(shared_ptr<ParticleData> pd, std::string name = std::string("noName")) : pd(pd), name(std::move(name)), allParticlesInGroup(true) { init(); }
  template <class ParticleSelector>
  ParticleGroup(ParticleSelector selector, shared_ptr<ParticleData> pd, std::string name = std::string("noName")) : pd(pd), name(std::move(name)) {
    const int N = pd->getNumParticles();
    std::vector<int> ids;
    {
      auto id = pd->getId(access::cpu, access::read);
      for (int i = 0; i < N; ++i) ids.push_back(id[i]);
    }
    for (int i = 0; i < N; ++i) if (selector.isSelected(i, pd)) myParticlesIds.push_back(ids[i]);
    std::sort(myParticlesIds.begin(), myParticlesIds.end());
    allParticlesInGroup = (int)myParticlesIds.size() == N;
    init();
  }
  template <class InputIterator>
  ParticleGroup(InputIterator begin, InputIterator end, shared_ptr<ParticleData> pd, std::string name = std::string("noName"))
      : pd(pd), name(std::move(name)), myParticlesIds(begin, end) {
    std::sort(myParticlesIds.begin(), myParticlesIds.end());
    allParticlesInGroup = (int)myParticlesIds.size() == pd->getNumParticles();
    init();
  }
  ParticleGroup(const ParticleGroup &) = delete;
  // ParticleGroup.cuh:221-227,498-600: an emptied group filled again by hand — by id, or by the particles' CURRENT indices in the
  // ParticleData (translated to ids at once: the membership is by id and survives reorders); `loc` says where the array lives.
  // Members keep the order they were added in.
  void clear() { myParticlesIds.clear(); allParticlesInGroup = false; needsIndexUpdate = true; }
  void addParticlesById(access::location loc, const int *ids, int N) {
    if (N <= 0) return;
    if (allParticlesInGroup) { myParticlesIds.resize(pd->getNumParticles()); for (size_t k = 0; k < myParticlesIds.size(); ++k) myParticlesIds[k] = (int)k; }
    const size_t before = myParticlesIds.size();
    myParticlesIds.resize(before + N);
    if (loc == access::gpu) detail::hipCheck(hipMemcpy(myParticlesIds.data() + before, ids, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost), "hipMemcpy");
    else std::copy(ids, ids + N, myParticlesIds.begin() + before);
    allParticlesInGroup = false;
    needsIndexUpdate = true;
  }
  void addParticlesByCurrentIndex(access::location loc, const int *indices, int N) {
    if (N <= 0) return;
    std::vector<int> idx(N);
    if (loc == access::gpu) detail::hipCheck(hipMemcpy(idx.data(), indices, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost), "hipMemcpy");
    else std::copy(indices, indices + N, idx.begin());
    {
      auto id = pd->getId(access::cpu, access::read);
      for (int &v : idx) v = id[v];
    }
    addParticlesById(access::cpu, idx.data(), N);
  }
  int getNumberParticles() const { return allParticlesInGroup ? pd->getNumParticles() : (int)myParticlesIds.size(); }
  shared_ptr<ParticleData> getParticleData() { return pd; }
  std::string getName() const { return name; }
  bool isAll() const { return allParticlesInGroup; }
  IndexIterator getIndexIterator(access::location loc) {
    updateIndices();
    if (allParticlesInGroup) return IndexIterator{nullptr};
    return IndexIterator{loc == access::gpu ? d_index.d : h_index.data()};
  }
  const int *getIndicesRawPtr(access::location loc) { return getIndexIterator(loc).raw(); }
  template <class T> PropertyIterator<T> getPropertyIterator(const property_ptr<T> &prop) {
    return PropertyIterator<T>{prop.raw(), getIndexIterator(prop.location())};
  }
};

namespace detail {
// property rows of the members of a group as one contiguous device array, in the group's order (what pg->getPropertyIterator(prop) is to
// a kernel of the reference): the property's own array for the group of all particles (pg == nullptr or pg->isAll()), a gathered copy in
// `buf` otherwise.  The modules below run their solvers on these arrays and scatter through the group's index in their update kernels.
template <class T> inline const T *groupRows(const T *all, ParticleGroup *pg, DeviceArray<T> &buf, hipStream_t st) {
  if (!all || !pg || pg->isAll()) return all;
  const int n = pg->getNumberParticles();
  if (buf.size() != (size_t)n) buf.resize(n);
  check(uammd_gather(all, pg->getIndicesRawPtr(access::gpu), buf.d, n, (int)sizeof(T), (void *)st));
  return buf.d;
}
// ... and the way back for the rows a module changed in place: all[index[i]] = rows[i] (nothing to do when rows IS the property)
template <class T> inline void scatterRows(const T *rows, T *all, ParticleGroup *pg, hipStream_t st) {
  if (!rows || rows == all || !pg || pg->isAll()) return;
  check(uammd_scatter(rows, pg->getIndicesRawPtr(access::gpu), all, pg->getNumberParticles(), (int)sizeof(T), (void *)st));
}
inline shared_ptr<ParticleGroup> subsetOrNull(const shared_ptr<ParticleGroup> &pg) { return (pg && !pg->isAll()) ? pg : nullptr; }
}  // namespace detail

// ---- misc/ParameterUpdatable.h:72-80, Interactor, Integrator ------------------------------------------------------------------
class ParameterUpdatable {
public:
  virtual ~ParameterUpdatable() = default;
  virtual void updateTimeStep(real) {}
  virtual void updateSimulationTime(real) {}
  virtual void updateBox(Box) {}
  virtual void updateTemperature(real) {}
  virtual void updateViscosity(real) {}
};

class Interactor : public virtual ParameterUpdatable {
protected:
  shared_ptr<ParticleData> pd;
  shared_ptr<System> sys;
  std::string name;
  shared_ptr<ParticleGroup> pg;        // the group the module acts on — ALWAYS valid, as in the reference (Interactor.cuh:46-60: built from a
                                       // ParticleData it is the group of all particles); user classes derived from this one say pg->...
  shared_ptr<ParticleGroup> subgroup;  // the same group when it is a proper subset, nullptr for all the particles (what the code below branches on)
public:
  struct Computables { bool force = false, energy = false, virial = false, stress = false; };
  Interactor(shared_ptr<ParticleData> pd, std::string name = "noName")
      : pd(pd), sys(pd->getSystem()), name(std::move(name)), pg(std::make_shared<ParticleGroup>(pd, "All")) {}
  Interactor(shared_ptr<ParticleGroup> pg, std::string name = "noName")
      : pd(pg->getParticleData()), sys(pd->getSystem()), name(std::move(name)), pg(pg), subgroup(pg->isAll() ? nullptr : pg) {}
  virtual ~Interactor() = default;
  virtual void sum(Computables comp, hipStream_t st = 0) = 0;
  std::string getName() { return name; }
  // Not part of UAMMD's interface: an Interactor that IS PairForces<Potential::LJ, CellList> on every particle lets
  // VerletNVT::GronbechJensen run its whole step through one fused library call (uammd_verletnvt_gj_lj_step, bit-identical to the
  // three-call sequence).  Everything else answers false and the integrator does what the reference does.
  struct FusedGronbechJensen {
    float *pos, *vel, *force; const float *mass; float defaultMass; int N; float dt, friction; bool is2D; float noiseAmplitude;
    uint stepNum, seed; hipStream_t st;
  };
  virtual bool fusedGronbechJensenStep(const FusedGronbechJensen &) { return false; }
};

class Integrator {
protected:
  shared_ptr<ParticleData> pd;
  shared_ptr<System> sys;
  std::string name;
  std::vector<shared_ptr<Interactor>> interactors;
  std::vector<shared_ptr<ParameterUpdatable>> updatables;
  shared_ptr<ParticleGroup> pg;        // ALWAYS valid (Integrator.cuh:41-55), see Interactor
  shared_ptr<ParticleGroup> subgroup;  // nullptr = all the particles
public:
  Integrator(shared_ptr<ParticleData> pd, std::string name = "noName")
      : pd(pd), sys(pd->getSystem()), name(std::move(name)), pg(std::make_shared<ParticleGroup>(pd, "All")) {}
  Integrator(shared_ptr<ParticleGroup> pg, std::string name = "noName")
      : pd(pg->getParticleData()), sys(pd->getSystem()), name(std::move(name)), pg(pg), subgroup(pg->isAll() ? nullptr : pg) {}
  virtual ~Integrator() = default;
  virtual void forwardTime() = 0;
  virtual real sumEnergy() { return 0; }
protected:
  // the particles this integrator moves: the members of its group (all of them without one)
  int groupSize() const { return subgroup ? subgroup->getNumberParticles() : pd->getNumParticles(); }
  const int *groupIndex() { return subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr; }
  void resetGroupForces(hipStream_t st) {  // thrust::fill(forceGroup, forceGroup + N, real4()) of the reference's integrators
    auto force = pd->getForce(access::gpu, access::write);
    if (subgroup) detail::check(uammd_fill_zero_indexed(force.raw(), groupIndex(), groupSize(), (int)sizeof(real4), (void *)st));
    else detail::check(uammd_fill_zero(force.raw(), sizeof(real4) * force.size(), (void *)st));
  }
public:
  void addInteractor(shared_ptr<Interactor> an_interactor) { interactors.push_back(an_interactor); addUpdatable(an_interactor); }
  std::vector<shared_ptr<Interactor>> getInteractors() { return interactors; }
  // (Integrator.cuh:109-124: the reference keeps a std::set — an object added twice, e.g. as an interactor and as an updatable, hears once)
  void addUpdatable(shared_ptr<ParameterUpdatable> u) { if (std::find(updatables.begin(), updatables.end(), u) == updatables.end()) updatables.push_back(u); }
  std::vector<shared_ptr<ParameterUpdatable>> getUpdatables() { return updatables; }
};

#if !defined(DOUBLE_PRECISION)   // (single-precision backends only: see PRECISION at the top)
// ---- library mode: CellListBase / BasicNeighbourListBase (no ParticleData) ---------------------------------------------------------
// Interactor/NeighbourList/CellList/CellListBase.cuh:97-172 and BasicList/BasicListBase.cuh:76-215, as
// examples/uammd_as_a_library/neighbour_list.cu:159-167 uses them: positions from any iterator, a Grid (or a Box and a cut-off), a stream.
namespace detail {
// the positions as one real4 array on the device: a raw device pointer goes through untouched; any other iterator (thrust device
// iterators, transform / permutation iterators over real3 or real4) is evaluated into `staged` — that needs device code: hipcc only
inline const real4 *stagePositions(const real4 *pos, int, DeviceArray<real4> &, hipStream_t) { return pos; }
inline const real4 *stagePositions(real4 *pos, int, DeviceArray<real4> &, hipStream_t) { return pos; }
#if defined(__HIPCC__)
struct ToReal4 {  // CellListBase.cuh:60-66
  template <class V> __host__ __device__ real4 operator()(const V &v) const { return make_real4(v); }
};
template <class PositionIterator>
inline const real4 *stagePositions(PositionIterator pos, int n, DeviceArray<real4> &staged, hipStream_t st) {
  if (staged.size() != (size_t)n) staged.resize(n);
  thrust::transform(thrust::hip::par.on(st), pos, pos + n, staged.d, ToReal4());
  return staged.d;
}
#else
template <class PositionIterator>
inline const real4 *stagePositions(PositionIterator, int, DeviceArray<real4> &, hipStream_t) {
  static_assert(sizeof(PositionIterator) == 0, "a position iterator other than a real4 device pointer needs device code: compile this TU with hipcc");
  return nullptr;
}
#endif
}  // namespace detail

class CellListBase {
protected:
  uammd_celllist *h = nullptr;
  detail::DeviceArray<real4> staged;
  Grid grid;
public:
  CellListBase() { detail::check(uammd_celllist_create(&h)); }
  CellListBase(const CellListBase &) = delete;
  CellListBase &operator=(const CellListBase &) = delete;
  ~CellListBase() { uammd_celllist_destroy(h); }
  // CellListBase.cuh:124-141.  The grid is the caller's: no createUpdateGrid here (that is CellList's, CellList.cuh:100-126)
  template <class PositionIterator> void update(PositionIterator pos, int numberParticles, Grid in_grid, hipStream_t st = 0) {
    grid = in_grid;
    float L[3]; int per[3];
    grid.box.toArrays(L, per);
    const int cd[3] = {grid.cellDim.x, grid.cellDim.y, grid.cellDim.z};
    const real4 *p = detail::stagePositions(pos, numberParticles, staged, st);
    detail::check(uammd_celllist_update(h, (const float *)p, numberParticles, L, per, cd, (void *)st));
  }
  // CellListBase.cuh:145-172: the reference's field names, on top of the C ABI's POD (what the device-side NeighbourContainer takes)
  struct CellListData : uammd_celllist_data {
    const uint *cellStart = nullptr;
    const int *cellEnd = nullptr;
    const real4 *sortPos = nullptr;
    const int *groupIndex = nullptr;
    Grid grid;
    CellListData() : uammd_celllist_data() {}
    explicit CellListData(const uammd_celllist_data &d)
        : uammd_celllist_data(d), cellStart(d.d_cellStart), cellEnd(d.d_cellEnd), sortPos((const real4 *)d.d_sortPos), groupIndex(d.d_groupIndex) {
      Box b(make_real3(d.boxSize[0], d.boxSize[1], d.boxSize[2]));
      b.setPeriodicity(d.periodic[0], d.periodic[1], d.periodic[2]);
      grid = Grid(b, make_int3(d.cellDim[0], d.cellDim[1], d.cellDim[2]));
    }
  };
  CellListData getCellList() {
    uammd_celllist_data d;
    detail::check(uammd_celllist_get(h, &d));
    return CellListData(d);
  }
  uammd_celllist *handle() { return h; }
};

class BasicNeighbourListBase {
protected:
  uammd_verletlist *h = nullptr;
  detail::DeviceArray<real4> staged;
  real currentCutOff = 0;
  Box currentBox;
public:
  // the list of a BasicNeighbourListBase holds exactly the pairs within the cut-off at the time of update (no skin) and is refilled by
  // every update: the Verlet list of the C ABI with a cut-off multiplier of one, forced
  BasicNeighbourListBase() {
    detail::check(uammd_verletlist_create(&h));
    detail::check(uammd_verletlist_set_cutoff_multiplier(h, real(1.0)));
  }
  BasicNeighbourListBase(const BasicNeighbourListBase &) = delete;
  BasicNeighbourListBase &operator=(const BasicNeighbourListBase &) = delete;
  ~BasicNeighbourListBase() { uammd_verletlist_destroy(h); }
  template <class PositionIterator> void update(PositionIterator pos, int numberParticles, Box box, real cutOff, hipStream_t st = 0) {  // :131-141
    currentBox = box;
    currentCutOff = cutOff;
    float L[3]; int per[3];
    box.toArrays(L, per);
    const real4 *p = detail::stagePositions(pos, numberParticles, staged, st);
    detail::check(uammd_verletlist_force_next_update(h));
    detail::check(uammd_verletlist_update(h, (const float *)p, numberParticles, L, per, cutOff, (void *)st, nullptr));
  }
  // particleStride[i] is the distance between consecutive neighbours of particle i in neighbourList (BasicListBase.cuh:90-103: a
  // constant, the number of particles): neighbour k of sorted particle i is neighbourList[particleStride[i] * k + i]
  struct StrideIterator {
    int stride = 0;
    UAMMD_HOSTDEV int operator[](int) const { return stride; }
    UAMMD_HOSTDEV int operator*() const { return stride; }
  };
  struct BasicNeighbourListData : uammd_verletlist_data {  // :144-152
    const int *neighbourList = nullptr;
    const int *numberNeighbours = nullptr;
    const real4 *sortPos = nullptr;
    const int *groupIndex = nullptr;
    StrideIterator particleStride;
    BasicNeighbourListData() : uammd_verletlist_data() {}
    explicit BasicNeighbourListData(const uammd_verletlist_data &d)
        : uammd_verletlist_data(d), neighbourList(d.d_neighbourList), numberNeighbours(d.d_numberNeighbours), sortPos((const real4 *)d.d_sortPos),
          groupIndex(d.d_groupIndex) { particleStride.stride = d.particleStride; }
  };
  BasicNeighbourListData getBasicNeighbourList(hipStream_t = 0) {
    uammd_verletlist_data d;
    detail::check(uammd_verletlist_get(h, &d));
    return BasicNeighbourListData(d);
  }
  uammd_verletlist *handle() { return h; }
};
#if defined(__HIPCC__)
namespace CellList_ns { using NeighbourContainer = device::NeighbourContainer; }                  // CellList/NeighbourContainer.cuh:54
namespace BasicNeighbourList_ns { using NeighbourContainer = device::VerletNeighbourContainer; }  // BasicList/NeighbourContainer.cuh:42
#endif

// ---- CellList ----------------------------------------------------------------------------------------------------------------
class CellList {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;          // nullptr = all the particles
  detail::DeviceArray<real4> groupPos;   // pg->getPropertyIterator(pos): the members' positions, gathered
  uammd_celllist *h = nullptr;
  bool force_next_update = true;
  real3 currentCutOff{0, 0, 0};
  Box currentBox;
  scoped_connection posWriteConnection, reorderConnection;
public:
  using CellListData = CellListBase::CellListData;
  explicit CellList(shared_ptr<ParticleData> pd) : pd(pd) {
    detail::check(uammd_celllist_create(&h));
    // CellList.cuh:88,133-143: the connection is a member and is dropped in the destructor, so a list may die before its ParticleData
    posWriteConnection = pd->getPosWriteRequestedSignal()->connect([this]() { force_next_update = true; });  // CellList.cuh:94-98
    reorderConnection = pd->getReorderSignal()->connect([this]() { force_next_update = true; });
  }
  explicit CellList(shared_ptr<ParticleGroup> group) : CellList(group->getParticleData()) {
    if (!group->isAll()) pg = group;
  }
  const int *groupIndex() { return pg ? pg->getIndicesRawPtr(access::gpu) : nullptr; }
  CellList(const CellList &) = delete;
  ~CellList() {
    posWriteConnection.disconnect();
    reorderConnection.disconnect();
    uammd_celllist_destroy(h);
  }
  void update(Box box, real cutOff, hipStream_t st = 0) { update(box, make_real3(cutOff), st); }
  void update(Box box, real3 cutOff, hipStream_t st = 0) {
    const bool rebuild = force_next_update || cutOff.x != currentCutOff.x || cutOff.y != currentCutOff.y ||
                         cutOff.z != currentCutOff.z || box != currentBox;  // CellList.cuh:192-204
    if (!rebuild) return;
    currentBox = box;
    currentCutOff = cutOff;
    float L[3], Lo[3], rc[3] = {cutOff.x, cutOff.y, cutOff.z};
    int per[3], cd[3], po[3];
    box.toArrays(L, per);
    detail::check(uammd_celllist_create_grid(L, per, rc, cd, Lo, po));
    auto pos = pd->getPos(access::gpu, access::read);
    if (pg) {
      const int n = pg->getNumberParticles();
      groupPos.resize(n);
      detail::check(uammd_gather(pos.raw(), pg->getIndicesRawPtr(access::gpu), groupPos.d, n, (int)sizeof(real4), (void *)st));
      detail::check(uammd_celllist_update(h, (const float *)groupPos.d, n, Lo, po, cd, (void *)st));
    } else {
      detail::check(uammd_celllist_update(h, (const float *)pos.raw(), pd->getNumParticles(), Lo, po, cd, (void *)st));
    }
    force_next_update = false;
  }
  CellListData getCellList() { uammd_celllist_data d; detail::check(uammd_celllist_get(h, &d)); return CellListData(d); }
#if defined(__HIPCC__)
  // CellList.cuh:186-189: a forward iterator over the neighbours of a particle for the caller's own kernels (hipcc TUs)
  CellList_ns::NeighbourContainer getNeighbourContainer() { return CellList_ns::NeighbourContainer(getCellList()); }
#endif
  uammd_celllist *handle() { return h; }
  bool isAllParticles() const { return !pg; }
  // the fused MD step built the list as update(box, cutOff) would have: the lazy-update bookkeeping follows
  void fusedBuilt(Box box, real3 cutOff) { currentBox = box; currentCutOff = cutOff; force_next_update = false; }
};

// ---- VerletList (Interactor/NeighbourList/VerletList.cuh:83-201) ------------------------------------------------------------------
class VerletList {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;          // nullptr = all the particles
  detail::DeviceArray<real4> groupPos;   // pg->getPropertyIterator(pos): the members' positions, gathered
  uammd_verletlist *h = nullptr;
  bool forceNextUpdate = true;
  Box currentBox;
  real currentCutOff = 0;
  scoped_connection posWriteConnection, reorderConnection;  // :88
public:
  using VerletListData = BasicNeighbourListBase::BasicNeighbourListData;  // VerletListBase.cuh:176-186
  explicit VerletList(shared_ptr<ParticleGroup> group) : VerletList(group->getParticleData()) {  // :96-103
    if (!group->isAll()) pg = group;
  }
  explicit VerletList(shared_ptr<ParticleData> pd) : pd(pd) {
    detail::check(uammd_verletlist_create(&h));
    posWriteConnection = pd->getPosWriteRequestedSignal()->connect([this]() { forceNextUpdate = true; });       // :99-100, :171-175
    reorderConnection = pd->getReorderSignal()->connect([this]() { forceNextUpdate = true; uammd_verletlist_force_next_update(h); });  // :101-102, :177-182
  }
  VerletList(const VerletList &) = delete;
  ~VerletList() {  // :106-110
    posWriteConnection.disconnect();
    reorderConnection.disconnect();
    uammd_verletlist_destroy(h);
  }
  void update(Box box, real cutOff, hipStream_t st = 0) {  // :112-124
    const bool rebuild = forceNextUpdate || box != currentBox || cutOff != currentCutOff;
    forceNextUpdate = false;
    if (!rebuild) return;
    pd->hintSortByHash(box, make_real3(cutOff * real(0.5)));
    currentBox = box;
    currentCutOff = cutOff;
    float L[3]; int per[3];
    box.toArrays(L, per);
    // VerletList reads the positions with access::read: that does not raise the pos-write signal (Property access only)
    auto pos = pd->getPos(access::gpu, access::read);
    const int N = pg ? pg->getNumberParticles() : pd->getNumParticles();
    detail::check(uammd_verletlist_update(h, (const float *)detail::groupRows((const real4 *)pos.raw(), pg.get(), groupPos, st), N, L, per, cutOff,
                                          (void *)st, nullptr));
  }
  void update(Box box, real3 cutOff, hipStream_t st = 0) {  // :130-136
    if (cutOff.x != cutOff.y || cutOff.x != cutOff.z) throw std::runtime_error("[VerletList] Invalid argument");
    update(box, cutOff.x, st);
  }
  VerletListData getVerletList() { uammd_verletlist_data d; detail::check(uammd_verletlist_get(h, &d)); return VerletListData(d); }
#if defined(__HIPCC__)
  // VerletList.cuh:160-164
  BasicNeighbourList_ns::NeighbourContainer getNeighbourContainer() { return BasicNeighbourList_ns::NeighbourContainer(getVerletList()); }
#endif
  void setCutOffMultiplier(real newMultiplier) { forceNextUpdate = true; detail::check(uammd_verletlist_set_cutoff_multiplier(h, newMultiplier)); }
  int getNumberOfStepsSinceLastUpdate() { int s = 0; detail::check(uammd_verletlist_get_steps_since_last_update(h, &s)); return s; }
  uammd_verletlist *handle() { return h; }
  bool isAllParticles() const { return !pg; }
  const int *groupIndex() { return pg ? pg->getIndicesRawPtr(access::gpu) : nullptr; }
};

// ---- Potential::LJ ---------------------------------------------------------------------------------------------------------------
namespace Potential {
class LJ {
  std::vector<uammd_lj_pair_parameters> table;
  detail::DeviceArray<uammd_lj_pair_parameters> d_table;
  int ntypes = 1;
  real cutOff = 0;
  bool dirty = true;
public:
  struct InputPairParameters { real cutOff, sigma, epsilon; bool shift = false; };
  LJ() : table(1) {}
  void setPotParameters(int ti, int tj, InputPairParameters p) {  // ParameterHandler.cuh:17-37
    cutOff = std::max(p.cutOff, cutOff);
    const int nn = std::max(ntypes, std::max(ti, tj) + 1);
    if (nn != ntypes) {
      std::vector<uammd_lj_pair_parameters> tmp((size_t)nn * nn);
      for (int i = 0; i < ntypes; ++i) for (int j = 0; j < ntypes; ++j) tmp[i + nn * j] = table[i + ntypes * j];
      table.swap(tmp);
      ntypes = nn;
    }
    uammd_lj_pair_parameters q;
    detail::check(uammd_lj_process_pair_parameters(p.cutOff, p.sigma, p.epsilon, p.shift, &q));
    table[ti + ntypes * tj] = q;
    if (ti != tj) table[tj + ntypes * ti] = q;
    dirty = true;
  }
  real getCutOff() { return cutOff; }
  int getNumberTypes() const { return ntypes; }
  const uammd_lj_pair_parameters *deviceTable() {
    if (dirty) {
      d_table.resize(table.size());
      detail::hipCheck(hipMemcpy(d_table.d, table.data(), sizeof(table[0]) * table.size(), hipMemcpyHostToDevice), "hipMemcpy");
      detail::check(uammd_lj_table_changed());  // (the same address may now hold other parameters: the lists read it again)
      dirty = false;
    }
    return d_table.d;
  }
};
}  // namespace Potential

// ---- PairForces<Potential::LJ, CellList> -------------------------------------------------------------------------------------------
template <class MyPotential, class NL = CellList> class PairForces;
template <class NL> class PairForces<Potential::LJ, NL> : public Interactor {
  Box box;
  shared_ptr<Potential::LJ> pot;
  shared_ptr<NL> nl;
  // NL::transverseList(Radial<LJFunctor>::Transverser): one fused entry point per neighbour-list type
  static int transverse(uammd_celllist *h, const uammd_lj_pair_parameters *t, int nt, const float *L, const int *per, float *f,
                        float *e, float *v, const int *globalIndex, void *st) {
    return uammd_lj_transverse_celllist(h, t, nt, L, per, f, e, v, globalIndex, UAMMD_LJ_ALGO_AUTO, st);
  }
  static int transverse(uammd_verletlist *h, const uammd_lj_pair_parameters *t, int nt, const float *L, const int *per, float *f,
                        float *e, float *v, const int *globalIndex, void *st) {
    return uammd_lj_transverse_verletlist(h, t, nt, L, per, f, e, v, globalIndex, st);
  }
  template <class List> static shared_ptr<List> makeList(shared_ptr<ParticleData> pd, shared_ptr<ParticleGroup> pg, List *) {
    return pg ? make_shared<List>(pg) : make_shared<List>(pd);  // both lists take a group (CellList.cuh:132, VerletList.cuh:96)
  }
public:
  struct Parameters { Box box; shared_ptr<NL> nl = nullptr; };
  PairForces(shared_ptr<ParticleData> pd, Parameters par, shared_ptr<Potential::LJ> pot = make_shared<Potential::LJ>())
      : Interactor(pd, "PairForces"), box(par.box), pot(pot), nl(par.nl) { noteListChoice((NL *)nullptr); }
  // PairForces(pg, par, pot): forces among the members of the group only (PairForces.cuh:103-107)
  PairForces(shared_ptr<ParticleGroup> pg, Parameters par, shared_ptr<Potential::LJ> pot = make_shared<Potential::LJ>())
      : Interactor(pg, "PairForces"), box(par.box), pot(pot), nl(par.nl) {}
  void updateBox(Box b) override { box = b; }
private:
  // On this GPU the cell list rebuilt every step is the faster neighbour list for a liquid: 0.19 ms per step against 0.35 with the
  // Verlet list at 1e6 particles, rho* = 0.8 (a rebuild costs more than the steps it saves).  Said once, at MESSAGE level.
  static void noteListChoice(CellList *) {}
  static void noteListChoice(VerletList *) {
    static bool said = false;
    if (!said) System::log<System::MESSAGE>("[PairForces] VerletList chosen: on MI355X PairForces<LJ, CellList> is ~1.9x faster per step for dense liquids (DESIGN.md 5.2b)");
    said = true;
  }
  static bool fusedStep(shared_ptr<CellList> &list, shared_ptr<ParticleData> pd, shared_ptr<Potential::LJ> pot, Box box,
                        const FusedGronbechJensen &a) {
    if (!list) list = make_shared<CellList>(pd);
    if (!list->isAllParticles()) return false;
    float L[3], Lo[3]; int per[3], cd[3], po[3];
    box.toArrays(L, per);
    const real rcut = pot->getCutOff();
    const float rc[3] = {rcut, rcut, rcut};
    detail::check(uammd_celllist_create_grid(L, per, rc, cd, Lo, po));
    detail::check(uammd_verletnvt_gj_lj_step(list->handle(), a.pos, a.vel, a.force, a.mass, a.defaultMass, a.N, L, per, Lo, po, cd,
                                             pot->deviceTable(), pot->getNumberTypes(), a.dt, a.friction, a.is2D, a.noiseAmplitude,
                                             a.stepNum, a.seed, UAMMD_LJ_ALGO_AUTO, (void *)a.st));
    list->fusedBuilt(box, make_real3(rcut));
    return true;
  }
  template <class List> static bool fusedStep(shared_ptr<List> &, shared_ptr<ParticleData>, shared_ptr<Potential::LJ>, Box,
                                              const FusedGronbechJensen &) { return false; }  // (only the CellList is fused)
public:
  bool fusedGronbechJensenStep(const FusedGronbechJensen &a) override {
    const real rcut = pot->getCutOff();
    if (subgroup || (box.boxSize.x <= 3 * rcut && box.boxSize.y <= 3 * rcut && box.boxSize.z <= 3 * rcut)) return false;
    return fusedStep(nl, pd, pot, box, a);
  }
  void sum(Computables comp, hipStream_t st = 0) override {  // PairForces.cu:43-78
    float L[3]; int per[3];
    box.toArrays(L, per);
    const real rcut = pot->getCutOff();
    const int N = subgroup ? subgroup->getNumberParticles() : pd->getNumParticles();
    const bool useNL = !(box.boxSize.x <= 3 * rcut && box.boxSize.y <= 3 * rcut && box.boxSize.z <= 3 * rcut);
    if (useNL) {
      if (!nl) nl = makeList(pd, subgroup, (NL *)nullptr);
      nl->update(box, rcut, st);
    }
    const int *globalIndex = subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr;
    auto force = comp.force ? pd->getForce(access::gpu, access::readwrite) : property_ptr<real4>();
    auto energy = comp.energy ? pd->getEnergy(access::gpu, access::readwrite) : property_ptr<real>();
    auto virial = comp.virial ? pd->getVirial(access::gpu, access::readwrite) : property_ptr<real>();
    if (useNL) {
      detail::check(transverse(nl->handle(), pot->deviceTable(), pot->getNumberTypes(), L, per, (float *)force.raw(),
                               energy.raw(), virial.raw(), globalIndex, (void *)st));
    } else {
      // all pairs among the members: the kernel reads pos[globalIndex[t]] and writes force[globalIndex[t]]
      // (NBody over pg->getPropertyIterator(pos), PairForces.cu:49-53), so the UN-gathered array goes in.
      auto pos = pd->getPos(access::gpu, access::read);
      detail::check(uammd_lj_transverse_nbody((const float *)pos.raw(), N, pot->deviceTable(), pot->getNumberTypes(), L,
                                              per, (float *)force.raw(), energy.raw(), virial.raw(), globalIndex, (void *)st));
    }
  }
};

// ---- VerletNVT -----------------------------------------------------------------------------------------------------------------------
namespace VerletNVT {
class Basic : public Integrator {
public:
  struct Parameters { real temperature = 0, dt = 0, friction = 1.0; bool is2D = false, initVelocities = true; real mass = -1.0; };
protected:
  real noiseAmplitude, dt, temperature, friction, defaultMass;
  uint seed;
  bool is2D;
  int steps = 0;
  hipStream_t stream = 0;
  virtual int kernelKind() const { return 0; }
  void callIntegrate(int step) {
    const int N = subgroup ? subgroup->getNumberParticles() : pd->getNumParticles();
    const int *index = subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr;  // subgroup->getIndexIterator(access::gpu)
    auto pos = pd->getPos(access::gpu, access::readwrite);
    auto vel = pd->getVel(access::gpu, access::readwrite);
    auto force = pd->getForce(access::gpu, access::readwrite);
    auto mass = defaultMass > 0 ? property_ptr<real>() : pd->getMassIfAllocated(access::gpu, access::read);
    auto fn = kernelKind() == 1 ? uammd_verletnvt_gj : uammd_verletnvt_basic;
    detail::check(fn(step, (float *)pos.raw(), (float *)vel.raw(), (float *)force.raw(), mass.raw(), defaultMass, index, N, dt,
                     friction, is2D, noiseAmplitude, (uint)steps, seed, (void *)stream));
  }
  void resetForces() {  // the members' forces only (Basic.cu:108-115)
    auto force = pd->getForce(access::gpu, access::write);
    if (subgroup) detail::check(uammd_fill_zero_indexed(force.raw(), subgroup->getIndicesRawPtr(access::gpu), subgroup->getNumberParticles(), (int)sizeof(real4), (void *)stream));
    else detail::check(uammd_fill_zero(force.raw(), sizeof(real4) * force.size(), (void *)stream));
  }
public:
  Basic(shared_ptr<ParticleData> pd, Parameters par, std::string name = "VerletNVT::Basic")
      : Basic(make_shared<ParticleGroup>(pd, "All"), par, name) {}
  Basic(shared_ptr<ParticleGroup> group, Parameters par, std::string name = "VerletNVT::Basic")
      : Integrator(group, name), dt(par.dt), temperature(par.temperature), friction(par.friction), is2D(par.is2D) {
    sys->rng().next32();  // Basic.cu:36-38
    sys->rng().next32();
    seed = sys->rng().next32();
    noiseAmplitude = std::sqrt(2 * dt * friction * temperature);
    defaultMass = par.mass;
    if (!pd->isMassAllocated() && defaultMass < 0) defaultMass = 1.0;
    if (par.initVelocities) {
      auto vel = pd->getVel(access::gpu, access::write);
      detail::check(uammd_verletnvt_initial_velocities((float *)vel.raw(), subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr,
                                                       (real)std::sqrt(3.0 * temperature), is2D,
                                                       subgroup ? subgroup->getNumberParticles() : pd->getNumParticles(), sys->rng().next32(), nullptr));
    }
  }
  void forwardTime() override {  // Basic.cu:148-171, GronbechJensen.cu:88-115
    for (auto &u : updatables) u->updateSimulationTime(steps * dt);
    steps++;
    if (steps == 1) {
      resetForces();
      for (auto &u : updatables) { u->updateTemperature(temperature); u->updateTimeStep(dt); }
      for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, stream); }
      detail::hipCheck(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    if (kernelKind() == 1 && !subgroup && interactors.size() == 1) {  // GronbechJensen + one interactor: try the fused step
      auto pos = pd->getPos(access::gpu, access::readwrite);
      auto vel = pd->getVel(access::gpu, access::readwrite);
      auto force = pd->getForce(access::gpu, access::readwrite);
      auto mass = defaultMass > 0 ? property_ptr<real>() : pd->getMassIfAllocated(access::gpu, access::read);
      const Interactor::FusedGronbechJensen a{(float *)pos.raw(), (float *)vel.raw(), (float *)force.raw(), mass.raw(), defaultMass,
                                              pd->getNumParticles(), dt, friction, is2D, noiseAmplitude, (uint)steps, seed, stream};
      if (interactors[0]->fusedGronbechJensenStep(a)) return;
    }
    callIntegrate(1);
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, stream); }
    callIntegrate(2);
  }
  real sumEnergy() override {  // sumKineticEnergy, Basic.cu:186-207: energy[i] += m v^2 / 2, returns 0
    const int N = subgroup ? subgroup->getNumberParticles() : pd->getNumParticles();
    auto vel = pd->getVel(access::gpu, access::read);
    auto energy = pd->getEnergy(access::gpu, access::readwrite);
    auto mass = defaultMass > 0 ? property_ptr<real>() : pd->getMassIfAllocated(access::gpu, access::read);
    detail::check(uammd_sum_kinetic_energy((const float *)vel.raw(), energy.raw(), mass.raw(), defaultMass,
                                           subgroup ? subgroup->getIndicesRawPtr(access::gpu) : nullptr, N, (void *)stream));
    return 0;
  }
};
class GronbechJensen final : public Basic {
  int kernelKind() const override { return 1; }
public:
  using Parameters = Basic::Parameters;
  GronbechJensen(shared_ptr<ParticleData> pd, Parameters par) : Basic(pd, par, "VerletNVT::GronbechJensen") {}
  GronbechJensen(shared_ptr<ParticleGroup> pg, Parameters par) : Basic(pg, par, "VerletNVT::GronbechJensen") {}
};
}  // namespace VerletNVT

#endif   // !DOUBLE_PRECISION

// ---- BD: Brownian dynamics without hydrodynamic interactions (Integrator/BrownianDynamics.cuh:57-183, .cu) ------------------------------------
// (both precisions: with -DDOUBLE_PRECISION — test/BD/Makefile:2 builds the reference's BD test so — the position update is
// uammd_bd_scheme_step_f64; the draws stay float Gaussians there too, as Saru::gf makes them in the reference's double build)
namespace BD {
struct Parameters {
  std::vector<real3> K = std::vector<real3>(3, real3());   // shear matrix, row by row
  real temperature = 0, viscosity = 1.0, hydrodynamicRadius = -1.0, dt = 0.0;
  bool is2D = false;
};
// what the four schemes share (BrownianDynamics.cuh:67-110, .cu:9-117): self mobility 1 / (6 pi eta a) — a from the parameters or, when it
// is left at -1 and the particles carry radii, per particle —, the shear matrix, the seed (third draw of the System generator), forces
class BaseBrownianIntegrator : public Integrator {
public:
  using Parameters = BD::Parameters;
  BaseBrownianIntegrator(shared_ptr<ParticleGroup> pg, Parameters par)
      : Integrator(pg, "BD::BaseBrownianIntegrator"), temperature(par.temperature), dt(par.dt), is2D(par.is2D) {
    sys->rng().next32();
    sys->rng().next32();
    seed = sys->rng().next32();
    selfMobility = real(1.0 / (6.0 * M_PI * par.viscosity));
    if (par.hydrodynamicRadius != real(-1.0)) {
      selfMobility /= par.hydrodynamicRadius;
      hydrodynamicRadius = par.hydrodynamicRadius;
    } else if (!pd->isRadiusAllocated()) hydrodynamicRadius = real(1.0);
    if (par.K.size() == 3)
      for (int i = 0; i < 3; ++i) { K[3 * i] = par.K[i].x; K[3 * i + 1] = par.K[i].y; K[3 * i + 2] = par.K[i].z; }
    for (real k : K) sheared = sheared || k != real(0.0);
  }
  BaseBrownianIntegrator(shared_ptr<ParticleData> pd, Parameters par) : BaseBrownianIntegrator(make_shared<ParticleGroup>(pd, "All"), par) {}
  real sumEnergy() override {   // 3/2 kT to every member's energy (:82-92)
    auto energy = pd->getEnergy(access::cpu, access::readwrite);
    auto index = pg->getIndexIterator(access::cpu);
    for (int k = 0; k < pg->getNumberParticles(); ++k) energy[index[k]] += real(1.5) * temperature;
    return 0;
  }
protected:
  real K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool sheared = false;
  real selfMobility, hydrodynamicRadius = real(-1.0), temperature, dt;
  bool is2D;
  hipStream_t st = 0;
  int steps = 0;
  uint seed;
  void updateInteractors() {   // .cu:77-88
    for (auto &u : updatables) u->updateSimulationTime(steps * dt);
    if (steps == 1) for (auto &u : updatables) { u->updateTemperature(temperature); u->updateTimeStep(dt); }
  }
  void computeCurrentForces() {   // .cu:90-106
    resetGroupForces(st);
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, st); }
  }
  // getParticleRadiusIfAvailable (.cu:108-117)
  property_ptr<real> radiusIfUsed() { return (hydrodynamicRadius == real(-1.0)) ? pd->getRadiusIfAllocated(access::gpu, access::read) : property_ptr<real>(); }
  // one launch of the library's kernel for the scheme (0 = EulerMaruyama)
  void advance(int scheme, int substep, real4 *aux, const int *originalIndex) {
    auto radius = radiusIfUsed();
    auto pos = pd->getPos(access::gpu, access::readwrite);
    auto force = pd->getForce(access::gpu, access::read);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bd_scheme_step_f64(scheme, substep, (double *)pos.raw(), (double *)aux, groupIndex(), originalIndex, (const double *)force.raw(),
                                           sheared ? K : nullptr, selfMobility, radius.raw(), dt, is2D, temperature, groupSize(), (uint)steps, seed, (void *)st));
#else
    if (scheme == 0)
      detail::check(uammd_bd_euler_maruyama((float *)pos.raw(), groupIndex(), (const float *)force.raw(), sheared ? K : nullptr, selfMobility, radius.raw(), dt,
                                            is2D, temperature, groupSize(), (uint)steps, seed, (void *)st));
    else
      detail::check(uammd_bd_scheme_step(scheme, substep, (float *)pos.raw(), (float *)aux, groupIndex(), originalIndex, (const float *)force.raw(),
                                         sheared ? K : nullptr, selfMobility, radius.raw(), dt, is2D, temperature, groupSize(), (uint)steps, seed, (void *)st));
#endif
  }
};
class EulerMaruyama : public BaseBrownianIntegrator {   // x += dt (K x + M F) + sqrt(2 T M dt) dW (.cu:119-170)
public:
  EulerMaruyama(shared_ptr<ParticleGroup> pg, Parameters par) : BaseBrownianIntegrator(pg, par) {}
  EulerMaruyama(shared_ptr<ParticleData> pd, Parameters par) : EulerMaruyama(make_shared<ParticleGroup>(pd, "All"), par) {}
  void forwardTime() override {
    steps++;
    updateInteractors();
    computeCurrentForces();
    advance(0, 0, nullptr, nullptr);
  }
};
class MidPoint : public BaseBrownianIntegrator {   // two force evaluations per step (.cu:172-232)
  detail::DeviceArray<real4> initialPositions;
public:
  MidPoint(shared_ptr<ParticleGroup> pg, Parameters par) : BaseBrownianIntegrator(pg, par) {}
  MidPoint(shared_ptr<ParticleData> pd, Parameters par) : MidPoint(make_shared<ParticleGroup>(pd, "All"), par) {}
  void forwardTime() override {
    steps++;
    updateInteractors();
    initialPositions.resize(groupSize());
    computeCurrentForces();
    advance(UAMMD_BD_MIDPOINT, 0, initialPositions.d, nullptr);
    computeCurrentForces();
    advance(UAMMD_BD_MIDPOINT, 1, initialPositions.d, nullptr);
  }
};
class AdamsBashforth : public BaseBrownianIntegrator {   // forces 3/2 F_n - 1/2 F_(n-1) (.cu:234-308)
  detail::DeviceArray<real4> previousForces, forceRows;
  void storeCurrentForces() {   // the members' forces, in the group's order (.cu:248-257)
    const int N = groupSize();
    previousForces.resize(N);
    auto force = pd->getForce(access::gpu, access::read);
    if (subgroup) detail::check(uammd_gather(force.raw(), groupIndex(), previousForces.d, N, (int)sizeof(real4), (void *)st));
    else detail::hipCheck(hipMemcpyAsync(previousForces.d, force.raw(), sizeof(real4) * (size_t)N, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
  }
public:
  AdamsBashforth(shared_ptr<ParticleGroup> pg, Parameters par) : BaseBrownianIntegrator(pg, par) {}
  AdamsBashforth(shared_ptr<ParticleData> pd, Parameters par) : AdamsBashforth(make_shared<ParticleGroup>(pd, "All"), par) {}
  void forwardTime() override {
    steps++;
    if (steps == 1) { updateInteractors(); computeCurrentForces(); }
    storeCurrentForces();
    updateInteractors();
    computeCurrentForces();
    advance(UAMMD_BD_ADAMS_BASHFORTH, 0, previousForces.d, nullptr);
  }
};
class Leimkuhler : public BaseBrownianIntegrator {   // noise (dW_n + dW_(n-1)) / 2 (.cu:310-384)
public:
  Leimkuhler(shared_ptr<ParticleGroup> pg, Parameters par) : BaseBrownianIntegrator(pg, par) {}
  Leimkuhler(shared_ptr<ParticleData> pd, Parameters par) : Leimkuhler(make_shared<ParticleGroup>(pd, "All"), par) {}
  void forwardTime() override {
    steps++;
    updateInteractors();
    computeCurrentForces();
    advance(UAMMD_BD_LEIMKUHLER, 0, nullptr, pd->getIdOrderedIndices(access::gpu));
  }
};
}  // namespace BD

// ---- BDHI::Parameters (Integrator/BDHI/BDHI.cuh:13-24) -----------------------------------------------------------------------------------
namespace BDHI {
struct Parameters {
  std::vector<real3> K;   // the 3x3 shear matrix as three rows
  real temperature = 0, viscosity = 1, hydrodynamicRadius = -1, tolerance = 1e-3, dt = 0;
  bool is2D = false;
  Box box;
};
template <class T> using cached_vector = uninitialized_cached_vector<T>;   // BDHI/FCM/utils.cuh:15
}  // namespace BDHI

namespace detail {
// the window descriptor the library takes, in the working precision (uammd_hip.h)
#if defined(DOUBLE_PRECISION)
using IBMDescriptor = uammd_ibm_kernel_f64;
#else
using IBMDescriptor = uammd_ibm_kernel;
#endif
inline IBMDescriptor gridWindow(int kind, int support, real h) {   // a window defined per grid cell (Peskin, six-point): no free parameters
  IBMDescriptor k{};
  k.kind = kind;
  k.support[0] = k.support[1] = k.support[2] = support;
  k.rmax = std::numeric_limits<real>::infinity();
  k.invh[0] = k.invh[1] = k.invh[2] = real(1.0) / h;
  return k;
}
UAMMD_HOSTDEV inline float expR(float x) { return ::expf(x); }
UAMMD_HOSTDEV inline double expR(double x) { return ::exp(x); }
UAMMD_HOSTDEV inline float sqrtR(float x) { return ::sqrtf(x); }
UAMMD_HOSTDEV inline double sqrtR(double x) { return ::sqrt(x); }
UAMMD_HOSTDEV inline float fmaR(float a, float b, float c) { return ::fmaf(a, b, c); }
UAMMD_HOSTDEV inline double fmaR(double a, double b, double c) { return ::fma(a, b, c); }
UAMMD_HOSTDEV inline float ceilR(float x) { return ::ceilf(x); }
UAMMD_HOSTDEV inline double ceilR(double x) { return ::ceil(x); }
// the two Gaussian windows of FCM choose their support the same way (FCM_kernels.cuh:36-44, :67-75): walk out in steps of h / 2 until the
// window is below the tolerance
template <class Phi> inline int supportForTolerance(Phi phi, real h, real tolerance) {
  const real dr = real(0.5) * h;
  real r = dr;
  while (phi(r) > tolerance) r += dr;
  return std::max(3, int(2 * r / h + 0.5));
}
}  // namespace detail

// ---- misc/IBM_kernels.cuh:26-240: the windows UAMMD ships.  Each is a small value type with phi(r) callable from host AND device code (a
// kernel of IBM<> in the user's TU evaluates it there), and describe(): the same window as the library's descriptor, which the host-only
// path of IBM<> and FCM_impl hand to the C ABI. ----
namespace IBM_kernels {
class Gaussian {  // phi(r) = exp(-r^2 / (2 width^2)) / sqrt(2 pi width^2), :28-40
  real prefactor, tau;
public:
  int support;    // nodes per axis when used with IBM<> directly (the reference asks a wrapping kernel for it)
  explicit Gaussian(real width, int support_ = 0)
      : prefactor(real(std::pow(2.0 * M_PI * double(width) * double(width), -0.5))), tau(real(-0.5 / (double(width) * double(width)))), support(support_) {}
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return prefactor * detail::expR(tau * r * r); }
  detail::IBMDescriptor describe() const {
    if (support <= 0) throw std::invalid_argument("IBM_kernels::Gaussian: give the support (nodes per axis) to spread or gather with it");
    detail::IBMDescriptor k{};
    k.kind = UAMMD_IBM_KERNEL_GAUSSIAN;
    k.support[0] = k.support[1] = k.support[2] = support;
    k.prefactor = prefactor; k.tau = tau; k.rmax = std::numeric_limits<real>::infinity();
    return k;
  }
};
// "exponential of a semicircle" (:82-112): phi(r) = exp(beta (sqrt(1 - (r / alpha)^2) - 1)) / norm inside |r| < alpha, alpha = half width
class BarnettMagland {
  real invnorm;
public:
  real alpha, beta;
  int support;
  BarnettMagland(real alpha_, real beta_, int support_ = 0) : alpha(alpha_), beta(beta_), support(support_) {
#if defined(DOUBLE_PRECISION)
    // norm = 2 int_0^alpha: composite Simpson rule on 20000 intervals, compensated sum (:44-79, :93-97)
    const int Nr = 20000;
    const double dx = double(alpha) / Nr;
    double sum = 0, c = 0;
    for (int i = 0; i <= Nr; ++i) {
      const double w = (i == 0 || i == Nr) ? 1.0 : ((i % 2) ? 4.0 : 2.0);
      const double y = w * shape(real(i * dx)) - c, t = sum + y;
      c = (t - sum) - y;
      sum = t;
    }
    invnorm = real(1.0 / (2.0 * dx / 3.0 * sum));
#else
    uammd_ibm_kernel k;   // (the library's evaluation of the same rule in single precision: one value for the host and the device side)
    uammd::detail::check(uammd_ibm_barnett_magland_kernel(alpha, beta, std::max(support, 1), real(1.0), &k));
    invnorm = k.prefactor;
#endif
  }
  UAMMD_HOSTDEV real shape(real r) const {
    const real z = r / alpha, dz2 = real(1.0) - z * z;
    return dz2 < real(0.0) ? real(0.0) : detail::expR(beta * (detail::sqrtR(dz2) - real(1.0)));
  }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return shape(r) * invnorm; }
  detail::IBMDescriptor describe(real lengthUnit = real(1.0)) const {
    if (support <= 0) throw std::invalid_argument("IBM_kernels::BarnettMagland: give the support (nodes per axis) to spread or gather with it");
    detail::IBMDescriptor k{};
    k.kind = UAMMD_IBM_KERNEL_BARNETT_MAGLAND;
    k.support[0] = k.support[1] = k.support[2] = support;
    k.prefactor = invnorm; k.tau = beta; k.rmax = alpha;
    k.invh[0] = k.invh[1] = k.invh[2] = lengthUnit;   // (this field carries the wrapper's length unit for this window)
    return k;
  }
};
namespace Peskin {
struct threePoint {  // :118-137
  real invh;
  static constexpr int support = 3;
  explicit threePoint(real h) : invh(real(1.0) / h) {}
  UAMMD_HOSTDEV real phi(real rr, real3 = real3()) const {
    const real r = (rr < 0 ? -rr : rr) * invh;
    if (r < real(0.5)) return invh * real(1 / 3.0) * (real(1.0) + detail::sqrtR(detail::fmaR(real(-3.0) * r, r, real(1.0))));
    if (r < real(1.5)) {
      const real omr = real(1.0) - r;
      return invh * real(1 / 6.0) * (detail::fmaR(real(-3.0), r, real(5.0)) - detail::sqrtR(detail::fmaR(real(-3.0) * omr, omr, real(1.0))));
    }
    return 0;
  }
  detail::IBMDescriptor describe() const { return detail::gridWindow(UAMMD_IBM_KERNEL_PESKIN3, 3, real(1.0) / invh); }
};
struct fourPoint {  // :140-160
  real invh;
  static constexpr int support = 4;
  explicit fourPoint(real h) : invh(real(1.0) / h) {}
  UAMMD_HOSTDEV real phi(real rr, real3 = real3()) const {
    const real r = (rr < 0 ? -rr : rr) * invh;
    if (r < real(1.0)) return invh * real(0.125) * (detail::fmaR(real(-2.0), r, real(3.0)) + detail::sqrtR(detail::fmaR(real(4.0) * r, real(1.0) - r, real(1.0))));
    if (r < real(2.0))
      return invh * real(0.125) * (detail::fmaR(real(-2.0), r, real(5.0)) - detail::sqrtR(detail::fmaR(-(real(4.0) * r), r, detail::fmaR(real(12.0), r, real(-7.0)))));
    return 0;
  }
  detail::IBMDescriptor describe() const { return detail::gridWindow(UAMMD_IBM_KERNEL_PESKIN4, 4, real(1.0) / invh); }
};
}  // namespace Peskin
namespace GaussianFlexible {
struct sixPoint {  // the C3 six-point window of Bao, Kaye and Peskin with K = 59/60 - sqrt(29)/20 (:162-236)
  real invh;
  static constexpr int support = 6;
  explicit sixPoint(real h, real = 1e-7) : invh(real(1.0) / h) {}
  UAMMD_HOSTDEV real phi(real rr, real3 = real3()) const {
    const real r = (rr < 0 ? -rr : rr) * invh;
    if (r >= real(3.0)) return 0;
    const real K = real(0.714075092976608);
    const real R = r - detail::ceilR(r) + real(1.0), R2 = R * R, R3 = R2 * R;
    const real b = real(9.0 / 4.0) - real(1.5) * (K + R2) + (real(22. / 3) - real(7.0) * K) * R - real(7. / 3.) * R3;
    const real g = real(0.25) * (real(0.5) * (real(161.0) / real(36.0) - real(59.0) / real(6.0) * K + real(5.0) * K * K) * R2 +
                                 real(1.0) / real(3.0) * (real(-109.0) / real(24.0) + real(5.0) * K) * R2 * R2 + real(5.0) / real(18.0) * R3 * R3);
    const real pre = real(1.0) / (real(2.0) * real(28.0)) * (-b + detail::sqrtR(b * b - real(4.0) * real(28.0) * g));   // the branch-independent root
    real v;
    if (r <= real(0.0)) { const real t = r + real(1.0); v = real(2.0) * pre + real(0.25) + real(1. / 6) * (real(4.0) - real(3.0) * K) * t - real(1. / 6) * t * t * t; }
    else if (r <= real(1.0)) v = real(2.0) * pre + real(5. / 8) - real(0.25) * (K + r * r);
    else if (r <= real(2.0)) { const real t = r + real(-1.0); v = real(-3.0) * pre + real(0.25) - real(1. / 6.) * (real(4.0) - real(3.0) * K) * t + real(1. / 6) * t * t * t; }
    else { const real t = r + real(-2.0); v = pre - real(1. / 16) + real(1. / 8) * (K + t * t) - real(1. / 12) * (real(3.0) * K - real(1.0)) * t - real(1. / 12) * t * t * t; }
    return v * invh;
  }
  detail::IBMDescriptor describe() const { return detail::gridWindow(UAMMD_IBM_KERNEL_SIXPOINT, 6, real(1.0) / invh); }
};
}  // namespace GaussianFlexible
}  // namespace IBM_kernels

// ---- FCM_ns::Kernels (BDHI/FCM/FCM_kernels.cuh): the windows FCM_impl is instantiated with — an IBM window plus what FCM needs to know
// about it: Kernel(h, tolerance), support, fixHydrodynamicRadius(.., h), adviseGridSize(a, tolerance). ----
namespace BDHI {
namespace FCM_ns {
namespace Kernels {
class Gaussian {  // :22-58
  uammd::detail::IBMDescriptor desc{};
  real a = 0;
public:
  int support;
  real rmax;
  Gaussian(real h, real tolerance) {
#if defined(DOUBLE_PRECISION)
    uammd::detail::check(uammd_fcm_gaussian_kernel_f64(h, tolerance, &desc, &a));
#else
    uammd::detail::check(uammd_fcm_gaussian_kernel(h, tolerance, &desc, &a));
#endif
    support = desc.support[0];
    rmax = desc.rmax;
  }
  static real adviseGridSize(real hydrodynamicRadius, real tolerance) {
#if defined(DOUBLE_PRECISION)
    return uammd_fcm_advise_grid_size_f64(hydrodynamicRadius, tolerance);
#else
    return uammd_fcm_advise_grid_size(hydrodynamicRadius, tolerance);
#endif
  }
  real fixHydrodynamicRadius(real, real) const { return a; }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return r >= rmax ? real(0) : desc.prefactor * uammd::detail::expR(desc.tau * r * r); }
  const uammd::detail::IBMDescriptor &describe() const { return desc; }
};
class GaussianTorque {  // :60-80
  uammd::detail::IBMDescriptor desc{};
public:
  int support;
  real rmax;
  GaussianTorque(real width, real h, real tolerance) {
    desc.kind = UAMMD_IBM_KERNEL_GAUSSIAN;
    desc.prefactor = real(std::pow(2.0 * M_PI * double(width) * double(width), -0.5));
    desc.tau = real(-0.5 / (double(width) * double(width)));
    const real pre = desc.prefactor, tau = desc.tau;
    support = uammd::detail::supportForTolerance([pre, tau](real r) { return pre * uammd::detail::expR(tau * r * r); }, h, tolerance);
    desc.support[0] = desc.support[1] = desc.support[2] = support;
    desc.rmax = rmax = real(support) * h;
  }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return r >= rmax ? real(0) : desc.prefactor * uammd::detail::expR(desc.tau * r * r); }
  const uammd::detail::IBMDescriptor &describe() const { return desc; }
};
class BarnettMagland {  // :82-155
  IBM_kernels::BarnettMagland bm;
  real a;
  static int computeSupport(real tol) {
    real w = std::max(1.5, int(-std::log10(tol) + 2) / 2.0);
    w = std::min(real(9.0), w);
    return (int)std::ceil(w);
  }
  static real computeUpsampling(real w) { return 1.36409985665115 * std::pow(w, -0.53028415751646); }
  static IBM_kernels::BarnettMagland make(real tolerance) {
    const int w = computeSupport(tolerance);
    return IBM_kernels::BarnettMagland(real(w * 0.5), real(1.8 * w * 2), (int)std::ceil(2 * (w * 0.5)));
  }
public:
  int support;
  BarnettMagland(real h, real tolerance) : bm(make(tolerance)), a(h), support(bm.support) {}
  static real adviseGridSize(real hydrodynamicRadius, real tolerance) { return hydrodynamicRadius * computeUpsampling(computeSupport(tolerance)); }
  real fixHydrodynamicRadius(real, real h) const { return h / computeUpsampling(support); }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return bm.phi(r / a) / a; }
  uammd::detail::IBMDescriptor describe() const { return bm.describe(a); }
};
namespace Peskin {
class threePoint {  // :159-176
  IBM_kernels::Peskin::threePoint kern;
public:
  static constexpr int support = 3;
  threePoint(real h, real) : kern(h) {}
  static real adviseGridSize(real hydrodynamicRadius, real) { return hydrodynamicRadius; }
  real fixHydrodynamicRadius(real, real h) const { return h; }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return kern.phi(r); }
  uammd::detail::IBMDescriptor describe() const { return kern.describe(); }
};
class fourPoint {  // :178-196
  IBM_kernels::Peskin::fourPoint kern;
public:
  static constexpr int support = 4;
  static constexpr real fac = 1.31;
  fourPoint(real h, real) : kern(h) {}
  static real adviseGridSize(real hydrodynamicRadius, real) { return hydrodynamicRadius / fac; }
  real fixHydrodynamicRadius(real, real h) const { return h * fac; }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return kern.phi(r); }
  uammd::detail::IBMDescriptor describe() const { return kern.describe(); }
};
}  // namespace Peskin
namespace GaussianFlexible {
class sixPoint {  // :199-219
  IBM_kernels::GaussianFlexible::sixPoint kern;
public:
  static constexpr int support = 6;
  static constexpr real fac = 1.5195;
  sixPoint(real h, real) : kern(h) {}
  static real adviseGridSize(real hydrodynamicRadius, real) { return hydrodynamicRadius / fac; }
  real fixHydrodynamicRadius(real, real h) const { return h * fac; }
  UAMMD_HOSTDEV real phi(real r, real3 = real3()) const { return kern.phi(r); }
  uammd::detail::IBMDescriptor describe() const { return kern.describe(); }
};
}  // namespace GaussianFlexible
}  // namespace Kernels
}  // namespace FCM_ns
}  // namespace BDHI

// ---- misc/IBM.cuh:63-203: IBM<Kernel, Grid, Index3D> — spread particle quantities to a grid, gather grid quantities to the particles
// (library mode: no ParticleData; iterators in, iterators out).  Two ways down:
//   * the LIBRARY path — a window UAMMD ships (it has describe()), real4 positions, real or real3 quantities through plain device pointers,
//     the default weights and node indexing: one call of uammd_ibm_spread / uammd_ibm_gather (their _f64 forms under DOUBLE_PRECISION).
//     Any C++ compiler.
//   * the TEMPLATE path — anything else (a user's kernel, int or vector quantities, thrust iterators, real3 positions, WeightCompute /
//     QuadratureWeights / Index3D functors): include/uammd/device/IBM.hip.hpp, compiled with the user's TU by hipcc.
// A grid with cellDim.z == 1 is treated as 2D (IBM.cuh:182-194). ----
namespace IBM_ns {
struct LinearIndex3D {
  UAMMD_HOSTDEV LinearIndex3D(int nx_, int ny_, int nz_) : nx(nx_), ny(ny_), nz(nz_) {}
  UAMMD_HOSTDEV int operator()(int3 c) const { return (*this)(c.x, c.y, c.z); }
  UAMMD_HOSTDEV int operator()(int i, int j, int k) const { return i + nx * (j + ny * k); }
  int nx, ny, nz;
};
namespace detail {
template <class K, class = void> struct has_describe : std::false_type {};
template <class K> struct has_describe<K, decltype(void(std::declval<const K &>().describe()))> : std::true_type {};
template <class P> struct pointee { using type = void; };
template <class T> struct pointee<T *> { using type = typename std::remove_cv<T>::type; };
// can this call go to the C ABI as it is?
template <class Kernel, class GridT, class Index3D, class Pos, class Q, class G> struct LibraryPath {
  using P = typename pointee<Pos>::type;
  using QT = typename pointee<Q>::type;
  using GT = typename pointee<G>::type;
  static constexpr bool value = has_describe<Kernel>::value && std::is_same<GridT, uammd::Grid>::value && std::is_same<Index3D, LinearIndex3D>::value &&
                                std::is_same<P, real4>::value && std::is_same<QT, GT>::value && (std::is_same<QT, real>::value || std::is_same<QT, real3>::value);
};
}  // namespace detail
}  // namespace IBM_ns
}  // namespace uammd
#if defined(__HIPCC__)
#include "device/IBM.hip.hpp"
#endif
namespace uammd {
#if !defined(__HIPCC__)
namespace IBM_ns {   // (named so that signatures with defaulted functor arguments read the same in a host-only TU; they hold device code under hipcc)
struct DefaultQuadratureWeights {};
struct DefaultWeightCompute {};
}  // namespace IBM_ns
#endif
template <class Kernel, class GridT = uammd::Grid, class Index3D = IBM_ns::LinearIndex3D> class IBM {
  shared_ptr<Kernel> kernel;
  GridT grid;
  Index3D cell2index;
  template <class Q> static int components() { return (int)(sizeof(Q) / sizeof(real)); }
  // ---- library path ----
  template <class Q> void spreadLibrary(const real4 *pos, const Q *v, Q *gridData, int N, hipStream_t st) const {
    int per[3];
    const int cd[3] = {grid.cellDim.x, grid.cellDim.y, grid.cellDim.z};
    const auto k = kernel->describe();
#if defined(DOUBLE_PRECISION)
    double L[3];
    grid.box.toArrays(L, per);
    uammd::detail::check(uammd_ibm_spread_f64((const double *)pos, 4, (const double *)v, components<Q>(), N, L, per, cd, cell2index.nx, &k, (double *)gridData, (void *)st));
#else
    float L[3];
    grid.box.toArrays(L, per);
    uammd::detail::check(uammd_ibm_spread((const float *)pos, 4, (const float *)v, components<Q>(), N, L, per, cd, cell2index.nx, &k, (float *)gridData, (void *)st));
#endif
  }
  template <class Q> void gatherLibrary(const real4 *pos, Q *Jq, const Q *gridData, int N, hipStream_t st) const {
    int per[3];
    const int cd[3] = {grid.cellDim.x, grid.cellDim.y, grid.cellDim.z};
    const auto k = kernel->describe();
#if defined(DOUBLE_PRECISION)
    double L[3];
    grid.box.toArrays(L, per);
    uammd::detail::check(uammd_ibm_gather_f64((const double *)pos, 4, (double *)Jq, components<Q>(), N, L, per, cd, cell2index.nx, &k, (const double *)gridData, (void *)st));
#else
    float L[3];
    grid.box.toArrays(L, per);
    uammd::detail::check(uammd_ibm_gather((const float *)pos, 4, (float *)Jq, components<Q>(), N, L, per, cd, cell2index.nx, &k, (const float *)gridData, (void *)st));
#endif
  }
  template <class Pos, class Q, class G> void spreadDefault(std::true_type, Pos pos, Q v, G gridData, int N, hipStream_t st) const {
    spreadLibrary(pos, v, gridData, N, st);
  }
  template <class Pos, class R, class G> void gatherDefault(std::true_type, Pos pos, R Jq, G gridData, int N, hipStream_t st) const {
    gatherLibrary(pos, Jq, gridData, N, st);
  }
#if defined(__HIPCC__)
  // ---- template path ----
  template <class Pos, class Q, class G> void spreadDefault(std::false_type, Pos pos, Q v, G gridData, int N, hipStream_t st) const {
    spread(pos, v, gridData, IBM_ns::DefaultWeightCompute(), N, st);
  }
  template <class Pos, class R, class G> void gatherDefault(std::false_type, Pos pos, R Jq, G gridData, int N, hipStream_t st) const {
    gather(pos, Jq, gridData, IBM_ns::DefaultQuadratureWeights(), IBM_ns::DefaultWeightCompute(), N, st);
  }
#else
  template <class Pos, class Q, class G> void spreadDefault(std::false_type, Pos, Q, G, int, hipStream_t) const {
    static_assert(sizeof(Pos) == 0, "IBM<>: a user-defined kernel or non-pointer / non-real quantities need the device template: compile this TU with hipcc");
  }
  template <class Pos, class R, class G> void gatherDefault(std::false_type, Pos, R, G, int, hipStream_t) const {
    static_assert(sizeof(Pos) == 0, "IBM<>: a user-defined kernel or non-pointer / non-real quantities need the device template: compile this TU with hipcc");
  }
#endif
  // (in a DOUBLE_PRECISION TU compiled by hipcc every call takes the template path: the library's double build holds the Gaussian and Peskin
  // windows only, the template holds whatever phi says)
  template <class Pos, class Q, class G> struct UseLibrary
      : std::integral_constant<bool, IBM_ns::detail::LibraryPath<Kernel, GridT, Index3D, Pos, Q, G>::value
#if defined(DOUBLE_PRECISION) && defined(__HIPCC__)
                                         && false
#endif
                               > {};
public:
  IBM(shared_ptr<Kernel> kern, GridT a_grid, Index3D index) : kernel(kern), grid(a_grid), cell2index(index) {}
  IBM(shared_ptr<Kernel> kern, GridT a_grid) : IBM(kern, a_grid, Index3D(a_grid.cellDim.x, a_grid.cellDim.y, a_grid.cellDim.z)) {}
  // gridData[node] += sum_i weightCompute(v[i], phi(node - pos[i]))   (IBM.cuh:118-138)
  template <class Pos, class Q, class G> void spread(Pos pos, Q v, G gridData, int numberParticles, hipStream_t st = 0) const {
    spreadDefault(UseLibrary<Pos, Q, G>(), pos, v, gridData, numberParticles, st);
  }
  // Jq[i] += sum_node qw(node) weightCompute(gridData[node], phi(node - pos[i]))   (IBM.cuh:140-180)
  template <class Pos, class R, class G> void gather(Pos pos, R Jq, G gridData, int numberParticles, hipStream_t st = 0) const {
    gatherDefault(UseLibrary<Pos, R, G>(), pos, Jq, gridData, numberParticles, st);
  }
#if defined(__HIPCC__)
  template <class Pos, class Q, class G, class WeightCompute, class = typename std::enable_if<!std::is_integral<WeightCompute>::value>::type>
  void spread(Pos pos, Q v, G gridData, WeightCompute weightCompute, int numberParticles, hipStream_t st = 0) const {
    if (grid.cellDim.z == 1) spread<true>(pos, v, gridData, weightCompute, numberParticles, st);
    else spread<false>(pos, v, gridData, weightCompute, numberParticles, st);
  }
  template <bool is2D, class Pos, class Q, class G, class WeightCompute, class = typename std::enable_if<!std::is_integral<WeightCompute>::value>::type>
  void spread(Pos pos, Q v, G gridData, WeightCompute weightCompute, int numberParticles, hipStream_t st = 0) const {
    IBM_ns::detail::launchSpread<is2D>(*kernel, grid, cell2index, pos, v, gridData, weightCompute, numberParticles, st);
    uammd::detail::hipCheck(hipGetLastError(), "IBM::spread");
  }
  template <bool is2D, class Pos, class Q, class G> void spread(Pos pos, Q v, G gridData, int numberParticles, hipStream_t st = 0) const {
    spread<is2D>(pos, v, gridData, IBM_ns::DefaultWeightCompute(), numberParticles, st);
  }
  template <class Pos, class R, class G, class QuadratureWeights, class WeightCompute>
  void gather(Pos pos, R Jq, G gridData, QuadratureWeights qw, WeightCompute wc, int numberParticles, hipStream_t st = 0) const {
    if (grid.cellDim.z == 1) gather<true>(pos, Jq, gridData, qw, wc, numberParticles, st);
    else gather<false>(pos, Jq, gridData, qw, wc, numberParticles, st);
  }
  template <bool is2D, class Pos, class R, class G, class QuadratureWeights, class WeightCompute>
  void gather(Pos pos, R Jq, G gridData, QuadratureWeights qw, WeightCompute wc, int numberParticles, hipStream_t st = 0) const {
    IBM_ns::detail::launchGather<is2D>(*kernel, grid, cell2index, pos, Jq, gridData, qw, wc, numberParticles, st);
    uammd::detail::hipCheck(hipGetLastError(), "IBM::gather");
  }
  template <bool is2D, class Pos, class R, class G> void gather(Pos pos, R Jq, G gridData, int numberParticles, hipStream_t st = 0) const {
    gather<is2D>(pos, Jq, gridData, IBM_ns::DefaultQuadratureWeights(), IBM_ns::DefaultWeightCompute(), numberParticles, st);
  }
#endif
  shared_ptr<Kernel> getKernel() { return kernel; }
};

namespace BDHI {
// FCM_impl (BDHI/FCM/FCM_impl.cuh:36-130): the solver without ParticleData — positions, forces (and torques) in, velocities out.
// Kernel / KernelTorque are the windows of FCM_ns::Kernels: the library evaluates them from their describe()d parameters (a user-written
// FCM window is not a library window: spreading it needs IBM<> above with the user's TU).
template <class Kernel = FCM_ns::Kernels::Gaussian, class KernelTorque = FCM_ns::Kernels::GaussianTorque> class FCM_impl {
  static_assert(IBM_ns::detail::has_describe<Kernel>::value && IBM_ns::detail::has_describe<KernelTorque>::value,
                "FCM_impl takes the windows of BDHI::FCM_ns::Kernels (the library evaluates them); spread a window of your own with IBM<>");
#if defined(DOUBLE_PRECISION)
  uammd_fcm_f64 *h = nullptr;
#else
  uammd_fcm *h = nullptr;
#endif
  Box box;
  real viscosity, hydrodynamicRadius;
  uint seed = 0, seed2 = 0;
  shared_ptr<Kernel> kernel;
  shared_ptr<KernelTorque> kernelTorque;
public:
  struct Parameters : BDHI::Parameters {   // FCM_impl.cuh:47-54
    int3 cells = make_int3(-1, -1, -1);    // number of Fourier nodes in each direction
    uint seed = 0;
    std::shared_ptr<Kernel> kernel = nullptr;
    std::shared_ptr<KernelTorque> kernelTorque = nullptr;
    bool adaptBoxSize = false;
  };
  explicit FCM_impl(Parameters par)
      : box(par.box), viscosity(par.viscosity), hydrodynamicRadius(par.hydrodynamicRadius), seed(par.seed), kernel(par.kernel), kernelTorque(par.kernelTorque) {
    // FCM_impl.cuh:56-92, in its order
    if (box.boxSize.x == real(0.0) && box.boxSize.y == real(0.0) && box.boxSize.z == real(0.0))
      System::log<System::CRITICAL>("[BDHI::FCM] Box of size zero detected, cannot work without a box! (make sure a box parameter was passed)");
    if (seed == 0) seed = (uint)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (par.box.boxSize.x <= 0 || par.cells.x <= 0) throw std::runtime_error("Invalid arguments");
    if (!par.kernel || !par.kernelTorque) {
      System::log<System::EXCEPTION>("FCM_impl requires instances of the spreading kernels");
      throw std::runtime_error("Invalid arguments");
    }
#if defined(DOUBLE_PRECISION)
    uammd_fcm_parameters_f64 p{};
#else
    uammd_fcm_parameters p{};
    p.seed = seed;
    p.hydrodynamicRadius = hydrodynamicRadius;
#endif
    p.boxSize[0] = par.box.boxSize.x; p.boxSize[1] = par.box.boxSize.y; p.boxSize[2] = par.box.boxSize.z;
    p.cells[0] = par.cells.x; p.cells[1] = par.cells.y; p.cells[2] = par.cells.z;
    p.viscosity = par.viscosity;
    p.kernel = kernel->describe();
#if defined(DOUBLE_PRECISION)
    uammd::detail::check(uammd_fcm_create_f64(&p, &h));   // (linear velocities; torques are not in the library's double-precision build)
#else
    uammd::detail::check(uammd_fcm_create(&p, &h));
    const uammd_ibm_kernel kt = kernelTorque->describe();
    uammd::detail::check(uammd_fcm_set_torque_kernel(h, &kt));
#endif
  }
  FCM_impl(const FCM_impl &) = delete;
#if defined(DOUBLE_PRECISION)
  ~FCM_impl() { uammd_fcm_destroy_f64(h); }
#else
  ~FCM_impl() { uammd_fcm_destroy(h); }
#endif
  real getHydrodynamicRadius() { return hydrodynamicRadius; }
  real getSelfMobility() { return (real)uammd_fcm_self_mobility(hydrodynamicRadius, viscosity, box.boxSize.x); }
  Box getBox() { return box; }
  // linear velocities into d_linearVelocity (real3[N])
  void computeHydrodynamicDisplacements(const real4 *pos, const real4 *force, real3 *d_linearVelocity, int N, real temperature,
                                        real prefactor, hipStream_t st) {
#if defined(DOUBLE_PRECISION)
    if (temperature > 0) ++seed2;   // addBrownianNoise's counter, FCM_impl.cuh:517-523
    uammd::detail::check(uammd_fcm_displacements_thermal_f64(h, (const double *)pos, (const double *)force, N, temperature, prefactor, seed, seed2,
                                                             (double *)d_linearVelocity, (void *)st));
#else
    uammd::detail::check(uammd_fcm_displacements(h, (const float *)pos, (const float *)force, N, temperature, prefactor,
                                                 (float *)d_linearVelocity, (void *)st));
#endif
  }
#if !defined(DOUBLE_PRECISION)
  // computeHydrodynamicDisplacements followed by integrateEulerMaruyamaD (BDHI_FCM.cu:67-92) in one library call: pos += v dt in place
  // positionsKept: pos is exactly what the previous call left (UAMMD_FCM_STEP_POSITIONS_KEPT: that call's binning is used)
  void stepEulerMaruyama(real4 *pos, const real4 *force, real3 *d_linearVelocity, int N, real temperature, real prefactor, real dt,
                         bool positionsKept, hipStream_t st) {
    uammd::detail::check(uammd_fcm_step_euler_maruyama(h, (float *)pos, (const float *)force, N, temperature, prefactor, dt,
                                                       (float *)d_linearVelocity, positionsKept ? UAMMD_FCM_STEP_POSITIONS_KEPT : 0, (void *)st));
  }
#endif
  // The reference's own signature (FCM_impl.cuh:126-129): owning containers returned by value, the second one empty without torques
  std::pair<cached_vector<real3>, cached_vector<real3>> computeHydrodynamicDisplacements(real4 *pos, real4 *force, real4 *torque,
                                                                                         int numberParticles, real temperature,
                                                                                         real prefactor, hipStream_t st) {
    cached_vector<real3> linear((size_t)numberParticles), angular(torque ? (size_t)numberParticles : 0);
    computeHydrodynamicDisplacements(pos, force, torque, linear.raw(), angular.raw(), numberParticles, temperature, prefactor, st);
    return std::make_pair(std::move(linear), std::move(angular));
  }
  // with torques (FCM_impl.cuh:306-358): linear and angular velocities; torque == nullptr falls back to the call above
  void computeHydrodynamicDisplacements(const real4 *pos, const real4 *force, const real4 *torque, real3 *d_linearVelocity,
                                        real3 *d_angularVelocity, int N, real temperature, real prefactor, hipStream_t st) {
    if (!torque) return computeHydrodynamicDisplacements(pos, force, d_linearVelocity, N, temperature, prefactor, st);
#if defined(DOUBLE_PRECISION)
    (void)d_angularVelocity;
    throw std::runtime_error("[BDHI::FCM] torques are not part of the double-precision build of the library (single precision has them)");
#else
    uammd::detail::check(uammd_fcm_displacements_torque(h, (const float *)pos, (const float *)force, (const float *)torque, N, temperature,
                                                        prefactor, (float *)d_linearVelocity, (float *)d_angularVelocity, (void *)st));
#endif
  }
};
namespace detail_fcm {
// initializeGrid, initializeKernel, initializeKernelTorque and fixHydrodynamicRadius as BDHI::FCM's and FCMIntegrator's constructors string
// them together (BDHI_FCM.cuh:29-80, :98-110)
template <class Kernel, class KernelTorque>
inline typename FCM_impl<Kernel, KernelTorque>::Parameters initialize(typename FCM_impl<Kernel, KernelTorque>::Parameters par, System &sys) {
  if (par.seed == 0) par.seed = sys.rng().next32();
  real h = 0;
  if (par.cells.x <= 0) {
    if (par.hydrodynamicRadius <= 0) System::log<System::CRITICAL>("[BDHI::FCM] I need an hydrodynamic radius if cell dimensions are not provided!");
    h = Kernel::adviseGridSize(par.hydrodynamicRadius, par.tolerance);
    par.cells = nextFFTWiseSize3D(make_int3(par.box.boxSize / h));
    if (par.adaptBoxSize) par.box = Box(make_real3(par.cells) * h);
  }
  const Grid grid(par.box, par.cells);
  const real hmin = std::min(grid.cellSize.x, std::min(grid.cellSize.y, grid.cellSize.z));
  if (!par.kernel) par.kernel = std::make_shared<Kernel>(hmin, par.tolerance);
  if (par.kernel->support >= grid.cellDim.x || par.kernel->support >= grid.cellDim.y || par.kernel->support >= grid.cellDim.z)   // said, not fatal (:58-64)
    System::log<System::ERROR>("[BDHI::FCM] Kernel support is too big, try lowering the tolerance or increasing the box size!.");
  par.hydrodynamicRadius = par.kernel->fixHydrodynamicRadius(par.hydrodynamicRadius, grid.cellSize.x);
  if (!par.kernelTorque) {
    const real width = par.hydrodynamicRadius / real(std::pow(6 * std::sqrt(M_PI), 1 / 3.));
    par.kernelTorque = std::make_shared<KernelTorque>(width, hmin, par.tolerance);
  }
  return par;
}
}  // namespace detail_fcm
class FCM {  // the Method concept of BDHI::EulerMaruyama (BDHI_FCM.cuh:84-147)
  using Kernel = FCM_ns::Kernels::Gaussian;
  using KernelTorque = FCM_ns::Kernels::GaussianTorque;
  using FCM_super = FCM_impl<Kernel, KernelTorque>;
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;  // nullptr = all the particles
  shared_ptr<FCM_super> fcm;
  real temperature, dt;
  detail::DeviceArray<real4> posRows, forceRows;
public:
  using Parameters = FCM_super::Parameters;
  FCM(shared_ptr<ParticleGroup> group, Parameters par)  // BDHI_FCM.cuh:98-110
      : pd(group->getParticleData()), pg(detail::subsetOrNull(group)), temperature(par.temperature), dt(par.dt) {
    fcm = make_shared<FCM_super>(detail_fcm::initialize<Kernel, KernelTorque>(par, *pd->getSystem()));
  }
  FCM(shared_ptr<ParticleData> pd, Parameters par) : FCM(make_shared<ParticleGroup>(pd, "All"), par) {}
  void setup_step(hipStream_t = 0) {}
  void computeMF(real3 *MF, hipStream_t st = 0) {
    auto force = pd->getForce(access::gpu, access::read);
    auto pos = pd->getPos(access::gpu, access::read);
    const int N = pg ? pg->getNumberParticles() : pd->getNumParticles();
    fcm->computeHydrodynamicDisplacements(detail::groupRows(pos.raw(), pg.get(), posRows, st), detail::groupRows(force.raw(), pg.get(), forceRows, st),
                                          MF, N, temperature, 1.0 / std::sqrt(dt), st);
  }
  void computeBdW(real3 *, hipStream_t = 0) {}
  void finish_step(hipStream_t = 0) {}
  real getHydrodynamicRadius() { return fcm->getHydrodynamicRadius(); }
  real getSelfMobility() { return fcm->getSelfMobility(); }
};
#if !defined(DOUBLE_PRECISION)
// BDHI_FCM.cuh:155-199, BDHI_FCM.cu:7-119.  The reference fixes Kernel = Gaussian; the template parameter exposes the
// alternatives it keeps commented out in FCM_impl.cuh:39-42.
template <class Kernel = FCM_ns::Kernels::Gaussian> class FCMIntegratorT : public Integrator {
  using KernelTorque = FCM_ns::Kernels::GaussianTorque;
  shared_ptr<FCM_impl<Kernel, KernelTorque>> fcm;
  detail::DeviceArray<real3> linearV, angularV;
  real temperature, dt;
  uint steps = 0;
  hipStream_t st = 0;
  bool posTouched = true;
  scoped_connection posWriteConnection, reorderConnection;  // dropped with the integrator
  detail::DeviceArray<real4> posRows, forceRows, torqueRows;  // the rows of a proper subgroup, gathered
public:
  using Parameters = typename FCM_impl<Kernel, KernelTorque>::Parameters;
  FCMIntegratorT(shared_ptr<ParticleGroup> group, Parameters par)  // BDHI_FCM.cuh:167-174
      : Integrator(group, "BDHI::FCMIntegrator"), linearV(group->getNumberParticles()), angularV(group->getNumberParticles()),
        temperature(par.temperature), dt(par.dt) {
    fcm = make_shared<FCM_impl<Kernel, KernelTorque>>(detail_fcm::initialize<Kernel, KernelTorque>(par, *sys));
    posWriteConnection = pd->getPosWriteRequestedSignal()->connect([this]() { posTouched = true; });
    reorderConnection = pd->getReorderSignal()->connect([this]() { posTouched = true; });
  }
  FCMIntegratorT(shared_ptr<ParticleData> pd, Parameters par) : FCMIntegratorT(make_shared<ParticleGroup>(pd, "All"), par) {}
  shared_ptr<FCM_impl<Kernel, KernelTorque>> getFCM_impl() { return fcm; }
  void forwardTime() override {
    steps++;
    for (auto &u : updatables) u->updateSimulationTime(steps * dt);
    if (steps == 1) for (auto &u : updatables) { u->updateTimeStep(dt); u->updateTemperature(temperature); u->updateBox(fcm->getBox()); }
    resetGroupForces(st);
    if (pd->isDirAllocated()) {  // computeCurrentForces, BDHI_FCM.cu:50-57
      auto torque = pd->getTorque(access::gpu, access::write);
      if (subgroup) detail::check(uammd_fill_zero_indexed(torque.raw(), groupIndex(), groupSize(), (int)sizeof(real4), (void *)st));
      else detail::check(uammd_fill_zero(torque.raw(), sizeof(real4) * torque.size(), (void *)st));
    }
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, st); }
    const int N = groupSize();
    if (!subgroup && !pd->isDirAllocated() && !pd->isTorqueAllocated()) {  // no rotation: the update rides in the solver's interpolation kernel
      const bool kept = !posTouched;  // (getPosWriteRequestedSignal: somebody may have moved the particles since our last step)
      auto pos = pd->getPos(access::gpu, access::readwrite);
      auto force = pd->getForce(access::gpu, access::read);
      fcm->stepEulerMaruyama(pos.raw(), force.raw(), linearV.d, N, temperature, 1.0 / std::sqrt(dt), dt, kept, st);
      posTouched = false;
      return;
    }
    {
      auto pos = pd->getPos(access::gpu, access::read);
      auto force = pd->getForce(access::gpu, access::read);
      auto torque = pd->getTorqueIfAllocated(access::gpu, access::read);
      fcm->computeHydrodynamicDisplacements(detail::groupRows(pos.raw(), subgroup.get(), posRows, st), detail::groupRows(force.raw(), subgroup.get(), forceRows, st),
                                            detail::groupRows(torque.raw(), subgroup.get(), torqueRows, st), linearV.d, angularV.d, N, temperature,
                                            1.0 / std::sqrt(dt), st);
    }
    auto pos = pd->getPos(access::gpu, access::readwrite);
    auto dir = pd->getDirIfAllocated(access::gpu, access::readwrite);
    detail::check(uammd_fcm_euler_maruyama_dir((float *)pos.raw(), (float *)dir.raw(), groupIndex(), (const float *)linearV.d,
                                               dir.raw() ? (const float *)angularV.d : nullptr, N, dt, (void *)st));
  }
};
using FCMIntegrator = FCMIntegratorT<>;
#else
// DOUBLE_PRECISION build of the same integrator (BDHI_FCM.cuh:155-199, BDHI_FCM.cu:7-119): the solve through FCM_impl's `_f64` entry
// points, the update pos += v dt through uammd_bdhi_euler_maruyama_f64.  No rotation in this build: particles with directions or torques
// are refused, as FCM_impl's double build refuses torques.
template <class Kernel = FCM_ns::Kernels::Gaussian> class FCMIntegratorT : public Integrator {
  using KernelTorque = FCM_ns::Kernels::GaussianTorque;
  shared_ptr<FCM_impl<Kernel, KernelTorque>> fcm;
  detail::DeviceArray<real3> linearV;
  real temperature, dt;
  uint steps = 0;
  hipStream_t st = 0;
  detail::DeviceArray<real4> posRows, forceRows;  // the rows of a proper subgroup, gathered
public:
  using Parameters = typename FCM_impl<Kernel, KernelTorque>::Parameters;
  FCMIntegratorT(shared_ptr<ParticleGroup> group, Parameters par)
      : Integrator(group, "BDHI::FCMIntegrator"), linearV(group->getNumberParticles()), temperature(par.temperature), dt(par.dt) {
    fcm = make_shared<FCM_impl<Kernel, KernelTorque>>(detail_fcm::initialize<Kernel, KernelTorque>(par, *sys));
  }
  FCMIntegratorT(shared_ptr<ParticleData> pd, Parameters par) : FCMIntegratorT(make_shared<ParticleGroup>(pd, "All"), par) {}
  shared_ptr<FCM_impl<Kernel, KernelTorque>> getFCM_impl() { return fcm; }
  void forwardTime() override {
    steps++;
    for (auto &u : updatables) u->updateSimulationTime(steps * dt);
    if (steps == 1) for (auto &u : updatables) { u->updateTimeStep(dt); u->updateTemperature(temperature); u->updateBox(fcm->getBox()); }
    if (pd->isDirAllocated() || pd->isTorqueAllocated())
      throw std::runtime_error("[BDHI::FCMIntegrator] rotation (directions / torques) is not part of the double-precision build on MI355X");
    resetGroupForces(st);
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, st); }
    const int N = groupSize();
    {
      auto pos = pd->getPos(access::gpu, access::read);
      auto force = pd->getForce(access::gpu, access::read);
      fcm->computeHydrodynamicDisplacements(detail::groupRows(pos.raw(), subgroup.get(), posRows, st), detail::groupRows(force.raw(), subgroup.get(), forceRows, st),
                                            linearV.d, N, temperature, 1.0 / std::sqrt(dt), st);
    }
    auto pos = pd->getPos(access::gpu, access::readwrite);
    detail::check(uammd_bdhi_euler_maruyama_f64((double *)pos.raw(), groupIndex(), (const double *)linearV.d, nullptr, nullptr, N, 0.0, dt, 0, (void *)st));
  }
  real getHydrodynamicRadius() { return fcm->getHydrodynamicRadius(); }
  real getSelfMobility() { return fcm->getSelfMobility(); }
};
using FCMIntegrator = FCMIntegratorT<>;
#endif
}  // namespace BDHI

// ---- BDHI::PSE (Integrator/BDHI/BDHI_PSE.cuh:79-176) and BDHI::EulerMaruyama<Method> (BDHI_EulerMaruyama.cu:125-166) --------------
namespace BDHI {
namespace pse_ns {
struct Parameters : BDHI::Parameters {  // PSE/utils.cuh:17-24
  real psi = 0.5;
  real shearStrain = 0;
};
}  // namespace pse_ns
#if defined(DOUBLE_PRECISION)
// DOUBLE_PRECISION build: the same interface on the `_f64` entry points — far field with rocFFT in double, near field over all pairs with the
// minimum image (the reference's double-precision tests hold one particle or a handful; the tuned cell-list / pair-record near field is
// the single-precision build below), near-field noise through the double-precision Lanczos solver.  No shear in this build.
class PSE {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;  // nullptr = all the particles (BDHI_PSE.cuh:171)
  detail::DeviceArray<real4> posRows, forceRows;
  int numberParticles() const { return pg ? pg->getNumberParticles() : pd->getNumParticles(); }
  const double *positions(const property_ptr<real4> &pos, hipStream_t st) { return (const double *)detail::groupRows((const real4 *)pos.raw(), pg.get(), posRows, st); }
  uammd_pse_near_f64 *nearField = nullptr;
  uammd_fcm_f64 *farField = nullptr;
  uammd_lanczos_f64 *solver = nullptr;
  real hydrodynamicRadius, M0, temperature, dt, tolerance;
  uint seedNear = 0, seedFar = 0;
  void far(const real4 *force, real3 *MF, real T, real prefactor, hipStream_t st) {
    const uint seed2 = T > 0 ? pd->getSystem()->rng().next32() : 0u;  // FarField.cuh:499
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_fcm_displacements_thermal_f64(farField, positions(pos, st), (const double *)force, numberParticles(), T, prefactor, seedFar, seed2,
                                                      (double *)MF, (void *)st));
  }
  void nearDeterministic(const real4 *force, real3 *MF, hipStream_t st) {   // NearField::Mdot, NearField.cuh:239-250: nothing without forces
    if (!force) return;
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_near_mdot_f64(nearField, positions(pos, st), (const double *)force, 4, numberParticles(), (double *)MF, (void *)st));
  }
  void nearStochastic(real3 *BdW, real T, real prefactor, hipStream_t st) {
    if (T == real(0.0)) return;
    const uint seed2 = pd->getSystem()->rng().next32();  // NearField.cuh:276
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_near_stochastic_f64(nearField, solver, positions(pos, st), numberParticles(), T, prefactor, seedNear, seed2, tolerance,
                                                (double *)BdW, (void *)st, nullptr));
  }
public:
  using Parameters = pse_ns::Parameters;
  PSE(shared_ptr<ParticleData> pd, Parameters par) : PSE(make_shared<ParticleGroup>(pd, "All"), par) {}  // BDHI_PSE.cuh:85-86
  PSE(shared_ptr<ParticleGroup> group, Parameters par)
      : pd(group->getParticleData()), pg(detail::subsetOrNull(group)), hydrodynamicRadius(par.hydrodynamicRadius), temperature(par.temperature),
        dt(par.dt), tolerance(par.tolerance) {
    M0 = (real)uammd_fcm_self_mobility(par.hydrodynamicRadius, par.viscosity, par.box.boxSize.x);
    const real3 L3 = par.box.boxSize;
    if (L3.x == real(0.0) && L3.y == real(0.0) && L3.z == real(0.0)) throw std::invalid_argument("Box of size zero detected");
    if (par.tolerance > 0.1) throw std::invalid_argument("Tolerance too high");  // PSE/initialization.cu:11-29
    if (par.shearStrain != real(0.0)) throw std::invalid_argument("[BDHI::PSE] the double-precision build of the library has no sheared boxes");
    const double L[3] = {L3.x, L3.y, L3.z};
    auto &rng = pd->getSystem()->rng();
    seedNear = rng.next32();  // NearField ctor first, then FarField (initialization.cu:57-59)
    detail::check(uammd_pse_near_create_f64(L, par.viscosity, par.hydrodynamicRadius, par.tolerance, par.psi, &nearField, nullptr, nullptr));
    seedFar = rng.next32();
    int c[3];
    detail::check(uammd_pse_far_raw_cells_f64(L, par.psi, par.tolerance, c));
    for (int &v : c) v = detail::nextFFTWiseSize(v);
    detail::check(uammd_pse_far_create_f64(L, c, par.viscosity, par.hydrodynamicRadius, par.tolerance, par.psi, par.shearStrain, &farField, nullptr, nullptr));
    detail::check(uammd_lanczos_create_f64(&solver));
  }
  PSE(const PSE &) = delete;
  ~PSE() {
    uammd_lanczos_destroy_f64(solver);
    uammd_pse_near_destroy_f64(nearField);
    uammd_fcm_destroy_f64(farField);
  }
  void setup_step(hipStream_t = 0) {}
  void finish_step(hipStream_t = 0) {}
  void computeMF(real3 *MF, hipStream_t st = 0) {  // :92-120
    const int N = numberParticles();
    detail::check(uammd_fill_zero(MF, sizeof(real3) * N, (void *)st));
    auto forceAll = pd->getForce(access::gpu, access::read);
    const real4 *force = detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st);
    far(force, MF, temperature, real(1.0 / std::sqrt(dt)), st);
    nearDeterministic(force, MF, st);
  }
  void computeBdW(real3 *BdW, hipStream_t st = 0) { nearStochastic(BdW, temperature, 1.0, st); }
  // the two halves of computeMF on their own (:101-120): each ADDS its part for the particles' current forces to MF
  void computeMFNearField(real3 *MF, hipStream_t st = 0) {
    auto forceAll = pd->getForce(access::gpu, access::read);
    nearDeterministic(detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st), MF, st);
  }
  void computeMFFarField(real3 *MF, hipStream_t st = 0) {
    auto forceAll = pd->getForce(access::gpu, access::read);
    far(detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st), MF, temperature, real(1.0 / std::sqrt(dt)), st);
  }
  void computeDivM(real3 *, hipStream_t = 0) {}   // (:125: the divergence of the PSE mobility is zero)
  void computeHydrodynamicDisplacements(real4 *force, real3 *MF, real T, real noise_prefactor, hipStream_t st = 0) {  // :135-155
    detail::check(uammd_fill_zero(MF, sizeof(real3) * numberParticles(), (void *)st));
    nearDeterministic(force, MF, st);
    nearStochastic(MF, T, noise_prefactor, st);
    far(force, MF, T, noise_prefactor, st);
  }
  real getHydrodynamicRadius() { return hydrodynamicRadius; }
  real getSelfMobility() { return M0; }
};
#else
class PSE {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;  // nullptr = all the particles (BDHI_PSE.cuh:171)
  detail::DeviceArray<real4> posRows, forceRows;
  int numberParticles() const { return pg ? pg->getNumberParticles() : pd->getNumParticles(); }
  // the members' positions, contiguous (the property itself without a proper subgroup)
  const float *positions(const property_ptr<real4> &pos, hipStream_t st) { return (const float *)detail::groupRows((const real4 *)pos.raw(), pg.get(), posRows, st); }
  uammd_pse_near *nearField = nullptr;
  uammd_fcm *farField = nullptr;
  scoped_connection posWriteConnection, reorderConnection;  // dropped in the destructor, before the handles they touch
  real hydrodynamicRadius, M0, temperature, dt;
  void far(const real4 *force, real3 *MF, real T, real prefactor, hipStream_t st) {
    const uint seed2 = T > 0 ? pd->getSystem()->rng().next32() : 0u;  // FarField.cuh:499
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_far_displacements(farField, positions(pos, st), (const float *)force, numberParticles(), T,
                                              prefactor, seed2, (float *)MF, (void *)st));
  }
  void nearStochastic(real3 *BdW, real T, real prefactor, hipStream_t st) {
    if (T == real(0.0)) return;
    const uint seed2 = pd->getSystem()->rng().next32();  // NearField.cuh:276
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_near_stochastic(nearField, positions(pos, st), numberParticles(), T, prefactor, seed2,
                                            (float *)BdW, (void *)st, nullptr));
  }
public:
  using Parameters = pse_ns::Parameters;
  PSE(shared_ptr<ParticleData> pd, Parameters par) : PSE(make_shared<ParticleGroup>(pd, "All"), par) {}  // BDHI_PSE.cuh:85-86
  PSE(shared_ptr<ParticleGroup> group, Parameters par)
      : pd(group->getParticleData()), pg(detail::subsetOrNull(group)), hydrodynamicRadius(par.hydrodynamicRadius), temperature(par.temperature),
        dt(par.dt) {
    M0 = (real)uammd_fcm_self_mobility(par.hydrodynamicRadius, par.viscosity, par.box.boxSize.x);
    const real3 L3 = par.box.boxSize;
    if (L3.x == real(0.0) && L3.y == real(0.0) && L3.z == real(0.0)) throw std::invalid_argument("Box of size zero detected");
    if (par.tolerance > 0.1) throw std::invalid_argument("Tolerance too high");  // PSE/initialization.cu:11-29
    const float L[3] = {L3.x, L3.y, L3.z};
    auto &rng = pd->getSystem()->rng();
    const uint seedNear = rng.next32();  // NearField ctor first, then FarField (initialization.cu:57-59)
    detail::check(uammd_pse_near_create(L, par.viscosity, par.hydrodynamicRadius, par.tolerance, par.psi, par.shearStrain, seedNear,
                                        &nearField, nullptr, nullptr));
    // CellList::update rebuilds only after a position write or a reorder (CellList.cuh:94-98,134-136)
    detail::check(uammd_pse_near_set_option(nearField, "lazy_list", 1));
    uammd_pse_near *nf = nearField;
    auto changed = [nf]() { uammd_pse_near_positions_changed(nf); };
    posWriteConnection = pd->getPosWriteRequestedSignal()->connect(changed);
    reorderConnection = pd->getReorderSignal()->connect(changed);
    const uint seedFar = rng.next32();
    int c[3];
    detail::check(uammd_pse_far_raw_cells(L, par.psi, par.tolerance, c));
    for (int &v : c) v = detail::nextFFTWiseSize(v);
    detail::check(uammd_pse_far_create(L, c, par.viscosity, par.hydrodynamicRadius, par.tolerance, par.psi, par.shearStrain, seedFar,
                                       &farField, nullptr, nullptr));
  }
  PSE(const PSE &) = delete;
  ~PSE() {
    posWriteConnection.disconnect();
    reorderConnection.disconnect();
    uammd_pse_near_destroy(nearField);
    uammd_fcm_destroy(farField);
  }
  void setup_step(hipStream_t = 0) {}
  void finish_step(hipStream_t = 0) {}
  void computeMF(real3 *MF, hipStream_t st = 0) {  // :92-120
    const int N = numberParticles();
    detail::check(uammd_fill_zero(MF, sizeof(real3) * N, (void *)st));
    auto forceAll = pd->getForce(access::gpu, access::read);
    const real4 *force = detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st);
    {  // (the near field's list and pair records are queued first: their one host read then happens while the far field runs)
      auto pos = pd->getPos(access::gpu, access::read);
      detail::check(uammd_pse_near_prepare(nearField, positions(pos, st), N, (void *)st));
    }
    far(force, MF, temperature, real(1.0 / std::sqrt(dt)), st);
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_near_mdot(nearField, positions(pos, st), (const float *)force, N, (float *)MF, (void *)st));
  }
  void computeBdW(real3 *BdW, hipStream_t st = 0) { nearStochastic(BdW, temperature, 1.0, st); }
  // the two halves of computeMF on their own (:101-120): each ADDS its part for the particles' current forces to MF
  void computeMFNearField(real3 *MF, hipStream_t st = 0) {
    auto forceAll = pd->getForce(access::gpu, access::read);
    const real4 *force = detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st);
    auto pos = pd->getPos(access::gpu, access::read);
    detail::check(uammd_pse_near_mdot(nearField, positions(pos, st), (const float *)force, numberParticles(), (float *)MF, (void *)st));
  }
  void computeMFFarField(real3 *MF, hipStream_t st = 0) {
    auto forceAll = pd->getForce(access::gpu, access::read);
    far(detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st), MF, temperature, real(1.0 / std::sqrt(dt)), st);
  }
  void computeDivM(real3 *, hipStream_t = 0) {}   // (:125: the divergence of the PSE mobility is zero)
  // computeMF and computeBdW (:92-126) as EulerMaruyama runs them when T > 0, queued so that the step's one wait for the GPU — the Lanczos
  // solve's convergence check — has work behind it: near-field list and pair records, the solve, the FAR FIELD from inside the solve
  // (uammd_pse_near_set_interleave: behind the check's kernels, while the host is busy with the check), then the near-field M F.  The same
  // calls with the same arguments; the two draws of System::rng() keep the reference's order (FarField.cuh:499, then NearField.cuh:276).
  void computeMFandBdW(real3 *MF, real3 *BdW, hipStream_t st = 0) {
    if (temperature == real(0.0)) { computeMF(MF, st); return; }
    const int N = numberParticles();
    detail::check(uammd_fill_zero(MF, sizeof(real3) * N, (void *)st));
    auto forceAll = pd->getForce(access::gpu, access::read);
    auto posAll = pd->getPos(access::gpu, access::read);
    const float *posRowsPtr = positions(posAll, st);
    const real4 *force = detail::groupRows((const real4 *)forceAll.raw(), pg.get(), forceRows, st);
    const uint seedFar = pd->getSystem()->rng().next32();
    const uint seedNear = pd->getSystem()->rng().next32();
    // (no uammd_pse_near_prepare here: the solve below starts with the same list and record launches and leaves the copy of the records'
    // counters to its own noise kernel)
    // the far field in two halves around the check: spreading and forward transforms while the host answers it, the rest while the host
    // reacts to its outcome
    struct Far { uammd_fcm *solver; const float *pos, *force; int N; float T, prefactor; uint seed; float *MF; };
    Far far{farField, posRowsPtr, (const float *)force, N, (float)temperature, (float)(1.0 / std::sqrt(dt)), seedFar, (float *)MF};
    uammd_interleave_fn firstHalf = [](void *c, void *stream) -> int {
      const Far *f = static_cast<const Far *>(c);
      return uammd_pse_far_displacements_half(f->solver, f->pos, f->force, f->N, f->T, f->prefactor, f->seed, f->MF, 1, stream);
    };
    uammd_interleave_fn secondHalf = [](void *c, void *stream) -> int {
      const Far *f = static_cast<const Far *>(c);
      return uammd_pse_far_displacements_half(f->solver, f->pos, f->force, f->N, f->T, f->prefactor, f->seed, f->MF, 2, stream);
    };
    detail::check(uammd_pse_near_set_interleave_early(nearField, firstHalf, &far));
    detail::check(uammd_pse_near_set_interleave(nearField, secondHalf, &far));
    // (the near field's M F rides on the solve's first product: the pair records are streamed once for the noise vector and F)
    detail::check(uammd_pse_near_set_mdot_rider(nearField, (const float *)force, (float *)MF));
    detail::check(uammd_pse_near_stochastic(nearField, posRowsPtr, N, temperature, real(1.0), seedNear, (float *)BdW, (void *)st, nullptr));
  }
  void computeHydrodynamicDisplacements(real4 *force, real3 *MF, real T, real noise_prefactor, hipStream_t st = 0) {  // :135-155
    const int N = numberParticles();
    detail::check(uammd_fill_zero(MF, sizeof(real3) * N, (void *)st));
    if (force) {   // NearField::Mdot does nothing without forces (NearField.cuh:239-250)
      auto pos = pd->getPos(access::gpu, access::read);
      detail::check(uammd_pse_near_mdot(nearField, positions(pos, st), (const float *)force, N, (float *)MF, (void *)st));
    }
    nearStochastic(MF, T, noise_prefactor, st);
    far(force, MF, T, noise_prefactor, st);
  }
  void setShearStrain(real g) {
    detail::check(uammd_pse_near_set_shear_strain(nearField, g));
    detail::check(uammd_pse_far_set_shear_strain(farField, g));
  }
  real getHydrodynamicRadius() { return hydrodynamicRadius; }
  real getSelfMobility() { return M0; }
};
#endif

// BDHI::EulerMaruyama<Method> (Integrator/BDHI/BDHI_EulerMaruyama.cuh:55-110, .cu:30-166): Method = FCM, PSE (both precisions), Lanczos, Cholesky
template <class Method> class EulerMaruyama : public Integrator {
  using Parameters_t = typename Method::Parameters;
  Parameters_t par;
  shared_ptr<Method> bdhi;
  detail::DeviceArray<real3> MF, BdW;
  int steps = 0;
  hipStream_t stream = 0;
public:
  using Parameters = Parameters_t;
  EulerMaruyama(shared_ptr<ParticleGroup> group, Parameters par)  // BDHI_EulerMaruyama.cuh:67, .cu:30-60
      : Integrator(group, "BDHI::EulerMaruyama"), par(par), bdhi(make_shared<Method>(group, par)), MF(group->getNumberParticles()),
        BdW(group->getNumberParticles() + 1) {}
  EulerMaruyama(shared_ptr<ParticleData> pd, Parameters par) : EulerMaruyama(make_shared<ParticleGroup>(pd, "All"), par) {}  // :69-70
  shared_ptr<Method> getScheme() { return bdhi; }  // :81
  real getHydrodynamicRadius() { return bdhi->getHydrodynamicRadius(); }
  real getSelfMobility() { return bdhi->getSelfMobility(); }
  shared_ptr<Method> getMethod() { return bdhi; }
private:
  // computeMF then computeBdW (BDHI_EulerMaruyama.cu:140-150) — or, for a method that can interleave the two (PSE: its far field behind
  // the Lanczos solve's one wait), its computeMFandBdW
  template <class M> auto mobilityAndNoise(M &m, int) -> decltype(m.computeMFandBdW(MF.d, BdW.d, stream), void()) {
    m.computeMFandBdW(MF.d, BdW.d, stream);
  }
  template <class M> void mobilityAndNoise(M &m, long) {
    m.computeMF(MF.d, stream);
    m.computeBdW(BdW.d, stream);
  }
public:
  void forwardTime() override {
    steps++;
    for (auto &u : updatables) u->updateSimulationTime(steps * par.dt);
    if (steps == 1)
      for (auto &u : updatables) { u->updateTimeStep(par.dt); u->updateTemperature(par.temperature); u->updateBox(par.box); u->updateViscosity(par.viscosity); }
    resetGroupForces(stream);  // .cu:115-123
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, stream); }
    bdhi->setup_step(stream);
    if (par.temperature > 0) mobilityAndNoise(*bdhi, 0);
    else bdhi->computeMF(MF.d, stream);
    const real sqrt2Tdt = std::sqrt(2 * par.dt * par.temperature);
    bdhi->finish_step(stream);
    real K[9] = {0};
    const bool shear = par.K.size() == 3;
    if (shear) for (int i = 0; i < 3; ++i) { K[3 * i] = par.K[i].x; K[3 * i + 1] = par.K[i].y; K[3 * i + 2] = par.K[i].z; }
    auto pos = pd->getPos(access::gpu, access::readwrite);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi_euler_maruyama_f64((double *)pos.raw(), groupIndex(), (const double *)MF.d, par.temperature > 0 ? (const double *)BdW.d : nullptr,
                                                shear ? K : nullptr, groupSize(), sqrt2Tdt, par.dt, par.is2D, (void *)stream));
#else
    detail::check(uammd_bdhi_euler_maruyama((float *)pos.raw(), groupIndex(), (const float *)MF.d, par.temperature > 0 ? (const float *)BdW.d : nullptr,
                                            shear ? K : nullptr, groupSize(), sqrt2Tdt, par.dt, par.is2D, (void *)stream));
#endif
  }
};
// BDHI::Lanczos (Integrator/BDHI/BDHI_Lanczos.cuh:20-67): open boundaries, dense RPY mobility, matrix free (both precisions)
class Lanczos {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;  // nullptr = all the particles (BDHI_Lanczos.cuh:51)
  detail::DeviceArray<real4> posRows, forceRows;
  detail::DeviceArray<real> radiusRows;
  int numberParticles() const { return pg ? pg->getNumberParticles() : pd->getNumParticles(); }
  BDHI::Parameters par;
#if defined(DOUBLE_PRECISION)
  uammd_lanczos_f64 *solver = nullptr;
#else
  uammd_lanczos *solver = nullptr;
#endif
  detail::DeviceArray<real3> noise;
  Xorshift128plus gen;  // the reference uses cuRAND here (stream unpinned): Box-Muller on the System generator family instead
public:
  using Parameters = BDHI::Parameters;
  Lanczos(shared_ptr<ParticleData> pd, Parameters par) : Lanczos(make_shared<ParticleGroup>(pd, "All"), par) {}  // :28-29
  Lanczos(shared_ptr<ParticleGroup> group, Parameters par)
      : pd(group->getParticleData()), pg(detail::subsetOrNull(group)), par(par), noise(group->getNumberParticles()) {
    if (par.hydrodynamicRadius < 0 && !pd->isRadiusAllocated())
      System::log<System::CRITICAL>("[BDHI::Lanczos] You need to provide Lanczos with either an hydrodynamic radius or via the individual particle radius.");
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_lanczos_create_f64(&solver));
#else
    detail::check(uammd_lanczos_create(&solver));
#endif
    gen.setSeed(pd->getSystem()->rng().next());
  }
  Lanczos(const Lanczos &) = delete;
#if defined(DOUBLE_PRECISION)
  ~Lanczos() { uammd_lanczos_destroy_f64(solver); }
#else
  ~Lanczos() { uammd_lanczos_destroy(solver); }
#endif
  void setup_step(hipStream_t = 0) {}
  void finish_step(hipStream_t = 0) {}
  real getHydrodynamicRadius() { return par.hydrodynamicRadius; }
  real getSelfMobility() { return par.hydrodynamicRadius < 0 ? real(-1.0) : real(1.0 / (6.0 * M_PI * par.viscosity * par.hydrodynamicRadius)); }
  void computeMF(real3 *MF, hipStream_t st = 0) {
    auto pos = pd->getPos(access::gpu, access::read);
    auto force = pd->getForce(access::gpu, access::read);
    auto radius = par.hydrodynamicRadius > 0 ? property_ptr<real>() : pd->getRadiusIfAllocated(access::gpu, access::read);
    const real4 *posNow = detail::groupRows((const real4 *)pos.raw(), pg.get(), posRows, st);
    const real4 *forceNow = detail::groupRows((const real4 *)force.raw(), pg.get(), forceRows, st);
    const real *radiusNow = detail::groupRows((const real *)radius.raw(), pg.get(), radiusRows, st);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_rpy_nbody_mdot_f64((const double *)posNow, (const double *)forceNow, 4, radiusNow, par.hydrodynamicRadius, par.viscosity,
                                           numberParticles(), (double *)MF, (void *)st));
#else
    detail::check(uammd_rpy_nbody_mdot((const float *)posNow, (const float *)forceNow, 4, radiusNow, par.hydrodynamicRadius, par.viscosity,
                                       numberParticles(), (float *)MF, (void *)st));
#endif
  }
  void computeBdW(real3 *BdW, hipStream_t st = 0) {
    if (!(par.temperature > real(0.0))) return;
    const int N = numberParticles();
    std::vector<real3> h(N);
    for (auto &v : h) {  // standard normals, Box-Muller
      real g[4];
      for (int k = 0; k < 4; k += 2) {
        const double u1 = gen.uniform(1e-300, 1.0), u2 = gen.uniform(0.0, 1.0);
        const double r = std::sqrt(-2.0 * std::log(u1));
        g[k] = real(r * std::cos(2 * M_PI * u2)); g[k + 1] = real(r * std::sin(2 * M_PI * u2));
      }
      v = make_real3(g[0], g[1], g[2]);
    }
    detail::hipCheck(hipMemcpy(noise.d, h.data(), sizeof(real3) * N, hipMemcpyHostToDevice), "hipMemcpy");
    auto pos = pd->getPos(access::gpu, access::read);
    auto radius = par.hydrodynamicRadius > 0 ? property_ptr<real>() : pd->getRadiusIfAllocated(access::gpu, access::read);
    const real4 *posNow = detail::groupRows((const real4 *)pos.raw(), pg.get(), posRows, st);
    const real *radiusNow = detail::groupRows((const real *)radius.raw(), pg.get(), radiusRows, st);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_rpy_lanczos_bdw_f64(solver, (const double *)posNow, radiusNow, par.hydrodynamicRadius, par.viscosity, N, (const double *)noise.d,
                                            par.tolerance, (double *)BdW, (void *)st, nullptr));
#else
    detail::check(uammd_rpy_lanczos_bdw(solver, (const float *)posNow, radiusNow, par.hydrodynamicRadius, par.viscosity, N, (const float *)noise.d,
                                        par.tolerance, (float *)BdW, (void *)st, nullptr));
#endif
  }
};

// BDHI::Cholesky (Integrator/BDHI/BDHI_Cholesky.cuh:37-80): open boundaries, dense RPY mobility stored in full, Cholesky factor for
// the noise (rocSOLVER potrf + rocBLAS symv / trmv behind uammd_bdhi_cholesky_*)
class Cholesky {
  shared_ptr<ParticleData> pd;
  shared_ptr<ParticleGroup> pg;  // nullptr = all the particles (BDHI_Cholesky.cuh:58); the C ABI takes the group's index
  int numberParticles() const { return pg ? pg->getNumberParticles() : pd->getNumParticles(); }
  const int *index() { return pg ? pg->getIndicesRawPtr(access::gpu) : nullptr; }
  BDHI::Parameters par;
#if defined(DOUBLE_PRECISION)
  uammd_bdhi_cholesky_f64 *h = nullptr;
#else
  uammd_bdhi_cholesky *h = nullptr;
#endif
  Xorshift128plus gen;  // cuRAND in the reference (stream unpinned): Box-Muller on the System generator family instead
public:
  using Parameters = BDHI::Parameters;
  Cholesky(shared_ptr<ParticleData> pd, Parameters par) : Cholesky(make_shared<ParticleGroup>(pd, "All"), par) {}  // :36-37
  Cholesky(shared_ptr<ParticleGroup> group, Parameters par) : pd(group->getParticleData()), pg(detail::subsetOrNull(group)), par(par) {
    if (par.hydrodynamicRadius < 0 && !pd->isRadiusAllocated())
      System::log<System::CRITICAL>("[BDHI::Cholesky] You need to provide Cholesky with either an hydrodynamic radius or via the individual particle radius.");
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi_cholesky_create_f64(numberParticles(), par.viscosity, par.hydrodynamicRadius, &h));
#else
    detail::check(uammd_bdhi_cholesky_create(numberParticles(), par.viscosity, par.hydrodynamicRadius, &h));
#endif
    gen.setSeed(pd->getSystem()->rng().next());
  }
  Cholesky(const Cholesky &) = delete;
#if defined(DOUBLE_PRECISION)
  ~Cholesky() { uammd_bdhi_cholesky_destroy_f64(h); }
#else
  ~Cholesky() { uammd_bdhi_cholesky_destroy(h); }
#endif
  void init() {}
  void finish_step(hipStream_t = 0) {}
  real getHydrodynamicRadius() { return par.hydrodynamicRadius; }
  real getSelfMobility() { return par.hydrodynamicRadius < 0 ? real(-1.0) : real(1.0 / (6.0 * M_PI * par.viscosity * par.hydrodynamicRadius)); }
  void setup_step(hipStream_t st = 0) {
    auto pos = pd->getPos(access::gpu, access::read);
    auto radius = pd->getRadiusIfAllocated(access::gpu, access::read);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi_cholesky_setup_step_f64(h, (const double *)pos.raw(), index(), radius.raw(), (void *)st));
#else
    detail::check(uammd_bdhi_cholesky_setup_step(h, (const float *)pos.raw(), index(), radius.raw(), (void *)st));
#endif
  }
  void computeMF(real3 *MF, hipStream_t st = 0) {
    auto pos = pd->getPos(access::gpu, access::read);
    auto force = pd->getForce(access::gpu, access::read);
    auto radius = pd->getRadiusIfAllocated(access::gpu, access::read);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi_cholesky_mf_f64(h, (const double *)pos.raw(), (const double *)force.raw(), index(), radius.raw(), (double *)MF, (void *)st));
#else
    detail::check(uammd_bdhi_cholesky_mf(h, (const float *)pos.raw(), (const float *)force.raw(), index(), radius.raw(), (float *)MF, (void *)st));
#endif
  }
  void computeBdW(real3 *BdW, hipStream_t st = 0) {
    const int N = numberParticles();
    std::vector<real3> hn(N);
    for (auto &v : hn) {  // standard normals, Box-Muller
      real g[4];
      for (int k = 0; k < 4; k += 2) {
        const double u1 = gen.uniform(1e-300, 1.0), u2 = gen.uniform(0.0, 1.0);
        const double r = std::sqrt(-2.0 * std::log(u1));
        g[k] = real(r * std::cos(2 * M_PI * u2)); g[k + 1] = real(r * std::sin(2 * M_PI * u2));
      }
      v = make_real3(g[0], g[1], g[2]);
    }
    detail::hipCheck(hipMemcpyAsync(BdW, hn.data(), sizeof(real3) * N, hipMemcpyHostToDevice, st), "hipMemcpy");
    detail::hipCheck(hipStreamSynchronize(st), "hipStreamSynchronize");
    auto pos = pd->getPos(access::gpu, access::read);
    auto radius = pd->getRadiusIfAllocated(access::gpu, access::read);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi_cholesky_bdw_f64(h, (const double *)pos.raw(), index(), radius.raw(), (double *)BdW, (void *)st));
#else
    detail::check(uammd_bdhi_cholesky_bdw(h, (const float *)pos.raw(), index(), radius.raw(), (float *)BdW, (void *)st));
#endif
  }
};

}  // namespace BDHI

// ---- BDHI::True2D / BDHI::Quasi2D (Integrator/Hydro/BDHI_quasi2D.cuh:155-257): hydrodynamics of particles confined to a plane ------
namespace BDHI {
namespace BDHI2D_ns {
struct True2D { static constexpr int id = UAMMD_BDHI2D_TRUE2D; static constexpr bool hasThermalDrift() { return false; } };
struct Quasi2D { static constexpr int id = UAMMD_BDHI2D_QUASI2D; static constexpr bool hasThermalDrift() { return true; } };
}  // namespace BDHI2D_ns
template <class HydroKernel> class BDHI2D : public Integrator {
#if defined(DOUBLE_PRECISION)
  uammd_bdhi2d_f64 *h = nullptr;
#else
  uammd_bdhi2d *h = nullptr;
#endif
  detail::DeviceArray<real2> particleVels;
  Box box;
  real temperature, dt, viscosity;
  int step = 0;
  int cellsOut[2] = {0, 0}, support = 0;
  hipStream_t st = 0;
  detail::DeviceArray<real4> posRows, forceRows;
public:
  struct Parameters : BDHI::Parameters {
    int2 cells = make_int2(-1, -1);
  };
  BDHI2D(shared_ptr<ParticleData> pd, Parameters par) : BDHI2D(make_shared<ParticleGroup>(pd, "All"), par) {}  // BDHI_quasi2D.cuh:189-190
  BDHI2D(shared_ptr<ParticleGroup> group, Parameters par)
      : Integrator(group, "BDHI::BDHI2D"), particleVels(group->getNumberParticles()), box(make_real3(par.box.boxSize.x, par.box.boxSize.y, 0)),
        temperature(par.temperature), dt(par.dt), viscosity(par.viscosity) {
#if defined(DOUBLE_PRECISION)
    uammd_bdhi2d_parameters_f64 p{};
#else
    uammd_bdhi2d_parameters p{};
#endif
    p.boxSize[0] = par.box.boxSize.x; p.boxSize[1] = par.box.boxSize.y;
    p.hydrodynamicRadius = par.hydrodynamicRadius; p.viscosity = par.viscosity; p.temperature = par.temperature; p.dt = par.dt;
    p.cells[0] = par.cells.x; p.cells[1] = par.cells.y;
    p.seed = sys->rng().next32();  // .cu:28
    p.kernel = HydroKernel::id;
    // "Invalid box" / "Invalid hydrodynamic radius"
#if defined(DOUBLE_PRECISION)
    if (uammd_bdhi2d_create_f64(&p, &h, cellsOut, &support) != 0) throw std::runtime_error(uammd_hip_last_error());
#else
    if (uammd_bdhi2d_create(&p, &h, cellsOut, &support) != 0) throw std::runtime_error(uammd_hip_last_error());
#endif
  }
  BDHI2D(const BDHI2D &) = delete;
#if defined(DOUBLE_PRECISION)
  ~BDHI2D() { uammd_bdhi2d_destroy_f64(h); }
#else
  ~BDHI2D() { uammd_bdhi2d_destroy(h); }
#endif
  int2 getCells() const { return make_int2(cellsOut[0], cellsOut[1]); }
  int getSupport() const { return support; }
  void forwardTime() override {
    for (auto &u : updatables) u->updateSimulationTime(step * dt);
    step++;
    if (step == 1) for (auto &u : updatables) { u->updateTemperature(temperature); u->updateBox(box); u->updateTimeStep(dt); }
    resetGroupForces(st);
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, st); }
    const int N = groupSize();
    {
      auto pos = pd->getPos(access::gpu, access::read);
      auto force = pd->getForce(access::gpu, access::read);
      const real4 *posRowsNow = detail::groupRows((const real4 *)pos.raw(), subgroup.get(), posRows, st);
      const real4 *forceRowsNow = interactors.empty() ? nullptr : detail::groupRows((const real4 *)force.raw(), subgroup.get(), forceRows, st);
#if defined(DOUBLE_PRECISION)
      detail::check(uammd_bdhi2d_velocities_f64(h, (const double *)posRowsNow, (const double *)forceRowsNow, N, (double *)particleVels.d, (void *)st));
#else
      detail::check(uammd_bdhi2d_velocities(h, (const float *)posRowsNow, (const float *)forceRowsNow, N, (float *)particleVels.d, (void *)st));
#endif
    }
    auto pos = pd->getPos(access::gpu, access::readwrite);
    real4 *rows = subgroup ? posRows.d : pos.raw();  // (a proper subgroup: the gathered rows above, moved, then written back through the index)
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_bdhi2d_update_positions_f64((double *)rows, (const double *)particleVels.d, N, dt, (void *)st));
#else
    detail::check(uammd_bdhi2d_update_positions((float *)rows, (const float *)particleVels.d, N, dt, (void *)st));
#endif
    detail::scatterRows((const real4 *)rows, pos.raw(), subgroup.get(), st);
  }
};
using True2D = BDHI2D<BDHI2D_ns::True2D>;
using Quasi2D = BDHI2D<BDHI2D_ns::Quasi2D>;
}  // namespace BDHI
#if !defined(DOUBLE_PRECISION)   // (single-precision backends only again, down to Poisson)

// ---- BDHI::FIB (Integrator/BDHI/FIB/FIB.cuh:131-236): fluctuating immersed boundary on a staggered grid ------------------------------
namespace BDHI {
class FIB : public Integrator {
  uammd_fib *h = nullptr;
  Box box;
  real temperature, viscosity, dt, hydrodynamicRadius = 0;
  int cellsOut[3] = {0, 0, 0};
  unsigned long long step = 0;
  detail::DeviceArray<real4> posRows, forceRows;  // the rows of a proper subgroup, gathered
public:
  enum Scheme { MIDPOINT, IMPROVED_MIDPOINT };
  struct Parameters {
    real temperature = 0;
    real viscosity = 1;
    real hydrodynamicRadius = -1;
    real dt = 0;
    Box box;
    int3 cells = make_int3(-1, -1, -1);
    Scheme scheme = Scheme::IMPROVED_MIDPOINT;
    real tolerance = 1e-5;
  };
  FIB(shared_ptr<ParticleData> pd, Parameters par) : FIB(make_shared<ParticleGroup>(pd, "All"), par) {}  // FIB.cuh:163-164
  FIB(shared_ptr<ParticleGroup> group, Parameters par)
      : Integrator(group, "BDHI::FIB"), box(par.box), temperature(par.temperature), viscosity(par.viscosity), dt(par.dt) {
    uammd_fib_parameters p{};
    p.boxSize[0] = par.box.boxSize.x; p.boxSize[1] = par.box.boxSize.y; p.boxSize[2] = par.box.boxSize.z;
    p.temperature = par.temperature; p.viscosity = par.viscosity; p.hydrodynamicRadius = par.hydrodynamicRadius; p.dt = par.dt;
    p.cells[0] = par.cells.x; p.cells[1] = par.cells.y; p.cells[2] = par.cells.z;
    p.scheme = (int)par.scheme;
    p.seed = sys->rng().next32();
    float rh = 0;
    if (uammd_fib_create(&p, &h, cellsOut, &rh) != 0) System::log<System::CRITICAL>("%s", uammd_hip_last_error());  // FIB.cu:95-103
    hydrodynamicRadius = rh;
  }
  FIB(const FIB &) = delete;
  ~FIB() { uammd_fib_destroy(h); }
  real getSelfMobility() { return uammd_fib_self_mobility(hydrodynamicRadius, viscosity, box.boxSize.x); }
  real getHydrodynamicRadius() { return hydrodynamicRadius; }
  real getCellSize() { return box.boxSize.x / cellsOut[0]; }
  void forwardTime() override {
    step++;
    if (step == 1) for (auto &u : updatables) { u->updateSimulationTime(0); u->updateTimeStep(dt); u->updateTemperature(temperature); u->updateBox(box); }
    resetGroupForces(nullptr);
    for (auto &f : interactors) { Interactor::Computables c; c.force = true; f->sum(c, 0); }
    {
      auto pos = pd->getPos(access::gpu, access::readwrite);
      auto force = pd->getForce(access::gpu, access::read);
      real4 *rows = const_cast<real4 *>(detail::groupRows((const real4 *)pos.raw(), subgroup.get(), posRows, nullptr));
      detail::check(uammd_fib_forward(h, (float *)rows, (const float *)detail::groupRows((const real4 *)force.raw(), subgroup.get(), forceRows, nullptr),
                                      groupSize(), nullptr));
      detail::scatterRows((const real4 *)rows, pos.raw(), subgroup.get(), nullptr);
    }
    for (auto &u : updatables) u->updateSimulationTime(step * dt);
  }
};
}  // namespace BDHI

// ---- Hydro::ICM (Integrator/Hydro/ICM.cuh:123-231): inertial coupling to an incompressible fluctuating fluid ---------------------
namespace Hydro {
class ICM : public Integrator {
  uammd_icm *h = nullptr;
  Box box;
  real temperature, viscosity, dt, hydrodynamicRadius = 0;
  int cellsOut[3] = {0, 0, 0};
  uint step = 0;
  detail::DeviceArray<real3> collocated;
  detail::DeviceArray<real4> posRows, forceRows;  // the rows of a proper subgroup, gathered
  std::vector<real3> h_collocated;
public:
  struct Parameters {
    real temperature = 0;
    real viscosity = -1;
    real density = -1;
    real hydrodynamicRadius = -1;
    real dt = 0;
    Box box;
    int3 cells = make_int3(-1, -1, -1);
    bool sumThermalDrift = false;
    bool removeTotalMomentum = true;
  };
  ICM(shared_ptr<ParticleData> pd, Parameters par) : ICM(make_shared<ParticleGroup>(pd, "All"), par) {}  // ICM.cuh:174-175
  ICM(shared_ptr<ParticleGroup> group, Parameters par)
      : Integrator(group, "Hydro::ICM"), box(par.box), temperature(par.temperature), viscosity(par.viscosity), dt(par.dt), collocated(0) {
    uammd_icm_parameters p{};
    p.boxSize[0] = par.box.boxSize.x; p.boxSize[1] = par.box.boxSize.y; p.boxSize[2] = par.box.boxSize.z;
    p.temperature = par.temperature; p.viscosity = par.viscosity; p.density = par.density; p.hydrodynamicRadius = par.hydrodynamicRadius;
    p.dt = par.dt;
    p.cells[0] = par.cells.x; p.cells[1] = par.cells.y; p.cells[2] = par.cells.z;
    p.sumThermalDrift = par.sumThermalDrift; p.removeTotalMomentum = par.removeTotalMomentum;
    p.seed = sys->rng().next32();
    float rh = 0;
    if (uammd_icm_create(&p, &h, cellsOut, &rh) != 0) System::log<System::CRITICAL>("%s", uammd_hip_last_error());  // ICM.cu:833-839, :869-872
    hydrodynamicRadius = rh;
  }
  ICM(const ICM &) = delete;
  ~ICM() { uammd_icm_destroy(h); }
  real sumEnergy() override { return 0; }
  real getSelfMobility() { return 1.0 / (6 * M_PI * viscosity * hydrodynamicRadius) * (1 - 2.837297 * hydrodynamicRadius / box.boxSize.x); }
  real getHydrodynamicRadius() { return hydrodynamicRadius; }
  int3 getNumberFluidCells() { return make_int3(cellsOut[0], cellsOut[1], cellsOut[2]); }
  // cell-centred fluid velocities, cell (i,j,k) at i + (j + k*n.y)*n.x (ICM.cuh:176-199)
  const real3 *getFluidVelocities(access::location dev) {
    const size_t nc = (size_t)cellsOut[0] * cellsOut[1] * cellsOut[2];
    collocated.resize(nc);
    detail::check(uammd_icm_get_fluid_velocity(h, (float *)collocated.d, 1, nullptr));
    if (dev == access::gpu) return collocated.d;
    if (dev != access::cpu) throw std::runtime_error("Invalid device");
    h_collocated.resize(nc);
    detail::hipCheck(hipMemcpy(h_collocated.data(), collocated.d, sizeof(real3) * nc, hipMemcpyDeviceToHost), "hipMemcpy");
    return h_collocated.data();
  }
  void forwardTime() override {
    step++;
    Interactor::Computables c; c.force = true;
    if (step == 1) {
      for (auto &u : updatables) { u->updateTemperature(temperature); u->updateTimeStep(dt); u->updateBox(box); u->updateSimulationTime(0); }
      for (auto &f : interactors) f->sum(c, 0);
    }
    const int N = groupSize();
    {
      auto pos = pd->getPos(access::gpu, access::readwrite);
      real4 *rows = const_cast<real4 *>(detail::groupRows((const real4 *)pos.raw(), subgroup.get(), posRows, nullptr));
      detail::check(uammd_icm_predictor(h, (float *)rows, N, nullptr));
      detail::scatterRows((const real4 *)rows, pos.raw(), subgroup.get(), nullptr);
    }
    for (auto &u : updatables) u->updateSimulationTime((step - 0.5) * dt);
    if (!interactors.empty()) {
      resetGroupForces(nullptr);
      for (auto &f : interactors) f->sum(c, 0);
    }
    {
      auto pos = pd->getPos(access::gpu, access::readwrite);
      auto force = pd->getForce(access::gpu, access::readwrite);
      real4 *rows = const_cast<real4 *>(detail::groupRows((const real4 *)pos.raw(), subgroup.get(), posRows, nullptr));
      detail::check(uammd_icm_fluid_and_corrector(h, (float *)rows, interactors.empty() ? nullptr : (const float *)detail::groupRows((const real4 *)force.raw(), subgroup.get(), forceRows, nullptr), N, nullptr));
      detail::scatterRows((const real4 *)rows, pos.raw(), subgroup.get(), nullptr);
      if (subgroup) detail::check(uammd_fill_zero_indexed(force.raw(), groupIndex(), N, (int)sizeof(real4), nullptr));
      else detail::check(uammd_fill_zero(force.raw(), sizeof(real4) * force.size(), nullptr));  // correctorStep, :1176-1181
    }
    for (auto &u : updatables) u->updateSimulationTime(step * dt);
  }
};
}  // namespace Hydro

#endif   // !DOUBLE_PRECISION

// ---- Poisson (Interactor/SpectralEwaldPoisson.cuh:83-136): triply periodic electrostatics, spectral Ewald (both precisions) -----
class Poisson : public Interactor {
#if defined(DOUBLE_PRECISION)
  uammd_poisson_f64 *h = nullptr;
  uammd_poisson_info_f64 info{};
#else
  uammd_poisson *h = nullptr;
  uammd_poisson_info info{};
#endif
  detail::DeviceArray<real4> posRows, forceRows;  // the rows of a proper subgroup, gathered
  detail::DeviceArray<real> chargeRows, energyRows;
  int numberParticles() const { return subgroup ? subgroup->getNumberParticles() : pd->getNumParticles(); }
public:
  struct Parameters {  // SpectralEwaldPoisson.cuh:94-103; cells and support are never read by the reference's constructor
    real upsampling = -1.0;
    int3 cells = make_int3(-1, -1, -1);
    Box box;
    real epsilon = -1;
    real tolerance = 1e-5;
    real gw = -1;
    int support = -1;
    real split = -1;
  };
  Poisson(shared_ptr<ParticleData> pd, Parameters par) : Poisson(make_shared<ParticleGroup>(pd, "All"), par) {}  // SpectralEwaldPoisson.cuh:103-104
  Poisson(shared_ptr<ParticleGroup> group, Parameters par) : Interactor(group, "IBM::Poisson") {
#if defined(DOUBLE_PRECISION)
    uammd_poisson_parameters_f64 p{};
#else
    uammd_poisson_parameters p{};
#endif
    p.boxSize[0] = par.box.boxSize.x; p.boxSize[1] = par.box.boxSize.y; p.boxSize[2] = par.box.boxSize.z;
    p.epsilon = par.epsilon; p.tolerance = par.tolerance; p.gw = par.gw; p.split = par.split; p.upsampling = par.upsampling;
#if defined(DOUBLE_PRECISION)
    if (uammd_poisson_create_f64(&p, &h, &info) != 0) {
#else
    if (uammd_poisson_create(&p, &h, &info) != 0) {
#endif
      const std::string msg = uammd_hip_last_error();
      if (msg.find("[Poisson]") != std::string::npos) throw std::invalid_argument(msg);  // .cu:95-102, :111-116
      throw std::runtime_error(msg);
    }
  }
  Poisson(const Poisson &) = delete;
#if defined(DOUBLE_PRECISION)
  ~Poisson() { uammd_poisson_destroy_f64(h); }
#else
  ~Poisson() { uammd_poisson_destroy(h); }
#endif
  // far field (always adds q E to the forces AND q phi to the energies, .cu:561-579), then the near-field passes
  void sum(Computables comp, hipStream_t st = 0) override {
    if (comp.virial) throw std::runtime_error("[Poisson] not implemented");
    auto pos = pd->getPos(access::gpu, access::read);
    auto charge = pd->getCharge(access::gpu, access::read);
    auto force = pd->getForce(access::gpu, access::readwrite);
    auto energy = pd->getEnergy(access::gpu, access::readwrite);
    ParticleGroup *g = subgroup.get();  // a proper subgroup: the members' rows gathered, the sums written back through the group's index
    real4 *f = const_cast<real4 *>(detail::groupRows((const real4 *)force.raw(), g, forceRows, st));
    real *e = const_cast<real *>(detail::groupRows((const real *)energy.raw(), g, energyRows, st));
    const real4 *posNow = detail::groupRows((const real4 *)pos.raw(), g, posRows, st);
    const real *chargeNow = detail::groupRows((const real *)charge.raw(), g, chargeRows, st);
#if defined(DOUBLE_PRECISION)
    detail::check(uammd_poisson_sum_f64(h, (const double *)posNow, chargeNow, numberParticles(), (double *)f, e, comp.force, comp.energy, (void *)st));
#else
    detail::check(uammd_poisson_sum(h, (const float *)posNow, chargeNow, numberParticles(), (float *)f, e, comp.force, comp.energy, (void *)st));
#endif
    detail::scatterRows((const real4 *)f, force.raw(), g, st);
    detail::scatterRows((const real *)e, energy.raw(), g, st);
  }
  // (Ex, Ey, Ez, phi) at the particles; like the reference's call, the far field also lands on the forces and energies
  std::vector<real4> computeFieldPotentialAtParticles() {
    const int N = numberParticles();
    detail::DeviceArray<real4> fp(N);
    detail::check(uammd_fill_zero(fp.d, sizeof(real4) * (size_t)N, nullptr));
    {
      auto pos = pd->getPos(access::gpu, access::read);
      auto charge = pd->getCharge(access::gpu, access::read);
      auto force = pd->getForce(access::gpu, access::readwrite);
      auto energy = pd->getEnergy(access::gpu, access::readwrite);
      ParticleGroup *g = subgroup.get();
      real4 *f = const_cast<real4 *>(detail::groupRows((const real4 *)force.raw(), g, forceRows, nullptr));
      real *e = const_cast<real *>(detail::groupRows((const real *)energy.raw(), g, energyRows, nullptr));
      const real4 *posNow = detail::groupRows((const real4 *)pos.raw(), g, posRows, nullptr);
      const real *chargeNow = detail::groupRows((const real *)charge.raw(), g, chargeRows, nullptr);
#if defined(DOUBLE_PRECISION)
      detail::check(uammd_poisson_field_potential_f64(h, (const double *)posNow, chargeNow, N, (double *)fp.d, (double *)f, e, nullptr));
#else
      detail::check(uammd_poisson_field_potential(h, (const float *)posNow, chargeNow, N, (float *)fp.d, (float *)f, e, nullptr));
#endif
      detail::scatterRows((const real4 *)f, force.raw(), g, nullptr);
      detail::scatterRows((const real *)e, energy.raw(), g, nullptr);
    }
    std::vector<real4> out(N);
    detail::hipCheck(hipMemcpy(out.data(), fp.d, sizeof(real4) * (size_t)N, hipMemcpyDeviceToHost), "hipMemcpy");
    return out;
  }
  int3 getCells() const { return make_int3(info.cells[0], info.cells[1], info.cells[2]); }
  int getSupport() const { return info.support; }
  real getNearFieldCutOff() const { return info.nearFieldCutOff; }
};

namespace lanczos {
struct MatrixDot {   // misc/LanczosAlgorithm/MatrixDot.h:7-25
  void setSize(int newsize) { m_size = newsize; }
  virtual void operator()(real *v, real *Mv) = 0;
  virtual ~MatrixDot() = default;
protected:
  int m_size = 0;
};
// any callable (v, Mv) as a MatrixDot (MatrixDot.h:14-23)
template <class Foo> struct MatrixDotAdaptor : public MatrixDot {
  Foo foo;
  explicit MatrixDotAdaptor(Foo f) : foo(std::move(f)) {}
  void operator()(real *v, real *Mv) override { foo(v, Mv); }
};
template <class Foo> MatrixDotAdaptor<typename std::decay<Foo>::type> createMatrixDotAdaptor(Foo &&foo) {
  return MatrixDotAdaptor<typename std::decay<Foo>::type>(std::forward<Foo>(foo));
}
class Solver {   // misc/LanczosAlgorithm.cuh:32-83
#if defined(DOUBLE_PRECISION)
  uammd_lanczos_f64 *h = nullptr;
#else
  uammd_lanczos *h = nullptr;
#endif
  static int trampoline(void *ctx, const real *v, real *Mv, int n, void *) {
    try {
      auto *dot = static_cast<MatrixDot *>(ctx);
      dot->setSize(n);
      (*dot)(const_cast<real *>(v), Mv);
      return 0;
    } catch (...) { return -99; }
  }
public:
#if defined(DOUBLE_PRECISION)
  Solver() { detail::check(uammd_lanczos_create_f64(&h)); }
  ~Solver() { uammd_lanczos_destroy_f64(h); }
#else
  Solver() { detail::check(uammd_lanczos_create(&h)); }
  ~Solver() { uammd_lanczos_destroy(h); }
#endif
  Solver(const Solver &) = delete;
  // Bv = sqrt(M) v to the tolerance; returns the number of iterations it took
  int run(MatrixDot *dot, real *Bv, const real *v, real tolerance, int N, hipStream_t st = 0) {
    int it = 0;
#if defined(DOUBLE_PRECISION)
    const int rc = uammd_lanczos_run_f64(h, &Solver::trampoline, dot, Bv, v, tolerance, N, (void *)st, &it);
#else
    const int rc = uammd_lanczos_run(h, &Solver::trampoline, dot, Bv, v, tolerance, N, (void *)st, &it);
#endif
    if (rc != 0) throw std::runtime_error(uammd_hip_last_error());  // "[Lanczos] Could not converge", LanczosAlgorithm.cu:227
    return it;
  }
  int run(MatrixDot &dot, real *Bv, const real *v, real tolerance, int N, hipStream_t st = 0) { return run(&dot, Bv, v, tolerance, N, st); }
  int run(std::function<void(real *, real *)> dot, real *Bv, const real *v, real tolerance, int N, hipStream_t st = 0) {   // :46-50
    auto adaptor = createMatrixDotAdaptor(std::move(dot));
    return run(&adaptor, Bv, v, tolerance, N, st);
  }
  // exactly numberIterations steps, no convergence test; returns the residual between the last two estimates (:54-67)
  real runIterations(MatrixDot *dot, real *Bz, const real *z, int numberIterations, int N) {
    real residual = 0;
#if defined(DOUBLE_PRECISION)
    const int rc = uammd_lanczos_run_iterations_f64(h, &Solver::trampoline, dot, Bz, z, numberIterations, N, nullptr, &residual);
#else
    const int rc = uammd_lanczos_run_iterations(h, &Solver::trampoline, dot, Bz, z, numberIterations, N, nullptr, &residual);
#endif
    if (rc != 0) throw std::runtime_error(uammd_hip_last_error());
    return residual;
  }
  real runIterations(MatrixDot &dot, real *Bv, const real *v, int numberIterations, int N) { return runIterations(&dot, Bv, v, numberIterations, N); }
  real runIterations(std::function<void(real *, real *)> dot, real *Bv, const real *v, int numberIterations, int N) {
    auto adaptor = createMatrixDotAdaptor(std::move(dot));
    return runIterations(&adaptor, Bv, v, numberIterations, N);
  }
#if defined(DOUBLE_PRECISION)
  void setIterationHardLimit(int newLimit) { detail::check(uammd_lanczos_set_iteration_hard_limit_f64(h, newLimit)); }
  int getLastRunRequiredSteps() { int s = 0; detail::check(uammd_lanczos_get_last_run_required_steps_f64(h, &s)); return s; }
#else
  void setIterationHardLimit(int newLimit) { detail::check(uammd_lanczos_set_iteration_hard_limit(h, newLimit)); }
  int getLastRunRequiredSteps() { int s = 0; detail::check(uammd_lanczos_get_last_run_required_steps(h, &s)); return s; }
#endif
};
}  // namespace lanczos

// ---- utils/InitialConditions.cuh: simple cubic stand-in for initLattice(L, N, sc) -------------------------------------------------------------
inline std::vector<real4> initLatticeSC(real3 L, uint N) {
  const int m = (int)std::ceil(std::cbrt((double)N));
  std::vector<real4> pos(N);
  for (uint i = 0; i < N; ++i) {
    const int ix = i % m, iy = (i / m) % m, iz = i / (m * m);
    pos[i] = {(ix + real(0.5)) / m * L.x - L.x / 2, (iy + real(0.5)) / m * L.y - L.y / 2, (iz + real(0.5)) / m * L.z - L.z / 2, 0};
  }
  return pos;
}

#if !defined(DOUBLE_PRECISION)
// ---- multi-GPU (new: the reference is single GPU) -----------------------------------------------------------------------------------------
// One process per GPU.  Rank 0 makes the 128-byte id (Comm::uniqueId) and hands it to the other processes by whatever the program
// uses to start them (MPI_Bcast, a file, a socket); every process then constructs its Comm after selecting its device.  The ranks form
// a periodic ring along z.  Thin RAII over uammd_comm_* (include/uammd_hip.h, RCCL over xGMI inside).
class Comm {
  uammd_comm *h = nullptr;
public:
  static std::vector<char> uniqueId() {
    std::vector<char> id(128);
    detail::check(uammd_comm_unique_id(id.data()));
    return id;
  }
  Comm(int rank, int world, const std::vector<char> &id) {
    if (id.size() != 128) throw std::runtime_error("Comm: the unique id has 128 bytes");
    detail::check(uammd_comm_init(&h, rank, world, id.data()));
  }
  Comm(const Comm &) = delete;
  ~Comm() { uammd_comm_destroy(h); }
  int rank() const { return uammd_comm_rank(h); }
  int world() const { return uammd_comm_world(h); }
  uammd_comm *handle() { return h; }
  // rows of `floatsPerRow` floats to rank + 1 / rank - 1, rows from rank - 1 / rank + 1 (asynchronous on st)
  void haloExchange(const real *sendUp, int nUp, const real *sendDown, int nDown, real *recvFromDown, int nFromDown, real *recvFromUp,
                    int nFromUp, int floatsPerRow, hipStream_t st = 0) {
    detail::check(uammd_comm_halo_exchange(h, sendUp, nUp, sendDown, nDown, recvFromDown, nFromDown, recvFromUp, nFromUp, floatsPerRow, (void *)st));
  }
  // the two message sizes of a refresh (synchronises st)
  void exchangeCounts(int toUp, int toDown, int &fromDown, int &fromUp, hipStream_t st = 0) {
    const int to[2] = {toUp, toDown};
    int from[2] = {0, 0};
    detail::check(uammd_comm_exchange_counts(h, to, from, (void *)st));
    fromDown = from[0];
    fromUp = from[1];
  }
  // the same with the two sizes still in device memory (uammd_slab_select's counts): one synchronisation returns all four
  void exchangeCountsDevice(const int *d_toUpDown, int &toUp, int &toDown, int &fromDown, int &fromUp, hipStream_t st = 0) {
    int all4[4] = {0, 0, 0, 0};
    detail::check(uammd_comm_exchange_counts_device(h, d_toUpDown, all4, (void *)st));
    toUp = all4[0]; toDown = all4[1]; fromDown = all4[2]; fromUp = all4[3];
  }
  void allToAll(const void *send, void *recv, size_t bytesPerPeer, hipStream_t st = 0) { detail::check(uammd_comm_alltoall(h, send, recv, bytesPerPeer, (void *)st)); }
  void allReduceSum(real *buf, int n, hipStream_t st = 0) { detail::check(uammd_comm_allreduce_sum(h, buf, n, (void *)st)); }
};
#endif   // !DOUBLE_PRECISION

}  // namespace uammd

// ---- utils/debugTools.h:14-15: UAMMD's own checking macros, used by programs written against it --------------------------------------
// CudaSafeCall(err) throws uammd::cuda_generic_error for a runtime call that did not return success; CudaCheckError() does the same for
// the last error the runtime recorded (after a device synchronisation when UAMMD_DEBUG is defined)
namespace uammd {
namespace detail {
inline void safeCall(hipError_t err, const char *file, int line) {
  if (err != hipSuccess) {
    (void)hipGetLastError();
    throw cuda_generic_error("CudaSafeCall() failed at " + std::string(file) + ":" + std::to_string(line) + ": " + hipGetErrorString(err) +
                                 " - code: " + std::to_string((int)err), (int)err);
  }
}
inline void checkError(const char *file, int line) {
#ifdef UAMMD_DEBUG
  safeCall(hipDeviceSynchronize(), file, line);
#endif
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess)
    throw cuda_generic_error("CudaCheckError() failed at " + std::string(file) + ":" + std::to_string(line) + ": " + hipGetErrorString(err) +
                                 " - code: " + std::to_string((int)err), (int)err);
}
}  // namespace detail
}  // namespace uammd
#define CudaSafeCall(err) ::uammd::detail::safeCall(err, __FILE__, __LINE__)
#define CudaCheckError() ::uammd::detail::checkError(__FILE__, __LINE__)
#endif
