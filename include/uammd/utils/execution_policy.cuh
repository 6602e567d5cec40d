// utils/execution_policy.cuh (reference: src/utils/execution_policy.cuh:17-24): thrust's device policy over the pool of temporary
// device memory, `thrust::sort(uammd::cached_device_execution_policy.on(st), ...)`.  rocThrust: hipcc only.
#pragma once
#include "../uammd.h"
#if defined(__HIPCC__)
#include <thrust/execution_policy.h>
namespace uammd {
namespace detail {
static const auto cached_device_execution_policy = thrust::device(System::allocator_thrust<char>());
}
using detail::cached_device_execution_policy;
}  // namespace uammd
#endif
