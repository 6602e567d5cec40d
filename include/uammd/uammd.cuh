// Forwarding header: same include path as the reference's src/uammd.cuh.
// The whole host interface of the MI355X build lives in uammd.h (C++14, no device code).
#pragma once
#include "uammd.h"
#include "third_party/saruprng.cuh"   // (the reference's umbrella header brings Saru into user code)
// A translation unit compiled by hipcc also gets what the reference's umbrella header brings into scope for user code: the thrust
// algorithms and iterators its tutorials call without including them (examples/basic_concepts/8-, 11-, 12-: thrust::fill,
// thrust::reduce, thrust::make_permutation_iterator).  rocThrust needs hipcc; a plain g++ TU gets the host interface alone.
// (User code spells two CUDA names that have no HIP spelling of their own: cudaStream_t -> hipStream_t, thrust::cuda::par ->
// thrust::hip::par — one-token edits, INTEGRATION.md.)
#if defined(__HIPCC__)
#include <thrust/device_vector.h>
#include <thrust/execution_policy.h>
#include <thrust/fill.h>
#include <thrust/host_vector.h>
#include <thrust/iterator/constant_iterator.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/permutation_iterator.h>
#include <thrust/iterator/transform_iterator.h>
#include <thrust/reduce.h>
#include <thrust/transform.h>
#include <thrust/transform_reduce.h>
// the reference's umbrella header also brings CUB (third_party/uammd_cub.cuh); its name on this platform is hipcub:: — programs that say
// cub::ThreadLoad / cub::DeviceScan spell it hipcub:: (a user-side edit like cudaStream_t -> hipStream_t; not aliased here)
#include <hipcub/hipcub.hpp>
#endif
