// Dense fast path for surfaces (placeholder until the tuned kernel lands): never dispatches.
#pragma once
#include "ivs_surface_generic.hpp"
namespace ivs {
inline int launch_surface_dense(const SurfaceParams&, int, hipStream_t, const char**) { return 0; }
}  // namespace ivs
