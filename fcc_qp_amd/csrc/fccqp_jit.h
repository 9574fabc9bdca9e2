// fccqp_jit.h -- run-time shape specialisation (internal header; implementation in fccqp_jit.cpp).
//
// The reference's constructor takes the problem sizes at run time (FCCQP::FCCQP(num_vars, num_equality_constraints, nc,
// lambda_c_start), /root/reference/src/fcc_qp.cpp:24-55); the specialised kernels of this library are templates on
// those sizes.  For a shape that is not in fused_shapes.def / coop_shapes.def the template is instantiated IN PROCESS
// with hiprtc (the kernel headers travel inside the library as text), the code object is kept on disk, and the generic
// any-shape kernel remains the fallback when compilation is not possible.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "fccqp_device.h"

namespace fccqp {
namespace jit {

enum Kind { kFused = 0, kCoop = 1 };

struct Kernel {
  hipFunction_t fn = nullptr;
  int kind = kFused;
  int waves = 1;                 // wavefronts per workgroup (block = 64 * waves)
  int scratch_bytes = 0;         // per-lane scratch of the compiled kernel (0: none)
};

// process-wide switch (default: on; FCCQP_JIT=0 in the environment or fccqp_set_jit(0) turn it off)
bool enabled();
void set_enabled(bool on);

// Does (n, m, nc, s) qualify for a run-time compiled kernel of this kind at all (the static_asserts of
// fused::Shape / coop::Shape that can be checked without compiling)?  `waves` receives the workgroup size chosen.
bool qualifies(int kind, int n, int m, int nc, int s, int *waves);

// Compiles the kernel for the shape (or finds its code object in the in-process / on-disk cache).  No GPU needed.
// Returns true on success; `msg` says why not otherwise, and `from_cache` whether hiprtc had to run.
bool prepare(int kind, int n, int m, int nc, int s, bool f32, std::string *msg, bool *from_cache = nullptr);

// prepare() + load on the current device.  Returns false (and a message) when the kernel cannot be had.
bool acquire(int kind, int n, int m, int nc, int s, bool f32, Kernel *out, std::string *msg);

hipError_t launch(const Kernel &k, const KernelArgs &a, hipStream_t stream);

// forget everything held in memory (code objects and loaded modules stay loaded but are no longer looked up): the next
// acquire() goes to the disk cache.  Tests only.
void forget();

// directory of the on-disk cache (created on first use)
std::string cache_dir();

}  // namespace jit
}  // namespace fccqp
