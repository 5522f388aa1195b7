// Shared host/device helpers for libcapnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>

namespace capnet {

// ---- error reporting (C-ABI: negative status + capnet_last_error()) ----
enum Status : int {
  kOk = 0,
  kErrInvalidArg = -1,
  kErrHip = -2,
  kErrNoDevice = -3,
  kErrWorkspace = -4,
  kErrUnsupported = -5,
};

void set_error(const char* fmt, ...);
const char* last_error();

#define CAPNET_HIP_CHECK(expr)                                                   \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) {                                                      \
      capnet::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,       \
                        hipGetErrorString(_e));                                  \
      return capnet::kErrHip;                                                    \
    }                                                                            \
  } while (0)

#define CAPNET_REQUIRE(cond, ...)                                                \
  do {                                                                           \
    if (!(cond)) {                                                               \
      capnet::set_error(__VA_ARGS__);                                            \
      return capnet::kErrInvalidArg;                                             \
    }                                                                            \
  } while (0)

// Timing events attached to ONE kernel dispatch (hipExtLaunchKernelGGL: the dispatch packet's own start / end
// timestamps -- no marker packets in the queue, one host call instead of three). The caller (csrc/trunk.cpp) parks a pair
// here; the next CAPNET_LAUNCH_TIMED on this thread takes it. Without a parked pair the macro is a plain launch.
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};
LaunchEvents& launch_events();     // thread-local
#define CAPNET_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                              \
  do {                                                                                                      \
    capnet::LaunchEvents& _le = capnet::launch_events();                                                    \
    if (_le.start) {                                                                                        \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, _le.start, _le.stop, 0, __VA_ARGS__);           \
      _le.start = _le.stop = nullptr;                                                                       \
    } else {                                                                                                \
      hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                      \
    }                                                                                                       \
  } while (0)

#define CAPNET_LAUNCH_CHECK()                                                    \
  do {                                                                           \
    hipError_t _e = hipGetLastError();                                           \
    if (_e != hipSuccess) {                                                      \
      capnet::set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__,   \
                        hipGetErrorString(_e));                                  \
      return capnet::kErrHip;                                                    \
    }                                                                            \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

static inline bool aligned16(const void* p) { return (((size_t)p) & 15) == 0; }

}  // namespace capnet
