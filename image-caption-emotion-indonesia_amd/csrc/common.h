// Shared host/device helpers for libcapnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
