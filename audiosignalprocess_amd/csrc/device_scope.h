// device_scope.h -- every entry point of the C-ABI selects its batch's device for its HIP calls; the
// caller's current device is put back when the entry point returns, so that a host thread that drives
// batches on several GPUs (or mixes this library with its own HIP code) never finds its device changed.
#pragma once
#include <hip/hip_runtime.h>

struct AspDeviceScope {
  int prev = -1;
  AspDeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~AspDeviceScope() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
  AspDeviceScope(const AspDeviceScope&) = delete;
  AspDeviceScope& operator=(const AspDeviceScope&) = delete;
};
