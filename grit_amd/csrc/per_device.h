// Idempotent per-DEVICE setup of the launchers (hipFuncSetAttribute and the CU count belong to a device; a process may drive several:
// ADVICE r04).  `static PerDevice<bool> x_pd; bool& x = x_pd();` keeps the launcher code as it was with a process-wide static.
#pragma once
#include <hip/hip_runtime.h>

namespace grit_detail {
template <class T>
struct PerDevice {
    T slot[64] = {};
    T& operator()() {
        int d = 0;
        (void)hipGetDevice(&d);
        return slot[d & 63];
    }
};
}  // namespace grit_detail
