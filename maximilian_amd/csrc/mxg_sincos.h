// mxg_sincos.h -- sin/cos of (phase * TWOPI) for maxiOsc::sinewave / coswave (C:230, C:278).
//
// The reference evaluates glibc sin()/cos() on the ROUNDED product x = phase*TWOPI.  glibc's
// result is within 1 ULP of the true value (in practice correctly rounded almost always).
// The device path must therefore produce sin(x) to well under 1 ULP so the two differ by at
// most 1 ULP per sample (the contract stated in DESIGN.md; measured in tests/test_osc_parity).
#pragma once
#include "mxg_common.h"

namespace mxg {

__device__ __forceinline__ double sin_2pi_phase(double phase) {
    double x = phase * (MXG_TWOPI);
    return sin(x);
}
__device__ __forceinline__ double cos_2pi_phase(double phase) {
    double x = phase * (MXG_TWOPI);
    return cos(x);
}

}  // namespace mxg
