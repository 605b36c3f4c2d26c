// A/B and ablation switches (include/fo1_ab.h) exist only in the test / bench build (-DFO1_ENABLE_AB, libfo1hip_ab.so).  In the product
// library (libfo1hip.so) they are compile-time constants: no process-global mutable state behind the public ABI, and the measured-slower
// kernel forms they select are not compiled in.
#pragma once
#ifdef FO1_ENABLE_AB
#include "../../include/fo1_ab.h"   // declarations carry the export visibility (the build is -fvisibility=hidden)
#define FO1_AB_VAR static int
#define FO1_AB_EXTERN_VAR int
#else
#define FO1_AB_VAR [[maybe_unused]] static constexpr int
#define FO1_AB_EXTERN_VAR [[maybe_unused]] static constexpr int
#endif
