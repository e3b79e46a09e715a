// Which RS_* environment variables a build of the library reads.
//   * Product switches, documented in INTEGRATION.md section 4, and the kernel-selection switches the test-suite's cross-checks flip
//     (one search / GEMM / iVector kernel against another on the same input) are read with std::getenv in every build.
//   * Measurement switches -- ablations, traces, tile-shape and schedule sweeps behind the numbers in profiles/ and DESIGN.md -- go
//     through TuneEnv(): they exist only in a build with -DRS_TUNING (make EXTRA=-DRS_TUNING; profiles/micro/*.sh build one into a
//     scratch copy) and read as "unset" in the shipped library.
#pragma once
#include <cstdlib>

namespace rs {
#ifdef RS_TUNING
inline const char *TuneEnv(const char *name) { return std::getenv(name); }
#else
inline const char *TuneEnv(const char *) { return nullptr; }
#endif
}  // namespace rs
