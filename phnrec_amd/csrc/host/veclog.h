// sLn over an array with this host's libm as the definition of logf (veclog.cpp)
#pragma once
#include <cstddef>

namespace phnrec {

// x[i] = x[i] > 0 ? logf(x[i]) : 0 for i < n -- the bits libm's scalar logf gives, sixteen values at a time where the CPU
// has AVX-512 and the form has been checked against this process's logf (first call); PHNREC_NO_VECTOR_LN=1: libm only
void LnInPlace(float *x, size_t n);
// which form LnInPlace uses in this process (text for -v / the self-test)
const char *LnForm();
// LnInPlace against logf over every non-negative bit pattern and a stride of the negative ones: number of differing values
long long LnSelfTest(int threads_hint);

}  // namespace phnrec
