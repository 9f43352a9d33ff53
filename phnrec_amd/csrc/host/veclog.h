// sLn over an array with this host's libm as the definition of logf (veclog.cpp)
#pragma once
#include <cstddef>

namespace phnrec {

// x[i] = x[i] > 0 ? logf(x[i]) : 0 for i < n -- the bits libm's scalar logf gives, sixteen values at a time where the CPU
// has AVX-512 and the form has been checked against this process's logf (first call); PHNREC_NO_VECTOR_LN=1: libm only
void LnInPlace(float *x, size_t n);
// which form LnInPlace uses in this process (text for -v / the self-test)
const char *LnForm();
// Which of glibc's two logf sequences this process's libm matches, found out by comparing scalar restatements of both with
// logf() on the probe values (every exponent, both ends of every table interval, a pseudo-random sweep; once per process):
// 1 = with fused multiply-adds, 2 = without (the values of LCRC_LN_GLIBC_FMA / LCRC_LN_GLIBC, include/lcrc.h), 0 = neither
// (another libc: the GPU front-end then cannot promise the host front-end's bits).
int LibmLogfForm();
// the scalar restatement itself (form 1 or 2), for self-tests
float LnRestated(float x, int form);
// LnInPlace against logf over every non-negative bit pattern and a stride of the negative ones: number of differing values
long long LnSelfTest(int threads_hint);

}  // namespace phnrec
