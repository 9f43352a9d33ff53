// sLn over an array (dspc.h:155-160: x > 0 ? logf(x) : 0), sixteen values at a time on AVX-512 -- with THIS HOST'S libm as
// the definition of logf: the reference calls libm's scalar logf per value, and its dumps are what the features must equal
// bit for bit.  glibc's logf (2.28+, sysdeps/ieee754/flt-32/e_logf.c) is a fixed sequence of IEEE double operations -- a
// 16-entry table (1/c, log c), a cubic in double, one rounding to float at the end -- so the same sequence on eight doubles
// per register gives the same bits.  Two things are not taken on trust:
//   * which sequence: glibc selects at load time between a build with fused multiply-adds and one without; both are here,
//   * that it IS this libm's: the first call checks the chosen form against logf() on 300 000 values (every exponent, both
//     ends of every table interval, a pseudo-random sweep); a libm that answers differently anywhere -- another glibc,
//     another libc -- switches the vector form off for the process and every value goes through logf() as before.
// tests/test_cli_cpu.py runs the check over ALL positive floats (phnrec --selftest-ln).
#include "veclog.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <initializer_list>

namespace phnrec {

namespace {

// glibc's __logf_data (LOGF_TABLE_BITS = 4, LOGF_POLY_ORDER = 4); read out of libm.so.6 of glibc 2.35 and equal to the
// published table of ARM's optimized-routines, which glibc took it from
const double kInvC[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                          0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                          0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                          0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
const double kLogC[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                          -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                          -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                          0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
const double kLn2 = 0x1.62e42fefa39efp-1;
const double kA[3] = {-0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2};
const uint32_t kOff = 0x3f330000u;

inline uint32_t AsUint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// the vector form takes positive normal numbers other than 1.0f (what glibc's main path takes); everything else -- zero and
// negative values (sLn: 0), subnormals, infinities, NaN, exactly 1 -- goes through the scalar expression
inline bool MainPath(uint32_t ix) { return ix - 0x00800000u < 0x7f800000u - 0x00800000u && ix != 0x3f800000u; }

inline float ScalarLn(float x) { return x > 0.0f ? logf(x) : 0.0f; }

// glibc's sequence on one value (e_logf.c), with fused multiply-adds or without; this file is compiled with
// -ffp-contract=off, so the second form stays unfused
__attribute__((always_inline)) inline float LnScalarImpl(float x, bool fma_form)
{
    if (!(x > 0.0f)) return 0.0f;
    uint32_t ix = AsUint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix == 0x7f800000u) return x;
    if (ix < 0x00800000u) ix = AsUint(x * 0x1p23f) - (23u << 23);
    const uint32_t tmp = ix - kOff;
    const int i = (int)((tmp >> 19) & 15u), k = (int)(int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf; memcpy(&zf, &iz, 4);
    const double z = (double)zf, invc = kInvC[i], logc = kLogC[i];
    double r, y0, y;
    if (fma_form) {
        r = __builtin_fma(z, invc, -1.0);
        y0 = __builtin_fma((double)k, kLn2, logc);
        const double r2 = r * r;
        y = __builtin_fma(kA[1], r, kA[2]);
        y = __builtin_fma(kA[0], r2, y);
        y = __builtin_fma(y, r2, y0 + r);
    } else {
        r = z * invc - 1.0;
        y0 = logc + (double)k * kLn2;
        const double r2 = r * r;
        y = kA[1] * r + kA[2];
        y = kA[0] * r2 + y;
        y = y * r2 + (y0 + r);
    }
    return (float)y;
}
#if defined(__x86_64__)
__attribute__((target("fma"))) float LnFmaHw(float x) { return LnScalarImpl(x, true); }      // vfmadd
#endif
float LnFmaSoft(float x) { return LnScalarImpl(x, true); }                                   // libm's fma()
float LnPlain(float x) { return LnScalarImpl(x, false); }
float LnFma(float x)
{
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("fma");
    if (hw) return LnFmaHw(x);
#endif
    return LnFmaSoft(x);
}

// the probe: for every exponent of the normal range, both ends and the middle of each of the 16 table intervals (the
// interval of a value is bits 19..22 of ix - OFF), values next to 1, special values, and a 32-bit LCG over all bit patterns
size_t FillProbe(float *probe, size_t cap)
{
    size_t n = 0;
    for (int k = -127; k <= 128; k++)
        for (uint32_t iv = 0; iv < 16; iv++)
            for (uint32_t m : {0u, 1u, 0x3ffffu, 0x40000u, 0x7fffeu, 0x7ffffu}) {
                const uint32_t ix = kOff + (((uint32_t)k << 23) | (iv << 19) | m);      // ix - OFF = k : interval : m
                float f; memcpy(&f, &ix, 4);
                probe[n++] = f;
            }
    for (int d = -64; d <= 64; d++) { const uint32_t ix = 0x3f800000u + (uint32_t)d; float f; memcpy(&f, &ix, 4); probe[n++] = f; }
    const float special[] = {0.0f, -0.0f, -1.0f, 1e-45f, 1e-39f, 1.17549435e-38f, 3.4028235e38f, INFINITY, -INFINITY, NAN};
    for (float f : special) probe[n++] = f;
    uint32_t s = 0x9e3779b9u;
    while (n < cap) { s = s * 1664525u + 1013904223u; float f; memcpy(&f, &s, 4); probe[n++] = f; }
    return n;
}

int DecideLibmForm()
{
    static float probe[300000];
    const size_t n = FillProbe(probe, sizeof probe / sizeof probe[0]);
    bool fma_ok = true, plain_ok = true;
    for (size_t i = 0; i < n && (fma_ok || plain_ok); i++) {
        const float want = ScalarLn(probe[i]);
        if (fma_ok) { const float a = LnFma(probe[i]); if (memcmp(&a, &want, 4) != 0) fma_ok = false; }
        if (plain_ok) { const float b = LnPlain(probe[i]); if (memcmp(&b, &want, 4) != 0) plain_ok = false; }
    }
    return fma_ok ? 1 : plain_ok ? 2 : 0;
}

#if defined(__x86_64__)
template <bool FMA>
__attribute__((target("avx512f,avx512dq,avx512vl,avx512bw,fma"))) inline __m256 Ln8(__m256i ix, __m512d invc_lo, __m512d invc_hi,
                                                                                  __m512d logc_lo, __m512d logc_hi)
{
    const __m256i tmp = _mm256_sub_epi32(ix, _mm256_set1_epi32((int)kOff));
    const __m256i i = _mm256_and_si256(_mm256_srli_epi32(tmp, 19), _mm256_set1_epi32(15));
    const __m256i k = _mm256_srai_epi32(tmp, 23);
    const __m256i iz = _mm256_sub_epi32(ix, _mm256_and_si256(tmp, _mm256_set1_epi32((int)0xff800000u)));
    const __m512i idx = _mm512_cvtepi32_epi64(i);
    const __m512d invc = _mm512_permutex2var_pd(invc_lo, idx, invc_hi);      // index bit 3 selects the second register
    const __m512d logc = _mm512_permutex2var_pd(logc_lo, idx, logc_hi);
    const __m512d z = _mm512_cvtps_pd(_mm256_castsi256_ps(iz));
    const __m512d kd = _mm512_cvtepi32_pd(k);
    const __m512d ln2 = _mm512_set1_pd(kLn2), a0 = _mm512_set1_pd(kA[0]), a1 = _mm512_set1_pd(kA[1]), a2 = _mm512_set1_pd(kA[2]);
    __m512d r, y0, y;
    if (FMA) {
        r = _mm512_fmsub_pd(z, invc, _mm512_set1_pd(1.0));                  // r = z * invc - 1
        y0 = _mm512_fmadd_pd(kd, ln2, logc);                                // y0 = logc + k * Ln2
        const __m512d r2 = _mm512_mul_pd(r, r);
        y = _mm512_fmadd_pd(a1, r, a2);                                     // y = A[1] * r + A[2]
        y = _mm512_fmadd_pd(a0, r2, y);                                     // y = A[0] * r2 + y
        y = _mm512_fmadd_pd(y, r2, _mm512_add_pd(y0, r));                   // y = y * r2 + (y0 + r)
    } else {
        r = _mm512_sub_pd(_mm512_mul_pd(z, invc), _mm512_set1_pd(1.0));
        y0 = _mm512_add_pd(logc, _mm512_mul_pd(kd, ln2));
        const __m512d r2 = _mm512_mul_pd(r, r);
        y = _mm512_add_pd(_mm512_mul_pd(a1, r), a2);
        y = _mm512_add_pd(_mm512_mul_pd(a0, r2), y);
        y = _mm512_add_pd(_mm512_mul_pd(y, r2), _mm512_add_pd(y0, r));
    }
    return _mm512_cvtpd_ps(y);
}

template <bool FMA>
__attribute__((target("avx512f,avx512dq,avx512vl,avx512bw,fma"))) void LnArray512(float *x, size_t n)
{
    const __m512d invc_lo = _mm512_loadu_pd(kInvC), invc_hi = _mm512_loadu_pd(kInvC + 8);
    const __m512d logc_lo = _mm512_loadu_pd(kLogC), logc_hi = _mm512_loadu_pd(kLogC + 8);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i ix = _mm512_loadu_si512(x + i);
        // lanes on glibc's main path: ix - 0x00800000 < 0x7f000000 (unsigned) and ix != 1.0f
        const __mmask16 in_range = _mm512_cmplt_epu32_mask(_mm512_sub_epi32(ix, _mm512_set1_epi32(0x00800000)), _mm512_set1_epi32(0x7f000000));
        const __mmask16 main = in_range & _mm512_cmpneq_epu32_mask(ix, _mm512_set1_epi32(0x3f800000));
        const __m256 lo = Ln8<FMA>(_mm512_castsi512_si256(ix), invc_lo, invc_hi, logc_lo, logc_hi);
        const __m256 hi = Ln8<FMA>(_mm512_extracti64x4_epi64(ix, 1), invc_lo, invc_hi, logc_lo, logc_hi);
        const __m512 y = _mm512_insertf32x8(_mm512_castps256_ps512(lo), hi, 1);
        if (main == 0xffff) {
            _mm512_storeu_ps(x + i, y);
        } else {
            float in[16];
            _mm512_storeu_ps(in, _mm512_castsi512_ps(ix));
            _mm512_mask_storeu_ps(x + i, main, y);
            for (int l = 0; l < 16; l++)
                if (!((main >> l) & 1)) x[i + l] = ScalarLn(in[l]);
        }
    }
    for (; i < n; i++) x[i] = ScalarLn(x[i]);
}

bool CpuHasAvx512()
{
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("fma");
}
#endif

// forms: 1 = scalar libm only, 2 = vector form with fused multiply-adds, 3 = vector form without

bool SameAsLibm(void (*fn)(float *, size_t), const float *probe, size_t n)
{
    float buf[4096];
    for (size_t i = 0; i < n; i += 4096) {
        const size_t m = n - i < 4096 ? n - i : 4096;
        memcpy(buf, probe + i, m * sizeof(float));
        fn(buf, m);
        for (size_t k = 0; k < m; k++) {
            const float want = ScalarLn(probe[i + k]);
            if (memcmp(&want, &buf[k], 4) != 0) return false;
        }
    }
    return true;
}

int Decide()
{
#if defined(__x86_64__)
    if (getenv("PHNREC_NO_VECTOR_LN") || getenv("PHNREC_NO_AVX512") || !CpuHasAvx512()) return 1;
    static float probe[300000];
    const size_t n = FillProbe(probe, sizeof probe / sizeof probe[0]);
    if (SameAsLibm(LnArray512<true>, probe, n)) return 2;
    if (SameAsLibm(LnArray512<false>, probe, n)) return 3;
#endif
    return 1;
}

int Form()
{
    static const int form = Decide();              // (initialised once, by the first thread that gets here; the others wait)
    return form;
}

}  // namespace

int LibmLogfForm()
{
    // PHNREC_LN_FORM=0/1/2 overrides the detection (tests: 0 is what a host with another libc reports)
    static const int form = getenv("PHNREC_LN_FORM") ? atoi(getenv("PHNREC_LN_FORM")) : DecideLibmForm();
    return form < 0 || form > 2 ? 0 : form;
}

float LnRestated(float x, int form) { return form == 1 ? LnFma(x) : LnPlain(x); }

void LnInPlace(float *x, size_t n)
{
#if defined(__x86_64__)
    const int f = Form();
    if (f == 2) { LnArray512<true>(x, n); return; }
    if (f == 3) { LnArray512<false>(x, n); return; }
#endif
    for (size_t i = 0; i < n; i++) x[i] = ScalarLn(x[i]);
}

const char *LnForm()
{
    switch (Form()) {
    case 2: return "avx512 (fused multiply-adds, as this libm's logf)";
    case 3: return "avx512 (separate multiplies and adds, as this libm's logf)";
    default: return "libm logf per value";
    }
}

// every positive float (and the rest of the bit patterns in strides) against libm: 0 = identical
long long LnSelfTest(int threads_hint)
{
    (void)threads_hint;
    long long bad = 0;
    static const size_t kChunk = 1 << 16;
    float *buf = static_cast<float *>(malloc(kChunk * sizeof(float)));
    float *ref = static_cast<float *>(malloc(kChunk * sizeof(float)));
    // positive half exhaustively: 0x00000000 .. 0x7fffffff; negative half: every 4099th pattern
    for (uint64_t base = 0; base < 0x80000000ull; base += kChunk) {
        for (size_t k = 0; k < kChunk; k++) { const uint32_t u = (uint32_t)(base + k); memcpy(&ref[k], &u, 4); }
        memcpy(buf, ref, kChunk * sizeof(float));
        LnInPlace(buf, kChunk);
        for (size_t k = 0; k < kChunk; k++) {
            const float want = ScalarLn(ref[k]);
            if (memcmp(&want, &buf[k], 4) != 0) bad++;
        }
    }
    size_t k = 0;
    for (uint64_t u = 0x80000000ull; u <= 0xffffffffull; u += 4099) { const uint32_t v = (uint32_t)u; memcpy(&ref[k], &v, 4); if (++k == kChunk) break; }
    memcpy(buf, ref, k * sizeof(float));
    LnInPlace(buf, k);
    for (size_t j = 0; j < k; j++) { const float want = ScalarLn(ref[j]); if (memcmp(&want, &buf[j], 4) != 0) bad++; }
    free(buf); free(ref);
    return bad;
}

}  // namespace phnrec
