// frontend_kernels.hip -- "next" row f1 of SURVEY.md 8: the mel-bank front-end on the GPU.
//
// raw samples -> log mel-bank energies, one wave per 10 ms frame, following the reference's
// arithmetic operation by operation; ln() is the host libm's own sequence of double operations when the caller names it
// (lcrc_frontend_set_ln: the dumps then agree with the reference's to the last bit), else log() in double rounded once
// (last bit or one ulp):
//   waveform decode   srec.cpp:709-791, alaw.cpp      lin16 / A-law, dc_shift, scale
//   framing           srec.cpp:945, melbanks.cpp:151   frame t = samples [t*step, t*step + vs)
//   pre-processing    dspc.h:64-84                     z_mean_source, pre-emphasis (both off in shipped configs)
//   Hamming           dspc.h:162-167                   table built on the host with the reference's expression
//   FFT               dspc.cpp:24-78                   radix-2 DIT on interleaved floats; twiddles from the
//                                                      reference's double-precision recurrence (table built on
//                                                      the host), butterfly products in double, rounded to float
//   power, mel        dspc.h:141-146, dspc.cpp:236-269 per bank, bins accumulated in ascending order
//   ln                dspc.h:155-160
// and the sentence mean normalisation (srec.cpp:1500-1511) with its sequential f32 column sums.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "frontend_dev.h"

namespace phnrec {

// G.711 A-law expansion (== 8 * ALawTableD5[b], alaw.cpp:14-48, srec.cpp:769)
__device__ __forceinline__ float alaw_to_linear(unsigned b)
{
    const unsigned a = b ^ 0x55u;
    int mant = (int)(a & 0x0Fu) << 4;
    const int seg = (int)(a & 0x70u) >> 4;
    if (seg == 0) mant += 8;
    else mant = (mant + 0x108) << (seg - 1);
    return (float)((a & 0x80u) ? mant : -mant);
}

// ---- ln() ------------------------------------------------------------------------------------------------------
// The reference takes ln() with libm's logf (sLn, dspc.h:155-160), so "the reference's bits" means the bits of the HOST's
// logf.  glibc's (2.28 and later; sysdeps/ieee754/flt-32/e_logf.c) is a fixed sequence of IEEE double operations -- a
// 16-entry table (1/c, ln c), a cubic in double, ONE rounding to float -- and exists in two builds that glibc selects
// between at load time: with fused multiply-adds and without.  Both sequences are restated here in the device's IEEE
// double arithmetic (the table and coefficients are glibc's __logf_data, as in host/veclog.cpp); the caller says which
// one its libm matches (lcrc_frontend_set_ln; the CLI finds out by checking 300 000 values, host/veclog.cpp) and gets that
// libm's bits.  Form 0 -- log() in double, rounded once -- stays the default of the C entry points: correctly rounded, i.e.
// glibc's result except where glibc's own 0.818-ulp error shows (about one value in 10^5).
__device__ const double kLnInvC[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                                       0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                                       0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                                       0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
__device__ const double kLnLogC[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                                       -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                                       -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                                       0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};

template <bool FMA>
__device__ __forceinline__ float ln_glibc(float x)
{
    if (!(x > 0.0f)) return 0.0f;                           // sLn's guard (zero, negative, NaN)
    unsigned ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;                     // e_logf.c: "fix sign of zero with downward rounding when x == 1"
    if (ix == 0x7f800000u) return x;                        // log(inf) == inf
    if (ix < 0x00800000u) ix = __float_as_uint(x * 0x1p23f) - (23u << 23);      // subnormal: normalised first
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u), k = (int)tmp >> 23;
    const double z = (double)__uint_as_float(ix - (tmp & 0xff800000u));
    const double invc = kLnInvC[i], logc = kLnLogC[i];
    double r, y0, y;
    if (FMA) {
        r = __builtin_fma(z, invc, -1.0);
        y0 = __builtin_fma((double)k, Ln2, logc);
        const double r2 = r * r;
        y = __builtin_fma(A1, r, A2);
        y = __builtin_fma(A0, r2, y);
        y = __builtin_fma(y, r2, y0 + r);
    } else {                                                // (compiled with -ffp-contract=off: nothing below is fused)
        r = z * invc - 1.0;
        y0 = logc + (double)k * Ln2;
        const double r2 = r * r;
        y = A1 * r + A2;
        y = A0 * r2 + y;
        y = y * r2 + (y0 + r);
    }
    return (float)y;
}

__device__ __forceinline__ float frontend_ln(float e, int form)
{
    if (form == 1) return ln_glibc<true>(e);
    if (form == 2) return ln_glibc<false>(e);
    return e > 0.0f ? (float)log((double)e) : 0.0f;
}

__global__ __launch_bounds__(256) void frontend_ln_kernel(float *x, size_t n, int form)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = frontend_ln(x[i], form);
}

hipError_t frontend_ln_launch(float *x, size_t n, int form, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>(4096, (n + 255) / 256);
    frontend_ln_kernel<<<blocks, 256, 0, stream>>>(x, n, form);
    return hipGetLastError();
}

template <int FFT>
__global__ __launch_bounds__(256) void melbank_kernel(const FrontendParams p)
{
    __shared__ __attribute__((aligned(16))) float buf[4][2 * FFT];   // interleaved (re, im) per wave
    __shared__ float pw[4][FFT / 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = blockIdx.x * 4 + wave;
    const bool live = fr < p.n_frames;
    float *d = buf[wave];

    // Eight workgroups fit a CU and each is a chain of dependent steps: the kernel's rate is (workgroups resident) /
    // (a workgroup's latency).  The search for a frame's utterance was seven or eight DEPENDENT L2 round trips in a
    // list's launch (36-150 utterances): the offsets now come into LDS with one coalesced load and the search runs there;
    // the filter weights of the tail's bin sums come along (they were a global load per bin, in a dependent loop).
    // (LDS per workgroup: FFT 256 8 + 2 + 4 + 0.5 = 14.5 KiB, eight and more per CU; FFT 512 16 + 4 + 1 + 1 = 22 KiB = seven
    //  per CU -- 20 KiB of it are the FFT's own buffers, so the offsets get 256 entries there (a list's launch carries
    //  36-150 utterances; longer offset tables are searched in global memory as before))
    constexpr int kOffLds = FFT >= 512 ? 256 : 1024;
    __shared__ int off_s[kOffLds];
    __shared__ float coef_s[FFT / 2];
    const bool off_in_lds = p.n_utts + 1 <= kOffLds;
    if (off_in_lds)
        for (int i = threadIdx.x; i <= p.n_utts; i += 256) off_s[i] = p.frame_off[i];
    for (int i = threadIdx.x; i < FFT / 2; i += 256) coef_s[i] = p.coeffs[i];
    __syncthreads();
    int u = 0, t = 0;
    long long s0 = 0, ns = 0;
    if (live) {
        int lo = 0, hi = p.n_utts;              // largest u with frame_off[u] <= fr
        if (off_in_lds) {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off_s[mid] <= fr) lo = mid; else hi = mid;
            }
            t = fr - off_s[lo];
        } else {
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (p.frame_off[mid] <= fr) lo = mid; else hi = mid;
            }
            t = fr - p.frame_off[lo];
        }
        u = lo;
        s0 = p.sample_start[u];
        ns = p.sample_start[p.n_utts + u];      // second half of the array: sample counts
    }
    // ---- decode + pre-process one frame into registers (vs <= 512: up to 8 samples per lane) ----
    const int vs = p.vector_size;
    float x[FFT / 64];
#pragma unroll
    for (int k = 0; k < FFT / 64; k++) {
        const int i = lane + 64 * k;
        float v = 0.0f;
        const long long si = (long long)t * p.vector_step + i;
        if (live && i < vs && si < ns) {        // samples past the end of a short signal are zeros
            if (p.wave_format == 1) {
                const short *w = reinterpret_cast<const short *>(p.bytes) + s0 + si;
                v = (float)*w;
            } else {
                v = alaw_to_linear(p.bytes[s0 + si]);
            }
            if (p.dc_shift != 0.0f) v = v + p.dc_shift;
            if (p.scale != 1.0f) v = v * p.scale;
        }
        x[k] = v;
    }
    if (p.z_mean_source || p.preem_coef != 0.0f) {
        // rarely used; done through LDS by lane 0 in the reference's sequential order
#pragma unroll
        for (int k = 0; k < FFT / 64; k++) d[lane + 64 * k] = x[k];
        __syncthreads();
        if (lane == 0 && live) {
            if (p.z_mean_source) {              // sSubtractAverage dspc.h:64-75
                float avg = 0.0f;
                for (int i = 0; i < vs; i++) avg += d[i];
                avg /= (float)vs;
                for (int i = 0; i < vs; i++) d[i] -= avg;
            }
            if (p.preem_coef != 0.0f) {         // sPreemphasisBW dspc.h:77-84
                for (int n = vs - 1; n > 0; --n) d[n] -= p.preem_coef * d[n - 1];
                d[0] *= (1.0f - p.preem_coef);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FFT / 64; k++) x[k] = d[lane + 64 * k];
        __syncthreads();
    }
    // ---- window; the first log2(FFT / 64) Danielson-Lanczos stages in registers ----
    // A lane's samples i = lane + 64 k land on the R = FFT / 64 CONSECUTIVE bit-reversed positions brev6(lane) * R +
    // brev(k): the butterflies of the stages with span < R pair values of one lane, and their twiddle index
    // (position & (span - 1)) is the same in every lane: wave-uniform constants.  Two (FFT 256) or three (512) of the
    // eight / nine stages need neither LDS nor a wait; twiddle (1, 0) -- the first butterfly of every group -- passes
    // its operand through as the host front-end does (host/frontend.cpp Fft8: exact up to the sign of a zero).
    constexpr int R = FFT / 64, LR = R == 4 ? 2 : 3;
    float re[R], im[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        constexpr int kRev4[4] = {0, 2, 1, 3}, kRev8[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        const int k = R == 4 ? kRev4[q & 3] : kRev8[q & 7];
        const int i = lane + 64 * k;
        re[q] = i < vs ? x[k] * p.hamming[i] : 0.0f;
        im[q] = 0.0f;
    }
#pragma unroll
    for (int st = 0; st < LR; st++) {
        const int h = 1 << st;
#pragma unroll
        for (int pq = 0; pq < R; pq++) {
            if (pq & h) continue;
            const int i = pq, j = pq + h, kk = pq & (h - 1);
            float tr, ti;
            if (kk == 0) {
                tr = re[j];
                ti = im[j];
            } else {
                const double wr = p.twiddle[2 * (h - 1 + kk)], wi = p.twiddle[2 * (h - 1 + kk) + 1];
                tr = (float)(wr * re[j] - wi * im[j]);
                ti = (float)(wr * im[j] + wi * re[j]);
            }
            re[j] = re[i] - tr;
            im[j] = im[i] - ti;
            re[i] += tr;
            im[i] += ti;
        }
    }
    {
        const int base = (int)(__brev((unsigned)lane) >> 26) * R;      // brev6(lane) * R
        float2 *d2 = reinterpret_cast<float2 *>(d);
#pragma unroll
        for (int q = 0; q < R; q++) d2[base + q] = make_float2(re[q], im[q]);
    }
    // A wave works on its own frame's buffer: between stages only ITS OWN LDS writes must be ordered before its reads
    // -- the LDS executes a wave's instructions in order; the fence keeps the compiler from reordering them -- so the
    // workgroup's barriers (eight or nine per frame, each waiting for the slowest of four unrelated waves) are gone.
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    wave_sync();
    // ---- the remaining stages through LDS; twiddle (stage, k) = entry h - 1 + k of the host-built table ----
    // (fully unrolled: the span is a compile-time constant in every stage -- masks and shifts fold --, and the twiddle's
    //  index is unsigned so that its load takes a 32-bit offset from the table's scalar base)
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const dbl2 *tw2 = reinterpret_cast<const dbl2 *>(p.twiddle);
#pragma unroll
    for (int h = R; h < FFT; h <<= 1) {
        float2 *d2 = reinterpret_cast<float2 *>(d);
#pragma unroll
        for (int q = 0; q < FFT / 128; q++) {
            const int bidx = lane + 64 * q;          // butterfly index 0 .. FFT/2-1
            const int k = bidx & (h - 1);
            const int i = ((bidx & ~(h - 1)) << 1) + k;
            const int j = i + h;
            const dbl2 w = tw2[(unsigned)(h - 1 + k)];
            const double wr = w.x, wi = w.y;
            const float2 vj = d2[j], vi = d2[i];
            const float tr = (float)(wr * vj.x - wi * vj.y);
            const float ti = (float)(wr * vj.y + wi * vj.x);
            d2[j] = make_float2(vi.x - tr, vi.y - ti);
            d2[i] = make_float2(vi.x + tr, vi.y + ti);
        }
        wave_sync();
    }
    // ---- power spectrum, mel filters, ln ----
#pragma unroll
    for (int k = 0; k < FFT / 128; k++) {
        const int i = lane + 64 * k;
        const float re = d[2 * i], im = d[2 * i + 1];
        pw[wave][i] = re * re + im * im;
    }
    __syncthreads();
    // The workgroup's 4 x nbanks outputs are dealt to its first threads -- thread q: frame q / nbanks, bank q % nbanks --
    // instead of nbanks lanes in each of the four waves: the bin sums and the double-precision ln (~150 instructions at
    // the f64 rate) are then issued by one or two waves, not by four waves with three quarters of their lanes idle.
    const int q = threadIdx.x, f = q / p.nbanks, b = q - f * p.nbanks;
    const int ofr = blockIdx.x * 4 + f;
    if (f < 4 && ofr < p.n_frames) {
        // bank b: bins whose falling edge is b (Banks == b) contribute p - v, then bins whose rising
        // edge... i.e. Banks == b + 1 contribute v; both runs are contiguous and in ascending order
        const float *pf = pw[f];
        const int a0 = p.run_begin[2 * b], a1 = p.run_end[2 * b], c0 = p.run_begin[2 * b + 1], c1 = p.run_end[2 * b + 1];
        float e = 0.0f;
        int i = a0;
        for (; i + 4 <= a1; i += 4) {               // four bins' operands requested together, added in bin order
            float pp[4], cc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { pp[k] = pf[i + k]; cc[k] = coef_s[i + k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const float v = cc[k] * pp[k]; e += (pp[k] - v); }
        }
        for (; i < a1; i++) {
            const float pp = pf[i], v = coef_s[i] * pp;
            e += (pp - v);
        }
        i = c0;
        for (; i + 4 <= c1; i += 4) {
            float pp[4], cc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { pp[k] = pf[i + k]; cc[k] = coef_s[i + k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) e += cc[k] * pp[k];
        }
        for (; i < c1; i++) e += coef_s[i] * pf[i];
        // ln: the host libm's sequence when the caller has named it, else in double, rounded once (the device's f32 logf is
        // only good to a few ulp)
        p.mel[(size_t)ofr * p.nbanks + b] = p.raw_energies ? e : frontend_ln(e, p.ln_form);
    }
}

// offlinenorm/sent_mean_norm (srec.cpp:1500-1511, matrix.h:194-199,245,2101-2116): the reference's column sums
// are SEQUENTIAL f32 sums over the frames, mean = sum * (1.0f / rows), x += -mean.  That exact order is the DEFAULT
// (lcrc_set_mean_order(ctx, 1), ABI 2); the fixed-shape tree further down is the opt-in (order 0).
//
// colmean_kernel: one workgroup per utterance; what bounds it is ONE dependent f32 add per frame and bank, so
// everything else is kept off that chain.  Wave 0 is the ADDER: lane b owns column b and does nothing but the chain.
// Waves 1-7 are LOADERS: they stream the utterance's rows (coalesced dword loads) into a ring of kMeanSlots LDS slots,
// TRANSPOSED -- slot[b][r], row pitch a multiple of 4 with an odd number of quads, so that the adder fetches four
// consecutive frames of its column with one conflict-free ds_read_b128 (a [r][b] image costs one LDS instruction per
// add, and the LDS issue, not the add, set the pace: 13 ns per frame in round 3).  A loader thread holds a whole
// chunk's share in registers and requests chunk c + 4 BEFORE it stores chunk c + 3, so a request has a whole step of
// the adder to arrive (with one request group in flight per step the loaders' round trips set the pace: 7 ns per
// frame); one barrier per chunk hands a slot over in each direction.  The adder requests the next 16 frames from LDS
// before it adds the current 16.
constexpr int kMeanSlots = 4;
constexpr int kMeanSlotFloats = 4352;           // 17 KiB per slot: 272 rows of 15 banks (+ pad), 176 rows of 23
constexpr int kMeanThreads = 512, kMeanLoaders = kMeanThreads - 64;
constexpr int kMeanPer = (kMeanSlotFloats + kMeanLoaders - 1) / kMeanLoaders;       // values per loader thread and chunk
constexpr int kMeanFuseRows = 2048;             // longest utterance of a launch up to which the means' workgroups subtract too
// SUBTRACT: the workgroup also subtracts its utterance's mean (x += -mean, srec.cpp:1510) once it has it -- the rows
// are in its L2 -- and the launch needs no submean_kernel behind it: the form for launches of short utterances (a list's
// 3-15 s files), where that kernel is one more launch that waits for CUs beside the posterior kernel's workgroups; one
// long utterance keeps the separate kernel (the whole chip subtracts in 5 us what one CU would in 9).
template <bool SUBTRACT>
__global__ __launch_bounds__(kMeanThreads) void colmean_kernel(float *mel, const int *frame_off, int nbanks,
                                                               float *means)
{
    extern __shared__ float ring[];             // [kMeanSlots][nbanks][RP]
    const int u = blockIdx.x;
    const int a = frame_off[u], rows = frame_off[u + 1] - a;
    if (rows <= 0) return;
    const int R = min(((kMeanSlotFloats / nbanks) - 4) & ~15, 1024);    // frames per slot: a multiple of 16
    const int RP = R + 4;                                               // pitch: RP / 4 odd -> 16 columns, 16 bank quads
    const int slot_floats = nbanks * RP;
    const float *x = mel + (size_t)a * nbanks;
    const int n_chunks = (rows + R - 1) / R;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // loaders: thread t takes elements t, t + kMeanLoaders, ... of a chunk's [r][b] image; (r, b) advance without division
    const int lt = (int)threadIdx.x - 64;
    const int step_r = kMeanLoaders / nbanks, step_b = kMeanLoaders % nbanks;
    const int r0 = lt / nbanks, b0 = lt % nbanks;
    auto request = [&](int c, float (&v)[kMeanPer]) {               // unconditional loads, clamped indices
        if (c >= n_chunks) return;
        const int n = min(R, rows - c * R) * nbanks;
        const float *src = x + (size_t)c * R * nbanks;
#pragma unroll
        for (int k = 0; k < kMeanPer; k++) v[k] = src[min(lt + k * kMeanLoaders, n - 1)];
    };
    auto commit = [&](int c, const float (&v)[kMeanPer]) {
        if (c >= n_chunks) return;
        const int n = min(R, rows - c * R) * nbanks;
        float *dst = ring + (size_t)(c % kMeanSlots) * slot_floats;
        int r = r0, b = b0;
#pragma unroll
        for (int k = 0; k < kMeanPer; k++) {
            if (lt + k * kMeanLoaders < n) dst[b * RP + r] = v[k];
            b += step_b; r += step_r;
            if (b >= nbanks) { b -= nbanks; r++; }
        }
    };
    float va[kMeanPer], vb[kMeanPer];
    if (wave > 0) {
        float vc[kMeanPer];
        request(0, va); request(1, vb); request(2, vc);             // the ring's first three chunks, all requested at once
        commit(0, va); commit(1, vb); commit(2, vc);
        request(3, vb);                                             // stored at step 0
    }
    __syncthreads();

    float sum = 0.0f;
    // step c: the adder consumes slot c % 4; the loaders store chunk c + 3 (requested a step ago) into the slot the adder
    // left at the last barrier and request chunk c + 4.  Two steps per pass: the register sets alternate by name.
    auto adder_step = [&](int c) {
        const int nr = min(R, rows - c * R);
        const float *col = ring + (size_t)(c % kMeanSlots) * slot_floats + lane * RP;
        const float4 *q = reinterpret_cast<const float4 *>(col);
        int r = 0;
        if (nr >= 16) {
            // two register sets, no copies: while one set's 16 frames are added the other set's are on their way
            // (the fences keep hipcc from sinking the requests behind the adds they are meant to overlap)
            float4 ra[4], rb[4];
            auto fetch = [&](float4 (&d)[4], int row) {
#pragma unroll
                for (int k = 0; k < 4; k++) d[k] = q[(row >> 2) + k];
            };
            auto add16 = [&](const float4 (&d)[4]) {
#pragma unroll
                for (int k = 0; k < 4; k++) { sum += d[k].x; sum += d[k].y; sum += d[k].z; sum += d[k].w; }
            };
            fetch(ra, 0);
            for (; r + 48 <= nr; r += 32) {
                fetch(rb, r + 16);
                __builtin_amdgcn_sched_barrier(0);
                add16(ra);
                __builtin_amdgcn_sched_barrier(0);
                fetch(ra, r + 32);
                __builtin_amdgcn_sched_barrier(0);
                add16(rb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (r + 32 <= nr) {
                fetch(rb, r + 16);
                __builtin_amdgcn_sched_barrier(0);
                add16(ra);
                add16(rb);
                r += 32;
            } else {
                add16(ra);
                r += 16;
            }
        }
        for (; r < nr; r++) sum += col[r];
    };
    for (int c = 0; c < n_chunks; c += 2) {
        if (wave > 0) { commit(c + 3, vb); request(c + 4, va); }
        else if (lane < nbanks) adder_step(c);
        __syncthreads();
        if (c + 1 >= n_chunks) break;
        if (wave > 0) { commit(c + 4, va); request(c + 5, vb); }
        else if (lane < nbanks) adder_step(c + 1);
        __syncthreads();
    }
    const float mean = sum * (1.0f / (float)rows);
    if (wave == 0 && lane < nbanks) means[(size_t)u * nbanks + lane] = mean;
    if constexpr (SUBTRACT) {
        __shared__ float mean_s[64];
        if (wave == 0 && lane < nbanks) mean_s[lane] = mean;
        __syncthreads();
        // element i of the utterance is bank i % nbanks: the bank advances with the thread's stride, no division
        const int n = rows * nbanks, tid = threadIdx.x;
        const int step_bb = kMeanThreads % nbanks;
        float *xw = mel + (size_t)a * nbanks;
        int b = tid % nbanks;
        for (int i = tid; i < n; i += 8 * kMeanThreads) {           // eight loads in flight per thread
            float v[8];
            int bb[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                v[k] = xw[min(i + k * kMeanThreads, n - 1)];
                bb[k] = b;
                b += step_bb;
                if (b >= nbanks) b -= nbanks;
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (i + k * kMeanThreads < n) xw[i + k * kMeanThreads] = v[k] + -mean_s[bb[k]];
        }
    }
}

// The OPT-IN order of the column sums (lcrc_set_mean_order(ctx, 0)): a FIXED-SHAPE TREE per utterance instead of the
// reference's dependent chain (the mean moves by ~1e-7 relative, posteriors by << 1e-4; 4.6 us whatever the length).  Rows
// are grouped in blocks of kMeanBlock rows counted from the utterance's first row; within a block, row lane q of
// Q = 256 / B' (B' = nbanks rounded up to a power of two) adds rows q, q+Q, ... in order, the Q lane sums are
// folded by halves (q += q + Q/2, ...), and submean_tree_kernel adds an utterance's block sums in block order.
// The shape depends on the utterance's own length and nbanks only: batching never changes a mean.
constexpr int kMeanBlock = 256;
__global__ __launch_bounds__(256) void colmean_block_kernel(const float *mel, const int *frame_off, const int *block_off,
                                                            int n_utts, int nbanks, float *partial)
{
    __shared__ float fold[256];
    const int blk = blockIdx.x;
    int lo = 0, hi = n_utts;                    // largest u with block_off[u] <= blk
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (block_off[mid] <= blk) lo = mid; else hi = mid;
    }
    const int first = frame_off[lo] + (blk - block_off[lo]) * kMeanBlock;
    const int rows = min(kMeanBlock, frame_off[lo + 1] - first);
    const int bp = nbanks <= 16 ? 16 : nbanks <= 32 ? 32 : 64, Q = 256 / bp;
    const int b = threadIdx.x % bp, q = threadIdx.x / bp;
    const float *x = mel + (size_t)first * nbanks + b;
    float sum = 0.0f;
    if (b < nbanks) {
        int r = q;
        for (; r + 7 * Q < rows; r += 8 * Q) {         // eight independent loads in flight, added in row order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = x[(size_t)(r + k * Q) * nbanks];
#pragma unroll
            for (int k = 0; k < 8; k++) sum += v[k];
        }
        for (; r < rows; r += Q) sum += x[(size_t)r * nbanks];
    }
    fold[threadIdx.x] = sum;
    __syncthreads();
    for (int h = Q / 2; h >= 1; h >>= 1) {
        if (q < h) fold[threadIdx.x] += fold[threadIdx.x + h * bp];
        __syncthreads();
    }
    if (q == 0 && b < nbanks) partial[(size_t)blk * nbanks + b] = fold[threadIdx.x];
}

// Second step of the tree: one thread per row; the workgroup first forms the means of the (few) utterances its
// 256 rows touch -- an utterance's block sums are added in block order, mean = sum * (1.0f / rows) -- in LDS,
// then every row subtracts.  (A separate "finish" launch for the means cost as much as either other kernel.)
__global__ __launch_bounds__(256) void submean_tree_kernel(float *mel, const float *partial, const int *frame_off,
                                                           const int *block_off, int n_utts, int n_rows, int nbanks)
{
    __shared__ float wg_means[4 * 64];          // up to 4 utterances x 64 banks cached; later ones are recomputed per row
    const int r = blockIdx.x * 256 + threadIdx.x;
    auto utt_of = [&](int row) {
        int lo = 0, hi = n_utts;                // largest u with frame_off[u] <= row (skips empty utterances)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (frame_off[mid] <= row) lo = mid; else hi = mid;
        }
        return lo;
    };
    auto mean_of = [&](int u, int b) {
        const int rows = frame_off[u + 1] - frame_off[u];
        const int k0 = block_off[u], k1 = block_off[u + 1];
        float sum = 0.0f;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {           // eight independent loads in flight, added in block order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = partial[(size_t)(k + q) * nbanks + b];
#pragma unroll
            for (int q = 0; q < 8; q++) sum += v[q];
        }
        for (; k < k1; k++) sum += partial[(size_t)k * nbanks + b];
        return sum * (1.0f / (float)rows);
    };
    const int row_first = blockIdx.x * 256, row_last = min(n_rows, row_first + 256) - 1;
    const int u_first = utt_of(row_first);
    {
        const int slot = threadIdx.x / 64, b = threadIdx.x % 64;       // utterance u_first + slot, bank b
        const int u = u_first + slot;
        if (b < nbanks && u < n_utts && frame_off[u] <= row_last && frame_off[u + 1] > frame_off[u])
            wg_means[threadIdx.x] = mean_of(u, b);
    }
    __syncthreads();
    if (r >= n_rows) return;
    const int u = utt_of(r);
    float *x = mel + (size_t)r * nbanks;
    if (u - u_first < 4) {
        const float *m = wg_means + (u - u_first) * 64;
        for (int b = 0; b < nbanks; b++) x[b] += -m[b];
    } else {
        for (int b = 0; b < nbanks; b++) x[b] += -mean_of(u, b);
    }
}

__global__ void submean_kernel(float *mel, const int *frame_off, int n_utts, int n_rows, int nbanks,
                               const float *means)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    int lo = 0, hi = n_utts;                    // largest u with frame_off[u] <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (frame_off[mid] <= r) lo = mid; else hi = mid;
    }
    float *x = mel + (size_t)r * nbanks;
    const float *m = means + (size_t)lo * nbanks;
    for (int b = 0; b < nbanks; b++) x[b] += -m[b];
}

// Host -> device by a kernel instead of a copy command: `src` is pinned host memory mapped into the device.  For callers
// whose upload must not stand in the copy engine's queue behind other contexts' large copy-backs (lcrc_wave_stage_energies).
__global__ __launch_bounds__(256) void pull_bytes_kernel(const uint4 *src, uint4 *dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

hipError_t pull_bytes_launch(const void *mapped_src, void *dst, size_t bytes, hipStream_t stream)
{
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>(512, (n16 + 255) / 256);
    pull_bytes_kernel<<<blocks, 256, 0, stream>>>(static_cast<const uint4 *>(mapped_src), static_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}

hipError_t frontend_launch(const FrontendParams &p, hipStream_t stream)
{
    if (p.n_frames <= 0) return hipSuccess;
    const dim3 grid((p.n_frames + 3) / 4), block(256);
    if (p.fft == 256) melbank_kernel<256><<<grid, block, 0, stream>>>(p);
    else if (p.fft == 512) melbank_kernel<512><<<grid, block, 0, stream>>>(p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// This file's code object onto the current device (the runtime loads a code object with the first launch of one of its
// kernels: 15-20 ms that a list's first waveform launch paid; lcrc_device_warmup pays them on its helper thread instead)
hipError_t frontend_preload_code()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&melbank_kernel<256>));
}

int meannorm_blocks(int rows) { return (rows + kMeanBlock - 1) / kMeanBlock; }

hipError_t meannorm_launch(float *mel, const int *frame_off, const int *block_off, int n_blocks, float *partial,
                           int n_utts, int n_rows, int nbanks, float *means, int max_utt_rows, hipStream_t stream)
{
    if (n_utts <= 0 || n_rows <= 0) return hipSuccess;
    if (nbanks > 64) return hipErrorInvalidValue;
    if (block_off == nullptr) {                 // the reference's sequential sums (lcrc_set_mean_order)
        const int R = std::min(((kMeanSlotFloats / nbanks) - 4) & ~15, 1024);
        const size_t lds = (size_t)kMeanSlots * nbanks * (R + 4) * sizeof(float);
        if (max_utt_rows > 0 && max_utt_rows <= kMeanFuseRows) {     // short utterances: means and subtraction in one launch
            colmean_kernel<true><<<n_utts, kMeanThreads, lds, stream>>>(mel, frame_off, nbanks, means);
            return hipGetLastError();
        }
        colmean_kernel<false><<<n_utts, kMeanThreads, lds, stream>>>(mel, frame_off, nbanks, means);
    } else {
        colmean_block_kernel<<<n_blocks, 256, 0, stream>>>(mel, frame_off, block_off, n_utts, nbanks, partial);
        submean_tree_kernel<<<(n_rows + 255) / 256, 256, 0, stream>>>(mel, partial, frame_off, block_off, n_utts, n_rows, nbanks);
        return hipGetLastError();
    }
    submean_kernel<<<(n_rows + 255) / 256, 256, 0, stream>>>(mel, frame_off, n_utts, n_rows, nbanks, means);
    return hipGetLastError();
}

}  // namespace phnrec
