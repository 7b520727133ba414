// pmc_device.h -- device-side arithmetic of the pair-energy path (gfx950).
//
// Build with -ffp-contract=off: every expression below is a sequence of single
// IEEE-754 double operations in the order the Julia reference performs them
// (src/models.jl, src/utils.jl); fused multiply-adds appear only where the
// reference itself calls muladd/fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PMC_WAVE 64

// ---- per-pair model (device copy of pmc_pair_model) -------------------------
struct DevModel {
    int kind;
    int n;
    double eps, eps4, sigma2, rcut2, shift;
    long long ndiv2_i;
    double ndiv2_f;
    double C0, C2_s2, C4_s4;
};

enum { K_LJ = 0, K_SOFT_INT = 1, K_SOFT_FLOAT = 2, K_SMOOTH_LJ = 3 };

// src/models.jl:30-34
__device__ __forceinline__ double dev_lennard_jones(double r2, double eps4, double sigma2)
{
    double x = sigma2 * (1.0 / r2);
    double x3 = x * x * x;
    return eps4 * (x3 * x3 - x3);
}

// Julia Base pow_body (Float64 ^ Int), see oracle/pmc_oracle.c:orc_pow_int
__device__ __forceinline__ double dev_pow_int(double x, long long n)
{
    double y = 1.0, xnlo = 0.0, ynlo = 0.0, err, hi;
    if (n == 3)
        return x * x * x;
    if (n < 0) {
        double rx = 1.0 / x;
        if (n == -2)
            return rx * rx;
        if (isfinite(x))
            xnlo = -__builtin_fma(x, rx, -1.0) * rx;
        x = rx;
        n = -n;
    }
    while (n > 1) {
        if (n & 1) {
            err = __builtin_fma(y, xnlo, x * ynlo);
            hi = x * y;
            ynlo = __builtin_fma(x, y, -hi);
            y = hi;
            ynlo += err;
        }
        err = x * 2.0 * xnlo;
        hi = x * x;
        xnlo = __builtin_fma(x, x, -hi);
        x = hi;
        xnlo += err;
        n >>= 1;
    }
    err = __builtin_fma(y, xnlo, x * ynlo);
    if (isfinite(x) && isfinite(err))
        return __builtin_fma(x, y, err);
    return x * y;
}

// n / d for operands whose quotient needs no range scaling (here: d = r2 inside a cutoff, n = 1
// or sigma^2).  These are exactly the operations hipcc emits for an IEEE double division --
// v_rcp_f64, two Newton steps, quotient, residual, correction -- minus v_div_scale / v_div_fixup,
// which are identities in that range: bit-identical to n / d there (tests/test_gpu_parity.py).
__device__ __forceinline__ double dev_div_inrange(double n, double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = n * y;
    double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, y, q);
}

// x^(n/2) for odd n (src/models.jl:66 makes ndiv2 = n / 2 a Float64 and calls Julia's table-driven
// Float64^Float64, < 1 ulp, not reproducible here). Evaluated as sqrt(x^n) with the compensated integer
// power: <= 0.75 ulp from the true value and the same bits on the host, the device and in the oracle.
__device__ __forceinline__ double dev_pow_half(double x, double ndiv2_f)
{
    return __builtin_sqrt(dev_pow_int(x, (long long)(2.0 * ndiv2_f)));
}

// throughput-engine variants of the pair potentials (same operations, guard-free division)
template <int KIND>
__device__ __forceinline__ double dev_potential_inrange(double r2, const DevModel &m);

// potential(r2, model): src/models.jl:72-74, 121-123, 162-167
template <int KIND>
__device__ __forceinline__ double dev_potential(double r2, const DevModel &m)
{
    if (KIND == K_LJ) {
        return dev_lennard_jones(r2, m.eps4, m.sigma2) - m.shift;
    } else if (KIND == K_SOFT_INT) {
        return m.eps * dev_pow_int(m.sigma2 / r2, m.ndiv2_i) - m.shift;
    } else if (KIND == K_SOFT_FLOAT) {
        return m.eps * dev_pow_half(m.sigma2 / r2, m.ndiv2_f) - m.shift;
    } else {
        double lj = dev_lennard_jones(r2, m.eps4, m.sigma2);
        double sh = m.eps4 * (m.C0 + r2 * __builtin_fma(r2, m.C4_s4, m.C2_s2));
        return lj + sh;
    }
}

template <int KIND>
__device__ __forceinline__ double dev_potential_inrange(double r2, const DevModel &m)
{
    if (KIND == K_LJ || KIND == K_SMOOTH_LJ) {
        double x = m.sigma2 * dev_div_inrange(1.0, r2);
        double x3 = x * x * x;
        double lj = m.eps4 * (x3 * x3 - x3);
        if (KIND == K_LJ) return lj - m.shift;
        double sh = m.eps4 * (m.C0 + r2 * __builtin_fma(r2, m.C4_s4, m.C2_s2));
        return lj + sh;
    } else if (KIND == K_SOFT_INT) {
        return m.eps * dev_pow_int(dev_div_inrange(m.sigma2, r2), m.ndiv2_i) - m.shift;
    } else {
        return dev_potential<KIND>(r2, m);
    }
}

__device__ __forceinline__ double dev_potential_dyn(double r2, const DevModel &m)
{
    switch (m.kind) {
    case K_LJ: return dev_potential<K_LJ>(r2, m);
    case K_SOFT_INT: return dev_potential<K_SOFT_INT>(r2, m);
    case K_SOFT_FLOAT: return dev_potential<K_SOFT_FLOAT>(r2, m);
    default: return dev_potential<K_SMOOTH_LJ>(r2, m);
    }
}

// src/utils.jl:15-18 vector_1D, literal form (division)
__device__ __forceinline__ double dev_vector_1d(double c1, double c2, double L)
{
    double dx = c1 - c2;
    return dx - rint(dx / L) * L;
}

// Same value whenever L >= 4 rcut: the quotient is only rounded differently when
// dx/L sits within an ulp of a half-integer, where |dx - n L| ~ L/2 > rcut for
// either n, so the pair contributes exactly 0 both ways (DESIGN.md "min-image").
__device__ __forceinline__ double dev_vector_1d_fast(double c1, double c2, double L, double invL)
{
    double dx = c1 - c2;
    return dx - rint(dx * invL) * L;
}

// ---- Julia float fld (base/div.jl), for get_cell src/neighbours.jl:178-180 ----
__device__ __forceinline__ double dev_jl_fld(double x, double y)
{
    double r = fmod(x, y);
    double m;
    if (r == 0.0)
        m = copysign(r, y);
    else if ((r > 0.0) != (y > 0.0))
        m = r + y;
    else
        m = r;
    return rint((x - m) / y);
}

__device__ __forceinline__ long long dev_fld_i(long long a, long long b)
{
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0)))
        q -= 1;
    return q;
}

// ---- exact (order-independent) energy accumulation --------------------------
// Pair energies are clamped to UMAX and rounded to multiples of 2^-32; sums of
// up to 2^10 such terms are exact in double, so any lane mapping, reduction tree
// or atomic order gives the same bits (DESIGN.md "exact accumulation").
#define PMC_UMAX 1024.0
#define PMC_CQ 1572864.0
#define PMC_FIX 4294967296.0

__device__ __forceinline__ double dev_quant(double u)
{
    double c = fmin(u, PMC_UMAX);
    // volatile-free: -ffp-contract=off and no fast-math keep (c + CQ) - CQ intact
    return (c + PMC_CQ) - PMC_CQ;
}

// ---- Philox4x32-10 ------------------------------------------------------------
__device__ __host__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { TAG_SEQ0 = 1, TAG_SEQ1 = 2, TAG_SHIFT = 3, TAG_SLOT0 = 4, TAG_SLOT1 = 5, TAG_PT = 6 };

// ---- deterministic log / sincos (fixed IEEE operation sequence) -----------------
__device__ __host__ __forceinline__ double det_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t b = (uint64_t)__builtin_bit_cast(uint64_t, x);
    long long k = (long long)((b >> 52) & 0x7ff) - 1023;
    uint64_t mant = b & 0x000fffffffffffffull;
    if (mant >= 0x6a09e667f3bcdull) {
        k += 1;
        b = mant | 0x3fe0000000000000ull;
    } else {
        b = mant | 0x3ff0000000000000ull;
    }
    double m = __builtin_bit_cast(double, b);
    double f = m - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) as a fixed sequence of IEEE operations (fdlibm e_exp.c: k = round(x / ln 2), r = x - k ln 2 in two pieces,
// exp(r) from a degree-5 rational correction, scaling by 2^k through the exponent field): the same bits on the host, the
// device and in the oracle (EnergyBias weights, src/moves.jl:270-273).  Finite |x| < 700 only; beyond that it saturates.
__device__ __host__ __forceinline__ double det_exp(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    double hi = x, lo = 0.0;
    long long k = 0;
    if (x > 0.34657359027997264 || x < -0.34657359027997264) {      // |x| > ln(2) / 2
        k = (long long)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
        const double t = (double)k;
        hi = x - t * ln2_hi;
        lo = t * ln2_lo;
        x = hi - lo;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y;
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    uint64_t b = (uint64_t)__builtin_bit_cast(uint64_t, y);
    b += (uint64_t)(k << 52);                                        // |k| <= 1010: y stays a normal number
    return __builtin_bit_cast(double, b);
}

__device__ __host__ __forceinline__ void det_sincos2pi(uint32_t w, double *sn, double *cs)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                 C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11,
                 PIO4 = 7.85398163397448278999e-01;
    uint32_t oct = w >> 29;
    uint32_t rem = w & 0x1fffffffu;
    uint32_t r = (oct & 1) ? (0x20000000u - rem) : rem;
    double y = ((double)r * (1.0 / 536870912.0)) * PIO4;
    double z = y * y;
    double ps = y + y * (z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))))));
    double pc = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
    double s, c;
    switch (oct) {
    case 0: s = ps;  c = pc;  break;
    case 1: s = pc;  c = ps;  break;
    case 2: s = pc;  c = -ps; break;
    case 3: s = ps;  c = -pc; break;
    case 4: s = -ps; c = -pc; break;
    case 5: s = -pc; c = -ps; break;
    case 6: s = -pc; c = ps;  break;
    default: s = -ps; c = pc; break;
    }
    *sn = s;
    *cs = c;
}

__device__ __host__ __forceinline__ void box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1)
{
    double u1 = ((double)w1 + 1.0) * (1.0 / 4294967296.0);
    double r = sqrt(-2.0 * det_log(u1));
    double s, c;
    det_sincos2pi(w2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

__device__ __host__ __forceinline__ double uniform53(uint32_t hi, uint32_t lo)
{
    double a = (double)(hi >> 5), b = (double)(lo >> 6);
    return ((a * 67108864.0 + b) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __host__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

// ---- wave helpers -------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
        v += __shfl_xor(v, s, PMC_WAVE);
    return v;
}

__device__ __forceinline__ int wave_lane() { return threadIdx.x & (PMC_WAVE - 1); }

__device__ __forceinline__ double wave_bcast(double v, int src)
{
    return __shfl(v, src, PMC_WAVE);
}
