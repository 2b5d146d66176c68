// f110_rng.hpp — the scan-noise stream of the reference, generated on the device.
//
// The reference adds `rng.normal(0., 0.01, size=num_beams)` to every scan (laser_models.py:450-452)
// with `rng = np.random.default_rng(seed=self.seed)` re-created at every reset
// (base_classes.py:204), i.e. NumPy's PCG64 bit generator (128-bit LCG, XSL-RR output) feeding
// NumPy's 256-layer ziggurat `random_standard_normal` (numpy/random/src/distributions/
// distributions.c; third-party dependency of the reference, pinned `numpy>=1.18,<=1.22` in its
// setup.py, present here as 2.2.6 — the algorithm and its constants are unchanged across those
// versions).  This header restates that algorithm so that the device produces the SAME doubles:
//
//   pcg_step / pcg_output      pcg64.h: state = state * MULT + inc; out = rotr64(hi ^ lo, hi >> 58)
//   pcg64_seed_from_u64        SeedSequence(entropy).generate_state(4, uint64) + pcg64_set_seed
//   zig_attempt                one pass of random_standard_normal's for(;;) body for the draw at a
//                              given stream position: accept (1 draw), wedge test (2 draws, may
//                              reject), tail loop (1 + 2k draws)
//   zig_chain_starts           which positions of a 64-draw chunk start an attempt, given how many
//                              draws every attempt consumes — the only sequential dependency of the
//                              stream, resolved with a few scalar iterations per chunk
//   log1p_glibc                glibc 2.35's log1p (the tail's output value goes through it, so it
//                              must match to the bit; checked on 5*10^7 arguments)
//
// Everything is __host__ __device__ so tests/host_harness runs the same code on the CPU against
// NumPy's live stream.  The wedge test compares against exp(); device and glibc exp may differ by
// an ulp, which can flip the comparison only when both sides agree to ~2^-52 relative (about one
// wedge test in 10^15) — accepted like the other measure-zero deviations listed in DESIGN.md.
#pragma once

#include "f110_math.hpp"
#include "f110_ziggurat_tables.hpp"

namespace f110 {

struct U128 {
    uint64_t hi, lo;
};

F110_HD uint64_t mulhi_u64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

F110_HD U128 mul128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = mulhi_u64(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
    return r;
}

F110_HD U128 add128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}

// PCG_DEFAULT_MULTIPLIER_128 (pcg64.h)
constexpr uint64_t kPcgMultHi = 2549297995355413924ULL, kPcgMultLo = 4865540595714422341ULL;

F110_HD U128 pcg_step(U128 s, U128 inc)
{
    const U128 m = {kPcgMultHi, kPcgMultLo};
    return add128(mul128(s, m), inc);
}

// pcg_output_xsl_rr_128_64 of the state AFTER the step (pcg64_next64 steps first)
F110_HD uint64_t pcg_output(U128 s)
{
    const uint64_t x = s.hi ^ s.lo;
    const unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

// next_double: (next_uint64 >> 11) * 2^-53
F110_HD double pcg_to_double(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

// Jump constants: the state i steps ahead is A^i * s + G_i * inc with G_i = 1 + A + ... + A^(i-1)
// (both independent of the seed), i = 0..64.
struct PcgJump {
    U128 a[65], g[65];
};

inline void pcg_jump_table(PcgJump &t)
{
    const U128 m = {kPcgMultHi, kPcgMultLo};
    t.a[0] = {0, 1};
    t.g[0] = {0, 0};
    for (int i = 1; i <= 64; ++i) {
        t.a[i] = mul128(t.a[i - 1], m);
        t.g[i] = add128(mul128(t.g[i - 1], m), U128{0, 1});
    }
}

// np.random.SeedSequence(seed).generate_state(4, np.uint64) -> pcg64_set_seed: out = {state.hi,
// state.lo, inc.hi, inc.lo} of np.random.PCG64(seed) (numpy/random/bit_generator.pyx,
// _pcg64.pyx, src/pcg64/pcg64.h).  seed is a non-negative integer below 2^64.
inline void pcg64_seed_from_u64(uint64_t seed, uint64_t out[4])
{
    const uint32_t INIT_A = 0x43b0d7e5u, MULT_A = 0x931e8875u, INIT_B = 0x8b51f9ddu, MULT_B = 0x58f38dedu;
    const uint32_t MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t entropy[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_ent = (seed >> 32) ? 2 : 1;
    uint32_t pool[4], hc = INIT_A;
    auto hashmix = [&](uint32_t v) {
        v ^= hc;
        hc *= MULT_A;
        v *= hc;
        v ^= v >> 16;
        return v;
    };
    auto mix = [&](uint32_t x, uint32_t y) {
        uint32_t r = MIX_L * x - MIX_R * y;
        r ^= r >> 16;
        return r;
    };
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? entropy[i] : 0u);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
    uint32_t w[8], hb = INIT_B;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= MULT_B;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t q[4];
    for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    // pcg64_set_seed: initstate = (q0 << 64) | q1, initseq = (q2 << 64) | q3; pcg_setseq_128_srandom_r
    const U128 initstate = {q[0], q[1]}, initseq = {q[2], q[3]};
    U128 inc = {(initseq.hi << 1) | (initseq.lo >> 63), (initseq.lo << 1) | 1ull};
    U128 st = {0, 0};
    st = pcg_step(st, inc);
    st = add128(st, initstate);
    st = pcg_step(st, inc);
    out[0] = st.hi;
    out[1] = st.lo;
    out[2] = inc.hi;
    out[3] = inc.lo;
}

// ---- glibc 2.35 log1p (sysdeps/ieee754/dbl-64/s_log1p.c: fdlibm's algorithm with the
// polynomial evaluated in the split form glibc uses), restated for x in (-1, 0] — the only
// arguments the ziggurat tail feeds it (-next_double).
F110_HD int32_t high_word(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2hiint(x);
#else
    uint64_t u;
    memcpy(&u, &x, 8);
    return (int32_t)(u >> 32);
#endif
}

F110_HD double with_high_word(double x, int32_t h)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double(h, __double2loint(x));
#else
    uint64_t u;
    memcpy(&u, &x, 8);
    u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)h << 32);
    memcpy(&x, &u, 8);
    return x;
#endif
}

F110_HD double log1p_glibc(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double two54 = 1.80143985094819840000e+16;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double f = 0., c = 0., u;
    int32_t hx = high_word(x), hu = 0, k = 1;
    const int32_t ax = hx & 0x7fffffff;
    if (hx < 0x3FDA827A) {  // x < 0.41422
        if (ax >= 0x3ff00000) return (x == -1.0) ? -HUGE_VAL : (x - x) / (x - x);
        if (ax < 0x3e200000) {  // |x| < 2^-29
            if (two54 + x > 0.0 && ax < 0x3c900000) return x;
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) {  // -0.2929 < x < 0.41422
            k = 0;
            f = x;
            hu = 1;
        }
    }
    if (k != 0) {
        u = 1.0 + x;
        hu = high_word(u);
        k = (hu >> 20) - 1023;
        c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
        c /= u;
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = with_high_word(u, hu | 0x3ff00000);
        } else {
            k += 1;
            u = with_high_word(u, hu | 0x3fe00000);
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    const double hfsq = 0.5 * f * f;
    if (hu == 0) {  // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        const double R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f), z = s * s;
    const double R1 = z * Lp1, z2 = z * z, R2 = Lp2 + z * Lp3, z4 = z2 * z2, R3 = Lp4 + z * Lp5, z6 = z4 * z2,
                 R4 = Lp6 + z * Lp7;
    const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// ---- ziggurat -------------------------------------------------------------------------------
constexpr double kZigR = 3.6541528853610087963519472518;       // ziggurat_nor_r
constexpr double kZigInvR = 0.27366123732975827203338247596;   // ziggurat_nor_inv_r

struct ZigTables {
    const uint64_t *k;  // ki_double
    const double *w;    // wi_double
    const double *f;    // fi_double
};

struct ZigAttempt {
    double val;  // the standard normal this attempt returns (when emit)
    int len;     // draws consumed, this one included
    bool emit;   // false: wedge test failed, random_standard_normal loops and the next attempt starts len draws on
};

// One pass of random_standard_normal's for(;;) body (distributions.c) for draw `r`; `st` is the
// generator state that produced r (further draws step from it).
F110_HD ZigAttempt zig_attempt(uint64_t r, U128 st, U128 inc, const ZigTables &t)
{
    ZigAttempt o;
    o.len = 1;
    o.emit = true;
    const int idx = (int)(r & 0xff);
    r >>= 8;
    const int sign = (int)(r & 0x1);
    const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
    double x = (double)rabs * t.w[idx];
    if (sign & 0x1) x = -x;
    o.val = x;
    if (rabs < t.k[idx]) return o;  // 99.3 % of the time
    if (idx == 0) {
        for (;;) {
            st = pcg_step(st, inc);
            const double u1 = pcg_to_double(pcg_output(st));
            st = pcg_step(st, inc);
            const double u2 = pcg_to_double(pcg_output(st));
            o.len += 2;
            const double xx = -kZigInvR * log1p_glibc(-u1);
            const double yy = -log1p_glibc(-u2);
            if (yy + yy > xx * xx) {
                o.val = ((rabs >> 8) & 0x1) ? -(kZigR + xx) : kZigR + xx;
                return o;
            }
        }
    }
    st = pcg_step(st, inc);
    const double u = pcg_to_double(pcg_output(st));
    o.len = 2;
    o.emit = ((t.f[idx - 1] - t.f[idx]) * u + t.f[idx]) < exp(-0.5 * x * x);
    return o;
}

F110_HD int ctz_u64(uint64_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__ffsll((unsigned long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}

F110_HD int popc_u64(uint64_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__popcll((unsigned long long)m);
#else
    return __builtin_popcountll(m);
#endif
}

// Positions (bits) of a 64-draw chunk at which an attempt starts.  `multi`: positions whose
// attempt — were one to start there — consumes more than one draw; len_of(p) its length;
// skip_in: leading positions already consumed by an attempt of the previous chunk; skip_out: the
// same for the next chunk.
template <typename LenOf>
F110_HD uint64_t zig_chain_starts(uint64_t multi, int skip_in, LenOf len_of, int &skip_out)
{
    if (skip_in >= 64) {
        skip_out = skip_in - 64;
        return 0ull;
    }
    uint64_t starts = ~0ull << skip_in;
    skip_out = 0;
    uint64_t m = multi & starts;
    while (m) {
        const int p = ctz_u64(m);
        m &= m - 1;
        const int last = p + len_of(p) - 1;  // the attempt at p consumes p .. last
        const uint64_t upto = last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull);
        const uint64_t kill = upto & ~((2ull << p) - 1ull);  // p+1 .. min(last, 63)
        starts &= ~kill;
        m &= ~kill;
        if (last >= 64) skip_out = last - 63;
    }
    return starts;
}

// position of the (n+1)-th set bit of m (n < popcount(m))
F110_HD int nth_set_bit(uint64_t m, int n)
{
    for (int i = 0; i < n; ++i) m &= m - 1;
    return ctz_u64(m);
}

// the state `n` steps ahead
F110_HD U128 pcg_advance(U128 s, U128 inc, const U128 *ja, const U128 *jg, int n)
{
    while (n > 64) {
        s = add128(mul128(ja[64], s), mul128(jg[64], inc));
        n -= 64;
    }
    return add128(mul128(ja[n], s), mul128(jg[n], inc));
}

}  // namespace f110
