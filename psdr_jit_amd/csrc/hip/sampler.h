// sampler.h — per-lane random stream of the reference Sampler, recomputed in registers.
//
// reference src/core/sampler.cpp:6-42: lane i of an N-lane drjit::PCG32 is seeded with
//   initstate = tea64(seed_i + PCG32_DEFAULT_STATE, i), initseq = tea64(i, seed_i + PCG32_DEFAULT_STATE)
// where the TEA rounds run on 64-bit lanes with a 32-bit sum (sampler.cpp:8,27).  drjit::PCG32 is
// pcg32 XSH-RR 64/32; next_float32 = bitcast((u >> 9) | 0x3f800000) - 1.
//
// The reference keeps 16 B of RNG state per lane in HBM (3 samplers x 8.4 M lanes at 512x512x32).
// Because every render call draws a FIXED number of values per lane (2 + 5*depth for the interior
// sampler, 1 + 10*depth for primary edges, 3 for secondary edges), the state after any number of
// calls is a closed-form function of (seed, lane, draws so far): the kernels re-derive it with
// pcg32's O(log n) skip-ahead and never touch memory.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define PSDR_RNG_HD __host__ __device__ inline
#else
#define PSDR_RNG_HD inline
#endif

namespace psdr {

constexpr uint64_t kPcgDefaultState = 0x853c49e6748fea9bULL;
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dULL;

PSDR_RNG_HD uint64_t tea64(uint64_t v0, uint64_t v1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cULL) ^ (v1 + (uint64_t) sum) ^ ((v1 >> 5) + 0xc8013ea4ULL);
        v1 += ((v0 << 4) + 0xad90777dULL) ^ (v0 + (uint64_t) sum) ^ ((v0 >> 5) + 0x7e95761eULL);
    }
    return v0 + (v1 << 32);
}

// pcg32's skip-ahead by `delta` draws is the affine map state -> mult * state + plus, and plus is LINEAR in the stream's increment
// (every term of the doubling recurrence below carries one factor inc): plus = inc * g with g the same recurrence run for inc = 1.
// (mult, g) depend on delta only, so the host runs the O(log delta) loop once per launch and every work item applies the map with two
// 64-bit multiplications - the same numbers mod 2^64 as the per-lane loop (a continuing sampler, e.g. iteration 1000 of an optimisation,
// otherwise pays ~35 instructions x log2(delta) per work item: tools/time_skip.py).
struct SkipAhead {
    uint64_t mult, g;
};
PSDR_RNG_HD SkipAhead skip_ahead(uint64_t delta) {
    uint64_t cur_mult = kPcgMult, cur_plus = 1u, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return SkipAhead{acc_mult, acc_plus};
}

struct LaneRng {
    uint64_t state, inc;

    PSDR_RNG_HD uint32_t next_u32() {
        uint64_t old = state;
        state = old * kPcgMult + inc;
        uint32_t xs = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    PSDR_RNG_HD float next_1d() {
        union { uint32_t u; float f; } c;
        c.u = (next_u32() >> 9) | 0x3f800000u;
        return c.f - 1.f;
    }
    PSDR_RNG_HD void advance(uint64_t delta) {
        uint64_t cur_mult = kPcgMult, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        while (delta > 0) {
            if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
    // Sampler::seed for one lane, then skip the draws earlier render calls consumed
    PSDR_RNG_HD void seed(uint64_t seed_value, uint64_t lane, uint64_t skip) {
        uint64_t s = seed_value + kPcgDefaultState;
        uint64_t initstate = tea64(s, lane), initseq = tea64(lane, s);
        state = 0;
        inc = (initseq << 1) | 1u;
        next_u32();
        state += initstate;
        next_u32();
        if (skip) advance(skip);
    }
    // ... with the skip-ahead as its precomputed map (the kernels' form)
    PSDR_RNG_HD void seed(uint64_t seed_value, uint64_t lane, const SkipAhead &sk) {
        seed(seed_value, lane, (uint64_t) 0);
        if (sk.mult != 1u || sk.g != 0u) state = sk.mult * state + sk.g * inc;      // (a launch-wide constant: a scalar branch)
    }
};

} // namespace psdr
