// ORACLE — TEST INFRASTRUCTURE ONLY (see num.h header).
//
// sampler.h — per-lane random number stream.
// Follows reference src/core/sampler.cpp:6-42 and include/psdr/core/sampler.h:8-40.
// drjit::PCG32 itself is NOT in /root/reference (ext/drjit is an empty, un-pinned submodule,
// .gitmodules:1-3; API generation drjit 0.4.x).  Its published algorithm (drjit/random.h, which
// is M. O'Neill's pcg32 "XSH RR 64/32") is restated here and pinned by the public pcg32-demo
// known-answer vector (tests/test_oracle_kat.py).
#pragma once
#include <cstdint>
#include <cstring>

namespace orc {

constexpr uint64_t PCG32_DEFAULT_STATE = 0x853c49e6748fea9bULL;   // Sampler::m_base_seed (sampler.h:38)
constexpr uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;

// sampler.cpp:6-17.  The reference instantiates this with UInt64 lanes (sampler.cpp:27), so the
// "TEA" rounds run in 64-bit arithmetic with a 32-bit running sum; the result is
// v0 + (v1 << 32) mod 2^64.  This is NOT the usual 32-bit TEA.
inline uint64_t sample_tea_64(uint64_t v0, uint64_t v1, int rounds = 4) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cULL) ^ (v1 + (uint64_t) sum) ^ ((v1 >> 5) + 0xc8013ea4ULL);
        v1 += ((v0 << 4) + 0xad90777dULL) ^ (v0 + (uint64_t) sum) ^ ((v0 >> 5) + 0x7e95761eULL);
    }
    return v0 + (v1 << 32);
}

struct PCG32 {
    uint64_t state = 0, inc = 0;
    // drjit PCG32::seed(size=1, initstate, initseq)
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0;
        inc = (initseq << 1) | 1u;
        next_uint32();
        state += initstate;
        next_uint32();
    }
    uint32_t next_uint32() {
        uint64_t old = state;
        state = old * PCG32_MULT + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    // drjit next_float32: bitcast((u >> 9) | 0x3f800000) - 1
    float next_float32() {
        uint32_t u = (next_uint32() >> 9) | 0x3f800000u;
        float f;
        std::memcpy(&f, &u, 4);
        return f - 1.f;
    }
    // advance by `delta` draws in O(log delta) (pcg32_advance, Brown 1994)
    void advance(uint64_t delta) {
        uint64_t cur_mult = PCG32_MULT, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
        while (delta > 0) {
            if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
};

// One lane of the reference's `Sampler` (an N-lane PCG32).  `seed_value` is the per-lane entry
// of the array handed to Sampler::seed (arange(N)+seed, integrator.cpp:24,61; scene.cpp:333),
// `lane` is its index.  Draw order for next_2d / next_nd<3>: x first (the reference leaves the
// argument evaluation order of Vector2f(next_1d(), next_1d()) to the compiler, sampler.h:19-21).
struct LaneSampler {
    PCG32 rng;
    void seed(uint64_t seed_value, uint64_t lane) {
        uint64_t s = seed_value + PCG32_DEFAULT_STATE;          // sampler.cpp:23
        rng.seed(sample_tea_64(s, lane), sample_tea_64(lane, s)); // sampler.cpp:27
    }
    float next_1d() { return rng.next_float32(); }
};

} // namespace orc
