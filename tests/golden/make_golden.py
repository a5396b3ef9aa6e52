"""Generates the integer known-answer fixtures under tests/golden/ from a PURE-PYTHON restatement
(arbitrary-precision ints, no numpy, no oracle, no product code) of the reference's sampler:
  * sample_tea_64 as instantiated with 64-bit lanes (reference src/core/sampler.cpp:6-17,27)
  * PCG32 (drjit/random.h == pcg32 XSH-RR 64/32; anchored by the public demo vector
    seed 42 / stream 54 -> a15c02b7 7b47f409 ba1d3330 83d2f293 bfa4784b cbed606e)
  * Sampler::seed + next_1d (sampler.cpp:19-42)
The reference itself cannot be imported here (drjit missing), so these are self-generated pins that
document the 64-bit-TEA quirk; both the C++ oracle and the HIP sampler must reproduce them bit-exactly.

    python tests/golden/make_golden.py
"""
import json
import os
import random
import struct

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1
MULT = 0x5851f42d4c957f2d
DEFAULT_STATE = 0x853c49e6748fea9b


def tea64(v0, v1, rounds=4):
    s = 0
    for _ in range(rounds):
        s = (s + 0x9e3779b9) & M32
        v0 = (v0 + ((((v1 << 4) & M64) + 0xa341316c) & M64 ^ ((v1 + s) & M64) ^ (((v1 >> 5) + 0xc8013ea4) & M64))) & M64
        v1 = (v1 + ((((v0 << 4) & M64) + 0xad90777d) & M64 ^ ((v0 + s) & M64) ^ (((v0 >> 5) + 0x7e95761e) & M64))) & M64
    return (v0 + ((v1 << 32) & M64)) & M64


class PCG32:
    def __init__(self, initstate, initseq):
        self.state = 0
        self.inc = ((initseq << 1) | 1) & M64
        self.next_u32()
        self.state = (self.state + initstate) & M64
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * MULT + self.inc) & M64
        xs = (((old >> 18) ^ old) >> 27) & M32
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & M32

    def next_f32_bits(self):
        u = (self.next_u32() >> 9) | 0x3f800000
        f = struct.unpack("<f", struct.pack("<I", u))[0] - 1.0
        return struct.unpack("<I", struct.pack("<f", f))[0]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    p = PCG32(42, 54)
    demo = [p.next_u32() for _ in range(6)]
    assert demo == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e], demo
    rnd = random.Random(2024)
    tea = [(0, 0), (1, 0), (0, 1), (DEFAULT_STATE, 0), (DEFAULT_STATE + 8388607, 8388607), (M64, M64)]
    tea += [(rnd.getrandbits(64), rnd.getrandbits(64)) for _ in range(26)]
    with open(os.path.join(here, "tea64.json"), "w") as fh:
        json.dump([[str(a), str(b), str(tea64(a, b))] for a, b in tea], fh, indent=0)
    rows = []
    for seed_value, lane in [(0, 0), (1, 1), (5, 3), (8388607, 8388607), (12345, 77), (2 ** 31 + 9, 2 ** 31 - 1)]:
        s = (seed_value + DEFAULT_STATE) & M64
        g = PCG32(tea64(s, lane), tea64(lane, s))
        rows.append({"seed_value": str(seed_value), "lane": str(lane), "bits": [g.next_f32_bits() for _ in range(16)]})
    with open(os.path.join(here, "sampler_floats.json"), "w") as fh:
        json.dump(rows, fh)
    print("wrote tea64.json, sampler_floats.json")


if __name__ == "__main__":
    main()
