// exr_piz.h — PIZ chunk decoder for the EXR reader (psdr_jit_amd/exr.py); see exr_piz.cpp
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace psdr_host {
// src: one PIZ-compressed chunk (nx samples x ny lines); words_per_sample: per channel, in file order, the number of 16-bit
// words of one sample (HALF 1, FLOAT / UINT 2).  out: ny lines, each = every channel's nx samples (the uncompressed chunk).
void piz_decode_chunk(const uint8_t *src, size_t n_src, int nx, int ny, const std::vector<int> &words_per_sample, uint16_t *out);
}  // namespace psdr_host
