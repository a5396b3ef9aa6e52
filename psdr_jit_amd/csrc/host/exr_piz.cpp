// exr_piz.cpp — decoder for OpenEXR's PIZ compression (one scanline chunk), written from the published description of the
// format (OpenEXR "Technical Introduction" / file-layout documents: range-compacting bitmap + lookup table, canonical Huffman
// code over 16-bit symbols with an 8-bit run-length escape, two-dimensional Haar-style wavelet with 14- and 16-bit variants).
// The reference reads its environment maps and textures through tinyexr (ext/, BitmapLoader::load_openexr_rgba,
// src/core/bitmap_loader.cpp); its tutorial environment map (tutorials/data/envmap/ballroom_1k.exr) is PIZ-compressed.
// Host-side setup code: runs once per file, nothing here is on the rendering path.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "exr_piz.h"

namespace psdr_host {
namespace {

constexpr int kEncBits = 16, kEncSize = (1 << kEncBits) + 1;     // 65536 symbols + the run-length escape
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;
constexpr int kMaxCodeLen = 58;

struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;
    int n = 0;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t get(int bits) {                                     // most significant bit first
        while (n < bits) {
            if (p >= end) throw std::runtime_error("EXR/PIZ: truncated Huffman data");
            acc = (acc << 8) | *p++;
            n += 8;
        }
        n -= bits;
        return (uint32_t) ((acc >> n) & ((1ull << bits) - 1ull));
    }
};

uint32_t rd32(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24); }

// Huffman section: [min symbol][max symbol][table bytes][data bits][reserved] (5 x u32), packed code lengths, code bits
void huf_uncompress(const uint8_t *src, size_t n_src, uint16_t *out, size_t n_out) {
    if (n_out == 0) return;
    if (n_src < 20) throw std::runtime_error("EXR/PIZ: truncated Huffman header");
    const uint32_t im = rd32(src), iM = rd32(src + 4), n_bits = rd32(src + 12);
    if (im >= (uint32_t) kEncSize || iM >= (uint32_t) kEncSize) throw std::runtime_error("EXR/PIZ: bad Huffman symbol range");
    std::vector<uint8_t> len((size_t) kEncSize, 0);
    BitReader tr(src + 20, src + n_src);
    for (uint32_t s = im; s <= iM; ++s) {                        // code lengths, 6 bits each, with two zero-run escapes
        const uint32_t l = tr.get(6);
        if (l == (uint32_t) kLongZeroRun) {
            uint32_t run = tr.get(8) + kShortestLongRun;
            if (s + run > iM + 1) throw std::runtime_error("EXR/PIZ: bad code-length table");
            s += run - 1;
        } else if (l >= (uint32_t) kShortZeroRun) {
            uint32_t run = l - kShortZeroRun + 2;
            if (s + run > iM + 1) throw std::runtime_error("EXR/PIZ: bad code-length table");
            s += run - 1;
        } else {
            len[s] = (uint8_t) l;
        }
    }
    const uint8_t *data = tr.p;                                  // the table is padded to a byte boundary
    if ((uint64_t) n_bits > 8ull * (uint64_t) (src + n_src - data)) throw std::runtime_error("EXR/PIZ: truncated Huffman data");
    // canonical codes: within a length in symbol order; the longest codes take the numerically smallest values
    uint64_t count[kMaxCodeLen + 1] = {0}, base[kMaxCodeLen + 1] = {0};
    for (int s = 0; s < kEncSize; ++s) ++count[len[s]];
    uint64_t c = 0;
    for (int l = kMaxCodeLen; l >= 1; --l) { const uint64_t nc = (c + count[l]) >> 1; base[l] = c; c = nc; }
    std::vector<uint32_t> first((size_t) kMaxCodeLen + 2, 0), syms;
    syms.reserve(65537);
    {
        uint32_t k = 0;
        for (int l = 1; l <= kMaxCodeLen; ++l) {
            first[l] = k;
            k += (uint32_t) count[l];
        }
        first[kMaxCodeLen + 1] = k;
        syms.resize(k);
        std::vector<uint32_t> fill(first.begin(), first.end());
        for (int s = 0; s < kEncSize; ++s) if (len[s]) syms[fill[len[s]]++] = (uint32_t) s;
    }
    const uint32_t rlc = iM;                                      // the run-length escape symbol
    BitReader br(data, data + (n_bits + 7) / 8);
    uint64_t left = n_bits;
    size_t o = 0;
    while (left > 0 && o < n_out) {
        uint64_t code = 0;
        int l = 0;
        uint32_t sym = 0;
        bool found = false;
        while (l < kMaxCodeLen && left > 0) {
            code = (code << 1) | br.get(1);
            ++l; --left;
            if (count[l] && code >= base[l] && code - base[l] < count[l]) { sym = syms[first[l] + (uint32_t) (code - base[l])]; found = true; break; }
        }
        if (!found) {
            if (left == 0) break;                                 // trailing pad bits
            throw std::runtime_error("EXR/PIZ: invalid Huffman code");
        }
        if (sym == rlc) {
            if (left < 8 || o == 0) throw std::runtime_error("EXR/PIZ: bad run");
            uint32_t run = br.get(8);
            left -= 8;
            if (o + run > n_out) throw std::runtime_error("EXR/PIZ: run past the end of the block");
            const uint16_t v = out[o - 1];
            while (run--) out[o++] = v;
        } else {
            out[o++] = (uint16_t) sym;
        }
    }
    if (o != n_out) throw std::runtime_error("EXR/PIZ: Huffman data ended early");
}

// inverse of the two-point transform: (average, difference) -> (a, b); 14-bit data fits signed shorts, 16-bit data wraps
inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int ls = (int16_t) l, hs = (int16_t) h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t) (int16_t) ai;
    b = (uint16_t) (int16_t) (ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - (1 << 15)) & 0xffff;
    b = (uint16_t) bb;
    a = (uint16_t) aa;
}
// in: nx x ny samples, ox / oy = strides between samples / rows; mx = largest value present
void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    while (p >= 1) {
        uint16_t *py = in;
        uint16_t *const ey = in + (ptrdiff_t) oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t) oy * p, oy2 = (ptrdiff_t) oy * p2, ox1 = (ptrdiff_t) ox * p, ox2 = (ptrdiff_t) ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) {
                    wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11);
                    wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11);
                } else {
                    wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11);
                    wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11);
                }
            }
            if (nx & p) {                                         // a column without a partner
                uint16_t *p10 = px + oy1;
                if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {                                             // a row without a partner
            uint16_t *px = py;
            uint16_t *const ex = py + (ptrdiff_t) ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

void piz_decode_chunk(const uint8_t *src, size_t n_src, int nx, int ny, const std::vector<int> &words_per_sample, uint16_t *out) {
    size_t total = 0;
    for (int w : words_per_sample) total += (size_t) nx * ny * w;
    if (n_src < 4) throw std::runtime_error("EXR/PIZ: truncated chunk");
    const uint16_t min_nz = (uint16_t) (src[0] | (src[1] << 8)), max_nz = (uint16_t) (src[2] | (src[3] << 8));
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (min_nz <= max_nz) {
        const size_t nb = (size_t) max_nz - min_nz + 1;
        if (max_nz >= 8192 || pos + nb > n_src) throw std::runtime_error("EXR/PIZ: bad bitmap");
        std::memcpy(bitmap.data() + min_nz, src + pos, nb);
        pos += nb;
    }
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int i = 0; i < 65536; ++i)
        if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t) i;
    const uint16_t max_value = (uint16_t) (k - 1);
    if (pos + 4 > n_src) throw std::runtime_error("EXR/PIZ: truncated chunk");
    const uint32_t huf_len = rd32(src + pos);
    pos += 4;
    if (pos + huf_len > n_src) throw std::runtime_error("EXR/PIZ: truncated chunk");
    std::vector<uint16_t> tmp(total);
    huf_uncompress(src + pos, huf_len, tmp.data(), total);
    // channels are stored one after the other, each as ny rows of nx samples of `w` interleaved 16-bit words
    std::vector<size_t> start;
    size_t q = 0;
    for (int w : words_per_sample) {
        start.push_back(q);
        for (int j = 0; j < w; ++j) wav2_decode(tmp.data() + q + j, nx, w, ny, nx * w, max_value);
        q += (size_t) nx * ny * w;
    }
    for (size_t i = 0; i < total; ++i) tmp[i] = lut[tmp[i]];
    // back to the file's scanline layout: for every line, every channel's nx samples
    size_t o = 0;
    for (int y = 0; y < ny; ++y)
        for (size_t c = 0; c < words_per_sample.size(); ++c) {
            const size_t n = (size_t) nx * words_per_sample[c];
            std::memcpy(out + o, tmp.data() + start[c] + (size_t) y * n, n * sizeof(uint16_t));
            o += n;
        }
}

}  // namespace psdr_host
