// rf_jpeg.cpp -- JPEG (baseline, extended sequential and progressive Huffman, 8 bit) -> RGBA8.
//
// The reference decodes textures with stb_image (nothings/stb @ beebb24b, un-vendored third party
// code: external/CMakeLists.txt:36-38, call site src/common/texture.cpp:12-31 with req_comp = 4).
// Entropy decoding and dequantisation are fixed by the JPEG standard; what differs between decoders
// is everything after the coefficients, so those stages restate stb_image's published integer
// algorithm: its 12-bit fixed-point Loeffler-style IDCT (columns to 10 bits, rows to 17, +128 level
// shift folded into the rounding constant), its chroma up-sampling filters (h2v1: 3:1 taps along the
// row; h2v2: 3:1 vertically then 3:1 horizontally with /16 rounding; v2: 3:1 between rows; otherwise
// replication) and its 20-bit fixed-point YCbCr -> RGB.  PARITY UNPINNED against stb_image itself
// (absent from the image and the mount); tests compare with Pillow's libjpeg-turbo within a few
// grey levels and pin the stages with known answers.
#include "rf_jpeg.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

namespace rf
{
namespace
{
[[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("JPEG: ") + what); }

constexpr uint8_t kDezigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                        6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                        39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                        // so that a corrupt run cannot index outside the block
                                        63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman
{
    // canonical code: for length L (1..16) codes first[L] .. first[L] + count[L] - 1 map to values[offset[L] ..]
    uint16_t count[17] = {};
    int32_t  first[17] = {};
    int32_t  offset[17] = {};
    uint8_t  values[256] = {};
    bool     defined = false;
    // 9-bit prefix table: (length << 8) | value, 0 = longer code
    uint16_t fast[512] = {};

    void build()
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len)
        {
            first[len] = code;
            offset[len] = k;
            if (count[len] && code + count[len] - 1 >= (1 << len)) fail("bad code lengths");
            code = (code + count[len]) << 1;
            k += count[len];
        }
        std::memset(fast, 0, sizeof fast);
        for (int len = 1; len <= 9; ++len)
            for (int i = 0; i < count[len]; ++i)
            {
                const int c = (first[len] + i) << (9 - len);
                for (int fill = 0; fill < (1 << (9 - len)); ++fill) fast[c + fill] = static_cast<uint16_t>((len << 8) | values[offset[len] + i]);
            }
        defined = true;
    }
};

struct Component
{
    int                  id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int                  dcPred = 0;
    int                  x = 0, y = 0, w2 = 0, h2 = 0;
    int                  coeffW = 0, coeffH = 0;
    std::vector<uint8_t> data;
    std::vector<int16_t> coeff; // progressive
};

struct Decoder
{
    const uint8_t* p;
    const uint8_t* end;
    // bit reader (MSB first); after a marker is met the stream reads as zeros
    uint32_t codeBuffer = 0;
    int      codeBits = 0;
    uint8_t  marker = 0xFF; // 0xFF = none pending
    bool     noMore = false;

    int       width = 0, height = 0, numComponents = 0;
    bool      progressive = false, rgb = false;
    int       hMax = 1, vMax = 1, mcuW = 0, mcuH = 0, mcuX = 0, mcuY = 0;
    Component comp[4];
    Huffman   dc[4], ac[4];
    uint16_t  dequant[4][64] = {};
    int       restartInterval = 0, todo = 0;
    int       scanN = 0, order[4] = {};
    int       specStart = 0, specEnd = 0, succHigh = 0, succLow = 0, eobRun = 0;
    int       adobeTransform = -1;
    bool      jfif = false;

    int get8()
    {
        if (p >= end) return 0;
        return *p++;
    }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void grow()
    {
        do
        {
            int b = noMore ? 0 : get8();
            if (b == 0xFF)
            {
                int c = get8();
                while (c == 0xFF) c = get8(); // fill bytes
                if (c != 0)
                {
                    marker = static_cast<uint8_t>(c);
                    noMore = true;
                    return;
                }
            }
            codeBuffer |= static_cast<uint32_t>(b) << (24 - codeBits);
            codeBits += 8;
        } while (codeBits <= 24);
    }

    int decode(const Huffman& h)
    {
        if (codeBits < 16) grow();
        const uint16_t f = h.fast[codeBuffer >> 23];
        if (f)
        {
            const int len = f >> 8;
            if (len > codeBits) fail("bad huffman code");
            codeBuffer <<= len;
            codeBits -= len;
            return f & 255;
        }
        const uint32_t top = codeBuffer >> 16;
        for (int len = 10; len <= 16; ++len)
        {
            const int code = static_cast<int>(top >> (16 - len));
            if (h.count[len] && code >= h.first[len] && code < h.first[len] + h.count[len])
            {
                if (len > codeBits) fail("bad huffman code");
                codeBuffer <<= len;
                codeBits -= len;
                return h.values[h.offset[len] + code - h.first[len]];
            }
        }
        fail("bad huffman code");
    }

    int getBits(int n)
    {
        if (n == 0) return 0;
        if (codeBits < n) grow();
        const uint32_t k = codeBuffer >> (32 - n);
        codeBuffer <<= n;
        codeBits -= n;
        return static_cast<int>(k);
    }
    int getBit() { return getBits(1); }
    // RECEIVE + EXTEND (ITU T.81 F.2.2.1)
    int extendReceive(int n)
    {
        if (n == 0) return 0;
        const int v = getBits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    void resetScan()
    {
        codeBuffer = 0;
        codeBits = 0;
        noMore = false;
        marker = 0xFF;
        for (Component& c : comp) c.dcPred = 0;
        todo = restartInterval ? restartInterval : 0x7fffffff;
        eobRun = 0;
    }
};

// ---- IDCT (stb_image's integer algorithm) -----------------------------------------------------
constexpr int f2f(float x) { return static_cast<int>(x * 4096 + 0.5); }
constexpr int fsh(int x) { return x * 4096; }
inline uint8_t clamp8(int x)
{
    if (static_cast<unsigned>(x) > 255) return x < 0 ? 0 : 255;
    return static_cast<uint8_t>(x);
}

#define RF_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                                  \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                                        \
    p2 = s2;                                                                                                       \
    p3 = s6;                                                                                                       \
    p1 = (p2 + p3) * f2f(0.5411961f);                                                                              \
    t2 = p1 + p3 * f2f(-1.847759065f);                                                                             \
    t3 = p1 + p2 * f2f(0.765366865f);                                                                              \
    p2 = s0;                                                                                                       \
    p3 = s4;                                                                                                       \
    t0 = fsh(p2 + p3);                                                                                             \
    t1 = fsh(p2 - p3);                                                                                             \
    x0 = t0 + t3;                                                                                                  \
    x3 = t0 - t3;                                                                                                  \
    x1 = t1 + t2;                                                                                                  \
    x2 = t1 - t2;                                                                                                  \
    t0 = s7;                                                                                                       \
    t1 = s5;                                                                                                       \
    t2 = s3;                                                                                                       \
    t3 = s1;                                                                                                       \
    p3 = t0 + t2;                                                                                                  \
    p4 = t1 + t3;                                                                                                  \
    p1 = t0 + t3;                                                                                                  \
    p2 = t1 + t2;                                                                                                  \
    p5 = (p3 + p4) * f2f(1.175875602f);                                                                            \
    t0 = t0 * f2f(0.298631336f);                                                                                   \
    t1 = t1 * f2f(2.053119869f);                                                                                   \
    t2 = t2 * f2f(3.072711026f);                                                                                   \
    t3 = t3 * f2f(1.501321110f);                                                                                   \
    p1 = p5 + p1 * f2f(-0.899976223f);                                                                             \
    p2 = p5 + p2 * f2f(-2.562915447f);                                                                             \
    p3 = p3 * f2f(-1.961570560f);                                                                                  \
    p4 = p4 * f2f(-0.390180644f);                                                                                  \
    t3 += p1 + p4;                                                                                                 \
    t2 += p2 + p3;                                                                                                 \
    t1 += p2 + p4;                                                                                                 \
    t0 += p1 + p3;

void idctBlock(uint8_t* out, int outStride, const int16_t data[64])
{
    int            val[64];
    int*           v = val;
    const int16_t* d = data;
    for (int i = 0; i < 8; ++i, ++d, ++v)
    {
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0)
        {
            const int dcterm = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        }
        else
        {
            RF_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512;
            x1 += 512;
            x2 += 512;
            x3 += 512;
            v[0] = (x0 + t3) >> 10;
            v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10;
            v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10;
            v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10;
            v[32] = (x3 - t0) >> 10;
        }
    }
    v = val;
    uint8_t* o = out;
    for (int i = 0; i < 8; ++i, v += 8, o += outStride)
    {
        RF_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17);
        x1 += 65536 + (128 << 17);
        x2 += 65536 + (128 << 17);
        x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17);
        o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17);
        o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17);
        o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17);
        o[4] = clamp8((x3 - t0) >> 17);
    }
}

// ---- entropy decoding -------------------------------------------------------------------------
void decodeBlock(Decoder& z, int16_t data[64], Component& c)
{
    const Huffman &hdc = z.dc[c.hd], &hac = z.ac[c.ha];
    const uint16_t* dq = z.dequant[c.tq];
    std::memset(data, 0, 64 * sizeof(int16_t));
    const int t = z.decode(hdc);
    if (t > 16) fail("bad DC size");
    const int diff = t ? z.extendReceive(t) : 0;
    const int dcv = c.dcPred + diff;
    c.dcPred = dcv;
    data[0] = static_cast<int16_t>(dcv * dq[0]);
    int k = 1;
    do
    {
        const int rs = z.decode(hac), s = rs & 15, r = rs >> 4;
        if (s == 0)
        {
            if (rs != 0xF0) break; // end of block
            k += 16;
        }
        else
        {
            k += r;
            const int zig = kDezigzag[k++];
            data[zig] = static_cast<int16_t>(z.extendReceive(s) * dq[zig]);
        }
    } while (k < 64);
}

void decodeBlockProgDc(Decoder& z, int16_t data[64], Component& c)
{
    if (z.specEnd != 0) fail("cannot merge DC and AC");
    if (z.succHigh == 0)
    {
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int t = z.decode(z.dc[c.hd]);
        if (t > 16) fail("bad DC size");
        const int diff = t ? z.extendReceive(t) : 0;
        const int dcv = c.dcPred + diff;
        c.dcPred = dcv;
        data[0] = static_cast<int16_t>(dcv * (1 << z.succLow));
    }
    else if (z.getBit()) data[0] = static_cast<int16_t>(data[0] + (1 << z.succLow));
}

void decodeBlockProgAc(Decoder& z, int16_t data[64], const Huffman& hac)
{
    if (z.specStart == 0) fail("cannot merge DC and AC");
    if (z.succHigh == 0)
    {
        const int shift = z.succLow;
        if (z.eobRun)
        {
            --z.eobRun;
            return;
        }
        int k = z.specStart;
        do
        {
            const int rs = z.decode(hac), s = rs & 15, r = rs >> 4;
            if (s == 0)
            {
                if (r < 15)
                {
                    z.eobRun = 1 << r;
                    if (r) z.eobRun += z.getBits(r);
                    --z.eobRun;
                    break;
                }
                k += 16;
            }
            else
            {
                k += r;
                const int zig = kDezigzag[k++];
                data[zig] = static_cast<int16_t>(z.extendReceive(s) * (1 << shift));
            }
        } while (k <= z.specEnd);
        return;
    }
    // refinement scan
    const int16_t bit = static_cast<int16_t>(1 << z.succLow);
    auto          refine = [&](int16_t* p) {
        if (z.getBit() && (*p & bit) == 0) *p = static_cast<int16_t>(*p > 0 ? *p + bit : *p - bit);
    };
    if (z.eobRun)
    {
        --z.eobRun;
        for (int k = z.specStart; k <= z.specEnd; ++k)
        {
            int16_t* p = &data[kDezigzag[k]];
            if (*p != 0) refine(p);
        }
        return;
    }
    int k = z.specStart;
    do
    {
        const int rs = z.decode(hac);
        int       s = rs & 15, r = rs >> 4;
        if (s == 0)
        {
            if (r < 15)
            {
                z.eobRun = (1 << r) - 1;
                if (r) z.eobRun += z.getBits(r);
                r = 64; // force end of block
            }
            // r == 15: a run of 16 zero coefficients
        }
        else
        {
            if (s != 1) fail("bad huffman code");
            s = z.getBit() ? bit : -bit;
        }
        while (k <= z.specEnd)
        {
            int16_t* p = &data[kDezigzag[k++]];
            if (*p != 0) refine(p);
            else
            {
                if (r == 0)
                {
                    *p = static_cast<int16_t>(s);
                    break;
                }
                --r;
            }
        }
    } while (k <= z.specEnd);
}

bool isRestart(uint8_t m) { return m >= 0xD0 && m <= 0xD7; }

// returns false when the entropy-coded segment ended early (stb_image keeps what it has)
bool afterMcu(Decoder& z)
{
    if (--z.todo <= 0)
    {
        if (z.codeBits < 24) z.grow();
        if (!isRestart(z.marker)) return false;
        z.resetScan();
    }
    return true;
}

void decodeScan(Decoder& z)
{
    z.resetScan();
    int16_t block[64];
    if (!z.progressive)
    {
        if (z.scanN == 1)
        {
            Component& c = z.comp[z.order[0]];
            const int  w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i)
                {
                    decodeBlock(z, block, c);
                    idctBlock(c.data.data() + static_cast<size_t>(c.w2) * j * 8 + i * 8, c.w2, block);
                    if (!afterMcu(z)) return;
                }
            return;
        }
        for (int j = 0; j < z.mcuY; ++j)
            for (int i = 0; i < z.mcuX; ++i)
            {
                for (int k = 0; k < z.scanN; ++k)
                {
                    Component& c = z.comp[z.order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x)
                        {
                            const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                            decodeBlock(z, block, c);
                            idctBlock(c.data.data() + static_cast<size_t>(c.w2) * y2 + x2, c.w2, block);
                        }
                }
                if (!afterMcu(z)) return;
            }
        return;
    }
    if (z.scanN == 1)
    {
        Component& c = z.comp[z.order[0]];
        const int  w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i)
            {
                int16_t* data = c.coeff.data() + 64 * (static_cast<size_t>(i) + static_cast<size_t>(j) * c.coeffW);
                if (z.specStart == 0) decodeBlockProgDc(z, data, c);
                else decodeBlockProgAc(z, data, z.ac[c.ha]);
                if (!afterMcu(z)) return;
            }
        return;
    }
    for (int j = 0; j < z.mcuY; ++j)
        for (int i = 0; i < z.mcuX; ++i)
        {
            for (int k = 0; k < z.scanN; ++k)
            {
                Component& c = z.comp[z.order[k]];
                for (int y = 0; y < c.v; ++y)
                    for (int x = 0; x < c.h; ++x)
                    {
                        const int x2 = i * c.h + x, y2 = j * c.v + y;
                        decodeBlockProgDc(z, c.coeff.data() + 64 * (static_cast<size_t>(x2) + static_cast<size_t>(y2) * c.coeffW), c);
                    }
            }
            if (!afterMcu(z)) return;
        }
}

void finishProgressive(Decoder& z)
{
    for (int n = 0; n < z.numComponents; ++n)
    {
        Component& c = z.comp[n];
        const int  w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
        for (int j = 0; j < h; ++j)
            for (int i = 0; i < w; ++i)
            {
                int16_t* data = c.coeff.data() + 64 * (static_cast<size_t>(i) + static_cast<size_t>(j) * c.coeffW);
                for (int k = 0; k < 64; ++k) data[k] = static_cast<int16_t>(data[k] * z.dequant[c.tq][k]);
                idctBlock(c.data.data() + static_cast<size_t>(c.w2) * j * 8 + i * 8, c.w2, data);
            }
    }
}

// ---- markers ----------------------------------------------------------------------------------
void processMarker(Decoder& z, int m)
{
    switch (m)
    {
    case 0xDD: // DRI
        if (z.get16() != 4) fail("bad DRI length");
        z.restartInterval = z.get16();
        return;
    case 0xDB: // DQT
    {
        int L = z.get16() - 2;
        while (L > 0)
        {
            const int q = z.get8(), prec = q >> 4, t = q & 15;
            if ((prec != 0 && prec != 1) || t > 3) fail("bad DQT");
            for (int i = 0; i < 64; ++i) z.dequant[t][kDezigzag[i]] = static_cast<uint16_t>(prec ? z.get16() : z.get8());
            L -= prec ? 129 : 65;
        }
        if (L != 0) fail("bad DQT length");
        return;
    }
    case 0xC4: // DHT
    {
        int L = z.get16() - 2;
        while (L > 0)
        {
            const int q = z.get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) fail("bad DHT header");
            Huffman& h = tc == 0 ? z.dc[th] : z.ac[th];
            int      n = 0;
            h.count[0] = 0;
            for (int i = 1; i <= 16; ++i)
            {
                h.count[i] = static_cast<uint16_t>(z.get8());
                n += h.count[i];
            }
            if (n > 256) fail("bad DHT header");
            L -= 17;
            for (int i = 0; i < n; ++i) h.values[i] = static_cast<uint8_t>(z.get8());
            h.build();
            L -= n;
        }
        if (L != 0) fail("bad DHT length");
        return;
    }
    default: break;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE)
    {
        int L = z.get16();
        if (L < 2) fail(m == 0xFE ? "bad COM length" : "bad APP length");
        L -= 2;
        if (m == 0xE0 && L >= 5)
        {
            static const char tag[5] = {'J', 'F', 'I', 'F', '\0'};
            bool              ok = true;
            for (int i = 0; i < 5; ++i)
                if (z.get8() != tag[i]) ok = false;
            L -= 5;
            if (ok) z.jfif = true;
        }
        else if (m == 0xEE && L >= 12)
        {
            static const char tag[6] = {'A', 'd', 'o', 'b', 'e', '\0'};
            bool              ok = true;
            for (int i = 0; i < 6; ++i)
                if (z.get8() != tag[i]) ok = false;
            L -= 6;
            if (ok)
            {
                z.get8();
                z.get16();
                z.get16();
                z.adobeTransform = z.get8();
                L -= 6;
            }
        }
        z.p = std::min(z.p + L, z.end);
        return;
    }
    fail("unknown marker");
}

void processFrameHeader(Decoder& z)
{
    const int Lf = z.get16();
    if (Lf < 11) fail("bad SOF length");
    if (z.get8() != 8) fail("only 8-bit JPEG is supported");
    z.height = z.get16();
    z.width = z.get16();
    if (z.height == 0 || z.width == 0) fail("zero-sized image");
    const int n = z.get8();
    if (n != 1 && n != 3) fail(n == 4 ? "CMYK JPEG is not supported" : "bad component count");
    z.numComponents = n;
    if (Lf != 8 + 3 * n) fail("bad SOF length");
    z.rgb = false;
    static const char rgbIds[3] = {'R', 'G', 'B'};
    int               rgbMatches = 0;
    for (int i = 0; i < n; ++i)
    {
        Component& c = z.comp[i];
        c.id = z.get8();
        if (n == 3 && c.id == rgbIds[i]) ++rgbMatches;
        const int q = z.get8();
        c.h = q >> 4;
        c.v = q & 15;
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) fail("bad sampling factors");
        c.tq = z.get8();
        if (c.tq > 3) fail("bad quantisation table index");
    }
    z.rgb = rgbMatches == 3;
    z.hMax = z.vMax = 1;
    for (int i = 0; i < n; ++i)
    {
        z.hMax = std::max(z.hMax, z.comp[i].h);
        z.vMax = std::max(z.vMax, z.comp[i].v);
    }
    for (int i = 0; i < n; ++i)
        if (z.hMax % z.comp[i].h != 0 || z.vMax % z.comp[i].v != 0) fail("bad sampling factors");
    z.mcuW = z.hMax * 8;
    z.mcuH = z.vMax * 8;
    z.mcuX = (z.width + z.mcuW - 1) / z.mcuW;
    z.mcuY = (z.height + z.mcuH - 1) / z.mcuH;
    if (static_cast<uint64_t>(z.width) * z.height > (1ull << 28)) fail("image too large");
    for (int i = 0; i < n; ++i)
    {
        Component& c = z.comp[i];
        c.x = (z.width * c.h + z.hMax - 1) / z.hMax;
        c.y = (z.height * c.v + z.vMax - 1) / z.vMax;
        c.w2 = z.mcuX * c.h * 8;
        c.h2 = z.mcuY * c.v * 8;
        c.data.assign(static_cast<size_t>(c.w2) * c.h2, 0);
        if (z.progressive)
        {
            c.coeffW = c.w2 / 8;
            c.coeffH = c.h2 / 8;
            c.coeff.assign(static_cast<size_t>(c.w2) * c.h2, 0);
        }
    }
}

void processScanHeader(Decoder& z)
{
    const int Ls = z.get16();
    z.scanN = z.get8();
    if (z.scanN < 1 || z.scanN > 4 || z.scanN > z.numComponents) fail("bad SOS component count");
    if (Ls != 6 + 2 * z.scanN) fail("bad SOS length");
    for (int i = 0; i < z.scanN; ++i)
    {
        const int id = z.get8(), q = z.get8();
        int       which = 0;
        for (; which < z.numComponents; ++which)
            if (z.comp[which].id == id) break;
        if (which == z.numComponents) fail("SOS names an unknown component");
        z.comp[which].hd = q >> 4;
        z.comp[which].ha = q & 15;
        if (z.comp[which].hd > 3 || z.comp[which].ha > 3) fail("bad huffman table index");
        z.order[i] = which;
    }
    z.specStart = z.get8();
    z.specEnd = z.get8();
    const int aa = z.get8();
    z.succHigh = aa >> 4;
    z.succLow = aa & 15;
    if (z.progressive)
    {
        if (z.specStart > 63 || z.specEnd > 63 || z.specStart > z.specEnd || z.succHigh > 13 || z.succLow > 13) fail("bad SOS");
    }
    else
    {
        if (z.specStart != 0) fail("bad SOS");
        if (z.succHigh != 0 || z.succLow != 0) fail("bad SOS");
        z.specEnd = 63;
    }
}

int nextMarker(Decoder& z)
{
    if (z.marker != 0xFF)
    {
        const int m = z.marker;
        z.marker = 0xFF;
        return m;
    }
    int x = z.get8();
    if (x != 0xFF) return 0xFF; // not a marker
    while (x == 0xFF) x = z.get8();
    return x;
}

// ---- up-sampling (stb_image's filters) ---------------------------------------------------------
inline uint8_t div4(int x) { return static_cast<uint8_t>(x >> 2); }
inline uint8_t div16(int x) { return static_cast<uint8_t>(x >> 4); }

const uint8_t* resampleRow1(uint8_t*, const uint8_t* nearRow, const uint8_t*, int, int) { return nearRow; }
const uint8_t* resampleRowV2(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int)
{
    for (int i = 0; i < w; ++i) out[i] = div4(3 * nearRow[i] + farRow[i] + 2);
    return out;
}
const uint8_t* resampleRowH2(uint8_t* out, const uint8_t* in, const uint8_t*, int w, int)
{
    if (w == 1)
    {
        out[0] = out[1] = in[0];
        return out;
    }
    out[0] = in[0];
    out[1] = div4(in[0] * 3 + in[1] + 2);
    int i;
    for (i = 1; i < w - 1; ++i)
    {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = div4(n + in[i - 1]);
        out[i * 2 + 1] = div4(n + in[i + 1]);
    }
    out[i * 2 + 0] = div4(in[w - 2] * 3 + in[w - 1] + 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t* resampleRowHV2(uint8_t* out, const uint8_t* nearRow, const uint8_t* farRow, int w, int)
{
    if (w == 1)
    {
        out[0] = out[1] = div4(3 * nearRow[0] + farRow[0] + 2);
        return out;
    }
    int t1 = 3 * nearRow[0] + farRow[0];
    out[0] = div4(t1 + 2);
    for (int i = 1; i < w; ++i)
    {
        const int t0 = t1;
        t1 = 3 * nearRow[i] + farRow[i];
        out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
        out[i * 2] = div16(3 * t1 + t0 + 8);
    }
    out[w * 2 - 1] = div4(t1 + 2);
    return out;
}
const uint8_t* resampleRowGeneric(uint8_t* out, const uint8_t* nearRow, const uint8_t*, int w, int hs)
{
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
    return out;
}

// ---- colour (stb_image's fixed point) ----------------------------------------------------------
constexpr int float2fixed(float x) { return static_cast<int>(x * 4096.0f + 0.5f) << 8; }

void ycbcrToRgbRow(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count)
{
    for (int i = 0; i < count; ++i)
    {
        const int yFixed = (y[i] << 20) + (1 << 19); // rounding
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int       r = yFixed + cr * float2fixed(1.40200f);
        int       g = yFixed + (cr * -float2fixed(0.71414f)) + ((cb * -float2fixed(0.34414f)) & 0xffff0000);
        int       b = yFixed + cb * float2fixed(1.77200f);
        r >>= 20;
        g >>= 20;
        b >>= 20;
        out[0] = clamp8(r);
        out[1] = clamp8(g);
        out[2] = clamp8(b);
        out[3] = 255;
        out += 4;
    }
}
} // namespace

bool looksLikeJpeg(std::span<const uint8_t> data) { return data.size() >= 3 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF; }

Rgba8Image decodeJpeg(std::span<const uint8_t> bytes)
{
    Decoder z;
    z.p = bytes.data();
    z.end = bytes.data() + bytes.size();
    if (nextMarker(z) != 0xD8) fail("no SOI");
    // header
    int m = nextMarker(z);
    while (!(m == 0xC0 || m == 0xC1 || m == 0xC2))
    {
        if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) fail("unsupported JPEG process (lossless, hierarchical or arithmetic)");
        processMarker(z, m);
        m = nextMarker(z);
        while (m == 0xFF)
        {
            if (z.p >= z.end) fail("no SOF");
            m = nextMarker(z);
        }
    }
    z.progressive = m == 0xC2;
    processFrameHeader(z);

    // scans
    m = nextMarker(z);
    while (m != 0xD9)
    {
        if (m == 0xDA)
        {
            processScanHeader(z);
            decodeScan(z);
            if (z.marker == 0xFF)
            {
                // look for the next marker past any junk after the entropy-coded data
                while (z.p < z.end)
                {
                    const int x = z.get8();
                    if (x == 0xFF)
                    {
                        const int y = z.p < z.end ? *z.p : 0;
                        if (y != 0x00 && y != 0xFF)
                        {
                            z.marker = static_cast<uint8_t>(z.get8());
                            break;
                        }
                    }
                }
                if (z.marker == 0xFF) break; // ran off the end: treat as EOI
            }
        }
        else if (m == 0xDC) // DNL
        {
            const int Ld = z.get16();
            const int NL = z.get16();
            if (Ld != 4) fail("bad DNL length");
            if (NL != z.height) fail("bad DNL height");
        }
        else
        {
            if (m == 0xFF && z.p >= z.end) break;
            processMarker(z, m);
        }
        m = nextMarker(z);
    }
    if (z.progressive) finishProgressive(z);

    // colour space: three components are YCbCr unless the ids spell RGB or an Adobe marker says "no transform"
    const bool isRgb = z.numComponents == 3 && (z.rgb || (z.adobeTransform == 0 && !z.jfif));

    // resample + convert, row by row
    struct Resample
    {
        const uint8_t* (*fn)(uint8_t*, const uint8_t*, const uint8_t*, int, int);
        const uint8_t *line0, *line1;
        int            hs, vs, wLores, ystep, ypos;
    } rs[3];
    std::vector<uint8_t> lineBuf[3];
    for (int k = 0; k < z.numComponents; ++k)
    {
        Resample& r = rs[k];
        lineBuf[k].assign(static_cast<size_t>(z.width) + 3 + 4 * 8, 0);
        r.hs = z.hMax / z.comp[k].h;
        r.vs = z.vMax / z.comp[k].v;
        r.ystep = r.vs >> 1;
        r.wLores = (z.width + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = z.comp[k].data.data();
        if (r.hs == 1 && r.vs == 1) r.fn = resampleRow1;
        else if (r.hs == 1 && r.vs == 2) r.fn = resampleRowV2;
        else if (r.hs == 2 && r.vs == 1) r.fn = resampleRowH2;
        else if (r.hs == 2 && r.vs == 2) r.fn = resampleRowHV2;
        else r.fn = resampleRowGeneric;
    }
    Rgba8Image img;
    img.width = static_cast<uint32_t>(z.width);
    img.height = static_cast<uint32_t>(z.height);
    img.rgba.resize(static_cast<size_t>(z.width) * z.height * 4);
    for (int j = 0; j < z.height; ++j)
    {
        const uint8_t* rows[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < z.numComponents; ++k)
        {
            Resample&  r = rs[k];
            const bool yBot = r.ystep >= (r.vs >> 1);
            rows[k] = r.fn(lineBuf[k].data(), yBot ? r.line1 : r.line0, yBot ? r.line0 : r.line1, r.wLores, r.hs);
            if (++r.ystep >= r.vs)
            {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
            }
        }
        uint8_t* out = img.rgba.data() + static_cast<size_t>(j) * z.width * 4;
        if (z.numComponents == 3)
        {
            if (isRgb)
                for (int i = 0; i < z.width; ++i)
                {
                    out[4 * i] = rows[0][i];
                    out[4 * i + 1] = rows[1][i];
                    out[4 * i + 2] = rows[2][i];
                    out[4 * i + 3] = 255;
                }
            else ycbcrToRgbRow(out, rows[0], rows[1], rows[2], z.width);
        }
        else
            for (int i = 0; i < z.width; ++i)
            {
                out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = rows[0][i];
                out[4 * i + 3] = 255;
            }
    }
    return img;
}
} // namespace rf
