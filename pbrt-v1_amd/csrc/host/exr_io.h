// exr_io.h -- minimal OpenEXR (RGBA half, scanline) writer / reader for the film output.
//
// The reference writes its image through OpenEXR's RgbaOutputFile (core/exrio.cpp:75-96): RGBA *half* pixels, a
// dataWindow equal to the rendered crop placed inside a displayWindow of the full resolution.  OpenEXR is a
// third-party library that is not vendored in the reference tree (exrinstall.txt: "latest") and is not installed
// here, so this is an independent writer of the published OpenEXR 2 single-part scanline layout: magic 20000630,
// version 2, attributes channels/compression/dataWindow/displayWindow/lineOrder/pixelAspectRatio/
// screenWindowCenter/screenWindowWidth, an offset table, one chunk per scanline with the channels in alphabetical
// order (A, B, G, R).  Compression is NO_COMPRESSION (RgbaOutputFile's default is PIZ; the pixel values -- the only
// thing parity is defined on -- are identical).  float -> half is round-to-nearest-even with overflow to infinity,
// as ImfHalf / half.h do.  The reader accepts exactly what the writer produces (enough for tile merging, the role of
// tools/exrassemble.cpp:42-67, and for tests).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace pbrthip {

inline uint16_t float_to_half(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = int32_t((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return uint16_t(sign | 0x7c00u | (m ? (0x200u | (m >> 13)) : 0u));   // inf / nan
    if (e >= 31) return uint16_t(sign | 0x7c00u);                                                        // overflow -> inf
    if (e <= 0) {                                                                                        // half denormal / zero
        if (e < -10) return uint16_t(sign);
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half_way = 1u << (shift - 1);
        if (rem > half_way || (rem == half_way && (h & 1))) ++h;
        return uint16_t(sign | h);
    }
    uint32_t h = (uint32_t(e) << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;                                                // may carry into the exponent: still correct
    return uint16_t(sign | h);
}
inline float half_to_float(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { e = 1; while (!(m & 0x400u)) { m <<= 1; --e; } m &= 0x3ffu; x = sign | ((e + 127 - 15) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 127 - 15) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4); return f;
}

struct ExrImage {
    int xRes = 0, yRes = 0, totalXRes = 0, totalYRes = 0, xOffset = 0, yOffset = 0;
    std::vector<float> rgb, alpha;      // [yRes][xRes][3], [yRes][xRes] (after half quantisation)
};

// same argument list as the reference's WriteRGBAImage (core/pbrt.h:236-238)
inline bool WriteRGBAImage(const std::string &name, const float *pixels, const float *alpha, int xRes, int yRes,
                           int totalXRes, int totalYRes, int xOffset, int yOffset) {
    FILE *f = std::fopen(name.c_str(), "wb");
    if (!f) return false;
    std::vector<unsigned char> hdr;
    auto put = [&](const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; hdr.insert(hdr.end(), c, c + n); };
    auto puts0 = [&](const char *s) { put(s, std::strlen(s) + 1); };
    auto puti = [&](int32_t v) { put(&v, 4); };
    auto putf = [&](float v) { put(&v, 4); };
    const unsigned char magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    put(magic, 8);
    puts0("channels"); puts0("chlist"); puti(4 * 18 + 1);
    for (const char *ch : {"A", "B", "G", "R"}) { puts0(ch); puti(1 /*HALF*/); const unsigned char lin[4] = {0, 0, 0, 0}; put(lin, 4); puti(1); puti(1); }
    hdr.push_back(0);
    puts0("compression"); puts0("compression"); puti(1); hdr.push_back(0);
    puts0("dataWindow"); puts0("box2i"); puti(16); puti(xOffset); puti(yOffset); puti(xOffset + xRes - 1); puti(yOffset + yRes - 1);
    puts0("displayWindow"); puts0("box2i"); puti(16); puti(0); puti(0); puti(totalXRes - 1); puti(totalYRes - 1);
    puts0("lineOrder"); puts0("lineOrder"); puti(1); hdr.push_back(0);
    puts0("pixelAspectRatio"); puts0("float"); puti(4); putf(1.f);
    puts0("screenWindowCenter"); puts0("v2f"); puti(8); putf(0.f); putf(0.f);
    puts0("screenWindowWidth"); puts0("float"); puti(4); putf(1.f);
    hdr.push_back(0);
    const uint64_t row_bytes = uint64_t(xRes) * 4 * 2, chunk = 8 + row_bytes;
    uint64_t off = hdr.size() + uint64_t(yRes) * 8;
    std::fwrite(hdr.data(), 1, hdr.size(), f);
    for (int y = 0; y < yRes; ++y, off += chunk) std::fwrite(&off, 8, 1, f);
    std::vector<uint16_t> row(size_t(xRes) * 4);
    for (int y = 0; y < yRes; ++y) {
        for (int x = 0; x < xRes; ++x) {
            const size_t i = size_t(y) * xRes + x;
            row[x] = float_to_half(alpha ? alpha[i] : 1.f);
            row[size_t(xRes) + x] = float_to_half(pixels[3 * i + 2]);
            row[size_t(2) * xRes + x] = float_to_half(pixels[3 * i + 1]);
            row[size_t(3) * xRes + x] = float_to_half(pixels[3 * i]);
        }
        const int32_t yy = yOffset + y, sz = int32_t(row_bytes);
        std::fwrite(&yy, 4, 1, f); std::fwrite(&sz, 4, 1, f); std::fwrite(row.data(), 2, row.size(), f);
    }
    std::fclose(f);
    return true;
}

// Reads the files WriteRGBAImage above produces (scanline, uncompressed, channels A B G R half).  Every size and offset taken
// from the file is checked against the buffer: a truncated or foreign file (other compression, tiled, multi-part, other
// channels) is refused, never read out of bounds.  The reference writes PIZ through OpenEXR (exrio.cpp:75-96); neither a PIZ
// encoder nor any PIZ file exists in this image to validate a decoder against, so PIZ input is refused, not guessed at.
inline bool ReadRGBAImage(const std::string &name, ExrImage &img) {
    FILE *f = std::fopen(name.c_str(), "rb");
    if (!f) return false;
    std::vector<unsigned char> buf;
    { std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET); if (n < 0) { std::fclose(f); return false; }
      buf.resize(size_t(n)); if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); return false; } }
    std::fclose(f);
    const size_t N = buf.size();
    if (N < 8 || buf[0] != 0x76 || buf[1] != 0x2f || buf[2] != 0x31 || buf[3] != 0x01 || buf[4] != 2 || buf[5] != 0) return false;   // version 2, no tiled / multi-part flags
    auto cstr = [&](size_t &p, std::string &out) {          // NUL-terminated string inside the buffer
        size_t q = p; while (q < N && buf[q]) ++q;
        if (q >= N) return false;
        out.assign((const char *)&buf[p], q - p); p = q + 1; return true;
    };
    size_t p = 8; int dw[4] = {0, 0, -1, -1}, disp[4] = {0, 0, -1, -1}; int compression = -1; std::string chans; bool have_dw = false, have_disp = false;
    for (;;) {
        if (p >= N) return false;
        if (buf[p] == 0) { ++p; break; }
        std::string an, ty;
        if (!cstr(p, an) || !cstr(p, ty) || p + 4 > N) return false;
        int32_t sz; std::memcpy(&sz, &buf[p], 4); p += 4;
        if (sz < 0 || size_t(sz) > N - p) return false;
        if (an == "dataWindow") { if (sz != 16) return false; std::memcpy(dw, &buf[p], 16); have_dw = true; }
        else if (an == "displayWindow") { if (sz != 16) return false; std::memcpy(disp, &buf[p], 16); have_disp = true; }
        else if (an == "compression") { if (sz != 1) return false; compression = buf[p]; }
        else if (an == "channels") {
            size_t q = p; const size_t end = p + size_t(sz);
            while (q < end && buf[q]) {
                std::string c; size_t qq = q; if (!cstr(qq, c) || qq + 16 > end) return false;
                int32_t ptype; std::memcpy(&ptype, &buf[qq], 4); if (ptype != 1) return false;          // HALF only
                chans += c; q = qq + 16;
            }
        }
        p += size_t(sz);
    }
    if (compression != 0) {                                   // said out loud: this is the one foreign-file case a pbrt user will actually meet
        static const char *const names[] = {"NO", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
        std::fprintf(stderr, "exr file \"%s\" uses %s compression: only uncompressed half RGBA scanline files (what this library's WriteRGBAImage "
                             "produces) can be read; a file written by OpenEXR's RgbaOutputFile defaults to PIZ\n",
                     name.c_str(), compression > 0 && compression < 10 ? names[compression] : "an unknown");
        return false;
    }
    if (chans != "ABGR" || !have_dw || !have_disp) return false;
    const long long xr = (long long)dw[2] - dw[0] + 1, yr = (long long)dw[3] - dw[1] + 1;
    if (xr < 1 || yr < 1 || xr > 65536 || yr > 65536 || disp[2] < 0 || disp[3] < 0 || disp[2] >= 65536 || disp[3] >= 65536) return false;
    img.xRes = int(xr); img.yRes = int(yr); img.xOffset = dw[0]; img.yOffset = dw[1];
    img.totalXRes = disp[2] + 1; img.totalYRes = disp[3] + 1;
    const size_t row_bytes = size_t(img.xRes) * 8;
    if (p + size_t(8) * img.yRes > N) return false;                                               // scanline offset table
    img.rgb.assign(size_t(3) * img.xRes * img.yRes, 0.f); img.alpha.assign(size_t(img.xRes) * img.yRes, 0.f);
    for (int y = 0; y < img.yRes; ++y) {
        uint64_t off; std::memcpy(&off, &buf[p + size_t(8) * y], 8);
        if (off > N || N - off < 8 + row_bytes) return false;
        int32_t yy, sz; std::memcpy(&yy, &buf[off], 4); std::memcpy(&sz, &buf[off + 4], 4);
        const long long ry = (long long)yy - img.yOffset;
        if (ry < 0 || ry >= img.yRes || size_t(sz) != row_bytes) return false;
        const unsigned char *row = &buf[off + 8];
        for (int x = 0; x < img.xRes; ++x) {
            const size_t i = size_t(ry) * img.xRes + x;
            uint16_t h[4]; for (int c = 0; c < 4; ++c) std::memcpy(&h[c], row + (size_t(c) * img.xRes + x) * 2, 2);
            img.alpha[i] = half_to_float(h[0]); img.rgb[3 * i + 2] = half_to_float(h[1]);
            img.rgb[3 * i + 1] = half_to_float(h[2]); img.rgb[3 * i] = half_to_float(h[3]);
        }
    }
    return true;
}

// tools/exrassemble.cpp:42-75: every input's data window is copied into a display-window-sized RGBA image (zero where no input
// covers it); inputs must agree on the display window.  Returns the fraction of the image covered, < 0 on error.
inline float AssembleRGBAImages(const std::vector<std::string> &inputs, const std::string &out) {
    int xres = 0, yres = 0; long long ndone = 0;
    std::vector<float> rgb, alpha;
    for (const std::string &fn : inputs) {
        ExrImage im;
        if (!ReadRGBAImage(fn, im)) { std::fprintf(stderr, "couldn't read exr file \"%s\"!\n", fn.c_str()); continue; }
        if (rgb.empty()) { xres = im.totalXRes; yres = im.totalYRes; rgb.assign(size_t(3) * xres * yres, 0.f); alpha.assign(size_t(xres) * yres, 0.f); }
        else if (xres != im.totalXRes || yres != im.totalYRes) return -1.f;
        if (im.xOffset < 0 || im.yOffset < 0 || im.xOffset + im.xRes > xres || im.yOffset + im.yRes > yres) return -1.f;
        ndone += (long long)im.xRes * im.yRes;
        for (int y = 0; y < im.yRes; ++y) {
            std::memcpy(&rgb[3 * (size_t(im.yOffset + y) * xres + im.xOffset)], &im.rgb[3 * size_t(y) * im.xRes], size_t(3) * im.xRes * sizeof(float));
            std::memcpy(&alpha[size_t(im.yOffset + y) * xres + im.xOffset], &im.alpha[size_t(y) * im.xRes], size_t(im.xRes) * sizeof(float));
        }
    }
    if (rgb.empty()) return -1.f;
    if (!WriteRGBAImage(out, rgb.data(), alpha.data(), xres, yres, xres, yres, 0, 0)) return -1.f;
    return float(ndone) / (float(xres) * float(yres));
}

}  // namespace pbrthip
