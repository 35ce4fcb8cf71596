// png_codec.cpp -- minimal PNG reader/writer on zlib for the host-side file helpers.
// Reader: 8-bit RGB (colour type 2) and RGBA (6) only, the formats the reference accepts
// (Src/Texture.cpp:97); anything else is reported as unsupported so the caller degrades
// to "no texture" like the reference does.  Writer: 8-bit RGB, what SaveBufferToPNG emits
// (Src/Texture.cpp:221-223).
#include "rtw_host.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>

namespace rtw {

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the per-scanline filters of one (sub-)image in place; rows are 1 + stride bytes
bool unfilter(uint8_t* data, size_t rows, size_t stride, int bpp)
{
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t* row = data + y * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)bpp ? cur[i - (size_t)bpp] : 0;
            const int b = prev[i];
            const int c = i >= (size_t)bpp ? prev[i - (size_t)bpp] : 0;
            int v = cur[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: return false;
            }
            cur[i] = (uint8_t)v;
        }
        prev = cur;
    }
    return true;
}

}  // namespace

std::string png_load(const std::string& path, std::vector<uint8_t>& texels, int& w, int& h, int& channels)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::vector<uint8_t> file;
    uint8_t buf[65536]; size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    std::fclose(f);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return "not a PNG: " + path;

    std::vector<uint8_t> idat;
    int bit_depth = 0, colour = -1, interlace = 0;
    w = h = 0;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return "truncated PNG: " + path;
        const uint8_t* body = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)be32(body); h = (int)be32(body + 4);
            bit_depth = body[8]; colour = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0) return "bad PNG header: " + path;
    if (!((colour == 2 || colour == 6) && bit_depth == 8)) return "unsupported PNG format: " + path;
    channels = colour == 2 ? 3 : 4;
    const int bpp = channels;

    // size of the filtered stream
    static const int xs[7] = { 0, 4, 0, 2, 0, 1, 0 }, ys[7] = { 0, 0, 4, 0, 2, 0, 1 };
    static const int dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
    size_t raw_size = 0;
    if (!interlace) raw_size = (size_t)h * ((size_t)w * (size_t)bpp + 1);
    else for (int p = 0; p < 7; p++) {
        const int pw = (w - xs[p] + dx[p] - 1) / dx[p], ph = (h - ys[p] + dy[p] - 1) / dy[p];
        if (pw > 0 && ph > 0) raw_size += (size_t)ph * ((size_t)pw * (size_t)bpp + 1);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf out_len = (uLongf)raw_size;
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw_size)
        return "PNG inflate failed: " + path;

    texels.assign((size_t)w * (size_t)h * (size_t)channels, 0);
    if (!interlace) {
        const size_t stride = (size_t)w * (size_t)bpp;
        if (!unfilter(raw.data(), (size_t)h, stride, bpp)) return "bad PNG filter: " + path;
        for (int y = 0; y < h; y++) std::memcpy(&texels[(size_t)y * stride], &raw[(size_t)y * (stride + 1) + 1], stride);
    } else {
        size_t off = 0;
        for (int p = 0; p < 7; p++) {
            const int pw = (w - xs[p] + dx[p] - 1) / dx[p], ph = (h - ys[p] + dy[p] - 1) / dy[p];
            if (pw <= 0 || ph <= 0) continue;
            const size_t stride = (size_t)pw * (size_t)bpp;
            if (!unfilter(&raw[off], (size_t)ph, stride, bpp)) return "bad PNG filter: " + path;
            for (int y = 0; y < ph; y++)
                for (int x = 0; x < pw; x++)
                    std::memcpy(&texels[((size_t)(ys[p] + y * dy[p]) * (size_t)w + (size_t)(xs[p] + x * dx[p])) * (size_t)bpp],
                                &raw[off + (size_t)y * (stride + 1) + 1 + (size_t)x * (size_t)bpp], (size_t)bpp);
            off += (size_t)ph * (stride + 1);
        }
    }
    return std::string();
}

std::string png_save_rgb(const std::string& path, const uint8_t* rgb, int w, int h)
{
    if (w <= 0 || h <= 0) return "bad image size";
    const size_t stride = (size_t)w * 3;
    std::vector<uint8_t> raw((size_t)h * (stride + 1));
    for (int y = 0; y < h; y++) {
        raw[(size_t)y * (stride + 1)] = 0;
        std::memcpy(&raw[(size_t)y * (stride + 1) + 1], rgb + (size_t)y * stride, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return "deflate failed";

    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return "cannot open " + path + " for writing";
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::fwrite(sig, 1, 8, f);
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8]; put_be32(hdr, len); std::memcpy(hdr + 4, type, 4);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0L, (const Bytef*)type, 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t c[4]; put_be32(c, (uint32_t)crc);
        std::fwrite(c, 1, 4, f);
    };
    uint8_t ihdr[13]; put_be32(ihdr, (uint32_t)w); put_be32(ihdr + 4, (uint32_t)h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    const bool ok = std::fclose(f) == 0;
    return ok ? std::string() : "write failed: " + path;
}

}  // namespace rtw
