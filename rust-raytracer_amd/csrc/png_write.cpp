// png_write.cpp -- rt_save_png: 8-bit RGB PNG of a framebuffer, the counterpart of
// raytracer-cli/src/output/mod.rs:7-17 + png_export.rs:1-14.  No image library is
// linked: the file is written with stored (uncompressed) deflate blocks, which every
// PNG reader accepts.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_error.h"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32(0, out.data() + at, out.size() - at));
}

}  // namespace

extern "C" int rt_save_png(const char *path, const float *fb_rgb, uint32_t width, uint32_t height) {
    return rt::guarded([&]() -> int {
        if (!path || !fb_rgb || width == 0 || height == 0) return rt::set_error(RT_ERR_INVALID, "null or empty argument");
        const std::string p(path);
        const size_t dot = p.find_last_of('.');
        const size_t slash = p.find_last_of('/');
        if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) {
            // output/mod.rs:10-14: any extension other than "png" is unimplemented
            const std::string ext = p.substr(dot + 1);
            if (ext != "png") return rt::set_error(RT_ERR_UNSUPPORTED, "The format '" + ext + "' is not supported");
        } else {
            return RT_OK;  // output/mod.rs:8: no extension, nothing is written
        }
        std::vector<uint8_t> rgb((size_t)width * height * 3);
        int rc = rt_framebuffer_to_rgb8(fb_rgb, width, height, rgb.data());
        if (rc) return rc;

        // raw scanlines: filter byte 0 + RGB
        const size_t row = (size_t)width * 3 + 1;
        std::vector<uint8_t> raw(row * height);
        for (uint32_t y = 0; y < height; ++y) {
            raw[y * row] = 0;
            std::memcpy(&raw[y * row + 1], &rgb[(size_t)y * width * 3], (size_t)width * 3);
        }
        // zlib stream of stored blocks
        std::vector<uint8_t> z;
        z.push_back(0x78); z.push_back(0x01);
        uint32_t a = 1, b = 0;
        for (size_t at = 0; at < raw.size();) {
            const size_t n = std::min<size_t>(65535, raw.size() - at);
            z.push_back(at + n == raw.size() ? 1 : 0);
            z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
            z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
            z.insert(z.end(), raw.begin() + (long)at, raw.begin() + (long)(at + n));
            for (size_t i = at; i < at + n; ++i) {  // adler32
                a = (a + raw[i]) % 65521u;
                b = (b + a) % 65521u;
            }
            at += n;
        }
        put_be32(z, (b << 16) | a);

        std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
        std::vector<uint8_t> ihdr;
        put_be32(ihdr, width); put_be32(ihdr, height);
        ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGB
        chunk(out, "IHDR", ihdr);
        chunk(out, "IDAT", z);
        chunk(out, "IEND", {});
        FILE *f = std::fopen(path, "wb");
        if (!f) return rt::set_error(RT_ERR_IO, std::string("cannot open ") + path);
        const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
        std::fclose(f);
        if (!ok) return rt::set_error(RT_ERR_IO, std::string("short write to ") + path);
        return RT_OK;
    });
}
