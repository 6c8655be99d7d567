// cniic_bench.cpp -- C++ counterpart of the reference's harness (src/bench.rs:15-83, src/main.rs:57-70):
//   cniic_bench --codec=<expr> <image>...
// per image: encode -> size -> ratio vs w*h*24 -> decode -> MSE -> CSV row
//   name,compressed_size,compression_ratio,error            (bench.rs:68-75)
// written to output/<codec.name()>.csv (bench.rs:85-91) and echoed to stdout with four more columns:
//   mpix_per_s, iters (K-means iterations), hbm_gbps (algorithmic bytes of the encode, SURVEY 8(d), over its wall time),
//   roofline_frac (that over the 8 TB/s of HBM3E).  A lossless codec whose
// decode differs is an error (bench.rs:50-59).  Images are PNG files (bench.rs:30 `image::open`; configs[0] is "one 512x512 RGB PNG":
// 8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced, inflated with zlib, alpha dropped like DynamicImage::to_rgb8),
// binary PPM (P6) files or synthetic specs "synth:P:4096x4096:2" / "synth:U:512x512:1" (kind, size, seed offset).
// The raw size is computed in 64 bits (the reference's u32 h*w*24 wraps above 13377^2 pixels).
// Images run one per worker thread, each with its own cniic_ctx (the reference uses rayon, bench.rs:24-28).
#include <sys/stat.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/cniic_hip.h"

struct Image { uint32_t w = 0, h = 0; std::vector<uint8_t> rgb; bool synth = false; int kind = 0; uint64_t seed = 0; };

static bool read_ppm(const std::string &path, Image &im) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P6") return false;
    auto next_int = [&]() -> long {
        for (;;) {
            int ch = f.peek();
            if (ch == '#') { std::string line; std::getline(f, line); }
            else if (isspace(ch)) f.get();
            else break;
        }
        long v; f >> v; return v;
    };
    long w = next_int(), h = next_int(), maxv = next_int();
    f.get();
    if (w <= 0 || h <= 0 || maxv != 255) return false;
    im.w = (uint32_t)w; im.h = (uint32_t)h;
    im.rgb.resize((size_t)w * h * 3);
    f.read(reinterpret_cast<char *>(im.rgb.data()), (std::streamsize)im.rgb.size());
    return (bool)f;
}

// A PNG of the kinds a photograph comes in (bench.rs:30: `image::open(path)`, then every codec takes `to_rgb8()`): bit depth 8, colour type
// 0 / 2 / 3 / 4 / 6, no interlace.  Chunks: IHDR, PLTE, IDAT (concatenated, one zlib stream), IEND; CRCs are not checked.
static bool read_png(const std::string &path, Image &im) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 33 || memcmp(b.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t at) { return ((uint32_t)b[at] << 24) | ((uint32_t)b[at + 1] << 16) | ((uint32_t)b[at + 2] << 8) | b[at + 3]; };
    uint32_t w = 0, h = 0, ctype = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t at = 8; at + 12 <= b.size();) {
        const uint32_t len = be32(at);
        if (at + 12 + (size_t)len > b.size()) return false;
        const char *ty = reinterpret_cast<const char *>(&b[at + 4]);
        const uint8_t *d = &b[at + 8];
        if (!memcmp(ty, "IHDR", 4)) {
            if (len != 13) return false;
            w = be32(at + 8); h = be32(at + 12); ctype = d[9];
            if (d[8] != 8 || d[10] != 0 || d[11] != 0 || d[12] != 0) return false;   // bit depth 8, deflate, adaptive filters, no interlace
        } else if (!memcmp(ty, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(ty, "IEND", 4)) break;
        at += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || !w || !h || (ctype == 3 && plte.size() < 3)) return false;
    if (w > 65535u || h > 65535u || (uint64_t)w * h > (1ull << 30)) return false;   // (a header that asks for more than any image here has)
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) return false;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    im.w = w; im.h = h;
    im.rgb.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = &raw[(stride + 1) * y];
        const uint8_t ft = row[0];
        for (size_t i = 0; i < stride; i++) {   // the five scanline filters of the PNG specification, per byte, on a stride of `ch` bytes
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, up = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = up;
            else if (ft == 3) pred = (a + up) >> 1;
            else if (ft == 4) { const int pq = a + up - c, pa = abs(pq - a), pb = abs(pq - up), pc = abs(pq - c); pred = (pa <= pb && pa <= pc) ? a : pb <= pc ? up : c; }
            else if (ft != 0) return false;
            cur[i] = (uint8_t)(row[1 + i] + pred);
        }
        uint8_t *o = &im.rgb[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *px = &cur[(size_t)x * ch];
            if (ctype == 2 || ctype == 6) { o[3 * x] = px[0]; o[3 * x + 1] = px[1]; o[3 * x + 2] = px[2]; }
            else if (ctype == 3) { const size_t e = (size_t)px[0] * 3; if (e + 3 > plte.size()) return false; o[3 * x] = plte[e]; o[3 * x + 1] = plte[e + 1]; o[3 * x + 2] = plte[e + 2]; }
            else { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = px[0]; }
        }
        prev.swap(cur);
    }
    return true;
}

static bool parse_synth(const std::string &s, Image &im) {  // synth:P:4096x4096:2
    char kind;
    unsigned w, h;
    unsigned long long off = 0;
    if (sscanf(s.c_str(), "synth:%c:%ux%u:%llu", &kind, &w, &h, &off) < 3) return false;
    im.synth = true; im.kind = (kind == 'U' || kind == 'u') ? CNIIC_SYNTH_UNIFORM : CNIIC_SYNTH_PHOTO;
    im.w = w; im.h = h; im.seed = 0x636E696963ULL + off;
    return true;
}

int main(int argc, char **argv) {
    if (argc < 3 || strncmp(argv[1], "--codec=", 8) != 0) {
        fprintf(stderr, "Usage: cniic_bench --codec=<codec> [<img file>..]\nAvailable codecs:\n  hufman\n  cluster-colors(<ncolors>)\n  voronoi(<k>)\n  delta\n  hilbert(rle)\n");
        return 2;
    }
    const std::string expr = argv[1] + 8;
    char name[64];
    if (cniic_codec_name(expr.c_str(), name, sizeof name) != CNIIC_OK) { fprintf(stderr, "Malformed codec argument: %s\n", expr.c_str()); return 2; }
    const bool lossless = cniic_codec_is_lossless(expr.c_str()) == 1;
    mkdir("output", 0755);
    const std::string csv_path = std::string("output/") + name + ".csv";
    FILE *csv = fopen(csv_path.c_str(), "w");
    if (!csv) { perror(csv_path.c_str()); return 1; }
    std::mutex mu;
    bool wrote_header = false;
    int failures = 0;
    std::vector<std::string> paths(argv + 2, argv + argc);
    const unsigned nthreads = std::min<unsigned>((unsigned)paths.size(), std::max(1u, std::min(4u, std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    size_t next = 0;
    for (unsigned t = 0; t < nthreads; t++)
        pool.emplace_back([&] {
            cniic_ctx *ctx = nullptr;
            if (cniic_ctx_create(0, nullptr, &ctx) != CNIIC_OK) { std::lock_guard<std::mutex> lk(mu); fprintf(stderr, "no usable MI355X\n"); failures++; return; }
            for (;;) {
                std::string p;
                { std::lock_guard<std::mutex> lk(mu); if (next >= paths.size()) break; p = paths[next++]; }
                Image im;
                void *dimg = nullptr;
                auto fail = [&](const char *what) { std::lock_guard<std::mutex> lk(mu); fprintf(stderr, "%s: %s (%s)\n", p.c_str(), what, cniic_last_error(ctx)); failures++; };
                if (!parse_synth(p, im) && !read_png(p, im) && !read_ppm(p, im)) { fail("cannot read image (PNG, binary PPM or synth:<P|U>:<w>x<h>[:seed])"); continue; }
                const uint64_t npx = (uint64_t)im.w * im.h;
                if (cniic_dev_alloc(ctx, npx * 3, &dimg) != CNIIC_OK) { fail("device allocation"); continue; }
                if (im.synth) cniic_synth_image(ctx, im.kind, im.seed, im.w, im.h, (uint8_t *)dimg);
                else cniic_memcpy(ctx, dimg, im.rgb.data(), npx * 3);
                uint64_t distinct = 0;  // (outside the timed encode: the algorithmic bytes of cluster-colors depend on it)
                if (strncmp(name, "cluster-colors", 14) == 0) cniic_hist_rgb24(ctx, (const uint8_t *)dimg, npx, nullptr, nullptr, 0, &distinct);
                std::vector<uint8_t> data(64 + npx * 16 + (1 << 16));
                uint64_t len = 0;
                cniic_kmeans_stats st{};
                auto t0 = std::chrono::steady_clock::now();
                int rc = cniic_codec_encode(ctx, expr.c_str(), (const uint8_t *)dimg, im.w, im.h, data.data(), data.size(), &len, &st);
                double enc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rc != CNIIC_OK) { fail("encode"); cniic_dev_free(ctx, dimg); continue; }
                const double raw_bits = (double)npx * 24.0;              // bench.rs:41 (in 64 bits)
                const double ratio = (double)len / raw_bits;             // bench.rs:43
                void *dback = nullptr;
                cniic_dev_alloc(ctx, npx * 3, &dback);
                uint32_t w2 = 0, h2 = 0;
                rc = cniic_codec_decode(ctx, expr.c_str(), data.data(), len, (uint8_t *)dback, npx * 3, &w2, &h2);
                if (rc != CNIIC_OK) { fail("Could not decode the image"); cniic_dev_free(ctx, dimg); cniic_dev_free(ctx, dback); continue; }
                double mse = 0;
                cniic_mse(ctx, (const uint8_t *)dimg, (const uint8_t *)dback, npx, &mse);   // bench.rs:95-104
                cniic_dev_free(ctx, dimg);
                cniic_dev_free(ctx, dback);
                std::lock_guard<std::mutex> lk(mu);
                if (mse != 0.0 && lossless) { fprintf(stderr, "%s: Decoded image doesn't match\n", p.c_str()); failures++; continue; }
                // algorithmic bytes of the encode: hufman 3 B/px histogram + 3 B/px pack; delta 3 B/px gather + 4 B/px symbols
                // written + 4 read; hilbert-rle 3 + 3 B/px; voronoi 7 B/px/iteration; cluster-colors 3 B/px histogram + 10 B/colour/iteration
                // + 4 B/px remap; every codec + the stream it writes
                double algo = (double)len;
                if (!strcmp(name, "Hufman")) algo += 6.0 * npx;
                else if (!strcmp(name, "delta")) algo += 11.0 * npx;
                else if (!strcmp(name, "hilbert-rle")) algo += 6.0 * npx;
                else if (!strncmp(name, "voronoi", 7)) algo += 7.0 * npx * (double)st.iterations;
                else algo += 7.0 * npx + 10.0 * (double)distinct * (double)st.iterations;
                const double gbps = algo / (enc_ms * 1e-3) / 1e9;
                if (!wrote_header) { fprintf(csv, "name,compressed_size,compression_ratio,error\n"); printf("name,compressed_size,compression_ratio,error,mpix_per_s,iters,hbm_gbps,roofline_frac\n"); wrote_header = true; }
                fprintf(csv, "%s,%llu,%.17g,%.17g\n", p.c_str(), (unsigned long long)len, ratio * 100.0, mse);
                printf("%s,%llu,%.6f,%.4f,%.1f,%llu,%.1f,%.4f\n", p.c_str(), (unsigned long long)len, ratio * 100.0, mse, npx / enc_ms / 1e3, (unsigned long long)st.iterations,
                       gbps, gbps / 8000.0);
            }
            cniic_ctx_destroy(ctx);
        });
    for (auto &th : pool) th.join();
    fclose(csv);
    return failures ? 1 : 0;
}
