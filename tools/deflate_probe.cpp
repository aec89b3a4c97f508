// Line-aware deflate against zlib on ASQG-like text: g++ -O2 -I siga_amd/host tools/deflate_probe.cpp -lz -o build/deflate_probe
// build/deflate_probe [file]   (without a file: synthetic VT and ED lines).  Every block is inflated again and compared.
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "line_deflate.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool inflate_equals(const std::string& raw, const char* in, size_t n) {
  std::string back(n + 16, '\0');
  z_stream z;
  memset(&z, 0, sizeof(z));
  inflateInit2(&z, -15);
  z.next_in = (Bytef*)raw.data();
  z.avail_in = (uInt)raw.size();
  z.next_out = (Bytef*)&back[0];
  z.avail_out = (uInt)back.size();
  int rc = inflate(&z, Z_FINISH);
  size_t got = z.total_out;
  inflateEnd(&z);
  return rc == Z_STREAM_END && got == n && memcmp(back.data(), in, n) == 0;
}

static void run(const char* what, const std::string& text) {
  const size_t kBlock = 1 << 20;
  for (int mode = 0; mode < 3; ++mode) {
    double t0 = now();
    size_t out = 0;
    bool ok = true;
    std::string o;
    for (size_t off = 0; off < text.size(); off += kBlock) {
      size_t m = std::min(kBlock, text.size() - off);
      if (mode == 0) {
        o.clear();
        sigah::ldef::deflate_lines((const unsigned char*)text.data() + off, m, true, &o);
      } else {
        z_stream z;
        memset(&z, 0, sizeof(z));
        deflateInit2(&z, mode == 1 ? 4 : 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        o.resize(deflateBound(&z, (uLong)m) + 16);
        z.next_in = (Bytef*)text.data() + off;
        z.avail_in = (uInt)m;
        z.next_out = (Bytef*)&o[0];
        z.avail_out = (uInt)o.size();
        deflate(&z, Z_FINISH);
        o.resize(z.total_out);
        deflateEnd(&z);
      }
      out += o.size();
    }
    double dt = now() - t0;
    if (mode == 0)
      for (size_t off = 0; off < text.size(); off += kBlock) {
        size_t m = std::min(kBlock, text.size() - off);
        o.clear();
        sigah::ldef::deflate_lines((const unsigned char*)text.data() + off, m, true, &o);
        ok = ok && inflate_equals(o, text.data() + off, m);
      }
    printf("%-8s %-12s %8.1f MB/s  ratio %.4f%s\n", what, mode == 0 ? "line-deflate" : mode == 1 ? "zlib -4" : "zlib -6", text.size() / dt / 1e6,
           (double)out / text.size(), mode == 0 ? (ok ? "  (inflates to the input)" : "  MISMATCH") : "");
  }
}

int main(int argc, char** argv) {
  if (argc > 1) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::string text;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
    fclose(f);
    run("file", text);
    return 0;
  }
  srand(1);
  const size_t G = 2000000, n = 300000;
  std::string g(G, 'A');
  for (size_t i = 0; i < G; ++i) g[i] = "ACGT"[rand() & 3];
  std::string vt, ed;
  char tmp[256];
  for (size_t i = 0; i < n; ++i) {
    vt.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "VT\tr%zu\t", i));
    vt.append(g, rand() % (G - 150), 150);
    vt += "\tSS:i:0\n";
  }
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      int ov = 45 + rand() % 100;
      ed.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "ED\tr%zu r%zu %d 149 150 0 %d 150 0 0\n", i, (size_t)(rand() % n), 150 - ov, ov - 1));
    }
  run("VT", vt);
  run("ED", ed);
  return 0;
}
