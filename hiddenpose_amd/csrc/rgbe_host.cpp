// Radiance .hdr container -> flat RGBE bytes (hp_rgbe_decode): the host half of the measurement ingest
// (utils/nlos_pose_dataloader.py:76 `cv2.imread(path, -1)`; the arithmetic lives in ingest_kernels.hip).  Parses
// UNTRUSTED file bytes: every read is bounds-checked against the end of the buffer, sizes come from a range-checked
// header, nothing throws across the C ABI.  HIP-free (see hp_host.h); fuzzed under ASan / UBSan by
// tests/test_host_asan.py.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "hp_host.h"

namespace hp {
struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool line(std::string& out) {
    out.clear();
    while (p < end && *p != '\n') out.push_back((char)*p++);
    if (p >= end) return false;
    ++p;
    return true;
  }
};

static int parse_header(Cursor& c, int& W, int& H) {
  std::string ln;
  if (!c.line(ln) || ln.size() < 2 || ln[0] != '#' || ln[1] != '?') {
    set_error("rgbe: missing '#?' signature line");
    return HP_ERR_BAD_ARG;
  }
  bool fmt = false;
  for (;;) {
    if (!c.line(ln)) {
      set_error("rgbe: header ends before the blank line");
      return HP_ERR_BAD_ARG;
    }
    if (ln.empty()) break;
    if (ln == "FORMAT=32-bit_rle_rgbe") fmt = true;
  }
  if (!fmt) {
    set_error("rgbe: FORMAT=32-bit_rle_rgbe line not found");
    return HP_ERR_BAD_ARG;
  }
  // "-Y H +X W": parsed by hand -- a scanf %d has no defined behaviour on overflow, and these digits come from a file
  auto number = [](const char*& q, int& out) -> bool {
    long v = 0;
    int digits = 0;
    while (*q >= '0' && *q <= '9' && digits < 10) v = v * 10 + (*q++ - '0'), ++digits;
    if (digits == 0 || (*q >= '0' && *q <= '9') || v < 1 || v > 0x7fffffffl) return false;
    out = (int)v;
    return true;
  };
  bool ok = c.line(ln) && ln.compare(0, 3, "-Y ") == 0;
  if (ok) {
    const char* q = ln.c_str() + 3;
    ok = number(q, H) && q[0] == ' ' && q[1] == '+' && q[2] == 'X' && q[3] == ' ';
    if (ok) {
      q += 4;
      ok = number(q, W);   // (trailing text after the width is ignored, as sscanf did)
    }
  }
  if (!ok) {
    set_error("rgbe: only the standard '-Y H +X W' orientation is supported (got '%.64s')", ln.c_str());
    return HP_ERR_UNSUPPORTED;
  }
  return HP_OK;
}


}  // namespace hp

using namespace hp;

extern "C" int hp_rgbe_decode(const unsigned char* file, size_t nbytes, int* width, int* height, unsigned char* rgbe,
                              size_t rgbe_capacity) {
  HP_REQUIRE(file && width && height, "hp_rgbe_decode: null argument");
  Cursor c{file, file + nbytes};
  int W = 0, H = 0;
  int rc = parse_header(c, W, H);
  if (rc) return rc;
  // A run-length scanline spends 2 bytes on at most 127 equal bytes, a flat file 1 for 1: a header that claims more pixels
  // than 64 x the file's size describes no file this decoder could finish, and a caller sizing its buffer from the query
  // must not be made to allocate W * H * 4 bytes on the word of 20 header characters.
  const size_t need = (size_t)W * (size_t)H * 4;   // W, H < 2^31: no overflow in 64 bits
  HP_REQUIRE(need / 64 <= nbytes, "rgbe: the header claims %d x %d pixels, more than a %zu-byte file can hold", W, H, nbytes);
  *width = W;
  *height = H;
  if (!rgbe) return HP_OK;  // size query
  HP_REQUIRE(rgbe_capacity >= need, "hp_rgbe_decode: output buffer holds %zu bytes, %zu needed", rgbe_capacity, need);
  std::string scan;
  try {
    scan.assign((size_t)W * 4, '\0');
  } catch (const std::bad_alloc&) {   // nothing may throw across the C ABI
    set_error("hp_rgbe_decode: out of memory for a %d-pixel scanline", W);
    return HP_ERR_BAD_ARG;
  }
  for (int y = 0; y < H; ++y) {
    uint8_t* dst = rgbe + (size_t)y * W * 4;
    HP_REQUIRE(c.end - c.p >= 4, "rgbe: file ends in scanline %d", y);
    const bool rle = W >= 8 && W < 32768 && c.p[0] == 2 && c.p[1] == 2 && !(c.p[2] & 0x80);
    if (!rle) {
      // flat pixels: the rest of the file is uncompressed (the decision is taken at the first such scanline)
      const size_t rest = (size_t)(H - y) * W * 4;
      HP_REQUIRE((size_t)(c.end - c.p) >= rest, "rgbe: file ends inside the flat pixel block");
      std::memcpy(dst, c.p, rest);
      return HP_OK;
    }
    HP_REQUIRE(((c.p[2] << 8) | c.p[3]) == W, "rgbe: scanline %d has a wrong width", y);
    c.p += 4;
    for (int ch = 0; ch < 4; ++ch) {
      int x = 0;
      while (x < W) {
        HP_REQUIRE(c.end - c.p >= 2, "rgbe: file ends in scanline %d", y);
        int cnt = *c.p++;
        if (cnt > 128) {
          cnt -= 128;
          HP_REQUIRE(cnt > 0 && x + cnt <= W, "rgbe: bad run in scanline %d", y);
          const uint8_t v = *c.p++;
          for (int i = 0; i < cnt; ++i) scan[(size_t)ch * W + x++] = (char)v;
        } else {
          HP_REQUIRE(cnt > 0 && x + cnt <= W && c.end - c.p >= cnt, "rgbe: bad literal block in scanline %d", y);
          std::memcpy(&scan[(size_t)ch * W + x], c.p, (size_t)cnt);
          c.p += cnt;
          x += cnt;
        }
      }
    }
    for (int x = 0; x < W; ++x)
      for (int ch = 0; ch < 4; ++ch) dst[(size_t)x * 4 + ch] = (uint8_t)scan[(size_t)ch * W + x];
  }
  return HP_OK;
}

