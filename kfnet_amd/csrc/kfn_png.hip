// kfn_png.hip -- host side of the image stream: PNG files -> uint8 RGB frames, on native threads.
//
// Replaces tf.image.decode_png(channels=3) inside the reference's queue runners (KFNet/train.py:195-239, decode at
// :213-217; eval.py feeds the same list).  The Python host decoded with PIL on a thread pool: PIL releases the GIL while it
// inflates, but every file still takes the interpreter lock a few dozen times (chunk parsing, buffer hand-overs), and the
// thread that enqueues the GPU work shares that lock -- measured on the GPU box (round 5, profiles/r05_eval_png_ab.log):
// the consumer's enqueue time grew in proportion to the number of decode threads (0.09 / 0.24 / 0.43 s of a 1.8 s run at
// 8 / 16 / 32 threads) while the decode rate stopped scaling.  Here a chunk is ONE call: the files are read, inflated (zlib),
// unfiltered and expanded to RGB by `threads` std::threads straight into the caller's (page-locked) staging buffer.
//
// Scope: what the reference's data holds and PIL's writer produces -- non-interlaced PNGs of bit depth <= 8, colour types
// 0 (gray), 2 (RGB), 3 (palette), 4 (gray + alpha), 6 (RGBA); alpha is dropped and gray replicated, as decode_png(channels=3)
// / PIL's convert('RGB') do.  Interlaced or 16-bit files, and files without the PNG signature, are reported per file as
// KFN_PNG_UNSUPPORTED and the host decodes those with PIL; a corrupt or wrong-sized file is an error that names the file.  Host code only: no device access.
#include "kfn_common.h"

#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {

struct PngResult {
  int status;          // KFN_PNG_OK / KFN_PNG_UNSUPPORTED / KFN_PNG_ERROR
  std::string what;
};

inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// scanline filters of the PNG specification (section 9): cur[] holds the filtered bytes and becomes the raw ones
void unfilter_row(int type, unsigned char* cur, const unsigned char* prev, int bpp, int n) {
  switch (type) {
    case 0: break;
    case 1:
      for (int i = bpp; i < n; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]);
      break;
    case 2:
      if (prev) for (int i = 0; i < n; ++i) cur[i] = (unsigned char)(cur[i] + prev[i]);
      break;
    case 3:
      for (int i = 0; i < n; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
        cur[i] = (unsigned char)(cur[i] + ((a + b) >> 1));
      }
      break;
    default:   // 4 (the caller has checked the range)
      for (int i = 0; i < n; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        cur[i] = (unsigned char)(cur[i] + paeth(a, b, c));
      }
  }
}

PngResult decode_one(const char* path, int H, int W, unsigned char* dst) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return {KFN_PNG_ERROR, std::string("cannot open ") + path};
  std::vector<unsigned char> file;
  {
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 8 + 25 + 12) { std::fclose(f); return {KFN_PNG_ERROR, std::string(path) + " is not a PNG file (too short)"}; }
    file.resize((size_t)sz);
    const size_t got = std::fread(file.data(), 1, (size_t)sz, f);
    std::fclose(f);
    if (got != (size_t)sz) return {KFN_PNG_ERROR, std::string("short read of ") + path};
  }
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  // (not a PNG at all -- a JPEG in the list, say: the host's general-purpose decoder decides, as it did before this file existed)
  if (std::memcmp(file.data(), sig, 8) != 0) return {KFN_PNG_UNSUPPORTED, std::string(path) + " is not a PNG file (signature)"};
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
  unsigned char palette[256 * 3] = {0};      // (an index past PLTE's entries reads black, as libpng's readers do)
  int n_pal = 0;
  std::vector<unsigned char> idat;
  idat.reserve(file.size());
  bool seen_end = false, seen_hdr = false;
  while (pos + 12 <= file.size() && !seen_end) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    if ((size_t)len > file.size() - pos - 12) return {KFN_PNG_ERROR, std::string(path) + ": truncated chunk"};
    const unsigned char* body = &file[pos + 8];
    // The CRC of every CRITICAL chunk (upper-case first letter: IHDR, PLTE, IDAT, IEND) is checked like libpng does (tf.image.decode_png
    // of the reference, PIL): zlib's adler32 only covers the IDAT stream, and a damaged IHDR / PLTE would otherwise decode to a
    // wrong frame with status OK (about 5 % of a 20k-case mutation fuzz).  Ancillary chunks are skipped unread.
    if ((type[0] & 0x20) == 0) {
      const uint32_t want = be32(body + len);
      const uint32_t got = (uint32_t)crc32(crc32(0L, type, 4), body, (uInt)len);
      if (want != got) return {KFN_PNG_ERROR, std::string(path) + ": CRC error in chunk " + std::string(reinterpret_cast<const char*>(type), 4)};
    }
    if (pos == 8 && std::memcmp(type, "IHDR", 4) != 0) return {KFN_PNG_ERROR, std::string(path) + ": first chunk is not IHDR"};
    if (!std::memcmp(type, "IHDR", 4)) {
      if (seen_hdr) return {KFN_PNG_ERROR, std::string(path) + ": duplicate IHDR"};
      if (len != 13) return {KFN_PNG_ERROR, std::string(path) + ": bad IHDR"};
      w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      if (body[10] != 0 || body[11] != 0) return {KFN_PNG_ERROR, std::string(path) + ": unknown compression / filter method"};
      seen_hdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len % 3 != 0 || len > 768) return {KFN_PNG_ERROR, std::string(path) + ": bad PLTE"};
      std::memcpy(palette, body, len);
      n_pal = (int)(len / 3);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      seen_end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!seen_hdr || idat.empty()) return {KFN_PNG_ERROR, std::string(path) + ": no IHDR / IDAT"};
  if (w != W || h != H) {
    char b[160];
    std::snprintf(b, sizeof b, "%s is %dx%d, expected %dx%d", path, h, w, H, W);
    return {KFN_PNG_ERROR, b};
  }
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return {KFN_PNG_ERROR, std::string(path) + ": unknown colour type"};
  }
  if (interlace != 0 || depth > 8) return {KFN_PNG_UNSUPPORTED, std::string(path) + ": interlaced or 16-bit"};
  if (!(depth == 8 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
    return {KFN_PNG_ERROR, std::string(path) + ": bit depth not allowed for its colour type"};
  if (ctype == 3 && n_pal == 0) return {KFN_PNG_ERROR, std::string(path) + ": palette image without PLTE"};
  const int row_bytes = (W * channels * depth + 7) / 8;
  const int bpp = (channels * depth + 7) / 8 > 0 ? (channels * depth + 7) / 8 : 1;
  std::vector<unsigned char> raw((size_t)H * (row_bytes + 1));
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return {KFN_PNG_ERROR, "inflateInit failed"};
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    const int rc = inflate(&zs, Z_FINISH);
    const size_t produced = raw.size() - zs.avail_out;
    inflateEnd(&zs);
    // (Z_BUF_ERROR with the output full = data behind the image: tolerated like libpng's "too much image data" warning)
    if (!(rc == Z_STREAM_END || (rc == Z_BUF_ERROR && produced == raw.size())) || produced != raw.size())
      return {KFN_PNG_ERROR, std::string(path) + ": corrupt or truncated image data"};
  }
  const unsigned char* prev = nullptr;
  for (int y = 0; y < H; ++y) {
    unsigned char* line = raw.data() + (size_t)y * (row_bytes + 1);
    const int ft = line[0];
    if (ft > 4) return {KFN_PNG_ERROR, std::string(path) + ": unknown filter type"};
    unsigned char* cur = line + 1;
    unfilter_row(ft, cur, prev, bpp, row_bytes);
    prev = cur;
    unsigned char* out = dst + (size_t)y * W * 3;
    if (ctype == 2) {
      std::memcpy(out, cur, (size_t)W * 3);
    } else if (ctype == 6) {
      for (int x = 0; x < W; ++x) { out[3 * x] = cur[4 * x]; out[3 * x + 1] = cur[4 * x + 1]; out[3 * x + 2] = cur[4 * x + 2]; }
    } else if (ctype == 4) {
      for (int x = 0; x < W; ++x) { const unsigned char g = cur[2 * x]; out[3 * x] = g; out[3 * x + 1] = g; out[3 * x + 2] = g; }
    } else {   // gray or palette samples of 1 / 2 / 4 / 8 bits, most significant bits first
      const int maxv = (1 << depth) - 1;
      for (int x = 0; x < W; ++x) {
        int v;
        if (depth == 8) v = cur[x];
        else {
          const int per = 8 / depth, byte = x / per, sh = (per - 1 - x % per) * depth;
          v = (cur[byte] >> sh) & maxv;
        }
        if (ctype == 3) {
          out[3 * x] = palette[3 * v]; out[3 * x + 1] = palette[3 * v + 1]; out[3 * x + 2] = palette[3 * v + 2];
        } else {
          const unsigned char g = (unsigned char)(depth == 8 ? v : v * 255 / maxv);   // 1/2/4-bit gray scaled to 0..255
          out[3 * x] = g; out[3 * x + 1] = g; out[3 * x + 2] = g;
        }
      }
    }
  }
  return {KFN_PNG_OK, std::string()};
}

}  // namespace

// Decodes n PNG files into dst [n][H][W][3] (uint8 RGB) on `threads` host threads (<= 0: one per file, at most the
// hardware's).  status [n] (optional) receives KFN_PNG_OK / KFN_PNG_UNSUPPORTED (interlaced or 16-bit: the caller's
// fallback decoder should take that file; its frame in dst is untouched) / KFN_PNG_ERROR.  Returns KFN_OK when no file
// is in error (unsupported files do not fail the call), else KFN_ERR_ARG with kfn_last_error() naming the first bad file.
extern "C" int kfn_decode_png_rgb8(const char* const* paths, int n, int H, int W, unsigned char* dst, int* status, int threads) {
  KFN_REQUIRE(n >= 0 && (n == 0 || (paths && dst)) && H > 0 && W > 0, "kfn_decode_png_rgb8: bad argument (n=%d, H=%d, W=%d)", n, H, W);
  if (n == 0) return KFN_OK;
  for (int i = 0; i < n; ++i) KFN_REQUIRE(paths[i] != nullptr, "kfn_decode_png_rgb8: paths[%d] is null", i);
  std::vector<PngResult> res((size_t)n);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > n) nt = n;
  std::atomic<int> next{0};
  const size_t frame = (size_t)H * W * 3;
  auto worker = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) return;
      try {
        res[(size_t)i] = decode_one(paths[i], H, W, dst + (size_t)i * frame);
      } catch (const std::exception& e) {      // (std::bad_alloc: no exception crosses the C ABI or a thread boundary)
        res[(size_t)i] = {KFN_PNG_ERROR, std::string(paths[i]) + ": " + e.what()};
      }
    }
  };
  if (nt == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  int first_bad = -1;
  for (int i = 0; i < n; ++i) {
    if (status) status[i] = res[(size_t)i].status;
    if (res[(size_t)i].status == KFN_PNG_ERROR && first_bad < 0) first_bad = i;
  }
  if (first_bad >= 0) return kfn::fail(KFN_ERR_ARG, "kfn_decode_png_rgb8: %s", res[(size_t)first_bad].what.c_str());
  return KFN_OK;
}
