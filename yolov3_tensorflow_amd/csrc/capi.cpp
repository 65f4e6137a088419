// Error plumbing and version entry points of the C-ABI (include/yolov3_amd.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void yolo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int yolo_abi_version(void) { return YOLO_ABI_VERSION; }
extern "C" const char* yolo_last_error(void) { return g_err; }
extern "C" int yolo_abi_dtype(void) {
#ifdef YOLO_FP16
  return YOLO_DTYPE_FP16;
#else
  return YOLO_DTYPE_BF16;
#endif
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a HOST buffer, slicing-by-8: the checksum of TensorFlow's checkpoint files
// (tensor bundle entries, SSTable block trailers; reference trainer.py:47-67,90-91 read / write them through TensorFlow).  seed = the crc of
// the preceding bytes (0 to start): crc32c(a || b) == yolo_crc32c(b, nb, yolo_crc32c(a, na, 0)).
static uint32_t g_crc_tab[8][256];
static bool g_crc_ready = false;
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
  g_crc_ready = true;
}

extern "C" uint32_t yolo_crc32c(const void* data, size_t n, uint32_t seed) {
  if (!g_crc_ready) crc_init();
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~seed;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t w;
    __builtin_memcpy(&w, p, 8);
    w ^= c;                                     // little-endian host (gfx950 hosts are x86-64)
    c = g_crc_tab[7][w & 0xFF] ^ g_crc_tab[6][(w >> 8) & 0xFF] ^ g_crc_tab[5][(w >> 16) & 0xFF] ^ g_crc_tab[4][(w >> 24) & 0xFF] ^
        g_crc_tab[3][(w >> 32) & 0xFF] ^ g_crc_tab[2][(w >> 40) & 0xFF] ^ g_crc_tab[1][(w >> 48) & 0xFF] ^ g_crc_tab[0][(w >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}
