// Error plumbing and version entry points of the C-ABI (include/yolov3_amd.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

// ---- launch sequencer (launch.h) ----
static YoloSeq* g_seq_rec = nullptr;
static std::vector<YoloSeq*> g_seqs;
YoloSeq* yolo_seq_recording() { return g_seq_rec; }

extern "C" int yolo_seq_begin(void) {
  YOLO_CHECK_ARG(g_seq_rec == nullptr, "a sequence is already being recorded");
  g_seq_rec = new YoloSeq();
  g_seqs.push_back(g_seq_rec);
  return (int)g_seqs.size() - 1;
}

extern "C" int yolo_seq_mark(void) { return g_seq_rec ? (int)g_seq_rec->items.size() : YOLO_ERR_INVALID_ARG; }

extern "C" int yolo_seq_end(void) {
  YOLO_CHECK_ARG(g_seq_rec != nullptr, "no sequence is being recorded");
  const int n = (int)g_seq_rec->items.size();
  g_seq_rec = nullptr;
  return n;
}

// Scope of the release an edge's event performs when it is recorded.  yolo_seq_fork: the runtime's default event -- a system-scope
// writeback / invalidate at the record, what a reader OUTSIDE this device's kernels needs (a copy engine, the host, a peer GPU: the
// gradient exchange's stream waits through these).  yolo_seq_fork_local: an edge between two streams whose consumers are kernels of the SAME
// device.  Every kernel dispatch carries its own agent-scope release / acquire (what makes consecutive kernels of one stream see each
// other's stores across XCDs), so such an event only has to ORDER the two queues: hipEventDisableSystemFence.  Measured on the training
// step (profiles/r04_fork_fence_ab.txt): +1.1 % (8.33 k against 8.24 k images/s; hipEventReleaseToDevice +0.35 %).
// YOLO_FORK_FENCE=system makes the local edges default events too, =device gives them the device-scope release.
static unsigned local_event_flags() {
  static const unsigned flags = [] {
    const char* v = getenv("YOLO_FORK_FENCE");
    unsigned f = (unsigned)hipEventDisableTiming;
    if (v && !strcmp(v, "system")) return f;
    if (v && !strcmp(v, "device")) return f | (unsigned)hipEventReleaseToDevice;
    return f | (unsigned)hipEventDisableSystemFence;
  }();
  return flags;
}

static int seq_fork(void* from_stream, void* to_stream, unsigned flags) {
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, flags);
  if (e != hipSuccess) { yolo_set_error("hipEventCreateWithFlags: %s", hipGetErrorString(e)); return (int)e; }
  if ((e = hipEventRecord(ev, (hipStream_t)from_stream)) != hipSuccess || (e = hipStreamWaitEvent((hipStream_t)to_stream, ev, 0)) != hipSuccess) {
    yolo_set_error("yolo_seq_fork: %s", hipGetErrorString(e));
    (void)hipEventDestroy(ev);
    return (int)e;
  }
  if (g_seq_rec) {                                   // the sequence keeps the event: the edge is replayed on it
    YoloSeqItem rec; rec.kind = 1; rec.fn = nullptr; rec.lds = 0; rec.stream = (hipStream_t)from_stream; rec.event = ev;
    YoloSeqItem wait = rec; wait.kind = 2; wait.stream = (hipStream_t)to_stream;
    g_seq_rec->items.push_back(rec);
    g_seq_rec->items.push_back(wait);
    g_seq_rec->events.push_back(ev);
  } else {
    (void)hipEventDestroy(ev);                       // released by the runtime once the recorded work has completed
  }
  return YOLO_OK;
}

extern "C" int yolo_seq_fork(void* from_stream, void* to_stream) { return seq_fork(from_stream, to_stream, (unsigned)hipEventDisableTiming); }
extern "C" int yolo_seq_fork_local(void* from_stream, void* to_stream) { return seq_fork(from_stream, to_stream, local_event_flags()); }

extern "C" int yolo_seq_run(int seq, int begin, int end) {
  YOLO_CHECK_ARG(seq >= 0 && seq < (int)g_seqs.size() && g_seqs[seq] && g_seqs[seq] != g_seq_rec, "bad sequence id");
  YoloSeq& s = *g_seqs[seq];
  YOLO_CHECK_ARG(begin >= 0 && begin <= end && end <= (int)s.items.size(), "bad item range");
  void* ptrs[64];
  for (int i = begin; i < end; ++i) {
    YoloSeqItem& it = s.items[i];
    hipError_t e;
    if (it.kind == 0) {
      const size_t n = it.offs.size();
      for (size_t k = 0; k < n; ++k) ptrs[k] = it.blob.data() + it.offs[k];
      e = hipLaunchKernel(it.fn, it.grid, it.block, ptrs, it.lds, it.stream);
    } else if (it.kind == 1) {
      e = hipEventRecord(it.event, it.stream);
    } else {
      e = hipStreamWaitEvent(it.stream, it.event, 0);
    }
    if (e != hipSuccess) { yolo_set_error("yolo_seq_run: item %d: %s", i, hipGetErrorString(e)); return (int)e; }
  }
  return YOLO_OK;
}

extern "C" int yolo_seq_free(int seq) {
  YOLO_CHECK_ARG(seq >= 0 && seq < (int)g_seqs.size() && g_seqs[seq] && g_seqs[seq] != g_seq_rec, "bad sequence id");
  for (hipEvent_t ev : g_seqs[seq]->events) (void)hipEventDestroy(ev);
  delete g_seqs[seq];
  g_seqs[seq] = nullptr;
  return YOLO_OK;
}
