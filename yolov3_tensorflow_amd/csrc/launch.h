// Launch sequencer: every kernel launch of the library goes through yolo_launch (the hipLaunchKernelGGL spelling in the sources is
// redirected here).  Normally it just launches.  While a sequence is being RECORDED (yolo_seq_begin .. yolo_seq_end) it also stores the
// launch -- kernel, grid, block, dynamic LDS, stream and a private copy of the argument values -- and yolo_seq_run replays the stored
// list with one C call: the training step is a fixed sequence of ~270 launches over static buffers on two or three streams, and enqueueing
// it from Python + ctypes costs ~2.8 ms per step against ~4.7 ms of GPU time.  Cross-stream edges (yolo_seq_fork: "stream B waits for
// what is queued on stream A now") are recorded as event record + stream wait on an event owned by the sequence, so the three-stream
// schedule replays as recorded (hipGraph replay serialises the forked weight-gradient branch on this runtime).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

struct YoloSeqItem {
  int kind;                 // 0 kernel launch, 1 event record on `stream`, 2 `stream` waits for `event`
  const void* fn;
  dim3 grid, block;
  size_t lds;
  hipStream_t stream;
  hipEvent_t event;
  std::vector<unsigned char> blob;      // argument values, each at its natural alignment
  std::vector<unsigned> offs;           // offset of every argument in blob
};

struct YoloSeq {
  std::vector<YoloSeqItem> items;
  std::vector<hipEvent_t> events;
};

YoloSeq* yolo_seq_recording();          // the sequence being recorded, or nullptr (capi.cpp)

template <typename T>
inline void yolo_seq_pack(YoloSeqItem& it, const T& v) {
  size_t off = (it.blob.size() + alignof(T) - 1) / alignof(T) * alignof(T);
  it.blob.resize(off + sizeof(T));
  std::memcpy(it.blob.data() + off, &v, sizeof(T));
  it.offs.push_back((unsigned)off);
}

template <typename... K, typename... A>
inline void yolo_launch(void (*kernel)(K...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A&&... a) {
  static_assert(sizeof...(K) == sizeof...(A), "kernel argument count");
  static_assert(sizeof...(K) <= 64, "yolo_seq_run (capi.cpp) replays launches through a 64-entry argument pointer array");
  std::tuple<std::remove_cv_t<std::remove_reference_t<K>>...> args{static_cast<std::remove_cv_t<std::remove_reference_t<K>>>(std::forward<A>(a))...};
  void* ptrs[sizeof...(K) ? sizeof...(K) : 1];
  std::apply([&](auto&... x) { size_t i = 0; ((ptrs[i++] = (void*)&x), ...); }, args);
  if (YoloSeq* s = yolo_seq_recording()) {
    YoloSeqItem it;
    it.kind = 0; it.fn = reinterpret_cast<const void*>(kernel); it.grid = grid; it.block = block; it.lds = lds; it.stream = stream; it.event = nullptr;
    it.blob.reserve(256);
    std::apply([&](auto&... x) { (yolo_seq_pack(it, x), ...); }, args);
    s->items.push_back(std::move(it));
  }
  (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, stream);
}

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) yolo_launch(kernel, grid, block, lds, stream, ##__VA_ARGS__)
