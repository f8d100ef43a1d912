// Frame feeder: the read-ahead half of the reference's `cap.read()` (geotrax/extract.py:146) for sources whose
// frames sit uncompressed in a file (.y4m, .npy) or are produced by a host thread (image folders, a decoder).
//
// The reference reads one frame synchronously at the top of every loop iteration. At 4K that is 12.4 MB (I420)
// or 24.9 MB (BGR) per frame; done on the thread that also drives the detector it is the whole pipeline's pace.
// Here the read, the PCIe transfer and the colour conversion run beside the pipeline:
//
//   reader threads   pread() whole frames straight into a ring of pinned (hipHostMalloc) slots -- one copy out
//                    of the page cache, no intermediate bytes object, frames of a batch read in parallel;
//   uploader thread  hipMemcpyAsync of every slot to its place in a ring of device batches on the feeder's own
//                    stream (pinned memory: the copy engines do it), I420 slots through yuv420_to_bgr_kernel on the
//                    same stream, one event per batch;
//   consumer         gtx_feeder_next() hands out batches in clip order as device pointers; gtx_feeder_wait() makes the
//                    consumer's stream wait for the batch's event (no host thread blocks on the transfer);
//                    gtx_feeder_release() returns slots once the consumer's pass over them is complete.
//
// Push mode (no file): the caller's own thread hands host frames to gtx_feeder_push(), which copies them into the
// pinned ring; everything behind that is the same.
#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../include/gtx.h"
#include "api_guard.hpp"
#include "common.hpp"
#include "detector.hpp"
#include "geometry.hpp"

struct gtx_feeder {
  gtx_ctx ctx;                       // device + the copy stream
  bool own_stream = true;            // false: the stream belongs to the context the feeder was created on
  int h = 0, w = 0, kind = 0;        // kind 0: BGR u8 frames, 1: I420 planes
  int B = 1, ring = 4;
  size_t src_bytes = 0, bgr_bytes = 0;
  uint8_t* pinned = nullptr;         // [ring][B][src_bytes], hipHostMalloc
  gtx::DevBuf dev, dev_yuv;          // [ring][B][bgr_bytes]; [ring][B][src_bytes] (I420 only)
  std::vector<hipEvent_t> ev;        // per ring slot: the batch's uploads (and conversions) are complete

  int fd = -1;
  std::vector<int64_t> offsets;      // file mode: payload offset of every frame to deliver, in delivery order
  std::vector<const uint8_t*> mem;   // memory mode: the frames as host pointers (the caller keeps them alive), in delivery order
  std::vector<std::thread> readers;
  std::thread uploader;

  std::mutex m;
  std::condition_variable cv;
  int64_t n_frames = -1;             // -1: not known yet (push mode before gtx_feeder_finish)
  int64_t next_read = 0;             // next frame a reader thread takes (file mode) / next frame pushed
  std::vector<int64_t> slot_frame;   // [ring*B]: frame whose bytes the pinned slot holds, -1 = none
  int64_t issued = 0;                // batches whose copies are on the stream (event recorded)
  int64_t handed = 0;                // batches returned by next()
  int64_t released = 0;              // batches whose slots may be overwritten
  bool stop = false;
  std::string error;                 // first failure of a worker thread

  int64_t n_batches() const { return n_frames < 0 ? -1 : (n_frames + B - 1) / B; }
  size_t slot_of(int64_t frame) const { return (size_t)((frame / B) % ring) * B + (size_t)(frame % B); }

  ~gtx_feeder() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : readers) if (t.joinable()) t.join();
    if (uploader.joinable()) uploader.join();
    (void)hipSetDevice(ctx.device);
    if (ctx.stream) (void)hipStreamSynchronize(ctx.stream);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (pinned) (void)hipHostFree(pinned);
    if (fd >= 0) ::close(fd);
    if (!own_stream) ctx.stream = nullptr;   // ~gtx_ctx must not destroy a borrowed stream
  }

  void set_error(const std::string& what) {
    std::lock_guard<std::mutex> lk(m);
    if (error.empty()) error = what;
    stop = true;
    cv.notify_all();
  }

  // reader thread: frames in file order, each into the pinned slot of its (batch, position)
  void read_loop() {
    for (;;) {
      int64_t i;
      {
        std::unique_lock<std::mutex> lk(m);
        if (stop || next_read >= n_frames) return;
        i = next_read++;
        cv.wait(lk, [&] { return stop || i / B < released + ring; });
        if (stop) return;
      }
      uint8_t* dst = pinned + slot_of(i) * src_bytes;
      size_t got = 0;
      if (!mem.empty()) {                       // memory mode: one copy into the pinned slot, like the page-cache copy of a pread
        memcpy(dst, mem[(size_t)i], src_bytes);
        got = src_bytes;
      }
      while (got < src_bytes) {
        const ssize_t r = ::pread(fd, dst + got, src_bytes - got, (off_t)(offsets[(size_t)i] + (int64_t)got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) {
          set_error("frame " + std::to_string(i) + " could not be read (" + (r == 0 ? std::string("file ends early") : std::string(strerror(errno))) + ")");
          return;
        }
        got += (size_t)r;
      }
      {
        std::lock_guard<std::mutex> lk(m);
        slot_frame[slot_of(i)] = i;
      }
      cv.notify_all();
    }
  }

  // uploader thread: slots in frame order onto the copy stream, one event per batch (a shorter last batch gets its event
  // when the end of the source is known)
  void upload_loop() {
    if (hipSetDevice(ctx.device) != hipSuccess) return set_error("hipSetDevice failed on the feeder's upload thread");
    try {
      for (int64_t i = 0;; ++i) {
        bool end;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return stop || (n_frames >= 0 && i >= n_frames) || slot_frame[slot_of(i)] == i; });
          if (stop) return;
          end = n_frames >= 0 && i >= n_frames;
        }
        if (end && i % B == 0) return;
        if (!end) {
          const size_t s = slot_of(i);
          uint8_t* bgr = dev.as<uint8_t>() + s * bgr_bytes;
          if (kind == 1) {
            uint8_t* yuv = dev_yuv.as<uint8_t>() + s * src_bytes;
            GTX_HIP(hipMemcpyAsync(yuv, pinned + s * src_bytes, src_bytes, hipMemcpyHostToDevice, ctx.stream));
            gtx::yuv420_to_bgr_dev(&ctx, yuv, h, w, bgr);
          } else {
            GTX_HIP(hipMemcpyAsync(bgr, pinned + s * src_bytes, src_bytes, hipMemcpyHostToDevice, ctx.stream));
          }
        }
        if (end || i % B == B - 1) {
          const int64_t j = end ? (i - 1) / B : i / B;
          GTX_HIP(hipEventRecord(ev[(size_t)(j % ring)], ctx.stream));
          {
            std::lock_guard<std::mutex> lk(m);
            issued = j + 1;
          }
          cv.notify_all();
          if (end) return;
        }
      }
    } catch (const std::exception& e) {
      set_error(e.what());
    }
  }
};

using gtx::guarded;

extern "C" {

namespace {
void feeder_create(int device, hipStream_t borrowed, int h, int w, int kind, int batch, int ring, gtx_feeder** out) {
  if (!out) gtx::fail(GTX_ERR_INVALID, "out is NULL");
  if (h <= 0 || w <= 0 || (kind != 0 && kind != 1) || batch < 1 || ring < 2)
    gtx::fail(GTX_ERR_INVALID, "feeder: bad geometry (h %d, w %d, kind %d, batch %d, ring %d)", h, w, kind, batch, ring);
  GTX_HIP(hipSetDevice(device));
  std::unique_ptr<gtx_feeder> f(new gtx_feeder);
  f->ctx.device = device;
  f->h = h, f->w = w, f->kind = kind, f->B = batch, f->ring = ring;
  f->bgr_bytes = (size_t)h * w * 3;
  f->src_bytes = kind == 1 ? (size_t)h * w + 2 * (size_t)((h + 1) / 2) * ((w + 1) / 2) : f->bgr_bytes;
  if (borrowed) {
    f->ctx.stream = borrowed;
    f->own_stream = false;
  } else {
    GTX_HIP(hipStreamCreateWithFlags(&f->ctx.stream, hipStreamNonBlocking));
  }
  const size_t slots = (size_t)ring * batch;
  GTX_HIP(hipHostMalloc((void**)&f->pinned, slots * f->src_bytes, hipHostMallocDefault));
  f->dev.alloc(slots * f->bgr_bytes);
  if (kind == 1) f->dev_yuv.alloc(slots * f->src_bytes);
  f->ev.resize((size_t)ring);
  for (auto& e : f->ev) GTX_HIP(hipEventCreateWithFlags(&e, gtx::wait_event_flags(false)));
  f->slot_frame.assign(slots, -1);
  *out = f.release();
}
}  // namespace

int gtx_feeder_create(int device, int h, int w, int kind, int batch, int ring, gtx_feeder** out) {
  return guarded([&] { feeder_create(device, nullptr, h, w, kind, batch, ring, out); });
}

int gtx_feeder_create_on(gtx_ctx* copy_ctx, int h, int w, int kind, int batch, int ring, gtx_feeder** out) {
  return guarded([&] {
    if (!copy_ctx) gtx::fail(GTX_ERR_INVALID, "copy_ctx is NULL");
    feeder_create(copy_ctx->device, copy_ctx->stream, h, w, kind, batch, ring, out);
  });
}

void gtx_feeder_destroy(gtx_feeder* f) { delete f; }

int gtx_feeder_open_file(gtx_feeder* f, const char* path, const int64_t* offsets, int64_t n_frames, int n_threads) {
  return guarded([&] {
    if (!f || !path || (n_frames > 0 && !offsets)) gtx::fail(GTX_ERR_INVALID, "feeder_open_file: NULL argument");
    if (f->fd >= 0 || f->uploader.joinable()) gtx::fail(GTX_ERR_STATE, "feeder already has a source");
    f->fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (f->fd < 0) gtx::fail(GTX_ERR_INVALID, "feeder: cannot open '%s': %s", path, strerror(errno));
    f->offsets.assign(offsets, offsets + n_frames);
    f->n_frames = n_frames;
    f->uploader = std::thread([f] { f->upload_loop(); });
    for (int t = 0; t < std::max(1, std::min(n_threads, 16)); ++t) f->readers.emplace_back([f] { f->read_loop(); });
  });
}

int gtx_feeder_open_memory(gtx_feeder* f, const void* const* frames, int64_t n_frames, int n_threads) {
  return guarded([&] {
    if (!f || (n_frames > 0 && !frames)) gtx::fail(GTX_ERR_INVALID, "feeder_open_memory: NULL argument");
    if (f->fd >= 0 || f->uploader.joinable()) gtx::fail(GTX_ERR_STATE, "feeder already has a source");
    for (int64_t i = 0; i < n_frames; ++i) {
      if (!frames[i]) gtx::fail(GTX_ERR_INVALID, "feeder_open_memory: frame %lld is NULL", (long long)i);
      f->mem.push_back(static_cast<const uint8_t*>(frames[i]));
    }
    f->n_frames = n_frames;
    f->uploader = std::thread([f] { f->upload_loop(); });
    for (int t = 0; t < std::max(1, std::min(n_threads, 16)); ++t) f->readers.emplace_back([f] { f->read_loop(); });
  });
}

int gtx_feeder_open_push(gtx_feeder* f) {
  return guarded([&] {
    if (!f) gtx::fail(GTX_ERR_INVALID, "feeder is NULL");
    if (f->fd >= 0 || f->uploader.joinable()) gtx::fail(GTX_ERR_STATE, "feeder already has a source");
    f->uploader = std::thread([f] { f->upload_loop(); });
  });
}

int gtx_feeder_push(gtx_feeder* f, const void* frame, size_t bytes) {
  return guarded([&] {
    if (!f || !frame) gtx::fail(GTX_ERR_INVALID, "feeder_push: NULL argument");
    if (bytes != f->src_bytes) gtx::fail(GTX_ERR_INVALID, "feeder_push: frame has %zu bytes, the feeder was built for %zu", bytes, f->src_bytes);
    int64_t i;
    {
      std::unique_lock<std::mutex> lk(f->m);
      if (f->n_frames >= 0) gtx::fail(GTX_ERR_STATE, "feeder_push after feeder_finish");
      i = f->next_read;
      f->cv.wait(lk, [&] { return f->stop || i / f->B < f->released + f->ring; });
      if (f->stop) gtx::fail(GTX_ERR_STATE, "feeder stopped: %s", f->error.c_str());
      f->next_read = i + 1;
    }
    memcpy(f->pinned + f->slot_of(i) * f->src_bytes, frame, bytes);
    {
      std::lock_guard<std::mutex> lk(f->m);
      f->slot_frame[f->slot_of(i)] = i;
    }
    f->cv.notify_all();
  });
}

int gtx_feeder_push_at(gtx_feeder* f, int64_t i, const void* frame, size_t bytes) {
  return guarded([&] {
    if (!f || !frame || i < 0) gtx::fail(GTX_ERR_INVALID, "feeder_push_at: bad argument");
    if (bytes != f->src_bytes) gtx::fail(GTX_ERR_INVALID, "feeder_push_at: frame has %zu bytes, the feeder was built for %zu", bytes, f->src_bytes);
    {
      std::unique_lock<std::mutex> lk(f->m);
      if (f->n_frames >= 0) gtx::fail(GTX_ERR_STATE, "feeder_push_at after feeder_finish");
      f->cv.wait(lk, [&] { return f->stop || i / f->B < f->released + f->ring; });
      if (f->stop) gtx::fail(GTX_ERR_STATE, "feeder stopped: %s", f->error.c_str());
      if (i / f->B < f->released) gtx::fail(GTX_ERR_STATE, "feeder_push_at: frame %lld belongs to a batch that was already consumed", (long long)i);
    }
    memcpy(f->pinned + f->slot_of(i) * f->src_bytes, frame, bytes);
    {
      std::lock_guard<std::mutex> lk(f->m);
      f->slot_frame[f->slot_of(i)] = i;
      if (i + 1 > f->next_read) f->next_read = i + 1;
    }
    f->cv.notify_all();
  });
}

int gtx_feeder_finish(gtx_feeder* f) {
  return guarded([&] {
    if (!f) gtx::fail(GTX_ERR_INVALID, "feeder is NULL");
    {
      std::lock_guard<std::mutex> lk(f->m);
      if (f->n_frames < 0) f->n_frames = f->next_read;
    }
    f->cv.notify_all();
  });
}

int gtx_feeder_stop(gtx_feeder* f) {
  return guarded([&] {
    if (!f) gtx::fail(GTX_ERR_INVALID, "feeder is NULL");
    {
      std::lock_guard<std::mutex> lk(f->m);
      if (f->error.empty()) f->error = "the feeder was stopped";
      f->stop = true;
    }
    f->cv.notify_all();
  });
}

int gtx_feeder_next(gtx_feeder* f, void** dptr, int* n, int64_t* batch_index) {
  return guarded([&] {
    if (!f || !dptr || !n) gtx::fail(GTX_ERR_INVALID, "feeder_next: NULL argument");
    std::unique_lock<std::mutex> lk(f->m);
    f->cv.wait(lk, [&] { return !f->error.empty() || f->issued > f->handed || (f->n_frames >= 0 && f->handed >= f->n_batches()); });
    if (f->issued > f->handed) {                       // batches that arrived before a failure are still delivered
      const int64_t j = f->handed++;
      *dptr = f->dev.as<uint8_t>() + (size_t)(j % f->ring) * f->B * f->bgr_bytes;
      *n = (int)((f->n_frames >= 0 && (j + 1) * f->B > f->n_frames) ? f->n_frames - j * f->B : f->B);
      if (batch_index) *batch_index = j;
      return;
    }
    if (!f->error.empty()) gtx::fail(GTX_ERR_INTERNAL, "%s", f->error.c_str());
    *dptr = nullptr;
    *n = 0;
    if (batch_index) *batch_index = f->handed;
  });
}

int gtx_feeder_wait(gtx_feeder* f, int64_t batch_index, gtx_ctx* consumer) {
  return guarded([&] {
    if (!f) gtx::fail(GTX_ERR_INVALID, "feeder is NULL");
    hipEvent_t e;
    {
      std::lock_guard<std::mutex> lk(f->m);
      if (batch_index < 0 || batch_index >= f->issued || batch_index < f->released)
        gtx::fail(GTX_ERR_STATE, "feeder_wait: batch %lld is not resident (issued %lld, released %lld)", (long long)batch_index, (long long)f->issued, (long long)f->released);
      e = f->ev[(size_t)(batch_index % f->ring)];
    }
    GTX_HIP(hipSetDevice(f->ctx.device));
    if (consumer) GTX_HIP(hipStreamWaitEvent(consumer->stream, e, 0));
    else GTX_HIP(hipEventSynchronize(e));
  });
}

int gtx_feeder_release(gtx_feeder* f, int64_t n_batches) {
  return guarded([&] {
    if (!f) gtx::fail(GTX_ERR_INVALID, "feeder is NULL");
    {
      std::lock_guard<std::mutex> lk(f->m);
      if (n_batches > f->handed) gtx::fail(GTX_ERR_STATE, "feeder_release(%lld): only %lld batches were handed out", (long long)n_batches, (long long)f->handed);
      if (n_batches > f->released) f->released = n_batches;
    }
    f->cv.notify_all();
  });
}

}  // extern "C"
