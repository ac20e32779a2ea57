// extern "C" surface of libhdp_hip.so (see include/hdp_hip.h): lifecycle, device-memory
// plumbing, plan construction for the metrics pass, and the host-pointer entry points that
// stage caller buffers through HBM in bounded chunks.
#include "hdp_internal.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace hdp {

static thread_local std::string g_err;
static int g_device = -1;
static hipStream_t g_stream = nullptr;
static std::string g_info;

int set_error(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

hipStream_t default_stream() { return g_stream; }
// Every extern "C" entry point passes through here first: besides the "hdp_init was called" check it binds the
// library's device to the CALLING thread (hipSetDevice is per thread; hdp_init only bound the thread that ran it,
// so a call from another host thread with LOCAL_RANK != 0 would otherwise allocate and launch on device 0).
bool device_ready() {
  if (g_device < 0) return false;
  // checked on EVERY call, not once per thread: the caller may have switched this thread to another device since
  // (torch.cuda.set_device), and the library's allocations, streams and launches all belong to g_device
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != g_device) {
    if (hipSetDevice(g_device) != hipSuccess) return false;
  }
  return true;
}

long long env_option(const char *name, long long dflt) {
  const char *v = getenv(name);
  if (!v || !*v) return dflt;
  static std::mutex mu;
  static std::vector<std::string> seen;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(seen.begin(), seen.end(), name) == seen.end()) {
      seen.emplace_back(name);
      fprintf(stderr, "[hdp] environment override %s=%s is in effect (kernel selector for tests and A/B runs; "
                      "results do not depend on it)\n", name, v);
    }
  }
  return atoll(v);
}

static hipStream_t pick(void *stream) { return stream ? static_cast<hipStream_t>(stream) : g_stream; }

// Pack an arbitrarily strided host array into [n_cells][T] contiguous floats.
static void pack_series(const float *x, int64_t c0, int64_t nc, int64_t T, int64_t sc, int64_t st,
                        float *dst) {
  if (st == 1) {
    for (int64_t c = 0; c < nc; ++c) std::memcpy(dst + c * T, x + (c0 + c) * sc, size_t(T) * 4);
    return;
  }
  // time-major (or general) input: walk time in the outer loop so reads stay sequential
  if (sc == 1) {
    constexpr int64_t TB = 256;
    for (int64_t t0 = 0; t0 < T; t0 += TB) {
      const int64_t t1 = std::min(T, t0 + TB);
      for (int64_t c = 0; c < nc; ++c) {
        float *d = dst + c * T;
        const float *s = x + (c0 + c);
        for (int64_t t = t0; t < t1; ++t) d[t] = s[t * st];
      }
    }
    return;
  }
  for (int64_t c = 0; c < nc; ++c)
    for (int64_t t = 0; t < T; ++t) dst[c * T + t] = x[(c0 + c) * sc + t * st];
}

// A freshly allocated result array (np.empty / np.zeros) is untouched virtual memory: a device-to-host copy into
// it takes one page fault per 4 KiB inside the copy (C2 thresholds: 77 ms instead of 34 for the 1.9 GB result).
// The chunk loops below therefore fault the NEXT chunk's part of the result from helper threads while the current
// chunk uploads, computes and downloads, and wait for a part before the copy that writes it is issued -- a part is
// never touched and written at the same time.  A touch rewrites the first byte of each page with its own value.
struct ResultPrefault {
  std::future<void> pending;
  void start(void *p, size_t bytes) {
    wait();
    if (!p || bytes < (size_t(8) << 20)) return;
    try {
      launch(p, bytes);
    } catch (...) {  // no helper thread available: the copy takes the faults itself, as before
      pending = std::future<void>();
    }
  }
  void launch(void *p, size_t bytes) {
    pending = std::async(std::launch::async, [p, bytes] {
      constexpr int kThreads = 4;
      constexpr size_t kPage = 4096;
      auto touch = [](volatile unsigned char *b, size_t n) {
        for (size_t i = 0; i < n; i += kPage) b[i] = b[i];
      };
      unsigned char *base = static_cast<unsigned char *>(p);
      const size_t part = ((bytes / kThreads) + kPage - 1) & ~(kPage - 1);
      std::thread th[kThreads];
      int started = 0;
      for (int k = 1; k < kThreads; ++k) {  // this thread takes part 0 (and whatever could not get a thread)
        const size_t a = size_t(k) * part;
        if (a >= bytes) break;
        try {
          th[started] = std::thread(touch, base + a, std::min(part, bytes - a));
          ++started;
        } catch (...) {
          touch(base + a, std::min(part, bytes - a));
        }
      }
      touch(base, std::min(part, bytes));
      for (int k = 0; k < started; ++k) th[k].join();
    });
  }
  void wait() {
    if (!pending.valid()) return;
    try {
      pending.get();
    } catch (...) {
    }
  }
  ~ResultPrefault() { wait(); }
};

// Device bytes a host entry point may spend on a resident copy of a complete time-major input.
static size_t whole_matrix_limit() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  return std::min<size_t>(free_b / 4, size_t(32) << 30);
}

static int64_t chunk_cells_for(int64_t n_cells, int64_t bytes_per_cell) {
  const int64_t budget = int64_t(1) << 30;  // ~1 GiB of device staging per chunk
  int64_t c = std::max<int64_t>(1, budget / std::max<int64_t>(1, bytes_per_cell));
  return std::min(c, n_cells);
}

// Bring series [c0, c0+nc) of a strided host array into dx [nc][T] on the device.
//  - time-contiguous rows: one plain copy;
//  - time-major (CMIP [time][cells], stride_cell == 1): strided 2-D copy of the [T][nc] slab into
//    `raw`, then a device transpose -- no host-side repacking of the bulk data;
//  - anything else: packed on the host.
struct SeriesUploader {
  DevBuf raw;
  std::vector<float> stage;
  bool whole = false;  // `raw` holds the complete time-major matrix [T][st]
  // Time-major input whose rows are dense over all n_cells series: one contiguous copy of the whole matrix (a
  // strided 2-D copy per chunk of cells runs at a sixth of the link rate from pageable memory) when it fits
  // `limit` bytes; the chunks are then transposed out of the resident copy.
  int prepare(const float *x, int64_t n_cells, int64_t T, int64_t sc, int64_t st, size_t limit, hipStream_t stream) {
    whole = false;
    const size_t bytes = size_t(T) * size_t(st) * 4;
    if (!(sc == 1 && st == n_cells && T > 1 && bytes <= limit)) return HDP_OK;
    HDP_HIP_TRY(raw.alloc(bytes));
    HDP_HIP_TRY(hipMemcpyAsync(raw.p, x, bytes, hipMemcpyHostToDevice, stream));
    whole = true;
    return HDP_OK;
  }
  int upload(const float *x, int64_t c0, int64_t nc, int64_t T, int64_t sc, int64_t st, float *dx,
             hipStream_t stream) {
    if (whole) return launch_transpose(raw.as<float>() + c0, st, T, nc, dx, stream);
    if (st == 1 && sc == T) {
      HDP_HIP_TRY(hipMemcpyAsync(dx, x + c0 * T, size_t(nc) * T * 4, hipMemcpyHostToDevice, stream));
      return HDP_OK;
    }
    if (sc == 1 && st >= nc) {
      if (raw.bytes < size_t(nc) * T * 4) HDP_HIP_TRY(raw.alloc(size_t(nc) * T * 4));
      HDP_HIP_TRY(hipMemcpy2DAsync(raw.p, size_t(nc) * 4, x + c0, size_t(st) * 4, size_t(nc) * 4, size_t(T),
                                   hipMemcpyHostToDevice, stream));
      return launch_transpose(raw.as<float>(), nc, T, nc, dx, stream);
    }
    if (stage.size() < size_t(nc) * T) stage.resize(size_t(nc) * T);
    pack_series(x, c0, nc, T, sc, st, stage.data());
    HDP_HIP_TRY(hipMemcpyAsync(dx, stage.data(), size_t(nc) * T * 4, hipMemcpyHostToDevice, stream));
    HDP_HIP_TRY(hipStreamSynchronize(stream));  // `stage` is reused by the next chunk
    return HDP_OK;
  }
};

// Upload of the NEXT chunk of series under the kernels and the download of the current one (inputs
// larger than HBM stream through the device in chunks / latitude bands).  A copy from pageable host memory blocks the
// calling thread while the runtime stages it through its pinned buffers, so the upload runs on a helper thread with its
// own stream into the other of two device buffers; the copy engine moves chunk k + 1 while the compute units work on
// chunk k.  The helper's error text travels back with its code (hdp_last_error is per thread).
struct AsyncUpload {
  hipStream_t stream = nullptr;
  std::future<int> pending;
  std::string err;
  ~AsyncUpload() {
    if (pending.valid()) {
      try { pending.get(); } catch (...) {}
    }
    if (stream) (void)hipStreamDestroy(stream);
  }
  int init() {
    HDP_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    return HDP_OK;
  }
  void start(SeriesUploader *up, const float *x, int64_t c0, int64_t nc, int64_t T, int64_t sc, int64_t st, float *dx) {
    const int dev = g_device;
    auto work = [this, up, x, c0, nc, T, sc, st, dx, dev]() -> int {
      if (hipSetDevice(dev) != hipSuccess) {
        err = "hipSetDevice failed on the upload thread";
        return HDP_EHIP;
      }
      int rc = up->upload(x, c0, nc, T, sc, st, dx, stream);
      if (rc == HDP_OK && hipStreamSynchronize(stream) != hipSuccess) rc = set_error(HDP_EHIP, "upload stream failed");
      if (rc != HDP_OK) err = g_err;
      return rc;
    };
    try {
      pending = std::async(std::launch::async, work);
    } catch (...) {  // no thread to be had: upload here, as before
      std::promise<int> pr;
      pr.set_value(work());
      pending = pr.get_future();
    }
  }
  int wait() {
    if (!pending.valid()) return HDP_OK;
    int rc = HDP_EHIP;
    try {
      rc = pending.get();
    } catch (...) {
      err = "the upload thread raised";
    }
    if (rc != HDP_OK) return set_error(rc, "%s", err.c_str());
    return HDP_OK;
  }
};

}  // namespace hdp

using namespace hdp;

extern "C" {

int hdp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hdp_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(HDP_ENODEV, "no HIP device visible (%s); libhdp_hip has no CPU fallback",
                     e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  HDP_REQUIRE(device >= 0 && device < n, HDP_EINVAL, "device %d outside [0, %d)", device, n);
  HDP_HIP_TRY(hipSetDevice(device));
  if (g_stream == nullptr || g_device != device) {
    if (g_stream) (void)hipStreamDestroy(g_stream);
    HDP_HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
  }
  g_device = device;
  hipDeviceProp_t prop;
  HDP_HIP_TRY(hipGetDeviceProperties(&prop, device));
  char buf[512];
  snprintf(buf, sizeof buf, "%s arch=%s CUs=%d LDS/block=%zu HBM=%.1f GiB", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, (size_t)prop.sharedMemPerBlock,
           double(prop.totalGlobalMem) / (1024.0 * 1024.0 * 1024.0));
  g_info = buf;
  return HDP_OK;
}

int hdp_shutdown(void) {
  if (g_stream) {
    (void)hipStreamSynchronize(g_stream);
    (void)hipStreamDestroy(g_stream);
  }
  g_stream = nullptr;
  g_device = -1;
  return HDP_OK;
}

const char *hdp_last_error(void) { return g_err.c_str(); }
const char *hdp_device_info(void) { return g_info.c_str(); }

void *hdp_dev_alloc(size_t bytes) {
  if (!device_ready()) {
    set_error(HDP_ENODEV, "hdp_init() has not selected a HIP device");
    return nullptr;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    set_error(HDP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

int hdp_dev_free(void *p) {
  if (p) HDP_HIP_TRY(hipFree(p));
  return HDP_OK;
}

int hdp_memcpy_h2d(void *dst, const void *src, size_t bytes) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return HDP_OK;
}

int hdp_memcpy_d2h(void *dst, const void *src, size_t bytes) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  HDP_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return HDP_OK;
}

int hdp_dev_memset(void *dst, int value, size_t bytes) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_HIP_TRY(hipMemsetAsync(dst, value, bytes, g_stream));
  return HDP_OK;
}

int hdp_sync(void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_HIP_TRY(hipStreamSynchronize(pick(stream)));
  return HDP_OK;
}

void *hdp_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreate(&ev) != hipSuccess) {
    set_error(HDP_EHIP, "hipEventCreate failed");
    return nullptr;
  }
  return ev;
}

int hdp_event_record(void *event, void *stream) {
  HDP_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), pick(stream)));
  return HDP_OK;
}

int hdp_event_elapsed_ms(void *start, void *stop, float *ms) {
  HDP_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  HDP_HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return HDP_OK;
}

int hdp_event_destroy(void *event) {
  if (event) HDP_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return HDP_OK;
}

// ---- thresholds ----------------------------------------------------------------------------------

int hdp_thresholds_f32_dev(const hdp_threshold_plan *plan, const float *x_dev, int64_t n_cells,
                           double *out_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan && (n_cells == 0 || (x_dev && out_dev)), HDP_EINVAL, "NULL plan or buffer");
  HDP_REQUIRE(n_cells >= 0, HDP_EINVAL, "negative n_cells");
  return launch_thresholds(plan, x_dev, n_cells, out_dev, pick(stream));
}

int hdp_thresholds_f32(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell,
                       int64_t stride_time, const int64_t *time_index, int64_t n_doy, int64_t S,
                       const int32_t *cols, int64_t W, const double *q, int64_t P, double *out) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_cells >= 0, HDP_EINVAL, "negative n_cells");
  HDP_REQUIRE(n_cells == 0 || (x && out), HDP_EINVAL, "NULL buffer");
  hdp_threshold_plan *plan = nullptr;
  int rc = hdp_threshold_plan_create(time_index, n_doy, S, cols, W, q, P, T, &plan);
  if (rc != HDP_OK) return rc;
  std::unique_ptr<hdp_threshold_plan> guard(plan);
  if (n_cells == 0) return HDP_OK;
  const int64_t chunk = chunk_cells_for(n_cells, T * 4 + n_doy * P * 8);
  DevBuf dx[2], dout, dref;
  HDP_HIP_TRY(dx[0].alloc(size_t(chunk) * T * 4));
  if (chunk < n_cells) HDP_HIP_TRY(dx[1].alloc(size_t(chunk) * T * 4));  // the next chunk uploads under this one's kernels
  HDP_HIP_TRY(dout.alloc(size_t(chunk) * n_doy * P * 8));
  HDP_HIP_TRY(dref.alloc(size_t(chunk) * n_doy * P * 8));
  SeriesUploader up;
  rc = up.prepare(x, n_cells, T, stride_cell, stride_time, whole_matrix_limit(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));  // the upload stream reads what prepare() put on the device
  AsyncUpload au;
  rc = au.init();
  if (rc != HDP_OK) return rc;
  ResultPrefault pre;
  pre.start(out, size_t(std::min(chunk, n_cells)) * n_doy * P * 8);
  au.start(&up, x, 0, std::min(chunk, n_cells), T, stride_cell, stride_time, dx[0].as<float>());
  int64_t k = 0;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk, ++k) {
    const int64_t nc = std::min(chunk, n_cells - c0);
    rc = au.wait();  // chunk k is on the device
    if (rc != HDP_OK) return rc;
    // dx[(k + 1) & 1] was last read by the kernels of chunk k - 1, finished before the previous iteration ended
    if (c0 + nc < n_cells)
      au.start(&up, x, c0 + nc, std::min(chunk, n_cells - c0 - nc), T, stride_cell, stride_time, dx[(k + 1) & 1].as<float>());
    rc = launch_thresholds(plan, dx[k & 1].as<float>(), nc, dout.as<double>(), g_stream);
    if (rc != HDP_OK) return rc;
    // device layout [cell][P][n_doy] -> the reference's (cell, doy, percentile)
    rc = launch_swap_last2_f64(dout.as<double>(), nc, P, n_doy, dref.as<double>(), g_stream);
    if (rc != HDP_OK) return rc;
    pre.wait();  // this chunk's part of `out` is resident
    if (c0 + nc < n_cells)
      pre.start(out + (c0 + nc) * n_doy * P, size_t(std::min(chunk, n_cells - c0 - nc)) * n_doy * P * 8);
    HDP_HIP_TRY(hipMemcpyAsync(out + c0 * n_doy * P, dref.p, size_t(nc) * n_doy * P * 8,
                               hipMemcpyDeviceToHost, g_stream));
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

int hdp_percentiles_table_f32(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell,
                              int64_t stride_time, const int64_t *win, int64_t n_doy, int64_t B,
                              const double *q, int64_t P, double *out) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_cells >= 0 && T > 0 && n_doy > 0 && B > 0 && P > 0, HDP_EINVAL, "bad sizes");
  HDP_REQUIRE(x && win && q && out, HDP_EINVAL, "NULL buffer");
  for (int64_t i = 0; i < n_doy * B; ++i)
    HDP_REQUIRE(win[i] >= -T && win[i] < T, HDP_EINVAL, "window index %lld out of range",
                (long long)win[i]);
  std::vector<QuantileParam> qp(P);
  std::vector<int32_t> klo(P), khi(P);
  for (int64_t p = 0; p < P; ++p) {
    int64_t a, b;
    if (quantile_param(q[p], B, &qp[p], &a, &b) != HDP_OK)
      return set_error(HDP_EQUANT, "Quantiles must be in the range [0, 1]");
    klo[p] = (int32_t)a;
    khi[p] = (int32_t)b;
  }
  DevBuf dwin, dqp, dklo, dkhi, dx, dout;
  HDP_HIP_TRY(dwin.upload(win, size_t(n_doy) * B * 8));
  HDP_HIP_TRY(dqp.upload(qp.data(), size_t(P) * sizeof(QuantileParam)));
  HDP_HIP_TRY(dklo.upload(klo.data(), size_t(P) * 4));
  HDP_HIP_TRY(dkhi.upload(khi.data(), size_t(P) * 4));
  const int64_t chunk = chunk_cells_for(n_cells, T * 4 + n_doy * P * 8);
  HDP_HIP_TRY(dx.alloc(size_t(chunk) * T * 4));
  HDP_HIP_TRY(dout.alloc(size_t(chunk) * n_doy * P * 8));
  SeriesUploader up;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk) {
    const int64_t nc = std::min(chunk, n_cells - c0);
    int rc = up.upload(x, c0, nc, T, stride_cell, stride_time, dx.as<float>(), g_stream);
    if (rc != HDP_OK) return rc;
    rc = launch_table_percentiles(dx.as<float>(), nc, T, dwin.as<int64_t>(), n_doy, B,
                                      dqp.as<QuantileParam>(), dklo.as<int32_t>(), dkhi.as<int32_t>(), P,
                                      dout.as<double>(), g_stream);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipMemcpyAsync(out + c0 * n_doy * P, dout.p, size_t(nc) * n_doy * P * 8,
                               hipMemcpyDeviceToHost, g_stream));
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

// ---- metrics ------------------------------------------------------------------------------------------

int hdp_metrics_plan_create(const int64_t *doy_map, int64_t T, int64_t n_doy, const int64_t *defs,
                            int64_t D, const int64_t *north, const int64_t *south, int64_t Y, int64_t P,
                            hdp_metrics_plan **plan_out) {
  HDP_REQUIRE(plan_out, HDP_EINVAL, "plan_out is NULL");
  *plan_out = nullptr;
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(doy_map && defs && (Y == 0 || (north && south)), HDP_EINVAL, "NULL table");
  HDP_REQUIRE(T > 0 && n_doy > 0 && D > 0 && P > 0 && Y >= 0, HDP_EINVAL, "bad sizes");
  HDP_REQUIRE(T < (int64_t(1) << 30), HDP_EUNSUP, "T too large");
  HDP_REQUIRE(n_doy < 65536, HDP_EUNSUP, "n_doy too large");
  const int64_t Tp = (T + 63) & ~int64_t(63);
  std::vector<uint16_t> dm(Tp, 0);
  for (int64_t t = 0; t < T; ++t) {
    int64_t v = doy_map[t];
    if (v < 0) v += n_doy;  // NumPy negative indexing into the threshold rows
    HDP_REQUIRE(v >= 0 && v < n_doy, HDP_EINVAL, "doy_map[%lld]=%lld outside the %lld threshold rows",
                (long long)t, (long long)doy_map[t], (long long)n_doy);
    dm[t] = (uint16_t)v;
  }
  bool regular = true;  // doy_map[t] == t mod n_doy: the year-aligned exceedance kernel applies
  for (int64_t t = 0; t < T && regular; ++t) regular = dm[t] == uint16_t(t % n_doy);
  std::vector<int32_t> dd(D * 3);
  for (int64_t i = 0; i < D * 3; ++i) {
    HDP_REQUIRE(defs[i] > -(int64_t(1) << 30) && defs[i] < (int64_t(1) << 30), HDP_EINVAL,
                "definition value out of range");
    dd[i] = (int32_t)defs[i];
  }
  std::vector<int2> ss(std::max<int64_t>(1, 2 * Y));
  std::vector<int64_t> r64(std::max<int64_t>(1, 4 * Y));
  bool ordered = true;
  for (int h = 0; h < 2; ++h) {
    const int64_t *r = h ? south : north;
    int64_t prev_end = 0;
    for (int64_t y = 0; y < Y; ++y) {
      const int64_t a = r[2 * y], b = r[2 * y + 1];
      // an empty or negative season slice makes the reference raise inside np.max (metric.py:136)
      HDP_REQUIRE(a >= 0 && b <= T && a < b, HDP_EINVAL,
                  "zero-size array to reduction operation maximum which has no identity "
                  "(season %lld of the %s table is [%lld, %lld))",
                  (long long)y, h ? "southern" : "northern", (long long)a, (long long)b);
      // tables that are not increasing and disjoint (the reference takes any ranges, its unit tests use overlapping
      // ones) are served by the per-series path, launch_metrics_any_ranges
      if (a < prev_end) ordered = false;
      HDP_REQUIRE(b - a < 32768, HDP_EUNSUP, "season longer than 32767 days does not fit int16");
      prev_end = std::max(prev_end, b);
      ss[h * Y + y] = make_int2((int)a, (int)b);
      r64[(h * Y + y) * 2] = a;
      r64[(h * Y + y) * 2 + 1] = b;
    }
  }
  auto *pl = new hdp_metrics_plan();
  pl->T = T; pl->n_doy = n_doy; pl->D = D; pl->Y = Y; pl->P = P;
  pl->ordered_seasons = ordered;
  pl->regular_calendar = regular;
  pl->defs_host.assign(defs, defs + D * 3);
  // HDP_METRICS_* selectors: read here, once (tests and A/B runs; every value gives the same results)
  pl->opt_general = env_option("HDP_METRICS_GENERAL", 0) != 0;
#ifdef HDP_CROSSCHECK_KERNELS
  pl->opt_fused = env_option("HDP_METRICS_FUSED", 0) != 0;
  pl->opt_cells = env_option("HDP_METRICS_CELLS", 1) != 0;
#else  // the round-1 (percentile, definition)-per-lane kernels are cross-checks: `make EXTRA=-DHDP_CROSSCHECK_KERNELS`
  pl->opt_fused = 0;
  pl->opt_cells = 1;
#endif
  pl->opt_packed = env_option("HDP_METRICS_PACKED", 1) != 0;
  pl->opt_overlap = env_option("HDP_METRICS_OVERLAP", 1) != 0;
#ifdef HDP_CROSSCHECK_KERNELS
  pl->opt_pairs = env_option("HDP_METRICS_PAIRS", 1) != 0;
#else
  pl->opt_pairs = 1;
#endif
  pl->opt_cw = (int32_t)env_option("HDP_METRICS_CW", 0);
  pl->opt_years = (int32_t)env_option("HDP_METRICS_YEARS", 1);   // 0: never, 1: records of >= 24 years, 2: any length
  pl->opt_years_lds = (int32_t)std::min<long long>(65536, std::max<long long>(0, env_option("HDP_METRICS_YEARS_LDS", 16384)));
  pl->opt_batch = std::max<long long>(0, env_option("HDP_METRICS_BATCH", 0));
  pl->opt_simple = env_option("HDP_METRICS_SIMPLE", 1) != 0;   // short path for pairs of definitions with max_break = 0
  pl->Ypitch = (Y + 15) & ~int64_t(15);  // 32-byte rows: sector-aligned packed stores
  int64_t dmax = 1;
  for (int64_t d = 0; d < D; ++d) dmax = std::max<int64_t>(dmax, dd[3 * d]);
  pl->dmax = dmax;
  bool uniform = ordered && dmax < (int64_t(1) << 20);
  for (int h = 0; h < 2 && uniform; ++h)
    for (int64_t y = 0; y + 1 < Y; ++y)
      if (ss[h * Y + y + 1].x - ss[h * Y + y].y < dmax + 64) uniform = false;
  pl->uniform_seasons = uniform;
  bool fit16 = true;
  for (int64_t d = 0; d < D; ++d)
    fit16 = fit16 && dd[3 * d] >= 0 && dd[3 * d] <= 16383 && dd[3 * d + 1] >= 0 && dd[3 * d + 1] <= 16383 &&
            dd[3 * d + 2] >= 0;
  pl->defs_fit16 = fit16;
  hipError_t e = pl->doy_map.upload(dm.data(), dm.size() * 2);
  if (e == hipSuccess) e = pl->defs.upload(dd.data(), dd.size() * 4);
  {  // the packed state machines pair definitions two by two: those with max_break = 0 first (stable), so that they pair up
    std::vector<int32_t> perm, d16(dd.size());
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t d = 0; d < D; ++d)
        if ((dd[3 * d + 1] == 0) == (pass == 0)) perm.push_back((int32_t)d);
    if (env_option("HDP_METRICS_SIMPLE", 1) == 0)
      for (int64_t d = 0; d < D; ++d) perm[d] = (int32_t)d;
    for (int64_t i = 0; i < D; ++i)
      for (int c = 0; c < 3; ++c) d16[3 * i + c] = dd[3 * perm[i] + c];
    pl->defs16_host = d16;
    if (e == hipSuccess) e = pl->defs16.upload(d16.data(), d16.size() * 4);
    if (e == hipSuccess) e = pl->def_perm.upload(perm.data(), perm.size() * 4);
  }
  if (e == hipSuccess) e = pl->seasons.upload(ss.data(), ss.size() * sizeof(int2));
  if (e == hipSuccess) e = pl->ranges64.upload(r64.data(), r64.size() * 8);
  if (e != hipSuccess) {
    delete pl;
    return set_error(HDP_EHIP, "uploading metrics plan tables failed: %s", hipGetErrorString(e));
  }
  *plan_out = pl;
  return HDP_OK;
}

int hdp_metrics_plan_destroy(hdp_metrics_plan *plan) {
  delete plan;
  return HDP_OK;
}

int64_t hdp_metrics_year_pitch(const hdp_metrics_plan *plan) { return plan ? plan->Y : 0; }

int hdp_metrics_plan_reserve(hdp_metrics_plan *plan, int64_t n_cells) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan && n_cells >= 0, HDP_EINVAL, "bad arguments");
  return reserve_metrics_scratch(plan, n_cells);
}

int64_t hdp_metrics_plan_batch_cells(const hdp_metrics_plan *plan, int64_t n_cells) {
  if (!plan || n_cells <= 0) return 0;
  return metrics_batch_cells(plan, n_cells, n_cells);
}

int hdp_metrics_f32_dev(const hdp_metrics_plan *plan, const float *x_dev, const double *thr_dev,
                        int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells,
                        int16_t *out_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan, HDP_EINVAL, "NULL plan");
  HDP_REQUIRE(n_cells >= 0 && n_thr_cells > 0, HDP_EINVAL, "bad cell counts");
  HDP_REQUIRE(n_cells == 0 || (x_dev && thr_dev && is_south_dev), HDP_EINVAL, "NULL buffer");
  if (plan->Y == 0 || n_cells == 0) return HDP_OK;
  HDP_REQUIRE(out_dev, HDP_EINVAL, "NULL output");
  return launch_metrics(plan, x_dev, thr_dev, n_thr_cells, is_south_dev, n_cells, out_dev, pick(stream));
}

// Host-pointer metrics calls.  The series are uploaded and processed in bounded chunks; `sink` receives every chunk's
// result in the device layout [4][P][D][Y][nc] (int16) and disposes of it (repack + download, widen + download,
// deposit in a shard buffer ...).  The forms:
//   hdp_metrics_f32                int16 [P][D][n_cells][4][Y]   the reference's gufunc block order
//   hdp_metrics_f32_planes_i64     int64 [4][P][D][n_cells][Y]   one plane per output variable
//   hdp_metrics_f32_layout_i16     int16 [4][P][D][Y][n_cells]   the device layout itself (what a collective moves)
//   hdp_metrics_f32_planes_i64_sharded   the rank's cells -> int16 all-gather on the device -> int64 planes of the grid
using MetricsSink = std::function<int(int64_t c0, int64_t nc, int64_t chunk, const int16_t *dout_dev)>;
static int metrics_host_chunks(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell, int64_t stride_time,
                               const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                               const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                               const int64_t *south, const uint8_t *is_south, int64_t Y, size_t extra_bytes_per_cell,
                               const MetricsSink &sink) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_cells >= 0 && n_thr_cells > 0, HDP_EINVAL, "bad cell counts");
  HDP_REQUIRE(n_cells == 0 || (x && thr && is_south), HDP_EINVAL, "NULL buffer");
  HDP_REQUIRE(n_cells % n_thr_cells == 0, HDP_EINVAL,
              "n_cells must be a multiple of the number of threshold cells");
  hdp_metrics_plan *plan = nullptr;
  int rc = hdp_metrics_plan_create(doy_map, T, n_doy, defs, D, north, south, Y, P, &plan);
  if (rc != HDP_OK) return rc;
  std::unique_ptr<hdp_metrics_plan> guard(plan);
  if (n_cells == 0 || Y == 0) return HDP_OK;
  // thresholds stay resident for the whole call; series are processed in chunks
  DevBuf dthr, dx, dsouth, dout;
  {  // the reference's (cell, doy, percentile) -> device layout [cell][P][n_doy]
    DevBuf dthr_ref;
    HDP_HIP_TRY(dthr_ref.upload(thr, size_t(n_thr_cells) * n_doy * P * 8));
    HDP_HIP_TRY(dthr.alloc(size_t(n_thr_cells) * n_doy * P * 8));
    rc = launch_swap_last2_f64(dthr_ref.as<double>(), n_thr_cells, n_doy, P, dthr.as<double>(), g_stream);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  // chunks are multiples of n_thr_cells when members share thresholds, so (c % n_thr_cells) holds
  int64_t chunk = chunk_cells_for(n_cells, T * 4 + 4 * P * D * Y * 2 + extra_bytes_per_cell);
  if (n_thr_cells < n_cells) {
    chunk = std::max<int64_t>(n_thr_cells, chunk / n_thr_cells * n_thr_cells);
  }
  DevBuf dx2;  // the next chunk uploads under this one's kernels (see AsyncUpload)
  HDP_HIP_TRY(dx.alloc(size_t(chunk) * T * 4));
  if (chunk < n_cells) HDP_HIP_TRY(dx2.alloc(size_t(chunk) * T * 4));
  HDP_HIP_TRY(dsouth.alloc(size_t(chunk)));
  HDP_HIP_TRY(dout.alloc(size_t(4) * P * D * chunk * Y * 2));
  SeriesUploader up;
  rc = up.prepare(x, n_cells, T, stride_cell, stride_time, whole_matrix_limit(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  AsyncUpload au;
  rc = au.init();
  if (rc != HDP_OK) return rc;
  float *const dxs[2] = {dx.as<float>(), dx2.as<float>()};
  au.start(&up, x, 0, std::min(chunk, n_cells), T, stride_cell, stride_time, dxs[0]);
  int64_t k = 0;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk, ++k) {
    const int64_t nc = std::min(chunk, n_cells - c0);
    rc = au.wait();
    if (rc != HDP_OK) return rc;
    if (c0 + nc < n_cells)
      au.start(&up, x, c0 + nc, std::min(chunk, n_cells - c0 - nc), T, stride_cell, stride_time, dxs[(k + 1) & 1]);
    HDP_HIP_TRY(hipMemcpyAsync(dsouth.p, is_south + c0, size_t(nc), hipMemcpyHostToDevice, g_stream));
    // with shared thresholds c0 is a multiple of n_thr_cells, so the modulo mapping is unchanged
    const double *thr_base = dthr.as<double>() + (n_thr_cells == n_cells ? c0 * n_doy * P : 0);
    const int64_t ntc = (n_thr_cells == n_cells) ? nc : n_thr_cells;
    rc = launch_metrics(plan, dxs[k & 1], thr_base, ntc, dsouth.as<uint8_t>(), nc, dout.as<int16_t>(), g_stream);
    if (rc != HDP_OK) return rc;
    rc = sink(c0, nc, chunk, dout.as<int16_t>());
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

enum MetForm { MET_REPACK16 = 0, MET_PLANES64 = 1, MET_LAYOUT16 = 2 };

static int metrics_host(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell, int64_t stride_time,
                        const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                        const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                        const int64_t *south, const uint8_t *is_south, int64_t Y, void *out, MetForm form) {
  HDP_REQUIRE(n_cells <= 0 || Y == 0 || out, HDP_EINVAL, "NULL buffer");
  const size_t esz = form == MET_PLANES64 ? 8 : 2;  // bytes per value of the result
  DevBuf dref;
  ResultPrefault pre;
  bool started = false;
  auto sink = [&](int64_t c0, int64_t nc, int64_t chunk, const int16_t *dout) -> int {
    if (!started) {
      started = true;
      if (form != MET_LAYOUT16) HDP_HIP_TRY(dref.alloc(size_t(4) * P * D * chunk * Y * esz));
      pre.start(out, size_t(P) * D * n_cells * 4 * Y * esz);  // the whole result, ahead of the first download
    }
    if (form == MET_LAYOUT16) {  // [4PD * Y] rows of nc values into rows of n_cells values
      pre.wait();
      HDP_HIP_TRY(hipMemcpy2DAsync(static_cast<char *>(out) + size_t(c0) * 2, size_t(n_cells) * 2, dout, size_t(nc) * 2,
                                   size_t(nc) * 2, size_t(4) * P * D * Y, hipMemcpyDeviceToHost, g_stream));
      return HDP_OK;
    }
    const bool planes = form == MET_PLANES64;
    int rc = planes ? launch_metrics_planes_i64(dout, P, D, nc, Y, dref.as<int64_t>(), g_stream)
                    : launch_metrics_repack(dout, P, D, nc, Y, dref.as<int16_t>(), g_stream);
    if (rc != HDP_OK) return rc;
    // the chunk's rows go straight to their places in the result: [P*D] rows of [nc][4][Y] into
    // [P*D][n_cells][4][Y], or [4*P*D] rows of [nc][Y] into [4*P*D][n_cells][Y]
    const size_t row_vals = planes ? size_t(Y) : size_t(4) * Y;
    const size_t n_rows = (planes ? size_t(4) : size_t(1)) * P * D;
    pre.wait();
    HDP_HIP_TRY(hipMemcpy2DAsync(static_cast<char *>(out) + size_t(c0) * row_vals * esz, size_t(n_cells) * row_vals * esz,
                                 dref.p, size_t(nc) * row_vals * esz, size_t(nc) * row_vals * esz, n_rows,
                                 hipMemcpyDeviceToHost, g_stream));
    return HDP_OK;
  };
  return metrics_host_chunks(x, n_cells, T, stride_cell, stride_time, thr, n_thr_cells, n_doy, P, doy_map, defs, D, north,
                             south, is_south, Y, form == MET_LAYOUT16 ? 0 : size_t(4) * P * D * Y * esz, sink);
}

int hdp_metrics_f32(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell, int64_t stride_time,
                    const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                    const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                    const int64_t *south, const uint8_t *is_south, int64_t Y, int16_t *out) {
  return metrics_host(x, n_cells, T, stride_cell, stride_time, thr, n_thr_cells, n_doy, P, doy_map, defs, D, north,
                      south, is_south, Y, out, MET_REPACK16);
}

int hdp_metrics_f32_planes_i64(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell, int64_t stride_time,
                               const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                               const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                               const int64_t *south, const uint8_t *is_south, int64_t Y, int64_t *out) {
  return metrics_host(x, n_cells, T, stride_cell, stride_time, thr, n_thr_cells, n_doy, P, doy_map, defs, D, north,
                      south, is_south, Y, out, MET_PLANES64);
}

int hdp_metrics_f32_layout_i16(const float *x, int64_t n_cells, int64_t T, int64_t stride_cell, int64_t stride_time,
                               const double *thr, int64_t n_thr_cells, int64_t n_doy, int64_t P,
                               const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                               const int64_t *south, const uint8_t *is_south, int64_t Y, int16_t *out) {
  return metrics_host(x, n_cells, T, stride_cell, stride_time, thr, n_thr_cells, n_doy, P, doy_map, defs, D, north,
                      south, is_south, Y, out, MET_LAYOUT16);
}

static ncclComm_t g_comm = nullptr;
static int g_comm_rank = -1, g_comm_world = 0;
static DevBuf g_comm_stat;  // [world + 1] int32: the status words of the sharded calls (allocated with the communicator)

// Sharded form of hdp_metrics_f32_planes_i64 (the reference's split is the dask graph of metric.py:444-452):
// this rank holds the `n_loc` cells [rank * shard, rank * shard + n_loc) of a grid of `n_total` cells, shard =
// ceil(n_total / world), as `n_mem` members x n_loc series (member-major) with the cells' own thresholds.  The int16
// result stays on the device in the layout [4][P][D][Y][n_mem * shard] (zero columns where the rank owns fewer cells), is
// all-gathered there by the library's communicator (2 bytes per value on the wire; the int64 planes would be four
// times that), and is widened and regrouped ONCE, on the gathered buffer, into the int64 planes of the whole grid,
// [4][P][D][n_mem * n_total][Y], which are downloaded into `out` in slabs.  *wire_bytes (optional) receives the bytes
// this rank handed to the collective.
int hdp_metrics_f32_planes_i64_sharded(const float *x, int64_t n_mem, int64_t n_loc, int64_t T, int64_t stride_cell,
                                       int64_t stride_time, const double *thr, int64_t n_doy, int64_t P,
                                       const int64_t *doy_map, const int64_t *defs, int64_t D, const int64_t *north,
                                       const int64_t *south, const uint8_t *is_south, int64_t Y, int64_t n_total,
                                       int64_t *out, int64_t *wire_bytes) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(g_comm, HDP_EINVAL, "no communicator (hdp_comm_init_rank)");
  if (wire_bytes) *wire_bytes = 0;
  const int64_t world = g_comm_world, rank = g_comm_rank;
  // Everything a rank can fail at BEFORE the data collective -- argument checks, the shard / gathered buffers (the
  // gathered one is world x the shard: the likeliest out-of-memory), the local pass itself -- is recorded in local_rc
  // instead of returned, and every rank then meets in a 4-byte status all-gather (its device words were allocated with the
  // communicator, so nothing can fail between here and there).  The data all-gather runs only when every rank reported
  // success: one failing rank makes every rank return an error instead of leaving the others blocked in ncclAllGather.
  int local_rc = HDP_OK;
  std::string local_msg;
  auto fail = [&](int code, const char *fmt, auto... a) {
    if (local_rc != HDP_OK) return;
    local_rc = set_error(code, fmt, a...);
    local_msg = hdp_last_error();
  };
  if (!(n_mem >= 1 && n_loc >= 0 && n_total >= 0 && P >= 1 && D >= 1 && Y >= 0)) fail(HDP_EINVAL, "bad sizes");
  const int64_t shard = world > 0 ? (std::max<int64_t>(n_total, 0) + world - 1) / world : 0;
  const int64_t lo = std::min(n_total, rank * shard), hi = std::min(n_total, (rank + 1) * shard);
  if (local_rc == HDP_OK && n_loc != hi - lo)
    fail(HDP_EINVAL, "rank %d of %d owns %lld cells of a grid of %lld, not %lld", (int)rank, (int)world,
         (long long)(hi - lo), (long long)n_total, (long long)n_loc);
  const bool empty = local_rc == HDP_OK && (n_total == 0 || Y == 0);   // the same on every rank that passed the checks
  if (local_rc == HDP_OK && !empty && !out) fail(HDP_EINVAL, "NULL output");
  const int64_t rows = 4 * P * D * Y;          // rows of the device layout
  const int64_t pad = n_mem * shard;           // series columns per rank, equal on every rank
  const size_t shard_bytes = local_rc == HDP_OK ? size_t(rows) * pad * 2 : 0;
  DevBuf dlocal, dgath;
  const char *inject = getenv("HDP_FAULT_INJECT");   // tests: "sharded_alloc" fails the gathered buffer's allocation
  if (local_rc == HDP_OK && !empty) {
    hipError_t e = dlocal.alloc(shard_bytes);
    if (e == hipSuccess) e = (inject && !strcmp(inject, "sharded_alloc")) ? hipErrorOutOfMemory : dgath.alloc(shard_bytes * world);
    if (e == hipSuccess) e = hipMemsetAsync(dlocal.p, 0, shard_bytes, g_stream);
    if (e != hipSuccess)
      fail(HDP_ENOMEM, "allocating the shard (%zu B) and gathered (%zu B) metric buffers failed: %s", shard_bytes,
           shard_bytes * size_t(world), hipGetErrorString(e));
  }
  auto sink = [&](int64_t c0, int64_t nc, int64_t /*chunk*/, const int16_t *dout) -> int {
    // series s = m * n_loc + c of the call -> column m * shard + c of the shard buffer: one 2-D copy per member touched
    for (int64_t s0 = c0; s0 < c0 + nc;) {
      const int64_t m = s0 / n_loc, c = s0 % n_loc;
      const int64_t n = std::min(n_loc - c, c0 + nc - s0);
      HDP_HIP_TRY(hipMemcpy2DAsync(dlocal.as<int16_t>() + m * shard + c, size_t(pad) * 2, dout + (s0 - c0), size_t(nc) * 2,
                                   size_t(n) * 2, size_t(rows), hipMemcpyDeviceToDevice, g_stream));
      s0 += n;
    }
    return HDP_OK;
  };
  if (local_rc == HDP_OK && !empty && n_loc > 0) {
    const int rc = metrics_host_chunks(x, n_mem * n_loc, T, stride_cell, stride_time, thr, n_loc, n_doy, P, doy_map, defs,
                                       D, north, south, is_south, Y, 0, sink);
    if (rc != HDP_OK) {
      local_rc = rc;
      local_msg = hdp_last_error();
    }
  }
  {
    // status first: [world] gathered words followed by this rank's own word (allocated by hdp_comm_init_rank)
    const int32_t mine = local_rc;
    int32_t *dstat = g_comm_stat.as<int32_t>();
    std::vector<int32_t> stat(world);
    hipError_t e = hipMemcpyAsync(dstat + world, &mine, 4, hipMemcpyHostToDevice, g_stream);
    ncclResult_t r = e == hipSuccess ? ncclAllGather(dstat + world, dstat, 4, ncclInt8, g_comm, g_stream) : ncclSuccess;
    if (e == hipSuccess && r == ncclSuccess)
      e = hipMemcpyAsync(stat.data(), dstat, size_t(4) * world, hipMemcpyDeviceToHost, g_stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(g_stream);
    if (r != ncclSuccess) return set_error(HDP_EHIP, "status ncclAllGather failed: %s", ncclGetErrorString(r));
    if (e != hipSuccess) return set_error(HDP_EHIP, "status exchange failed: %s", hipGetErrorString(e));
    if (local_rc != HDP_OK) return set_error(local_rc, "%s", local_msg.c_str());
    for (int64_t r2 = 0; r2 < world; ++r2)
      if (stat[r2] != HDP_OK)
        return set_error(HDP_EHIP, "the sharded metrics call failed on rank %d (code %d); no rank entered the data all-gather",
                         (int)r2, (int)stat[r2]);
    if (empty) return HDP_OK;
    r = ncclAllGather(dlocal.p, dgath.p, shard_bytes, ncclInt8, g_comm, g_stream);
    if (r != ncclSuccess) return set_error(HDP_EHIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
    if (wire_bytes) *wire_bytes = (int64_t)shard_bytes;
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  dlocal.release();
  // widen + regroup on the gathered buffer, slab by slab of (metric, percentile, definition) planes
  const int64_t series = n_mem * n_total;
  const int64_t mpd = 4 * P * D;
  int64_t slab = std::max<int64_t>(1, (int64_t(1) << 30) / std::max<int64_t>(1, series * Y * 8));
  slab = std::min(slab, mpd);
  DevBuf dpl;
  HDP_HIP_TRY(dpl.alloc(size_t(slab) * series * Y * 8));
  ResultPrefault pre;
  pre.start(out, size_t(mpd) * series * Y * 8);
  for (int64_t r0 = 0; r0 < mpd; r0 += slab) {
    const int64_t nr = std::min(slab, mpd - r0);
    const int rc = launch_metrics_planes_i64_gathered(dgath.as<int16_t>(), mpd, Y, world, shard, n_mem, n_total, r0, nr,
                                                      dpl.as<int64_t>(), g_stream);
    if (rc != HDP_OK) return rc;
    pre.wait();
    HDP_HIP_TRY(hipMemcpyAsync(out + size_t(r0) * series * Y, dpl.p, size_t(nr) * series * Y * 8, hipMemcpyDeviceToHost,
                               g_stream));
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

// Unit-level entry for the second half of the sharded call, without a communicator: `gathered` is what the data
// all-gather of `world` ranks leaves on every rank -- [world][4 * P * D][Y][n_mem * shard] int16 (host), shard =
// ceil(n_total / world), zero columns where a rank owns fewer cells -- and `out` receives the int64 planes of the whole
// grid, [4][P][D][n_mem * n_total][Y], exactly as hdp_metrics_f32_planes_i64_sharded regroups them.  Lets a one-GPU test
// pin the multi-rank layout (world > 1, a short last shard, several members).
int hdp_metrics_planes_i64_regroup(const int16_t *gathered, int64_t world, int64_t n_mem, int64_t n_total, int64_t P,
                                   int64_t D, int64_t Y, int64_t *out) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(world >= 1 && n_mem >= 1 && n_total >= 0 && P >= 1 && D >= 1 && Y >= 0, HDP_EINVAL, "bad sizes");
  if (n_total == 0 || Y == 0) return HDP_OK;
  HDP_REQUIRE(gathered && out, HDP_EINVAL, "NULL buffer");
  const int64_t shard = (n_total + world - 1) / world, mpd = 4 * P * D, series = n_mem * n_total;
  const size_t gbytes = size_t(world) * mpd * Y * n_mem * shard * 2, obytes = size_t(mpd) * series * Y * 8;
  DevBuf dg, dp;
  HDP_HIP_TRY(dg.alloc(gbytes));
  HDP_HIP_TRY(dp.alloc(obytes));
  HDP_HIP_TRY(hipMemcpyAsync(dg.p, gathered, gbytes, hipMemcpyHostToDevice, g_stream));
  const int rc = launch_metrics_planes_i64_gathered(dg.as<int16_t>(), mpd, Y, world, shard, n_mem, n_total, 0, mpd,
                                                    dp.as<int64_t>(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipMemcpyAsync(out, dp.p, obytes, hipMemcpyDeviceToHost, g_stream));
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  return HDP_OK;
}

const char *hdp_metrics_plan_describe(const hdp_metrics_plan *plan) {
  static thread_local char buf[256];
  if (!plan) return "";
  if (!plan->ordered_seasons)
    snprintf(buf, sizeof buf, "per-series kernels (hot days, index_heatwaves, season metrics): season tables overlap or are unordered");
  else if (!plan->uniform_seasons || plan->opt_general)
    snprintf(buf, sizeof buf, "metrics_kernel_general (seasons closed per lane)");
  else if (plan->opt_fused)
    snprintf(buf, sizeof buf, "metrics_kernel_uniform<fused> (one kernel, no exceedance scratch)");
  else if (!plan->opt_cells)
    snprintf(buf, sizeof buf, "exceed_kernel + metrics_kernel_uniform (lane = (percentile, definition)) + transpose");
  else
    snprintf(buf, sizeof buf, "%s + metrics_kernel_cells%s (lane = series; batches of series, the two kernels of "
             "consecutive batches overlap on the plan's streams)",
             hdp::metrics_year_words(plan) ? "exceed_years_kernel" : (plan->opt_pairs ? "exceed_pairs_kernel" : "exceed_kernel"),
             (plan->defs_fit16 && plan->T <= 65535 && plan->opt_packed)
                 ? "16 (packed 16-bit state)" : "");
  return buf;
}

// ---- time-major device input (CMIP order [T][cells]; reference workflow docs/example_cmip_workflow/run_cmip_workflow.py:31-32)

int hdp_thresholds_f32_tm_dev(const hdp_threshold_plan *plan, const float *x_tm_dev, int64_t pitch_cells, int64_t n_cells,
                              double *out_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan && (n_cells == 0 || (x_tm_dev && out_dev)), HDP_EINVAL, "NULL plan or buffer");
  HDP_REQUIRE(n_cells >= 0 && pitch_cells >= n_cells, HDP_EINVAL, "bad cell count or pitch");
  return launch_thresholds_tm(plan, x_tm_dev, pitch_cells, n_cells, out_dev, pick(stream));
}

int hdp_metrics_f32_tm_dev(const hdp_metrics_plan *plan, const float *x_tm_dev, int64_t pitch_cells, const double *thr_dev,
                           int64_t n_thr_cells, const uint8_t *is_south_dev, int64_t n_cells, int16_t *out_dev,
                           void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(plan, HDP_EINVAL, "NULL plan");
  HDP_REQUIRE(n_cells >= 0 && n_thr_cells > 0 && pitch_cells >= n_cells && pitch_cells > 0, HDP_EINVAL, "bad cell counts");
  HDP_REQUIRE(n_cells == 0 || (x_tm_dev && thr_dev && is_south_dev), HDP_EINVAL, "NULL buffer");
  if (plan->Y == 0 || n_cells == 0) return HDP_OK;
  HDP_REQUIRE(out_dev, HDP_EINVAL, "NULL output");
  return launch_metrics(plan, x_tm_dev, thr_dev, n_thr_cells, is_south_dev, n_cells, out_dev, pick(stream), pitch_cells);
}

// ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------------------------
// north_star: "grid cells shard embarrassingly across the 8 GPUs of one node with an RCCL all-gather over xGMI only to
// reassemble the final metrics Dataset".  The reference has no collective of its own (its data movement is implicit in
// the dask graph: xarray.map_blocks threshold.py:161, metric.py:444; concat/merge threshold.py:229, metric.py:520).

int hdp_comm_unique_id(void *id_out) {
  static_assert(sizeof(ncclUniqueId) == HDP_COMM_ID_BYTES, "ncclUniqueId size");
  HDP_REQUIRE(id_out, HDP_EINVAL, "NULL id buffer");
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return set_error(HDP_EHIP, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
  std::memcpy(id_out, &id, sizeof id);
  return HDP_OK;
}

int hdp_comm_init_rank(const void *id_bytes, int rank, int world) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(id_bytes && world >= 1 && rank >= 0 && rank < world, HDP_EINVAL, "bad communicator arguments");
  HDP_REQUIRE(g_comm == nullptr, HDP_EINVAL, "a communicator already exists (hdp_comm_destroy first)");
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof id);
  const ncclResult_t r = ncclCommInitRank(&g_comm, world, id, rank);
  if (r != ncclSuccess) {
    g_comm = nullptr;
    return set_error(HDP_EHIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
  }
  g_comm_rank = rank;
  g_comm_world = world;
  const hipError_t e = g_comm_stat.alloc(size_t(4) * (world + 1));
  if (e != hipSuccess) {
    (void)ncclCommDestroy(g_comm);
    g_comm = nullptr;
    g_comm_rank = -1;
    g_comm_world = 0;
    return set_error(HDP_ENOMEM, "allocating the communicator's status words failed: %s", hipGetErrorString(e));
  }
  return HDP_OK;
}

int hdp_comm_destroy(void) {
  if (g_comm) (void)ncclCommDestroy(g_comm);
  g_comm_stat.release();
  g_comm = nullptr;
  g_comm_rank = -1;
  g_comm_world = 0;
  return HDP_OK;
}

// RCCL's version code (ncclGetVersion), e.g. 22105: which librccl a process mapped is part of a multi-GPU diagnosis
int hdp_rccl_version(int *version) {
  HDP_REQUIRE(version, HDP_EINVAL, "NULL version");
  const ncclResult_t r = ncclGetVersion(version);
  if (r != ncclSuccess) return set_error(HDP_EHIP, "ncclGetVersion failed: %s", ncclGetErrorString(r));
  return HDP_OK;
}

int hdp_comm_rank(void) { return g_comm_rank; }
int hdp_comm_world(void) { return g_comm_world; }

// recv_dev [world][bytes_per_rank] <- every rank's send_dev [bytes_per_rank].  Bytes travel as ncclInt8 (RCCL has no
// int16): the int16 metrics shard of config 4 is 6.2 GB per rank.
int hdp_allgather_dev(const void *send_dev, size_t bytes_per_rank, void *recv_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(g_comm, HDP_EINVAL, "no communicator (hdp_comm_init_rank)");
  HDP_REQUIRE(bytes_per_rank == 0 || (send_dev && recv_dev), HDP_EINVAL, "NULL buffer");
  if (bytes_per_rank == 0) return HDP_OK;
  const ncclResult_t r = ncclAllGather(send_dev, recv_dev, bytes_per_rank, ncclInt8, g_comm, pick(stream));
  if (r != ncclSuccess) return set_error(HDP_EHIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
  return HDP_OK;
}

// The same exchange as one grouped send/recv per peer: on the full xGMI mesh (7 links x ~153 GB/s per GPU) every
// peer's shard can cross its own link at once, where a ring all-gather is bound by one link.  Which of
// the two is faster is a measurement for an 8-GPU node; bench.py reports both.
int hdp_allgather_direct_dev(const void *send_dev, size_t bytes_per_rank, void *recv_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(g_comm, HDP_EINVAL, "no communicator (hdp_comm_init_rank)");
  HDP_REQUIRE(bytes_per_rank == 0 || (send_dev && recv_dev), HDP_EINVAL, "NULL buffer");
  if (bytes_per_rank == 0) return HDP_OK;
  hipStream_t s = pick(stream);
  char *recv = static_cast<char *>(recv_dev);
  ncclResult_t r = ncclGroupStart();
  for (int peer = 0; r == ncclSuccess && peer < g_comm_world; ++peer) {
    if (peer == g_comm_rank) continue;
    r = ncclSend(send_dev, bytes_per_rank, ncclInt8, peer, g_comm, s);
    if (r == ncclSuccess) r = ncclRecv(recv + size_t(peer) * bytes_per_rank, bytes_per_rank, ncclInt8, peer, g_comm, s);
  }
  const ncclResult_t e = ncclGroupEnd();
  if (r == ncclSuccess) r = e;
  if (r != ncclSuccess) return set_error(HDP_EHIP, "grouped ncclSend/ncclRecv failed: %s", ncclGetErrorString(r));
  HDP_HIP_TRY(hipMemcpyAsync(recv + size_t(g_comm_rank) * bytes_per_rank, send_dev, bytes_per_rank,
                             hipMemcpyDeviceToDevice, s));
  return HDP_OK;
}

// ---- unit-level mirrors ---------------------------------------------------------------------------------

int hdp_index_heatwaves(const uint8_t *hot, int64_t n_series, int64_t T, int64_t min_duration,
                        int64_t max_break, int64_t max_subs, int64_t *ids) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_series >= 0 && T >= 0, HDP_EINVAL, "bad sizes");
  if (n_series * T == 0) return HDP_OK;
  HDP_REQUIRE(hot && ids, HDP_EINVAL, "NULL buffer");
  DevBuf dh, di;
  HDP_HIP_TRY(dh.upload(hot, size_t(n_series) * T));
  HDP_HIP_TRY(di.alloc(size_t(n_series) * T * 8));
  int rc = launch_index_heatwaves(dh.as<uint8_t>(), n_series, T, min_duration, max_break, max_subs,
                                  di.as<int64_t>(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  HDP_HIP_TRY(hipMemcpy(ids, di.p, size_t(n_series) * T * 8, hipMemcpyDeviceToHost));
  return HDP_OK;
}

int hdp_season_metrics(const int64_t *ids, int64_t n_series, int64_t T, const int64_t *ranges, int64_t Y,
                       int64_t *out, double *hwa) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_series >= 0 && T >= 0 && Y >= 0, HDP_EINVAL, "bad sizes");
  if (n_series * Y == 0) return HDP_OK;
  HDP_REQUIRE(ids && ranges && out && hwa, HDP_EINVAL, "NULL buffer");
  for (int64_t y = 0; y < Y; ++y) {
    int64_t a = ranges[2 * y], b = ranges[2 * y + 1];
    HDP_REQUIRE(a >= 0 && b <= T && a < b, HDP_EINVAL,
                "zero-size array to reduction operation maximum which has no identity");
  }
  DevBuf di, dr, dout, dh;
  HDP_HIP_TRY(di.upload(ids, size_t(n_series) * T * 8));
  HDP_HIP_TRY(dr.upload(ranges, size_t(Y) * 16));
  HDP_HIP_TRY(dout.alloc(size_t(n_series) * 4 * Y * 8));
  HDP_HIP_TRY(dh.alloc(size_t(n_series) * Y * 8));
  int rc = launch_season_metrics(di.as<int64_t>(), n_series, T, dr.as<int64_t>(), Y, dout.as<int64_t>(),
                                 dh.as<double>(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  HDP_HIP_TRY(hipMemcpy(out, dout.p, size_t(n_series) * 4 * Y * 8, hipMemcpyDeviceToHost));
  HDP_HIP_TRY(hipMemcpy(hwa, dh.p, size_t(n_series) * Y * 8, hipMemcpyDeviceToHost));
  return HDP_OK;
}

int hdp_indicate_hot_days(const float *measure, int64_t n_series, int64_t T, const double *thr,
                          int64_t n_doy, const int64_t *doy_map, uint8_t *hot) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_series >= 0 && T >= 0 && n_doy > 0, HDP_EINVAL, "bad sizes");
  if (n_series * T == 0) return HDP_OK;
  HDP_REQUIRE(measure && thr && doy_map && hot, HDP_EINVAL, "NULL buffer");
  std::vector<int64_t> dm(T);
  for (int64_t t = 0; t < T; ++t) {
    int64_t v = doy_map[t];
    if (v < 0) v += n_doy;
    HDP_REQUIRE(v >= 0 && v < n_doy, HDP_EINVAL, "doy_map[%lld] out of range", (long long)t);
    dm[t] = v;
  }
  DevBuf dx, dt, dd, dh;
  HDP_HIP_TRY(dx.upload(measure, size_t(n_series) * T * 4));
  HDP_HIP_TRY(dt.upload(thr, size_t(n_series) * n_doy * 8));
  HDP_HIP_TRY(dd.upload(dm.data(), size_t(T) * 8));
  HDP_HIP_TRY(dh.alloc(size_t(n_series) * T));
  int rc = launch_indicate_hot_days(dx.as<float>(), n_series, T, dt.as<double>(), n_doy, dd.as<int64_t>(),
                                    dh.as<uint8_t>(), g_stream);
  if (rc != HDP_OK) return rc;
  HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  HDP_HIP_TRY(hipMemcpy(hot, dh.p, size_t(n_series) * T, hipMemcpyDeviceToHost));
  return HDP_OK;
}

int hdp_heat_index_f32_dev(const float *temp_f_dev, const float *rel_humid_dev, int64_t n, float *out_dev,
                           void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n >= 0 && (n == 0 || (temp_f_dev && rel_humid_dev && out_dev)), HDP_EINVAL, "bad arguments");
  return launch_heat_index(temp_f_dev, rel_humid_dev, n, out_dev, false, pick(stream));
}

int hdp_heat_index_celsius_f32_dev(const float *temp_c_dev, const float *rel_humid_dev, int64_t n,
                                   float *out_c_dev, void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n >= 0 && (n == 0 || (temp_c_dev && rel_humid_dev && out_c_dev)), HDP_EINVAL, "bad arguments");
  return launch_heat_index(temp_c_dev, rel_humid_dev, n, out_c_dev, true, pick(stream));
}

int hdp_heat_index_f32(const float *temp_f, const float *rel_humid, int64_t n, float *out) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n >= 0 && (n == 0 || (temp_f && rel_humid && out)), HDP_EINVAL, "bad arguments");
  const int64_t chunk = int64_t(1) << 27;  // 512 MiB per operand
  DevBuf dt, dr, dout;
  const int64_t cap = std::min(n, chunk);
  HDP_HIP_TRY(dt.alloc(size_t(cap) * 4));
  HDP_HIP_TRY(dr.alloc(size_t(cap) * 4));
  HDP_HIP_TRY(dout.alloc(size_t(cap) * 4));
  for (int64_t i0 = 0; i0 < n; i0 += chunk) {
    const int64_t m = std::min(chunk, n - i0);
    HDP_HIP_TRY(hipMemcpyAsync(dt.p, temp_f + i0, size_t(m) * 4, hipMemcpyHostToDevice, g_stream));
    HDP_HIP_TRY(hipMemcpyAsync(dr.p, rel_humid + i0, size_t(m) * 4, hipMemcpyHostToDevice, g_stream));
    int rc = launch_heat_index(dt.as<float>(), dr.as<float>(), m, dout.as<float>(), false, g_stream);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipMemcpyAsync(out + i0, dout.p, size_t(m) * 4, hipMemcpyDeviceToHost, g_stream));
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

int hdp_weighted_mean_i16_dev(const int16_t *v_dev, int64_t n_rows, int64_t n, const double *w_dev, double *out_dev,
                              void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_rows >= 0 && n > 0, HDP_EINVAL, "bad sizes");
  HDP_REQUIRE(n_rows == 0 || (v_dev && w_dev && out_dev), HDP_EINVAL, "NULL buffer");
  return launch_weighted_row_mean_i16(v_dev, n_rows, n, w_dev, out_dev, pick(stream));
}

int hdp_weighted_mean_f64(const double *v, int64_t n_rows, int64_t n, const double *w, double *out) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_rows >= 0 && n > 0, HDP_EINVAL, "bad sizes");
  HDP_REQUIRE(n_rows == 0 || (v && w && out), HDP_EINVAL, "NULL buffer");
  if (n_rows == 0) return HDP_OK;
  const int64_t chunk = std::max<int64_t>(1, (int64_t(1) << 27) / n);  // ~1 GiB of rows per pass
  DevBuf dv, dw, dout;
  HDP_HIP_TRY(dw.upload(w, size_t(n) * 8));
  HDP_HIP_TRY(dv.alloc(size_t(std::min(chunk, n_rows)) * n * 8));
  HDP_HIP_TRY(dout.alloc(size_t(std::min(chunk, n_rows)) * 8));
  for (int64_t r0 = 0; r0 < n_rows; r0 += chunk) {
    const int64_t m = std::min(chunk, n_rows - r0);
    HDP_HIP_TRY(hipMemcpyAsync(dv.p, v + r0 * n, size_t(m) * n * 8, hipMemcpyHostToDevice, g_stream));
    int rc = launch_weighted_row_mean_f64(dv.as<double>(), m, n, dw.as<double>(), dout.as<double>(), g_stream);
    if (rc != HDP_OK) return rc;
    HDP_HIP_TRY(hipMemcpyAsync(out + r0, dout.p, size_t(m) * 8, hipMemcpyDeviceToHost, g_stream));
    HDP_HIP_TRY(hipStreamSynchronize(g_stream));
  }
  return HDP_OK;
}

int hdp_generate_series_dev(float *x_dev, int64_t n_cells, int64_t T, int64_t cell_offset,
                            const float *lat_dev, uint64_t seed, float noise_scale, float trend_per_day,
                            void *stream) {
  HDP_REQUIRE(device_ready(), HDP_ENODEV, "hdp_init() has not selected a HIP device");
  HDP_REQUIRE(n_cells >= 0 && T >= 0, HDP_EINVAL, "bad sizes");
  if (n_cells * T == 0) return HDP_OK;
  HDP_REQUIRE(x_dev && lat_dev, HDP_EINVAL, "NULL buffer");
  return launch_generate(x_dev, n_cells, T, cell_offset, lat_dev, seed, noise_scale, trend_per_day,
                         pick(stream));
}

}  // extern "C"
