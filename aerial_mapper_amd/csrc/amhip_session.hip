// amhip_session.hip -- one map served through HOST matrices by one or several GPUs.
//
// The reference's host is ONE C++ process whose state is the GridMap's float matrices
// (main-dsm.cc:103-107, main-ortho-backward-grid.cc:128-141): Dsm::process and
// OrthoBackwardGrid::process read and write them.  A session is what the drop-in classes
// (aerial_mapper_amd/cpp/*.cc) share per map:
//   * windows    the map cut into tiles_i x tiles_j windows, window k on devices[k] (one
//                context each; a device may serve several windows).  A DSM call uploads a
//                slice of the cloud to every device, each selects what every window needs
//                (its cells + the halo margin, k_halo_select) and the selections travel
//                device to device (hipMemcpyPeerAsync: xGMI) -- SURVEY 8e option (i), halo
//                POINTS, inside one process; every window then runs the single-GPU kernels.
//                Frames and poses are replicated; the mosaic needs no exchange.
//   * residency  the layers stay on the devices between calls.  Whether a host matrix still
//                holds what the device holds is decided by CONTENT: two 64-bit sums of a
//                NON-LINEAR mix of (cell bits, cell position) over the matrix -- any single
//                changed cell changes both, and no arithmetic relation between two edits cancels
//                (round 2's sum was linear in the bits: d1 (2 g1 + 1) = -d2 (2 g2 + 1) went
//                unnoticed) --, computed with host threads on the way in and by a kernel on
//                the way out.  Equal -> no transfer;
//                a matrix that holds its initial constant -> a lazy device-side reset instead
//                of an upload; anything else is uploaded.  Outputs the kernels did not change
//                (num_observations: `+= itself`, ortho-backward-grid.cc:183) are not downloaded.
//                tuning knob session_always_copy: every matrix up and down, like round 1.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "amhip_common.h"
#include "amhip_content_sum.h"

namespace amhip {

struct Win {
  int i0, j0, rows, cols;
};

struct Hash128 {
  unsigned long long a = 0, b = 0;
  bool operator==(const Hash128& o) const { return a == o.a && b == o.b; }
  bool operator!=(const Hash128& o) const { return !(*this == o); }
};

// What the session knows about one layer of one window.  `device_valid`: the device layer holds
// a matrix with content sum `device`.  It is cleared BEFORE any kernel that writes the layer is
// enqueued and set again only by sync_out, from the device's own sum: a call that fails midway
// leaves it cleared, so the retry uploads the host matrix instead of trusting a half-written
// layer.  `host`: content sum of the host matrix as this session last saw or wrote it.
struct LayerSync {
  bool device_valid = false;
  Hash128 device;
  bool host_known = false;
  Hash128 host;
  // sync_in found (or put) the device layer in its lazily-reset initial state AND the host matrix
  // holds that constant: once a kernel has materialised the layer, sync_out downloads it without
  // waiting for its content sum (the sum then runs on a second stream beside the download)
  bool fresh_in = false;
};

// Where the last host-buffer call spent its time (amhip_session_last_profile; window 0's view):
// wall-clock phases on the host, HIP-event times for what the stream did.
struct CallProfile {
  double total_ms = 0, h2d_ms = 0, host_sum_ms = 0, kernel_ms = 0, dev_sum_wait_ms = 0, d2h_ms = 0;
  double up_bytes = 0, down_bytes = 0;
};

struct Session {
  amhip_grid_desc grid;
  int ti = 1, tj = 1;
  std::vector<int> edges_i, edges_j;
  std::vector<amhip_ctx*> ctx;
  std::vector<Win> win;
  std::vector<int> dev;
  std::vector<LayerSync> sync;                 // [window][layer]
  std::vector<Hash128> const_hash;             // [window][layer]: content sum of the initial constant
  bool always_copy = false;
  // DSM routing (W > 1): count, then compact -- a slice's selections for all windows take
  // (points selected) rows, not (slice x windows)
  std::vector<double*> route_out;
  std::vector<size_t> route_cap;               // doubles
  std::vector<long long*> route_counts;        // device, W per window
  std::vector<hipEvent_t> route_done;          // window k's selections are complete
  std::vector<double*> cloud;
  std::vector<size_t> cloud_cap;               // doubles
  std::vector<unsigned long long*> dev_hash;   // device scratch: two u64 per layer
  std::vector<unsigned long long*> pin_hash;   // its pinned host mirror
  std::vector<float*> pin;                     // pinned host staging of partial downloads
  std::vector<size_t> pin_cap;                 // floats
  bool verify_partial = false;                 // tuning knob session_verify_partial: re-sum the host matrix
  std::atomic<unsigned long long> up_bytes{0}, down_bytes{0};  // layer traffic (amhip_session_transfer_stats)
  // per window: a second stream for the content sums that run beside a download, and the events
  // that time a call's kernels / downloads (CallProfile)
  std::vector<hipStream_t> aux_stream;
  std::vector<hipEvent_t> ev_k0, ev_k1, ev_s1, ev_d1;
  CallProfile prof;
  std::atomic<unsigned long long> prof_down{0};   // (the windows' threads add to it: the call's downloads)
  int W() const { return (int)ctx.size(); }
};

// tuning knob session_trace: wall time of every phase of a call on stderr (where does a host-matrix
// call spend its time: content sums, uploads, kernels, downloads?)
struct PhaseClock {
  bool on;
  const char* call;
  std::chrono::steady_clock::time_point t0;
  explicit PhaseClock(const char* name)
      : on(tuning_on("session_trace")), call(name), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* phase, hipStream_t wait_for = nullptr, bool wait = false) {
    if (!on) return;
    if (wait) (void)hipStreamSynchronize(wait_for);
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[amhip session] %s: %-28s %8.3f ms\n", call, phase,
                 std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// ---- content sums (amhip_content_sum.h: cell_mix, host_column_sum) ----------------------
// layer == nullptr: the sums of a window filled with `constant` (no memory is read)
__global__ void __launch_bounds__(256)
k_layer_hash(const float* __restrict__ layer, float constant, int rows, int cols, int i0, int j0,
             int map_rows, unsigned long long* __restrict__ out) {
  unsigned long long a = 0, b = 0;
  const size_t n = (size_t)rows * (size_t)cols;
  const size_t stride = (size_t)gridDim.x * 256;
  const unsigned cbits = __float_as_uint(constant);
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
    const int i = (int)(k % (size_t)rows), j = (int)(k / (size_t)rows);
    const unsigned long long g = (unsigned long long)(i0 + i) +
                                 (unsigned long long)(j0 + j) * (unsigned long long)map_rows;
    cell_mix(layer ? __float_as_uint(layer[k]) : cbits, g, &a, &b);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    a += __shfl_xor(a, d, 64);
    b += __shfl_xor(b, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, a);
    atomicAdd(out + 1, b);
  }
}

// CPUs this process may keep busy: hardware threads, capped by its affinity mask and by a cgroup
// CPU quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1).  *quota_cpus: the quota
// alone (0: none).  A GPU box's pod runs under such a quota: 256 hardware threads, 16 CPUs' worth of
// time per 100 ms (tools/ubench/host_sum_bench.cc: 128 summing threads run 12 ms at 200 GB/s,
// then the whole process is frozen until the period ends).
struct CpuBudget {
  int cpus;
  double quota;
};
static CpuBudget measure_cpu_budget() {
  int n = (int)std::thread::hardware_concurrency();
  if (n <= 0) n = 8;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
  double q = 0.0;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64] = {0};
    double period = 0.0;
    if (std::fscanf(f, "%63s %lf", a, &period) == 2 && period > 0.0 && a[0] >= '0' && a[0] <= '9')
      q = std::atof(a) / period;
    std::fclose(f);
  } else {
    double qu = -1.0, period = 0.0;
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      if (std::fscanf(g, "%lf", &qu) != 1) qu = -1.0;
      std::fclose(g);
    }
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(g, "%lf", &period) != 1) period = 0.0;
      std::fclose(g);
    }
    if (qu > 0.0 && period > 0.0) q = qu / period;
  }
  if (q > 0.0) n = std::min(n, std::max(1, (int)std::ceil(q)));
  return {n, q};
}
static int usable_cpus(double* quota_cpus) {
  static const CpuBudget b = measure_cpu_budget();  // (once per process; thread-safe initialisation)
  if (quota_cpus) *quota_cpus = b.quota;
  return b.cpus;
}

// threads for a pass of host threads over `bytes` of matrices
static int host_threads(size_t bytes) {
  if (tuning("session_threads", 0.0) > 0.0) return std::max(1, (int)tuning("session_threads", 0.0));
  unsigned hw = std::thread::hardware_concurrency();
  double quota = 0.0;
  const int cpus = usable_cpus(&quota);
  // (the AVX-512 loop sums 33 GB/s per thread on the pool's EPYCs: a few threads per memory
  // channel group saturate the host's memory, more only burn CPU time in stalls)
  if (host_sum_is_vectorized()) return std::max(1, std::min(cpus, 32));
  // (the scalar mix costs three multiplies per cell, 5 GB/s per thread: 128 threads keep the six
  // matrices of a mosaic call -- 2.4 GB at cfg3 -- near the host's memory bandwidth)
  int T = (int)std::min<unsigned>(hw ? hw : 8u, 128u);
  if (quota > 0.0 && quota < (double)T) {
    // under a CPU quota: a short pass may burst over four times the quota's CPUs (64 threads draw
    // 175 GB/s; more only burn the period's allowance in memory stalls); a pass worth more than
    // ~ 60 % of one period's allowance (100 ms x quota; a thread sums ~ 4.5 GB/s) would get the
    // process frozen mid-way: one thread per CPU of the quota, steadily.  Measured on the pool's
    // boxes (quota 16): cfg3's six matrices 75 - 82 ms per first pass whatever the burst width
    // (32 .. 128), 100 ms at 16 threads; the five 1.6 GB matrices of a 20 000^2 map 180 ms at
    // 128 threads, 150 ms at 16.
    const double cpu_seconds = (double)bytes / 4.5e9;
    T = cpu_seconds <= 0.06 * quota ? std::min(T, 4 * cpus) : cpus;
  } else {
    T = std::min(T, std::max(cpus, 1));
  }
  return std::max(1, T);
}

// per-window sums of `nl` host matrices (column-major, the map's size) with host threads
static void host_hashes(const Session& s, const float* const* mats, int nl,
                        std::vector<Hash128>* out /* [nl][W] */) {
  const int W = s.W(), R = s.grid.rows, Cc = s.grid.cols;
  out->assign((size_t)nl * W, Hash128());
  size_t bytes = 0;
  for (int l = 0; l < nl; ++l)
    if (mats[l]) bytes += (size_t)R * (size_t)Cc * 4;
  const int T = std::max(1, std::min(host_threads(bytes), Cc));
  std::vector<std::vector<Hash128>> part(T, std::vector<Hash128>((size_t)nl * W));
  auto work = [&](int t) {
    const int c0 = (int)((long long)Cc * t / T), c1 = (int)((long long)Cc * (t + 1) / T);
    std::vector<Hash128>& acc = part[t];
    int b = 0;
    for (int j = c0; j < c1; ++j) {
      while (j >= s.edges_j[b + 1]) ++b;
      for (int a = 0; a < s.ti; ++a) {
        const int k = a + b * s.ti;
        const int ia = s.edges_i[a], ib = s.edges_i[a + 1];
        const unsigned long long g0 = (unsigned long long)j * (unsigned long long)R;
        for (int l = 0; l < nl; ++l) {
          if (!mats[l]) continue;
          const unsigned* col = reinterpret_cast<const unsigned*>(mats[l]) + (size_t)j * R;
          host_column_sum(col + ia, (size_t)(ib - ia), g0 + (unsigned long long)ia,
                          &acc[(size_t)l * W + k].a, &acc[(size_t)l * W + k].b);
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  for (int t = 0; t < T; ++t)
    for (size_t q = 0; q < out->size(); ++q) {
      (*out)[q].a += part[t][q].a;
      (*out)[q].b += part[t][q].b;
    }
}

static inline float* map_at(float* base, const Session& s, const Win& w) {
  return base + (size_t)w.i0 + (size_t)w.j0 * (size_t)s.grid.rows;
}
static inline const float* map_at(const float* base, const Session& s, const Win& w) {
  return base + (size_t)w.i0 + (size_t)w.j0 * (size_t)s.grid.rows;
}

// window block <-> map-shaped host matrix (bytes: the grid_map message's payloads are not
// aligned).  A window that spans all rows of the map is ONE contiguous range of both: a plain
// copy instead of a pitched one (the runtime stages pitched copies of pageable memory row by row).
static hipError_t copy_window(void* dst_host_or_dev, const void* src, const Session& s, const Win& w,
                              bool to_host, hipStream_t stream) {
  const size_t col = (size_t)w.rows * 4;
  if (w.rows == s.grid.rows)
    return hipMemcpyAsync(dst_host_or_dev, src, col * (size_t)w.cols,
                          to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, stream);
  return to_host ? hipMemcpy2DAsync(dst_host_or_dev, (size_t)s.grid.rows * 4, src, col, col,
                                    (size_t)w.cols, hipMemcpyDeviceToHost, stream)
                 : hipMemcpy2DAsync(dst_host_or_dev, col, src, (size_t)s.grid.rows * 4, col,
                                    (size_t)w.cols, hipMemcpyHostToDevice, stream);
}

// host matrix -> window layer, as far as needed (asynchronous on the context's stream).
// will_write: the call that follows writes this layer on the device.
static int sync_in(Session& s, int k, int layer, const float* host, const Hash128& host_hash,
                   bool will_write) {
  Ctx* c = &s.ctx[k]->impl;
  LayerSync& st = s.sync[(size_t)k * AMHIP_NUM_LAYERS + layer];
  int rc = ctx_use_device(c);
  if (rc) return rc;
  st.host = host_hash;
  st.host_known = !s.always_copy;
  st.fresh_in = false;
  bool have = false;
  if (!s.always_copy) {
    const bool host_is_const = host_hash == s.const_hash[(size_t)k * AMHIP_NUM_LAYERS + layer];
    if (st.device_valid && st.device == host_hash) {
      have = true;  // the device holds exactly this
      st.fresh_in = host_is_const && ctx_layer_is_initial(c, layer);
    } else if (host_is_const) {
      // (any matrix with these sums is taken for the initial constant: 2^-128)
      if ((rc = ctx_layer_set_initial(c, layer))) return rc;
      have = true;
      st.fresh_in = ctx_layer_is_initial(c, layer);
    }
  }
  if (!have) {
    const Win& w = s.win[k];
    st.device_valid = false;
    ctx_overwrite(c, layer);
    AMHIP_TRY(copy_window(c->layers[layer], map_at(host, s, w), s, w, false, c->stream));
    s.up_bytes += (unsigned long long)w.rows * (unsigned long long)w.cols * 4ull;
  }
  st.device = host_hash;
  // (a layer the next kernel writes is unknown until sync_out has summed it again)
  st.device_valid = !s.always_copy && !will_write;
  return AMHIP_OK;
}

// columns [0, cols) of a packed rows x cols block -> the map-shaped host matrix at (i, j)
static void unpack_block(const float* packed, float* host, const Session& s, int i, int j, int rows,
                         int cols) {
  const size_t bytes = (size_t)rows * (size_t)cols * 4;
  const int T = bytes < (8u << 20) ? 1 : std::min(16, cols);
  auto work = [&](int t) {
    const int c0 = (int)((long long)cols * t / T), c1 = (int)((long long)cols * (t + 1) / T);
    for (int q = c0; q < c1; ++q)
      std::memcpy(host + (size_t)i + (size_t)(j + q) * (size_t)s.grid.rows, packed + (size_t)q * rows,
                  (size_t)rows * 4);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

// window layers -> host matrices where the device content differs from what the host holds.
// dirty_ok: the call before was a DSM / backward-mosaic call, whose context knows which cells of
// the window it can have written (ctx_last_dirty): a small cloud's sub-window, the bounding box of
// a small batch's tile list.  A host matrix that equalled the device layer BEFORE the call (what
// sync_in establishes) differs from it only there: that rectangle is downloaded (device -> pinned
// staging -> the matrix's columns) instead of the window -- the incremental use case on a large
// map downloads megabytes instead of gigabytes per call.  tuning knob session_no_partial: always the
// whole window (A-B, tests).
static int sync_out(Session& s, int k, const int* layers, float* const* hosts, int nl,
                    bool dirty_ok = false) {
  PhaseClock clock("sync_out");
  Ctx* c = &s.ctx[k]->impl;
  int rc = ctx_use_device(c);
  if (rc) return rc;
  const Win& w = s.win[k];
  Hash128 h[AMHIP_NUM_LAYERS];
  bool run[AMHIP_NUM_LAYERS] = {};
  int rect[4] = {0, 0, w.rows, w.cols};
  bool partial = false;
  if (dirty_ok && !s.always_copy && !tuning_on("session_no_partial")) {
    if ((rc = ctx_last_dirty(c, rect))) return rc;
    partial = rect[2] > 0 && rect[3] > 0 && rect[0] >= 0 && rect[1] >= 0 &&
              rect[0] + rect[2] <= w.rows && rect[1] + rect[3] <= w.cols &&
              2 * (size_t)rect[2] * (size_t)rect[3] <= (size_t)w.rows * (size_t)w.cols;
  }
  AMHIP_TRY(hipEventRecord(s.ev_k1[k], c->stream));  // (the call's kernels end here)
  clock.mark("kernels (wait)", c->stream, true);
  // A download is asynchronous: until the stream has synchronised without an error the host
  // matrix may be stale or half written, so the layers being downloaded are marked "host
  // unknown" FIRST and "host == device sum" only after the wait (a failed copy must
  // not leave the session believing the host holds h[q] -- the next call would skip it).
  bool copied[AMHIP_NUM_LAYERS] = {};
  bool packed[AMHIP_NUM_LAYERS] = {};
  bool early[AMHIP_NUM_LAYERS] = {};
  bool any = false;
  const auto wall0 = std::chrono::steady_clock::now();
  // (1) "fresh" layers -- the host holds the initial constant, the device layer was lazily initial
  // when the call began and a kernel has materialised it since: their download starts NOW, on the
  // context's stream, and their content sums (which only decide what LATER calls may skip) run
  // beside it on the second stream.  A layer whose every written cell happens to equal the
  // constant is downloaded for nothing; nothing else changes.
  if (!s.always_copy && !partial && !tuning_on("session_serial_sums")) {
    for (int q = 0; q < nl; ++q) {
      if (!hosts[q]) continue;
      const LayerSync& st = s.sync[(size_t)k * AMHIP_NUM_LAYERS + layers[q]];
      early[q] = st.fresh_in && st.host_known && !ctx_layer_is_initial(c, layers[q]);
      any = any || early[q];
    }
  }
  const bool beside = any;
  unsigned long long* const got = s.pin_hash[k];   // (pinned: the copy below must not block the host)
  if (!s.always_copy) {
    hipStream_t hs = c->stream;
    if (beside) {  // (beside the downloads; ordered behind the kernels by their end event)
      hs = s.aux_stream[k];
      AMHIP_TRY(hipStreamWaitEvent(hs, s.ev_k1[k], 0));
    }
    AMHIP_TRY(hipMemsetAsync(s.dev_hash[k], 0, sizeof(unsigned long long) * 2 * AMHIP_NUM_LAYERS, hs));
    for (int q = 0; q < nl; ++q) {
      if (!hosts[q]) continue;
      const int l = layers[q];
      if (ctx_layer_is_initial(c, l)) {  // nothing wrote it since its (lazy) reset
        h[q] = s.const_hash[(size_t)k * AMHIP_NUM_LAYERS + l];
        continue;
      }
      run[q] = true;
      hipLaunchKernelGGL(k_layer_hash, dim3(2048), dim3(256), 0, hs, c->layers[l], 0.0f, w.rows,
                         w.cols, w.i0, w.j0, s.grid.rows, s.dev_hash[k] + 2 * q);
    }
    AMHIP_TRY(hipMemcpyAsync(got, s.dev_hash[k], sizeof(unsigned long long) * 2 * AMHIP_NUM_LAYERS,
                             hipMemcpyDeviceToHost, hs));
    // (a download into pageable memory keeps the calling thread until it is done: the sums are
    // already running on the second stream by then)
    for (int q = 0; q < nl; ++q) {
      if (!early[q]) continue;
      LayerSync& st = s.sync[(size_t)k * AMHIP_NUM_LAYERS + layers[q]];
      st.host_known = false;
      AMHIP_TRY(copy_window(map_at(hosts[q], s, w), c->layers[layers[q]], s, w, true, c->stream));
      s.down_bytes += (unsigned long long)w.rows * (unsigned long long)w.cols * 4ull;
      s.prof_down += (unsigned long long)w.rows * (unsigned long long)w.cols * 4ull;
      copied[q] = true;
    }
    // (the downloads' own start, where the sums ran in front of them on the same stream)
    AMHIP_TRY(hipEventRecord(s.ev_s1[k], c->stream));
    AMHIP_TRY(hipStreamSynchronize(hs));
    for (int q = 0; q < nl; ++q)
      if (run[q]) {
        h[q].a = got[2 * q];
        h[q].b = got[2 * q + 1];
      }
  }
  clock.mark("device content sums");
  const auto wall1 = std::chrono::steady_clock::now();
  const size_t block = (size_t)rect[2] * (size_t)rect[3];
  if (partial) {  // (staging for every layer of the call, at most 1 GiB: beyond, the plain way)
    const size_t need = block * (size_t)nl;
    if (need * 4 > (1ull << 30)) {
      partial = false;
    } else if (need > s.pin_cap[k]) {
      if (s.pin[k]) (void)hipHostFree(s.pin[k]);
      s.pin[k] = nullptr;
      s.pin_cap[k] = 0;
      if (hipHostMalloc(reinterpret_cast<void**>(&s.pin[k]), need * 4, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        s.pin[k] = nullptr;
        partial = false;
      } else {
        s.pin_cap[k] = need;
      }
    }
  }
  // (2) every other layer: downloaded where the device's sum differs from what the host holds
  for (int q = 0; q < nl; ++q) {
    if (!hosts[q]) continue;
    const int l = layers[q];
    LayerSync& st = s.sync[(size_t)k * AMHIP_NUM_LAYERS + l];
    const bool host_has_it = !s.always_copy && st.host_known && st.host == h[q];
    // (the host matrix is what the device layer was before the call)
    const bool host_is_before = !s.always_copy && st.host_known && st.host == st.device;
    st.device_valid = !s.always_copy;
    st.device = h[q];
    if (early[q]) continue;
    if (!host_has_it) {
      st.host_known = false;
      if ((rc = ctx_materialize(c, l))) return rc;
      if (partial && host_is_before) {
        AMHIP_TRY(hipMemcpy2DAsync(s.pin[k] + block * (size_t)q, (size_t)rect[2] * 4,
                                   c->layers[l] + (size_t)rect[0] + (size_t)rect[1] * (size_t)w.rows,
                                   (size_t)w.rows * 4, (size_t)rect[2] * 4, (size_t)rect[3],
                                   hipMemcpyDeviceToHost, c->stream));
        packed[q] = true;
        s.down_bytes += (unsigned long long)block * 4ull;
        s.prof_down += (unsigned long long)block * 4ull;
      } else {
        AMHIP_TRY(copy_window(map_at(hosts[q], s, w), c->layers[l], s, w, true, c->stream));
        s.down_bytes += (unsigned long long)w.rows * (unsigned long long)w.cols * 4ull;
        s.prof_down += (unsigned long long)w.rows * (unsigned long long)w.cols * 4ull;
      }
      copied[q] = true;
      any = true;
    }
  }
  AMHIP_TRY(hipEventRecord(s.ev_d1[k], c->stream));
  if (any) AMHIP_TRY(hipStreamSynchronize(c->stream));
  clock.mark("downloads");
  for (int q = 0; q < nl; ++q)
    if (packed[q])
      unpack_block(s.pin[k] + block * (size_t)q, hosts[q], s, w.i0 + rect[0], w.j0 + rect[1], rect[2],
                   rect[3]);
  clock.mark("unpack");
  if (s.verify_partial) {  // (tests: the matrix as a whole must now carry the device's sums)
    for (int q = 0; q < nl; ++q) {
      if (!packed[q]) continue;
      std::vector<Hash128> hv;
      const float* m[1] = {hosts[q]};
      host_hashes(s, m, 1, &hv);
      if (hv[(size_t)k] != h[q])
        return arg_failure("session: a partial download left the host matrix different from the device layer");
    }
  }
  for (int q = 0; q < nl; ++q) {
    if (!hosts[q] || !copied[q]) continue;
    LayerSync& st = s.sync[(size_t)k * AMHIP_NUM_LAYERS + layers[q]];
    st.host_known = !s.always_copy;
    st.host = h[q];
  }
  if (k == 0) {  // (the call's profile: window 0's view)
    AMHIP_TRY(hipEventSynchronize(s.ev_d1[k]));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.ev_k0[k], s.ev_k1[k]) == hipSuccess) s.prof.kernel_ms = ms;
    if (hipEventElapsedTime(&ms, beside || s.always_copy ? s.ev_k1[k] : s.ev_s1[k], s.ev_d1[k]) == hipSuccess)
      s.prof.d2h_ms = ms;
    (void)hipGetLastError();
    // (the sums' share of the wall clock: the whole wait when nothing was downloaded beside them)
    s.prof.dev_sum_wait_ms = beside ? 0.0 : std::chrono::duration<double, std::milli>(wall1 - wall0).count();
  }
  return AMHIP_OK;
}

// dsm.cc:127-144: the largest squared radius of the ladder -> metres a window grows by
static double halo_margin_m(int radius_sq, double res) {
  double tmax = (double)radius_sq, lambda = 1.0;
  for (;;) {
    tmax = std::max(tmax, lambda * radius_sq);
    lambda *= 1.1;
    if (lambda * radius_sq > 7.0) break;
  }
  return std::sqrt(tmax) + res;
}

template <typename F>
static int for_windows(Session& s, F fn) {
  const int W = s.W();
  std::vector<int> rcs(W, AMHIP_OK);
  std::vector<std::string> msgs(W);
  if (W == 1) return fn(0);
  std::vector<std::thread> th;
  for (int k = 0; k < W; ++k)
    th.emplace_back([&, k]() {
      rcs[k] = fn(k);
      if (rcs[k]) msgs[k] = amhip_last_error();  // (thread-local message)
    });
  for (auto& x : th) x.join();
  for (int k = 0; k < W; ++k)
    if (rcs[k]) {
      set_last_error(msgs[k]);
      return rcs[k];
    }
  return AMHIP_OK;
}

}  // namespace amhip

using namespace amhip;

struct amhip_session {
  amhip::Session impl;
};

extern "C" {

int amhip_session_create(const amhip_grid_desc* grid, int tiles_i, int tiles_j,
                         const int32_t* devices, amhip_session** out) {
  if (!grid || !out || tiles_i < 1 || tiles_j < 1)
    return arg_failure("amhip_session_create: bad argument");
  if (tiles_i > grid->rows || tiles_j > grid->cols || (long long)tiles_i * tiles_j > 64)
    return arg_failure("amhip_session_create: more windows than the map can be cut into (<= 64)");
  *out = nullptr;
  amhip_session* h = new (std::nothrow) amhip_session();
  if (!h) return AMHIP_ERR_NOMEM;
  Session& s = h->impl;
  s.grid = *grid;
  s.ti = tiles_i;
  s.tj = tiles_j;
  s.always_copy = tuning_on("session_always_copy");
  s.verify_partial = tuning_on("session_verify_partial");
  // window edges on multiples of the gather tile (64 x 32 cells) except at the map border
  auto edges = [](int n, int parts, int align, std::vector<int>* e) {
    e->assign(1, 0);
    for (int k = 1; k < parts; ++k) {
      int v = (int)std::llround((double)n * k / parts / align) * align;
      v = std::min(std::max(v, e->back() + 1), n - (parts - k));
      e->push_back(v);
    }
    e->push_back(n);
  };
  edges(grid->rows, tiles_i, 64, &s.edges_i);
  edges(grid->cols, tiles_j, 32, &s.edges_j);
  const int W = tiles_i * tiles_j;
  int rc = AMHIP_OK;
  for (int k = 0; k < W && rc == AMHIP_OK; ++k) {
    const int a = k % tiles_i, b = k / tiles_i;
    Win w = {s.edges_i[a], s.edges_j[b], s.edges_i[a + 1] - s.edges_i[a],
             s.edges_j[b + 1] - s.edges_j[b]};
    amhip_ctx* c = nullptr;
    const int d = devices ? devices[k] : 0;
    rc = amhip_ctx_create_window(grid, w.i0, w.j0, w.rows, w.cols, d, &c);
    if (rc) break;
    s.ctx.push_back(c);
    s.win.push_back(w);
    s.dev.push_back(d);
    if ((rc = ctx_use_device(&c->impl))) break;
    unsigned long long* dh = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dh), sizeof(unsigned long long) * 2 * AMHIP_NUM_LAYERS) !=
        hipSuccess) {
      rc = hip_fail(hipGetLastError(), "hipMalloc(session scratch)", __FILE__, __LINE__);
      break;
    }
    s.dev_hash.push_back(dh);
    unsigned long long* ph = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&ph), sizeof(unsigned long long) * 2 * AMHIP_NUM_LAYERS,
                      hipHostMallocDefault) != hipSuccess) {
      rc = hip_fail(hipGetLastError(), "hipHostMalloc(session scratch)", __FILE__, __LINE__);
      break;
    }
    s.pin_hash.push_back(ph);
    hipStream_t aux = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess || hipEventCreate(&e2) != hipSuccess || hipEventCreate(&e3) != hipSuccess) {
      rc = hip_fail(hipGetLastError(), "session: second stream / timing events", __FILE__, __LINE__);
      if (aux) (void)hipStreamDestroy(aux);
      for (hipEvent_t e : {e0, e1, e2, e3})
        if (e) (void)hipEventDestroy(e);
      break;
    }
    s.aux_stream.push_back(aux);
    s.ev_k0.push_back(e0);
    s.ev_k1.push_back(e1);
    s.ev_d1.push_back(e2);
    s.ev_s1.push_back(e3);
  }
  if (rc) {
    amhip_session_destroy(h);
    return rc;
  }
  s.sync.assign((size_t)W * AMHIP_NUM_LAYERS, LayerSync());
  // content sums of the windows' initial constants: summed on the device without reading
  // memory (the non-linear mix has no closed form)
  s.const_hash.resize((size_t)W * AMHIP_NUM_LAYERS);
  for (int k = 0; k < W && rc == AMHIP_OK; ++k) {
    Ctx* c = &s.ctx[k]->impl;
    if ((rc = ctx_use_device(c))) break;
    hipError_t e = hipMemsetAsync(s.dev_hash[k], 0, sizeof(unsigned long long) * 2 * AMHIP_NUM_LAYERS,
                                  c->stream);
    for (int l = 0; l < AMHIP_NUM_LAYERS && e == hipSuccess; ++l)
      hipLaunchKernelGGL(k_layer_hash, dim3(1024), dim3(256), 0, c->stream, (const float*)nullptr,
                         ctx_layer_init_value(l), s.win[k].rows, s.win[k].cols, s.win[k].i0,
                         s.win[k].j0, s.grid.rows, s.dev_hash[k] + 2 * l);
    unsigned long long got[2 * AMHIP_NUM_LAYERS];
    if (e == hipSuccess)
      e = hipMemcpyAsync(got, s.dev_hash[k], sizeof(got), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      rc = hip_fail(e, "session: content sums of the initial layers", __FILE__, __LINE__);
      break;
    }
    for (int l = 0; l < AMHIP_NUM_LAYERS; ++l) {
      s.const_hash[(size_t)k * AMHIP_NUM_LAYERS + l].a = got[2 * l];
      s.const_hash[(size_t)k * AMHIP_NUM_LAYERS + l].b = got[2 * l + 1];
    }
  }
  if (rc) {
    amhip_session_destroy(h);
    return rc;
  }
  s.route_out.assign(W, nullptr);
  s.route_done.assign(W, nullptr);
  s.route_cap.assign(W, 0);
  s.route_counts.assign(W, nullptr);
  s.cloud.assign(W, nullptr);
  s.cloud_cap.assign(W, 0);
  s.pin.assign(W, nullptr);
  s.pin_cap.assign(W, 0);
  // direct device-to-device copies where the hardware allows them (xGMI)
  for (int a = 0; a < W; ++a)
    for (int b = 0; b < W; ++b)
      if (s.dev[a] != s.dev[b]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, s.dev[a], s.dev[b]) == hipSuccess && can &&
            hipSetDevice(s.dev[a]) == hipSuccess)
          (void)hipDeviceEnablePeerAccess(s.dev[b], 0);
        (void)hipGetLastError();  // (already enabled is fine)
      }
  *out = h;
  return AMHIP_OK;
}

void amhip_session_destroy(amhip_session* h) {
  if (!h) return;
  Session& s = h->impl;
  for (size_t k = 0; k < s.ctx.size(); ++k) {
    if (s.ctx[k]) {
      (void)ctx_use_device(&s.ctx[k]->impl);
      (void)hipStreamSynchronize(s.ctx[k]->impl.stream);
      if (k < s.route_out.size() && s.route_out[k]) (void)hipFree(s.route_out[k]);
      if (k < s.route_counts.size() && s.route_counts[k]) (void)hipFree(s.route_counts[k]);
      if (k < s.route_done.size() && s.route_done[k]) (void)hipEventDestroy(s.route_done[k]);
      if (k < s.cloud.size() && s.cloud[k]) (void)hipFree(s.cloud[k]);
      if (k < s.dev_hash.size() && s.dev_hash[k]) (void)hipFree(s.dev_hash[k]);
      if (k < s.pin.size() && s.pin[k]) (void)hipHostFree(s.pin[k]);
      if (k < s.pin_hash.size() && s.pin_hash[k]) (void)hipHostFree(s.pin_hash[k]);
      if (k < s.aux_stream.size() && s.aux_stream[k]) (void)hipStreamDestroy(s.aux_stream[k]);
      for (auto* v : {&s.ev_k0, &s.ev_k1, &s.ev_s1, &s.ev_d1})
        if (k < v->size() && (*v)[k]) (void)hipEventDestroy((*v)[k]);
      amhip_ctx_destroy(s.ctx[k]);
    }
  }
  delete h;
}

int amhip_session_num_windows(const amhip_session* h) { return h ? h->impl.W() : 0; }

amhip_ctx* amhip_session_context(amhip_session* h, int k) {
  if (!h || k < 0 || k >= h->impl.W()) return nullptr;
  return h->impl.ctx[k];
}

int amhip_session_window(const amhip_session* h, int k, int32_t* i0_j0_rows_cols) {
  if (!h || k < 0 || k >= h->impl.W() || !i0_j0_rows_cols)
    return arg_failure("amhip_session_window: bad argument");
  const Win& w = h->impl.win[k];
  i0_j0_rows_cols[0] = w.i0;
  i0_j0_rows_cols[1] = w.j0;
  i0_j0_rows_cols[2] = w.rows;
  i0_j0_rows_cols[3] = w.cols;
  return AMHIP_OK;
}

int amhip_session_set_always_copy(amhip_session* h, int on) {
  if (!h) return arg_failure("null session");
  h->impl.always_copy = on != 0;
  for (auto& st : h->impl.sync) st.device_valid = st.host_known = false;
  return AMHIP_OK;
}

int amhip_session_transfer_stats(const amhip_session* h, uint64_t* uploaded_bytes,
                                 uint64_t* downloaded_bytes) {
  if (!h) return arg_failure("null session");
  if (uploaded_bytes) *uploaded_bytes = h->impl.up_bytes.load();
  if (downloaded_bytes) *downloaded_bytes = h->impl.down_bytes.load();
  return AMHIP_OK;
}

int amhip_session_last_profile(const amhip_session* h, double* out8) {
  if (!h || !out8) return arg_failure("amhip_session_last_profile: null argument");
  const CallProfile& p = h->impl.prof;
  out8[0] = p.total_ms;
  out8[1] = p.h2d_ms;
  out8[2] = p.host_sum_ms;
  out8[3] = p.kernel_ms;
  out8[4] = p.dev_sum_wait_ms;
  out8[5] = p.d2h_ms;
  out8[6] = p.up_bytes;
  out8[7] = (double)h->impl.prof_down.load();
  return AMHIP_OK;
}

int amhip_session_set_dsm_precision(amhip_session* h, int mode) {
  if (!h) return arg_failure("null session");
  for (amhip_ctx* c : h->impl.ctx) {
    const int rc = amhip_ctx_set_dsm_precision(c, mode);
    if (rc) return rc;
  }
  return AMHIP_OK;
}

// dsm::Dsm::process (dsm.cc:186-201) on host buffers: `elevation` is the GridMap's matrix
// (map rows x cols, column-major), read and written like the reference does.
int amhip_session_dsm_process(amhip_session* h, const double* host_xyz, size_t n, int radius_sq,
                              double center_easting, double center_northing, float* elevation) {
  if (!h) return arg_failure("null session");
  if (n == 0) return AMHIP_OK;  // "Passed empty point cloud to DSM module" (dsm.cc:189-192)
  if (!host_xyz || !elevation) return arg_failure("amhip_session_dsm_process: null buffer");
  if (radius_sq <= 0) return arg_failure("interpolation_radius must be >= 1");
  Session& s = h->impl;
  const int W = s.W();
  // (1) what do the devices already hold of `elevation`?  (host threads; the cloud's upload
  // is enqueued first where there is a single window, so that both overlap)
  std::vector<Hash128> hh;
  const float* mats[1] = {elevation};
  int rc;
  const auto call0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  s.prof = CallProfile();
  s.prof_down = 0;
  s.prof.up_bytes = (double)n * 24.0;
  if (W == 1) {
    PhaseClock clock("dsm");
    Ctx* c = &s.ctx[0]->impl;
    if ((rc = ctx_use_device(c))) return rc;
    if (n >= 0x7FFFFFFFull) return arg_failure("more than 2^31-1 points");
    if ((rc = ensure_capacity(&c->stage_points, &c->stage_points_cap, 3 * n))) return rc;
    // (a pageable source makes this call return only once the data is staged; the sums run
    // in a second thread meanwhile)
    double sum_ms = 0.0;
    std::thread hasher([&]() {
      const auto t = std::chrono::steady_clock::now();
      if (!s.always_copy) host_hashes(s, mats, 1, &hh);
      else hh.assign(1, Hash128());
      sum_ms = ms_since(t);
    });
    hipError_t e = hipMemcpyAsync(c->stage_points, host_xyz, 3 * n * sizeof(double),
                                  hipMemcpyHostToDevice, c->stream);
    s.prof.h2d_ms = ms_since(call0);
    clock.mark("cloud enqueued");
    hasher.join();
    s.prof.host_sum_ms = sum_ms;
    clock.mark("host content sum");
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(cloud)", __FILE__, __LINE__);
    if ((rc = sync_in(s, 0, AMHIP_LAYER_ELEVATION, elevation, hh[0], true))) return rc;
    AMHIP_TRY(hipEventRecord(s.ev_k0[0], c->stream));
    if ((rc = amhip_dsm_process_dev(s.ctx[0], c->stage_points, n, radius_sq, center_easting,
                                    center_northing)))
      return rc;
    clock.mark("sync_in + dsm enqueued");
    const int lay[1] = {AMHIP_LAYER_ELEVATION};
    float* outs[1] = {elevation};
    if ((rc = sync_out(s, 0, lay, outs, 1, true))) return rc;
    rc = ctx_fetch_status(c);
    s.prof.total_ms = ms_since(call0);
    return rc;
  }

  if (!s.always_copy) host_hashes(s, mats, 1, &hh);
  else hh.assign(W, Hash128());
  // (2) slice k of the cloud goes to window k's device, which COUNTS what every window needs
  // of it (its cells grown by the halo margin) -- one pass without output --
  const double margin = halo_margin_m(radius_sq, s.grid.resolution);
  std::vector<int32_t> wins(4 * (size_t)W);
  for (int d = 0; d < W; ++d) {
    wins[4 * d + 0] = s.win[d].i0;
    wins[4 * d + 1] = s.win[d].j0;
    wins[4 * d + 2] = s.win[d].rows;
    wins[4 * d + 3] = s.win[d].cols;
  }
  std::vector<size_t> lo(W + 1);
  for (int k = 0; k <= W; ++k) lo[k] = n * (size_t)k / (size_t)W;
  std::vector<long long> counts((size_t)W * W, 0);
  // one selection pass of slice k over destinations [d0, d0 + nd): count only (caps 0) or
  // compacting into route_out[k] at the offsets the counts gave
  auto select_pass = [&](int k, bool count_only, const std::vector<unsigned long long>* offs) -> int {
    Ctx* c = &s.ctx[k]->impl;
    const size_t nk = lo[k + 1] - lo[k];
    for (int d0 = 0; d0 < W; d0 += kMaxHaloDests) {
      const int nd = std::min(kMaxHaloDests, W - d0);
      HaloParams hp;
      int r = make_halo_params(*c, center_easting, center_northing, &wins[4 * (size_t)d0], nd, margin,
                               0, &hp);
      if (r) return r;
      for (int d = 0; d < nd; ++d) {
        hp.off[d] = count_only ? 0ull : (*offs)[(size_t)k * W + d0 + d];
        hp.cap_d[d] = count_only ? 0ull : (unsigned long long)counts[(size_t)k * W + d0 + d];
      }
      if ((r = halo_select_run(c, c->stage_points, nk, hp, s.route_out[k],
                               reinterpret_cast<unsigned long long*>(s.route_counts[k] + d0))))
        return r;
    }
    return AMHIP_OK;
  };
  rc = for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    const size_t nk = lo[k + 1] - lo[k];
    if ((r = sync_in(s, k, AMHIP_LAYER_ELEVATION, elevation, hh[k], true))) return r;
    if (!s.route_done[k])
      AMHIP_TRY(hipEventCreateWithFlags(&s.route_done[k], hipEventDisableTiming));
    if (nk == 0) return AMHIP_OK;
    if ((r = ensure_capacity(&c->stage_points, &c->stage_points_cap, 3 * nk))) return r;
    AMHIP_TRY(hipMemcpyAsync(c->stage_points, host_xyz + 3 * lo[k], 3 * nk * sizeof(double),
                             hipMemcpyHostToDevice, c->stream));
    if (!s.route_counts[k])
      AMHIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.route_counts[k]), sizeof(long long) * W));
    if ((r = select_pass(k, true, nullptr))) return r;
    AMHIP_TRY(hipMemcpyAsync(&counts[(size_t)k * W], s.route_counts[k], sizeof(long long) * W,
                             hipMemcpyDeviceToHost, c->stream));
    // (the one host synchronisation of the routing: the sizes of everything that follows)
    AMHIP_TRY(hipStreamSynchronize(c->stream));
    return AMHIP_OK;
  });
  if (rc) return rc;
  // (3) the same pass again, now writing: slice k's selections for window d land behind those
  // for windows < d in ONE buffer of (points selected) rows -- not (slice x windows) --, and an
  // event marks them complete; no host synchronisation from here to the DSM's own
  std::vector<unsigned long long> offs((size_t)W * W, 0ull);
  for (int k = 0; k < W; ++k) {
    unsigned long long run = 0;
    for (int d = 0; d < W; ++d) {
      offs[(size_t)k * W + d] = run;
      run += (unsigned long long)counts[(size_t)k * W + d];
    }
  }
  rc = for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    const size_t nk = lo[k + 1] - lo[k];
    size_t sel = 0;
    for (int d = 0; d < W; ++d) sel += (size_t)counts[(size_t)k * W + d];
    if (nk && sel) {
      if ((r = ensure_capacity(&s.route_out[k], &s.route_cap[k], 3 * sel))) return r;
      if ((r = select_pass(k, false, &offs))) return r;
    }
    AMHIP_TRY(hipEventRecord(s.route_done[k], c->stream));
    return AMHIP_OK;
  });
  if (rc) return rc;
  // (4) every window collects its points from all slices, device to device (xGMI peer copies
  // behind the producers' events), and runs Dsm::process on them
  rc = for_windows(s, [&](int d) -> int {
    Ctx* c = &s.ctx[d]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    size_t total = 0;
    for (int k = 0; k < W; ++k) total += (size_t)counts[(size_t)k * W + d];
    if (total >= 0x7FFFFFFFull) return arg_failure("more than 2^31-1 points in one window");
    if (total) {
      if ((r = ensure_capacity(&s.cloud[d], &s.cloud_cap[d], 3 * total))) return r;
      size_t off = 0;
      for (int k = 0; k < W; ++k) {
        const size_t cnt = (size_t)counts[(size_t)k * W + d];
        if (!cnt) continue;
        if (k != d) AMHIP_TRY(hipStreamWaitEvent(c->stream, s.route_done[k], 0));
        const double* src = s.route_out[k] + 3 * (size_t)offs[(size_t)k * W + d];
        if (s.dev[k] == s.dev[d])
          AMHIP_TRY(hipMemcpyAsync(s.cloud[d] + 3 * off, src, cnt * 24, hipMemcpyDeviceToDevice,
                                   c->stream));
        else
          AMHIP_TRY(hipMemcpyPeerAsync(s.cloud[d] + 3 * off, s.dev[d], src, s.dev[k], cnt * 24,
                                       c->stream));
        off += cnt;
      }
      AMHIP_TRY(hipEventRecord(s.ev_k0[d], c->stream));
      if ((r = amhip_dsm_process_dev(s.ctx[d], s.cloud[d], total, radius_sq, center_easting,
                                     center_northing)))
        return r;
    } else {
      AMHIP_TRY(hipEventRecord(s.ev_k0[d], c->stream));
    }
    const int lay[1] = {AMHIP_LAYER_ELEVATION};
    float* outs[1] = {elevation};
    if ((r = sync_out(s, d, lay, outs, 1, total != 0))) return r;  // (no call: no dirty cells on record)
    return ctx_fetch_status(c);
  });
  // (a later call may overwrite route_out[k] while a slower peer still reads it: every window
  // finished its copies before its own sync_out returned, and all threads were joined)
  s.prof.total_ms = ms_since(call0);
  return rc;
}

// ortho::OrthoBackwardGrid::process (ortho-backward-grid.cc:223-239) on host buffers.
int amhip_session_ortho_backward_process(
    amhip_session* h, const amhip_camera* cam, const double* host_T_G_C, size_t F,
    const uint8_t* const* images, const size_t* steps, int channels, int colored,
    const float* elevation, float* elevation_angle, float* observation_index,
    float* num_observations, float* ortho, float* colored_ortho) {
  if (!h || !cam) return arg_failure("null session / camera");
  if (F == 0 || !host_T_G_C || !images || !steps)
    return arg_failure("empty pose / image list (CHECK(!T_G_Bs.empty()))");
  if (cam->width <= 0 || cam->height <= 0) return arg_failure("bad image size");
  if (!elevation || !elevation_angle || !observation_index || !num_observations ||
      !(colored ? colored_ortho : ortho))
    return arg_failure("amhip_session_ortho_backward_process: null layer");
  Session& s = h->impl;
  const int W = s.W();
  const size_t row = (size_t)cam->width * (size_t)channels;
  const size_t frame = row * (size_t)cam->height;
  for (size_t f = 0; f < F; ++f) {
    if (!images[f]) return arg_failure("null image");
    if (steps[f] < row) return arg_failure("image step smaller than a row");
  }
  // layer order of the context: ORTHO, ELEVATION, ELEVATION_ANGLE, NUM_OBSERVATIONS,
  // OBSERVATION_INDEX, COLORED_ORTHO
  // (the mosaic touches ONE of the two output layers -- ortho-backward-grid.cc:186-208 --: the
  // other matrix is neither summed, uploaded nor downloaded; 400 MB of host hashing per call)
  const float* ins[AMHIP_NUM_LAYERS] = {colored ? nullptr : ortho, elevation, elevation_angle,
                                        num_observations, observation_index,
                                        colored ? colored_ortho : nullptr};
  std::vector<Hash128> hh;
  auto upload_frames = [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    if ((r = ensure_capacity(&c->stage_frames, &c->stage_frames_cap, frame * F))) return r;
    for (size_t f = 0; f < F; ++f) {
      if (steps[f] == row)
        AMHIP_TRY(hipMemcpyAsync(c->stage_frames + f * frame, images[f], frame,
                                 hipMemcpyHostToDevice, c->stream));
      else
        AMHIP_TRY(hipMemcpy2DAsync(c->stage_frames + f * frame, row, images[f], steps[f], row,
                                   (size_t)cam->height, hipMemcpyHostToDevice, c->stream));
    }
    return AMHIP_OK;
  };
  // the frames go up while host threads hash the six matrices
  PhaseClock clock("ortho_backward");
  const auto call0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  s.prof = CallProfile();
  s.prof_down = 0;
  s.prof.up_bytes = (double)frame * (double)F;
  int rc_up = AMHIP_OK;
  std::string up_msg;
  double up_ms = 0.0;
  std::thread uploader([&]() {
    rc_up = for_windows(s, upload_frames);
    if (rc_up) up_msg = amhip_last_error();
    up_ms = ms_since(call0);
  });
  if (!s.always_copy) host_hashes(s, ins, AMHIP_NUM_LAYERS, &hh);
  else hh.assign((size_t)AMHIP_NUM_LAYERS * W, Hash128());
  s.prof.host_sum_ms = ms_since(call0);
  clock.mark("host content sums");
  uploader.join();
  s.prof.h2d_ms = up_ms;
  clock.mark("frames enqueued");
  if (rc_up) {
    set_last_error(up_msg);
    return rc_up;
  }
  const int rc_all = for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    // (the mosaic reads the elevation and writes the other layers)
    for (int l = 0; l < AMHIP_NUM_LAYERS; ++l)
      if (ins[l] && (r = sync_in(s, k, l, ins[l], hh[(size_t)l * W + k], l != AMHIP_LAYER_ELEVATION)))
        return r;
    if (k == 0) clock.mark("sync_in");
    AMHIP_TRY(hipEventRecord(s.ev_k0[k], c->stream));
    if ((r = amhip_ortho_backward_process_dev(s.ctx[k], cam, host_T_G_C, F, c->stage_frames, frame,
                                              row, channels, colored)))
      return r;
    if (k == 0) clock.mark("mosaic enqueued");
    const int lay[5] = {AMHIP_LAYER_ORTHO, AMHIP_LAYER_ELEVATION_ANGLE, AMHIP_LAYER_NUM_OBSERVATIONS,
                        AMHIP_LAYER_OBSERVATION_INDEX, AMHIP_LAYER_COLORED_ORTHO};
    float* outs[5] = {colored ? nullptr : ortho, elevation_angle, num_observations, observation_index,
                      colored ? colored_ortho : nullptr};
    if ((r = sync_out(s, k, lay, outs, 5, true))) return r;
    return ctx_fetch_status(c);
  });
  s.prof.total_ms = ms_since(call0);
  return rc_all;
}

// ortho::OrthoFromPcl::process (ortho-from-pcl.cc:20-113) on host buffers: `ortho` = the
// GridMap's matrix.  Every window gets the whole cloud (its binning drops what lies outside the
// window + margin): intensities travel with their points, and this rank-1 "next" path is not
// worth a routing pass of its own.
int amhip_session_ortho_from_pcl_process(amhip_session* h, const double* host_xyz,
                                         const int32_t* host_intensities, size_t n, int radius_sq,
                                         int adaptive, float* ortho) {
  if (!h) return arg_failure("null session");
  if (n == 0 || !host_xyz || !host_intensities || !ortho)
    return arg_failure("empty point cloud / null buffer (CHECK(!pointcloud.empty()))");
  if (n >= 0x7FFFFFFFull) return arg_failure("more than 2^31-1 points");
  Session& s = h->impl;
  std::vector<Hash128> hh;
  const float* mats[1] = {ortho};
  const auto call0 = std::chrono::steady_clock::now();
  s.prof = CallProfile();
  s.prof_down = 0;
  s.prof.up_bytes = (double)n * 28.0 * (double)s.W();   // (every window gets the whole cloud)
  if (!s.always_copy) host_hashes(s, mats, 1, &hh);
  else hh.assign(s.W(), Hash128());
  s.prof.host_sum_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call0).count();
  const int rc_all = for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    if ((r = sync_in(s, k, AMHIP_LAYER_ORTHO, ortho, hh[k], true))) return r;
    if ((r = ensure_capacity(&c->stage_points, &c->stage_points_cap, 3 * n))) return r;
    if ((r = ensure_capacity(&c->stage_values, &c->stage_values_cap, n))) return r;
    AMHIP_TRY(hipMemcpyAsync(c->stage_points, host_xyz, 3 * n * sizeof(double),
                             hipMemcpyHostToDevice, c->stream));
    AMHIP_TRY(hipMemcpyAsync(c->stage_values, host_intensities, n * sizeof(int32_t),
                             hipMemcpyHostToDevice, c->stream));
    AMHIP_TRY(hipEventRecord(s.ev_k0[k], c->stream));
    if ((r = amhip_ortho_from_pcl_process_dev(s.ctx[k], c->stage_points, c->stage_values, n,
                                              radius_sq, adaptive)))
      return r;
    const int lay[1] = {AMHIP_LAYER_ORTHO};
    float* outs[1] = {ortho};
    if ((r = sync_out(s, k, lay, outs, 1))) return r;
    return ctx_fetch_status(c);
  });
  s.prof.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - call0).count();
  return rc_all;
}

// ---- what leaves the map (formats: amhip_export.hip) ---------------

// grid_map_msgs/GridMap in ROS 1 wire format (AerialGridMap::publishOnce,
// aerial-mapper-grid-map.cc:66-72: setTimestamp + GridMapRosConverter::toMessage + publish):
// the resident layers travel from the devices STRAIGHT into the message buffer -- no host matrix
// in between.  layer_ids[l] = the amhip layer behind message layer l, or -1: then host_layers[l]
// (a rows x cols column-major matrix) is copied, or, if that is null, the layer is NaN (the three
// layers of AerialGridMap the path never touches).
int amhip_session_grid_map_msg(amhip_session* h, uint64_t stamp_ns, const char* frame_id,
                               int num_layers, const char* const* layer_names,
                               const int32_t* layer_ids, const float* const* host_layers,
                               uint8_t* out, size_t cap, size_t* written) {
  if (!h || !frame_id || num_layers < 1 || num_layers > 64 || !layer_names || !layer_ids || !out)
    return arg_failure("amhip_session_grid_map_msg: bad argument");
  Session& s = h->impl;
  for (int l = 0; l < num_layers; ++l)
    if (layer_ids[l] >= AMHIP_NUM_LAYERS) return arg_failure("amhip_session_grid_map_msg: bad layer id");
  std::vector<size_t> at(num_layers);
  int rc = amhip_grid_map_msg_layout(&s.grid, stamp_ns, frame_id, num_layers, layer_names, out, cap,
                                     at.data());
  if (rc) return rc;
  const size_t cells = (size_t)s.grid.rows * (size_t)s.grid.cols;
  // the layers without a device copy are filled by host threads WHILE the others travel
  std::vector<std::thread> fillers;
  {
    std::vector<float> nan_block(std::min<size_t>(cells, 1u << 16), std::nanf(""));
    const size_t nb = nan_block.size();
    auto fill = [&, nb](int l, size_t k0, size_t k1, const std::vector<float>* block) {
      uint8_t* p = out + at[l];  // (not aligned: bytes only)
      if (host_layers && host_layers[l]) {
        std::memcpy(p + 4 * k0, host_layers[l] + k0, 4 * (k1 - k0));
      } else {
        for (size_t k = k0; k < k1; k += nb)
          std::memcpy(p + 4 * k, block->data(), 4 * std::min(nb, k1 - k));
      }
    };
    const auto shared = std::make_shared<std::vector<float>>(std::move(nan_block));
    const int parts = cells >= (size_t(1) << 22) ? 4 : 1;
    for (int l = 0; l < num_layers; ++l) {
      if (layer_ids[l] >= 0) continue;
      for (int q = 0; q < parts; ++q) {
        const size_t k0 = cells * q / parts, k1 = cells * (q + 1) / parts;
        fillers.emplace_back([=]() { fill(l, k0, k1, shared.get()); });
      }
    }
  }
  rc = for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    const Win& w = s.win[k];
    for (int l = 0; l < num_layers; ++l) {
      const int id = layer_ids[l];
      if (id < 0) continue;
      if ((r = ctx_materialize(c, id))) return r;
      uint8_t* dst = out + at[l] + 4 * ((size_t)w.i0 + (size_t)w.j0 * (size_t)s.grid.rows);
      AMHIP_TRY(copy_window(dst, c->layers[id], s, w, true, c->stream));
    }
    AMHIP_TRY(hipStreamSynchronize(c->stream));
    return AMHIP_OK;
  });
  for (auto& t : fillers) t.join();
  if (rc) return rc;
  if (written) *written = amhip_grid_map_msg_bytes(&s.grid, frame_id, num_layers, layer_names);
  return AMHIP_OK;
}

// The whole map's layer as an image (grid_map_cv's toImage orientation: image row = index 0):
// every window is turned on its own device (k_layer_to_image) and lands in its block of the
// host image.
int amhip_session_layer_to_image(amhip_session* h, int layer, int bgr, float lower, float upper,
                                 uint8_t* host_image, size_t step) {
  if (!h || !host_image || layer < 0 || layer >= AMHIP_NUM_LAYERS)
    return arg_failure("amhip_session_layer_to_image: bad argument");
  Session& s = h->impl;
  const size_t bpp = bgr ? 3u : 1u;
  if (step < (size_t)s.grid.cols * bpp)
    return arg_failure("amhip_session_layer_to_image: step smaller than an image row");
  if (!bgr && !(upper > lower)) return arg_failure("amhip_session_layer_to_image: upper <= lower");
  return for_windows(s, [&](int k) -> int {
    Ctx* c = &s.ctx[k]->impl;
    int r = ctx_use_device(c);
    if (r) return r;
    const Win& w = s.win[k];
    const size_t row = (size_t)w.cols * bpp;
    uint8_t* dev = nullptr;
    AMHIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), row * (size_t)w.rows));
    r = amhip_layer_to_image_dev(s.ctx[k], layer, bgr, lower, upper, dev, row);
    if (r == AMHIP_OK) {
      hipError_t e = hipMemcpy2DAsync(host_image + (size_t)w.i0 * step + (size_t)w.j0 * bpp, step,
                                      dev, row, row, (size_t)w.rows, hipMemcpyDeviceToHost,
                                      c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) r = hip_fail(e, "image download", __FILE__, __LINE__);
    }
    (void)hipFree(dev);
    return r;
  });
}

}  // extern "C"
