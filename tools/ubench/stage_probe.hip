// stage_probe.hip -- what can a host-buffer call reach on this box?  (round 6)
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread tools/ubench/stage_probe.hip -o tools/ubench/stage_probe
// Measures, for a 1.2 GB pageable buffer (the cfg3 cloud) and a 0.4 GB one (a layer):
//   (1) hipMemcpyAsync straight from / to pageable memory (what amhip_session did until round 5)
//   (2) the same from / to pinned memory (the link's own rate)
//   (3) worker threads staging chunks through their own pinned buffers (memcpy -> DMA, DMA -> memcpy)
//   (4) plain multi-threaded memcpy and read rates (the host side's ceiling)
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Worker {
  char* pin[2] = {nullptr, nullptr};
  hipEvent_t ev[2];
};

// staged upload: worker t takes chunks t, t + T, ... ; memcpy into its pinned buffer, DMA, event
static double staged_up(char* dev, const char* host, size_t bytes, int T, size_t chunk, hipStream_t* streams,
                        int nstreams, std::vector<Worker>& ws) {
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  const double t0 = now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      Worker& w = ws[t];
      int q = 0;
      for (size_t c = t; c < nchunks; c += T, ++q) {
        const int b = q & 1;
        if (q >= 2) CK(hipEventSynchronize(w.ev[b]));
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        std::memcpy(w.pin[b], host + off, len);
        hipStream_t s = streams[t % nstreams];
        CK(hipMemcpyAsync(dev + off, w.pin[b], len, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(w.ev[b], s));
      }
    });
  for (auto& x : th) x.join();
  for (int i = 0; i < nstreams; ++i) CK(hipStreamSynchronize(streams[i]));
  return now() - t0;
}

static double staged_down(char* host, const char* dev, size_t bytes, int T, size_t chunk, hipStream_t* streams,
                          int nstreams, std::vector<Worker>& ws) {
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  const double t0 = now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      Worker& w = ws[t];
      hipStream_t s = streams[t % nstreams];
      size_t mine = 0;
      for (size_t c = t; c < nchunks; c += T) ++mine;
      // depth-2 pipeline: issue chunk q + 1 before draining chunk q
      auto issue = [&](size_t q) {
        const size_t c = t + q * T;
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        CK(hipMemcpyAsync(w.pin[q & 1], dev + off, len, hipMemcpyDeviceToHost, s));
        CK(hipEventRecord(w.ev[q & 1], s));
      };
      if (mine) issue(0);
      for (size_t q = 0; q < mine; ++q) {
        CK(hipEventSynchronize(w.ev[q & 1]));
        const size_t c = t + q * T;
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        // (buffer (q + 1) & 1 was drained at step q - 1)
        if (q + 1 < mine) issue(q + 1);
        std::memcpy(host + off, w.pin[q & 1], len);
      }
    });
  for (auto& x : th) x.join();
  return now() - t0;
}

int main(int argc, char** argv) {
  const size_t big = (size_t)1200 << 20, small = (size_t)400 << 20;
  char* host = nullptr;
  if (posix_memalign(reinterpret_cast<void**>(&host), 4096, big)) return 1;
  std::memset(host, 1, big);
  char* host2 = nullptr;
  if (posix_memalign(reinterpret_cast<void**>(&host2), 4096, big)) return 1;
  std::memset(host2, 2, big);
  char* dev = nullptr;
  CK(hipMalloc(reinterpret_cast<void**>(&dev), big));
  hipStream_t streams[4];
  for (auto& s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  // (1) pageable
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    CK(hipMemcpyAsync(dev, host, big, hipMemcpyHostToDevice, streams[0]));
    CK(hipStreamSynchronize(streams[0]));
    double t1 = now();
    CK(hipMemcpyAsync(host2, dev, small, hipMemcpyDeviceToHost, streams[0]));
    CK(hipStreamSynchronize(streams[0]));
    double t2 = now();
    std::printf("pageable  H2D 1.2GB %.2f ms (%.1f GB/s)   D2H 0.4GB %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3,
                big / (t1 - t0) / 1e9, (t2 - t1) * 1e3, small / (t2 - t1) / 1e9);
  }
  // (2) pinned
  {
    char* pin = nullptr;
    double t0 = now();
    CK(hipHostMalloc(reinterpret_cast<void**>(&pin), small, hipHostMallocDefault));
    double t1 = now();
    std::printf("hipHostMalloc 0.4GB %.2f ms\n", (t1 - t0) * 1e3);
    std::memset(pin, 3, small);
    for (int rep = 0; rep < 2; ++rep) {
      t0 = now();
      CK(hipMemcpyAsync(dev, pin, small, hipMemcpyHostToDevice, streams[0]));
      CK(hipStreamSynchronize(streams[0]));
      t1 = now();
      CK(hipMemcpyAsync(pin, dev, small, hipMemcpyDeviceToHost, streams[0]));
      CK(hipStreamSynchronize(streams[0]));
      double t2 = now();
      std::printf("pinned    H2D 0.4GB %.2f ms (%.1f GB/s)   D2H 0.4GB %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3,
                  small / (t1 - t0) / 1e9, (t2 - t1) * 1e3, small / (t2 - t1) / 1e9);
    }
    // duplex: H2D on one stream, D2H on another
    t0 = now();
    CK(hipMemcpyAsync(dev, pin, small / 2, hipMemcpyHostToDevice, streams[0]));
    CK(hipMemcpyAsync(pin + small / 2, dev + small / 2, small / 2, hipMemcpyDeviceToHost, streams[1]));
    CK(hipStreamSynchronize(streams[0]));
    CK(hipStreamSynchronize(streams[1]));
    t1 = now();
    std::printf("pinned duplex 0.2GB up + 0.2GB down %.2f ms\n", (t1 - t0) * 1e3);
    // register in place
    t0 = now();
    hipError_t e = hipHostRegister(host2, small, hipHostRegisterDefault);
    t1 = now();
    std::printf("hipHostRegister 0.4GB: %s %.2f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
    if (e == hipSuccess) {
      t0 = now();
      CK(hipMemcpyAsync(dev, host2, small, hipMemcpyHostToDevice, streams[0]));
      CK(hipStreamSynchronize(streams[0]));
      t1 = now();
      std::printf("registered H2D 0.4GB %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, small / (t1 - t0) / 1e9);
      t0 = now();
      CK(hipHostUnregister(host2));
      t1 = now();
      std::printf("hipHostUnregister %.2f ms\n", (t1 - t0) * 1e3);
    }
    (void)hipGetLastError();
    CK(hipHostFree(pin));
  }
  // (4) host memcpy / read rates
  for (int T : {1, 2, 4, 8, 12, 16, 24, 32}) {
    std::vector<std::thread> th;
    double t0 = now();
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        const size_t a = small * t / T, b = small * (t + 1) / T;
        std::memcpy(host2 + a, host + a, b - a);
      });
    for (auto& x : th) x.join();
    double t1 = now();
    std::atomic<unsigned long long> sink{0};
    th.clear();
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        const size_t a = small * t / T / 8, b = small * (t + 1) / T / 8;
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(host);
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (size_t k = a; k + 3 < b; k += 4) {
          s0 += p[k];
          s1 += p[k + 1];
          s2 += p[k + 2];
          s3 += p[k + 3];
        }
        sink += s0 + s1 + s2 + s3;
      });
    for (auto& x : th) x.join();
    double t2 = now();
    std::printf("host T=%2d  memcpy 0.4GB %.2f ms (%.1f GB/s copied)   read 0.4GB %.2f ms (%.1f GB/s)\n", T,
                (t1 - t0) * 1e3, small / (t1 - t0) / 1e9, (t2 - t1) * 1e3, small / (t2 - t1) / 1e9);
  }
  // (3) staged
  for (size_t chunk : {(size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20}) {
    for (int T : {4, 6, 8, 12, 16}) {
      for (int ns : {1, 2}) {
        std::vector<Worker> ws(T);
        for (auto& w : ws)
          for (int b = 0; b < 2; ++b) {
            CK(hipHostMalloc(reinterpret_cast<void**>(&w.pin[b]), chunk, hipHostMallocDefault));
            CK(hipEventCreateWithFlags(&w.ev[b], hipEventDisableTiming));
          }
        double up = 1e9, down = 1e9;
        for (int rep = 0; rep < 2; ++rep) {
          up = std::min(up, staged_up(dev, host, big, T, chunk, streams, ns, ws));
          down = std::min(down, staged_down(host2, dev, small, T, chunk, streams, ns, ws));
        }
        std::printf("staged chunk %zu MB T=%2d streams=%d  H2D 1.2GB %.2f ms (%.1f GB/s)   D2H 0.4GB %.2f ms (%.1f GB/s)\n",
                    chunk >> 20, T, ns, up * 1e3, big / up / 1e9, down * 1e3, small / down / 1e9);
        for (auto& w : ws)
          for (int b = 0; b < 2; ++b) {
            CK(hipHostFree(w.pin[b]));
            CK(hipEventDestroy(w.ev[b]));
          }
      }
    }
  }
  // correctness of the staged paths
  CK(hipMemset(dev, 0, big));
  {
    std::vector<Worker> ws(8);
    for (auto& w : ws)
      for (int b = 0; b < 2; ++b) {
        CK(hipHostMalloc(reinterpret_cast<void**>(&w.pin[b]), 4 << 20, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&w.ev[b], hipEventDisableTiming));
      }
    for (size_t k = 0; k < big; k += 4099) host[k] = (char)(k * 7);
    staged_up(dev, host, big, 8, 4 << 20, streams, 2, ws);
    std::memset(host2, 0, big);
    staged_down(host2, dev, big, 8, 4 << 20, streams, 2, ws);
    std::printf("staged round trip %s\n", std::memcmp(host, host2, big) == 0 ? "identical" : "DIFFERENT");
  }
  return 0;
}
