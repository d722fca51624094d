// amhip_io.hip -- the point-cloud text format on MI355X.
//
// Replaces io::AerialMapperIO::loadPointCloudFromFile
// (aerial_mapper_io/src/aerial-mapper-io.cc:309-347): a text file of
// whitespace separated records `x y z intensity` read with
// `infile >> x >> y >> z >> intensity`, points with z <= -100 dropped.  At the
// reference's configurations this is the step in front of dsm::Dsm::process
// (50 M points = 2.3 GB of text; iostream extraction runs at a few MB/s per
// core), and its output is exactly the AoS f64 cloud + int32 intensities the
// DSM / OrthoFromPcl entry points take, so the cloud can stay in HBM.
//
//   k_io_token_count / k_io_token_emit   token starts (non-space after space),
//                                        16 bytes per lane, block scan
//   k_io_parse      one lane per token: libstdc++'s num_get grammar, decimal ->
//                   double with the Eisel-Lemire algorithm (D. Lemire, "Number
//                   Parsing at a Gigabyte per Second", SP&E 2021: one or two
//                   64x64->128-bit products against a table of powers of five,
//                   correctly rounded; the rare undecidable inputs -- more than
//                   19 significant digits straddling a rounding boundary -- are
//                   flagged and re-done with strtod on the host)
//   k_io_tail       ONE lane, from the first record whose tokens are not each consumed whole by
//                   their field's extraction: the iostream loop itself, character by character
//                   (libstdc++'s num_get: what a double / an int extraction accepts, where it
//                   stops, where the NEXT extraction resumes -- "4.7" read as an int leaves ".7"
//                   for the next double; "1-2-3-4" is a whole record)
//   k_io_keep_count / k_io_keep_emit     z > -100 filter, file order kept
// The stream semantics are the reference's for EVERY input: while every whitespace-delimited token
// is consumed whole by its field (any well-formed file), tokens map to extractions one to one and
// are parsed in parallel; from the first one that is not, the sequential kernel takes over until
// the stream fails or ends.  An incomplete last record is dropped, as `while (infile >> ...)` does.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "amhip_common.h"
#include "amhip_pow5_table.h"

namespace amhip {

namespace {

__device__ __forceinline__ bool io_space(unsigned char c) {
  // std::isspace in the "C" locale (what operator>> skips)
  return c == ' ' || (c >= 9 && c <= 13);
}

constexpr int kIoThreads = 256;
constexpr int kIoBytesPerLane = 16;
constexpr size_t kIoChunk = (size_t)kIoThreads * kIoBytesPerLane;

// token starts among the 16 bytes of this lane (bit k set: byte k starts a token)
__device__ __forceinline__ unsigned lane_token_mask(const unsigned char* __restrict__ text,
                                                    size_t len, size_t pos) {
  if (pos >= len) return 0u;
  unsigned char b[kIoBytesPerLane];
  if (pos + kIoBytesPerLane <= len) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + pos);  // pos is 16-byte aligned
    memcpy(b, &v, 16);
  } else {
    for (int k = 0; k < kIoBytesPerLane; ++k) b[k] = pos + k < len ? text[pos + k] : (unsigned char)' ';
  }
  bool prev_space = pos == 0 ? true : io_space(text[pos - 1]);
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < kIoBytesPerLane; ++k) {
    const bool sp = io_space(b[k]);
    if (!sp && prev_space) m |= 1u << k;
    prev_space = sp;
  }
  return m;
}

__global__ void __launch_bounds__(kIoThreads)
k_io_token_count(const unsigned char* __restrict__ text, size_t len,
                 uint32_t* __restrict__ block_counts) {
  __shared__ unsigned s_sum[kIoThreads / 64];
  const size_t pos = (size_t)blockIdx.x * kIoChunk + (size_t)threadIdx.x * kIoBytesPerLane;
  unsigned c = __popc(lane_token_mask(text, len, pos));
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

__global__ void __launch_bounds__(kIoThreads)
k_io_token_emit(const unsigned char* __restrict__ text, size_t len,
                const uint32_t* __restrict__ block_start, uint64_t* __restrict__ tok) {
  __shared__ unsigned s_wave[kIoThreads / 64];
  const size_t pos = (size_t)blockIdx.x * kIoChunk + (size_t)threadIdx.x * kIoBytesPerLane;
  const unsigned m = lane_token_mask(text, len, pos);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned incl = __popc(m);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_wave[wid] = incl;
  __syncthreads();
  unsigned base = block_start[blockIdx.x];
  for (int w = 0; w < wid; ++w) base += s_wave[w];
  unsigned at = base + incl - __popc(m);
  for (unsigned mm = m; mm; mm &= mm - 1) tok[at++] = pos + (unsigned)(__ffs(mm) - 1);
}

// ---- exclusive scan of the per-block token counts (single block, loops) -------
__global__ void __launch_bounds__(1024)
k_io_scan(uint32_t* __restrict__ v, size_t n, unsigned long long* __restrict__ total) {
  __shared__ unsigned s_w[1024 / 64];
  __shared__ unsigned long long s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (size_t base = 0; base < n; base += 1024) {
    const size_t i = base + threadIdx.x;
    const unsigned x = i < n ? v[i] : 0u;
    unsigned incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    unsigned pre = 0;
    for (int w = 0; w < wid; ++w) pre += s_w[w];
    const unsigned long long carry = s_carry;
    if (i < n) v[i] = (uint32_t)(carry + pre + incl - x);
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + pre + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}

// ---- Eisel-Lemire ------------------------------------------------------------------
struct AdjMant {
  uint64_t mantissa;
  int power2;  // < 0: undecided, needs the slow path
};

__device__ __forceinline__ AdjMant eisel_lemire(int64_t q, uint64_t w,
                                                const uint64_t (*__restrict__ T)[2]) {
  AdjMant a;
  a.mantissa = 0;
  a.power2 = 0;
  if (w == 0 || q < kPow5Min) return a;  // zero
  if (q > kPow5Max) {
    a.power2 = 0x7FF;  // infinity
    return a;
  }
  const int lz = __clzll((long long)w);
  w <<= lz;
  const uint64_t* t = T[q - kPow5Min];
  uint64_t lo = w * t[0], hi = __umul64hi(w, t[0]);
  if ((hi & 0x1FFull) == 0x1FFull) {  // 55 bits of precision not settled: refine
    const uint64_t hi2 = __umul64hi(w, t[1]);
    lo += hi2;
    if (hi2 > lo) ++hi;
    if (lo == ~0ull && !(q >= -27 && q <= 55)) {
      a.power2 = -1;
      return a;
    }
  }
  const int upperbit = (int)(hi >> 63);
  const int shift = upperbit + 64 - 52 - 3;
  a.mantissa = hi >> shift;
  // floor(log2(10^q)) + 63, valid for |q| <= 350
  const int p10 = (int)(((152170 + 65536) * q) >> 16) + 63;
  a.power2 = p10 + upperbit - lz + 1023;
  if (a.power2 <= 0) {  // subnormal
    if (-a.power2 + 1 >= 64) {
      a.mantissa = 0;
      a.power2 = 0;
      return a;
    }
    a.mantissa >>= -a.power2 + 1;
    a.mantissa += (a.mantissa & 1);
    a.mantissa >>= 1;
    a.power2 = (a.mantissa < (1ull << 52)) ? 0 : 1;
    return a;
  }
  // exactly half-way: round to even
  if (lo <= 1 && q >= -4 && q <= 23 && ((a.mantissa & 3) == 1)) {
    if ((a.mantissa << shift) == hi) a.mantissa &= ~1ull;
  }
  a.mantissa += (a.mantissa & 1);
  a.mantissa >>= 1;
  if (a.mantissa >= (2ull << 52)) {
    a.mantissa = 1ull << 52;
    ++a.power2;
  }
  a.mantissa &= ~(1ull << 52);
  if (a.power2 >= 0x7FF) {
    a.mantissa = 0;
    a.power2 = 0x7FF;
  }
  return a;
}

enum : int { kTokOk = 0, kTokSlow = 1, kTokBad = 2, kTokOkThenStop = 3 };

// ---- what libstdc++'s num_get ACCEPTS from a character sequence (locale_facets.tcc, "C" locale:
// no grouping, '.' the decimal point).  Returns the number of characters taken; the conversion of
// exactly those characters (strtod / the integer rules) then succeeds or sets failbit.
// _M_extract_float: [+-] 0* then { digit | '.' once, not after the exponent | (e|E) once, only after
// a mantissa digit, followed by an optional sign }.
__device__ size_t accept_float(const unsigned char* s, size_t room) {
  size_t i = 0;
  if (i < room && (s[i] == '+' || s[i] == '-')) ++i;
  bool mantissa = false, dec = false, sci = false;
  while (i < room && s[i] == '0') {
    mantissa = true;
    ++i;
  }
  while (i < room) {
    const unsigned char c = s[i];
    if (c >= '0' && c <= '9') {
      mantissa = true;
      ++i;
    } else if (c == '.' && !dec && !sci) {
      dec = true;
      ++i;
    } else if ((c == 'e' || c == 'E') && !sci && mantissa) {
      sci = true;
      ++i;
      if (i < room && (s[i] == '+' || s[i] == '-')) ++i;
    } else {
      break;
    }
  }
  return i;
}
// _M_extract_int (base 10): [+-] digits, every digit taken even past an overflow
__device__ size_t accept_int(const unsigned char* s, size_t room) {
  size_t i = 0;
  if (i < room && (s[i] == '+' || s[i] == '-')) ++i;
  while (i < room && s[i] >= '0' && s[i] <= '9') ++i;
  return i;
}

// libstdc++ num_get<...>::do_get(double&): [+-] digits [. digits] [(e|E) [+-] digits]
__device__ int parse_double(const unsigned char* s, int len,
                            const uint64_t (*__restrict__ T)[2], double* out) {
  int i = 0;
  bool neg = false;
  if (i < len && (s[i] == '+' || s[i] == '-')) {
    neg = s[i] == '-';
    ++i;
  }
  uint64_t w = 0;
  int nd = 0;
  long long exp10 = 0;
  bool seen_digit = false, seen_nonzero = false, trunc_nonzero = false;
  for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i) {
    const unsigned d = s[i] - '0';
    seen_digit = true;
    if (!seen_nonzero && d == 0) continue;
    seen_nonzero = true;
    if (nd < 19) {
      w = w * 10 + d;
      ++nd;
    } else {
      ++exp10;
      if (d) trunc_nonzero = true;
    }
  }
  if (i < len && s[i] == '.') {
    ++i;
    for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i) {
      const unsigned d = s[i] - '0';
      seen_digit = true;
      if (!seen_nonzero && d == 0) {
        --exp10;
        continue;
      }
      seen_nonzero = true;
      if (nd < 19) {
        w = w * 10 + d;
        ++nd;
        --exp10;
      } else if (d) {
        trunc_nonzero = true;
      }
    }
  }
  if (!seen_digit) return kTokBad;
  if (i < len && (s[i] == 'e' || s[i] == 'E')) {
    ++i;
    bool eneg = false;
    if (i < len && (s[i] == '+' || s[i] == '-')) {
      eneg = s[i] == '-';
      ++i;
    }
    if (!(i < len && s[i] >= '0' && s[i] <= '9')) return kTokBad;  // "1e": strtod stops early
    long long e = 0;
    for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i)
      if (e < 100000) e = e * 10 + (s[i] - '0');
    exp10 += eneg ? -e : e;
  }
  if (i != len) return kTokBad;  // trailing characters
  AdjMant a = eisel_lemire(exp10, w, T);
  if (a.power2 < 0) return kTokSlow;
  if (trunc_nonzero) {  // more than 19 digits: decided only if w and w + 1 agree
    const AdjMant b = eisel_lemire(exp10, w + 1, T);
    if (b.power2 < 0 || b.mantissa != a.mantissa || b.power2 != a.power2) return kTokSlow;
  }
  if (a.power2 >= 0x7FF) return kTokBad;  // overflow: libstdc++ sets failbit
  const uint64_t bits = ((uint64_t)neg << 63) | ((uint64_t)a.power2 << 52) | a.mantissa;
  *out = __longlong_as_double((long long)bits);
  return kTokOk;
}

// num_get<...>::do_get(int&): [+-] digits; what follows a valid prefix stays in the stream
__device__ int parse_int(const unsigned char* s, int len, int32_t* out) {
  int i = 0;
  bool neg = false;
  if (i < len && (s[i] == '+' || s[i] == '-')) {
    neg = s[i] == '-';
    ++i;
  }
  if (!(i < len && s[i] >= '0' && s[i] <= '9')) return kTokBad;
  long long v = 0;
  for (; i < len && s[i] >= '0' && s[i] <= '9'; ++i) {
    if (v < (1ll << 40)) v = v * 10 + (s[i] - '0');
  }
  if (neg) v = -v;
  if (v > 2147483647ll || v < -2147483648ll) return kTokBad;  // failbit on overflow
  *out = (int32_t)v;
  return i == len ? kTokOk : kTokOkThenStop;
}

constexpr int kMaxTokenLen = 768;  // longer tokens go to the slow path

__global__ void __launch_bounds__(256)
k_io_parse(const unsigned char* __restrict__ text, size_t len, const uint64_t* __restrict__ tok,
           size_t ntok, const uint64_t (*__restrict__ T)[2], double* __restrict__ xyz,
           int32_t* __restrict__ inten, unsigned long long* __restrict__ first_bad,
           unsigned long long* __restrict__ slow_count, uint64_t* __restrict__ slow_list,
           size_t slow_cap, size_t nrec_cap) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntok) return;
  const size_t start = tok[t];
  const unsigned char* s = text + start;
  const size_t room = len - start;
  int n = 0;
  while ((size_t)n < room && n < kMaxTokenLen && !io_space(s[n])) ++n;
  const bool too_long = n == kMaxTokenLen && (size_t)n < room && !io_space(s[n]);
  const int field = (int)(t & 3);
  const size_t rec = t >> 2;
  // (tokens of an incomplete last record are only CHECKED: an anomaly among them still hands the
  // stream to k_io_tail, which may find a whole record in them)
  const bool store = rec < nrec_cap;
  int st;
  if (field < 3) {
    double v = 0.0;
    st = too_long ? kTokBad : parse_double(s, n, T, &v);
    if (st == kTokOk && store) xyz[3 * rec + field] = v;
  } else {
    int32_t v = 0;
    st = too_long ? kTokBad : parse_int(s, n, &v);
    if (st == kTokOk && store) inten[rec] = v;
    if (st == kTokOkThenStop) st = kTokBad;   // (what follows the digits is the next extraction's)
  }
  // an ANOMALY: the extraction does not consume this token whole (or fails on it) -- from this
  // token's record on the sequential kernel reads the stream
  if (st == kTokBad) atomicMin(first_bad, (unsigned long long)t);
  if (st == kTokSlow && store) {
    const unsigned long long k = atomicAdd(slow_count, 1ull);
    if (k < slow_cap) {
      slow_list[3 * k + 0] = start;
      slow_list[3 * k + 1] = (uint64_t)n;
      slow_list[3 * k + 2] = 3 * rec + (uint64_t)field;
    }
  }
}

// The iostream loop of aerial-mapper-io.cc:316-323 / :337-345 itself, on ONE lane, from byte `pos`
// (where record `rec0`'s first extraction begins) until an extraction fails or the text ends:
//   while (infile >> x >> y >> z >> intensity) { push; if (infile.eof()) break; }
// Records rec0, rec0 + 1, ... are stored while they fit `cap`; *nrec_out = records the loop
// completed in all (the caller enlarges the buffers and runs it again if that exceeds cap).
__global__ void k_io_tail(const unsigned char* __restrict__ text, size_t len, size_t pos, size_t rec0,
                          size_t cap, const uint64_t (*__restrict__ T)[2], double* __restrict__ xyz,
                          int32_t* __restrict__ inten, unsigned long long* __restrict__ nrec_out,
                          unsigned long long* __restrict__ slow_count, uint64_t* __restrict__ slow_list,
                          size_t slow_cap) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  size_t rec = rec0;
  for (;;) {
    double v[3] = {0.0, 0.0, 0.0};
    size_t slow_at[3] = {0, 0, 0}, slow_len[3] = {0, 0, 0};
    bool ok = true;
    for (int f = 0; f < 3 && ok; ++f) {
      while (pos < len && io_space(text[pos])) ++pos;   // (the sentry: skipws)
      const size_t span = accept_float(text + pos, len - pos);
      const int st = span <= 0x7FFFFFFFull ? parse_double(text + pos, (int)span, T, &v[f]) : kTokBad;
      if (st == kTokBad) ok = false;   // (nothing accepted, strtod would stop early, or overflow: failbit)
      if (st == kTokSlow) {
        slow_at[f] = pos;
        slow_len[f] = span;
      }
      pos += span;
    }
    if (!ok) break;
    while (pos < len && io_space(text[pos])) ++pos;
    const size_t span = accept_int(text + pos, len - pos);
    int32_t iv = 0;
    const int st = span <= 0x7FFFFFFFull ? parse_int(text + pos, (int)span, &iv) : kTokBad;
    if (st != kTokOk) break;           // (no digit, or outside int: failbit)
    pos += span;
    if (rec < cap) {
      for (int f = 0; f < 3; ++f) {
        xyz[3 * rec + f] = v[f];
        if (slow_len[f]) {
          const unsigned long long k = atomicAdd(slow_count, 1ull);
          if (k < slow_cap) {
            slow_list[3 * k + 0] = slow_at[f];
            slow_list[3 * k + 1] = slow_len[f];
            slow_list[3 * k + 2] = 3 * rec + (uint64_t)f;
          }
        }
      }
      inten[rec] = iv;
    }
    ++rec;
  }
  *nrec_out = rec;
}

// ---- z > -100, in file order -------------------------------------------------------
__global__ void __launch_bounds__(256)
k_io_keep_count(const double* __restrict__ xyz, size_t nrec, uint32_t* __restrict__ block_counts) {
  __shared__ unsigned s_sum[4];
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned c = (r < nrec && xyz[3 * r + 2] > -100.0) ? 1u : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

__global__ void __launch_bounds__(256)
k_io_keep_emit(const double* __restrict__ xyz, const int32_t* __restrict__ inten, size_t nrec,
               const uint32_t* __restrict__ block_start, double* __restrict__ out_xyz,
               int32_t* __restrict__ out_inten) {
  __shared__ unsigned s_wave[4];
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool keep = r < nrec && xyz[3 * r + 2] > -100.0;
  const unsigned long long m = __ballot(keep);
  if (lane == 0) s_wave[wid] = __popcll(m);
  __syncthreads();
  unsigned at = block_start[blockIdx.x];
  for (int w = 0; w < wid; ++w) at += s_wave[w];
  at += __popcll(m & ((1ull << lane) - 1ull));
  if (keep) {
    out_xyz[3 * (size_t)at + 0] = xyz[3 * r + 0];
    out_xyz[3 * (size_t)at + 1] = xyz[3 * r + 1];
    out_xyz[3 * (size_t)at + 2] = xyz[3 * r + 2];
    out_inten[at] = inten[r];
  }
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    AMHIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    return AMHIP_OK;
  }
  template <typename T>
  T* as() {
    return static_cast<T*>(p);
  }
  void* release() {
    void* q = p;
    p = nullptr;
    return q;
  }
};

int io_fail(const char* msg) {
  set_last_error(msg);
  return AMHIP_ERR_ARG;
}

}  // namespace

int io_parse_point_cloud(int device, const char* host_text, size_t len, double** out_xyz,
                         int32_t** out_inten, size_t* out_n, size_t* out_slow) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    set_last_error("no HIP device available (libaerial_mapper_hip has no CPU fallback)");
    return AMHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) return io_fail("amhip_io: bad device index");
  AMHIP_TRY(hipSetDevice(device));
  *out_xyz = nullptr;
  *out_inten = nullptr;
  *out_n = 0;
  if (out_slow) *out_slow = 0;
  if (len == 0) return AMHIP_OK;
  hipStream_t stream = nullptr;  // the default stream: this entry point is synchronous
  const size_t nblk = (len + kIoChunk - 1) / kIoChunk;
  if (nblk >= 0xFFFFFFFFull) return io_fail("amhip_io: text larger than 16 TB");
  DevBuf text, counts, total, table;
  int rc;
  if ((rc = text.alloc(len + 16))) return rc;
  if ((rc = counts.alloc((nblk + 1) * sizeof(uint32_t)))) return rc;
  if ((rc = total.alloc(4 * sizeof(unsigned long long)))) return rc;
  if ((rc = table.alloc(sizeof(kPow5Host)))) return rc;
  AMHIP_TRY(hipMemcpyAsync(text.p, host_text, len, hipMemcpyHostToDevice, stream));
  AMHIP_TRY(hipMemcpyAsync(table.p, kPow5Host, sizeof(kPow5Host), hipMemcpyHostToDevice, stream));
  const unsigned char* dtext = text.as<unsigned char>();
  hipLaunchKernelGGL(k_io_token_count, dim3((unsigned)nblk), dim3(kIoThreads), 0, stream, dtext,
                     len, counts.as<uint32_t>());
  hipLaunchKernelGGL(k_io_scan, dim3(1), dim3(1024), 0, stream, counts.as<uint32_t>(), nblk,
                     total.as<unsigned long long>());
  AMHIP_TRY(hipGetLastError());
  unsigned long long ntok = 0;
  AMHIP_TRY(hipMemcpyAsync(&ntok, total.p, sizeof(ntok), hipMemcpyDeviceToHost, stream));
  AMHIP_TRY(hipStreamSynchronize(stream));
  if (ntok >= 0xFFFFFFFFull) return io_fail("amhip_io: more than 2^32 tokens in one call");
  const size_t nrec_all = (size_t)(ntok / 4);
  if (ntok == 0) return AMHIP_OK;

  DevBuf tok, raw_xyz, raw_int, slow;
  const size_t slow_cap = 1 << 20;
  size_t rec_cap = nrec_all + 1;
  if ((rc = tok.alloc((size_t)ntok * sizeof(uint64_t)))) return rc;
  if ((rc = raw_xyz.alloc(rec_cap * 3 * sizeof(double)))) return rc;
  if ((rc = raw_int.alloc(rec_cap * sizeof(int32_t)))) return rc;
  if ((rc = slow.alloc(3 * slow_cap * sizeof(uint64_t)))) return rc;
  unsigned long long init[3] = {~0ull, 0ull, 0ull};  // first anomaly, slow_count, the tail's record count
  unsigned long long* ctl = total.as<unsigned long long>() + 1;
  AMHIP_TRY(hipMemcpyAsync(ctl, init, sizeof(init), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_io_token_emit, dim3((unsigned)nblk), dim3(kIoThreads), 0, stream, dtext, len,
                     counts.as<uint32_t>(), tok.as<uint64_t>());
  // every token is checked; the tokens of whole records are stored (an incomplete last record
  // never reaches the vectors)
  hipLaunchKernelGGL(k_io_parse, dim3((unsigned)((ntok + 255) / 256)), dim3(256), 0, stream,
                     dtext, len, tok.as<uint64_t>(), (size_t)ntok,
                     reinterpret_cast<const uint64_t(*)[2]>(table.p), raw_xyz.as<double>(),
                     raw_int.as<int32_t>(), ctl, ctl + 1, slow.as<uint64_t>(), slow_cap, nrec_all);
  AMHIP_TRY(hipGetLastError());
  unsigned long long res[3];
  AMHIP_TRY(hipMemcpyAsync(res, ctl, sizeof(res), hipMemcpyDeviceToHost, stream));
  AMHIP_TRY(hipStreamSynchronize(stream));
  size_t nrec = nrec_all;
  // (numbers the PARALLEL pass filed for strtod; those of records the tail re-reads are dropped)
  const unsigned long long nslow_parallel = res[1];
  size_t tail_from = ~size_t(0);
  if (res[0] != ~0ull) {
    // ---- the first token its extraction does not consume whole: from its record on, the
    // iostream loop itself (k_io_tail) ------------------------------------------------------
    const size_t rec0 = (size_t)(res[0] / 4);
    tail_from = rec0;
    uint64_t off0 = 0;
    AMHIP_TRY(hipMemcpy(&off0, tok.as<uint64_t>() + 4 * rec0, sizeof(off0), hipMemcpyDeviceToHost));
    for (int attempt = 0; attempt < 2; ++attempt) {
      // (the tail files its own strtod entries behind the parallel pass's)
      AMHIP_TRY(hipMemcpyAsync(ctl + 1, &nslow_parallel, sizeof(nslow_parallel), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k_io_tail, dim3(1), dim3(1), 0, stream, dtext, len, (size_t)off0, rec0, rec_cap,
                         reinterpret_cast<const uint64_t(*)[2]>(table.p), raw_xyz.as<double>(),
                         raw_int.as<int32_t>(), ctl + 2, ctl + 1, slow.as<uint64_t>(), slow_cap);
      AMHIP_TRY(hipGetLastError());
      AMHIP_TRY(hipMemcpyAsync(res, ctl, sizeof(res), hipMemcpyDeviceToHost, stream));
      AMHIP_TRY(hipStreamSynchronize(stream));
      nrec = (size_t)res[2];
      if (nrec <= rec_cap) break;
      if (attempt == 1) return io_fail("amhip_io: internal: the sequential reader's record count changed");
      // more records than tokens / 4 (tokens that hold several numbers): larger buffers, the
      // records in front of rec0 carried over, once more
      DevBuf big_xyz, big_int;
      if ((rc = big_xyz.alloc((nrec + 1) * 3 * sizeof(double)))) return rc;
      if ((rc = big_int.alloc((nrec + 1) * sizeof(int32_t)))) return rc;
      if (rec0) {
        AMHIP_TRY(hipMemcpyAsync(big_xyz.p, raw_xyz.p, rec0 * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        AMHIP_TRY(hipMemcpyAsync(big_int.p, raw_int.p, rec0 * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
      }
      AMHIP_TRY(hipStreamSynchronize(stream));
      std::swap(raw_xyz.p, big_xyz.p);
      std::swap(raw_int.p, big_int.p);
      rec_cap = nrec + 1;
    }
  }
  // ---- slow path: the host re-does the flagged numbers with strtod -------------------
  // (more than 19 significant digits straddling a rounding boundary; syntactically valid by
  // construction.  Entries of records the stream never completed are ignored.)
  size_t nslow = (size_t)res[1];
  if (nslow > slow_cap) return io_fail("amhip_io: too many tokens need the strtod path");
  if (nslow) {
    std::vector<uint64_t> list(3 * nslow);
    AMHIP_TRY(hipMemcpy(list.data(), slow.p, 3 * nslow * sizeof(uint64_t), hipMemcpyDeviceToHost));
    size_t first_bad_rec = ~size_t(0);
    for (size_t k = 0; k < nslow; ++k) {
      const size_t off = (size_t)list[3 * k], n = (size_t)list[3 * k + 1], dest = (size_t)list[3 * k + 2];
      if (dest / 3 >= nrec) continue;
      if (k < (size_t)nslow_parallel && dest / 3 >= tail_from) continue;   // (re-read by the tail)
      const std::string str(host_text + off, n);
      char* endp = nullptr;
      errno = 0;
      const double v = std::strtod(str.c_str(), &endp);
      const bool bad = endp == str.c_str() || *endp != '\0' || v == HUGE_VAL || v == -HUGE_VAL;
      if (bad) {   // (overflow: failbit -- the stream ends in this record)
        first_bad_rec = std::min(first_bad_rec, dest / 3);
        continue;
      }
      AMHIP_TRY(hipMemcpy(raw_xyz.as<double>() + dest, &v, sizeof(v), hipMemcpyHostToDevice));
    }
    if (first_bad_rec < nrec) nrec = first_bad_rec;
  }
  if (out_slow) *out_slow = nslow;
  if (nrec == 0) return AMHIP_OK;
  // ---- z > -100 -----------------------------------------------------------------------
  const size_t kblk = (nrec + 255) / 256;
  DevBuf kcounts;
  if ((rc = kcounts.alloc((kblk + 1) * sizeof(uint32_t)))) return rc;
  hipLaunchKernelGGL(k_io_keep_count, dim3((unsigned)kblk), dim3(256), 0, stream,
                     raw_xyz.as<double>(), nrec, kcounts.as<uint32_t>());
  hipLaunchKernelGGL(k_io_scan, dim3(1), dim3(1024), 0, stream, kcounts.as<uint32_t>(), kblk,
                     total.as<unsigned long long>());
  unsigned long long nkeep = 0;
  AMHIP_TRY(hipMemcpyAsync(&nkeep, total.p, sizeof(nkeep), hipMemcpyDeviceToHost, stream));
  AMHIP_TRY(hipStreamSynchronize(stream));
  DevBuf oxyz, oint;
  if ((rc = oxyz.alloc((size_t)nkeep * 3 * sizeof(double)))) return rc;
  if ((rc = oint.alloc((size_t)nkeep * sizeof(int32_t)))) return rc;
  hipLaunchKernelGGL(k_io_keep_emit, dim3((unsigned)kblk), dim3(256), 0, stream,
                     raw_xyz.as<double>(), raw_int.as<int32_t>(), nrec, kcounts.as<uint32_t>(),
                     oxyz.as<double>(), oint.as<int32_t>());
  AMHIP_TRY(hipGetLastError());
  AMHIP_TRY(hipStreamSynchronize(stream));
  *out_xyz = static_cast<double*>(oxyz.release());
  *out_inten = static_cast<int32_t*>(oint.release());
  *out_n = (size_t)nkeep;
  return AMHIP_OK;
}

}  // namespace amhip

extern "C" {

int amhip_io_parse_point_cloud_text(int device, const char* host_text, size_t len,
                                    double** dev_xyz, int32_t** dev_intensities,
                                    size_t* num_points, size_t* num_strtod_tokens) {
  if ((!host_text && len) || !dev_xyz || !dev_intensities || !num_points) {
    amhip::set_last_error("amhip_io_parse_point_cloud_text: null argument");
    return AMHIP_ERR_ARG;
  }
  return amhip::io_parse_point_cloud(device, host_text, len, dev_xyz, dev_intensities, num_points,
                                     num_strtod_tokens);
}

int amhip_io_download_point_cloud(const double* dev_xyz, const int32_t* dev_intensities, size_t n,
                                  double* host_xyz, int32_t* host_intensities) {
  if (n == 0) return AMHIP_OK;
  if (!dev_xyz) {
    amhip::set_last_error("amhip_io_download_point_cloud: null argument");
    return AMHIP_ERR_ARG;
  }
  if (host_xyz) AMHIP_TRY(hipMemcpy(host_xyz, dev_xyz, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
  if (host_intensities && dev_intensities)
    AMHIP_TRY(hipMemcpy(host_intensities, dev_intensities, n * sizeof(int32_t),
                        hipMemcpyDeviceToHost));
  return AMHIP_OK;
}

int amhip_io_free(void* dev_ptr) {
  if (dev_ptr) AMHIP_TRY(hipFree(dev_ptr));
  return AMHIP_OK;
}

}  // extern "C"
