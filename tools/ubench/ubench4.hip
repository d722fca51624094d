// ubench4 -- round 3: packed single precision (VOP3P v_pk_*_f32) for the DSM gather's candidate
// loop.  A lane serves TWO cells per candidate (rows j, j + 1): the two squared distances, the two
// hit masks, the two weight updates are pairs -- one packed instruction each, if those issue at the
// rate of a plain f32 instruction.
//
//   part P  cycles per wave-instruction per SIMD (s_memtime deltas, as ubench3 part A) of
//           v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 clamp, next to v_fma_f32
//   part S  semantics the loop relies on, checked on the part: clamp applies to both halves,
//           op_sel_hi broadcasts a low half, the clamped fma with a power-of-two scale IS the
//           comparison (0 / 1 exactly, nothing in between) for neighbouring floats
//   part L  the loop body as shipped in round 2 (v_cmpx hit blocks) against the packed body, same
//           LDS records, same harness as ubench3 part B; the sums must agree bit for bit
// Build: hipcc --offload-arch=gfx950 -O3 ubench4.hip -o ubench4 ; run: ./ubench4 [P|S|L]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);         \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// part P
// ---------------------------------------------------------------------------------------------
#define DEF_KERNEL_PK(NAME, BODY)                                                             \
  __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* stamps, int iters) { \
    extern __shared__ unsigned char dyn_lds[];                                                \
    const float f = threadIdx.x + 1.5f;                                                       \
    v2f a0 = {f, f + 1}, a1 = {f + 2, f + 3}, a2 = {f + 4, f + 5}, a3 = {f + 6, f + 7};       \
    if (iters < 0) dyn_lds[threadIdx.x] = 1;                                                  \
    __syncthreads();                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                           \
    for (int it = 0; it < iters; ++it) {                                                      \
      asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY                                    \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));                                 \
    }                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                           \
    if ((threadIdx.x & 63) == 0) {                                                            \
      unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));     \
      s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;                                             \
    }                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y; \
  }
DEF_KERNEL_PK(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n")
DEF_KERNEL_PK(k_pkmul, "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n")
DEF_KERNEL_PK(k_pkadd, "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n")
DEF_KERNEL_PK(k_pkfmac, "v_pk_fma_f32 %0, %0, %1, %2 clamp\n v_pk_fma_f32 %1, %1, %2, %3 clamp\n v_pk_fma_f32 %2, %2, %3, %0 clamp\n v_pk_fma_f32 %3, %3, %0, %1 clamp\n")
DEF_KERNEL_PK(k_pkfmab, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %1, %1, %2, %3 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %2, %2, %3, %0 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %3, %3, %0, %1 op_sel_hi:[1,1,0]\n")
// plain f32 in the same harness, as the yardstick
__global__ void __launch_bounds__(256) k_fma32(float* out, unsigned long long* stamps, int iters) {
  extern __shared__ unsigned char dyn_lds[];
  float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  if (iters < 0) dyn_lds[threadIdx.x] = 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define FMA4 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
  for (int it = 0; it < iters; ++it)
    asm volatile(FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 FMA4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

static int run_p(void (*kernel)(float*, unsigned long long*, int), int waves_per_simd, int iters, float* out,
                 unsigned long long* stamps, double* cyc, double* ghz) {
  const int blocks = 256 * waves_per_simd;
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                         (int)lds));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, stamps, iters);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h((size_t)blocks * 16);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double sc = 0, sr = 0;
  for (int w = 0; w < blocks * 4; ++w) {
    sc += (double)(h[4 * w + 1] - h[4 * w]);
    sr += (double)(h[4 * w + 3] - h[4 * w + 2]);
  }
  *cyc = sc / (blocks * 4) / ((double)iters * 32) / waves_per_simd;
  *ghz = sc / sr * 0.1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// part S
// ---------------------------------------------------------------------------------------------
__global__ void k_sem(const float* in, float* out, int n) {
  // in: triples (d2, scale, offset) per thread; out: 4 floats per thread
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const v2f d2 = {in[4 * t + 0], in[4 * t + 1]};
  const v2f sc = {in[4 * t + 2], in[4 * t + 2]};
  const v2f of = {in[4 * t + 3], -1.0f};
  v2f m, b;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(d2), "v"(sc), "v"(of));
  // broadcast of the LOW half of the third source to both results
  asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel_hi:[1,1,0]" : "=v"(b) : "v"(d2), "v"(of));
  out[4 * t + 0] = m.x;
  out[4 * t + 1] = m.y;
  out[4 * t + 2] = b.x;
  out[4 * t + 3] = b.y;
}

// ---------------------------------------------------------------------------------------------
// part L: the two loop bodies
// ---------------------------------------------------------------------------------------------
struct LoopConsts {
  float thi, tlo_below, big;  // threshold, the float below the band's lower end, 2 / ulp
};

template <int BODY>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_loop(float* out, unsigned long long* stamps, int len, int trips, LoopConsts lc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* s_rec = reinterpret_cast<uint4*>(smem);
  // points scattered over a few cells around the lanes' cells: about 3/4 of the candidates hit
  for (int k = threadIdx.x; k < 1024; k += 512)
    s_rec[k] = make_uint4(((k * 2654435761u) >> 3) + (29u << 28), ((k * 40503u * 977u) >> 3) + (2u << 28),
                          __float_as_uint(0.25f * (k & 15) - 2.0f), 0u);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned Ui = (unsigned)(28 + (lane & 3)) << 28, Vj = (unsigned)(1 + ((threadIdx.x >> 6) & 3)) << 28;
  const float one_cell = (float)(1u << 28);
  const float thi = lc.thi, thiB = thi;
  float NA = 0, DA = 0, NB = 0, DB = 0, mA = 0, mB = 0;
  v2f AMB = {0.f, 0.f};
  const v2f negbig = {-lc.big, -lc.big}, posbig = {lc.big, lc.big};
  const v2f thibig = {lc.big * thi, lc.big * thiB};
  const v2f tlobig = {-lc.big * lc.tlo_below, -lc.big * lc.tlo_below};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
    float nA = 0, dA = 0, nB = 0, dB = 0;
    v2f Dp = {0.f, 0.f}, Np = {0.f, 0.f};
    const uint4* pr = s_rec + ((lane * 3 + t * 7) & 511);
    const uint4* const pe = pr + len;
    for (; pr < pe; ++pr) {
      const uint4 rec = *pr;
      if (BODY == 0) {
        float t0_, t1_, t3_;
        unsigned long long sv;
        asm volatile(
            "v_sub_u32 %[t0], %[Ui], %[x]\n\t"
            "v_sub_u32 %[t1], %[Vj], %[y]\n\t"
            "v_cvt_f32_i32 %[t0], %[t0]\n\t"
            "v_cvt_f32_i32 %[t1], %[t1]\n\t"
            "v_mul_f32 %[t0], %[t0], %[t0]\n\t"
            "v_add_f32 %[t3], %[one], %[t1]\n\t"
            "v_fma_f32 %[t1], %[t1], %[t1], %[t0]\n\t"
            "v_fma_f32 %[t3], %[t3], %[t3], %[t0]\n\t"
            "s_mov_b64 %[sv], exec\n\t"
            "v_cmpx_gt_f32 %[thi], %[t1]\n\t"
            "v_rcp_f32 %[t0], %[t1]\n\t"
            "v_max_f32 %[mA], %[mA], %[t1]\n\t"
            "v_add_f32 %[dA], %[dA], %[t0]\n\t"
            "v_fmac_f32 %[nA], %[t0], %[z]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_cmpx_gt_f32 %[thiB], %[t3]\n\t"
            "v_rcp_f32 %[t0], %[t3]\n\t"
            "v_max_f32 %[mB], %[mB], %[t3]\n\t"
            "v_add_f32 %[dB], %[dB], %[t0]\n\t"
            "v_fmac_f32 %[nB], %[t0], %[z]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [t0] "=&v"(t0_), [t1] "=&v"(t1_), [t3] "=&v"(t3_), [sv] "=&s"(sv), [mA] "+v"(mA),
              [mB] "+v"(mB), [nA] "+v"(nA), [dA] "+v"(dA), [nB] "+v"(nB), [dB] "+v"(dB)
            : [Ui] "v"(Ui), [Vj] "v"(Vj), [x] "v"(rec.x), [y] "v"(rec.y), [z] "v"(rec.z),
              [one] "v"(one_cell), [thi] "v"(thi), [thiB] "v"(thiB)
            : "vcc");
      } else {
        // packed body: the pairs live in 64-bit registers, the scalar halves are plain C++ (the
        // register coalescer writes them in place); modifiers need the asm
        const float dx = (float)(int)(Ui - rec.x);
        const float dy = (float)(int)(Vj - rec.y);
        v2f X2, Y, d2, m, a, w, Z;
        X2.x = dx * dx;
        X2.y = 0.f;
        Y.x = dy;
        Y.y = dy + one_cell;
        Z.x = __uint_as_float(rec.z);
        Z.y = 0.f;
        asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel_hi:[1,1,0]" : "=v"(d2) : "v"(Y), "v"(X2));
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(d2), "v"(negbig), "v"(thibig));
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(a) : "v"(d2), "v"(posbig), "v"(tlobig));
        w.x = __builtin_amdgcn_rcpf(d2.x);
        w.y = __builtin_amdgcn_rcpf(d2.y);
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(w) : "v"(w), "v"(m));
        asm("v_pk_add_f32 %0, %0, %1" : "+v"(Dp) : "v"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(Np) : "v"(w), "v"(Z));
        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(AMB) : "v"(a), "v"(m));
      }
    }
    if (BODY == 0) {
      NA += nA; DA += dA; NB += nB; DB += dB;
    } else {
      NA += Np.x; DA += Dp.x; NB += Np.y; DB += Dp.y;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    unsigned long long* s = stamps + 4 * ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6));
    s[0] = t0; s[1] = t1; s[2] = r0; s[3] = r1;
  }
  float* o = out + 6 * ((size_t)blockIdx.x * 512 + threadIdx.x);
  o[0] = NA; o[1] = DA; o[2] = NB; o[3] = DB;
  // ambiguity: the old body's max d2 of the hits against the band, the new body's count
  o[4] = BODY == 0 ? (float)(mA > lc.tlo_below) : (float)(AMB.x > 0.f);
  o[5] = BODY == 0 ? (float)(mB > lc.tlo_below) : (float)(AMB.y > 0.f);
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "PSL";
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* out;
  unsigned long long* stamps;
  CK(hipMalloc(&out, (size_t)1024 * 512 * 6 * 4 * 2));
  CK(hipMalloc(&stamps, (size_t)256 * 8 * 16 * 8 * 8));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, prop.multiProcessorCount);

  if (strchr(which, 'P')) {
    struct T {
      const char* name;
      void (*fn)(float*, unsigned long long*, int);
    } tests[] = {{"v_fma_f32 (yardstick)", k_fma32}, {"v_pk_fma_f32", k_pkfma},
                 {"v_pk_mul_f32", k_pkmul},          {"v_pk_add_f32", k_pkadd},
                 {"v_pk_fma_f32 clamp", k_pkfmac},   {"v_pk_fma_f32 op_sel_hi:[1,1,0]", k_pkfmab}};
    for (auto& t : tests) {
      printf("{\"part\": \"P\", \"instr\": \"%s\"", t.name);
      for (int w : {1, 2, 4, 8}) {
        double cyc, ghz;
        if (run_p(t.fn, w, 2048, out, stamps, &cyc, &ghz)) return 1;
        printf(", \"w%d\": {\"cyc\": %.3f, \"GHz\": %.3f}", w, cyc, ghz);
      }
      printf("}\n");
    }
  }

  if (strchr(which, 'S')) {
    // thresholds across binades; d2 = the neighbours of the threshold and far values
    std::vector<float> in;
    std::vector<int> expect;  // expected m.x
    const float ths[] = {1.0f, 3.0f, 4.6116860e18f * 1.000002f, 1.5e19f, 7.3e17f, 16777216.0f * 16777216.0f * 64.0f};
    for (float th : ths) {
      int e;
      std::frexp(th, &e);
      const float ulp = std::ldexp(1.0f, e - 24);
      const float big = 2.0f / ulp;
      const float cand[] = {std::nextafterf(th, 0.f), th, std::nextafterf(th, 1e38f),
                            std::nextafterf(std::nextafterf(th, 0.f), 0.f), th * 0.5f, th * 2.0f, 0.0f, th * 1e-6f,
                            std::ldexp(1.0f, 63)};
      for (float d : cand) {
        in.push_back(d);
        in.push_back(std::nextafterf(d, 1e38f));
        in.push_back(-big);
        in.push_back(big * th);
        expect.push_back(d < th ? 1 : 0);
      }
    }
    const int n = (int)expect.size();
    float *din, *dout;
    CK(hipMalloc(&din, in.size() * 4));
    CK(hipMalloc(&dout, in.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sem, dim3((n + 63) / 64), dim3(64), 0, 0, din, dout, n);
    std::vector<float> h(in.size());
    CK(hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost));
    int bad_cmp = 0, bad_hi = 0, bad_bcast = 0, frac = 0;
    for (int t = 0; t < n; ++t) {
      const float d0 = in[4 * t], d1 = in[4 * t + 1], of = in[4 * t + 3];
      if (h[4 * t] != (float)expect[t]) ++bad_cmp;
      if (h[4 * t] != 0.f && h[4 * t] != 1.f) ++frac;
      // high half: offset -1, scale -big: never a hit; clamp must give 0 (not a negative number)
      if (h[4 * t + 1] != 0.f) ++bad_hi;
      const float b0 = std::fmaf(d0, d0, of), b1 = std::fmaf(d1, d1, of);
      if (!(h[4 * t + 2] == b0 || (std::isinf(b0) && std::isinf(h[4 * t + 2])))) ++bad_bcast;
      if (!(h[4 * t + 3] == b1 || (std::isinf(b1) && std::isinf(h[4 * t + 3])))) ++bad_bcast;
    }
    printf("{\"part\": \"S\", \"cases\": %d, \"clamped_fma_differs_from_comparison\": %d, \"fractional_masks\": %d, "
           "\"high_half_not_clamped\": %d, \"op_sel_hi_broadcast_wrong\": %d}\n",
           n, bad_cmp, frac, bad_hi, bad_bcast);
  }

  if (strchr(which, 'L')) {
    LoopConsts lc;
    const float one_cell = (float)(1u << 28);
    lc.thi = 16.0f * one_cell * one_cell * 1.000002f;
    const float tlo = 16.0f * one_cell * one_cell * 0.999998f;
    lc.tlo_below = std::nextafterf(tlo, 0.f);
    int e;
    std::frexp(lc.tlo_below, &e);
    lc.big = 2.0f / std::ldexp(1.0f, e - 24);
    for (int len : {9, 13, 45}) {
      const int blocks = 256 * 4, trips = 45 * 40 / len;
      const size_t lds = 37 * 1024;
      std::vector<float> res[2];
      for (int body = 0; body < 2; ++body) {
        float ms = 0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        float* o = out + (size_t)body * blocks * 512 * 6;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          if (body == 0) hipLaunchKernelGGL(k_loop<0>, dim3(blocks), dim3(512), lds, 0, o, stamps, len, trips, lc);
          else hipLaunchKernelGGL(k_loop<1>, dim3(blocks), dim3(512), lds, 0, o, stamps, len, trips, lc);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> h((size_t)blocks * 8 * 4);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double sc = 0, sr = 0;
        for (int w = 0; w < blocks * 8; ++w) {
          sc += (double)(h[4 * w + 1] - h[4 * w]);
          sr += (double)(h[4 * w + 3] - h[4 * w + 2]);
        }
        const double cands = (double)trips * len;
        res[body].resize((size_t)blocks * 512 * 6);
        CK(hipMemcpy(res[body].data(), o, res[body].size() * 4, hipMemcpyDeviceToHost));
        double hits = 0;
        printf("{\"part\": \"L\", \"body\": \"%s\", \"trip_len\": %d, \"cycles_per_candidate_per_wave\": %.2f, "
               "\"cycles_per_candidate_per_simd\": %.2f, \"GHz\": %.3f, \"ms\": %.3f",
               body ? "packed" : "cmpx (round 2)", len, sc / (blocks * 8) / cands, sc / (blocks * 8) / cands / 8.0,
               sc / sr * 0.1, ms);
        (void)hits;
        if (body == 1) {
          size_t diff = 0, amb = 0, nonzero = 0;
          for (size_t k = 0; k < res[0].size(); ++k) {
            if (memcmp(&res[0][k], &res[1][k], 4)) ++diff;
            if (k % 6 >= 4 && res[1][k] != 0.f) ++amb;
            if (k % 6 == 1 && res[1][k] != 0.f) ++nonzero;
          }
          printf(", \"words_differing_from_round2_body\": %zu, \"cells_with_hits\": %zu, \"ambiguous_cells\": %zu",
                 diff, nonzero, amb);
        }
        printf("}\n");
      }
    }
  }
  return 0;
}
