// tools/micro/gemm2_probe.hip -- stand-alone A/B harness for the second-generation NT GEMM (refign_amd/csrc/gemm2.h):
// every variant against the shipped kernel (rfn_gemm_nt of librefign_hip.so, loaded with dlopen) on the teacher / student
// shapes of the HRDA step -- bit-for-bit comparison of the whole result (same MFMA, same k order, same rounding point),
// an fp64 host check of sampled entries, HIP-event timing over rotating operand sets (so that the operands come from
// HBM, as they do in the step).  Not part of the product: builds to gpurun_out/, run by tools/gemm2_probe.sh.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm2_probe.hip -o gpurun_out/gemm2_probe -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../refign_amd/csrc/gemm2.h"

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

using namespace rfn;

typedef int (*gemm_nt_fn)(const void*, const void*, const void*, const void*, const float*, int, int, void*, long, long,
                          long, long, long, long, int, void*);

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct Variant {
  const char* name;
  int bm, bn, nsk;
  void (*launch)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long, long, long, Gemm2Epi, bool, hipStream_t);
  bool force_res_kernel = false;
};

template <int BM, int BN, int NSK, int D0, int D1, int D3, int ABL = 0>
static void launch_v(const uint16_t* X, const uint16_t* W, uint16_t* Y, int M, int N, int K, long ldx, long ldw, long ldy,
                     Gemm2Epi epi, bool res, hipStream_t s) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
  const int total = tiles_m * tiles_n;
  static const int cap = getenv("G2_GRID") ? atoi(getenv("G2_GRID")) : 256;
  dim3 grid(total < cap ? total : cap), block(256);
  if (res)
    hipLaunchKernelGGL((gemm_nt2_kernel<1, BM, BN, 2, 2, true, true, 0, NSK, D0, D1, D3, ABL>), grid, block, 0, s, X, W, Y,
                       M, N, K, ldx, ldw, ldy, tiles_n, total, epi);
  else
    hipLaunchKernelGGL((gemm_nt2_kernel<1, BM, BN, 2, 2, true, false, 0, NSK, D0, D1, D3, ABL>), grid, block, 0, s, X, W, Y,
                       M, N, K, ldx, ldw, ldy, tiles_n, total, epi);
}

static const Variant kVariants[] = {
    {"192x320 nsk3 d664", 192, 320, 3, launch_v<192, 320, 3, 6, 6, 4>},
    {"192x320 nsk3 d556", 192, 320, 3, launch_v<192, 320, 3, 5, 5, 6>},
    {"192x320 nsk1 d664", 192, 320, 1, launch_v<192, 320, 1, 6, 6, 4>},
    {"192x320 nsk3 d448", 192, 320, 3, launch_v<192, 320, 3, 4, 4, 8>},
    {"192x320 d556 dbg8", 192, 320, 3, launch_v<192, 320, 3, 5, 5, 6, 8>},
    {"192x320 RESkernel/nullres", 192, 320, 3, launch_v<192, 320, 3, 5, 5, 6>, true},
};

struct Shape {
  int M, K, N;
  bool res;
};

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "refign_amd/lib/librefign_hip.so";
  void* h = dlopen(libpath, RTLD_NOW);
  if (!h) {
    fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror());
    return 1;
  }
  gemm_nt_fn old_gemm = (gemm_nt_fn)dlsym(h, "rfn_gemm_nt");
  if (!old_gemm) return 1;
  const Shape shapes[] = {
      {81600, 320, 320, true}, {81600, 320, 320, false}, {81600, 1280, 320, true}, {81000, 512, 1280, true},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  const int reps = 20;
  for (const Shape& sh : shapes) {
    const long M = sh.M, K = sh.K, N = sh.N;
    const size_t xb = (size_t)M * K * 2, wb = (size_t)N * K * 2, yb = (size_t)M * N * 2;
    int R = (int)((768ul << 20) / (xb + yb + (sh.res ? yb : 0)) + 1);   // rotating sets: ~768 MB of operands in all
    if (R > 8) R = 8;
    if (R < 2) R = 2;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hb(N), hr;
    for (auto& v : hx) v = f2bf(nd(rng));
    for (auto& v : hw) v = f2bf(nd(rng) / sqrtf((float)K));
    for (auto& v : hb) v = f2bf(nd(rng));
    std::vector<float> hrs;
    const int rps = 2040;
    if (sh.res) {
      hr.resize((size_t)M * N);
      for (auto& v : hr) v = f2bf(nd(rng));
      hrs.resize((M + rps - 1) / rps);
      for (auto& v : hrs) v = 0.5f + 0.1f * nd(rng);
    }
    std::vector<uint16_t*> dx(R), dy(R), dr(R, nullptr);
    uint16_t *dw, *db, *dyref;
    float* drs = nullptr;
    for (int r = 0; r < R; ++r) {
      CK(hipMalloc(&dx[r], xb));
      CK(hipMalloc(&dy[r], yb));
      CK(hipMemcpy(dx[r], hx.data(), xb, hipMemcpyHostToDevice));
      if (sh.res) {
        CK(hipMalloc(&dr[r], yb));
        CK(hipMemcpy(dr[r], hr.data(), yb, hipMemcpyHostToDevice));
      }
    }
    CK(hipMalloc(&dw, wb));
    CK(hipMalloc(&db, N * 2));
    CK(hipMalloc(&dyref, yb));
    CK(hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 2, hipMemcpyHostToDevice));
    if (sh.res) {
      CK(hipMalloc(&drs, hrs.size() * 4));
      CK(hipMemcpy(drs, hrs.data(), hrs.size() * 4, hipMemcpyHostToDevice));
    }
    const double flop = 2.0 * M * N * K;
    // ---- shipped kernel
    auto run_old = [&](int r, uint16_t* y) {
      int rc = old_gemm(dx[r], dw, db, sh.res ? dr[r] : nullptr, drs, rps, 0, y, M, N, K, K, K, N, 1, st);
      if (rc != 0) {
        fprintf(stderr, "old kernel rc %d\n", rc);
        exit(1);
      }
    };
    run_old(0, dyref);
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < 3; ++i) run_old(i % R, dy[i % R]);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run_old(i % R, dy[i % R]);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_old = ms * 1e3 / reps;
    std::vector<uint16_t> yref((size_t)M * N), yv((size_t)M * N);
    CK(hipMemcpy(yref.data(), dyref, yb, hipMemcpyDeviceToHost));
    // fp64 host check of the shipped kernel on sampled entries
    double worst = 0;
    for (int t = 0; t < 2000; ++t) {
      const long m = (t < 64) ? (M - 1 - t) : (long)(rng() % M), n = rng() % N;
      double acc = 0;
      for (long k = 0; k < K; ++k) acc += (double)bf2f(hx[m * K + k]) * (double)bf2f(hw[n * K + k]);
      acc += bf2f(hb[n]);
      if (sh.res) acc = bf2f(hr[m * N + n]) + hrs[m / rps] * acc;
      const double got = bf2f(yref[m * N + n]);
      const double err = fabs(got - acc) / (fabs(acc) + 1.0);
      if (err > worst) worst = err;
    }
    printf("== %ld x %ld -> %ld%s  (R=%d)  shipped: %8.1f us %7.1f TF/s   fp64 sample rel err %.2e\n", M, K, N,
           sh.res ? " +res" : "", R, us_old, flop / us_old / 1e6, worst);
    fflush(stdout);
    // ---- variants
    for (const Variant& v : kVariants) {
      if (N % v.bn != 0 || K / 64 < v.nsk) continue;
      if ((M + v.bm - 1) / v.bm * (N / v.bn) < 64) continue;
      Gemm2Epi epi{db, sh.res ? dr[0] : nullptr, drs, rps, nullptr};
      CK(hipMemsetAsync(dy[0], 0xff, yb, st));
      if (v.force_res_kernel && sh.res) continue;
      v.launch(dx[0], dw, dy[0], (int)M, (int)N, (int)K, K, K, N, epi, sh.res || v.force_res_kernel, st);
      CK(hipStreamSynchronize(st));
      hipError_t le = hipGetLastError();
      if (le != hipSuccess) {
        printf("   %-20s launch error %s\n", v.name, hipGetErrorString(le));
        continue;
      }
      CK(hipMemcpy(yv.data(), dy[0], yb, hipMemcpyDeviceToHost));
      size_t nbad = 0;
      double maxd = 0;
      double worst_new = 0;
      {
        std::mt19937 r2(99);
        for (int t = 0; t < 4000; ++t) {
          const long m = (t < 2000) ? (long)((r2() % (M / 16)) * 16 + 12 + (t & 3)) : (long)(r2() % M), n = r2() % N;
          if (m >= M) continue;
          double acc = 0;
          for (long k = 0; k < K; ++k) acc += (double)bf2f(hx[m * K + k]) * (double)bf2f(hw[n * K + k]);
          acc += bf2f(hb[n]);
          if (sh.res) acc = bf2f(hr[m * N + n]) + hrs[m / rps] * acc;
          const double err = fabs((double)bf2f(yv[m * N + n]) - acc) / (fabs(acc) + 1.0);
          if (!(err <= worst_new)) worst_new = err;
        }
      }
      int shown = 8;
      std::vector<long> hm(192, 0), hn(320, 0);
      for (size_t i = 0; i < yv.size(); ++i)
        if (yv[i] != yref[i]) {
          ++nbad;
          if (!(fabs((double)bf2f(yv[i]) - (double)bf2f(yref[i])) < 1.0)) { hm[(i / N) % 192]++; hn[(i % N) % 320]++; }
          if (shown < 12 && !(fabs((double)bf2f(yv[i]) - (double)bf2f(yref[i])) < 1.0)) {
            printf("      bad m=%zu n=%zu got %04x (%g) ref %04x (%g)\n", i / N, i % N, yv[i], bf2f(yv[i]), yref[i], bf2f(yref[i]));
            ++shown;
          }
          const double d = fabs((double)bf2f(yv[i]) - (double)bf2f(yref[i]));
          if (d > maxd || std::isnan(d)) maxd = std::isnan(d) ? 1e30 : d;
        }
      {
        long tb = 0;
        for (int r = 0; r < 192; ++r) tb += hm[r];
        if (tb) printf("      %ld entries off by >= 1\n", tb);
      }
      for (int i = 0; i < 3; ++i) {
        epi.res = sh.res ? dr[i % R] : nullptr;
        v.launch(dx[i % R], dw, dy[i % R], (int)M, (int)N, (int)K, K, K, N, epi, sh.res, st);
      }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) {
        epi.res = sh.res ? dr[i % R] : nullptr;
        v.launch(dx[i % R], dw, dy[i % R], (int)M, (int)N, (int)K, K, K, N, epi, sh.res, st);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("   %-20s %8.1f us %7.1f TF/s  x%.2f   mismatching %zu of %zu (max |d| %.3g)  fp64 sample rel err %.2e\n", v.name, us, flop / us / 1e6,
             us_old / us, nbad, yv.size(), maxd, worst_new);
      fflush(stdout);
    }
    if (N % 320 == 0 && K >= 192) {   // s_memtime trace of workgroup 0, wave 0 (per K-step: top, g0, g1, g2, waited, barrier, reads, g3; + epilogue end)
      unsigned long long* dtr;
      CK(hipMalloc(&dtr, 512 * 8));
      CK(hipMemset(dtr, 0, 512 * 8));
      Gemm2Epi epi{db, nullptr, nullptr, rps, dtr};
      const int tiles_n = (int)N / 320, total = (int)((M + 191) / 192) * tiles_n;
      dim3 grid(total < 256 ? total : 256);
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL((gemm_nt2_kernel<1, 192, 320, 2, 2, true, false, 0, 3, 5, 5, 6, 0, true>), grid, dim3(256), 0, st, dx[0], dw,
                         dy[0], (int)M, (int)N, (int)K, K, K, N, tiles_n, total, epi);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> tr(512);
      CK(hipMemcpy(tr.data(), dtr, 512 * 8, hipMemcpyDeviceToHost));
      int n = 0;
      while (n < 512 && tr[n] != 0) ++n;
      printf("   trace 192x320 nsk3: %d stamps, kernel %.1f us, first->last %llu ticks\n", n, ms * 1e3, n ? tr[n - 1] - tr[0] : 0ull);
      const int nk = (int)K / 64, per = 8 * nk + 1;
      for (int t = 0; t < 3 && (t + 1) * per <= n; ++t) {
        printf("     tile %d:", t);
        for (int k = 0; k < nk && k < 6; ++k) {
          const unsigned long long* q = &tr[t * per + 8 * k];
          printf(" [g0 %llu g1 %llu g2 %llu wait %llu bar %llu rd %llu g3 %llu]", q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3],
                 q[5] - q[4], q[6] - q[5], q[7] - q[6]);
        }
        printf(" epilogue %llu  tile total %llu\n", tr[t * per + per - 1] - tr[t * per + per - 2], tr[t * per + per - 1] - tr[t * per]);
      }
      CK(hipFree(dtr));
    }
    for (int r = 0; r < R; ++r) {
      CK(hipFree(dx[r]));
      CK(hipFree(dy[r]));
      if (dr[r]) CK(hipFree(dr[r]));
    }
    CK(hipFree(dw));
    CK(hipFree(db));
    CK(hipFree(dyref));
    if (drs) CK(hipFree(drs));
  }
  return 0;
}
