// refign_amd/csrc/gemm.hip -- library GEMMs of the MiT Linear layers through a plan cache.
//
// The dense GEMMs of the step stay on the ROCm library (hipBLASLt; "plain library GEMMs"): what is hand-made here is the
// CALL PATH.  One Refign step issues ~2 100 of them from the student's forward/backward, and through the framework each
// costs ~35 us of host time (descriptor and layout objects built per call, a heuristic query per call, the extended
// launch path) -- 75 ms of a ~290 ms, host-bound step (profiles/r01_step_profile_host_time.txt).  The same few dozen
// problems repeat every step, so the descriptors, layouts and the heuristic's algorithm are made once per problem and
// kept; a call is then a map lookup, one attribute update (bias pointer) and hipblasLtMatmul.
//
// Row-major operands are handed to the column-major library the usual way (C^T = B^T A^T):
//   kind 0  forward   Y[T,N]  = X[T,K] . W[N,K]^T (+ bias[N])     -> m=N, n=T, k=K, opA=T (W), opB=N (X)
//   kind 1  dgrad     dX[T,K] = dY[T,N] . W[N,K]                  -> m=K, n=T, k=N, opA=N (W), opB=N (dY)
//   kind 2  wgrad     P[s][N,K] = dY_s[Ts,N]^T . X_s[Ts,K], s < S -> m=K, n=N, k=Ts, opA=N (X_s), opB=T (dY_s), batched
// dtype 1 = bfloat16 in/out, 0 = float32; fp32 accumulation.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <tuple>

#include "common.h"

namespace rfn {

struct GemmPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};

using GemmKey = std::tuple<int, long, long, long, int, int, int>;   // kind, T, N, K, S, dtype, has_bias

struct GemmCtx {
  hipblasLtHandle_t handle = nullptr;
  std::map<GemmKey, GemmPlan> plans;
};

// one context per host thread (forward runs on the Python thread, backward on the autograd engine's thread): no locks
static GemmCtx& gemm_ctx() {
  static thread_local GemmCtx ctx;
  return ctx;
}

constexpr size_t kGemmWorkspace = 32u << 20;

static int make_plan(GemmCtx& ctx, GemmPlan& p, int kind, long T, long N, long K, int S, int dtype, int has_bias) {
  const hipDataType dt = dtype == 1 ? HIP_R_16BF : HIP_R_32F;
  long m, n, lda, ldb, ldc, sa = 0, sb = 0, sc = 0;
  hipblasOperation_t opa, opb;
  long rows_a, cols_a, rows_b, cols_b;
  if (kind == 0) {            // A_blas = W (col-major K x N, ld K) transposed; B_blas = X (col-major K x T, ld K)
    m = N; n = T; opa = HIPBLAS_OP_T; opb = HIPBLAS_OP_N;
    rows_a = K; cols_a = N; lda = K; rows_b = K; cols_b = T; ldb = K; ldc = N;
  } else if (kind == 1) {     // A_blas = W (K x N, ld K); B_blas = dY (N x T, ld N)
    m = K; n = T; opa = HIPBLAS_OP_N; opb = HIPBLAS_OP_N;
    rows_a = K; cols_a = N; lda = K; rows_b = N; cols_b = T; ldb = N; ldc = K;
  } else {                    // per slab: A_blas = X_s (K x Ts, ld K); B_blas = dY_s (N x Ts, ld N) transposed
    const long Ts = T / S;
    m = K; n = N; opa = HIPBLAS_OP_N; opb = HIPBLAS_OP_T;
    rows_a = K; cols_a = Ts; lda = K; rows_b = N; cols_b = Ts; ldb = N; ldc = K;
    sa = Ts * K; sb = Ts * N; sc = N * K;
  }
  if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return 1;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb));
  if (has_bias) {
    hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
    int32_t bdt = (int32_t)dt;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bdt, sizeof(bdt));
  }
  if (hipblasLtMatrixLayoutCreate(&p.la, dt, rows_a, cols_a, lda) != HIPBLAS_STATUS_SUCCESS) return 2;
  if (hipblasLtMatrixLayoutCreate(&p.lb, dt, rows_b, cols_b, ldb) != HIPBLAS_STATUS_SUCCESS) return 2;
  if (hipblasLtMatrixLayoutCreate(&p.lc, dt, m, n, ldc) != HIPBLAS_STATUS_SUCCESS) return 2;
  if (kind == 2 && S > 1) {
    int32_t bc = S;
    int64_t a64 = sa, b64 = sb, c64 = sc;
    hipblasLtMatrixLayoutSetAttribute(p.la, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
    hipblasLtMatrixLayoutSetAttribute(p.lb, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
    hipblasLtMatrixLayoutSetAttribute(p.lc, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc));
    hipblasLtMatrixLayoutSetAttribute(p.la, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &a64, sizeof(a64));
    hipblasLtMatrixLayoutSetAttribute(p.lb, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &b64, sizeof(b64));
    hipblasLtMatrixLayoutSetAttribute(p.lc, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &c64, sizeof(c64));
  }
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return 3;
  uint64_t wsmax = kGemmWorkspace;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax));
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  const hipblasStatus_t st =
      hipblasLtMatmulAlgoGetHeuristic(ctx.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return 4;
  p.algo = res[0].algo;
  p.ws = res[0].workspaceSize;
  p.ok = true;
  return 0;
}

}  // namespace rfn

using namespace rfn;

extern "C" {

unsigned long rfn_gemm_workspace_bytes(void) { return (unsigned long)kGemmWorkspace; }

// A: weight W (kind 0, 1) or activations X (kind 2); B: activations X (kind 0) or grad_y (kind 1, 2); C: result.
int rfn_linear_gemm(int kind, const void* A, const void* B, void* C, const void* bias, void* workspace, long T, long N,
                    long K, int S, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(A && B && C && workspace, "rfn_linear_gemm: null pointer");
  RFN_REQUIRE(kind >= 0 && kind <= 2 && T > 0 && N > 0 && K > 0 && S >= 1 && (dtype == 0 || dtype == 1),
              "rfn_linear_gemm: bad arguments");
  RFN_REQUIRE(kind == 2 ? (T % S == 0) : (S == 1), "rfn_linear_gemm: S must divide T (kind 2) / be 1 (kinds 0, 1)");
  RFN_REQUIRE(bias == nullptr || kind == 0, "rfn_linear_gemm: bias only with kind 0");
  GemmCtx& ctx = gemm_ctx();
  if (ctx.handle == nullptr && hipblasLtCreate(&ctx.handle) != HIPBLAS_STATUS_SUCCESS)
    return fail(RFN_ELAUNCH, "rfn_linear_gemm: hipblasLtCreate failed");
  const GemmKey key{kind, T, N, K, S, dtype, bias != nullptr ? 1 : 0};
  auto it = ctx.plans.find(key);
  if (it == ctx.plans.end()) {
    GemmPlan p;
    const int rc = make_plan(ctx, p, kind, T, N, K, S, dtype, bias != nullptr ? 1 : 0);
    if (rc != 0) return fail(RFN_ELAUNCH, "rfn_linear_gemm: no hipBLASLt plan (stage %d) for kind %d T=%ld N=%ld K=%ld S=%d",
                             rc, kind, T, N, K, S);
    it = ctx.plans.emplace(key, p).first;
  }
  GemmPlan& p = it->second;
  if (bias != nullptr)
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  const float alpha = 1.0f, beta = 0.0f;
  const hipblasStatus_t st = hipblasLtMatmul(ctx.handle, p.desc, &alpha, A, p.la, B, p.lb, &beta, C, p.lc, C, p.lc, &p.algo,
                                             workspace, kGemmWorkspace, (hipStream_t)stream);
  if (st != HIPBLAS_STATUS_SUCCESS) return fail(RFN_ELAUNCH, "rfn_linear_gemm: hipblasLtMatmul status %d", (int)st);
  return RFN_OK;
}

}  // extern "C"
