// a1/a2: one query against n candidate vectors, fp64 on the device.
// Replaces DenseReranker._cosine x N (app/modules/retrieval/retrieval_backend.py:192-197,245):
// out[i] = dot / (|q| |c_i|), 0.0 when either norm is zero.  One wave per candidate,
// coalesced fp64 loads, shuffle reduction; HBM-bound: reads (n+1)*dim*8 bytes.
#include "common.h"

namespace mrag {

__global__ __launch_bounds__(256) void cosine_f64_kernel(const double* __restrict__ q, const double* __restrict__ c,
                                                         int64_t n, int dim, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* v = c + row * (int64_t)dim;
  double dot = 0.0, qq = 0.0, vv = 0.0;
  for (int i = lane; i < dim; i += 64) {
    const double a = q[i], b = v[i];
    dot += a * b;
    qq += a * a;
    vv += b * b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dot += __shfl_xor(dot, off);
    qq += __shfl_xor(qq, off);
    vv += __shfl_xor(vv, off);
  }
  if (lane == 0) {
    const double na = sqrt(qq), nb = sqrt(vv);
    out[row] = (na != 0.0 && nb != 0.0) ? dot / (na * nb) : 0.0;
  }
}


// all pairs: one wave per (i, j >= i) pair block row; n is small (graph sentences, MMR pools)
__global__ __launch_bounds__(256) void cosine_matrix_f64_kernel(const double* __restrict__ x, int64_t n, int dim,
                                                                double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n * n) return;
  const int64_t i = pair / n, j = pair % n;
  if (j < i) return;
  const double *a = x + i * (int64_t)dim, *b = x + j * (int64_t)dim;
  double dot = 0.0, aa = 0.0, bb = 0.0;
  for (int t = lane; t < dim; t += 64) {
    const double u = a[t], v = b[t];
    dot += u * v; aa += u * u; bb += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dot += __shfl_xor(dot, off); aa += __shfl_xor(aa, off); bb += __shfl_xor(bb, off);
  }
  if (lane == 0) {
    const double na = sqrt(aa), nb = sqrt(bb);
    const double c = (na != 0.0 && nb != 0.0) ? dot / (na * nb) : 0.0;
    out[i * n + j] = c;
    out[j * n + i] = c;
  }
}

// adjacent rows: out[i] = dot(x_i, x_{i+1}) / (|x_i| |x_{i+1}| + eps) -- the cut rule of embed-mode segmentation
// (app/modules/graph_construction/segmenter.py:40-42: eps = 1e-9 inside the denominator, no zero-norm branch)
__global__ __launch_bounds__(256) void cosine_adjacent_f64_kernel(const double* __restrict__ x, int64_t n, int dim, double eps,
                                                                  double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i + 1 >= n) return;
  const double *a = x + i * (int64_t)dim, *b = a + dim;
  double dot = 0.0, aa = 0.0, bb = 0.0;
  for (int t = lane; t < dim; t += 64) {
    const double u = a[t], v = b[t];
    dot += u * v; aa += u * u; bb += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dot += __shfl_xor(dot, off); aa += __shfl_xor(aa, off); bb += __shfl_xor(bb, off);
  }
  if (lane == 0) out[i] = dot / (sqrt(aa) * sqrt(bb) + eps);
}

}  // namespace mrag

using namespace mrag;

extern "C" int mrag_cosine_adjacent_f64(int device, const double* x, int64_t n, int dim, double eps, double* out, void* stream_) {
  if (n < 0 || dim <= 0) return fail(MRAG_ERR_INVALID, "bad shape n=%lld dim=%d", (long long)n, dim);
  if (n < 2) return MRAG_OK;
  if (!x || !out) return fail(MRAG_ERR_INVALID, "NULL buffer");
  MRAG_TRY(use_device(device));
  hipStream_t stream = (hipStream_t)stream_;
  void* tmp = nullptr;
  const size_t xb = (size_t)n * dim * 8, ob = (size_t)(n - 1) * 8;
  MRAG_HIP(hipMalloc(&tmp, xb + ob));
  hipError_t e = hipMemcpyAsync(tmp, x, xb, hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cosine_adjacent_f64_kernel, dim3((unsigned)((n + 2) / 4)), dim3(256), 0, stream, (const double*)tmp, n, dim, eps,
                       (double*)((char*)tmp + xb));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, (char*)tmp + xb, ob, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(MRAG_ERR_HIP, "cosine_adjacent_f64 failed: %s", hipGetErrorString(e));
  return MRAG_OK;
}

extern "C" int mrag_cosine_matrix_f64(int device, const double* x, int64_t n, int dim, double* out, int is_device, void* stream_) {
  if (n < 0 || dim <= 0 || n > 32768) return fail(MRAG_ERR_INVALID, "bad shape n=%lld dim=%d (n <= 32768)", (long long)n, dim);
  if (n == 0) return MRAG_OK;
  if (!x || !out) return fail(MRAG_ERR_INVALID, "NULL buffer");
  MRAG_TRY(use_device(device));
  hipStream_t stream = (hipStream_t)stream_;
  const double* dx = x;
  double* dout = out;
  void* tmp = nullptr;
  const size_t xb = (size_t)n * dim * 8, ob = (size_t)n * n * 8;
  if (!is_device) {
    MRAG_HIP(hipMalloc(&tmp, xb + ob));
    if (hipMemcpyAsync(tmp, x, xb, hipMemcpyHostToDevice, stream) != hipSuccess) { (void)hipFree(tmp); return fail(MRAG_ERR_HIP, "H2D copy failed"); }
    dx = (const double*)tmp;
    dout = (double*)((char*)tmp + xb);
  }
  hipLaunchKernelGGL(cosine_matrix_f64_kernel, dim3((unsigned)((n * n + 3) / 4)), dim3(256), 0, stream, dx, n, dim, dout);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !is_device) {
    e = hipMemcpyAsync(out, dout, ob, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  if (tmp) (void)hipFree(tmp);
  if (e != hipSuccess) return fail(MRAG_ERR_HIP, "cosine_matrix_f64 failed: %s", hipGetErrorString(e));
  return MRAG_OK;
}


extern "C" int mrag_cosine_f64(int device, const double* query, const double* cands, int64_t n, int dim,
                               double* out_scores, int is_device, void* stream_) {
  if (n < 0 || dim <= 0) return fail(MRAG_ERR_INVALID, "bad shape n=%lld dim=%d", (long long)n, dim);
  if (n == 0) return MRAG_OK;
  if (!query || !cands || !out_scores) return fail(MRAG_ERR_INVALID, "NULL buffer");
  MRAG_TRY(use_device(device));
  hipStream_t stream = (hipStream_t)stream_;
  const double *dq = query, *dc = cands;
  double* dout = out_scores;
  void* tmp = nullptr;
  if (!is_device) {
    const size_t qb = (size_t)dim * 8, cb = (size_t)n * dim * 8, ob = (size_t)n * 8;
    MRAG_HIP(hipMalloc(&tmp, qb + cb + ob));
    char* base = (char*)tmp;
    hipError_t e1 = hipMemcpyAsync(base, query, qb, hipMemcpyHostToDevice, stream);
    hipError_t e2 = hipMemcpyAsync(base + qb, cands, cb, hipMemcpyHostToDevice, stream);
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(tmp); return fail(MRAG_ERR_HIP, "H2D copy failed"); }
    dq = (const double*)base;
    dc = (const double*)(base + qb);
    dout = (double*)(base + qb + cb);
  }
  hipLaunchKernelGGL(cosine_f64_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, dq, dc, n, dim, dout);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !is_device) {
    e = hipMemcpyAsync(out_scores, dout, (size_t)n * 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  if (tmp) (void)hipFree(tmp);
  if (e != hipSuccess) return fail(MRAG_ERR_HIP, "cosine_f64 failed: %s", hipGetErrorString(e));
  return MRAG_OK;
}
