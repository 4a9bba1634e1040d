// Library plumbing: error strings, handle registry, device selection, row preparation (K1).
#include "common.h"
#include <type_traits>

#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>

namespace mrag {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MRAG_ERR_NO_DEVICE, "no HIP device available (%s); libmrag_hip has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (device < 0 || device >= n) return fail(MRAG_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
  MRAG_HIP(hipSetDevice(device));
  return MRAG_OK;
}

static std::mutex g_mu;
static std::unordered_map<uint64_t, Object*> g_objs;
static uint64_t g_next = 0x4d520001ull;

mrag_handle register_object(Object* o) {
  std::lock_guard<std::mutex> lk(g_mu);
  uint64_t h = g_next++;
  g_objs[h] = o;
  return h;
}

Object* lookup(mrag_handle h, HandleKind kind) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_objs.find(h);
  if (it == g_objs.end() || it->second->kind != kind) {
    set_error("invalid handle 0x%llx", (unsigned long long)h);
    return nullptr;
  }
  return it->second;
}

Object* take(mrag_handle h, HandleKind kind) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_objs.find(h);
  if (it == g_objs.end() || it->second->kind != kind) {
    set_error("invalid handle 0x%llx", (unsigned long long)h);
    return nullptr;
  }
  Object* o = it->second;
  g_objs.erase(it);
  return o;
}

int DevBuf::ensure(size_t need) {
  if (need <= bytes) return MRAG_OK;
  if (p) {
    (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  size_t want = need + need / 4;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    want = need;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    p = nullptr;
    return fail(MRAG_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  bytes = want;
  return MRAG_OK;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

// ------------------------------------------------------------------------------------------
// K1: row preparation.  One wave per row: fp64 sum of squares (squares of fp32 values are
// exact in fp64), fp64 divide, one rounding to fp32, one rounding to the storage type --
// the same arithmetic as oracle/dense_search.py:l2_normalize + astype.  HBM-bound:
// reads n*dim*sizeof(src), writes n*ld*2 bytes.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ inline double load_as_f64(const T* p, int64_t i);
template <>
__device__ inline double load_as_f64<float>(const float* p, int64_t i) { return (double)p[i]; }
template <>
__device__ inline double load_as_f64<_Float16>(const _Float16* p, int64_t i) { return (double)(float)p[i]; }
template <>
__device__ inline double load_as_f64<__bf16>(const __bf16* p, int64_t i) { return (double)(float)p[i]; }
template <>
__device__ inline double load_as_f64<double>(const double* p, int64_t i) { return p[i]; }

template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void prep_rows_kernel(const SRC* __restrict__ src, int64_t n, int dim,
                                                        DST* __restrict__ dst, int ld, int normalize) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const SRC* s = src + row * (int64_t)dim;
  DST* d = dst + row * (int64_t)ld;
  double inv = 1.0;
  if (normalize) {
    double ss = 0.0;
    for (int i = lane; i < dim; i += 64) {
      double x = load_as_f64<SRC>(s, i);
      ss += x * x;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    inv = ss > 0.0 ? sqrt(ss) : 1.0;
  }
  for (int i = lane; i < ld; i += 64) {
    float y = 0.f;
    if (i < dim) {
      double x = load_as_f64<SRC>(s, i);
      y = normalize ? (float)(x / inv) : (float)x;
    }
    d[i] = (DST)y;
  }
}

// fp32 sources with dim % 8 == 0 and ld <= 4096: one wave per row, the row is read ONCE as 2 x 16-byte
// loads per 8-element chunk (chunk c of lane l: c = l, l + 64, ...) and kept in registers for the fp64
// sum of squares and the division; stores are 16 bytes.  Same arithmetic as the scalar kernel (fp64
// square sum in the same per-lane order is NOT kept: the order of the partial sums differs, the fp64
// sum is then rounded once -- see the parity test on device normalisation).
template <typename DST>
__global__ __launch_bounds__(256) void prep_rows_vec_kernel(const float* __restrict__ src, int64_t n, int dim,
                                                            DST* __restrict__ dst, int ld, int normalize) {
  typedef DST d8 __attribute__((ext_vector_type(8)));
  constexpr int MAXC = 8;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float4* s = (const float4*)(src + row * (int64_t)dim);
  d8* d = (d8*)(dst + row * (int64_t)ld);
  const int nch = dim >> 3, nch_out = ld >> 3;
  float v[MAXC][8];
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const float4 a = s[2 * c], b = s[2 * c + 1];
      v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (double)v[i][j] * (double)v[i][j];
    }
  }
  double inv = 1.0;
  if (normalize) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    inv = ss > 0.0 ? sqrt(ss) : 1.0;
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch_out) {
      d8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = 0.f;
        if (c < nch) y = normalize ? (float)((double)v[i][j] / inv) : v[i][j];
        o[j] = (DST)y;
      }
      d[c] = o;
    }
  }
}

template <typename SRC>
static int prep_dispatch_dst(const void* src, int64_t n, int dim, void* dst, int ld, int storage_dtype,
                             int normalize, hipStream_t stream) {
  if (n == 0) return MRAG_OK;
  dim3 grid((unsigned)((n + 3) / 4)), block(256);
  if (std::is_same<SRC, float>::value && dim % 8 == 0 && ld % 8 == 0 && ld <= 4096 && ((uintptr_t)src & 15) == 0 &&
      (storage_dtype == MRAG_F16 || storage_dtype == MRAG_BF16)) {
    if (storage_dtype == MRAG_F16)
      hipLaunchKernelGGL((prep_rows_vec_kernel<_Float16>), grid, block, 0, stream, (const float*)src, n, dim, (_Float16*)dst, ld, normalize);
    else
      hipLaunchKernelGGL((prep_rows_vec_kernel<__bf16>), grid, block, 0, stream, (const float*)src, n, dim, (__bf16*)dst, ld, normalize);
    MRAG_HIP(hipGetLastError());
    return MRAG_OK;
  }
  if (storage_dtype == MRAG_F16) {
    hipLaunchKernelGGL((prep_rows_kernel<SRC, _Float16>), grid, block, 0, stream, (const SRC*)src, n, dim,
                       (_Float16*)dst, ld, normalize);
  } else if (storage_dtype == MRAG_BF16) {
    hipLaunchKernelGGL((prep_rows_kernel<SRC, __bf16>), grid, block, 0, stream, (const SRC*)src, n, dim,
                       (__bf16*)dst, ld, normalize);
  } else {
    return fail(MRAG_ERR_INVALID, "storage dtype %d not supported (fp16/bf16 only)", storage_dtype);
  }
  MRAG_HIP(hipGetLastError());
  return MRAG_OK;
}

int launch_prep_rows(const void* src, int src_dtype, int64_t n, int dim, void* dst, int ld, int storage_dtype,
                     int normalize, hipStream_t stream) {
  switch (src_dtype) {
    case MRAG_F32: return prep_dispatch_dst<float>(src, n, dim, dst, ld, storage_dtype, normalize, stream);
    case MRAG_F16: return prep_dispatch_dst<_Float16>(src, n, dim, dst, ld, storage_dtype, normalize, stream);
    case MRAG_BF16: return prep_dispatch_dst<__bf16>(src, n, dim, dst, ld, storage_dtype, normalize, stream);
    case MRAG_F64: return prep_dispatch_dst<double>(src, n, dim, dst, ld, storage_dtype, normalize, stream);
    default: return fail(MRAG_ERR_INVALID, "unknown source dtype %d", src_dtype);
  }
}

}  // namespace mrag

extern "C" {

int mrag_abi_version(void) { return MRAG_ABI_VERSION; }

const char* mrag_last_error(void) { return mrag::g_err; }

int mrag_device_count(int* out_count) {
  if (!out_count) return mrag::fail(MRAG_ERR_INVALID, "out_count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *out_count = n;
  return MRAG_OK;
}

}  // extern "C"
