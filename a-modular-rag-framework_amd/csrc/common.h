// Internal helpers shared by the libmrag_hip translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/mrag.h"

namespace mrag {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define MRAG_HIP(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      return ::mrag::fail(_e == hipErrorOutOfMemory ? MRAG_ERR_OOM : MRAG_ERR_HIP,              \
                          "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,      \
                          __LINE__);                                                            \
    }                                                                                           \
  } while (0)

#define MRAG_TRY(expr)            \
  do {                            \
    int _s = (expr);              \
    if (_s != MRAG_OK) return _s; \
  } while (0)

// verifies a usable device and makes it current
int use_device(int device);

enum HandleKind : uint32_t { KIND_BF_INDEX = 1, KIND_IVF = 2, KIND_ENCODER = 3 };

struct Object {
  HandleKind kind;
  int device;
  virtual ~Object() {}
};

mrag_handle register_object(Object* o);
Object* lookup(mrag_handle h, HandleKind kind);  // nullptr + error set when invalid
Object* take(mrag_handle h, HandleKind kind);    // removes from the registry

// growable device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need);  // grows (contents NOT preserved); returns status
  void release();
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// rows of any supported dtype -> normalised/rounded storage rows (fp16/bf16), zero padded to `ld`
// src/dst are device pointers.  dst row r lives at dst + r*ld (16-bit elements).
int launch_prep_rows(const void* src, int src_dtype, int64_t n, int dim, void* dst, int ld,
                     int storage_dtype, int normalize, hipStream_t stream);

}  // namespace mrag
