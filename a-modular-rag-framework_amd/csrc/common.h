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

enum HandleKind : uint32_t { KIND_BF_INDEX = 1, KIND_IVF = 2, KIND_ENCODER = 3, KIND_BM25 = 4 };

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


// One fused similarity + top-k pass (K2) and the candidate merge (K4) over PREPARED operands
// (storage dtype, padded to `ld`, query buffer padded to a multiple of 256 rows).
struct BfLaunch {
  const uint16_t* corpus = nullptr;   // [rows padded to 256][ld]
  const uint16_t* queries = nullptr;  // dense: [ceil(nq/256)*256][ld]; descriptor mode: gathered [n_wg*256][ld]
  int ld = 0, dtype = MRAG_F16, k = 0;
  int64_t nq = 0;                     // queries reported (rows of out_scores / out_ids)
  int64_t n_rows = 0;                 // dense mode: valid corpus rows
  // descriptor mode (IVF list scan)
  const int* wg_desc = nullptr;       // device int[n_wg][8]
  int n_wg = 0;
  const void* pair_loc = nullptr;     // device int2[nq][nprobe]
  int nprobe = 0;
  const int64_t* row_ids = nullptr;   // device: stored position -> original row
  int64_t id_base = 0;
  float* out_scores = nullptr;        // device [nq][k]
  int64_t* out_ids = nullptr;         // device [nq][k]
  DevBuf* lists = nullptr;            // workspace (grown as needed)
  DevBuf* counts = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev_k2_begin = nullptr, ev_k2_end = nullptr;   // optional
  long long* clock = nullptr;         // optional device long long[4]: batch kernel workgroup 0's clock stamps
};
int bf_launch(const BfLaunch& a);
int bf_max_k();
int bf_max_k_wide();
int64_t bf_round_rows(int64_t n);     // rows rounded up to the kernel's tile (256)

// "Score segments + select" regime of the IVF search (ivf_scan.hip).  One scan workgroup = one descriptor:
//   [0] gq base: slot s of the workgroup scores query row gq[base + s] (< 0: none); or -1 - first for the
//       identity map (queries first .. first + 127)          [1] queries (<= 128)
//   [2] first corpus row            [3] rows to score          [4] index of the first row inside the segment
//   [5],[6] float offset of the workgroup's score block S[slot][pitch] (low, high)        [7] pitch (floats, % 32 == 0)
constexpr int IVFS_DESC_WORDS = 8;
constexpr int IVFS_QUERIES = 128;       // queries per scan workgroup
#ifndef MRAG_IVFS_CHUNK_ROWS
#define MRAG_IVFS_CHUNK_ROWS 128
#endif
constexpr int IVFS_CHUNK_ROWS = MRAG_IVFS_CHUNK_ROWS;   // rows of a list one scan descriptor covers = one tile of the scan kernel (longer lists are cut: no single-workgroup tails)
constexpr int IVFS_PITCH_ALIGN = 32;    // floats: every query's score segment starts on a 128-byte line
static inline int64_t ivfs_pitch(int64_t rows) { return (rows + IVFS_PITCH_ALIGN - 1) / IVFS_PITCH_ALIGN * IVFS_PITCH_ALIGN; }
constexpr int IVFS_DENSE_ROWS = 128;    // corpus rows per descriptor of the dense (probe selection) case: one tile, so that the persistent grid balances
// (n_desc_dev != nullptr: the descriptor count lives on the device and n_desc is its host-side upper bound = the grid)
int ivfs_scan(const uint16_t* corpus, const uint16_t* queries, int ld, int dtype, int64_t zero_row, const int* desc, int n_desc,
              const int* n_desc_dev, const int64_t* gq, float* S, hipStream_t stream, long long* dbg = nullptr);
// pinfo: int4 per (query, probe) = (float offset of the pair's score segment: low, high; its rows; its first stored row)
int ivfs_select_lists(const float* S, const void* pinfo, int nprobe, int64_t nq, int k, const int64_t* row_ids,
                      int64_t id_base, float* out_scores, int64_t* out_ids, hipStream_t stream);
int ivfs_dense_n_desc(int64_t nq, int n_rows);
// count_out != nullptr: every selected row r with row_weight[r] > 0 also bumps count_out[r] (the IVF search's
// "queries per list" histogram, otherwise a kernel of its own)
int ivfs_dense_topk(const uint16_t* corpus, int n_rows, const uint16_t* queries, int64_t nq, int ld, int dtype, int k, int* desc,
                    float* S, float* out_scores, int64_t* out_ids, hipStream_t stream, int* count_out = nullptr,
                    const int* row_weight = nullptr);

}  // namespace mrag
