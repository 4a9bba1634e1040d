// Entry points declared in include/mrag.h whose kernels are not written yet in this tree
// state: they fail loudly with MRAG_ERR_UNSUPPORTED (never a CPU fallback).  Each one is
// removed from this file when its implementation (ivf.hip / encoder.hip) lands.
#include "common.h"
using namespace mrag;
#define PENDING(name) return fail(MRAG_ERR_UNSUPPORTED, #name " is not implemented in this build")
extern "C" {
#ifndef MRAG_HAVE_IVF
int mrag_ivf_create(int, int, int, int, int, mrag_handle*) { PENDING(mrag_ivf_create); }
int mrag_ivf_destroy(mrag_handle) { PENDING(mrag_ivf_destroy); }
int mrag_ivf_train(mrag_handle, const void*, int64_t, int, int, int, int, uint64_t, void*) { PENDING(mrag_ivf_train); }
int mrag_ivf_set_centroids(mrag_handle, const void*, int, int, int, void*) { PENDING(mrag_ivf_set_centroids); }
int mrag_ivf_get_centroids(mrag_handle, float*, int, void*) { PENDING(mrag_ivf_get_centroids); }
int mrag_ivf_add(mrag_handle, const void*, int64_t, int, int, int, void*) { PENDING(mrag_ivf_add); }
int mrag_ivf_size(mrag_handle, int64_t*) { PENDING(mrag_ivf_size); }
int mrag_ivf_set_id_base(mrag_handle, int64_t) { PENDING(mrag_ivf_set_id_base); }
int mrag_ivf_get_assignments(mrag_handle, int32_t*, int, void*) { PENDING(mrag_ivf_get_assignments); }
int mrag_ivf_search(mrag_handle, const void*, int64_t, int, int, int, int, int, float*, int64_t*, int, void*) { PENDING(mrag_ivf_search); }
#endif
#ifndef MRAG_HAVE_ENCODER
int mrag_encoder_create(const mrag_encoder_config*, int, mrag_handle*) { PENDING(mrag_encoder_create); }
int mrag_encoder_destroy(mrag_handle) { PENDING(mrag_encoder_destroy); }
int mrag_encoder_set_param(mrag_handle, const char*, const float*, int64_t, int, void*) { PENDING(mrag_encoder_set_param); }
int mrag_encoder_missing_params(mrag_handle, int*) { PENDING(mrag_encoder_missing_params); }
int mrag_encoder_forward(mrag_handle, const int32_t*, const int32_t*, int, int, float*, int, int, int, void*) { PENDING(mrag_encoder_forward); }
#endif
}
