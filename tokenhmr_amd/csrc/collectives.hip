// RCCL collectives of the data-parallel path behind the C ABI, for hosts that do not have torch.distributed.
//
// The reference has no collective on this path (inference is single-device, tokenhmr/eval.py:52-54); crops are independent, so
// the path shards with no data-path collective.  Two collectives surround it: ONE broadcast of the packed weight arena at
// start-up and ONE all-gather per batch of the packed per-crop records (85,128 B / crop).  tokenhmr_amd/dist.py issues them
// through torch.distributed (backend "nccl" == RCCL); these entry points issue the same two calls on an ncclComm_t the caller
// created.  RCCL is NOT linked: its symbols are resolved at first use from the librccl the process already has (dlopen by
// soname returns the loaded copy — e.g. the one bundled with PyTorch), so the communicator and the calls belong to one RCCL.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "../../include/tokenhmr_hip.h"
#include "common.h"

namespace {

typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*errstr_fn)(int);

struct Rccl {
    bcast_fn bcast = nullptr;
    allgather_fn allgather = nullptr;
    errstr_fn errstr = nullptr;
    std::string err;
    bool tried = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!r.tried) {
        r.tried = true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);         // the copy the process already uses, if any
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { r.err = std::string("librccl not found: ") + dlerror(); return r; }
        r.bcast = reinterpret_cast<bcast_fn>(dlsym(h, "ncclBroadcast"));
        r.allgather = reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather"));
        r.errstr = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        if (!r.bcast || !r.allgather) r.err = "librccl lacks ncclBroadcast / ncclAllGather";
    }
    return r;
}

constexpr int NCCL_UINT8 = 1, NCCL_FLOAT32 = 7;          // ncclDataType_t (rccl.h)

// one thread per float4-aligned word group would need per-field alignment; records are 21282 words, fields are not 16-byte
// aligned within a record, so the pack kernel moves single words: 5.4 MB at 64 crops, a few microseconds
struct PackSrc {
    const float* p[8];
    int n[8];           // words per crop of each field
};
__global__ __launch_bounds__(256) void pack_records_kernel(PackSrc s, float* __restrict__ rec, int B) {
    const int b = blockIdx.y;
    float* out = rec + (int64_t)b * THMR_RECORD_WORDS;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < THMR_RECORD_WORDS; w += gridDim.x * 256) {
        int f = 0, off = w;
#pragma unroll
        for (int i = 0; i < 7; ++i)
            if (f == i && off >= s.n[i]) { off -= s.n[i]; f = i + 1; }
        out[w] = s.p[f][(int64_t)b * s.n[f] + off];
    }
}

thread_local std::string g_cerr;
int cfail(int code, const std::string& m) { g_cerr = m; return code; }

}  // namespace

extern "C" {

const char* thmr_collective_last_error(void) { return g_cerr.c_str(); }

int thmr_pack_records(const thmr_outputs* o, int32_t B, float* rec_dev, void* stream) {
    if (!o || !rec_dev || B < 1) return cfail(THMR_ERR_INVALID, "bad argument");
    if (!o->pred_vertices || !o->pred_keypoints_3d || !o->pred_keypoints_2d || !o->rotmat || !o->betas || !o->pred_cam ||
        !o->pred_cam_t || !o->token_idx)
        return cfail(THMR_ERR_INVALID, "thmr_pack_records needs pred_vertices, pred_keypoints_3d/2d, rotmat, betas, pred_cam, pred_cam_t and token_idx");
    PackSrc s;
    const float* p[8] = {o->pred_vertices, o->pred_keypoints_3d, o->pred_keypoints_2d, o->rotmat, o->betas, o->pred_cam,
                         o->pred_cam_t, reinterpret_cast<const float*>(o->token_idx)};      // indices travel bit-exactly as words
    const int n[8] = {6890 * 3, 44 * 3, 44 * 2, 24 * 9, 10, 3, 3, 160};
    int tot = 0;
    for (int i = 0; i < 8; ++i) { s.p[i] = p[i]; s.n[i] = n[i]; tot += n[i]; }
    static_assert(6890 * 3 + 44 * 3 + 44 * 2 + 24 * 9 + 10 + 3 + 3 + 160 == THMR_RECORD_WORDS, "record layout");
    (void)tot;
    hipLaunchKernelGGL(pack_records_kernel, dim3(21, B), dim3(256), 0, static_cast<hipStream_t>(stream), s, rec_dev, B);
    return hipGetLastError() == hipSuccess ? 0 : cfail(THMR_ERR_HIP, "pack_records launch failed");
}

int thmr_bcast_weights(thmr_engine* e, void* nccl_comm, int32_t root, void* stream) {
    if (!e || !nccl_comm) return cfail(THMR_ERR_INVALID, "null engine / communicator");
    Rccl& r = rccl();
    if (!r.bcast) return cfail(THMR_ERR_STATE, r.err);
    void* ptr = nullptr;
    size_t bytes = 0;
    if (int rc = thmr_weight_arena(e, &ptr, &bytes)) return rc;
    const int rc = r.bcast(ptr, ptr, bytes, NCCL_UINT8, root, nccl_comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return cfail(THMR_ERR_HIP, std::string("ncclBroadcast: ") + (r.errstr ? r.errstr(rc) : "error"));
    return 0;
}

int thmr_allgather_records(void* nccl_comm, const float* rec_dev, int32_t rows, float* recv_dev, void* stream) {
    if (!nccl_comm || !rec_dev || !recv_dev || rows < 0) return cfail(THMR_ERR_INVALID, "bad argument");
    Rccl& r = rccl();
    if (!r.allgather) return cfail(THMR_ERR_STATE, r.err);
    const int rc = r.allgather(rec_dev, recv_dev, (size_t)rows * THMR_RECORD_WORDS, NCCL_FLOAT32, nccl_comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return cfail(THMR_ERR_HIP, std::string("ncclAllGather: ") + (r.errstr ? r.errstr(rc) : "error"));
    return 0;
}

}  // extern "C"
