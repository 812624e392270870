// Gradient all-reduce over RCCL / xGMI for the data-parallel training step (reference
// basicsr/models/base_model.py:108-115 -- torch DistributedDataParallel's bucketed all-reduce(SUM) / world -- and :448, the
// loss reduce).  One thin entry point on a CALLER-PROVIDED communicator: the library neither creates communicators nor
// bootstraps ranks (that is the launcher's / torch.distributed's job), it only enqueues `ncclAllReduce(SUM)` in place on the
// caller's stream followed by the 1/world scaling, so that a host that does not use torch's DDP can reduce a flat gradient
// buffer (or a DDP communication hook can hand its bucket) through the same C ABI as the compute kernels.
//
// RCCL is resolved at first use from the process image: the copy that is already loaded (torch bundles one:
// torch/lib/librccl.so, SONAME librccl.so.1) or, failing that, the system's librccl.so.1 -- the library itself does not link
// against RCCL, so single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "dcpt_common.h"
#include "../../include/dcpt_hip.h"

namespace {
typedef ncclResult_t (*allreduce_fn)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef const char* (*errstr_fn)(ncclResult_t);
std::mutex g_mu;
allreduce_fn g_allreduce = nullptr;
errstr_fn g_errstr = nullptr;
bool g_tried = false;

bool resolve() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_tried) return g_allreduce != nullptr;
    g_tried = true;
    void* h = nullptr;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!sym) {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (int pass = 0; pass < 2 && !sym; ++pass)   // pass 0: only a copy that is already mapped (RTLD_NOLOAD)
            for (const char* n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h && (sym = dlsym(h, "ncclAllReduce"))) break;
            }
    }
    g_allreduce = (allreduce_fn)sym;
    g_errstr = (errstr_fn)(h ? dlsym(h, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return g_allreduce != nullptr;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ buf, size_t n4, size_t n, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 v = ldg4(buf + 4 * i);
        stg4(buf + 4 * i, f4_scale(v, scale));
    }
    if (i == 0)
        for (size_t t = 4 * n4; t < n; ++t) buf[t] *= scale;   // ragged tail (< 4 elements)
}
}  // namespace

extern "C" int dcpt_allreduce_flat(float* buf, size_t n, void* rccl_comm, float scale, dcpt_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DCPT_CHECK_ARG(buf && rccl_comm, "allreduce_flat: null buffer or communicator");
    DCPT_CHECK_ARG(((uintptr_t)buf & 15) == 0, "allreduce_flat: buffer must be 16-byte aligned");
    if (n == 0) return DCPT_OK;
    if (!resolve()) {
        dcpt_set_error("allreduce_flat: RCCL (librccl.so.1) is not loadable in this process: %s", dlerror());
        return DCPT_ERR_HIP;
    }
    const ncclResult_t r = g_allreduce(buf, buf, n, ncclFloat32, ncclSum, (ncclComm_t)rccl_comm, s);
    if (r != ncclSuccess) {
        dcpt_set_error("allreduce_flat: ncclAllReduce failed: %s", g_errstr ? g_errstr(r) : "?");
        return DCPT_ERR_HIP;
    }
    if (scale != 1.0f) {
        const size_t n4 = n / 4;
        scale_kernel<<<dim3((unsigned)((n4 > 0 ? n4 : 1) + 255) / 256), dim3(256), 0, s>>>(buf, n4, n, scale);
        DCPT_CHECK_LAUNCH("allreduce_flat scale");
    }
    return DCPT_OK;
}
