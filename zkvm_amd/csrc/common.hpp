// Shared host-side plumbing of the library: error codes, HIP call checking, a grow-only device buffer.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/zkgpu.h"
#include <cstdio>
#include <cstdint>
#include <string>
#include <stdexcept>

namespace zk {

// Error codes returned across the C boundary: the ZKGPU_ERR_* macros of include/zkgpu.h.

struct ZkError : std::runtime_error {
    int code;
    ZkError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define ZK_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            throw ::zk::ZkError(e_ == hipErrorOutOfMemory ? ZKGPU_ERR_OOM : ZKGPU_ERR_HIP,       \
                                std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    } while (0)

#define ZK_REQUIRE(cond, code, msg)                       \
    do {                                                  \
        if (!(cond)) throw ::zk::ZkError((code), (msg));  \
    } while (0)

// Device allocation that only ever grows; freed with the owning context.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) ZK_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        ZK_HIP(hipMalloc(&p, bytes));
        cap = bytes;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace zk
