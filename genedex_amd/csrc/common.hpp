// common.hpp -- host-side helpers shared by all translation units of libgdx.so
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/gdx.h"
#include "errors.hpp"

namespace gdx {

#define GDX_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            ::gdx::fail(GDX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

// RAII device allocation
template <class T>
struct DeviceBuffer {
    T *ptr = nullptr;
    size_t count = 0;
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { alloc(n); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    DeviceBuffer(DeviceBuffer &&o) noexcept : ptr(o.ptr), count(o.count)
    {
        o.ptr = nullptr;
        o.count = 0;
    }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept
    {
        if (this != &o) {
            release();
            ptr = o.ptr;
            count = o.count;
            o.ptr = nullptr;
            o.count = 0;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void alloc(size_t n)
    {
        release();
        count = n;
        if (n) GDX_HIP(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    size_t bytes() const { return count * sizeof(T); }
    T *get() const { return ptr; }
};

inline uint64_t div_ceil(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Grow-only scratch memory for the `_dev` entry points, one arena per (calling thread, stream, slot).
// Work on one stream is ordered, so a later call may reuse the bytes of an earlier one without waiting;
// hipMallocAsync / hipFreeAsync per call cost ~0.1 ms each on this runtime.  Growing an arena waits for the
// stream once.  Arenas live until the thread exits.
void *stream_scratch(hipStream_t stream, int slot, size_t bytes);

// grid size for a grid-stride kernel: enough blocks to fill 256 CUs several times over
inline unsigned grid_for(uint64_t work_items, unsigned block, unsigned max_blocks = 256u * 16u)
{
    uint64_t g = div_ceil(work_items ? work_items : 1, block);
    return static_cast<unsigned>(g < max_blocks ? g : max_blocks);
}

}  // namespace gdx
