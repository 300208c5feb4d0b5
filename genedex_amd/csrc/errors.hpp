// errors.hpp -- the error type of libgdx.so's host code (no HIP: also compiled into the sanitised CPU checks of
// tests/host_checks.cpp)
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/gdx.h"

namespace gdx {

// error carried through the host code and turned into a gdx_status at the C ABI
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

[[noreturn]] inline void fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(status, buf);
}

inline uint64_t div_ceil_u64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

}  // namespace gdx
