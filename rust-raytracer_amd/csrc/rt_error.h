// rt_error.h -- per-thread last-error message behind rt_last_error().
#pragma once

#include <exception>
#include <new>
#include <string>

#include "../../include/rt_hip.h"

namespace rt {

inline std::string &last_error_slot() {
    static thread_local std::string msg;
    return msg;
}

// Records `msg` and returns `code`, so call sites read `return set_error(...)`.
inline int set_error(int code, const std::string &msg) {
    last_error_slot() = msg;
    return code;
}

// The body of a C entry point: no exception leaves the library (a host allocation that fails, a container
// asked for an absurd size), it becomes an error code and a message like every other failure.
template <class Body>
int guarded(Body &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return set_error(RT_ERR_DEVICE, "out of host memory");
    } catch (const std::exception &e) {
        return set_error(RT_ERR_DEVICE, std::string("internal error: ") + e.what());
    } catch (...) {
        return set_error(RT_ERR_DEVICE, "internal error");
    }
}

}  // namespace rt
